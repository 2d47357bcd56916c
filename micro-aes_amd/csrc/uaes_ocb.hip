/*
 * uaes_ocb.hip -- OCB (RFC 7253), the remaining block-parallel AEAD of the
 * reference.
 *
 *   ocb_setup_lds  <- getSubkeys :593-604 (L_*, L_$, L_0 ...), the nonce part of
 *                     OCB_cipher :1705-1719 (K_top, stretch, Offset_0)
 *   ocb_main_body  <- the block loop of OCB_cipher :1721-1730 with getDelta :1662-1680
 *   ocb_final_body <- the partial block :1736-1741, the tag :1743-1744 and the
 *                     PMAC of the associated data :1746-1760
 * ONE launch per call (k_ocb; k_ocb_small for short messages): every workgroup derives the L table and
 * Offset_0 for itself (two block encryptions side by side + one shift per table row, ~1 us), and the
 * workgroup that finishes last computes the tag.
 *
 * The reference walks the blocks one by one and recomputes Offset_i from scratch
 * for every i (getDelta doubles L up to 64 times per block).  Here the offsets are
 * never chained:  Offset_i = Offset_0 ^ XOR{ L_j : bit j of gray(i) },  gray(i) =
 * i ^ (i >> 1).  For i = 256 c + l the low eight bits of gray(i) depend on the lane
 * position l only (plus bit 0 of c, which toggles L_7), the rest on the chunk
 * number c, so a lane holds four constant 16-byte masks and a wave carries one
 * uniform mask per 256-block chunk that moves to the next chunk with two XORs
 * (gray(c+1) ^ gray(c) = 1 << ntz(c+1)).  The checksum is XOR-reduced per lane,
 * per wave, then with four 32-bit atomics per wave.
 *
 * Block i (1-based, as in the RFC) of the text sits at byte offset 16 (i - 1).
 */
#include <hip/hip_runtime.h>
#include <string.h>
#include <type_traits>
#include "uaes_aes.hip.h"
#include "uaes_device.h"
#include "uaes_plan.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

#define OCB_NL       64u                      /* table rows: 0 L_*, 1 L_$, 2+j L_j      */
#define OCB_LDS_L    (128u * 1024u)           /* L table in LDS, after the cipher tables */
#define OCB_LDS_ACC  (OCB_LDS_L + OCB_NL * 16u)
#define OCB_LDS_OFF0 (OCB_LDS_ACC + 32u)        /* behind the two 16-byte accumulators: Offset_0 */
#define OCB_LDS_FLAG (OCB_LDS_OFF0 + 16u)       /* "this workgroup finishes the call" */
#define OCB_LDS_TE0  (OCB_LDS_FLAG + 16u)        /* decryption: 1 KiB, plain Te0 (EncPlain) */
#define OCB_LDS      (OCB_LDS_TE0 + 1024u)
#define OCB_CHUNK    256u                     /* blocks per chunk: one wave x 4 per lane */
#define OCB_RUN_MAX  16u                      /* most consecutive chunks a wave takes at once */
#define UAES_U       4                        /* blocks per lane per chunk               */
/* scratch rows (uint4): [65..] one checksum share per workgroup of k_ocb (plain stores: thousands of
 * same-address atomics cost 0.2 ms); rows [0..64] are the hand-over rows of k_ocb_small (L table, Offset_0) */
#define OCB_ROW_OFF0 64u
#define OCB_ROW_PART 65u
#define OCB_MAX_WGS  1024u
#define OCB_ROW_HASH (OCB_ROW_PART + OCB_MAX_WGS)  /* ... and one share of HASH(K, A) per workgroup when the associated data is long */
#define OCB_HASH_SPREAD 8192u                   /* whole blocks of associated data from which every workgroup hashes a part (below: the finishing workgroup alone) */
#define OCB_PLAIN_AAD (16u * 1024u)            /* decryption: associated data up to here is hashed through the plain Te0 */

struct B16 {
    u32 w[4];
};

__device__ __forceinline__ B16 b16(uint4 v) { B16 b = { { v.x, v.y, v.z, v.w } }; return b; }
__device__ __forceinline__ uint4 u4(const B16 &b) { return make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]); }
__device__ __forceinline__ void bx(B16 &a, const B16 &b)
{
    a.w[0] ^= b.w[0]; a.w[1] ^= b.w[1]; a.w[2] ^= b.w[2]; a.w[3] ^= b.w[3];
}

__device__ __forceinline__ B16 lds_row(u32 row)
{
    return b16(*(const uint4 *)(uaes_lds + OCB_LDS_L + 16u * row));
}

/* a wave-uniform block, moved to scalar registers */
__device__ __forceinline__ B16 uniform(const B16 &b)
{
    B16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r.w[q] = (u32)__builtin_amdgcn_readfirstlane((int)b.w[q]);
    return r;
}

/* doubleBblock (:434-443): the block as a 128-bit big-endian integer, << 1, carry -> ^0x87 */
__device__ __forceinline__ B16 ocb_double(const B16 &b)
{
    u64 hi = ((u64)bswap32(b.w[0]) << 32) | bswap32(b.w[1]);
    u64 lo = ((u64)bswap32(b.w[2]) << 32) | bswap32(b.w[3]);
    const u64 carry = hi >> 63;
    hi = (hi << 1) | (lo >> 63);
    lo = (lo << 1) ^ (carry ? 0x87ull : 0ull);
    B16 r;
    r.w[0] = bswap32((u32)(hi >> 32)); r.w[1] = bswap32((u32)hi);
    r.w[2] = bswap32((u32)(lo >> 32)); r.w[3] = bswap32((u32)lo);
    return r;
}

/* One block encryption, two ways.  EncRep: the replicated tables of the encryption direction are in LDS.
 * EncPlain: a 1 KiB copy of Te0 at OCB_LDS_TE0 -- what a DECRYPTING kernel uses for the handful of encryptions OCB
 * needs whatever the direction (L_*, K_top, the pad of a ragged block, the tag, short associated data), instead of
 * swapping 128 KiB of tables in and out around them.                                                        */
template <int NR>
struct EncRep {
    const uaesk_rk &rk;
    const LaneConst &lc;
    __device__ __forceinline__ void operator()(B16 &b) const
    {
        u32 s[1][4] = { { b.w[0], b.w[1], b.w[2], b.w[3] } };
        enc_blocks<NR, 1>(s, rk, lc);
        b.w[0] = s[0][0]; b.w[1] = s[0][1]; b.w[2] = s[0][2]; b.w[3] = s[0][3];
    }
};

template <int NR>
struct EncPlain {
    const uaesk_rk &rk;
    __device__ __forceinline__ void operator()(B16 &b) const
    {
        plain_encrypt<NR>((const u32 *)(uaes_lds + OCB_LDS_TE0), rk, b.w);
    }
};

__device__ __forceinline__ void ocb_fill_plain_te0(const u32 *__restrict__ te0)      /* a barrier must follow */
{
    if (threadIdx.x < 256u) ((u32 *)(uaes_lds + OCB_LDS_TE0))[threadIdx.x] = te0[threadIdx.x];
}

/* XOR of the L_j selected by the bits of g, j counted from `first` (LDS table) */
__device__ __forceinline__ B16 ocb_gray_sum(u64 g, u32 first)
{
    B16 d = { { 0, 0, 0, 0 } };
    while (g) {
        const u32 j = (u32)__builtin_ctzll(g);
        g &= g - 1;
        bx(d, lds_row(2u + first + j));
    }
    return d;
}

/* Offset_i - Offset_0 for an arbitrary index (the tails; one lane) */
__device__ __forceinline__ B16 ocb_delta(u64 i) { return ocb_gray_sum(i ^ (i >> 1), 0); }

/* the two checksum accumulators of a workgroup */
__device__ __forceinline__ void ocb_clear_acc()
{
    if (threadIdx.x < 2) *(uint4 *)(uaes_lds + OCB_LDS_ACC + 16u * threadIdx.x) = make_uint4(0, 0, 0, 0);
}

__device__ __forceinline__ void wave_xor_reduce(u32 (&v)[4])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] ^= __shfl_xor(v[k], off, 64);
    }
}

/* ------------------------------------------------------------------------ */
/* per-call setup: L table, Offset_0, cleared checksum                         */
/* ------------------------------------------------------------------------ */
/* v * x^r (0 <= r < 64) in GF(2^128) as OCB numbers its bits (the block is a big-endian integer, doubleBblock
 * :434-443 is r = 1): the r bits shifted out come back multiplied by x^7 + x^2 + x + 1, which stays below x^71 */
__device__ __forceinline__ B16 ocb_times_x_pow(const B16 &b, u32 r)
{
    u64 hi = ((u64)bswap32(b.w[0]) << 32) | bswap32(b.w[1]);
    u64 lo = ((u64)bswap32(b.w[2]) << 32) | bswap32(b.w[3]);
    if (r) {
        const u64 o = hi >> (64u - r);
        hi = (hi << r) | (lo >> (64u - r));
        lo <<= r;
        lo ^= o ^ (o << 1) ^ (o << 2) ^ (o << 7);
        hi ^= (o >> 63) ^ (o >> 62) ^ (o >> 57);
    }
    B16 v;
    v.w[0] = bswap32((u32)(hi >> 32)); v.w[1] = bswap32((u32)hi);
    v.w[2] = bswap32((u32)(lo >> 32)); v.w[3] = bswap32((u32)lo);
    return v;
}

/* Call with the whole first wave, what `enc` needs in LDS; a barrier must follow.
 * Lane 0 encrypts the zero block (L_*), lane 1 the nonce block (K_top) -- a lane alone walks its rounds at the
 * pace of a lone wave whatever the other lanes do, so the second block is free.  Row r of the table is
 * L_* x^r (row 0 L_*, 1 L_$, 2 + j L_j; getSubkeys doubles its way up, :593-604): every lane shifts its own row
 * out of L_* in one step.  Offset_0 = bits bottom .. bottom + 127 of K_top || (K_top ^ K_top << 8)[0..63].  */
template <typename ENC>
__device__ __forceinline__ void ocb_setup_lds(const ENC &enc, uint4 nonce_block, u32 bottom)
{
    B16 b = { { 0, 0, 0, 0 } };
    if (threadIdx.x == 1) b = b16(nonce_block);
    enc(b);
    B16 ls;
#pragma unroll
    for (int q = 0; q < 4; ++q) ls.w[q] = (u32)__builtin_amdgcn_readfirstlane((int)b.w[q]);
    ((uint4 *)(uaes_lds + OCB_LDS_L))[threadIdx.x] = u4(ocb_times_x_pow(ls, threadIdx.x));
    if (threadIdx.x == 1) {
        const u64 hi = ((u64)bswap32(b.w[0]) << 32) | bswap32(b.w[1]);
        const u64 lo = ((u64)bswap32(b.w[2]) << 32) | bswap32(b.w[3]);
        const u64 ext = hi ^ ((hi << 8) | (lo >> 56));    /* Stretch = K_top || ext        */
        const u64 ohi = bottom ? (hi << bottom) | (lo >> (64u - bottom)) : hi;
        const u64 olo = bottom ? (lo << bottom) | (ext >> (64u - bottom)) : lo;
        *(uint4 *)(uaes_lds + OCB_LDS_OFF0) = make_uint4(bswap32((u32)(ohi >> 32)), bswap32((u32)ohi),
                                                         bswap32((u32)(olo >> 32)), bswap32((u32)olo));
    }
    ocb_clear_acc();
}

/* ------------------------------------------------------------------------ */
/* the block loop                                                              */
/* ------------------------------------------------------------------------ */
/* the caller has put the L table and Offset_0 (ocb_setup_lds) and the cipher tables of the direction into LDS */
/* HASH: the same walk over the ASSOCIATED DATA (HASH(K, A), :1749-1754: Sum ^= Enc(A_i ^ Offset_i) with offsets
 * counted from zero): nothing is written, the sum collects the cipher's outputs.  share = the workgroup's row.    */
template <int NR, bool DEC, bool HASH = false>
__device__ __forceinline__ void ocb_main_body(const uaesk_rk &rk, const LaneConst &lc,
                                              uint4 *__restrict__ share, u64 nblocks, u32 run,
                                              const uint4 *in, uint4 *out)
{
    /* run (a power of two <= OCB_RUN_MAX) = consecutive chunks per wave; short texts use
     * short runs so that every CU gets work                                        */
    const u32 lane = threadIdx.x & 63u;
    /* launched with 16 waves per workgroup, or 4 for short texts (more CUs, see launch_ocb) */
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);
    const u64 nchunks = (nblocks >> 8) + 1;           /* indices 0..nblocks, index 0 unused */

    /* Offset_i - Offset_0 for i = 256 c + 64 u + lane splits into a lane part (bits 0..5 of
     * gray(lane), bit 5 taken as lane bit 5), a part that depends on u only (L_5..L_7) and
     * the chunk part; the last two are wave-uniform and live in scalar registers.        */
    const B16 lm = ocb_gray_sum(lane ^ (lane >> 1), 0);
    const B16 zero16 = { { 0, 0, 0, 0 } };
    const B16 off0 = HASH ? zero16 : uniform(b16(*(const uint4 *)(uaes_lds + OCB_LDS_OFF0)));
    const B16 l5 = uniform(lds_row(2u + 5u)), l6 = uniform(lds_row(2u + 6u)), l7 = uniform(lds_row(2u + 7u));

    u32 sum[4] = { 0, 0, 0, 0 };
    /* the chunk mask of chunk c from scratch: the set bits of gray(c) from bit 8 on, and L_7 for bit 0 of c */
    auto chunk_mask = [&](u64 c) {
        B16 m = off0;
        bx(m, ocb_gray_sum(c ^ (c >> 1), 8));
        if (c & 1) bx(m, l7);
        return uniform(m);
    };
    /* one chunk, text in d[]: whiten, encrypt, whiten, store, checksum.  FULL: all 256 positions exist (every chunk
     * but the first, whose index 0 does not, and the last) -- no masks on the checksum, no exec-masked stores */
    auto chunk = [&](auto FULLT, u64 c, const B16 &cm, const uint4 (&d)[UAES_U]) {
        constexpr bool FULL = decltype(FULLT)::value;
        const u32 lo = c == 0 ? 1u : 0u;
        const u64 left = nblocks - c * OCB_CHUNK;
        const u32 hi = left < OCB_CHUNK - 1 ? (u32)left : OCB_CHUNK - 1;
        auto live = [&](int u) -> u32 {
            if (FULL) return 0xffffffffu;
            const u32 j = 64u * u + lane;
            return (j >= lo && j <= hi) ? 0xffffffffu : 0u;
        };
        B16 um[UAES_U];                               /* uniform masks: chunk ^ u part */
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            um[u] = cm;
            if (u & 1) bx(um[u], l5);
            if ((u ^ (u >> 1)) & 1) bx(um[u], l6);
            if (u >> 1) bx(um[u], l7);
        }
        u32 s[UAES_U][4];
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            s[u][0] = xor3(d[u].x, um[u].w[0], lm.w[0]); s[u][1] = xor3(d[u].y, um[u].w[1], lm.w[1]);
            s[u][2] = xor3(d[u].z, um[u].w[2], lm.w[2]); s[u][3] = xor3(d[u].w, um[u].w[3], lm.w[3]);
            if (!DEC && !HASH) {
                const u32 lv = live(u);
                sum[0] ^= d[u].x & lv; sum[1] ^= d[u].y & lv;
                sum[2] ^= d[u].z & lv; sum[3] ^= d[u].w & lv;
            }
        }
        if (DEC) {
            dec_blocks<NR, UAES_U>(s, rk, lc);
        } else {
            enc_blocks_skewed<NR>(s[0], s[1], rk, lc);
            enc_blocks_skewed<NR>(s[2], s[3], rk, lc);
        }
        unsigned char *obase = (unsigned char *)out + c * (OCB_CHUNK * 16u) - 16;
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) {
            const u32 lv = live(u);
            if (HASH) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] ^= s[u][q] & lv;
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s[u][q] = xor3(s[u][q], um[u].w[q], lm.w[q]);
            if (DEC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] ^= s[u][q] & lv;
            }
            if (FULL || lv) *(uint4 *)(obase + 16u * (64u * u + lane)) = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    };

    /* The first chunk (index 0 does not exist) and the last (it ends where the text ends) are the only ones that need
     * masks: they go LAST, to the two waves that are next in the round-robin of the runs, with clamped loads of their
     * own; the loop runs over whole chunks only -- in one expansion with the masks it was 3 % slower, as a second
     * expansion inside the loop it did not fit 128 registers, and neither did the edge chunks in front of the loop with
     * the loop's first text already requested.                                                                      */
    const u64 nfull = nchunks > 2 ? nchunks - 2 : 0;  /* whole chunks: 1 .. nchunks - 2 */
    /* run is a power of two: shifts and masks, not the 64-bit divisions `k / run`, `k % run` compile to */
    const u32 lrun = (u32)__builtin_ctz(run), mrun = run - 1u;
    auto chunk_of = [&](u64 k) { return ((((k >> lrun) * nwaves + wave) << lrun)) + (k & mrun); };
    uint4 dn[UAES_U];
    auto fetch = [&](u64 f) {
        const unsigned char *base = (const unsigned char *)in + (f + 1) * (OCB_CHUNK * 16u) - 16;
#pragma unroll
        for (int u = 0; u < UAES_U; ++u) dn[u] = *(const uint4 *)(base + 16u * (64u * u + lane));
    };
    u64 k = 0, f = chunk_of(0);                       /* f counts the whole chunks: chunk c = f + 1 */
    if (f < nfull) fetch(f);
    {
        B16 cm = off0;                                /* chunk mask */
        while (f < nfull) {
            const u64 c = f + 1;
            if ((k & mrun) == 0) cm = chunk_mask(c);  /* first chunk of a run */
            uint4 d[UAES_U];
#pragma unroll
            for (int u = 0; u < UAES_U; ++u) d[u] = dn[u];
            const u64 fn = chunk_of(k + 1);
            if (fn < nfull) fetch(fn);
            chunk(std::true_type{}, c, cm, d);
            /* next chunk of the run: bit 0 of c flips (L_7), gray(c) gains or loses bit ntz(c+1) */
            ++k;
            bx(cm, l7);
            bx(cm, lds_row(2u + 8u + (u32)__builtin_ctzll(c + 1)));
            cm = uniform(cm);
            f = fn;
        }
    }
    {
        const u64 runs = (nfull + mrun) >> lrun;
        const u64 e0 = runs % nwaves, e1 = (runs + 1) % nwaves;          /* owners of the first / the last chunk */
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && nchunks < 2) break;
            if (wave != (pass ? e1 : e0)) continue;
            const u64 ce = pass ? nchunks - 1 : 0;
            const u32 lo = ce == 0 ? 1u : 0u;
            const u64 left = nblocks - ce * OCB_CHUNK;
            const u32 hi = left < OCB_CHUNK - 1 ? (u32)left : OCB_CHUNK - 1;
            const unsigned char *base = (const unsigned char *)in + ce * (OCB_CHUNK * 16u) - 16;
            uint4 d[UAES_U];
#pragma unroll
            for (int u = 0; u < UAES_U; ++u) {
                u32 j = 64u * u + lane;
                j = j < lo ? lo : (j > hi ? hi : j);          /* clamped: no branch before the load */
                d[u] = *(const uint4 *)(base + 16u * j);
            }
            chunk(std::false_type{}, ce, chunk_mask(ce), d);
        }
    }
    wave_xor_reduce(sum);
    if (lane == 0) {
        u32 *acc = (u32 *)(uaes_lds + OCB_LDS_ACC);
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicXor(acc + q, sum[q]);
    }
    __syncthreads();
    /* the share goes out as four device-scope atomic exchanges whose results have come back before the workgroup
     * counts itself in (k_ocb): they are performed where every XCD sees them, so the counting needs no release
     * fence -- which on this part writes back the XCD's whole L2, full of ciphertext that nobody is waiting for */
    if (threadIdx.x == 0) {
        u32 *row = (u32 *)share;
        const u32 *acc = (const u32 *)(uaes_lds + OCB_LDS_ACC);
        u32 old = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) old |= __hip_atomic_exchange(row + q, acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");
    }
}

/* ------------------------------------------------------------------------ */
/* tails, PMAC of the associated data, tag                                     */
/* ------------------------------------------------------------------------ */
__device__ __forceinline__ B16 ocb_load_bytes(const unsigned char *p, u32 n)
{
    B16 b = { { 0, 0, 0, 0 } };
    if (n == 16 && (((uintptr_t)p) & 15u) == 0) return b16(*(const uint4 *)p);
#pragma unroll                                                /* constant word indices: a block indexed at run time lives
                                                                 in scratch memory, a round trip to it per access */
    for (u32 i = 0; i < 16; ++i)
        if (i < n) b.w[i >> 2] |= (u32)p[i] << (8 * (i & 3));
    return b;
}

/* byte `pos` (0..15, run time) of a block in registers: ^= v, and its value */
__device__ __forceinline__ void b16_xor_byte(B16 &b, u32 pos, u32 v)
{
#pragma unroll
    for (u32 q = 0; q < 4; ++q) b.w[q] ^= q == (pos >> 2) ? v << (8 * (pos & 3)) : 0u;
}

__device__ __forceinline__ u32 b16_byte(const B16 &b, u32 pos)
{
    const u32 q = pos >> 2;
    const u32 w = q == 0 ? b.w[0] : q == 1 ? b.w[1] : q == 2 ? b.w[2] : b.w[3];
    return (w >> (8 * (pos & 3))) & 0xffu;
}

/* the caller has put the L table, Offset_0, CLEARED accumulators (ocb_clear_acc) and what `enc` needs into
 * LDS; parts = the checksum shares of the nparts workgroups of the block loop (any address space), hparts = their
 * nhash shares of HASH(K, A) over the first na_done blocks of the associated data (the rest is hashed here) */
template <typename ENC>
__device__ __forceinline__ void ocb_final_body(const ENC &enc,
                                               const uint4 *parts, u32 nparts,
                                               const uint4 *hparts, u32 nhash, u64 na_done, int decrypt,
                                               const unsigned char *__restrict__ aad, u64 aad_len,
                                               const unsigned char *in, unsigned char *out, u64 len,
                                               int *status, u32 tag_len)
{

    /* HASH(K, A): Sum ^= Enc(A_i ^ Offset_i), Offset from zero (:1749-1754); all threads */
    u32 h[4] = { 0, 0, 0, 0 };
    const u64 na = aad_len >> 4;
    for (u32 i = threadIdx.x; i < nhash; i += blockDim.x) {
        const uint4 v = hparts[i];
        h[0] ^= v.x; h[1] ^= v.y; h[2] ^= v.z; h[3] ^= v.w;
    }
    for (u64 i = na_done + threadIdx.x + 1; i <= na; i += blockDim.x) {
        B16 b = ocb_load_bytes(aad + 16 * (i - 1), 16);
        bx(b, ocb_delta(i));
        enc(b);
        h[0] ^= b.w[0]; h[1] ^= b.w[1]; h[2] ^= b.w[2]; h[3] ^= b.w[3];
    }
    if (threadIdx.x == 0 && (aad_len & 15u)) {        /* A_* || 1 || 0.., Offset_* = Offset_m ^ L_* (:1755-1760) */
        const u32 r = (u32)(aad_len & 15u);
        B16 b = ocb_load_bytes(aad + 16 * na, r);
        b16_xor_byte(b, r, 0x80u);
        bx(b, ocb_delta(na));
        bx(b, lds_row(0));
        enc(b);
        h[0] ^= b.w[0]; h[1] ^= b.w[1]; h[2] ^= b.w[2]; h[3] ^= b.w[3];
    }
    wave_xor_reduce(h);
    /* checksum shares of the k_ocb workgroups */
    u32 cs[4] = { 0, 0, 0, 0 };
    for (u32 i = threadIdx.x; i < nparts; i += blockDim.x) {
        const uint4 v = parts[i];
        cs[0] ^= v.x; cs[1] ^= v.y; cs[2] ^= v.z; cs[3] ^= v.w;
    }
    wave_xor_reduce(cs);
    if ((threadIdx.x & 63u) == 0) {
        u32 *acc = (u32 *)(uaes_lds + OCB_LDS_ACC);
#pragma unroll
        for (int q = 0; q < 4; ++q) { atomicXor(acc + q, h[q]); atomicXor(acc + 4 + q, cs[q]); }
    }
    __syncthreads();
    /* the rest is one thread's work; the others fall through to the end of the kernel (a completion ticket
     * riding on k_ocb_small needs every thread there, ticket_release)                                   */
    if (threadIdx.x == 0) {
        const u64 n = len >> 4;
        const u32 r = (u32)(len & 15u);
        B16 d = b16(*(const uint4 *)(uaes_lds + OCB_LDS_OFF0));
        bx(d, ocb_delta(n));                              /* Offset_m */
        B16 ck = b16(*(const uint4 *)(uaes_lds + OCB_LDS_ACC + 16u));
        if (r) {                                          /* :1736-1741 */
            bx(d, lds_row(0));                            /* Offset_* = Offset_m ^ L_* */
            B16 pad = d;
            enc(pad);
            for (u32 i = 0; i < r; ++i) {
                const u32 x = in[16 * n + i];
                const u32 y = x ^ b16_byte(pad, i);
                b16_xor_byte(ck, i, decrypt ? y : x);
                out[16 * n + i] = (unsigned char)y;
            }
            b16_xor_byte(ck, r, 0x80u);
        }
        bx(ck, d);
        bx(ck, lds_row(1));                               /* ^ L_$ */
        enc(ck);
        bx(ck, b16(*(const uint4 *)(uaes_lds + OCB_LDS_ACC)));
        if (!decrypt) {                                   /* the first OCB_TAG_LEN bytes (:1783 / :1807) */
#pragma unroll
            for (u32 i = 0; i < 16; ++i)
                if (i < tag_len) out[len + i] = (unsigned char)(ck.w[i >> 2] >> (8 * (i & 3)));
        } else {
            u32 diff = 0;
#pragma unroll
            for (u32 i = 0; i < 16; ++i)
                if (i < tag_len) diff |= (u32)in[len + i] ^ ((ck.w[i >> 2] >> (8 * (i & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
        }
    }
}

/* One call = one launch.  Every workgroup: encryption tables, L table + Offset_0 (first wave), for decryption
 * the tables of the other direction, then its runs of chunks.  Its checksum share goes to the scratch row of the
 * workgroup (plain store, released to the device by thread 0) and the workgroup counts itself in on *done_word;
 * whoever arrives last acquires the others' shares and finishes the call: PMAC of the associated data, the ragged
 * block, the tag (decryption: the encryption tables come back first).  *done_word is zero between calls (the last
 * arrival puts it back; the host layer hands out a word that nothing else writes).  A completion ticket may ride
 * on the launch (every thread ends in ticket_release).
 * Against setup / block loop / tag as three launches (round 4): 1 MiB 25.4 -> 17.0 us, 16 MiB 32.3 -> 26.6, 1 GiB +-0. */
struct OcbArgs {
    uaesk_rk ek, dk;                                  /* dk = ek for encryption */
    uaesk_tables tb;
    uint4 nonce_block;
    uint4 *scr;
    unsigned *done_word;
    u64 nblocks, aad_len, len;
    u64 hash_blocks;                                  /* whole blocks of associated data hashed by all workgroups (0: by the last one) */
    const unsigned char *aad, *in;
    unsigned char *out;
    int *status;
    u32 bottom, run, run_aad, tag_len;
    uaesk_done done;
};

/* The arguments are read through the kernel-argument pointer where they are used: as plain parameters all of them
 * are loaded at the entry and what only the last phase needs (pointers, lengths, for decryption a second key
 * schedule) sits in scalar registers across the block loop, which has none to spare (22 more v_readlane per trip). */
template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_ocb(OcbArgs)
{
    const OcbArgs *const a = (const OcbArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    if (DEC) {
        ocb_fill_plain_te0(a->tb.te0);
        fill_dec_tables(a->tb.td0);
    } else {
        fill_enc_tables(a->tb.te0);
    }
    const LaneConst lc = make_lane_const();
    if (threadIdx.x < 64) {
        const uaesk_rk ek = a->ek;
        if (DEC) ocb_setup_lds(EncPlain<NR>{ ek }, a->nonce_block, a->bottom);
        else ocb_setup_lds(EncRep<NR>{ ek, lc }, a->nonce_block, a->bottom);
    }
    __syncthreads();
    const u64 nblocks = a->nblocks;
    if (nblocks) {
        const uaesk_rk rk = DEC ? a->dk : a->ek;
        ocb_main_body<NR, DEC>(rk, lc, a->scr + OCB_ROW_PART + blockIdx.x, nblocks, a->run, (const uint4 *)a->in, (uint4 *)a->out);
    }
    /* long associated data: every workgroup hashes its part of it (a decrypting launch swaps the tables first) --
     * left to the finishing workgroup alone, 32 MiB of it took 7.4 ms */
    const u64 hash_blocks = a->hash_blocks;
    const bool enc_tables = !DEC || hash_blocks;      /* the replicated tables of the encryption direction are in LDS */
    if (hash_blocks) {
        __syncthreads();
        ocb_clear_acc();
        if (DEC) fill_enc_tables(a->tb.te0); else __syncthreads();
        const uaesk_rk ek = a->ek;
        ocb_main_body<NR, false, true>(ek, lc, a->scr + OCB_ROW_HASH + blockIdx.x, hash_blocks, a->run_aad,
                                       (const uint4 *)a->aad, nullptr);
    }
    bool last = true;
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) {                       /* after ocb_main_body's store of the share, same thread */
            unsigned *const done_word = a->done_word;
            const unsigned arrived = __hip_atomic_fetch_add(done_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned l = arrived == gridDim.x - 1u ? 1u : 0u;
            if (l) __hip_atomic_store(done_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(volatile u32 *)(uaes_lds + OCB_LDS_FLAG) = l;
        }
        __syncthreads();
        last = __builtin_amdgcn_readfirstlane((int)*(volatile u32 *)(uaes_lds + OCB_LDS_FLAG)) != 0;
    }
    if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        ocb_clear_acc();
        const uaesk_rk ek = a->ek;
        const u32 nparts = nblocks ? gridDim.x : 0u, nhash = hash_blocks ? gridDim.x : 0u;
        /* decryption: the few encryptions of the tag go through the plain table; associated data beyond a few
         * blocks per thread is worth bringing the replicated encryption tables back for */
        if (enc_tables || a->aad_len > OCB_PLAIN_AAD) {
            if (!enc_tables) fill_enc_tables(a->tb.te0); else __syncthreads();
            ocb_final_body(EncRep<NR>{ ek, lc }, a->scr + OCB_ROW_PART, nparts, a->scr + OCB_ROW_HASH, nhash, hash_blocks,
                           DEC ? 1 : 0, a->aad, a->aad_len, a->in, a->out, a->len, a->status, a->tag_len);
        } else {
            __syncthreads();
            ocb_final_body(EncPlain<NR>{ ek }, a->scr + OCB_ROW_PART, nparts, a->scr + OCB_ROW_HASH, 0u, 0ull,
                           1, a->aad, a->aad_len, a->in, a->out, a->len, a->status, a->tag_len);
        }
    }
    ticket_release(a->done);
}

/* Short messages (<= 1024 whole blocks, <= 64 KiB of associated data): ONE workgroup, one block per lane -- the
 * three phases run one after the other with everything handed over in LDS; decryption swaps the cipher tables
 * between the phases (setup and tag use Enc).
 * 4 KiB call: 38.5 -> 30 us (what is left is serial: tables, L_* / K_top / tag encryptions by one lane).                                                                            */
#define OCB_SMALL_BLOCKS 1024u
#define OCB_SMALL_AAD    65536u
#define OCB_SMALL_SCR    ((OCB_LDS + 63u) & ~63u)                 /* the checksum share, behind the flag */
#define OCB_SMALL_LDS    (OCB_SMALL_SCR + 16u)

template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_ocb_small(uaesk_rk ek, uaesk_rk dk, uaesk_tables tb,
                                                       uint4 nonce_block, u32 bottom,
                                                       const unsigned char *__restrict__ aad, u64 aad_len,
                                                       const unsigned char *in, unsigned char *out, u64 len,
                                                       int *status, u32 tag_len, uaesk_done done)
{
#define OT(i) do { } while (0)
    OT(0);
    uint4 *const part = (uint4 *)(uaes_lds + OCB_SMALL_SCR);
    const u64 nblocks = len >> 4;
    const bool dec_tables = DEC && nblocks;                   /* else: an encrypting launch in all but name */
    if (dec_tables) {
        ocb_fill_plain_te0(tb.te0);
        fill_dec_tables(tb.td0);
    } else {
        fill_enc_tables(tb.te0);
    }
    const LaneConst lc = make_lane_const();
    OT(1);
    if (threadIdx.x < 64) {
        if (dec_tables) ocb_setup_lds(EncPlain<NR>{ ek }, nonce_block, bottom);
        else ocb_setup_lds(EncRep<NR>{ ek, lc }, nonce_block, bottom);
    }
    __syncthreads();
    OT(2);
    u32 nparts = 0;
    if (nblocks) {
        /* one block per lane (block i = thread + 1, Offset_i = Offset_0 ^ the L_j of gray(i)): a 4 KiB text runs on
         * four waves with the latency of ONE block encryption; ocb_main_body's wave-per-256-blocks layout would
         * put it on a single wave, four blocks in a row                                                       */
        {
            const u64 i = (u64)threadIdx.x + 1;
            const bool live = i <= nblocks;
            const u64 ic = live ? i : nblocks;                 /* clamped: no branch around the rounds */
            B16 off = b16(*(const uint4 *)(uaes_lds + OCB_LDS_OFF0));
            bx(off, ocb_delta(ic));
            const uint4 d = ((const uint4 *)in)[ic - 1];
            const u32 lv = live ? 0xffffffffu : 0u;
            u32 sum[4] = { 0, 0, 0, 0 };
            u32 s1[1][4] = { { d.x ^ off.w[0], d.y ^ off.w[1], d.z ^ off.w[2], d.w ^ off.w[3] } };
            if (!DEC) { sum[0] = d.x & lv; sum[1] = d.y & lv; sum[2] = d.z & lv; sum[3] = d.w & lv; }
            if (DEC) dec_blocks<NR, 1>(s1, dk, lc); else enc_blocks<NR, 1>(s1, ek, lc);
#pragma unroll
            for (int q = 0; q < 4; ++q) s1[0][q] ^= off.w[q];
            if (DEC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] = s1[0][q] & lv;
            }
            if (live) ((uint4 *)out)[i - 1] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
            wave_xor_reduce(sum);
            if ((threadIdx.x & 63u) == 0) {
                u32 *acc = (u32 *)(uaes_lds + OCB_LDS_ACC);
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicXor(acc + q, sum[q]);
            }
            __syncthreads();
            if (threadIdx.x == 0) *part = *(const uint4 *)(uaes_lds + OCB_LDS_ACC);
        }
        nparts = 1;
        __syncthreads();
        ocb_clear_acc();
    }
    OT(3);
    if (dec_tables && aad_len > OCB_PLAIN_AAD) {
        fill_enc_tables(tb.te0);
        OT(4);
        ocb_final_body(EncRep<NR>{ ek, lc }, part, nparts, part, 0u, 0ull, 1, aad, aad_len, in, out, len, status, tag_len);
    } else {
        __syncthreads();
        OT(4);
        if (dec_tables) ocb_final_body(EncPlain<NR>{ ek }, part, nparts, part, 0u, 0ull, 1, aad, aad_len, in, out, len, status, tag_len);
        else ocb_final_body(EncRep<NR>{ ek, lc }, part, nparts, part, 0u, 0ull, DEC ? 1 : 0, aad, aad_len, in, out, len, status, tag_len);
    }
#undef OT
    ticket_release(done);
}

/* ------------------------------------------------------------------------ */
/* launcher                                                                    */
/* ------------------------------------------------------------------------ */
static unsigned cu_count()
{
    static int cus = 0;
    if (!cus) uaesk_device_info(&cus, nullptr);
    return cus > 0 ? (unsigned)cus : 256u;
}

static hipError_t want_lds(const void *kern)
{
    return uaesk_want_lds(kern, (unsigned)(OCB_LDS));
}

/* OCB's rows of the table of arrangements (uaes_plan.h): one workgroup for a short message with short associated data,
 * else runs of chunks over as many workgroups as there are CUs -- both ONE launch, the last workgroup to arrive makes
 * the tag */
bool uaesk_arr_on(int id);                                       /* uaes_kernels.hip */
static uaes_plan plan_ocb(u64 len, u64 aad_len)
{
    uaes_plan p = { UAES_ARR_OCB_RUNS, 1, 0, 0 };
    if ((len >> 4) <= OCB_SMALL_BLOCKS && aad_len <= OCB_SMALL_AAD && uaesk_arr_on(UAES_ARR_OCB_SMALL)) { p.arrangement = UAES_ARR_OCB_SMALL; p.grid = 1; }
    return p;
}
int uaesk_plan_ocb(int dir, size_t len, size_t aad_len, uaes_plan *p)
{
    if (dir < 0 || dir > 1) return (int)hipErrorInvalidValue;
    *p = plan_ocb(len, aad_len);
    return 0;
}

template <int NR>
static int launch_ocb(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_rk *dk,
                      int decrypt, uint4 nb, u32 bottom, u32 tag_len, const void *aad, size_t aad_len,
                      const void *in, size_t len, void *out, void *scratch, unsigned *done_word, int *status)
{
    uint4 *scr = (uint4 *)scratch;
    hipError_t e;
    /* one launch either way, which can carry the call's completion ticket (a decryption's status word must then
     * be host-visible: the host layer arms a ticket only when it passes a pinned status pointer) */
    const uaesk_done done = uaesk_ticket_take();
    if (plan_ocb(len, aad_len).arrangement == UAES_ARR_OCB_SMALL) {         /* short message: one workgroup */
        const void *ks = decrypt ? (const void *)k_ocb_small<NR, true> : (const void *)k_ocb_small<NR, false>;
        if ((e = uaesk_want_lds(ks, (unsigned)OCB_SMALL_LDS)) != hipSuccess) return (int)e;
        if (decrypt)
            hipLaunchKernelGGL((k_ocb_small<NR, true>), dim3(1), dim3(UAES_WG), OCB_SMALL_LDS, st, *ek, *dk, *tb, nb, bottom,
                               (const unsigned char *)aad, (u64)aad_len, (const unsigned char *)in,
                               (unsigned char *)out, (u64)len, status, tag_len, done);
        else
            hipLaunchKernelGGL((k_ocb_small<NR, false>), dim3(1), dim3(UAES_WG), OCB_SMALL_LDS, st, *ek, *ek, *tb, nb, bottom,
                               (const unsigned char *)aad, (u64)aad_len, (const unsigned char *)in,
                               (unsigned char *)out, (u64)len, status, tag_len, done);
        return (int)hipGetLastError();
    }
    if ((e = want_lds(decrypt ? (const void *)k_ocb<NR, true> : (const void *)k_ocb<NR, false>)) != hipSuccess) return (int)e;
    const u64 nblocks = len >> 4;
    /* long associated data (whole blocks, 16-byte aligned) is hashed by all workgroups as a second walk */
    const u64 hblocks = ((u64)(aad_len >> 4) >= OCB_HASH_SPREAD && ((uintptr_t)aad & 15u) == 0) ? (u64)(aad_len >> 4) : 0;
    unsigned wg = 256u;
    u64 grid = 1;
    u32 runs_of[2] = { 1, 1 };
    const u64 walk[2] = { nblocks, hblocks };
    for (int w = 0; w < 2; ++w) {
        if (!walk[w]) continue;
        const u64 nchunks = (walk[w] >> 8) + 1;
        /* at least 8 runs per wave, so that an uneven split costs at most 1/8 */
        u32 run = OCB_RUN_MAX;
        while (run > 1 && nchunks / run < 8ull * cu_count() * (UAES_WG / 64)) run >>= 1;
        const u64 runs = (nchunks + run - 1) / run;
        /* short texts: 4-wave workgroups, so that the few chunks spread over more CUs */
        const unsigned wgw = (runs + UAES_WG / 64 - 1) / (UAES_WG / 64) * 2 <= cu_count() ? 256u : UAES_WG;
        runs_of[w] = run;
        if (wgw > wg) wg = wgw;
    }
    for (int w = 0; w < 2; ++w) {
        if (!walk[w]) continue;
        const u64 runs = (((walk[w] >> 8) + 1) + runs_of[w] - 1) / runs_of[w];
        u64 g = (runs + wg / 64 - 1) / (wg / 64);
        if (g > grid) grid = g;
    }
    if (grid > cu_count()) grid = cu_count();
    if (grid > OCB_MAX_WGS) grid = OCB_MAX_WGS;
    const u32 run = runs_of[0];
    OcbArgs ka;
    ka.ek = *ek; ka.dk = decrypt ? *dk : *ek; ka.tb = *tb; ka.nonce_block = nb; ka.scr = scr; ka.done_word = done_word;
    ka.nblocks = nblocks; ka.aad_len = aad_len; ka.len = len; ka.aad = (const unsigned char *)aad;
    ka.hash_blocks = hblocks; ka.run_aad = runs_of[1];
    ka.in = (const unsigned char *)in; ka.out = (unsigned char *)out; ka.status = status;
    ka.bottom = bottom; ka.run = run; ka.tag_len = tag_len; ka.done = done;
    if (decrypt) hipLaunchKernelGGL((k_ocb<NR, true>), dim3((unsigned)grid), dim3(wg), OCB_LDS, st, ka);
    else hipLaunchKernelGGL((k_ocb<NR, false>), dim3((unsigned)grid), dim3(wg), OCB_LDS, st, ka);
    return (int)hipGetLastError();
}

extern "C" size_t uaesk_ocb_scratch_bytes(void) { return 16u * (OCB_ROW_HASH + OCB_MAX_WGS); }

/* nonce is a host pointer; everything else device memory (in/out 16-byte aligned).  done_word: a device word that
 * is zero between calls and that nothing else writes (the workgroups of a launch count themselves in on it).
 * encrypt: tag_len bytes of tag written at out+len.  decrypt: tag read at in+len, *status = 0 / 0x1A,
 * the text is written either way (as in the reference, :1804-1809).
 * nonce_len / tag_len = the reference's OCB_NONCE_LEN (1..15) / OCB_TAG_LEN (1..16), micro_aes.h:115-116 */
extern "C" int uaesk_ocb(void *stream, const uaesk_tables *tb, int nr,
                         const uaesk_rk *ek, const uaesk_rk *dk, int decrypt, const uint8_t *nonce,
                         size_t nonce_len, size_t tag_len,
                         const void *aad, size_t aad_len, const void *in, size_t len, void *out,
                         void *scratch, unsigned *done_word, int *status)
{
    if (nonce_len < 1 || nonce_len > 15 || tag_len < 1 || tag_len > 16) return (int)hipErrorInvalidValue;
    uint8_t kt[16];
    memset(kt, 0, sizeof kt);
    memcpy(kt + 16 - nonce_len, nonce, nonce_len);    /* :1706 */
    kt[0] |= (uint8_t)(tag_len << 4);                 /* :1707 (tag length 128 encodes as 0 in the top 7 bits) */
    kt[15 - nonce_len] |= 1;                          /* :1708 */
    const u32 bottom = kt[15] & 63u;
    kt[15] &= 0xC0;
    uint4 nb;
    memcpy(&nb, kt, 16);
    const u32 tl = (u32)tag_len;
    switch (nr) {
    case 10: return launch_ocb<10>(S(stream), tb, ek, dk, decrypt, nb, bottom, tl, aad, aad_len, in, len, out, scratch, done_word, status);
    case 12: return launch_ocb<12>(S(stream), tb, ek, dk, decrypt, nb, bottom, tl, aad, aad_len, in, len, out, scratch, done_word, status);
    case 14: return launch_ocb<14>(S(stream), tb, ek, dk, decrypt, nb, bottom, tl, aad, aad_len, in, len, out, scratch, done_word, status);
    default: return (int)hipErrorInvalidValue;
    }
}
