/*
 * uaes_ctr.hip.h -- the counter-mode core shared by the CTR kernel (uaes_kernels.hip)
 * and the fused GCM encrypt kernel (uaes_gcm.hip).
 *
 *   ctr_words        <- the counter block of stream block i (incBlock, micro_aes.c:421-427)
 *   ctr_edges        <- blocks outside the whole chunks + the byte tail (mixThenXor, :949)
 *   ctr_shared_loop  <- CTR_cipher (:943-949) with rounds 1-2 shared between the 256
 *                       counters of a group
 */
#ifndef UAES_CTR_HIP_H_
#define UAES_CTR_HIP_H_

#include "uaes_aes.hip.h"

/* counter block for stream block i: bytes 0..8 fixed, bytes 9..15 = 56-bit
 * big-endian (v0 + i) mod 2^56 (reference N2)                               */
__device__ __forceinline__ void ctr_words(const uaesk_ctr &c, u64 i, u32 (&w)[4])
{
    if (c.le32) {                              /* GCM-SIV: LE32 counter in bytes 0..3 (wave-uniform branch) */
        w[0] = c.w0 + (u32)i;
        w[1] = c.w1; w[2] = c.w2; w[3] = c.w3;
        return;
    }
    const u64 v = (c.v0 + i) & 0x00FFFFFFFFFFFFFFull;
    w[0] = c.w0;
    w[1] = c.w1;
    w[2] = bswap32((c.b8 << 24) | (u32)(v >> 32));
    w[3] = bswap32((u32)v);
}

/* one block through the plain NR-round path: out[i] = in[i] ^ Enc(counter i); returns the ciphertext */
template <int NR, typename LC>
__device__ __forceinline__ uint4 ctr_one_block(const uaesk_rk &rk, const uaesk_ctr &ctr, const uint4 *in, uint4 *out,
                                               u64 i, const LC &lc)
{
    u32 s1[1][4];
    ctr_words(ctr, i, s1[0]);
    enc_blocks<NR, 1>(s1, rk, lc);
    const uint4 d = in[i];
    const uint4 c = make_uint4(d.x ^ s1[0][0], d.y ^ s1[0][1], d.z ^ s1[0][2], d.w ^ s1[0][3]);
    out[i] = c;
    return c;
}

/* reference N3: len%16 tail bytes use Enc(ctr_final) (mixThenXor, :949); one thread */
template <int NR, typename LC>
__device__ __forceinline__ void ctr_byte_tail(const uaesk_rk &rk, const uaesk_ctr &ctr, const uint4 *in, uint4 *out,
                                              u64 nfull, u32 rem, const LC &lc)
{
    u32 s1[1][4];
    ctr_words(ctr, nfull, s1[0]);
    enc_blocks<NR, 1>(s1, rk, lc);
    const unsigned char *src = (const unsigned char *)(in + nfull);
    unsigned char *dst = (unsigned char *)(out + nfull);
    for (u32 i = 0; i < rem; ++i)
        dst[i] = src[i] ^ (unsigned char)(s1[0][i >> 2] >> (8 * (i & 3)));
}

/* Blocks [0, pre_end) and [suf, nfull) and the byte tail, spread evenly over all
 * workgroups, one block per thread through the plain 10/12/14-round path, so no
 * separate launch (and no second table fill) is needed.                         */
template <int NR, typename LC>
__device__ __forceinline__ void ctr_edge_blocks(const uaesk_rk &rk, const uaesk_ctr &ctr, const uint4 *in, uint4 *out,
                                                u64 pre_end, u64 suf, u64 nfull, u32 rem, const LC &lc)
{
    const u64 nedge = pre_end + (nfull - suf);
    const u64 per = (nedge + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per;
    u64 hi = lo + per;
    if (hi > nedge) hi = nedge;
    for (u64 e = lo + threadIdx.x; e < hi; e += UAES_WG)
        (void)ctr_one_block<NR>(rk, ctr, in, out, e < pre_end ? e : suf + (e - pre_end), lc);
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) ctr_byte_tail<NR>(rk, ctr, in, out, nfull, rem, lc);
}

#define CTRS_CHUNK 64u                       /* groups (of 256 counters) per U-buffer refill: one barrier each;
                                                256 measured no faster (profiles/HISTORY.md) */

/* ------------------------------------------------------------------------ */
/* CTR with shared rounds 1-2                                                 */
/* ------------------------------------------------------------------------ */
/* Consecutive counter blocks differ only in their low bytes.  Cut the stream
 * into GROUPS of 256 counters that share bytes 0..14 (group G, position p =
 * counter byte 15), and pin every lane to one p for the whole kernel:
 *
 *   after AddRoundKey(0) only state byte 15 depends on p; after round 1 only
 *   column 0 does:  col0 = A(G) ^ Te3[p ^ rk0.b15],  col1..3 = uniform(G);
 *   after round 2 every column is  Te_k[one byte of col0] ^ U_c(G).
 *
 * A(G) changes only when counter bits 40..47 change, so the four round-2
 * lookups on col0's bytes are per-LANE constants L_c, computed once; the
 * uniform parts U_c(G) cost 27 lookups per GROUP (one lane of wave 0 per group,
 * handed over through LDS).  A block therefore enters round 3 as L ^ U(G):
 * 4 XORs instead of 32 table lookups -- 128 lookups per AES-128 block instead
 * of 160 on a path whose bound is the LDS lookup rate (32 lanes/clk/CU).
 *
 * Workgroup = 16 waves = 4 quads; wave w owns positions p = 64*(w&3) + lane.  One
 * ITERATION of the workgroup covers 8 consecutive groups (2048 blocks): quad q
 * takes groups q and q+4 of them, two blocks per lane, half a round out of phase
 * (enc_rounds_skewed), so each wave-level load/store is one contiguous 1 KiB
 * segment.  The 8-group stripes are dealt round-robin: iteration `it` of workgroup b
 * takes groups g_lo + 8*(b + grid*it) ...  Block (b, it, q, u, p) then sits at  j + S*it
 * with the lane index j = 2048 b + 256 (q + 4u) + p and S = 2048 grid: exactly the strided
 * Horner layout of GHASH, so a lane can fold its own ciphertext blocks as it produces them
 * (the fused GCM kernel), and every workgroup gets floor or ceil of stripes/grid stripes.
 *
 * U-buffer: the uniform parts of the next 8 iterations (64 groups), double buffered,
 * one s_barrier per 8 iterations.                                                */
/* A(G) is the same for every group of ONE LAUNCH: it depends on the counter's bits 40..47 only, and the launchers cut
 * a text at the (one in 2^40 blocks) place where those move -- ctr_stripes_cross_a() below, launch_ctr_shared and the
 * GCM launchers -- so the lane constants L are made once per kernel and are loop invariants.  (Until round 5 the loop
 * watched A per block and redefined L under a wave-uniform condition: a loop-carried value with a conditional
 * definition, which cost eight v_mov per trip and a second copy of the flags in the U-buffer.)                   */
static inline bool ctr_stripes_cross_a(const uaesk_ctr *c, u64 g_lo, u64 n8)
{
    if (!n8) return false;
    const u64 vbase = c->v0 - (c->v0 & 0xffu);
    const u64 first = vbase + (g_lo << 8), last = vbase + ((g_lo + 8 * n8) << 8) - 1;     /* (a wrap at 2^56 counts) */
    return (first >> 40) != (last >> 40);
}

struct CtrGeo {
    u64 first;          /* g_lo: first group of the striped region                    */
    u64 iters;          /* iterations (stripes) of THIS workgroup                     */
};

__device__ __forceinline__ u64 ctr_geo_group0(const CtrGeo &g, u64 it)
{
    return g.first + 8ull * ((u64)blockIdx.x + (u64)gridDim.x * it);
}

/* FOLD: functor called once per iteration with two blocks of this lane (u = 0, 1), in stream
 * order per lane; `void operator()(uint4 b0, uint4 b1)`.  FOLD::of_input selects which side of the
 * XOR it sees: the blocks just written (encrypting GCM hashes its output) or the blocks just
 * read (decrypting GCM hashes its input).                                            */
struct CtrNoFold {
    static constexpr bool of_input = false;
    static constexpr int round_prio = 1;       /* wave priority while a round's lookups are issued */
    static constexpr bool expand2 = true;      /* the loop body twice per trip, text buffers swapped (no register moves) */
    static constexpr bool text_ahead = true;   /* the text of iteration it + 1 is requested before the rounds of iteration it */
    __device__ __forceinline__ void operator()(const uint4 &, const uint4 &) const {}
};

/* lds_buf: byte offset of 2 x 64 x 32 B of LDS for the U-buffer */
template <int NR, typename LC, typename FOLD>
__device__ __forceinline__ void ctr_shared_loop(const uaesk_rk &rk, const uaesk_ctr &ctr, const uint4 *in, uint4 *out,
                                                const CtrGeo &geo, u32 lds_buf, const LC &lc, FOLD &fold)
{
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 p = ((wave & 3u) << 6) | lane;               /* counter byte 15 of this lane */
    const u32 quad = wave >> 2;
    const u32 c0 = (u32)ctr.v0 & 0xffu;                    /* position of stream block 0   */
    const u64 vbase = ctr.v0 - c0;                         /* group-aligned counter        */
    uint4 *buf = (uint4 *)(uaes_lds + lds_buf);
    const u32 lane_blk = (quad << 8) | p;                  /* lane's block offset inside the 8-group stripe */

    /* byte 15 after AddRoundKey(0), as a Te3 lookup operand in byte 3 */
    const u32 x15 = ((p << 24) ^ rk.w[3]) & 0xff000000u;
    /* round keys 3..NR (wave-uniform: they stay in SGPRs; forcing them into VGPRs to
     * speed up v_bitop3 issue measured 3 % SLOWER -- more VGPRs, lower clock)        */
    struct { u32 w[4 * (NR - 2)]; } rkv;
#pragma unroll
    for (int i = 0; i < 4 * (NR - 2); ++i) rkv.w[i] = rk.w[12 + i];
    u32 L0 = 0, L1 = 0, L2 = 0, L3 = 0;
    u32 parity = 0;

    /* Block index of (it, u, lane) = ((group0(it) + 4u) << 8) - c0 [uniform] + lane_blk [per lane].
     * The plaintext of iteration it+1 is requested before the rounds of iteration it, so
     * HBM latency hides under ~260 table lookups.                                       */
    u64 it = 0;
    uint4 d_cur[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
    if (FOLD::text_ahead && it < geo.iters) {
        const u64 g0 = ctr_geo_group0(geo, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) d_cur[u] = (in + (((g0 + 4u * u) << 8) - c0))[lane_blk];
    }

    /* One iteration = two blocks per lane.  The body is written once and expanded twice per trip with the two text
     * buffers swapped, so that "next becomes current" is a renaming, not eight register moves; the text addresses are a
     * wave-uniform 64-bit base (SGPRs, advanced by scalar instructions) plus the lane's constant 32-bit offset, so no
     * 64-bit vector adds either: the VALU is the second-busiest unit of this kernel.          */
    const u32 lane_byte = lane_blk * 16u;
    /* the text of a stripe through a buffer resource whose base is the stripe (four SGPRs, rebuilt by scalar
     * instructions every iteration) + the lane's constant byte offset + 16 KiB for the second block of the lane */
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define STRIPE_RSRC(p) __builtin_amdgcn_make_buffer_rsrc( \
        (void *)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)(p) >> 32)) << 32) | \
                 (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)(p))), 0, 0x7fffffff, 0x00020000)
#define TL(k, b, w) tlook_true<k, b>(w, lc)
    if (geo.iters) {                                        /* the lane constants: A of the launch, then L0..L3 */
        const u64 v = (vbase + (ctr_geo_group0(geo, 0) << 8)) & 0x00FFFFFFFFFFFFFFull;
        const u32 s0 = ctr.w0 ^ rk.w[0], s1 = ctr.w1 ^ rk.w[1];
        const u32 s2 = bswap32((ctr.b8 << 24) | (u32)(v >> 32)) ^ rk.w[2];
        const u32 A = xor3(TL(0, 0, s0), TL(1, 1, s1), TL(2, 2, s2)) ^ rk.w[4];
        const u32 col0 = A ^ TL(3, 3, x15);
        L0 = TL(0, 0, col0);
        L1 = TL(3, 3, col0);
        L2 = TL(2, 2, col0);
        L3 = TL(1, 1, col0);
    }
    auto refill = [&](u64 it) {
        {
            if (wave == 0) {
                /* uniform part of rounds 1 and 2 for the 64 groups of iterations it .. it+7:
                 * lane gi handles group group0(it + gi/8) + gi%8                          */
                const u32 gi = lane;
                const u64 g = ctr_geo_group0(geo, it + (gi >> 3)) + (gi & 7u);
                const u64 v = (vbase + (g << 8)) & 0x00FFFFFFFFFFFFFFull;
                const u32 s0 = ctr.w0 ^ rk.w[0], s1 = ctr.w1 ^ rk.w[1];
                const u32 s2 = bswap32((ctr.b8 << 24) | (u32)(v >> 32)) ^ rk.w[2];
                const u32 s3 = (bswap32((u32)v) ^ rk.w[3]) & 0x00ffffffu;      /* byte 15 excluded */
                const u32 c1 = xor3(xor3(TL(0, 0, s1), TL(1, 1, s2), TL(2, 2, s3)), TL(3, 3, s0), rk.w[5]);
                const u32 c2 = xor3(xor3(TL(0, 0, s2), TL(1, 1, s3), TL(2, 2, s0)), TL(3, 3, s1), rk.w[6]);
                const u32 c3 = xor3(xor3(TL(0, 0, s3), TL(1, 1, s0), TL(2, 2, s1)), TL(3, 3, s2), rk.w[7]);
                const u32 u0 = xor3(TL(1, 1, c1), TL(2, 2, c2), TL(3, 3, c3)) ^ rk.w[8];
                const u32 u1 = xor3(TL(0, 0, c1), TL(1, 1, c2), TL(2, 2, c3)) ^ rk.w[9];
                const u32 u2 = xor3(TL(0, 0, c2), TL(1, 1, c3), TL(3, 3, c1)) ^ rk.w[10];
                const u32 u3 = xor3(TL(0, 0, c3), TL(2, 2, c1), TL(3, 3, c2)) ^ rk.w[11];
                buf[(parity * CTRS_CHUNK + gi) * 2 + 0] = make_uint4(u0, u1, u2, u3);
            }
            __syncthreads();
        }

    };
    auto body = [&](u64 it, uint4 (&d_cur)[2], uint4 (&d_nxt)[2]) {
        /* request the next iteration's plaintext (clamped to this workgroup's last one) -- or, for a fold that cannot
         * afford a second text buffer (FOLD::text_ahead == false), THIS iteration's: the text is not needed before the
         * rounds are done, and one iteration (~4.7 us per 2048 blocks of a workgroup) hides the load either way */
        const u64 g0 = ctr_geo_group0(geo, it);
        {
            const u64 gn = FOLD::text_ahead ? ctr_geo_group0(geo, it + 1 < geo.iters ? it + 1 : it) : g0;
            uint4 (&d_ld)[2] = FOLD::text_ahead ? d_nxt : d_cur;
            const __amdgpu_buffer_rsrc_t rin = STRIPE_RSRC(in + ((gn << 8) - c0));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, lane_byte, 16384 * u, 0);
                d_ld[u] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }

        u32 s[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32 gl = 8u * ((u32)it & 7u) + quad + 4u * u;            /* slot in the U-buffer */
            const uint4 uu = buf[(parity * CTRS_CHUNK + gl) * 2 + 0];
            s[u][0] = L0 ^ uu.x; s[u][1] = L1 ^ uu.y; s[u][2] = L2 ^ uu.z; s[u][3] = L3 ^ uu.w;
        }
        enc_rounds_skewed<NR, 3, decltype(rkv), false, LC, FOLD::round_prio>(s[0], s[1], rkv, lc);
        uint4 ct[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ct[u] = make_uint4(d_cur[u].x ^ s[u][0], d_cur[u].y ^ s[u][1], d_cur[u].z ^ s[u][2], d_cur[u].w ^ s[u][3]);
            /* the ciphertext goes out through a plain global pointer, NOT through a buffer resource like the loads:
             * `buffer_store_dwordx4 v[a:a+3], voff, s[rsrc], s_off offen` followed at once by a VALU write to v[a..a+3]
             * stores garbage for some lanes on gfx950 (seen in the one-pass GCM decrypt, whose GHASH selects reuse the
             * registers of the second store: ~1 % of the blocks came out as the selects' zeros, different ones every run,
             * profiles/r04_buffer_store_hazard.log).  The ISA manuals list that hazard -- store data of more than 64 bits,
             * then a VALU write of the data registers: 1 wait state -- with the exception "not if SOFFSET is an SGPR",
             * and the compiler's hazard recogniser follows them; for global_store it always inserts the wait state.
             * (The store path that reproduces it, and the timing-only builds without loads / stores, are a patch for the
             * measuring tools: tools/experiments/ctr_measurement_switches.patch.)                                      */
            (out + (((g0 + 4u * u) << 8) - c0))[lane_blk] = ct[u];
        }
        if (FOLD::of_input) fold(d_cur[0], d_cur[1]); else fold(ct[0], ct[1]);
    };
    uint4 d_a[2], d_b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) d_a[u] = d_cur[u];
    constexpr bool twice = FOLD::expand2;
    /* chunk by chunk (8 iterations = 64 groups of the U-buffer): refill, then the chunk's iterations */
    while (it < geo.iters) {
        const u64 end = it + 8 < geo.iters ? it + 8 : geo.iters;
        refill(it);
        if constexpr (twice) {
            for (; it + 2 <= end; it += 2) {
                body(it, d_a, d_b);
                body(it + 1, d_b, d_a);
            }
            if (it < end) { body(it, d_a, d_b); ++it; }      /* an odd count: only the launch's last stripe */
        } else {                                            /* a fold too big to hold twice (GCM: it would spill) */
            for (; it < end; ++it) {
                body(it, d_a, d_b);
                if (FOLD::text_ahead) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) d_a[u] = d_b[u];
                }
            }
        }
        parity ^= 1u;
    }
#undef TL
#undef STRIPE_RSRC
}

#endif
