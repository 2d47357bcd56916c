/*
 * uaes_kernels.hip -- gfx950 kernels for the block-parallel AES modes and
 * their C launchers (interface: uaes_device.h).
 *
 *   k_ecb   <- AES_ECB_encrypt/decrypt loops      micro_aes.c:644-651, :671-674
 *   k_ctr   <- CTR_cipher loop + incBlock         micro_aes.c:943-949, :421-427
 *   k_xts*  <- XTS_cipher + doubleLblock          micro_aes.c:1008-1055, :449-458
 *
 * Launch shape: persistent workgroups of 1024 threads (16 waves), one per CU
 * (the replicated tables take 96-128 KiB of the CU's 160 KiB LDS), grid-stride
 * over 16-byte blocks with 128-bit coalesced global loads/stores: lane l of a
 * workgroup touches block base + u*1024 + l, so every wave-level access is one
 * contiguous 1 KiB segment.  U independent blocks per lane give the LDS
 * pipeline enough parallel lookups to hide its latency.
 */
#include <hip/hip_runtime.h>
#include <string.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include "uaes_aes.hip.h"
#include "uaes_ctr.hip.h"
#include "uaes_gf.h"
#include "uaes_device.h"
#include "uaes_plan.h"
#include <atomic>

#define UAES_U 4            /* blocks per lane per iteration */

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

/* ------------------------------------------------------------------------ */
/* ECB                                                                        */
/* ------------------------------------------------------------------------ */
/* U = blocks per lane per iteration: 4 (two skewed pairs) for bulk texts, 1 for short ones, where
 * spreading the blocks over four times as many CUs beats instruction-level parallelism          */
template <int NR, bool DEC, int U>
__global__ __launch_bounds__(UAES_WG) void k_ecb(uaesk_rk rk, uaesk_tables tb,
                                                 const uint4 *in, uint4 *out,
                                                 u64 nfull, u32 rem, u32 padding, uaesk_done done,
                                                 u64 tail_from)        /* blocks from here on: one per thread (== nfull: none) */
{
    /* the first pass's text is requested before the tables are made (see k_ctr) */
    const u64 first = (u64)blockIdx.x * UAES_WG * U;
    uint4 d0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u64 i0 = first + (u64)u * UAES_WG + threadIdx.x;
        d0[u] = make_uint4(0, 0, 0, 0);
        if (i0 < nfull) d0[u] = in[i0];
    }
    if (DEC) fill_dec_tables(tb.td0); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * U;

    const u32 lane16 = threadIdx.x * 16u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    for (u64 base = first; base < tail_from; base += stride) {
        u32 s[U][4];
        u64 idx[U];
        /* a tile that lies completely inside the text (all but the last one of the launch) needs no bounds checks, and
         * its address is a wave-uniform base + the thread's constant offset: loads through a buffer resource (SGPR base,
         * one VGPR offset), no compares, no exec masking, no 64-bit vector adds -- ~9 VALU instructions per block less */
        const bool whole = base + (u64)UAES_WG * U <= nfull;
        if (whole && base != first) {
            const u64 b = (u64)(in + base);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                         (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)b)), 0, (int)(UAES_WG * U * 16u), 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane16, (int)(UAES_WG * 16u) * u, 0);
                idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
                s[u][0] = v.x; s[u][1] = v.y; s[u][2] = v.z; s[u][3] = v.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
                uint4 d = d0[u];
                if (base != first && idx[u] < nfull) d = in[idx[u]];
                s[u][0] = d.x; s[u][1] = d.y; s[u][2] = d.z; s[u][3] = d.w;
            }
        }
        if (DEC) {
            dec_blocks<NR, U>(s, rk, lc);
        } else if (U == 4) {                       /* two pairs, each half a round out of phase */
            enc_blocks_skewed<NR>(s[0], s[1], rk, lc);
            enc_blocks_skewed<NR>(s[2 % U], s[3 % U], rk, lc);
        } else {
            enc_blocks<NR, U>(s, rk, lc);
        }
        if (whole) {
            uint4 *o = out + base + threadIdx.x;              /* (global stores: uaes_ctr.hip.h says why not buffer stores) */
#pragma unroll
            for (int u = 0; u < U; ++u) o[(u32)u * UAES_WG] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (idx[u] < nfull) out[idx[u]] = make_uint4(s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    }

    /* A last round of tiles that covers only part of the grid would leave the other workgroups idle for a whole tile
     * (20 MiB = 320 tiles of 64 KiB on 256 workgroups: 1.25 rounds cost 2): the host hands such a remainder over as
     * [tail_from, nfull), one block per thread over ALL workgroups (as the CTR kernel's edge path).      */
    if (U > 1) {
        for (u64 i = tail_from + (u64)blockIdx.x * UAES_WG + threadIdx.x; i < nfull; i += (u64)gridDim.x * UAES_WG) {
            const uint4 d = in[i];
            u32 s1[1][4] = { { d.x, d.y, d.z, d.w } };
            if (DEC) dec_blocks<NR, 1>(s1, rk, lc); else enc_blocks<NR, 1>(s1, rk, lc);
            out[i] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
        }
    }

    /* reference N1: a trailing partial block is zero padded and ENCRYPTED into
     * a full output block (micro_aes.c:648-651); decrypt never gets here with
     * rem != 0 (host copies the ragged tail through and reports 0x1D).
     * padBlock (micro_aes.c:610-621): AES_PADDING 1 = PKCS#7 (n bytes of value n,
     * a whole block of 0x10 when rem == 0), 2 = ISO/IEC 7816-4 (0x80 then zeros);
     * both ALWAYS add a block.                                                 */
    if (!DEC && (rem || padding) && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned char *src = (const unsigned char *)(in + nfull);
        unsigned char pad[16];
        const unsigned char fillv = padding == 1 ? (unsigned char)(16u - rem) : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)               /* static indices only: no private segment for the bulk kernel */
            pad[i] = (u32)i < rem ? src[i] : (padding == 2 && (u32)i == rem) ? (unsigned char)0x80 : fillv;
        u32 s1[1][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
            s1[0][w] = pad[4 * w] | (pad[4 * w + 1] << 8) | (pad[4 * w + 2] << 16) | ((u32)pad[4 * w + 3] << 24);
        enc_blocks<NR, 1>(s1, rk, lc);
        out[nfull] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
    }
    ticket_release(done);
}

/* ------------------------------------------------------------------------ */
/* CTR (counter arithmetic, edges and the shared-round loop: uaes_ctr.hip.h)   */
/* ------------------------------------------------------------------------ */
/* the generic CTR kernel: work items are single blocks, U = 4 per lane per iteration */
template <int NR, int U>
__device__ __forceinline__ void ctr_generic_body(const uaesk_rk &rk, const uaesk_tables &tb, const uaesk_ctr &ctr,
                                                 const uint4 *in, uint4 *out,
                                                 u64 nfull, u32 rem, const int *__restrict__ gate, const uaesk_done &done)
{
    if (gate && *gate != 0) { ticket_release(done); return; }      /* GCM decrypt: tag mismatch -> untouched */
    /* the first pass's text is requested BEFORE the tables are made: a short call is this one pass, and the load
     * (from the host's pinned window, across the link, for a host caller) then runs beside the 0.7 us of table stores
     * instead of after them */
    const u64 first = (u64)blockIdx.x * UAES_WG * U;
    uint4 d0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u64 i0 = first + (u64)u * UAES_WG + threadIdx.x;
        d0[u] = make_uint4(0, 0, 0, 0);
        if (i0 < nfull) d0[u] = in[i0];
    }
    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * U;

    for (u64 base = first; base < nfull; base += stride) {
        u32 s[U][4];
        uint4 d[U];
        u64 idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
            d[u] = d0[u];
            if (base != first && idx[u] < nfull) d[u] = in[idx[u]];
            ctr_words(ctr, idx[u], s[u]);
        }
        if (U == 4) {                              /* two pairs, each half a round out of phase (as k_ecb) */
            enc_blocks_skewed<NR>(s[0], s[1], rk, lc);
            enc_blocks_skewed<NR>(s[2 % U], s[3 % U], rk, lc);
        } else {
            enc_blocks<NR, U>(s, rk, lc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (idx[u] < nfull)
                out[idx[u]] = make_uint4(d[u].x ^ s[u][0], d[u].y ^ s[u][1], d[u].z ^ s[u][2], d[u].w ^ s[u][3]);
    }

    /* reference N3: len%16 tail bytes use Enc(ctr_final) (mixThenXor, :949) */
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) ctr_byte_tail<NR>(rk, ctr, in, out, nfull, rem, lc);
    ticket_release(done);
}

template <int NR, int U>
__global__ __launch_bounds__(UAES_WG, 4) void k_ctr(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr,
                                                 const uint4 *in, uint4 *out,
                                                 u64 nfull, u32 rem, const int *__restrict__ gate, uaesk_done done)
{
    ctr_generic_body<NR, U>(rk, tb, ctr, in, out, nfull, rem, gate, done);
}

/* the same kernel with its key schedule and counter description in DEVICE memory: what an earlier kernel of the same
 * stream made them from (a long GCM-SIV message: the per-nonce key and the tag that is the counter never visit the
 * host, uaesk_gcmsiv_long).  Read at the kernel's entry, before any store: uniform, unclobbered -> scalar loads. */
template <int NR, int U>
__global__ __launch_bounds__(UAES_WG, 4) void k_ctr_ind(const uaesk_rk *__restrict__ rkp, uaesk_tables tb,
                                                     const uaesk_ctr *__restrict__ ctrp,
                                                     const uint4 *in, uint4 *out, u64 nfull, u32 rem)
{
    uaesk_rk rk;
#pragma unroll
    for (int i = 0; i < 4 * (NR + 1); ++i) rk.w[i] = (u32)__builtin_amdgcn_readfirstlane((int)rkp->w[i]);
    uaesk_ctr ctr;
    ctr.w0 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->w0);
    ctr.w1 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->w1);
    ctr.w2 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->w2);
    ctr.w3 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->w3);
    ctr.b8 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->b8);
    ctr.le32 = (u32)__builtin_amdgcn_readfirstlane((int)ctrp->le32);
    {
        const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)ctrp->v0);
        const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(ctrp->v0 >> 32));
        ctr.v0 = ((u64)hi << 32) | lo;
    }
    const uaesk_done none = { nullptr, nullptr, 0 };
    ctr_generic_body<NR, U>(rk, tb, ctr, in, out, nfull, rem, nullptr, none);
}

#define CTRS_BUF   (UAES_LDS_ENC)            /* 2 x 64 x 32 B after the tables */
#define UAES_LDS_CTRS (UAES_LDS_ENC + 2u * CTRS_CHUNK * 32u)

/* The shared-round CTR kernel (uaes_ctr.hip.h): the hot loop handles only whole 8-group
 * stripes (2048 blocks, 32 KiB) that lie completely inside the stream, so it has no bounds
 * checks; the blocks before and after them (up to block nfull) and a byte tail run in the
 * prologue.  The stripes are dealt round-robin over the workgroups, so every workgroup gets
 * floor or ceil of stripes/grid and the kernel's time follows the text size in 32 KiB steps
 * (round 1 dealt 256 KiB chunks: 80 MiB ran 10 % slower than 64 MiB,
 * profiles/HISTORY.md).  A lock-step four-block version
 * measured 4 % slower (profiles/HISTORY.md).                            */
template <int NR>
__global__ __launch_bounds__(UAES_WG, 4) void k_ctr_shared2(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr,
                                                           const uint4 *in, uint4 *out,
                                                           u64 g_lo, u64 stripes, u64 nfull, u32 rem,
                                                           const int *__restrict__ gate)
{
    if (gate && *gate != 0) return;
    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u32 c0 = (u32)ctr.v0 & 0xffu;
    CtrGeo geo;
    geo.first = g_lo;
    geo.iters = stripes / gridDim.x + (blockIdx.x < stripes % gridDim.x ? 1 : 0);
    ctr_edge_blocks<NR>(rk, ctr, in, out, g_lo * 256 - (g_lo ? c0 : 0),              /* blocks [0, pre_end) */
                        (g_lo + 8 * stripes) * 256 - c0, nfull, rem, lc);           /* blocks [suf, nfull) */
    CtrNoFold nofold;
    ctr_shared_loop<NR>(rk, ctr, in, out, geo, CTRS_BUF, lc, nofold);
}

/* ------------------------------------------------------------------------ */
/* XTS                                                                        */
/* ------------------------------------------------------------------------ */
/* tweak arithmetic (T * alpha^k by shifting): uaes_gf.h, struct Tw          */
#define XTS_CHUNK 256u      /* blocks per chunk = one wave x 4 blocks per lane */

/* pre-pass: one thread per data unit.  T0 = Enc_key2(tweak) (:1026-1027),
 * then the tweak at the start of every 256-block chunk of the unit; for units of
 * many chunks (serial != 0 only below XTS_SERIAL_CPS) k_xts_expand does the latter. */
/* PLAIN (round 5): up to 2^16 units -- one-wave workgroups that encrypt through an UNREPLICATED 1 KiB copy of Te0
 * (plain_encrypt; the lanes' lookups collide in the banks, which costs a wave's 160 lookups a microsecond): filling
 * 128 KiB of replicated tables for one block per lane was most of this kernel's 8.5 us (8 MiB of 4 KiB sectors:
 * the kernel 8.5 -> ~4 us).                                                                                    */
template <int NR, bool PLAIN = false>
__global__ __launch_bounds__(UAES_WG) void k_xts_tweaks(uaesk_rk k2, uaesk_tables tb,
                                                        uint4 raw_tweak, u32 use_raw, u64 first_sector,
                                                        u64 nsectors, u64 chunks_per_sector, u32 serial,
                                                        uint4 *__restrict__ chunk_tw)
{
    LaneConst lc;
    if (PLAIN) {
        for (u32 i = threadIdx.x; i < 256u; i += blockDim.x) ((u32 *)uaes_lds)[i] = tb.te0[i];
        __syncthreads();
    } else {
        fill_enc_tables(tb.te0);
        lc = make_lane_const();
    }
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 sct = (u64)blockIdx.x * blockDim.x + threadIdx.x; sct < nsectors; sct += stride) {
        u32 s[1][4];
        if (use_raw) {
            s[0][0] = raw_tweak.x; s[0][1] = raw_tweak.y; s[0][2] = raw_tweak.z; s[0][3] = raw_tweak.w;
        } else {
            const u64 id = first_sector + sct;  /* copyLint, micro_aes.c:399-404 */
            s[0][0] = (u32)id; s[0][1] = (u32)(id >> 32); s[0][2] = 0; s[0][3] = 0;
        }
        if (PLAIN) plain_encrypt<NR>((const u32 *)uaes_lds, k2, s[0]); else enc_blocks<NR, 1>(s, k2, lc);
        Tw t;
        t.lo = s[0][0] | ((u64)s[0][1] << 32);
        t.hi = s[0][2] | ((u64)s[0][3] << 32);
        for (u64 c = 0; c < (serial ? chunks_per_sector : 1); ++c) {
            chunk_tw[sct * chunks_per_sector + c] =
                make_uint4((u32)t.lo, (u32)(t.lo >> 32), (u32)t.hi, (u32)(t.hi >> 32));
            t = tw_mul_pow64(tw_mul_pow64(tw_mul_pow64(tw_mul_pow64(t))));
        }
    }
}

/* Chunk tweaks of LONG data units (the reference API is one unit per call, so a bulk
 * AES_XTS_encrypt is one unit of many chunks): instead of walking T <- T * alpha^256
 * down the unit, one wave takes 64 consecutive chunks.  The tweak of its first chunk,
 * T0 * alpha^(2^14 r), is a product of T0 with the host-made constants
 * alpha^(2^(14+i)) (bits i of r), each product computed by the whole wave (lane l
 * contributes coefficients l and l+64, butterfly XOR); lane j then shifts by 256 j.  */
#define XTS_POW_N 40
struct XtsPow {
    u64 lo[XTS_POW_N], hi[XTS_POW_N];
};

__device__ __forceinline__ Tw wave_tw_mul(Tw x, u64 elo, u64 ehi, u32 lane)
{
    const Tw xl = tw_mul_pow(x, lane), xh = tw_mul_pow64(xl);
    Tw z = { 0, 0 };
    if ((elo >> lane) & 1) z = xl;
    if ((ehi >> lane) & 1) { z.lo ^= xh.lo; z.hi ^= xh.hi; }
    u32 w[4] = { (u32)z.lo, (u32)(z.lo >> 32), (u32)z.hi, (u32)(z.hi >> 32) };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] ^= __shfl_xor(w[q], off, 64);
    }
    z.lo = w[0] | ((u64)w[1] << 32);
    z.hi = w[2] | ((u64)w[3] << 32);
    return z;
}

__global__ __launch_bounds__(UAES_WG) void k_xts_expand(XtsPow pw, u64 nsectors, u64 chunks_per_sector,
                                                        uint4 *__restrict__ chunk_tw)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 groups = (chunks_per_sector + 63) / 64;
    const u64 nwaves = (u64)gridDim.x * (UAES_WG / 64);
    for (u64 w = (u64)blockIdx.x * (UAES_WG / 64) + (threadIdx.x >> 6); w < nsectors * groups; w += nwaves) {
        const u64 sct = w / groups, r = w - sct * groups;
        const uint4 t0 = chunk_tw[sct * chunks_per_sector];
        Tw t;
        t.lo = t0.x | ((u64)t0.y << 32);
        t.hi = t0.z | ((u64)t0.w << 32);
        for (u32 i = 0; i < XTS_POW_N; ++i)
            if ((r >> i) & 1) t = wave_tw_mul(t, pw.lo[i], pw.hi[i], lane);       /* wave-uniform */
        t = tw_mul_a256(t, lane);                                                 /* * alpha^(256 lane): uaes_gf.h */
        const u64 c = r * 64 + lane;
        if (c != 0 && c < chunks_per_sector)
            chunk_tw[sct * chunks_per_sector + c] =
                make_uint4((u32)t.lo, (u32)(t.lo >> 32), (u32)t.hi, (u32)(t.hi >> 32));
    }
}

/* 16-byte accesses at addresses that are only known to be byte aligned (data units whose
 * size is not a multiple of 16 start at odd offsets from the second unit on): a packed
 * type tells the compiler the truth and it picks what the memory system allows.     */
struct __attribute__((packed, aligned(1))) Unaligned16 {
    u32 x, y, z, w;
};

template <bool ALIGNED>
__device__ __forceinline__ uint4 load16(const unsigned char *p)
{
    if (ALIGNED) return *(const uint4 *)p;
    const Unaligned16 v = *(const Unaligned16 *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <bool ALIGNED>
__device__ __forceinline__ void store16(unsigned char *p, uint4 v)
{
    if (ALIGNED) { *(uint4 *)p = v; return; }
    Unaligned16 u;
    u.x = v.x; u.y = v.y; u.z = v.z; u.w = v.w;
    *(Unaligned16 *)p = u;
}

/* main pass: one wave per chunk; lane l handles blocks l, l+64, l+128, l+192
 * of the chunk so each wave-level load/store is a contiguous 1 KiB segment.
 * The next chunk's data and tweak are requested before this chunk's rounds;
 * encryption runs the four blocks as two skewed pairs (enc_blocks_skewed).   */
/* in == out is allowed (every lane reads its blocks before it writes them): no __restrict__ */
/* PACKED (round 5): data units SHORTER than a chunk -- 64 B .. 4080 B, a multiple of 64 bytes; the 512-byte
 * sector is the classic one.  With a chunk per unit a 512-byte unit filled 32 of a wave's 256 block slots (140 GiB/s
 * against 1000 for 4 KiB units).  Here the text is one flat run of blocks cut into 256-block chunks whatever the unit
 * size, and lane l takes the four CONSECUTIVE blocks 4 l .. 4 l + 3 of the chunk: they lie in one unit (a unit is a
 * multiple of four blocks), so the lane needs one unit tweak T_s (a per-lane load from the pre-pass' table instead of
 * a wave-uniform one), one shift by the block's position in the unit and three doublings.  Its loads and stores are
 * 64 bytes apart from its neighbours' instead of 16 (a wave-level access is 4 KiB wide, a quarter of it used, the
 * four accesses of a chunk use all of it); the memory path has room for that in a kernel that needs 16 + 16 bytes
 * every eight cycles per CU.  Arguments then: chunks_per_sector = blocks per unit, main_blocks = blocks of the whole
 * text, (step_q, step_r) = (256 * waves) / and % blocks per unit, magic = ceil(2^24 / blocks per unit).           */
/* ALLFULL: the host knows that every chunk has all 256 blocks (units of a multiple of 4 KiB, the C3 shape): no block
 * count per chunk, no exec-masked stores */
template <int NR, bool DEC, bool ALIGNED, bool PACKED, bool ALLFULL = false>
__device__ __forceinline__ void xts_body(const uaesk_rk &k1, const uaesk_tables &tb,
                                         const uint4 *__restrict__ chunk_tw,
                                         u64 nsectors, u64 chunks_per_sector,
                                         u64 main_blocks,      /* whole blocks handled here, per unit */
                                         u64 sector_bytes,
                                         u64 step_q, u64 step_r,   /* (waves of the grid) / and % chunks_per_sector */
                                         const unsigned char *in,
                                         unsigned char *out,
                                         u64 nmain,            /* chunks [nmain, all) go by quarters (== all: none) */
                                         u32 magic)
{
    const u32 lane = threadIdx.x & 63u;
    if (DEC) fill_dec_tables(tb.td0); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    /* launched with 16 waves per workgroup, or 4 for short texts (more CUs, see launch_xts).  The wave's number is made
     * a SCALAR explicitly: everything that positions a chunk (unit, chunk in the unit, byte offset, block count) then
     * lives in SGPRs and is stepped by scalar instructions -- as vector arithmetic it was ~20 VALU instructions per
     * chunk in a kernel whose VALU is as busy as its LDS.                                        */
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);
    const u64 nchunks = nmain;                         /* the main loop's share; the rest follows behind it */
    const u32 lane16 = lane * 16u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));

    /* (unit, chunk inside the unit) of the chunk being fetched: one division here, then stepped by the
     * grid's wave count as (step_q, step_r) -- a 64-bit division per chunk cost ~8 % of the loop     */
    /* PACKED: (unit, block inside the unit) of the chunk's first block */
    u64 sctn = (PACKED ? wave * XTS_CHUNK : wave) / chunks_per_sector;
    u64 withn = (PACKED ? wave * XTS_CHUNK : wave) - sctn * chunks_per_sector;
    struct Fetched {
        uint4 d[UAES_U], tb;
        u32 cnt;
        u32 rb;                                      /* PACKED: (unit, position in it) of the chunk's first block */
        u64 sb;
        u64 off;                                     /* byte offset of the chunk's first block */
    };
    const u32 lane_off = PACKED ? lane * 64u : lane16, u_off = PACKED ? 16u : 1024u;
    auto fetch = [&](u64 ch, Fetched &f) {
        const u64 sct = sctn;
        const u64 first = PACKED ? ch * XTS_CHUNK : withn * XTS_CHUNK;
        const u32 rb = (u32)withn;
        sctn += step_q; withn += step_r;             /* the next chunk of this wave */
        if (withn >= chunks_per_sector) { withn -= chunks_per_sector; ++sctn; }
        const u64 left = main_blocks - first;
        f.cnt = ALLFULL ? XTS_CHUNK : (left < XTS_CHUNK ? (u32)left : XTS_CHUNK);
        f.off = PACKED ? first * 16 : sct * sector_bytes + first * 16;
        if (PACKED) { f.rb = rb; f.sb = sct; }                   /* the lane's unit tweak follows later (fetch_tw) */
        else f.tb = chunk_tw[ch];
        const unsigned char *src = in + f.off;
        if (ALIGNED) {
            /* a buffer resource over exactly the chunk's blocks: base in SGPRs, the lane's constant offset, and the
             * range check returns zeros for the blocks a short last chunk does not have (no clamping, no branch) */
            const u64 b = (u64)src;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                         (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)b)),
                0, (int)(f.cnt * 16u), 0x00020000);
#pragma unroll
            for (int u = 0; u < UAES_U; ++u) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off + u_off * u, 0, 0);
                f.d[u] = make_uint4(v.x, v.y, v.z, v.w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < UAES_U; ++u) {
                const u32 j = lane + 64u * u;
                const u32 jc = j < f.cnt ? j : (f.cnt ? f.cnt - 1 : 0);     /* clamped: no branch before the load */
                f.d[u] = f.cnt ? load16<ALIGNED>(src + 16u * jc) : make_uint4(0, 0, 0, 0);
            }
        }
    };

    /* PACKED: the unit tweak of the lane's four blocks, a per-lane load (four vector registers where the other
     * arrangement has four scalars); requested for the NEXT chunk between the two pairs of this one, when the first
     * pair's text registers are free -- with the text loads at the top the kernel does not fit 128 registers */
    auto lane_pos = [&](const Fetched &f, u32 &q) {
        const u32 x = f.rb + 4u * lane;                          /* < 512 */
        q = (x * magic) >> 24;                                   /* x / blocks per unit */
        return x - q * (u32)chunks_per_sector;
    };
    auto fetch_tw = [&](Fetched &f) {
        u32 q;
        (void)lane_pos(f, q);
        /* a buffer resource over the table from the chunk's first unit on: base in SGPRs, 16 q as the lane's offset,
         * zeros for lanes past the last unit */
        const u64 left = nsectors - f.sb;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(chunk_tw + f.sb), 0, (int)(left < 1024 ? left * 16u : 16384u), 0x00020000);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, q * 16u, 0, 0);
        f.tb = make_uint4(v.x, v.y, v.z, v.w);
    };

    /* one chunk: request the wave's next one into `nx`, then whiten, encrypt and store `cur`.  Expanded twice per loop
     * trip with the two buffers swapped: "next becomes current" is a renaming, not twenty register moves.  (Stores go
     * through plain global pointers: uaes_ctr.hip.h explains why not through a buffer resource.)               */
    auto body = [&](u64 ch, const Fetched &cur, Fetched &nx) {
        const u64 nxt = ch + nwaves;
        if (nxt < nchunks) fetch(nxt, nx);

        Tw t;
        t.lo = cur.tb.x | ((u64)cur.tb.y << 32);
        t.hi = cur.tb.z | ((u64)cur.tb.w << 32);
        if (PACKED) {
            u32 q;
            const u32 p0 = lane_pos(cur, q);
            t = tw_mul_pow(t, p0 & 63u);
            const Tw t64 = tw_mul_pow64(t);
            if (p0 & 64u) t = t64;
            const Tw t128 = tw_mul_pow64(tw_mul_pow64(t));
            if (p0 & 128u) t = t128;
        } else {
            t = tw_mul_pow(t, lane);
        }
        unsigned char *dst = out + cur.off;
        /* two skewed pairs one after the other: only one pair's states and tweaks are live at a time
         * (all four at once spilled 8-18 registers)                                                  */
#pragma unroll
        for (int pr = 0; pr < UAES_U; pr += 2) {
            u32 s[2][4], tw[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                tw[u][0] = (u32)t.lo; tw[u][1] = (u32)(t.lo >> 32);
                tw[u][2] = (u32)t.hi; tw[u][3] = (u32)(t.hi >> 32);
                s[u][0] = cur.d[pr + u].x ^ tw[u][0]; s[u][1] = cur.d[pr + u].y ^ tw[u][1];
                s[u][2] = cur.d[pr + u].z ^ tw[u][2]; s[u][3] = cur.d[pr + u].w ^ tw[u][3];
                t = PACKED ? tw_mul_pow(t, 1) : tw_mul_pow64(t);
            }
            if (DEC) dec_blocks_skewed<NR>(s[0], s[1], k1, lc);
            else enc_blocks_skewed<NR>(s[0], s[1], k1, lc);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32 j = PACKED ? 4u * lane + (u32)(pr + u) : lane + 64u * (pr + u);
                if (ALLFULL || j < cur.cnt)
                    store16<ALIGNED>(dst + 16u * j, make_uint4(s[u][0] ^ tw[u][0], s[u][1] ^ tw[u][1],
                                                               s[u][2] ^ tw[u][2], s[u][3] ^ tw[u][3]));
            }
        }
        if (PACKED && nxt < nchunks) fetch_tw(nx);
    };

    Fetched fa, fb;
    u64 ch = wave;
    if (ch < nchunks) { fetch(ch, fa); if (PACKED) fetch_tw(fa); }
    while (ch < nchunks) {
        body(ch, fa, fb);
        ch += nwaves;
        if (ch >= nchunks) break;
        body(ch, fb, fa);
        ch += nwaves;
    }

    /* A last round of chunks that covers only part of the grid's waves would leave the others idle for a whole chunk
     * (a round is 4096 waves x 4 KiB = 16 MiB: 18 MiB ran at 0.75 of the 16 MiB rate).  The host hands such a remainder
     * over as chunks [nmain, all); here they go by QUARTERS -- 64 blocks, one per lane -- over four times as many waves.
     * Quarter k of a chunk starts at its tweak * alpha^(64 k).                                                       */
    const u64 all = PACKED ? nmain : nsectors * chunks_per_sector;
    for (u64 q = wave; q < 4 * (all - nmain); q += nwaves) {
        const u64 c = nmain + (q >> 2);
        const u32 k = (u32)q & 3u;
        const u64 sct = c / chunks_per_sector, first = (c - sct * chunks_per_sector) * XTS_CHUNK;
        const u64 left = main_blocks - first;
        const u32 cnt = left < XTS_CHUNK ? (u32)left : XTS_CHUNK;
        const u32 j = 64u * k + lane;
        const uint4 tb4 = chunk_tw[c];
        Tw t;
        t.lo = tb4.x | ((u64)tb4.y << 32);
        t.hi = tb4.z | ((u64)tb4.w << 32);
        for (u32 i = 0; i < k; ++i) t = tw_mul_pow64(t);
        t = tw_mul_pow(t, lane);
        if (j < cnt) {
            const u64 off = sct * sector_bytes + (first + j) * 16;
            const uint4 d = load16<ALIGNED>(in + off);
            const u32 tw[4] = { (u32)t.lo, (u32)(t.lo >> 32), (u32)t.hi, (u32)(t.hi >> 32) };
            u32 s1[1][4] = { { d.x ^ tw[0], d.y ^ tw[1], d.z ^ tw[2], d.w ^ tw[3] } };
            if (DEC) dec_blocks<NR, 1>(s1, k1, lc); else enc_blocks<NR, 1>(s1, k1, lc);
            store16<ALIGNED>(out + off, make_uint4(s1[0][0] ^ tw[0], s1[0][1] ^ tw[1], s1[0][2] ^ tw[2], s1[0][3] ^ tw[3]));
        }
    }
}

template <int NR, bool DEC, bool ALIGNED, bool PACKED = false, bool ALLFULL = false>
__global__ __launch_bounds__(UAES_WG) void k_xts(uaesk_rk k1, uaesk_tables tb,
                                                 const uint4 *__restrict__ chunk_tw,
                                                 u64 nsectors, u64 chunks_per_sector, u64 main_blocks, u64 sector_bytes,
                                                 u64 step_q, u64 step_r, const unsigned char *in, unsigned char *out,
                                                 u64 nmain, u32 magic)
{
    xts_body<NR, DEC, ALIGNED, PACKED, ALLFULL>(k1, tb, chunk_tw, nsectors, chunks_per_sector, main_blocks, sector_bytes,
                                                step_q, step_r, in, out, nmain, magic);
}

/* ONE data unit of up to 64 chunks = 256 KiB (the reference API's call shape, a unit per call, micro_aes.c:1066-1093):
 * T0 = Enc_key2(tweak) is computed HERE, by every workgroup for itself, instead of by the k_xts_tweaks pre-pass, so a
 * call is one launch (4 KiB: 28.3 -> 22.5 us in round 2; 64 KiB 16.2 -> 11.7 us in round 3 when the kernel learnt to run
 * on several workgroups, 1 MiB 20.3 -> 13.9 us).  Wave 0 encrypts the tweak through an unreplicated 1 KiB copy of Te0 behind the cipher
 * tables (all lanes read the same entry: a broadcast, no conflicts; the decrypt direction has no Te tables in LDS)
 * while the other waves fill the tables; then one block per lane: wave w of the grid takes the 64-block runs
 * w, w + waves, ... of the unit, run q = 4c + k has the base tweak T0 * alpha^(256 c) * alpha^(64 k) (c = 64 g + l:
 * k_xts_expand's arithmetic, a wave-wide product per set bit of g and the sparse shifts of tw_mul_a256 for l).  The chunk
 * tweaks T0 * alpha^(256 c) are also written to chunk_tw for k_xts_cts.                                       */
#define XTS_SMALL_CHUNKS 1024u       /* 4 MiB: every wave of 256 CUs still has one run; longer units take the pre-pass
                                        and the bulk kernel (four blocks per lane, skewed pairs) */
#define XTS_SMALL_LDS    (UAES_LDS_ENC + 1024u + 16u)

/* MULTI: several units per call (a compile-time split: with both shapes in one kernel the tweak key stays live in the
 * loop beside the data key and the power table, and 85 scalar registers spill into the hot path of the one-unit call) */
template <int NR, bool DEC, bool MULTI>
__global__ __launch_bounds__(UAES_WG) void k_xts_small(uaesk_rk k1, uaesk_rk k2, uaesk_tables tb,
                                                       uint4 raw_tweak, u32 use_raw, u64 sector_id,
                                                       uint4 *__restrict__ chunk_tw, u64 chunks,
                                                       u64 main_blocks, const unsigned char *in, unsigned char *out,
                                                       uaesk_done done, XtsPow pw, u64 nsectors, u64 sector_bytes)
{
    /* nsectors == 1: the one unit described above.  nsectors > 1 (units of a multiple of 16 bytes, tweak = the unit's
     * number: a 64 KiB write to a disk of 4 KiB sectors): the same kernel with one more level -- run R of the call is
     * run R % rps of unit R / rps, and every WAVE encrypts the tweak of the unit its run lies in by itself (no
     * k_xts_tweaks pre-pass for a call of a few hundred units either) */
    u32 *te_plain = (u32 *)(uaes_lds + UAES_LDS_ENC);
    uint4 *t0_slot = (uint4 *)(uaes_lds + UAES_LDS_ENC + 1024u);
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    /* (run counts fit 32 bits by far -- launch_xts: at most 4 * XTS_SMALL_CHUNKS runs with blocks -- and one unit needs
     * no division at all: a 64-bit division is ~1.5 us of a lone wave) */
    const u32 runs_per_unit = (u32)((main_blocks + 63) >> 6), heads = (u32)(4 * chunks);   /* runs with blocks / runs that may head a chunk */
    const u32 rps = MULTI ? runs_per_unit : (runs_per_unit > heads ? runs_per_unit : heads);
    const u32 total = rps * (u32)nsectors;
    const u32 run0 = blockIdx.x * (UAES_WG / 64u) + wave;             /* this wave's first run: its text is requested now */
    uint4 d0 = make_uint4(0, 0, 0, 0);
    if (run0 < total) {
        const u32 u0 = MULTI ? run0 / rps : 0u, r0 = run0 - u0 * rps;
        if (64ull * r0 < main_blocks) {
            const u64 b0 = 64ull * r0 + lane;
            d0 = load16<true>(in + (u64)u0 * sector_bytes + 16u * (b0 < main_blocks ? b0 : main_blocks - 1));
        }
    }
    if (threadIdx.x >= blockDim.x - 256u) te_plain[threadIdx.x - (blockDim.x - 256u)] = tb.te0[threadIdx.x - (blockDim.x - 256u)];
    __syncthreads();
    if (wave == 0 && !MULTI) {                         /* the tweak's encryption, while the other waves fill the tables */
        u32 s[4];
        if (use_raw) { s[0] = raw_tweak.x; s[1] = raw_tweak.y; s[2] = raw_tweak.z; s[3] = raw_tweak.w; }
        else { s[0] = (u32)sector_id; s[1] = (u32)(sector_id >> 32); s[2] = 0; s[3] = 0; }   /* copyLint, micro_aes.c:399-404 */
        plain_encrypt<NR>(te_plain, k2, s);
        if (lane == 0) *t0_slot = make_uint4(s[0], s[1], s[2], s[3]);
    } else if (!MULTI) {
        fill_enc_tables_share(DEC ? tb.td0 : tb.te0, threadIdx.x - 64u, blockDim.x - 64u);
    } else {
        fill_enc_tables_share(DEC ? tb.td0 : tb.te0, threadIdx.x, blockDim.x);
    }
    const LaneConst lc = make_lane_const();
    __syncthreads();
    const u32 waves = gridDim.x * (UAES_WG / 64u);
    for (u32 run = run0; run < total; run += waves) {                     /* wave-uniform */
        const u32 unit = MULTI ? run / rps : 0u, ru = run - unit * rps;
        Tw t;
        if (!MULTI) {
            const uint4 t4 = *t0_slot;
            t.lo = t4.x | ((u64)t4.y << 32);
            t.hi = t4.z | ((u64)t4.w << 32);
        } else {                                                          /* T0 of this run's unit, by this wave */
            const u64 sid = sector_id + unit;
            u32 s[4] = { (u32)sid, (u32)(sid >> 32), 0, 0 };
            plain_encrypt<NR>(te_plain, k2, s);
            t.lo = s[0] | ((u64)s[1] << 32);
            t.hi = s[2] | ((u64)s[3] << 32);
        }
        const u32 c = (u32)__builtin_amdgcn_readfirstlane((int)(ru >> 2)), k = ru & 3u;
        for (u32 i = 0; (c >> 6) >> i; ++i)                               /* chunk 64 g + l: alpha^(2^14 g) by the wave, */
            if (((c >> 6) >> i) & 1) t = wave_tw_mul(t, pw.lo[i], pw.hi[i], lane);
        if (c & 63u) t = tw_mul_a256(t, c & 63u);                          /* alpha^(256 l) by sparse shifts (uaes_gf.h)   */
        for (u32 q = 0; q < k; ++q) t = tw_mul_pow64(t);
        if (!MULTI && k == 0 && c < chunks && lane == 0)          /* chunk tweaks for k_xts_cts */
            chunk_tw[c] = make_uint4((u32)t.lo, (u32)(t.lo >> 32), (u32)t.hi, (u32)(t.hi >> 32));
        if (64ull * ru >= main_blocks) continue;
        const u64 blk = 64ull * ru + lane;
        const Tw tl = tw_mul_pow(t, lane);
        const u32 tw[4] = { (u32)tl.lo, (u32)(tl.lo >> 32), (u32)tl.hi, (u32)(tl.hi >> 32) };
        const u64 bc = blk < main_blocks ? blk : main_blocks - 1;        /* clamped: no branch around the rounds */
        const unsigned char *src = in + (u64)unit * sector_bytes;
        const uint4 d = run == run0 ? d0 : load16<true>(src + 16u * bc);
        u32 s[1][4] = { { d.x ^ tw[0], d.y ^ tw[1], d.z ^ tw[2], d.w ^ tw[3] } };
        if (DEC) dec_blocks<NR, 1>(s, k1, lc); else enc_blocks<NR, 1>(s, k1, lc);
        if (blk < main_blocks)
            store16<true>(out + (u64)unit * sector_bytes + 16u * blk,
                          make_uint4(s[0][0] ^ tw[0], s[0][1] ^ tw[1], s[0][2] ^ tw[2], s[0][3] ^ tw[3]));
    }
    ticket_release(done);
}

/* ciphertext stealing (micro_aes.c:1037-1053): one thread per data unit handles
 * the last whole block m and the r-byte tail.  All inputs are read before any
 * output is written, so in == out is fine.                                  */
template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_xts_cts(uaesk_rk k1, uaesk_tables tb,
                                                     const uint4 *__restrict__ chunk_tw,
                                                     u64 nsectors, u64 chunks_per_sector,
                                                     u64 m, u32 r, u64 sector_bytes,
                                                     const unsigned char *in,
                                                     unsigned char *out)
{
    if (DEC) fill_dec_tables(tb.td0); else fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG;
    for (u64 sct = (u64)blockIdx.x * UAES_WG + threadIdx.x; sct < nsectors; sct += stride) {
        /* tweak of block m: chunk base * alpha^(m mod 256) */
        const uint4 tb4 = chunk_tw[sct * chunks_per_sector + m / XTS_CHUNK];
        Tw tm;
        tm.lo = tb4.x | ((u64)tb4.y << 32);
        tm.hi = tb4.z | ((u64)tb4.w << 32);
        const u32 off = (u32)(m % XTS_CHUNK);
        tm = tw_mul_pow(tm, off & 63u);
        for (u32 q = 0; q < (off >> 6); ++q) tm = tw_mul_pow64(tm);
        const Tw tn = tw_mul_pow(tm, 1);
        /* encrypt: block m uses T_m, the stolen block T_{m+1}; decrypt swaps (:1041) */
        const Tw ta = DEC ? tn : tm, tbk = DEC ? tm : tn;

        const unsigned char *src = in + sct * sector_bytes + m * 16;
        unsigned char *dst = out + sct * sector_bytes + m * 16;
        const uint4 d = load16<false>(src);          /* units of ragged size start at odd addresses */
        unsigned char tail[16];
        for (u32 i = 0; i < r; ++i) tail[i] = src[16 + i];

        u32 s[1][4];
        s[0][0] = d.x ^ (u32)ta.lo; s[0][1] = d.y ^ (u32)(ta.lo >> 32);
        s[0][2] = d.z ^ (u32)ta.hi; s[0][3] = d.w ^ (u32)(ta.hi >> 32);
        if (DEC) dec_blocks<NR, 1>(s, k1, lc); else enc_blocks<NR, 1>(s, k1, lc);
        u32 cc[4] = { s[0][0] ^ (u32)ta.lo, s[0][1] ^ (u32)(ta.lo >> 32),
                      s[0][2] ^ (u32)ta.hi, s[0][3] ^ (u32)(ta.hi >> 32) };
        /* second block = tail bytes followed by cc[r..16) */
        u32 p[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            u32 v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const u32 i = 4 * w + b;
                const u32 byte = i < r ? tail[i] : (cc[w] >> (8 * b)) & 0xffu;
                v |= byte << (8 * b);
            }
            p[w] = v;
        }
        s[0][0] = p[0] ^ (u32)tbk.lo; s[0][1] = p[1] ^ (u32)(tbk.lo >> 32);
        s[0][2] = p[2] ^ (u32)tbk.hi; s[0][3] = p[3] ^ (u32)(tbk.hi >> 32);
        if (DEC) dec_blocks<NR, 1>(s, k1, lc); else enc_blocks<NR, 1>(s, k1, lc);
        store16<false>(dst, make_uint4(s[0][0] ^ (u32)tbk.lo, s[0][1] ^ (u32)(tbk.lo >> 32),
                                       s[0][2] ^ (u32)tbk.hi, s[0][3] ^ (u32)(tbk.hi >> 32)));
        for (u32 i = 0; i < r; ++i) dst[16 + i] = (unsigned char)(cc[i >> 2] >> (8 * (i & 3)));
    }
}

/* ------------------------------------------------------------------------ */
/* self test of the primitives (diagnostics for a fresh box)                  */
/* ------------------------------------------------------------------------ */
__global__ __launch_bounds__(UAES_WG) void k_selftest(uaesk_rk ek, uaesk_rk dk, uaesk_tables tb,
                                                      unsigned *result)
{
    unsigned bad = 0;
    /* v_perm_b32 byte numbering: selector n picks byte n of {S0,S1}, S1 low */
    if (__builtin_amdgcn_perm(0x44332211u, 0x88776655u, 0x07040300u) != 0x44118855u) bad |= 1;
    if (__builtin_amdgcn_perm(0x44332211u, 0x88776655u, 0x0c0c0c0cu) != 0u) bad |= 2;
    if (bswap32(0x11223344u) != 0x44332211u) bad |= 4;
    if (rotl32(0x80000001u, 8) != 0x00000180u) bad |= 8;
    /* the table lookups address LDS absolutely: dynamic segment must start at 0 */
    if ((u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)uaes_lds != 0u) bad |= 256;
    if (xor3(0xf0f0f0f0u, 0xccccccccu, 0xaaaaaaaau) != 0x96969696u) bad |= 512;
    if (or_xor(0xf0f0f0f0u, 0xccccccccu, 0xaaaaaaaau) != 0x56565656u) bad |= 1024;

    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    /* FIPS-197 appendix C.1: key 00..0f, pt 00112233..ff */
    u32 s[1][4] = { { 0x33221100u, 0x77665544u, 0xbbaa9988u, 0xffeeddccu } };
    enc_blocks<10, 1>(s, ek, lc);
    if (s[0][0] != 0xd8e0c469u || s[0][1] != 0x30047b6au || s[0][2] != 0x80b7cdd8u || s[0][3] != 0x5ac5b470u) bad |= 16;
    __syncthreads();
    fill_dec_tables(tb.td0);
    dec_blocks<10, 1>(s, dk, lc);
    if (s[0][0] != 0x33221100u || s[0][1] != 0x77665544u || s[0][2] != 0xbbaa9988u || s[0][3] != 0xffeeddccu) bad |= 32;
    /* XTS tweak arithmetic: alpha^k by shifting == k doublings */
    Tw t = { 0x8000000000000001ull * (threadIdx.x + 1), 0xC3A5C85C97CB3127ull ^ threadIdx.x };
    Tw a = t;
    const u32 k = threadIdx.x & 63u;
    for (u32 i = 0; i < k; ++i) a = tw_mul_pow(a, 1);
    Tw b = tw_mul_pow(t, k);
    if (a.lo != b.lo || a.hi != b.hi) bad |= 64;
    a = t;
    for (u32 i = 0; i < 64; ++i) a = tw_mul_pow(a, 1);
    b = tw_mul_pow64(t);
    if (a.lo != b.lo || a.hi != b.hi) bad |= 128;
    /* sixteen lanes per block (row_encrypt: DPP row_ror / quad_perm directions, table slots) */
    __syncthreads();
    row_fill_tables(tb.te0, ek);
    {
        const RowLane<10> L = row_lane<10>();
        const u32 pt[4] = { 0x33221100u, 0x77665544u, 0xbbaa9988u, 0xffeeddccu };
        const u32 ct[4] = { 0xd8e0c469u, 0x30047b6au, 0x80b7cdd8u, 0x5ac5b470u };
        if (row_encrypt<10>(row_pick(pt, L.c), L) != row_pick(ct, L.c)) bad |= 2048;
    }
    if (bad) atomicOr(result, bad);
}

/* ------------------------------------------------------------------------ */
/* launchers                                                                  */
/* ------------------------------------------------------------------------ */
static int g_cus = 0;

extern "C" int uaesk_device_info(int *cu_count, int *lds_bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    g_cus = p.multiProcessorCount;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    return 0;
}

static unsigned grid_for(u64 work_items, u64 per_wg)
{
    if (g_cus <= 0) uaesk_device_info(nullptr, nullptr);
    u64 want = (work_items + per_wg - 1) / per_wg;
    if (want < 1) want = 1;
    const u64 cap = (u64)(g_cus > 0 ? g_cus : 256);
    return (unsigned)(want < cap ? want : cap);
}

/* > 64 KiB of dynamic LDS needs the attribute; set it once per (kernel, device): hipFuncSetAttribute takes
 * a runtime-wide lock, which the host threads of the synchronous API would otherwise meet on every call.
 * Keyed by the kernel's address (instantiations share C++ types, so a per-type static would be wrong),
 * open addressing on a few address bits.  A racing first call merely sets the attribute twice; a full
 * neighbourhood falls back to setting it every time. */
hipError_t uaesk_want_lds(const void *kern, unsigned bytes)
{
    enum { SLOTS = 512, PROBE = 16 };
    static const void *volatile seen[16][SLOTS];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    const unsigned h = (unsigned)(((uintptr_t)kern >> 3) * 2654435761u) >> 16;
    int free_slot = -1;
    for (int i = 0; i < PROBE; ++i) {
        const unsigned s = (h + i) % SLOTS;
        const void *v = seen[dev][s];
        if (v == kern) return hipSuccess;
        if (!v) { free_slot = (int)s; break; }
    }
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && free_slot >= 0) seen[dev][free_slot] = kern;
    return e;
}

template <typename K>
static hipError_t set_lds(K kern, unsigned bytes)
{
    return uaesk_want_lds((const void *)kern, bytes);
}

#define DISPATCH_NR(nr, CALL)                         \
    switch (nr) {                                     \
    case 10: { constexpr int NR = 10; CALL; } break;  \
    case 12: { constexpr int NR = 12; CALL; } break;  \
    case 14: { constexpr int NR = 14; CALL; } break;  \
    default: return (int)hipErrorInvalidValue;        \
    }

/* ------------------------------------------------------------------------ */
/* the table of arrangements (uaes_plan.h): switch, names, ECB / CTR / XTS planners */
/* ------------------------------------------------------------------------ */
static std::atomic<unsigned> g_plan_disabled{ ~0u };
extern "C" unsigned uaesk_plan_disabled(void)
{
    unsigned m = g_plan_disabled.load(std::memory_order_relaxed);
    if (m == ~0u) {
        const char *e = getenv("UAES_PLAN_DISABLE");
        m = (e && *e) ? (unsigned)strtoul(e, nullptr, 0) & 0x7fffffffu : 0u;
        g_plan_disabled.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" void uaesk_plan_disable(unsigned mask) { g_plan_disabled.store(mask & 0x7fffffffu, std::memory_order_relaxed); }
bool uaesk_arr_on(int id) { return !((uaesk_plan_disabled() >> id) & 1u); }

extern "C" const char *uaesk_arrangement_name(int id)
{
    static const char *const names[UAES_ARR_COUNT] = {
        "ecb.single", "ecb.tiled", "ctr.single", "ctr.quad", "ctr.striped", "xts.small", "xts.packed", "xts.bulk",
        "gcm.small", "gcm.chunks", "gcm.twophase", "gcm.striped", "gcm.levels", "ocb.small", "ocb.runs",
        "siv.small", "siv.chunks", "siv.levels" };
    return id >= 0 && id < UAES_ARR_COUNT ? names[id] : "?";
}

/* short texts: one block per lane, so that up to four times as many CUs take part */
static bool short_text(u64 nblocks)
{
    return grid_for(nblocks, (u64)UAES_WG * UAES_U) * 2 <= grid_for(~0ull, 1);
}

static uaes_plan plan_ecb(u64 items)
{
    uaes_plan p = { UAES_ARR_ECB_TILED, 1, 0, 0 };
    if (short_text(items) && uaesk_arr_on(UAES_ARR_ECB_SINGLE)) { p.arrangement = UAES_ARR_ECB_SINGLE; p.grid = grid_for(items, UAES_WG); }
    else p.grid = grid_for(items, (u64)UAES_WG * UAES_U);
    return p;
}

/* CTR.  The striped kernel (k_ctr_shared2: rounds 1-2 shared by the 256 counters of a group, eight groups per stripe)
 * from ONE grid of stripes on: below that the generic kernel's one- or four-block work items spread better over the
 * CUs (8 MiB on 256 CUs; profiles/HISTORY.md "CTR size sweep").  A last round of stripes that covers less than
 * CTR_TAIL_PCT % of the grid is handed to the kernel's edge path -- one block per thread through the plain rounds,
 * spread over ALL workgroups (20 MiB = 640 stripes on 256 workgroups: 2.5 rounds would cost 3).  g_lo / n8: the
 * stripes' first group and their number (uaes_ctr.hip.h). */
#define CTR_TAIL_PCT 80u
static uaes_plan plan_ctr(const uaesk_ctr *ctr, size_t len, u64 *g_lo_out, u64 *n8_out)
{
    const u64 nfull = len / 16, nblocks = (len + 15) / 16;
    uaes_plan p = { UAES_ARR_CTR_QUAD, 1, 0, 0 };
    const unsigned grid = grid_for(~0ull, 1);
    if (!ctr->le32 && uaesk_arr_on(UAES_ARR_CTR_STRIPED)) {      /* (the shared rounds assume the 56-bit big-endian counter) */
        const u32 c0 = (u32)ctr->v0 & 0xffu;
        const u64 g_lo = c0 ? 1 : 0, groups = (c0 + nfull) / 256;
        u64 n8 = groups > g_lo ? (groups - g_lo) / 8 : 0;
        if (n8 >= (u64)grid) {
            const u64 r = n8 % grid;
            if (r && n8 > grid && r * 100 < (u64)grid * CTR_TAIL_PCT) n8 -= r;
            p.arrangement = UAES_ARR_CTR_STRIPED; p.grid = grid;
            /* a text in which counter bits 40..47 move -- once in 2^40 blocks -- is cut there into two launches */
            if (ctr_stripes_cross_a(ctr, g_lo, n8)) p.launches = 2;
            if (g_lo_out) *g_lo_out = g_lo;
            if (n8_out) *n8_out = n8;
            return p;
        }
    }
    if (short_text(nblocks) && uaesk_arr_on(UAES_ARR_CTR_SINGLE)) { p.arrangement = UAES_ARR_CTR_SINGLE; p.grid = grid_for(nblocks, UAES_WG); }
    else p.grid = grid_for(nblocks, (u64)UAES_WG * UAES_U);
    return p;
}

template <int NR, bool DEC>
static int launch_ecb(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *keys,
                      const void *in, void *out, size_t nfull, unsigned rem, unsigned padding)
{
    const unsigned lds = DEC ? UAES_LDS_DEC : UAES_LDS_ENC;
    hipError_t e = set_lds(k_ecb<NR, DEC, UAES_U>, lds);
    if (e == hipSuccess) e = set_lds(k_ecb<NR, DEC, 1>, lds);
    if (e != hipSuccess) return (int)e;
    const u64 items = nfull + ((rem || padding) ? 1 : 0);
    const uaesk_done done = uaesk_ticket_take();              /* the call's only kernel: it carries the ticket */
    const uaes_plan pl = plan_ecb(items);
    if (pl.arrangement == UAES_ARR_ECB_SINGLE) {
        hipLaunchKernelGGL((k_ecb<NR, DEC, 1>), dim3(pl.grid), dim3(UAES_WG), lds, st, *keys, *tb,
                           (const uint4 *)in, (uint4 *)out, (u64)nfull, (u32)rem, (u32)padding, done, (u64)nfull);
    } else {
        const unsigned grid = pl.grid;
        const u64 tile = (u64)UAES_WG * UAES_U, per_round = tile * grid;
        const u64 rounds = (u64)nfull / per_round, left = (u64)nfull - rounds * per_round;
        /* the remainder of the last round as single blocks when it covers less than 80 % of the grid */
        const u64 tail_from = (rounds && left && left * 100 < per_round * 80) ? rounds * per_round : (u64)nfull;
        hipLaunchKernelGGL((k_ecb<NR, DEC, UAES_U>), dim3(grid), dim3(UAES_WG), lds, st,
                           *keys, *tb, (const uint4 *)in, (uint4 *)out, (u64)nfull, (u32)rem, (u32)padding, done, tail_from);
    }
    return (int)hipGetLastError();
}

extern "C" int uaesk_ecb(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *keys,
                         int decrypt, const void *in, void *out, size_t nfull, unsigned rem, unsigned padding)
{
    if (decrypt) padding = 0;
    if (nfull == 0 && rem == 0 && padding == 0) return 0;
    if (padding > 2) return (int)hipErrorInvalidValue;
    if (decrypt) { DISPATCH_NR(nr, return (launch_ecb<NR, true>(S(stream), tb, keys, in, out, nfull, 0, 0))); }
    else         { DISPATCH_NR(nr, return (launch_ecb<NR, false>(S(stream), tb, keys, in, out, nfull, rem, padding))); }
    return 0;
}

template <int NR, int U>
static int launch_ctr_u(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *ctr,
                        const void *in, void *out, size_t len, const int *gate)
{
    const unsigned lds = UAES_LDS_ENC;
    hipError_t e = set_lds(k_ctr<NR, U>, lds);
    if (e != hipSuccess) return (int)e;
    const u64 nfull = len / 16;
    const u32 rem = (u32)(len % 16);
    const unsigned grid = grid_for(nfull + (rem ? 1 : 0), (u64)UAES_WG * U);
    /* an armed ticket rides on an ungated launch only (the gated one is the second half of a GCM decryption,
     * which never arms) */
    uaesk_done done = { nullptr, nullptr, 0 };
    if (!gate) done = uaesk_ticket_take();
    hipLaunchKernelGGL((k_ctr<NR, U>), dim3(grid), dim3(UAES_WG), lds, st, *ek, *tb, *ctr,
                       (const uint4 *)in, (uint4 *)out, nfull, rem, gate, done);
    return (int)hipGetLastError();
}

/* one launch of the striped kernel over a text whose stripes do not cross a 2^40-block boundary */
template <int NR>
static int launch_ctr_striped(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *ctr,
                              const void *in, void *out, size_t len, const int *gate, unsigned grid, u64 g_lo, u64 n8)
{
    hipError_t e = set_lds(k_ctr_shared2<NR>, UAES_LDS_CTRS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_ctr_shared2<NR>), dim3(grid), dim3(UAES_WG), UAES_LDS_CTRS, st, *ek, *tb, *ctr,
                       (const uint4 *)in, (uint4 *)out, g_lo, n8, (u64)(len / 16), (u32)(len % 16), gate);
    return (int)hipGetLastError();
}

template <int NR>
static int launch_ctr(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *ctr,
                      const void *in, void *out, size_t len, const int *gate)
{
    u64 g_lo = 0, n8 = 0;
    const uaes_plan pl = plan_ctr(ctr, len, &g_lo, &n8);
    if (pl.arrangement == UAES_ARR_CTR_SINGLE) return launch_ctr_u<NR, 1>(st, tb, ek, ctr, in, out, len, gate);
    if (pl.arrangement == UAES_ARR_CTR_QUAD) return launch_ctr_u<NR, UAES_U>(st, tb, ek, ctr, in, out, len, gate);
    if (pl.launches == 2) {
        /* the striped kernel makes its lane constants once per launch (uaes_ctr.hip.h): cut at the block where
         * counter bits 40..47 move; each half is planned on its own */
        const u64 nb = ((u64)1 << 40) - (ctr->v0 & (((u64)1 << 40) - 1));          /* blocks up to the boundary */
        if (nb == 0 || nb >= len / 16) return launch_ctr_u<NR, UAES_U>(st, tb, ek, ctr, in, out, len, gate);   /* (cannot happen) */
        uaesk_ctr second = *ctr;
        second.v0 = (ctr->v0 + nb) & 0x00FFFFFFFFFFFFFFull;
        int rc = launch_ctr<NR>(st, tb, ek, ctr, in, out, (size_t)(nb * 16), gate);
        if (rc) return rc;
        return launch_ctr<NR>(st, tb, ek, &second, (const unsigned char *)in + nb * 16, (unsigned char *)out + nb * 16,
                              len - (size_t)(nb * 16), gate);
    }
    return launch_ctr_striped<NR>(st, tb, ek, ctr, in, out, len, gate, pl.grid, g_lo, n8);
}

template <int NR, int U>
static int launch_ctr_ind_u(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *d_rk, const uaesk_ctr *d_ctr,
                            const void *in, void *out, size_t len)
{
    const unsigned lds = UAES_LDS_ENC;
    hipError_t e = set_lds(k_ctr_ind<NR, U>, lds);
    if (e != hipSuccess) return (int)e;
    const u64 nfull = len / 16;
    const u32 rem = (u32)(len % 16);
    const unsigned grid = grid_for(nfull + (rem ? 1 : 0), (u64)UAES_WG * U);
    hipLaunchKernelGGL((k_ctr_ind<NR, U>), dim3(grid), dim3(UAES_WG), lds, st, d_rk, *tb, d_ctr,
                       (const uint4 *)in, (uint4 *)out, nfull, rem);
    return (int)hipGetLastError();
}

/* the generic CTR kernel with key schedule and counter description read from device memory (k_ctr_ind) */
extern "C" int uaesk_ctr_xcrypt_ind(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *d_rk,
                                    const uaesk_ctr *d_ctr, const void *in, void *out, size_t len)
{
    if (len == 0) return 0;
    if (short_text((len + 15) / 16)) { DISPATCH_NR(nr, return (launch_ctr_ind_u<NR, 1>(S(stream), tb, d_rk, d_ctr, in, out, len))); }
    else                             { DISPATCH_NR(nr, return (launch_ctr_ind_u<NR, 4>(S(stream), tb, d_rk, d_ctr, in, out, len))); }
    return 0;
}

extern "C" int uaesk_ctr_xcrypt(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                                const uaesk_ctr *ctr, const void *in, void *out, size_t len,
                                const int *gate)
{
    if (len == 0) return 0;
    DISPATCH_NR(nr, return (launch_ctr<NR>(S(stream), tb, ek, ctr, in, out, len, gate)));
    return 0;
}

#define XTS_SERIAL_CPS 128u      /* up to this many chunks per unit the pre-pass thread walks them itself */

/* alpha^(2^(14+i)), i < XTS_POW_N: repeated squaring from alpha^(2^14) = 256 word shifts of 1 */
static const XtsPow *xts_pow_table()
{
    static XtsPow tab;
    static std::once_flag once;
    std::call_once(once, [] {
        Tw e = { 1, 0 };
        for (int i = 0; i < 256; ++i) e = tw_mul_pow64(e);
        for (int i = 0; i < XTS_POW_N; ++i) {
            tab.lo[i] = e.lo; tab.hi[i] = e.hi;
            Tw a = e, acc = { 0, 0 };
            for (int b = 0; b < 128; ++b) {          /* e * e, bit-serial */
                if (((b < 64 ? e.lo : e.hi) >> (b & 63)) & 1) { acc.lo ^= a.lo; acc.hi ^= a.hi; }
                a = tw_mul_pow(a, 1);
            }
            e = acc;
        }
    });
    return &tab;
}

static void xts_geometry(size_t sector_bytes, u64 *main_blocks, u32 *rem, u64 *cps)
{
    const u32 r = (u32)(sector_bytes % 16);
    const u64 whole = sector_bytes / 16;
    const u64 mb = whole - (r ? 1 : 0);        /* blocks before the stealing pair */
    u64 c = (whole + XTS_CHUNK - 1) / XTS_CHUNK;   /* covers block m as well */
    if (c == 0) c = 1;
    *main_blocks = mb; *rem = r; *cps = c;
}

extern "C" size_t uaesk_xts_scratch_bytes(size_t sector_bytes, size_t nsectors)
{
    u64 mb, cps; u32 r;
    xts_geometry(sector_bytes, &mb, &r, &cps);
    return (size_t)(cps * nsectors * 16);
}

/* XTS (uaes_plan.h).  XTS_SMALL: one unit of up to 8 MiB (one block per lane and a run loop: 5 MiB 27.2 -> 23.8 us,
 * 8 MiB 31.5 -> 28.5, 12 MiB 32.3 against 39.8 for the pre-pass + bulk kernel), or up to 4 MiB of whole-block units
 * numbered from first_sector, in ONE launch.  XTS_PACKED: units shorter than a chunk (whole blocks, a multiple of
 * four) in the flat, packed arrangement.  XTS_BULK: the tweak pre-pass and the chunk kernel -- everything else, the C3
 * shape (2^20 sectors of 4 KiB) included.  (Round 5 had a fourth arrangement between SMALL and BULK for 4 KiB sectors up
 * to 128 MiB, the tweaks made inside the chunk kernel: -22 % .. +3.9 % against BULK on two boxes, deleted in round 6:
 * profiles/r06_plan_ab.log.) */
static uaes_plan plan_xts(size_t sector_bytes, size_t nsectors, bool explicit_tweak)
{
    u64 mb, cps; u32 r;
    xts_geometry(sector_bytes, &mb, &r, &cps);
    uaes_plan p = { UAES_ARR_XTS_BULK, 0, 0, 0 };
    const bool one_unit = nsectors == 1 && mb > 0 && cps <= 2ull * XTS_SMALL_CHUNKS;
    const bool few_units = nsectors > 1 && !explicit_tweak && r == 0 && mb > 0 &&
                           (u64)nsectors * ((mb + 63) / 64) <= 4ull * XTS_SMALL_CHUNKS;
    if ((one_unit || few_units) && uaesk_arr_on(UAES_ARR_XTS_SMALL)) {
        p.arrangement = UAES_ARR_XTS_SMALL; p.launches = r ? 2 : 1;
        p.grid = grid_for((u64)nsectors * ((mb + 63) / 64), UAES_WG / 64);
        return p;
    }
    const bool serial = cps <= XTS_SERIAL_CPS;
    p.launches = 1 + (serial ? 0 : 1) + (mb > 0 ? 1 : 0) + (r ? 1 : 0);
    if (!explicit_tweak && r == 0 && mb >= 4 && mb < XTS_CHUNK && mb % 4 == 0 && nsectors > 1 && uaesk_arr_on(UAES_ARR_XTS_PACKED)) {
        p.arrangement = UAES_ARR_XTS_PACKED; p.launches = 2;
    }
    return p;
}

template <int NR, bool DEC>
static int launch_xts(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k1, const uaesk_rk *k2,
                      const uint8_t *tweak16, u64 first_sector, size_t sector_bytes, size_t nsectors,
                      const void *in, void *out, void *scratch)
{
    u64 mb, cps; u32 r;
    xts_geometry(sector_bytes, &mb, &r, &cps);
    const unsigned lds = DEC ? UAES_LDS_DEC : UAES_LDS_ENC;
    hipError_t e = set_lds(k_xts_tweaks<NR>, UAES_LDS_ENC);
    if (e == hipSuccess) e = set_lds(k_xts<NR, DEC, true>, lds);
    if (e == hipSuccess) e = set_lds(k_xts<NR, DEC, false>, lds);
    if (e == hipSuccess) e = set_lds(k_xts_cts<NR, DEC>, lds);
    if (e != hipSuccess) return (int)e;

    uint4 raw = make_uint4(0, 0, 0, 0);
    if (tweak16) memcpy(&raw, tweak16, 16);
    const uaes_plan pl = plan_xts(sector_bytes, nsectors, tweak16 != nullptr);
    const bool few_units = pl.arrangement == UAES_ARR_XTS_SMALL && nsectors > 1;     /* (one unit, or several whole-block ones) */
    if (pl.arrangement == UAES_ARR_XTS_SMALL) {
        e = few_units ? set_lds((k_xts_small<NR, DEC, true>), XTS_SMALL_LDS) : set_lds((k_xts_small<NR, DEC, false>), XTS_SMALL_LDS);
        if (e != hipSuccess) return (int)e;
        uaesk_done done = { nullptr, nullptr, 0 };
        if (!r) done = uaesk_ticket_take();                   /* no stealing kernel behind it: it carries the ticket */
        /* a workgroup per 1024 blocks (one per lane), as many as half the CUs: a 64 KiB unit runs on four CUs at the
         * latency of one block */
        const unsigned sgrid = pl.grid;
        if (few_units)
            hipLaunchKernelGGL((k_xts_small<NR, DEC, true>), dim3(sgrid), dim3(UAES_WG), XTS_SMALL_LDS, st, *k1, *k2, *tb, raw,
                               0u, first_sector, (uint4 *)scratch, cps, mb,
                               (const unsigned char *)in, (unsigned char *)out, done, *xts_pow_table(), (u64)nsectors,
                               (u64)sector_bytes);
        else
            hipLaunchKernelGGL((k_xts_small<NR, DEC, false>), dim3(sgrid), dim3(UAES_WG), XTS_SMALL_LDS, st, *k1, *k2, *tb, raw,
                               (u32)(tweak16 != nullptr), first_sector, (uint4 *)scratch, cps, mb,
                               (const unsigned char *)in, (unsigned char *)out, done, *xts_pow_table(), (u64)1,
                               (u64)sector_bytes);
        if (r)
            hipLaunchKernelGGL((k_xts_cts<NR, DEC>), dim3(1), dim3(UAES_WG), lds, st,
                               *k1, *tb, (const uint4 *)scratch, (u64)1, cps, mb, r, (u64)sector_bytes,
                               (const unsigned char *)in, (unsigned char *)out);
        return (int)hipGetLastError();
    }
    const bool serial = cps <= XTS_SERIAL_CPS;
    if (!serial && ((cps + 63) / 64) >> XTS_POW_N) return (int)hipErrorInvalidValue;
    if (nsectors <= 65536) {                 /* few units: one-wave workgroups on a plain table (k_xts_tweaks) */
        if ((e = set_lds((k_xts_tweaks<NR, true>), 1024u)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_xts_tweaks<NR, true>), dim3((unsigned)((nsectors + 63) / 64)), dim3(64), 1024u, st,
                           *k2, *tb, raw, (u32)(tweak16 != nullptr), first_sector, (u64)nsectors, cps, (u32)serial,
                           (uint4 *)scratch);
    } else {
        hipLaunchKernelGGL((k_xts_tweaks<NR>), dim3(grid_for(nsectors, UAES_WG)), dim3(UAES_WG), UAES_LDS_ENC, st,
                           *k2, *tb, raw, (u32)(tweak16 != nullptr), first_sector, (u64)nsectors, cps, (u32)serial,
                           (uint4 *)scratch);
    }
    if (!serial)
        hipLaunchKernelGGL(k_xts_expand, dim3(grid_for((u64)nsectors * ((cps + 63) / 64), UAES_WG / 64)),
                           dim3(UAES_WG), 0, st, *xts_pow_table(), (u64)nsectors, cps, (uint4 *)scratch);
    /* units shorter than a chunk (whole blocks, a multiple of four): the flat, packed arrangement (k_xts) */
    if (pl.arrangement == UAES_ARR_XTS_PACKED) {
        if ((e = set_lds((k_xts<NR, DEC, true, true>), lds)) != hipSuccess) return (int)e;
        const u64 total = (u64)nsectors * mb, nchunks = (total + XTS_CHUNK - 1) / XTS_CHUNK;
        const unsigned wg = grid_for(nchunks, UAES_WG / 64) * 2 <= grid_for(~0ull, 1) ? 256u : UAES_WG;
        const unsigned xgrid = grid_for(nchunks, wg / 64);
        const u64 adv = (u64)xgrid * (wg / 64) * XTS_CHUNK;
        hipLaunchKernelGGL((k_xts<NR, DEC, true, true>), dim3(xgrid), dim3(wg), lds, st,
                           *k1, *tb, (const uint4 *)scratch, (u64)nsectors, mb, total, (u64)sector_bytes, adv / mb, adv % mb,
                           (const unsigned char *)in, (unsigned char *)out, nchunks, (u32)(((1u << 24) + mb - 1) / mb));
        return (int)hipGetLastError();
    }
    if (mb > 0) {
        const u64 nchunks = (u64)nsectors * ((mb + XTS_CHUNK - 1) / XTS_CHUNK);
        /* one wave per 256-block chunk: a short text on 16-wave workgroups would sit on a few CUs,
         * so below half a GPU's worth of chunks the workgroups shrink to 4 waves              */
        const unsigned wg = grid_for(nchunks, UAES_WG / 64) * 2 <= grid_for(~0ull, 1) ? 256u : UAES_WG;
        /* every block address is a multiple of 16 unless units of ragged size follow one another */
        const unsigned xgrid = grid_for(nchunks, wg / 64);
        const u64 nwaves = (u64)xgrid * (wg / 64), step_q = nwaves / cps, step_r = nwaves % cps;
        /* the remainder of the last round by quarter chunks when it covers less than 80 % of the waves (k_xts) */
        const u64 pct = 80;                                   /* (as CTR_TAIL_PCT) */
        const u64 rounds = nchunks / nwaves, left = nchunks % nwaves;
        const u64 nmain = (rounds && left && left * 100 < nwaves * pct) ? rounds * nwaves : nchunks;
        if ((sector_bytes % 16 == 0 || nsectors == 1) && mb % XTS_CHUNK == 0) {        /* every chunk whole: k_xts<.., ALLFULL> */
            if ((e = set_lds((k_xts<NR, DEC, true, false, true>), lds)) != hipSuccess) return (int)e;
            hipLaunchKernelGGL((k_xts<NR, DEC, true, false, true>), dim3(xgrid), dim3(wg), lds, st,
                               *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, (u64)sector_bytes, step_q, step_r,
                               (const unsigned char *)in, (unsigned char *)out, nmain, 0u);
        } else if (sector_bytes % 16 == 0 || nsectors == 1)
            hipLaunchKernelGGL((k_xts<NR, DEC, true>), dim3(xgrid), dim3(wg), lds, st,
                               *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, (u64)sector_bytes, step_q, step_r,
                               (const unsigned char *)in, (unsigned char *)out, nmain, 0u);
        else
            hipLaunchKernelGGL((k_xts<NR, DEC, false>), dim3(xgrid), dim3(wg), lds, st,
                               *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, (u64)sector_bytes, step_q, step_r,
                               (const unsigned char *)in, (unsigned char *)out, nmain, 0u);
    }
    if (r) {
        hipLaunchKernelGGL((k_xts_cts<NR, DEC>), dim3(grid_for(nsectors, UAES_WG)), dim3(UAES_WG), lds, st,
                           *k1, *tb, (const uint4 *)scratch, (u64)nsectors, cps, mb, r, (u64)sector_bytes,
                           (const unsigned char *)in, (unsigned char *)out);
    }
    return (int)hipGetLastError();
}

extern "C" int uaesk_xts(void *stream, const uaesk_tables *tb, int nr,
                         const uaesk_rk *k1, const uaesk_rk *k2_enc, int decrypt,
                         const uint8_t *tweak16, uint64_t first_sector,
                         size_t sector_bytes, size_t nsectors,
                         const void *in, void *out, void *scratch)
{
    if (nsectors == 0) return 0;
    if (sector_bytes < 16) return (int)hipErrorInvalidValue;
    if (decrypt) { DISPATCH_NR(nr, return (launch_xts<NR, true>(S(stream), tb, k1, k2_enc, tweak16, first_sector, sector_bytes, nsectors, in, out, scratch))); }
    else         { DISPATCH_NR(nr, return (launch_xts<NR, false>(S(stream), tb, k1, k2_enc, tweak16, first_sector, sector_bytes, nsectors, in, out, scratch))); }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* completion ticket of a synchronous call                                      */
/* ------------------------------------------------------------------------ */
/* One wave behind the call's kernels on the lane's stream: copies up to 64 bytes of result (a status word, a tag,
 * a MAC) from device memory into the lane's PINNED page and then releases the ticket number to the host, which
 * spins on that word instead of entering hipStreamSynchronize (tools/ubench/threadfloor.hip: an empty kernel +
 * hipStreamSynchronize costs 11 us and tops out at 0.31 M calls/s over all host threads; kernel + ticket kernel
 * 8.9 us and 0.6 M calls/s at 8 threads).  Stream order puts it behind the kernels' end-of-kernel release.  */
static thread_local uaesk_done g_armed = { nullptr, nullptr, 0 };

extern "C" void uaesk_ticket_arm(void *pinned_flag, void *d_count, unsigned seq)
{
    g_armed.flag = (unsigned *)pinned_flag; g_armed.count = (unsigned *)d_count; g_armed.seq = seq;
}

static thread_local int g_unused = 0;

/* A word that is ZERO BETWEEN CALLS, for the next kernel-level call of this thread that runs several workgroups in one
 * launch and lets one of them finish the job (the GCM chunk kernel's finisher): the host layer owns such words (the
 * tail of its scratch buffers) and arms one right before the call; a routine that can use it takes it at its entry. */
static thread_local unsigned *g_done_word = nullptr;
extern "C" void uaesk_done_word_arm(unsigned *w) { g_done_word = w; }
unsigned *uaesk_done_word_take()
{
    /* UAES_GCM_FOLD=0: the one-launch arrangements of the GCM family (chunk workgroups + a fold inside the same
     * launch) are off; every such call takes its multi-launch form (chunks, then k_gcm_combine) */
    static const bool fold_on = [] { const char *e = getenv("UAES_GCM_FOLD"); return !(e && e[0] == '0'); }();
    unsigned *w = g_done_word;
    g_done_word = nullptr;
    return fold_on ? w : nullptr;
}

extern "C" int uaesk_ticket_disarm(void)
{
    const int still = g_armed.flag != nullptr || g_unused;
    g_armed.flag = nullptr;
    g_unused = 0;
    return still;
}

/* a launcher took the ticket (so that its building blocks would not) and then found no kernel to put it on */
void uaesk_ticket_unused() { g_unused = 1; }

uaesk_done uaesk_ticket_take()
{
    const uaesk_done d = g_armed;
    g_armed.flag = nullptr;
    return d;
}

__global__ __launch_bounds__(64) void k_ticket(unsigned *flag, unsigned seq, const unsigned *src, unsigned *dst, unsigned nwords)
{
    if (threadIdx.x < nwords) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int uaesk_ticket(void *stream, void *pinned_flag, unsigned seq, const void *d_src, void *pinned_dst, unsigned nbytes)
{
    if (nbytes > 64u || (nbytes & 3u)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ticket, dim3(1), dim3(64), 0, S(stream), (unsigned *)pinned_flag, seq, (const unsigned *)d_src,
                       (unsigned *)pinned_dst, nbytes / 4u);
    return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* shader clock probe (measurement only)                                        */
/* ------------------------------------------------------------------------ */
/* One wave spins for `ticks` periods of the 100 MHz reference counter (s_memrealtime) and reports how many shader
 * cycles (s_memtime) went by: the clock the chip actually runs at while other streams keep it busy.  bench.py
 * launches it on a second stream beside the measured workload (under the 1.4 kW cap a cipher
 * kernel settles near 2.1 GHz, not at the 2.4 GHz the device properties quote).                              */
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

extern "C" int uaesk_clock_probe(void *stream, void *d_out16, unsigned long long ticks_100mhz)
{
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, S(stream), (unsigned long long *)d_out16, ticks_100mhz);
    return (int)hipGetLastError();
}

extern "C" int uaesk_selftest(void *stream, const uaesk_tables *tb, const uaesk_rk *ek128,
                              const uaesk_rk *dk128, unsigned *d_result)
{
    hipError_t e = set_lds(k_selftest, UAES_LDS_ENC);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_selftest, dim3(2), dim3(UAES_WG), UAES_LDS_ENC, S(stream), *ek128, *dk128, *tb, d_result);
    return (int)hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* the table as data (uaes_plan.h)                                             */
/* ------------------------------------------------------------------------ */
int uaesk_plan_gcm(int dir, size_t len, size_t aad_len, unsigned flags, uaes_plan *p);     /* uaes_gcm.hip */
int uaesk_plan_siv(int dir, size_t len, size_t aad_len, unsigned flags, uaes_plan *p);     /* uaes_gcm.hip */
int uaesk_plan_ocb(int dir, size_t len, size_t aad_len, uaes_plan *p);                     /* uaes_ocb.hip */

extern "C" int uaesk_plan(int mode, int dir, size_t a, size_t b, unsigned flags, uaes_plan *p)
{
    if (!p) return (int)hipErrorInvalidValue;
    if (uaesk_device_info(nullptr, nullptr) != 0) g_cus = 0;    /* no device: the table of a 256-CU MI355X (grid_for, plan_cus) */
    switch (mode) {
    case UAES_PLAN_ECB:
        *p = plan_ecb((u64)((a + 15) / 16));
        return 0;
    case UAES_PLAN_CTR: {
        uaesk_ctr c;
        memset(&c, 0, sizeof c);
        c.v0 = 1;                                   /* CTR_START_VALUE; the planner looks at the low byte only */
        c.le32 = (flags >> 3) & 1u;
        *p = plan_ctr(&c, a, nullptr, nullptr);
        return 0;
    }
    case UAES_PLAN_XTS:
        if (a < 16 || b == 0) return (int)hipErrorInvalidValue;
        *p = plan_xts(a, b, (flags >> 1) & 1u);
        return 0;
    case UAES_PLAN_GCM: return uaesk_plan_gcm(dir, a, b, flags, p);
    case UAES_PLAN_SIV: return uaesk_plan_siv(dir, a, b, flags, p);
    case UAES_PLAN_OCB: return uaesk_plan_ocb(dir, a, b, p);
    default: return (int)hipErrorInvalidValue;
    }
}
