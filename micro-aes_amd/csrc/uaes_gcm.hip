/*
 * uaes_gcm.hip -- GCM for gfx950: GHASH as a strided-Horner polynomial
 * evaluation with LDS-resident multiplication tables, plus the GCM driver.
 *
 * Reference behaviour reproduced (micro_aes.c):
 *   mulGF128 :476-493, xMac :551-570, gHash :1127-1137 (N6: AAD and CT are
 *   zero-padded separately, then one block of two 64-bit BE bit lengths),
 *   GCMsetup :1140-1152 (H = Enc(0), J0 = nonce || 00000001),
 *   AES_GCM_encrypt :1164-1179 (CTR from J0+1, tag = Enc(J0) ^ GHASH),
 *   AES_GCM_decrypt :1192-1212 (authenticate first; plaintext untouched and
 *   0x1A returned on mismatch, N7).
 *
 * The reference multiplies bit-serially (128 shift/xor steps per block).
 * gfx950 has no carry-less multiply instruction, so the design is:
 *
 *   GHASH(X_0..X_{M-1}) = sum_u X_u * H^(M-u).  With a stride S, lane j keeps
 *   acc_j = sum_k X_{kS+j} * (H^S)^(K-1-k)  (acc <- acc*H^S ^ X), and
 *   GHASH(X) = GHASH(acc_0..acc_{S-1}): the accumulators of one level are the
 *   input blocks of the next, so the same kernel recurses with smaller
 *   strides (2^18|..|2^12 -> 2^14 -> 1024 -> 64 -> 4 -> 1).  Multiplication by the
 *   FIXED element H^S is GF(2)-linear in the other operand, hence 16 lookups
 *   of 16 bytes in a 64 KiB byte-indexed table (ds_read_b128) + 15 XORs per
 *   block on the bulk levels, and 32 lookups in an 8 KiB nibble-indexed table
 *   on the three tiny last levels.  Tables and the powers H^(2^k) are built
 *   on the GPU by one setup workgroup (wave-cooperative multiply: lane l
 *   contributes coefficients l and l+64, butterfly XOR over the wave).
 *
 * HBM traffic: the bulk level reads each ciphertext block exactly once
 * (16 B/block); everything else is O(2 MiB).
 */
#include <hip/hip_runtime.h>
#include <atomic>
#include <string.h>
#include <stdlib.h>
#include "uaes_ghash.hip.h"

/* ------------------------------------------------------------------------ */
/* bulk level: stride S = gridDim.x * 256                                     */
/* ------------------------------------------------------------------------ */
#define GH_PF 4     /* blocks prefetched per lane */

__global__ __launch_bounds__(GH_PT) void k_ghash_pass(GSrc src, u64 nv, const uint4 *__restrict__ tab8,
                                                     uint4 *__restrict__ accs)
{
    uint4 *T = (uint4 *)uaes_lds;                  /* LDS address 0 (absolute addressing in tabmul8_xor) */
    for (u32 i = threadIdx.x; i < 4096u; i += GH_PT) T[i] = tab8[i];
    __syncthreads();
    const GhLane gl = gh_lane_setup();

    const u64 stride = (u64)gridDim.x * GH_PT;
    const u64 j = (u64)blockIdx.x * GH_PT + threadIdx.x;
    const u64 steps = (nv + stride - 1) / stride;
    const u64 pad = steps * stride - nv;           /* virtual zero blocks in front */
    uint4 acc = make_uint4(0, 0, 0, 0);

    u64 k = 0;
    /* first (possibly padded) step, then whole groups of GH_PF with the loads
     * issued ahead of the dependent multiply chain                          */
    for (; k < steps && (k == 0 || (steps - k) % GH_PF != 0); ++k) {
        const u64 u = k * stride + j;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (u >= pad) x = load_vblock(src, u - pad);
        acc = tabmul8_xor(acc, x, gl);
    }
    for (; k < steps; k += GH_PF) {
        uint4 x[GH_PF];
#pragma unroll
        for (int p = 0; p < GH_PF; ++p) x[p] = load_vblock(src, (k + p) * stride + j - pad);
#pragma unroll
        for (int p = 0; p < GH_PF; ++p) acc = tabmul8_xor(acc, x[p], gl);
    }
    accs[j] = acc;
}


/* ------------------------------------------------------------------------ */
/* last levels (strides 256, 16, 1) + tag handling; one workgroup             */
/* ------------------------------------------------------------------------ */
/* mode 0: write tag = GHASH ^ EJ0 to tag_io (encrypt)
 * mode 1: compare with the 16 bytes at tag_io, *status = 0 / 0x1A (decrypt)
 * mode 2: write the raw GHASH value to tag_io (tests)
 * Levels inside the workgroup: stride 1024, then the radix-4 tree (gh_tree).    */

/* TC: the six nibble tables in LDS, buf: GT_BUF entries behind them, ej0 = Enc(J0) */
__device__ __forceinline__ void ghash_final_body(const GSrc &src, u64 nv, uint4 *TC, uint4 *buf, uint4 ej0,
                                                 int mode, unsigned char *tag_io, int *status)
{
    /* The sequence is front-padded with zero blocks to steps * 1024; a level's leading rows that
     * hold only padding are skipped (their accumulators stay zero), and so is the multiplication
     * of a still-zero accumulator: a short message (the per-call floor) costs a handful of
     * dependent table multiplications instead of 3 + 3 + 3 + 3 + 4 + steps, and the tables of the
     * levels it never multiplies in need not have been built (launch_setup: maxlog).        */
    const u64 steps = (nv + GH_T - 1) / GH_T;
    const u64 pad = steps * GH_T - nv;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (u64 k = 0; k < steps; ++k) {
        const u64 u = k * GH_T + threadIdx.x;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (u >= pad) x = load_vblock(src, u - pad);
        acc = k ? x4(tabmul4(TC, acc), x) : x;
    }
    const u32 live = nv < GH_T ? (u32)nv : GH_T;               /* non-padding entries at the end of buf[0..1024) */
    acc = gh_tree<true>(buf, TC, acc, live);
    if (threadIdx.x == 0) {
        if (mode != 2) acc = x4(acc, ej0);
        const u32 w[4] = { acc.x, acc.y, acc.z, acc.w };
        if (mode == 1) {
            u32 diff = 0;
            for (u32 i = 0; i < 16; ++i) diff |= (u32)tag_io[i] ^ ((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
        } else {
            for (u32 i = 0; i < 16; ++i) tag_io[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
}

__global__ __launch_bounds__(GH_T) void k_ghash_final(GSrc src, u64 nv, const unsigned char *__restrict__ scratch,
                                                      int mode, unsigned char *tag_io, int *status)
{
    uint4 *TC = (uint4 *)uaes_lds;            /* the six nibble tables; TC[0..512) = H^1024 */
    uint4 *buf = TC + GT_NTAB * 512u;
    const uint4 *g4 = (const uint4 *)(scratch + GS_TAB4);
    for (u32 i = threadIdx.x; i < GT_NTAB * 512u; i += GH_T) TC[i] = g4[i];
    __syncthreads();
    ghash_final_body(src, nv, TC, buf, *(const uint4 *)(scratch + GS_EJ0), mode, tag_io, status);
}

/* ------------------------------------------------------------------------ */
/* setup: H, Enc(J0), powers H^(2^k), multiplication tables                   */
/* ------------------------------------------------------------------------ */
#define SETUP_LDS   (UAES_LDS_ENC + 24576u)

/* logA: log2 of the bulk stride (12..17), 0 = no bulk level.  needB: build the
 * H^4096 table.  h_given: skip AES, use hval as H and 0 as Enc(J0) (tests, POLYVAL); 2: H from GS_SIV_HG.
 * logF != 0: also build the table of H^(2^logF) for the fused encrypt kernel.   */
template <int NR>
__global__ __launch_bounds__(UAES_WG) void k_gcm_setup(uaesk_rk ek, uaesk_tables tb, uint4 j0,
                                                       unsigned char *__restrict__ scratch,
                                                       u32 logA, u32 needB, u32 h_given, uint4 hval, u32 want_pow64,
                                                       u32 logF, u32 maxlog)
{
    /* A long one-shot call (logF) is launched on TWO workgroups that both make H and its powers and then split the
     * rest: workgroup 0 the tables, workgroup 1 the fused kernel's weights (latency-bound dependent products that
     * would otherwise wait behind the table stores: 10.7 -> ~7.5 us).  Every output has exactly one writer.      */
    const bool tables_wg = blockIdx.x == 0, weights_wg = blockIdx.x == gridDim.x - 1;
    Gf *shPow = (Gf *)(uaes_lds + UAES_LDS_ENC);          /* up to 21 powers   */
    Gf *shGen = shPow + 32;                                /* 9 x 128 generators: 18 KiB */
    uint4 *gH = (uint4 *)(scratch + GS_H);

    if (!h_given) {
        /* H = Enc(0) and Enc(J0): two blocks, a quad of lanes each (quad_encrypt: 4 KiB of table
         * stores instead of 128 KiB, ~2x shorter latency than one lane per block)               */
        const u32 kb = UAES_LDS_ENC + 19200u;                  /* behind shPow / shGen */
        quad_fill_tables(tb.te0, ek, kb);
        const LaneConst lc = quad_lane_const();
        if (threadIdx.x < 64) {                                /* wave 0: quad 0 -> H, the other quads -> Enc(J0) */
            u32 s[4] = { 0, 0, 0, 0 };
            if (threadIdx.x >= 4) { s[0] = j0.x; s[1] = j0.y; s[2] = j0.z; s[3] = j0.w; }
            quad_encrypt<NR>(s, ek, lc, kb);
            if (tables_wg && (threadIdx.x == 0 || threadIdx.x == 4)) gH[threadIdx.x >> 2] = make_uint4(s[0], s[1], s[2], s[3]);
            if (threadIdx.x == 0) shPow[0] = gf_from_words(s[0], s[1], s[2], s[3]);
        }
    } else if (threadIdx.x == 0) {
        if (h_given == 2) hval = *(const uint4 *)(scratch + GS_SIV_HG);     /* made by k_siv_prep on this stream */
        if (tables_wg) {
            gH[0] = hval;
            gH[1] = make_uint4(0, 0, 0, 0);
        }
        shPow[0] = gf_from_words(hval.x, hval.y, hval.z, hval.w);
    }
    __syncthreads();

    /* the highest power any requested table needs; the workgroup weights take Y^(2^i), Z^(2^i), i < 4 = H^(2^(11..18)) */
    const u32 last = want_pow64 ? 63u : (logF && maxlog < 18u ? 18u : maxlog);
    if (tb.frob) {
        /* H^(2^k) = F^k H: squaring is the Frobenius map, linear over GF(2) with a matrix that depends on
         * the field only, so every power is ONE bit-matrix product with a constant (two parities per
         * lane, two ballots per wave) instead of k dependent multiplications; the waves take different k */
        uint4 *g64 = (uint4 *)(scratch + GS_POW64);
        const Gf h = shPow[0];
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        if (threadIdx.x == 0 && want_pow64 && tables_wg) {
            u32 w[4];
            gf_to_words(h, w);
            g64[0] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (u32 k = wave + 1; k <= last; k += UAES_WG / 64) {
            const uint64_t *rows = tb.frob + (u64)(k - 1) * 256u;
            const u32 b0 = (u32)(__popcll(rows[2 * lane] & h.hi) + __popcll(rows[2 * lane + 1] & h.lo)) & 1u;
            const u32 b1 = (u32)(__popcll(rows[128 + 2 * lane] & h.hi) + __popcll(rows[128 + 2 * lane + 1] & h.lo)) & 1u;
            Gf p;
            p.hi = __ballot(b0);
            p.lo = __ballot(b1);
            if (lane == 0) {
                if (k <= GF_MAXLOG) shPow[k] = p;
                if (want_pow64 && tables_wg) {
                    u32 w[4];
                    gf_to_words(p, w);
                    g64[k] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    } else if (threadIdx.x < 64) {             /* no matrices: squaring chain, wave 0 */
        Gf p = shPow[0];
        uint4 *g64 = (uint4 *)(scratch + GS_POW64);
        for (u32 k = 0; k <= last; ++k) {
            if (k) p = wave_gfmul(p, p, threadIdx.x);
            if (threadIdx.x == 0) {
                if (k <= GF_MAXLOG) shPow[k] = p;
                if (want_pow64 && tables_wg) {
                    u32 w[4];
                    gf_to_words(p, w);
                    g64[k] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    }
    __syncthreads();

    if (tables_wg) {
    /* generators M*x^q of the tables: A (2^logA), B (2^14), the six nibble tables (2^10, 2^8, 2^6, 2^4, 2^2,
     * 2^0), fused (2^logF)                                                                              */
    const u32 logs[9] = { logA, GH_LOGB, 10u, 8u, 6u, 4u, 2u, 0u, logF };
    for (u32 idx = threadIdx.x; idx < 9u * 128u; idx += UAES_WG) {
        const u32 t = idx >> 7, q = idx & 127u;
        shGen[idx] = gf_mul_xq128(shPow[logs[t]], q);
    }
    __syncthreads();

    /* byte-indexed tables: entry (j, v) = sum_i bit(v, 7-i) * gen[8j+i] */
    for (u32 t = 0; t < 3; ++t) {
        if ((t == 0 && !logA) || (t == 1 && !needB) || (t == 2 && !logF)) continue;
        uint4 *dst = (uint4 *)(scratch + (t == 0 ? GS_TAB8_A : t == 1 ? GS_TAB8_B : GS_TAB8_F));
        const Gf *gen = shGen + 128 * (t == 2 ? 8 : t);
        /* entry (v, j) sits at v * 16 + j: a thread keeps its slot j = tid & 15 (eight generators, read once) and
         * takes rows v = tid / 16 + 64 q, so that a wave stores 1 KiB of consecutive entries per pass -- row by row
         * (v = tid & 255) every lane wrote into a different 256-byte row, 64 cache lines per store: 2.9 us a table */
        const u32 j = threadIdx.x & 15u;
        Gf g[8];
#pragma unroll
        for (u32 i = 0; i < 8; ++i) g[i] = gen[8 * j + i];
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
            const u32 v = (threadIdx.x >> 4) + 64u * q;
            Gf e = { 0, 0 };
#pragma unroll
            for (u32 i = 0; i < 8; ++i) {
                const u64 m = 0 - (u64)((v >> (7 - i)) & 1u);
                e.hi ^= g[i].hi & m;
                e.lo ^= g[i].lo & m;
            }
            u32 w[4];
            gf_to_words(e, w);
            dst[v * 16 + j] = make_uint4(w[0], w[1], w[2], w[3]);      /* row v, slot j */
        }
    }
    /* nibble-indexed tables: entry (p, v) = sum_i bit(v, 3-i) * gen[4p+i] */
    if (threadIdx.x < 512) {
        const u32 p = threadIdx.x >> 4, v = threadIdx.x & 15u;
        for (u32 t = 0; t < GT_NTAB; ++t) {
            uint4 *dst = (uint4 *)(scratch + GS_TAB4 + 8192u * t);
            const Gf *gen = shGen + 128 * (2 + t);
            Gf e = { 0, 0 };
#pragma unroll
            for (u32 i = 0; i < 4; ++i) {
                const u64 m = 0 - (u64)((v >> (3 - i)) & 1u);
                e.hi ^= gen[4 * p + i].hi & m;
                e.lo ^= gen[4 * p + i].lo & m;
            }
            u32 w[4];
            gf_to_words(e, w);
            dst[nib_entry(p, v)] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    }                                          /* tables_wg */
    if (logF && weights_wg) {
        /* weights of the fused kernel's workgroups: Y^k and Z^k = Y^(16k), k < 16 (Y = H^2048).  The squarings
         * Y^(2^i) = H^(2^(11+i)), Z^(2^i) = H^(2^(15+i)) are among the powers made above, so every other exponent
         * is a product of at most three of them: TWO rounds of independent products (six, then five per family,
         * one wave each) instead of four doubling rounds (4.2 -> 2.2 us of every one-shot long call)          */
        Gf *shY = (Gf *)(uaes_lds + UAES_LDS_ENC + 19456u);       /* [0..15] Y^k, [16..31] Z^k */
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const u32 fam = wave >> 3, i = wave & 7u;                  /* product number i of this family */
        if (threadIdx.x < 2) {
            const Gf one = { 0x8000000000000000ull, 0 };
            const u32 base = threadIdx.x ? 15u : 11u;
            shY[16 * threadIdx.x] = one;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) shY[16 * threadIdx.x + (1u << q)] = shPow[base + q];
        }
        __syncthreads();
        {   /* 3 = 2+1, 5 = 4+1, 6 = 4+2, 9 = 8+1, 10 = 8+2, 12 = 8+4 */
            const u32 a = i < 1 ? 2u : i < 3 ? 4u : 8u, b = i == 0 || i == 1 || i == 3 ? 1u : (i == 2 || i == 4 ? 2u : 4u);
            if (i < 6) {
                const Gf pr = wave_gfmul(shY[16 * fam + a], shY[16 * fam + b], lane);
                if (lane == 0) shY[16 * fam + a + b] = pr;
            }
        }
        __syncthreads();
        {   /* 7 = 4+3, 11 = 8+3, 13 = 8+5, 14 = 8+6, 15 = 12+3 */
            const u32 a = i == 0 ? 4u : i < 4 ? 8u : 12u, b = i < 2 || i == 4 ? 3u : (i == 2 ? 5u : 6u);
            if (i < 5) {
                const Gf pr = wave_gfmul(shY[16 * fam + a], shY[16 * fam + b], lane);
                if (lane == 0) shY[16 * fam + a + b] = pr;
            }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            u32 w[4];
            gf_to_words(shY[threadIdx.x], w);
            ((uint4 *)(scratch + GS_YLO))[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (threadIdx.x == 32) *(uint4 *)(scratch + GS_T) = make_uint4(0, 0, 0, 0);
    }
    if (tables_wg && threadIdx.x <= (want_pow64 ? GH_MAXLOG : (maxlog < GH_MAXLOG ? maxlog : GH_MAXLOG))) {
        u32 w[4];
        gf_to_words(shPow[threadIdx.x], w);
        ((uint4 *)(scratch + GS_POW))[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

/* ------------------------------------------------------------------------ */
/* fused encrypt: CTR with shared rounds + GHASH of the ciphertext in one pass  */
/* ------------------------------------------------------------------------ */
/* The separate GHASH pass re-reads the gigabyte the CTR kernel has just written (1.5x the
 * algorithmic HBM traffic).  Here every lane folds its ciphertext blocks into its own
 * GHASH accumulators while they are still in registers:
 *
 *   the striped geometry of ctr_shared_loop puts block (b, it, q, u, p) at  j + S it,
 *   j = 2048 b + 256 (q + 4u) + p,  S = 2048 * workgroups  -- a strided Horner layout -- so
 *   acc_j <- acc_j * H^S ^ C  per block (tabmul8_xor_half: the 64 KiB byte table of H^S in LDS
 *   next to the 64 KiB split-halves AES tables).
 *
 * When its stripes are done a workgroup reduces its 2048 accumulators to ONE block
 *   R_b = sum_q acc_q * H^(2047 - q)        (nibble tables H^1024, H^64, H^4, H loaded over the
 *                                            AES tables: 1 + 16 + 16 + 4 dependent multiplies)
 * which stands for all the blocks the workgroup encrypted, as a polynomial that ends with its
 * last stripe.  k_b = stripes dealt after that stripe (0..workgroups-1; workgroups may differ
 * by one stripe), so the workgroup's share of the hash, taken to the end of the striped
 * region, is R_b * Y^(k_b), Y = H^2048 = Ylo[k_b & 15] * Zhi[k_b >> 4] (two wave-cooperative
 * products); the shares are XORed into one 16-byte block T in device memory.
 *
 * Message layout: [AAD blocks][head: ciphertext blocks up to the first group boundary of the
 * counter, < 256][striped region][tail: < 2048 blocks + ragged bytes][length block].  The
 * nfront = AAD + head blocks are the INITIAL values of the last nfront accumulators (they sit
 * exactly one stride before those lanes' first blocks); a head block is encrypted by the lane
 * that absorbs it.  The tail is encrypted in the prologue; the tag is
 * GHASH([T][tail][lengths]) ^ Enc(J0), one short k_ghash_final.
 *
 * LDS: [0, 64K) GHASH table, [64K, 128K) AES tables (later: nibble tables + reduction
 * buffer), then the U-buffer.                                                            */
#define GF_LDS_AES    65536u
#define GF_LDS_BUF    131072u
#define GF_LDS_TOTAL  (GF_LDS_BUF + 2u * CTRS_CHUNK * 32u)

/* tabmul8_xor in two halves of 8 lookups: 32 registers of table entries in flight instead
 * of 64 (the fused kernel shares its 128-register budget with the cipher's two states)   */
__device__ __forceinline__ uint4 tabmul8_xor_half(uint4 a, uint4 x, const GhLane &gl)
{
    u32 r[4];
    gh_words(a, gl.g, r);
    u32 z[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32x4 e[8];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 8 * h + i;
            e[i] = *(lds_cu128 *)(uintptr_t)__builtin_amdgcn_perm(r[t >> 2], gl.so[t >> 2], gl.sel[t & 3]);
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) z[w] = xor3(z[w], e[i][w], e[i + 1][w]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return make_uint4(z[0], z[1], z[2], z[3]);
}

template <bool DEC>
struct GhFold {
    static constexpr bool of_input = DEC;
    static constexpr int round_prio = 2;       /* cipher lookups 2 > GHASH lookups 1 > XOR work 0 */
    static constexpr bool expand2 = false;     /* the loop body once per trip: twice would spill (128 VGPRs are all in use) */
    static constexpr bool text_ahead = false;  /* one text buffer, loaded at the head of its own iteration: no second buffer,
                                                  no eight v_mov per trip to rotate it (profiles/r05_gcm_text_ab.log) */
    uint4 acc[2];
    GhLane gl;
    __device__ __forceinline__ void operator()(const uint4 &c0, const uint4 &c1)
    {
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = tabmul8_xor_half(acc[0], c0, gl);
        acc[1] = tabmul8_xor_half(acc[1], c1, gl);
    }
};



template <int NR, bool DEC>
__global__ __launch_bounds__(UAES_WG) void k_gcm_fused(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr,
                                                           const uint4 *in, uint4 *out,
                                                           u64 g_lo, u64 stripes, u64 h1, u64 nfull, u32 rem,
                                                           GSrc front, u64 nfront,
                                                           unsigned char *__restrict__ scratch)
{
    uint4 *T = (uint4 *)uaes_lds;                  /* LDS address 0 (absolute addressing in tabmul8_xor) */
    const uint4 *tab8 = (const uint4 *)(scratch + GS_TAB8_F);
    for (u32 i = threadIdx.x; i < 4096u; i += UAES_WG) T[i] = tab8[i];
    fill_tables64(tb.te0, GF_LDS_AES);             /* ends with a barrier */
    const LaneConst2 lc = make_lane_const2(GF_LDS_AES);
    GhFold<DEC> fold;
    fold.gl = gh_lane_setup();

    /* tail blocks [h1, nfull) and the ragged bytes */
    ctr_edge_blocks<NR>(rk, ctr, in, out, 0, h1, nfull, rem, lc);

    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 p = ((wave & 3u) << 6) | lane, quad = wave >> 2;
    const u64 S = 2048ull * gridDim.x;
    const u64 ablk = (front.aad_len + 15) >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const u64 j = 2048ull * blockIdx.x + 256u * (quad + 4u * u) + p;
        fold.acc[u] = make_uint4(0, 0, 0, 0);
        if (j >= S - nfront) {                     /* a block in front of the striped region: AAD, or a head block */
            const u64 f = j - (S - nfront);
            if (f < ablk) {
                fold.acc[u] = load_vblock_fwd(front, f);
            } else {                               /* GHASH takes the ciphertext: the block read when decrypting */
                const uint4 d = in[f - ablk];
                const uint4 c = ctr_one_block<NR>(rk, ctr, in, out, f - ablk, lc);
                fold.acc[u] = DEC ? d : c;
            }
        }
    }
    CtrGeo geo;
    geo.first = g_lo;
    geo.iters = stripes / gridDim.x + (blockIdx.x < stripes % gridDim.x ? 1 : 0);
    ctr_shared_loop<NR>(rk, ctr, in, out, geo, GF_LDS_BUF, lc, fold);

    /* ---- the workgroup's 2048 accumulators -> R_b -> weighted share into T ---- */
    /* the six nibble tables go over the AES tables: requested BEFORE the barrier (the loads touch no LDS), stored
     * behind it -- the cache round trip hides behind the wait for the workgroup's slowest wave               */
    uint4 *TC = (uint4 *)(uaes_lds + GF_LDS_AES);
    uint4 *buf = (uint4 *)uaes_lds;                /* GT_BUF entries, over the GHASH table (every wave is past its last fold) */
    {
        static_assert(GT_NTAB * 512u == 3u * UAES_WG, "three table entries per thread");
        const uint4 *g4 = (const uint4 *)(scratch + GS_TAB4);
        const uint4 e0 = g4[threadIdx.x], e1 = g4[threadIdx.x + UAES_WG], e2 = g4[threadIdx.x + 2u * UAES_WG];
        __syncthreads();                           /* every wave is done with the AES tables */
        TC[threadIdx.x] = e0; TC[threadIdx.x + UAES_WG] = e1; TC[threadIdx.x + 2u * UAES_WG] = e2;
    }
    __syncthreads();
    /* thread q' = 256 quad + p holds accumulators q' and q' + 1024; then the radix-4 tree, last term H^0 */
    uint4 acc = gh_tree<false>(buf, TC, x4(tabmul4(TC, fold.acc[0]), fold.acc[1]), 1024u);
    if (threadIdx.x == 0) buf[GT_BUF - 1] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {                        /* wave 0: R_b * Ylo[k & 15] * Zhi[k >> 4] */
        const u64 last = (u64)blockIdx.x + (u64)gridDim.x * (geo.iters - 1);        /* this workgroup's last stripe */
        const u32 k = (u32)(stripes - 1 - last);
        const uint4 *yl = (const uint4 *)(scratch + GS_YLO), *zh = (const uint4 *)(scratch + GS_ZHI);
        Gf w = wave_gfmul(gf_from4(yl[k & 15u]), gf_from4(zh[k >> 4]), threadIdx.x);
        w = wave_gfmul(gf_from4(buf[GT_BUF - 1]), w, threadIdx.x);
        if (threadIdx.x == 0) {
            u32 ww[4];
            gf_to_words(w, ww);
            unsigned *t = (unsigned *)(scratch + GS_T);
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicXor(t + i, ww[i]);
        }
    }
}

template <int NR, bool DEC>
static int launch_fused(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c,
                        const void *in, void *out, unsigned grid, u64 g_lo, u64 stripes, u64 h1, u64 nfull, u32 rem,
                        const GSrc &front, u64 nfront, unsigned char *sc)
{
    hipError_t e = uaesk_want_lds((const void *)k_gcm_fused<NR, DEC>, (unsigned)(GF_LDS_TOTAL));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_gcm_fused<NR, DEC>), dim3(grid), dim3(UAES_WG), GF_LDS_TOTAL, st, *ek, *tb, *c,
                       (const uint4 *)in, (uint4 *)out, g_lo, stripes, h1, nfull, rem, front, nfront, sc);
    return (int)hipGetLastError();
}

template <bool DEC>
static int launch_fused_nr(int nr, hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c,
                           const void *in, void *out, unsigned grid, u64 g_lo, u64 stripes, u64 h1, u64 nfull, u32 rem,
                           const GSrc &front, u64 nfront, unsigned char *sc)
{
    switch (nr) {
    case 10: return launch_fused<10, DEC>(st, tb, ek, c, in, out, grid, g_lo, stripes, h1, nfull, rem, front, nfront, sc);
    case 12: return launch_fused<12, DEC>(st, tb, ek, c, in, out, grid, g_lo, stripes, h1, nfull, rem, front, nfront, sc);
    case 14: return launch_fused<14, DEC>(st, tb, ek, c, in, out, grid, g_lo, stripes, h1, nfull, rem, front, nfront, sc);
    default: return (int)hipErrorInvalidValue;
    }
}

/* one-pass decrypt, failed authentication: the plaintext written before the tag was known is zeroed */
__global__ __launch_bounds__(1024) void k_wipe_if_failed(const int *__restrict__ status, unsigned char *out, u64 len)
{
    if (*status == 0) return;
    const u64 n16 = len >> 4, stride = (u64)gridDim.x * blockDim.x;
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (((uintptr_t)out & 15u) == 0) {
        for (u64 i = tid; i < n16; i += stride) ((uint4 *)out)[i] = make_uint4(0, 0, 0, 0);
        for (u64 i = (n16 << 4) + tid; i < len; i += stride) out[i] = 0;
    } else {
        for (u64 i = tid; i < len; i += stride) out[i] = 0;
    }
}

/* ------------------------------------------------------------------------ */
/* key context: what is left of the setup when the tables of a key are kept    */
/* ------------------------------------------------------------------------ */
/* Enc(J0) -> GS_EJ0 and T <- 0: the only per-message inputs of the tag besides the text.  One wave,
 * a quad per block (quad_encrypt), 4 KiB of table stores.                                       */
template <int NR>
__global__ __launch_bounds__(64) void k_gcm_ej0(uaesk_rk ek, uaesk_tables tb, uint4 j0, unsigned char *__restrict__ scratch)
{
    quad_fill_tables(tb.te0, ek);
    const LaneConst lc = quad_lane_const();
    u32 s[4] = { j0.x, j0.y, j0.z, j0.w };
    quad_encrypt<NR>(s, ek, lc);
    if (threadIdx.x == 0) {
        *(uint4 *)(scratch + GS_EJ0) = make_uint4(s[0], s[1], s[2], s[3]);
        *(uint4 *)(scratch + GS_T) = make_uint4(0, 0, 0, 0);
    }
}

template <int NR>
static int launch_ej0(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, uint4 j0, unsigned char *sc)
{
    hipError_t e = uaesk_want_lds((const void *)k_gcm_ej0<NR>, (unsigned)(UAES_LDS_QUAD));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_gcm_ej0<NR>), dim3(1), dim3(64), UAES_LDS_QUAD, st, *ek, *tb, j0, sc);
    return (int)hipGetLastError();
}


/* ------------------------------------------------------------------------ */
/* short messages: the whole of GCM in one workgroup                           */
/* ------------------------------------------------------------------------ */
/* AAD blocks + text blocks + the length block <= 2047 (a 16 KiB TLS record, a 4 KiB page, a packet):
 * CTR, GHASH and the tag in ONE launch of one 1024-thread workgroup instead of three launches
 * (Enc(J0) / CTR / last GHASH levels) -- per call that is what counts (profiles/HISTORY.md).
 * The GHASH input sequence is front-padded to 1024 or 2048 positions; thread t owns positions t and
 * t + 1024.  A thread whose position is a text block computes that block's keystream itself, so the
 * ciphertext it hashes is the one it has in registers; thread 0's first position is always padding and
 * computes Enc(J0) in that slot (every thread runs the same two block encryptions: no divergence
 * around the rounds).  Decrypt hashes the ciphertext it READS, keeps the plaintext in registers, and
 * writes it only after the tag has matched (N7, micro_aes.c:1200-1208) -- in the same launch.
 * The six nibble tables (H^1024 .. H) come from `scratch` (a key context, k_gcm_setup) or are made here.
 * LDS: [0, 64K) the split-halves AES tables, then the six nibble tables and the reduction buffer.    */

template <int NR, bool DEC>
__global__ __launch_bounds__(GH_T) void k_gcm_small(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr, uint4 j0,
                                                    GSrc src, const uint4 *in, uint4 *out,
                                                    const unsigned char *__restrict__ scratch,
                                                    unsigned char *tag_io, int *status, u32 build, uaesk_done done)
{
    /* build != 0: a one-shot call -- the key's nibble tables are not in `scratch`, they are made here from
     * H = Enc(0), which thread 1's padding slot computes (gcm_build_nibble_tables)                        */
    uint4 *TC = (uint4 *)(uaes_lds + GSM_LDS_TAB);            /* the six nibble tables; TC[0..512) = H^1024 */
    uint4 *buf = TC + GT_NTAB * 512u;                         /* GT_BUF entries (the last three: H, Enc(J0), verdict) */
    const u64 len = src.ct_len;
    const u64 ablk = (src.aad_len + 15) >> 4, cblk = (len + 15) >> 4, nv = ablk + cblk + 1;
    const u32 steps = nv + 2 > GH_T ? 2u : 1u;                /* pad >= 2 either way (nv <= 2046) */
    const u64 pad = (u64)steps * GH_T - nv;
    /* what this thread's positions READ (text, AAD, lengths) is requested before the tables are made: the loads -- across
     * the link, for a host caller's pinned window -- then run beside 1.5-2 us of table stores instead of after them */
    uint4 xk[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
    bool is_text[2] = { false, false };
    u64 ti[2] = { 0, 0 };
    u32 tn[2] = { 0, 0 };
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
    for (u32 k = 0; k < 2; ++k) {
        if (k >= steps) break;
        const u64 u = (u64)k * GH_T + threadIdx.x;
        const bool live = u >= pad;
        const u64 v = live ? u - pad : 0;
        is_text[k] = live && v >= ablk && v < ablk + cblk;
        if (is_text[k]) {
            ti[k] = v - ablk;
            const u64 avail = len - 16 * ti[k];
            tn[k] = avail < 16 ? (u32)avail : 16u;
            xk[k] = tn[k] == 16 ? in[ti[k]] : load_bytes_padded((const unsigned char *)(in + ti[k]), tn[k]);
        } else if (live) {
            GSrc rest = src;                                  /* AAD blocks and the length block */
            rest.ct_len = 0;
            xk[k] = load_vblock_fwd(rest, v < ablk ? v : ablk);
        }
    }
    if (!build) {
        const uint4 *g4 = (const uint4 *)(scratch + GS_TAB4);
        for (u32 i = threadIdx.x; i < GT_NTAB * 512u; i += GH_T) TC[i] = g4[i];
    }
    fill_tables64(tb.te0, 0);                                 /* ends with a barrier */
    const LaneConst2 lc = make_lane_const2(0);

    uint4 hold[2];                                            /* decrypt: plaintext waiting for the verdict */
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
    for (u32 k = 0; k < 2; ++k) {
        if (k >= steps) break;
        const bool is_j0 = k == 0 && threadIdx.x == 0;        /* positions 0 and 1 are padding */
        const bool is_h = k == 0 && threadIdx.x == 1;
        u32 s1[1][4];
        ctr_words(ctr, ti[k], s1[0]);
        if (is_j0) { s1[0][0] = j0.x; s1[0][1] = j0.y; s1[0][2] = j0.z; s1[0][3] = j0.w; }
        if (is_h) { s1[0][0] = 0; s1[0][1] = 0; s1[0][2] = 0; s1[0][3] = 0; }
        /* a wave whose sixty-four positions are all padding, AAD or the length block has nothing to encrypt: a 4 KiB
         * text occupies five of the sixteen waves, and the LDS is what the block phase waits for (wave-uniform) */
        if (__builtin_amdgcn_ballot_w64(is_text[k] || is_j0 || is_h) != 0) enc_blocks<NR, 1>(s1, rk, lc);
        if (is_j0) buf[GT_BUF - 2] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
        if (is_h) buf[GT_BUF - 3] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
        if (is_text[k]) {
            const uint4 d = xk[k];
            const u32 nb = tn[k];
            u32 o[4] = { d.x ^ s1[0][0], d.y ^ s1[0][1], d.z ^ s1[0][2], d.w ^ s1[0][3] };
            if (nb < 16) {                                    /* the keystream beyond the text is not part of it (N3) */
#pragma unroll
                for (u32 w = 0; w < 4; ++w) {
                    const u32 keep = nb >= 4 * w + 4 ? 0xffffffffu : nb <= 4 * w ? 0u : (1u << (8 * (nb - 4 * w))) - 1u;
                    o[w] &= keep;
                }
            }
            const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
            if (DEC) {
                hold[k] = ov;                                 /* GHASH takes the ciphertext read, zero padded (N6): xk[k] stays */
            } else {
                xk[k] = ov;
                if (nb == 16) {
                    out[ti[k]] = ov;
                } else {
                    unsigned char *dst = (unsigned char *)(out + ti[k]);
                    for (u32 b = 0; b < nb; ++b) dst[b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
    }
    if (build) {
        __syncthreads();                                      /* H is in its slot */
        gcm_build_nibble_tables(TC, buf, tb.frob);
    }
    uint4 acc = xk[0];
    if (steps == 2) acc = x4(tabmul4(TC, acc), xk[1]);
    const u32 live_n = steps == 1 ? (u32)nv : GH_T;
    acc = gh_tree<true>(buf, TC, acc, live_n);
    if (threadIdx.x == 0) {
        acc = x4(acc, buf[GT_BUF - 2]);
        const u32 w[4] = { acc.x, acc.y, acc.z, acc.w };
        if (DEC) {
            u32 diff = 0;
            for (u32 b = 0; b < 16; ++b) diff |= (u32)tag_io[b] ^ ((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
            buf[GT_BUF - 1] = make_uint4(diff, 0, 0, 0);
        } else {
            for (u32 b = 0; b < 16; ++b) tag_io[b] = (unsigned char)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
    if (DEC) {
        __syncthreads();
        if (buf[GT_BUF - 1].x == 0) {
#pragma unroll
            for (u32 k = 0; k < 2; ++k) {
                if (!is_text[k]) continue;
                if (tn[k] == 16) {
                    out[ti[k]] = hold[k];
                } else {
                    const u32 o[4] = { hold[k].x, hold[k].y, hold[k].z, hold[k].w };
                    unsigned char *dst = (unsigned char *)(out + ti[k]);
                    for (u32 b = 0; b < tn[k]; ++b) dst[b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
    }
    ticket_release(done);
}

template <int NR, bool DEC>
static int launch_small(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c, uint4 j0,
                        const GSrc &src, const void *in, void *out, const unsigned char *sc,
                        unsigned char *tag_io, int *status, u32 build, const uaesk_done &done)
{
    hipError_t e = uaesk_want_lds((const void *)k_gcm_small<NR, DEC>, (unsigned)(GSM_LDS_TOTAL));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_gcm_small<NR, DEC>), dim3(1), dim3(GH_T), GSM_LDS_TOTAL, st, *ek, *tb, *c, j0, src,
                       (const uint4 *)in, (uint4 *)out, sc, tag_io, status, build, done);
    return (int)hipGetLastError();
}

template <bool DEC>
static int launch_small_nr(int nr, hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c,
                           uint4 j0, const GSrc &src, const void *in, void *out, const unsigned char *sc,
                           unsigned char *tag_io, int *status, u32 build, const uaesk_done &done)
{
    switch (nr) {
    case 10: return launch_small<10, DEC>(st, tb, ek, c, j0, src, in, out, sc, tag_io, status, build, done);
    case 12: return launch_small<12, DEC>(st, tb, ek, c, j0, src, in, out, sc, tag_io, status, build, done);
    case 14: return launch_small<14, DEC>(st, tb, ek, c, j0, src, in, out, sc, tag_io, status, build, done);
    default: return (int)hipErrorInvalidValue;
    }
}


/* ------------------------------------------------------------------------ */
/* medium messages (up to 256 chunks of 2048 GHASH blocks: 8 MiB): chunk kernel + combine kernel */
/* ------------------------------------------------------------------------ */
/* Between the single-workgroup kernel (<= 2046 blocks) and the striped one-pass kernel (>= 8 MiB) a text used to
 * take k_gcm_setup + k_ctr + [k_ghash_pass] + k_ghash_final: three or four launches of which all but k_ctr run on
 * ONE workgroup, 35-45 us over a CTR call whatever the size.  Here the GHASH input sequence (AAD, text, lengths),
 * front-padded to W * 2048 positions, is cut into W chunks; workgroup w of k_gcm_chunks is k_gcm_small's body on
 * chunk w -- it encrypts the text blocks among its 2048 positions, hashes them and writes the chunk's hash
 * G_w = sum x_q H^(2048 - q) -- and k_gcm_combine folds the W partial hashes with the SAME radix-4 tree over the
 * powers of Y = H^2048 (Y^64, Y^16, Y^4, Y are Frobenius powers H^(2^17), H^(2^15), H^(2^13), H^(2^11)):
 * GHASH = sum_w G_w Y^(W-1-w), tag = GHASH ^ Enc(J0).  Every workgroup makes the tables it needs from H = Enc(0)
 * itself (or takes them from a key context).  Two launches, 22 us of device time from 64 KiB to 8 MiB.
 * MODE 0: encrypt; 1: hash the ciphertext only (decrypt, tag first: N7); 2: decrypt while hashing (one pass).  */
#define GMC_MAXW 1024u          /* the combine kernel folds up to 1024 partial hashes: 32 MiB */

/* FOLD: ONE launch of W chunk workgroups + one PREPARING workgroup (the last of the grid).  The preparing workgroup
 * makes what the fold needs beside the chunk work -- Enc(J0) and the tables of Y, 6-7 us of a one-shot call -- and the
 * fold itself is done by WHOEVER OF THE W + 1 ARRIVES LAST on *done_word (a fetch-add each; the one that reads W is
 * last), the way k_ocb makes its tag.  No workgroup ever waits for another one to make progress: correctness does not
 * depend on the order or the concurrency in which the device runs the workgroups (CU masking, a debugger, a
 * partitioned or oversubscribed device), and nothing traps.  Before it counts itself in, the preparing workgroup LOOKS
 * at the counter for a bounded time (GMC_LOOK_TICKS of the 100 MHz clock = 1 ms; the chunk work of the largest
 * one-launch call is ~25 us): on a healthy device every chunk workgroup has arrived by then, the preparing workgroup
 * is the last and folds with the tables it already holds in LDS -- the fast path, the timing of round 5 (64 KiB
 * 18.9 us per call, 1 MiB 21.1, with a key context 15.3).  If the look runs out it counts in and leaves; the chunk
 * workgroup that then arrives last makes the tables itself (gcm_combine_body<NR, false>: 6-7 us more, only then).
 * A chunk's hash goes out as four device-scope atomic exchanges whose results are back before the workgroup counts
 * itself in (uaes_ocb.hip explains why not a release fence); the one that folds resets the counter for the next call
 * (a word that is zero between calls, uaes_device.h; the host layer re-zeroes it after a launch that failed).        */
#define GMC_LOOK_TICKS 100000ull

/* the look as the kernels get it (GmcFin.look_ticks): GMC_LOOK_TICKS, or UAES_GCM_LOOK_TICKS / uaesk_debug_gcm_look()
 * -- 0 makes the preparing workgroup count in without looking, so that a chunk workgroup is usually the last
 * (tests/test_gpu_parity.py runs every one-launch size both ways) */
static std::atomic<unsigned long long> g_look_ticks{ ~0ull };
static u64 gcm_look_ticks(void)
{
    unsigned long long v = g_look_ticks.load(std::memory_order_relaxed);
    if (v == ~0ull) {
        const char *e = getenv("UAES_GCM_LOOK_TICKS");
        v = e && *e ? strtoull(e, nullptr, 10) : GMC_LOOK_TICKS;
        g_look_ticks.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" void uaesk_debug_gcm_look(unsigned long long ticks) { g_look_ticks.store(ticks, std::memory_order_relaxed); }

/* how many folds a CHUNK workgroup has done on this device since the module was loaded (diagnostic) */
__device__ unsigned g_gcm_chunk_folds;
extern "C" int uaesk_debug_gcm_chunk_folds(unsigned *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gcm_chunk_folds), sizeof *out, 0, hipMemcpyDeviceToHost);
}
struct GmcFin {
    uint4 j0;
    unsigned *done_word;
    u64 look_ticks;                                           /* the preparing workgroup's bounded look (100 MHz ticks) */
    unsigned char *tag_io;
    int *status;
    int mode;                                                 /* 0: write the tag, 1: compare it, 2: a piece of a streamed
                                                               * message -- fold its raw hash into the running value */
    u32 ylog;
    u32 fin_build;                                            /* mode 2: the finisher makes the tables of Y (the chunk
                                                               * workgroups take theirs from the scratch: `build`) */
    u64 m;                                                    /* mode 2: blocks of the piece */
    u64 len_aad, len_ct;                                      /* modes 0 / 1: the length block, which the FINISHER absorbs --
                                                               * (S ^ lengths) H, S the hash of AAD and text the chunks made: a
                                                               * text of 2^k blocks is 2^k positions, not one more          */
    uaesk_done done;
};

template <int NR, bool WAIT>
__device__ __forceinline__ void gcm_combine_body(const uaesk_rk &ek, const uaesk_tables &tb, uint4 j0, const uint4 *partial, u32 W,
                                                 const unsigned char *__restrict__ scratch, u32 build,
                                                 int mode, unsigned char *tag_io, int *status, u32 ylog,
                                                 unsigned char *wipe_out, u64 wipe_len, unsigned *done_word,
                                                 u64 len_aad, u64 len_ct, u64 look_ticks = 0);

template <int NR, int MODE, bool FOLD = false>
__global__ __launch_bounds__(GH_T) void k_gcm_chunks(uaesk_rk rk, uaesk_tables tb, uaesk_ctr ctr,
                                                     GSrc src, const uint4 *in, uint4 *out,
                                                     const unsigned char *__restrict__ scratch, uint4 *partial, u32 build,
                                                     u32 steps, GmcFin fin)
{
    if (FOLD && blockIdx.x == gridDim.x - 1u) {
        gcm_combine_body<NR, true>(rk, tb, fin.j0, partial, gridDim.x - 1u, scratch, fin.mode == 2 ? fin.fin_build : build, fin.mode,
                                   fin.tag_io, fin.status, fin.ylog, nullptr, fin.m, fin.done_word, fin.len_aad, fin.len_ct,
                                   fin.look_ticks);
        ticket_release(fin.done);
        return;
    }
    const u32 wg = blockIdx.x, nwg = FOLD ? gridDim.x - 1u : gridDim.x;
    uint4 *TC = (uint4 *)(uaes_lds + GSM_LDS_TAB);
    uint4 *buf = TC + GT_NTAB * 512u;
    const u64 len = src.ct_len;
    const u64 ablk = (src.aad_len + 15) >> 4, cblk = (len + 15) >> 4, nv = ablk + cblk + (src.has_len ? 1u : 0u);
    const u64 chunk = (u64)steps * GH_T;                      /* positions per workgroup: one or two per thread */
    const u64 pad = chunk * nwg - nv;                   /* zero positions in front of the sequence */
    GSrc rest = src;                                          /* AAD blocks and the length block */
    rest.ct_len = 0;
    /* hash only: the step's block is all a step needs from memory, so the blocks of up to four steps are requested
     * ahead of the dependent products (as k_ghash_pass does) -- and the first group's HERE, before the tables are made:
     * they travel beside the 5 us of table building instead of behind it */
    auto load_x = [&](u32 k) -> uint4 {
        const u64 P = chunk * wg + (u64)k * GH_T + threadIdx.x;
        if (P < pad) return make_uint4(0, 0, 0, 0);
        const u64 v = P - pad;
        if (v >= ablk && v < ablk + cblk) {
            const u64 i = v - ablk, avail = len - 16 * i;
            const uint4 d = avail >= 16 ? in[i] : load_bytes_padded((const unsigned char *)(in + i), (u32)avail);
            return src.rev ? rev16(d) : d;
        }
        return load_vblock(rest, v < ablk ? v : ablk);        /* (byte-reversed for POLYVAL) */
    };
    uint4 x0[4] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
    if (MODE == 1) {
#pragma unroll
        for (u32 p = 0; p < 4; ++p)
            if (p < steps) x0[p] = load_x(p);
    }
    if (!build) {
        const uint4 *g4 = (const uint4 *)(scratch + GS_TAB4);
        for (u32 i = threadIdx.x; i < GT_NTAB * 512u; i += GH_T) TC[i] = g4[i];
    }
    const LaneConst2 lc = make_lane_const2(0);
    GhLane gl = gh_lane_setup();
    if (MODE == 1) {
        /* hash only (a decryption that authenticates first, GMAC, GCM-SIV's POLYVAL): no cipher tables -- their 64 KiB
         * hold the BYTE table of H^1024 for the stride products (sixteen ds_read_b128 per product where the nibble
         * table takes thirty-two: the loop's only LDS traffic, so a step costs half), made from nibble table 0 once
         * the tables are there: T8[v][j] = T4[2j][v >> 4] ^ T4[2j + 1][v & 15]                                    */
        if (build) {
            if (src.rev) {                                    /* POLYVAL of a long GCM-SIV message: the key k_siv_prep made */
                if (threadIdx.x == 0) buf[GT_BUF - 3] = *(const uint4 *)(scratch + GS_SIV_HG);
            } else {                                          /* H = Enc(0) through an unreplicated Te0 (the table's place) */
                u32 *te_plain = (u32 *)uaes_lds;
                if (threadIdx.x < 256u) te_plain[threadIdx.x] = tb.te0[threadIdx.x];
                __syncthreads();
                if (threadIdx.x < 64u) {
                    u32 s1[4] = { 0, 0, 0, 0 };
                    plain_encrypt<NR>(te_plain, rk, s1);
                    if (threadIdx.x == 0) buf[GT_BUF - 3] = make_uint4(s1[0], s1[1], s1[2], s1[3]);
                }
            }
            __syncthreads();
            gcm_build_nibble_tables(TC, buf, tb.frob);        /* ends with a barrier */
        } else {
            __syncthreads();                                  /* the key context's tables are in */
        }
        if (steps > 1) {
            uint4 *T8 = (uint4 *)uaes_lds;
            for (u32 e = threadIdx.x; e < 4096u; e += GH_T) {
                const u32 v = e >> 4, j = e & 15u;
                T8[e] = x4(TC[nib_entry(2u * j, v >> 4)], TC[nib_entry(2u * j + 1u, v & 15u)]);
            }
            __syncthreads();
        }
    } else {
        fill_tables64(tb.te0, 0);                             /* ends with a barrier */
        if (build) {                                          /* H = Enc(0): one uniform pass, thread 0 keeps it */
            u32 s1[1][4] = { { 0, 0, 0, 0 } };
            enc_blocks<NR, 1>(s1, rk, lc);
            if (threadIdx.x == 0) buf[GT_BUF - 3] = make_uint4(s1[0][0], s1[0][1], s1[0][2], s1[0][3]);
            __syncthreads();
            gcm_build_nibble_tables(TC, buf, tb.frob);
        }
    }
    /* position k * 1024 + t of the chunk belongs to thread t: Horner over the thread's own positions with H^1024
     * (table 0), steps = 1, 2, 4 ... so that the chunk length stays a power of two (the finisher's Y = H^chunk is a
     * Frobenius power) and one round of workgroups covers the text */
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (MODE == 1) {
        acc = x0[0];
        if (steps >= 2) acc = tabmul8_xor(acc, x0[1], gl);    /* acc * H^1024 ^ x through the byte table at LDS 0 */
        if (steps >= 4) {
            acc = tabmul8_xor(acc, x0[2], gl);
            acc = tabmul8_xor(acc, x0[3], gl);
        }
        for (u32 k = 4; k < steps; k += 4) {
            uint4 x[4];
#pragma unroll
            for (u32 p = 0; p < 4; ++p) x[p] = load_x(k + p);
#pragma unroll
            for (u32 p = 0; p < 4; ++p) acc = tabmul8_xor(acc, x[p], gl);
        }
    }
    for (u32 k = 0; MODE != 1 && k < steps; ++k) {
        if (k) {                                 /* (first, while little else is live; an opaque copy of the
                                                               * thread number, or tabmul4's sixteen slot constants are
                                                               * hoisted out of the loop and spilled) */
            u32 tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            acc = tabmul4<4>(TC, acc, tid);
        }
        const u64 P = chunk * wg + (u64)k * GH_T + threadIdx.x;
        const bool live = P >= pad;
        const u64 v = live ? P - pad : 0;
        const bool is_text = live && v >= ablk && v < ablk + cblk;
        const u64 i = is_text ? v - ablk : 0;
        u32 s1[1][4] = { { 0, 0, 0, 0 } };
        if (MODE != 1) {                                      /* every thread, whatever its position is: no divergence around the rounds */
            ctr_words(ctr, i, s1[0]);
            enc_blocks<NR, 1>(s1, rk, lc);
        }
        uint4 x = make_uint4(0, 0, 0, 0);
        if (is_text) {
            const u64 avail = len - 16 * i;
            const u32 nb = avail < 16 ? (u32)avail : 16u;
            const uint4 d = nb == 16 ? in[i] : load_bytes_padded((const unsigned char *)(in + i), nb);
            x = (MODE == 1 && src.rev) ? rev16(d) : d;
            if (MODE != 1) {
                u32 o[4] = { d.x ^ s1[0][0], d.y ^ s1[0][1], d.z ^ s1[0][2], d.w ^ s1[0][3] };
                if (nb < 16) {
#pragma unroll
                    for (u32 w = 0; w < 4; ++w) {
                        const u32 keep = nb >= 4 * w + 4 ? 0xffffffffu : nb <= 4 * w ? 0u : (1u << (8 * (nb - 4 * w))) - 1u;
                        o[w] &= keep;
                    }
                    unsigned char *dst = (unsigned char *)(out + i);
                    for (u32 b = 0; b < nb; ++b) dst[b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
                } else {
                    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (MODE == 0) x = make_uint4(o[0], o[1], o[2], o[3]);   /* GHASH takes the ciphertext */
            }
        } else if (live) {
            x = load_vblock(rest, v < ablk ? v : ablk);       /* (byte-reversed for POLYVAL) */
        }
        acc = x4(acc, x);
    }
    acc = gh_tree<true>(buf, TC, acc, GH_T);
    if (!FOLD) {
        if (threadIdx.x == 0) partial[wg] = acc;
        return;
    }
    if (threadIdx.x == 0) {
        u32 *row = (u32 *)(partial + wg);
        const u32 v[4] = { acc.x, acc.y, acc.z, acc.w };
        u32 old = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) old |= __hip_atomic_exchange(row + q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");
        const unsigned arrived = __hip_atomic_fetch_add(fin.done_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == nwg) __hip_atomic_store(fin.done_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        buf[GT_BUF - 4] = make_uint4(arrived == nwg ? 1u : 0u, 0, 0, 0);
    }
    __syncthreads();
    if (buf[GT_BUF - 4].x != 0) {
        /* the last of the W + 1 to arrive, and it is a chunk workgroup (the preparing workgroup's look ran out): the
         * whole fold here, tables included */
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&g_gcm_chunk_folds, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gcm_combine_body<NR, false>(rk, tb, fin.j0, partial, nwg, scratch, fin.mode == 2 ? fin.fin_build : build, fin.mode,
                                    fin.tag_io, fin.status, fin.ylog, nullptr, fin.m, nullptr, fin.len_aad, fin.len_ct);
    }
    ticket_release(fin.done);
}

/* mode 0: tag = sum_w G_w Y^(W-1-w) ^ Enc(J0) written to tag_io; mode 1: compared with the 16 bytes there */
template <int NR, bool WAIT>
__device__ __forceinline__ void gcm_combine_body(const uaesk_rk &ek, const uaesk_tables &tb, uint4 j0, const uint4 *partial, u32 W,
                                                 const unsigned char *__restrict__ scratch, u32 build,
                                                 int mode, unsigned char *tag_io, int *status, u32 ylog,
                                                 unsigned char *wipe_out, u64 wipe_len, unsigned *done_word,
                                                 u64 len_aad, u64 len_ct, u64 look_ticks)
{
    uint4 *TC = (uint4 *)uaes_lds;
    uint4 *buf = TC + GT_NTAB * 512u;
    u32 *te_plain = (u32 *)(uaes_lds + GHF_LDS);
    /* the W partial hashes are the LAST W of 1024 entries (up to 256 of them the 1024 -> 256 level only copies);
     * requested first: they travel while the tables are made (WAIT: they do not exist yet) */
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (!WAIT && threadIdx.x >= GH_T - W) acc = partial[threadIdx.x - (GH_T - W)];
    if (threadIdx.x < 256) te_plain[threadIdx.x] = tb.te0[threadIdx.x];
    __syncthreads();
    const u32 wave = threadIdx.x >> 6;
    if (wave < 2) {                                           /* wave 0: H; wave 1: Enc(J0) */
        u32 s1[4] = { 0, 0, 0, 0 };
        if (mode >= 3) {                                      /* GCM-SIV: the POLYVAL key k_siv_prep made; j0 is the nonce */
            if (!wave) { const uint4 h = *(const uint4 *)(scratch + GS_SIV_HG); s1[0] = h.x; s1[1] = h.y; s1[2] = h.z; s1[3] = h.w; }
        } else {
            if (wave) { s1[0] = j0.x; s1[1] = j0.y; s1[2] = j0.z; s1[3] = j0.w; }
            if ((wave && mode != 2) || (!wave && build)) plain_encrypt<NR>(te_plain, ek, s1);
        }
        if ((threadIdx.x & 63u) == 0) buf[GT_BUF - 3 + wave] = make_uint4(s1[0], s1[1], s1[2], s1[3]);
    }
    if (!build) {                                             /* a key context holds the tables (k_gcm_ytables): waves 2..15
                                                                 bring them in while wave 1 encrypts J0 */
        const uint4 *gy = (const uint4 *)(scratch + GS_YTAB + (ylog - 10u) * GS_YTAB_SET);
        const uint4 *gh = (const uint4 *)(scratch + GS_TAB4 + 5u * 8192u);      /* the key's table of H itself */
        if (threadIdx.x >= 128u)
            for (u32 i = threadIdx.x - 128u; i < 6u * 512u; i += GH_T - 128u) TC[i] = i < 512u ? gh[i] : gy[i - 512u];
    }
    __syncthreads();
    const uint4 ej0 = buf[GT_BUF - 2];
    if (build) gcm_build_nibble_tables<true>(TC, buf, tb.frob, ylog);    /* Y^256, Y^64, Y^16, Y^4, Y in tables 1..5 */
    if (WAIT) {
        /* the preparing workgroup of a one-launch call: a bounded look at the counter, then it counts itself in like
         * everybody else; only the LAST of the W + 1 folds (GMC_LOOK_TICKS above) */
        if (threadIdx.x == 0) {
            const u64 t0 = wall_clock64();
            while (__hip_atomic_load(done_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != W &&
                   wall_clock64() - t0 < look_ticks)
                __builtin_amdgcn_s_sleep(8);
            const unsigned arrived = __hip_atomic_fetch_add(done_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == W) __hip_atomic_store(done_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     /* for the next call */
            buf[GT_BUF - 4] = make_uint4(arrived == W ? 1u : 0u, 0, 0, 0);
        }
        __syncthreads();
        if (buf[GT_BUF - 4].x == 0) return;                   /* a chunk workgroup is still out: IT folds (k_gcm_chunks) */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x >= GH_T - W) acc = partial[threadIdx.x - (GH_T - W)];
    }
    acc = gh_tree<false>(buf, TC, acc, W);
    if (mode != 2 && threadIdx.x < 4) {                       /* the length block (N6): (S ^ lengths) H, table 0 = H */
        const u64 abits = len_aad * 8, cbits = len_ct * 8;
        const uint4 lb = mode >= 3 ? rev16(make_uint4((u32)abits, (u32)(abits >> 32), (u32)cbits, (u32)(cbits >> 32)))
                                   : make_uint4(bswap32((u32)(abits >> 32)), bswap32((u32)abits),
                                                bswap32((u32)(cbits >> 32)), bswap32((u32)cbits));
        acc = tabmul4q(TC, x4(acc, lb));
    }
    if (mode >= 3) {
        /* GCM-SIV (3: encrypt, 4: decrypt): tag = Enc_k((POLYVAL ^ nonce) with the top bit cleared) (GCM_SIVtag
         * :1453-1460) under the schedule k_siv_prep left in the scratch; written behind the text together with the
         * counter made of it, or compared */
        if (threadIdx.x == 0) {
            unsigned char *sc = (unsigned char *)scratch;
            uaesk_rk rk;
            const u32 *rkw = (const u32 *)(sc + GS_SIV_RK);
#pragma unroll
            for (int i = 0; i < 4 * (NR + 1); ++i) rk.w[i] = rkw[i];
            const uint4 pv = rev16(acc);
            u32 s1[4] = { pv.x ^ j0.x, pv.y ^ j0.y, pv.z ^ j0.z, pv.w & 0x7fffffffu };
            plain_encrypt<NR>(te_plain, rk, s1);
            if (mode == 4) {
                u32 diff = 0;
                for (u32 b = 0; b < 16; ++b) diff |= (u32)tag_io[b] ^ ((s1[b >> 2] >> (8 * (b & 3))) & 0xffu);
                *status = diff ? 0x1A : 0;
            } else {
                for (u32 b = 0; b < 16; ++b) tag_io[b] = (unsigned char)(s1[b >> 2] >> (8 * (b & 3)));
                uaesk_ctr *c = (uaesk_ctr *)(sc + GS_SIV_CTR);
                c->w0 = s1[0]; c->w1 = s1[1]; c->w2 = s1[2]; c->w3 = s1[3] | 0x80000000u;
                c->b8 = 0; c->v0 = 0; c->le32 = 1;
            }
        }
        return;
    }
    if (mode == 2) {
        /* a streamed piece: Y <- Y * H^m ^ P (k_gcm_fold's arithmetic, wave 0; wipe_len carries m) */
        if (threadIdx.x == 0) buf[GT_BUF - 1] = acc;
        __syncthreads();
        if (threadIdx.x < 64) {
            unsigned char *sc = (unsigned char *)scratch;
            const uint4 y4 = *(const uint4 *)(sc + GS_RUN);
            Gf run = gf_from_words(y4.x, y4.y, y4.z, y4.w);
            const uint4 *pw = (const uint4 *)(sc + GS_POW64);
            for (u32 k = 0; k < 64; ++k) {
                if ((wipe_len >> k) & 1) {                    /* wave-uniform */
                    const uint4 h = pw[k];
                    run = wave_gfmul(run, gf_from_words(h.x, h.y, h.z, h.w), threadIdx.x);
                }
            }
            if (threadIdx.x == 0) {
                u32 w[4];
                gf_to_words(run, w);
                const uint4 p4 = buf[GT_BUF - 1];
                *(uint4 *)(sc + GS_RUN) = make_uint4(w[0] ^ p4.x, w[1] ^ p4.y, w[2] ^ p4.z, w[3] ^ p4.w);
            }
        }
        return;
    }
    if (threadIdx.x == 0) {
        acc = x4(acc, ej0);
        const u32 w[4] = { acc.x, acc.y, acc.z, acc.w };
        if (mode == 1) {
            u32 diff = 0;
            for (u32 i = 0; i < 16; ++i) diff |= (u32)tag_io[i] ^ ((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
            buf[GT_BUF - 1] = make_uint4(diff, 0, 0, 0);
        } else {
            for (u32 i = 0; i < 16; ++i) tag_io[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
    /* one-pass decryption (the chunk kernel has written the plaintext already): a wrong tag takes it back HERE, by
     * this one workgroup -- slow for megabytes, but only a forgery pays it, and every good message saves the launch
     * of a wipe kernel behind this one */
    if (wipe_out) {
        __syncthreads();
        if (buf[GT_BUF - 1].x != 0) {
            const u64 head = wipe_len < 16 ? wipe_len : (16 - ((uintptr_t)wipe_out & 15u)) & 15u;
            for (u64 i = threadIdx.x; i < head; i += GH_T) wipe_out[i] = 0;
            uint4 *v = (uint4 *)(wipe_out + head);
            const u64 nv16 = (wipe_len - head) >> 4;
            for (u64 i = threadIdx.x; i < nv16; i += GH_T) v[i] = make_uint4(0, 0, 0, 0);
            for (u64 i = head + (nv16 << 4) + threadIdx.x; i < wipe_len; i += GH_T) wipe_out[i] = 0;
        }
    }
}

template <int NR>
__global__ __launch_bounds__(GH_T) void k_gcm_combine(uaesk_rk ek, uaesk_tables tb, uint4 j0, const uint4 *partial, u32 W,
                                                      const unsigned char *__restrict__ scratch, u32 build,
                                                      int mode, unsigned char *tag_io, int *status, u32 ylog,
                                                      unsigned char *wipe_out, u64 wipe_len, u64 len_aad, u64 len_ct)
{
    gcm_combine_body<NR, false>(ek, tb, j0, partial, W, scratch, build, mode, tag_io, status, ylog, wipe_out, wipe_len, nullptr,
                                len_aad, len_ct);
}

/* key context: the table sets k_gcm_combine would otherwise make in every call (workgroup i: Y = H^(1024 << i)) */
__global__ __launch_bounds__(GH_T) void k_gcm_ytables(uaesk_tables tb, unsigned char *__restrict__ scratch)
{
    uint4 *TC = (uint4 *)uaes_lds;
    uint4 *buf = TC + GT_NTAB * 512u;
    if (threadIdx.x == 0) buf[GT_BUF - 3] = *(const uint4 *)(scratch + GS_H);
    __syncthreads();
    gcm_build_nibble_tables<true>(TC, buf, tb.frob, 10u + blockIdx.x);
    uint4 *gy = (uint4 *)(scratch + GS_YTAB + blockIdx.x * GS_YTAB_SET);
    for (u32 i = threadIdx.x; i < 5u * 512u; i += GH_T) gy[i] = TC[512u + i];
}

static u32 log2_u32(u32 v)
{
    u32 lg = 0;
    while ((1u << lg) < v) ++lg;
    return lg;
}

template <int NR>
static int launch_medium(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c, uint4 j0,
                         const GSrc &src, const void *in, void *out, unsigned char *sc, u32 W, u32 steps, u32 build, int decrypt,
                         unsigned char *tag_io, int *status, unsigned *done_word, TicketScope &ticket, bool hash_only = false)
{
    /* hash_only: an encryption without text (GMAC): the hash-only chunk workgroups, the tag written by the finisher */
    uint4 *partial = (uint4 *)(sc + GS_ACC1);
    hipError_t e;
    GmcFin fin;
    memset(&fin, 0, sizeof fin);
    /* with a counter word from the host layer, encryption and the hash-only pass of a decryption are ONE launch */
    /* (as long as the chunk workgroups are one round; see medium_steps about W = the number of CUs) */
    int cus_f = 0;
    if (uaesk_device_info(&cus_f, nullptr) != 0) cus_f = 0;
    const bool fold = done_word != nullptr && decrypt != 2 && (int)W <= cus_f;
    if (fold) {
        fin.j0 = j0; fin.done_word = done_word; fin.look_ticks = gcm_look_ticks(); fin.tag_io = tag_io; fin.status = status; fin.mode = decrypt ? 1 : 0;
        fin.ylog = 10u + log2_u32(steps);
        fin.len_aad = src.len_aad; fin.len_ct = src.len_ct;
        if (!decrypt) fin.done = ticket.use();
    }
#define GMC_LAUNCH(M, F)                                                                                            \
    do {                                                                                                            \
        e = uaesk_want_lds((const void *)k_gcm_chunks<NR, M, F>, (unsigned)(GSM_LDS_TOTAL)); \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((k_gcm_chunks<NR, M, F>), dim3(W + (F ? 1u : 0u)), dim3(GH_T), GSM_LDS_TOTAL, st, *ek, *tb, *c, src, \
                           (const uint4 *)in, (uint4 *)out, (const unsigned char *)sc, partial, build, steps, fin);  \
    } while (0)
    if (fold) {
        if (decrypt == 0 && !hash_only) GMC_LAUNCH(0, true); else GMC_LAUNCH(1, true);
        return (int)hipGetLastError();
    }
    if (decrypt == 0 && !hash_only) GMC_LAUNCH(0, false); else if (decrypt == 1 || hash_only) GMC_LAUNCH(1, false); else GMC_LAUNCH(2, false);
#undef GMC_LAUNCH
    e = uaesk_want_lds((const void *)k_gcm_combine<NR>, (unsigned)(GHFB_LDS));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_gcm_combine<NR>), dim3(1), dim3(GH_T), GHFB_LDS, st, *ek, *tb, j0, (const uint4 *)partial, W,
                       (const unsigned char *)sc, build, decrypt ? 1 : 0, tag_io, status, 10u + log2_u32(steps),
                       decrypt == 2 ? (unsigned char *)out : nullptr, (u64)src.ct_len, (u64)src.len_aad, (u64)src.len_ct);
    return (int)hipGetLastError();
}


/* ------------------------------------------------------------------------ */
/* host-side drivers                                                          */
/* ------------------------------------------------------------------------ */

struct GPlan {
    u32 logA;       /* 0 = no bulk level */
    u32 needB;
    u64 nv;         /* blocks the plan was made for */
};

static GPlan plan_for(u64 nv)
{
    GPlan p = { 0, 0, nv };
    if (nv <= GH_DIRECT) return p;
    u32 lg = GH_LOGB;
    while (lg < GH_MAXLOG && ((u64)64 << lg) < nv) ++lg;     /* ~64+ steps per lane */
    p.logA = lg;
    p.needB = ((u64)1 << lg) > GH_DIRECT;
    return p;
}

static int run_ghash_levels(hipStream_t st, const GSrc &msg, u64 nv, const GPlan &pl,
                            unsigned char *scratch, int mode, unsigned char *tag_io, int *status)
{
    GSrc cur = msg;
    u64 n = nv;
    hipError_t e;
    if (pl.logA) {
        e = uaesk_want_lds((const void *)k_ghash_pass, (unsigned)(65536));
        if (e != hipSuccess) return (int)e;
        const u64 sA = (u64)1 << pl.logA;
        hipLaunchKernelGGL(k_ghash_pass, dim3((unsigned)(sA / GH_PT)), dim3(GH_PT), 65536, st,
                           cur, n, (const uint4 *)(scratch + GS_TAB8_A), (uint4 *)(scratch + GS_ACC1));
        cur.aad = nullptr; cur.aad_len = 0; cur.ct = scratch + GS_ACC1; cur.ct_len = sA * 16; cur.has_len = 0;
        cur.rev = 0;
        n = sA;
        if (pl.needB) {
            const u64 sB = (u64)1 << GH_LOGB;
            hipLaunchKernelGGL(k_ghash_pass, dim3((unsigned)(sB / GH_PT)), dim3(GH_PT), 65536, st,
                               cur, n, (const uint4 *)(scratch + GS_TAB8_B), (uint4 *)(scratch + GS_ACC2));
            cur.ct = scratch + GS_ACC2; cur.ct_len = sB * 16;
            n = sB;
        }
    }
    e = uaesk_want_lds((const void *)k_ghash_final, (unsigned)(GHF_LDS));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_ghash_final, dim3(1), dim3(GH_T), GHF_LDS, st,
                       cur, n, (const unsigned char *)scratch, mode, tag_io, status);
    return (int)hipGetLastError();
}

extern "C" size_t uaesk_gcm_scratch_bytes(void) { return GS_TOTAL; }
extern "C" size_t uaesk_gcm_stream_scratch_bytes(void) { return GS_SMALL; }

template <int NR>
static int launch_setup(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, uint4 j0,
                        unsigned char *scratch, const GPlan &pl, u32 h_given, uint4 hval, u32 want_pow64 = 0,
                        u32 logF = 0)
{
    hipError_t e = uaesk_want_lds((const void *)k_gcm_setup<NR>, (unsigned)(SETUP_LDS));
    if (e != hipSuccess) return (int)e;
    /* powers actually needed: the bulk tables, and of the last-levels tables H^1024 / H^64 / H^4 / H
     * only those a message of nv blocks reaches (k_ghash_final skips the empty levels)       */
    u32 maxlog = pl.nv > 1024 ? 10u : pl.nv > 256 ? 8u : pl.nv > 64 ? 6u : pl.nv > 16 ? 4u : pl.nv > 4 ? 2u : 0u;
    if (pl.logA > maxlog) maxlog = pl.logA;
    if (pl.needB && GH_LOGB > maxlog) maxlog = GH_LOGB;
    if (logF > maxlog) maxlog = logF;
    /* with the fused kernel's weights to make: a second workgroup for them (k_gcm_setup: tables_wg / weights_wg) */
    hipLaunchKernelGGL((k_gcm_setup<NR>), dim3(logF ? 2u : 1u), dim3(UAES_WG), SETUP_LDS, st, *ek, *tb, j0, scratch,
                       pl.logA, pl.needB, h_given, hval, want_pow64, logF, maxlog);
    return (int)hipGetLastError();
}

/* J0 of a nonce whose length is not 12 bytes (GCMsetup's first branch, micro_aes.c:1145-1149):
 * gHash(H, no AAD, nonce) = GHASH(nonce zero padded || [0]_64 || [8 len]_64), raw, to j0_out16
 * (device).  d_iv = the nonce in device memory.                                            */
extern "C" int uaesk_gcm_j0(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                            const void *d_iv, size_t iv_len, void *scratch, void *j0_out16)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    GSrc msg;
    msg.aad = nullptr; msg.aad_len = 0;
    msg.ct = (const unsigned char *)d_iv; msg.ct_len = iv_len;
    msg.has_len = 1; msg.len_aad = 0; msg.len_ct = iv_len; msg.rev = 0;
    const u64 nv = ((iv_len + 15) >> 4) + 1;
    const GPlan pl = plan_for(nv);
    const uint4 z = make_uint4(0, 0, 0, 0);
    int rc;
    GCM_NR((launch_setup<NR>(st, tb, ek, z, sc, pl, 0, z)));
    if (rc) return rc;
    return run_ghash_levels(st, msg, nv, pl, sc, 2, (unsigned char *)j0_out16, nullptr);
}

/* ------------------------------------------------------------------------ */
/* GCM's rows of the table of arrangements (uaes_plan.h)                        */
/* ------------------------------------------------------------------------ */
/* GCM_CHUNKS: a workgroup takes 1024 * steps GHASH positions, steps the smallest power of two with which ONE round of
 * workgroups covers the text (W <= CUs; W = CUs leaves the preparing workgroup without a CU until the first chunk
 * workgroup retires, which still beats twice the positions per thread on half the CUs).  An encryption takes it up to
 * GMC_ONEPASS_MAX_NV positions (16 MiB): there the chunk workgroups -- CTR and GHASH together -- beat the striped pass
 * and its three launches (35-40 us of fixed cost).  A tag-first decryption hashes with them as far as one round reaches
 * (2^GMC_MAXLOGSTEPS positions per thread: 512 MiB on 256 CUs) and lets the gated CTR kernel write.
 * GCM_TWOPHASE: from there to GMC_TWOPHASE_MAX_NV (128 MiB) an encryption runs the bulk CTR kernel and then the
 * hash-only chunk workgroups over its OUTPUT, which still sits in the last-level cache (64 MiB 84 -> 75 us against the
 * striped kernel); past that the second pass no longer finds it there.
 * GCM_STRIPED: CTR and GHASH in one pass over 8-group stripes (k_gcm_fused): needs a power-of-two number of lanes in
 * the grid (2048 * CUs: 2^19 on MI355X), at least one stripe per workgroup, and the AAD + head blocks in front of one
 * round.  GCM_LEVELS: setup, CTR kernel, the GHASH levels -- takes anything.                                      */
#define GMC_MAXLOGSTEPS 7u
#define GMC_ONEPASS_MAX_NV  ((u64)1 << 20)
#define GMC_TWOPHASE_MAX_NV ((u64)1 << 23)

bool uaesk_arr_on(int id);                                       /* uaes_kernels.hip: the switch of uaes_plan.h */

static int plan_cus(void)
{
    int cus = 0;
    if (uaesk_device_info(&cus, nullptr) != 0 || cus <= 0) cus = 256;      /* (no device: planned for an MI355X) */
    return cus;
}

static u32 medium_steps(u64 nv, int cus)
{
    if (cus < 2) return 0;
    for (u32 lg = 0; lg <= GMC_MAXLOGSTEPS; ++lg)
        if ((nv + (1024ull << lg) - 1) / (1024ull << lg) <= (u64)cus) return 1u << lg;
    return 0;
}

/* the striped region of a text whose keystream starts at counter c: h0 head blocks up to the first group boundary,
 * n8 stripes of eight 256-counter groups from group g_lo on, h1 = blocks in front of the tail */
struct GcmStripes { bool ok; u32 logF; u64 h0, g_lo, n8, h1; int cus; };
static GcmStripes gcm_stripes(const uaesk_ctr &c, u64 len, u64 ablk)
{
    GcmStripes g;
    g.cus = plan_cus();
    const u64 Sl = 2048ull * (u64)g.cus;
    g.logF = 0;
    while (((u64)1 << g.logF) < Sl) ++g.logF;
    const u64 nfull = len / 16;
    const u32 c0 = (u32)c.v0 & 0xffu;
    g.h0 = (256u - c0) & 255u; g.g_lo = c0 ? 1 : 0;
    const u64 groups = (c0 + nfull) / 256;
    g.n8 = groups > g.g_lo ? (groups - g.g_lo) / 8 : 0;
    g.h1 = g.h0 + 2048 * g.n8;
    g.ok = ((u64)1 << g.logF) == Sl && g.logF <= GF_MAXLOG && g.n8 >= (u64)g.cus && ablk + g.h0 <= Sl &&
           !ctr_stripes_cross_a(&c, g.g_lo, g.n8);       /* (one in 2^40 blocks: the two-pass path, whose CTR kernel cuts there) */
    return g;
}

struct GcmPlan { uaes_plan p; GcmStripes s; };
/* dir: 0 encrypt, 1 decrypt tag first (N7), 2 decrypt in one pass, 3 tag only.  piece: a piece of a streamed message
 * (no AAD, no small arrangement; chunk workgroups only with the tables of Y in the stream's scratch: chunks_ready). */
static GcmPlan gcm_plan(int dir, u64 len, u64 aad_len, const uaesk_ctr &c, bool has_word, bool piece = false, bool chunks_ready = true)
{
    GcmPlan g;
    const u64 ablk = (aad_len + 15) >> 4, nvh = ablk + ((len + 15) >> 4);     /* positions the chunk workgroups hash */
    g.s = gcm_stripes(c, len, ablk);
    g.p.arrangement = UAES_ARR_GCM_LEVELS; g.p.launches = 0; g.p.grid = 0; g.p.steps = 0;
    if (!piece && dir != 3 && nvh + 1 <= GSM_MAXNV && uaesk_arr_on(UAES_ARR_GCM_SMALL)) {
        g.p.arrangement = UAES_ARR_GCM_SMALL; g.p.launches = 1; g.p.grid = 1;
        return g;
    }
    const u32 steps = medium_steps(nvh, g.s.cus);
    const bool gmac = len == 0 && dir != 3 && !piece;     /* nothing to encrypt: hash-only chunks whatever the direction */
    const bool chunks_can = dir != 3 && steps && (!piece || (chunks_ready && has_word && nvh >= 1024));
    const bool striped_can = (dir == 0 || dir == 2 || piece) && g.s.ok && uaesk_arr_on(UAES_ARR_GCM_STRIPED);
    if (chunks_can) {
        g.p.steps = steps;
        g.p.grid = (unsigned)((nvh + 1024ull * steps - 1) / (1024ull * steps));
        const bool fold_ok = has_word && (int)g.p.grid <= g.s.cus;       /* one launch: chunk workgroups + fold */
        if ((dir == 1 && !piece) || gmac || nvh <= GMC_ONEPASS_MAX_NV) {
            if (uaesk_arr_on(UAES_ARR_GCM_CHUNKS)) {
                const int dmode = (gmac && dir == 2) ? 1 : dir;            /* (gcm_body) */
                g.p.arrangement = UAES_ARR_GCM_CHUNKS;
                g.p.launches = ((fold_ok && dmode != 2) || piece ? 1 : 2) + ((dmode == 1 && !gmac && !piece) ? 1 : 0);
                return g;
            }
        } else if (nvh <= GMC_TWOPHASE_MAX_NV && uaesk_arr_on(UAES_ARR_GCM_TWOPHASE)) {
            /* encrypt: CTR kernel, hash-only chunks; one-pass decrypt: hash-only chunks, gated CTR kernel, wipe */
            g.p.arrangement = UAES_ARR_GCM_TWOPHASE;
            g.p.launches = 1 + (fold_ok || piece ? 1 : 2) + (dir == 2 && !piece ? 1 : 0);
            return g;
        }
        if (piece && !striped_can && uaesk_arr_on(UAES_ARR_GCM_CHUNKS)) {      /* (a piece that cannot be striped) */
            g.p.arrangement = UAES_ARR_GCM_CHUNKS; g.p.launches = 1;
            return g;
        }
    }
    if (striped_can) {
        g.p.arrangement = UAES_ARR_GCM_STRIPED; g.p.launches = 3 + (dir == 2 ? 2 : 0); g.p.grid = (unsigned)g.s.cus; g.p.steps = 0;
        return g;
    }
    g.p.steps = 0; g.p.grid = 0;
    return g;
}

static void gcm_counter_from_j0(uaesk_ctr &c, const unsigned char j0b[16], u64 block_offset)
{
    memset(&c, 0, sizeof c);
    memcpy(&c.w0, j0b, 4);
    memcpy(&c.w1, j0b + 4, 4);
    c.b8 = j0b[8];
    uint64_t v = 0;                             /* bytes 9..15 of J0: the reference's 56-bit counter (N2); + 1: pre-increment (N4) */
    for (int i = 9; i < 16; ++i) v = (v << 8) | j0b[i];
    c.v0 = (v + 1 + block_offset) & 0x00FFFFFFFFFFFFFFull;
}

int uaesk_plan_gcm(int dir, size_t len, size_t aad_len, unsigned flags, uaes_plan *p)
{
    if (dir < 0 || dir > 3) return (int)hipErrorInvalidValue;
    unsigned char j0b[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 };     /* a 12-byte nonce: J0 ends in 00000001 */
    uaesk_ctr c;
    gcm_counter_from_j0(c, j0b, 0);
    *p = gcm_plan(dir, len, aad_len, c, !((flags >> 2) & 1u)).p;
    if (p->arrangement == UAES_ARR_GCM_LEVELS) {
        const u64 nv = ((aad_len + 15) >> 4) + ((len + 15) >> 4) + 1;
        const GPlan pl = plan_for(nv);
        p->launches = 1 + (dir == 3 ? 0 : 1) + (pl.logA ? 1 : 0) + (pl.needB ? 1 : 0) + 1;
    }
    return 0;
}

/* GCM-SIV: one workgroup up to 2046 POLYVAL positions; then k_siv_prep, the hash-only chunk workgroups whose fold also
 * makes the tag and the counter (needs the counter word: ONE launch, no two-launch form), the CTR kernel -- as far as one
 * round of chunk workgroups reaches (512 MiB on 256 CUs); beyond that POLYVAL by the GHASH levels and k_siv_tag. */
static uaes_plan siv_plan(u64 len, u64 aad_len, bool has_word, bool long_only = false)
{
    uaes_plan p = { UAES_ARR_SIV_LEVELS, 0, 0, 0 };
    const u64 nvh = ((aad_len + 15) >> 4) + ((len + 15) >> 4);
    if (!long_only && nvh + 1 <= GSM_MAXNV && uaesk_arr_on(UAES_ARR_SIV_SMALL)) { p.arrangement = UAES_ARR_SIV_SMALL; p.launches = 1; p.grid = 1; return p; }
    const u32 steps = medium_steps(nvh, plan_cus());
    if (has_word && steps && uaesk_arr_on(UAES_ARR_SIV_CHUNKS)) {
        p.arrangement = UAES_ARR_SIV_CHUNKS; p.launches = 3; p.steps = steps;
        p.grid = (unsigned)((nvh + 1024ull * steps - 1) / (1024ull * steps));
        return p;
    }
    const GPlan pl = plan_for(nvh + 1);
    p.launches = 2 + (pl.logA ? 1 : 0) + (pl.needB ? 1 : 0) + 1 + 1 + 1;      /* prep, setup, levels, final, tag, CTR */
    return p;
}

int uaesk_plan_siv(int dir, size_t len, size_t aad_len, unsigned flags, uaes_plan *p)
{
    if (dir < 0 || dir > 1) return (int)hipErrorInvalidValue;
    *p = siv_plan(len, aad_len, !((flags >> 2) & 1u));
    return 0;
}

/* keyed != 0: `scratch` belongs to a key context whose tables (nibble tables, H^S table, Y/Z powers, all
 * H^(2^k)) uaesk_gcm_key_tables built; then only Enc(J0) is computed per message unless the text needs
 * a size-dependent bulk table (two-pass texts over 512 KiB), for which the full setup runs as usual. */
static int gcm_body(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                    int decrypt, const uint8_t *j0_16,
                    const void *aad, size_t aad_len,
                    const void *in, size_t len, void *out,
                    void *scratch, int *status, int keyed)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    /* an armed completion ticket (uaes_device.h) may ride on a ONE-launch encryption only; taking it here also
     * keeps the building blocks below (uaesk_ctr_xcrypt) from picking it up in the middle of a longer sequence */
    TicketScope ticket;
    /* ... and so may the counter word a one-launch arrangement of several workgroups needs (uaesk_done_word_arm) */
    unsigned *const done_word = uaesk_done_word_take();
    /* J0 = nonce || 00000001 for the 12-byte nonce (GCMsetup, micro_aes.c:1150-1151), or
     * GHASH(nonce) (uaesk_gcm_j0); the host layer passes the 16 bytes                     */
    uint4 j0;
    unsigned char j0b[16];
    memcpy(j0b, j0_16, 16);
    memcpy(&j0, j0b, 16);
    /* keystream counter starts at J0 + 1 (pre-increment, N4); the reference's
     * incBlock carries through bytes 15..9, so this is a 56-bit counter      */
    uaesk_ctr c;
    gcm_counter_from_j0(c, j0b, 0);

    GSrc msg;
    msg.aad = (const unsigned char *)aad; msg.aad_len = aad_len;
    msg.ct = (const unsigned char *)(decrypt ? in : out); msg.ct_len = len;
    msg.has_len = 1; msg.len_aad = aad_len; msg.len_ct = len; msg.rev = 0;
    const u64 nv = ((aad_len + 15) >> 4) + ((len + 15) >> 4) + 1;
    const u64 nfull = len / 16, ablk = (aad_len + 15) >> 4;

    int rc = 0;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const GcmPlan plan = gcm_plan(decrypt, len, aad_len, c, done_word != nullptr);
    const GcmStripes &sp = plan.s;

    switch (plan.p.arrangement) {
    case UAES_ARR_GCM_SMALL: {
        /* short message: one workgroup does all of it (k_gcm_small).  The nibble tables of this key: a key context has
         * them, a one-shot call makes them inside the same launch */
        const u32 build = keyed ? 0u : 1u;
        GSrc sm = msg;
        sm.ct = (const unsigned char *)in;     /* the kernel reads the text itself */
        if (decrypt)                               /* (the host layer arms a ticket only with a host-visible status word) */
            return launch_small_nr<true>(nr, st, tb, ek, &c, j0, sm, in, out, sc, (unsigned char *)in + len, status, build, ticket.use());
        return launch_small_nr<false>(nr, st, tb, ek, &c, j0, sm, in, out, sc, (unsigned char *)out + len, nullptr, build, ticket.use());
    }
    case UAES_ARR_GCM_CHUNKS:
    case UAES_ARR_GCM_TWOPHASE: {
        /* chunk workgroups + a fold (k_gcm_chunks; two launches with k_gcm_combine where the one-launch arrangement
         * cannot be used), tables made in the kernels for a one-shot call.  Decrypt mode 1 hashes first and lets the
         * gated CTR kernel write; mode 2 decrypts in the chunk kernel and zeroes the output if the tag turns out wrong.
         * TWOPHASE: an encryption runs the bulk CTR kernel and then hashes its OUTPUT with the hash-only chunks; a
         * one-pass decryption takes the tag-first order there (the faster one in this range) and keeps its contract
         * -- a ZEROED output on a forgery -- with the wipe kernel the striped order launches too */
        const bool gmac = len == 0;
        const bool two = plan.p.arrangement == UAES_ARR_GCM_TWOPHASE;
        const u32 steps = plan.p.steps, W = plan.p.grid, build = keyed ? 0u : 1u;
        GSrc sm = msg;
        sm.ct = (const unsigned char *)in;
        sm.has_len = 0;
        const int dmode = ((gmac || two) && decrypt == 2) ? 1 : decrypt;   /* (without a text there is nothing a one-pass order could write early) */
        const bool hash_only = (gmac || two) && !decrypt;
        const void *text = in;
        if (two && !decrypt) {
            if ((rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr)) != 0) return rc;
            sm.ct = (const unsigned char *)out;
            text = out;
        }
        unsigned char *tagp = decrypt ? (unsigned char *)in + len : (unsigned char *)out + len;
        GCM_NR((launch_medium<NR>(st, tb, ek, &c, j0, sm, text, out, sc, W, steps, build, dmode, tagp, status, done_word, ticket, hash_only)));
        if (rc && dmode == 2) (void)hipMemsetAsync(out, 0, len, st);   /* the chunk kernel may have been enqueued: see STRIPED */
        if (rc || !decrypt) return rc;
        if (dmode == 1) {
            rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, status);
            if (!rc && two && decrypt == 2) {                    /* the one-pass contract: zeroed on a forgery */
                hipLaunchKernelGGL(k_wipe_if_failed, dim3((unsigned)sp.cus * 2u), dim3(1024), 0, st, status,
                                   (unsigned char *)out, (u64)len);
                rc = (int)hipGetLastError();
            }
        }
        return rc;                             /* mode 2: the combine kernel takes a forgery's plaintext back itself */
    }
    case UAES_ARR_GCM_STRIPED: {
        /* Long text: CTR and GHASH in one pass (k_gcm_fused).  Decrypt authenticates BEFORE it writes (N7), so by
         * default it never comes here; decrypt == 2 (the caller accepts a zeroed output on failure, or the output is a
         * private staging buffer) runs one pass as well: the striped region is decrypted while its ciphertext is
         * hashed, the tag is checked over [T][tail][lengths] with the tail still ciphertext, then the tail is
         * decrypted (gated on the status) and, on a mismatch, everything written is zeroed.    */
        GSrc fin;                                   /* [T][tail][lengths] */
        fin.aad = sc + GS_T; fin.aad_len = 16;
        fin.ct = (const unsigned char *)(decrypt ? in : out) + sp.h1 * 16; fin.ct_len = len - sp.h1 * 16;
        fin.has_len = 1; fin.len_aad = aad_len; fin.len_ct = len; fin.rev = 0;
        const u64 nvf = 1 + ((fin.ct_len + 15) >> 4) + 1;
        const GPlan plf = plan_for(nvf);
        GSrc front = msg;                           /* [AAD][head] in front of the striped region */
        front.ct_len = sp.h0 * 16; front.has_len = 0;
        if (keyed) { GCM_NR((launch_ej0<NR>(st, tb, ek, j0, sc))); }
        else       { GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, plf, 0, z, 0, sp.logF))); }
        if (rc) return rc;
        if (!decrypt) {
            rc = launch_fused_nr<false>(nr, st, tb, ek, &c, in, out, (unsigned)sp.cus, sp.g_lo, sp.n8, sp.h1, nfull,
                                        (u32)(len % 16), front, ablk + sp.h0, sc);
            if (rc) return rc;
            return run_ghash_levels(st, fin, nvf, plf, sc, 0, (unsigned char *)out + len, nullptr);
        }
        /* the fused kernel leaves the tail alone (h1 = nfull, no ragged bytes) */
        rc = launch_fused_nr<true>(nr, st, tb, ek, &c, in, out, (unsigned)sp.cus, sp.g_lo, sp.n8, nfull, nfull, 0,
                                   front, ablk + sp.h0, sc);
        if (rc) return rc;                          /* nothing has been written yet */
        /* from here on `out` holds plaintext nobody has authenticated: if anything below cannot be
         * enqueued, the conditional wipe would never run -- wipe unconditionally instead (N7) */
        rc = run_ghash_levels(st, fin, nvf, plf, sc, 1, (unsigned char *)in + len, status);
        if (!rc && len > sp.h1 * 16) {
            uaesk_ctr ct = c;
            ct.v0 = (c.v0 + sp.h1) & 0x00FFFFFFFFFFFFFFull;
            rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &ct, (const unsigned char *)in + sp.h1 * 16,
                                  (unsigned char *)out + sp.h1 * 16, len - sp.h1 * 16, status);
        }
        if (!rc) {
            hipLaunchKernelGGL(k_wipe_if_failed, dim3((unsigned)sp.cus * 2u), dim3(1024), 0, st, status,
                               (unsigned char *)out, (u64)len);
            rc = (int)hipGetLastError();
        }
        if (rc) (void)hipMemsetAsync(out, 0, len, st);
        return rc;
    }
    default:
        break;
    }

    /* GCM_LEVELS: setup (or Enc(J0) alone where a key context holds every table this text needs), the CTR kernel, the
     * GHASH levels.  A one-pass decryption that lands here takes the tag-first order: same results, N7 kept. */
    const GPlan pl = plan_for(nv);
    if (keyed && pl.logA == 0) {
        GCM_NR((launch_ej0<NR>(st, tb, ek, j0, sc)));
    } else {
        /* (a key context keeps its tables: the full setup rewrites them with the same values) */
        const u32 all = keyed ? 1u : 0u;
        const u32 logFk = (keyed && ((u64)1 << sp.logF) == 2048ull * (u64)sp.cus && sp.logF <= GF_MAXLOG) ? sp.logF : 0u;
        GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, pl, 0, z, all, logFk)));
    }
    if (rc) return rc;

    /* decrypt == 3: only the tag of (aad, `in` as ciphertext), 16 bytes written at `status`; nothing is decrypted.
     * The host layer compares a TRUNCATED tag (GCM_TAG_LEN < 16, micro_aes.c:1204) itself and then runs
     * uaesk_gcm_ctr -- the fused kernels compare all sixteen bytes on the device.                        */
    if (decrypt == 3) return run_ghash_levels(st, msg, nv, pl, sc, 0, (unsigned char *)status, nullptr);

    if (!decrypt) {
        rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr);
        if (rc) return rc;
        return run_ghash_levels(st, msg, nv, pl, sc, 0, (unsigned char *)out + len, nullptr);
    }
    rc = run_ghash_levels(st, msg, nv, pl, sc, 1, (unsigned char *)in + len, status);
    if (rc) return rc;
    return uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, status);
}

extern "C" int uaesk_gcm(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                         int decrypt, const uint8_t *j0_16,
                         const void *aad, size_t aad_len,
                         const void *in, size_t len, void *out,
                         void *scratch, int *status)
{
    return gcm_body(stream, tb, nr, ek, decrypt, j0_16, aad, aad_len, in, len, out, scratch, status, 0);
}

/* the CTR half alone: in -> out under the keystream that starts at J0 + 1 (CTR_cipher mode CCM_GCM, :938-940) */
extern "C" int uaesk_gcm_ctr(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                             const uint8_t *j0_16, const void *in, size_t len, void *out)
{
    uaesk_ctr c;
    memset(&c, 0, sizeof c);
    memcpy(&c.w0, j0_16, 4);
    memcpy(&c.w1, j0_16 + 4, 4);
    c.b8 = j0_16[8];
    uint64_t v = 0;
    for (int i = 9; i < 16; ++i) v = (v << 8) | j0_16[i];
    c.v0 = (v + 1) & 0x00FFFFFFFFFFFFFFull;
    return uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr);
}

extern "C" int uaesk_gcm_keyed(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                               int decrypt, const uint8_t *j0_16,
                               const void *aad, size_t aad_len,
                               const void *in, size_t len, void *out,
                               void *key_scratch, int *status)
{
    return gcm_body(stream, tb, nr, ek, decrypt, j0_16, aad, aad_len, in, len, out, key_scratch, status, 1);
}

/* everything of the GCM setup that depends on the key only, into a key context's scratch */
extern "C" int uaesk_gcm_key_tables(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, void *key_scratch)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)key_scratch;
    GPlan pl = plan_for(2048);                 /* no bulk table; the nibble tables of all four last levels */
    int cus = 0;
    u32 logF = 0;
    if (uaesk_device_info(&cus, nullptr) == 0 && cus > 0) {
        const u64 Sl = 2048ull * (u64)cus;
        while (((u64)1 << logF) < Sl) ++logF;
        if (((u64)1 << logF) != Sl || logF > GF_MAXLOG) logF = 0;
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    int rc;
    GCM_NR((launch_setup<NR>(st, tb, ek, z, sc, pl, 0, z, 1, logF)));
    if (rc) return rc;
    hipError_t e = uaesk_want_lds((const void *)k_gcm_ytables, (unsigned)GHF_LDS);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_gcm_ytables, dim3(GS_YTAB_SETS), dim3(GH_T), GHF_LDS, st, *tb, sc);
    return (int)hipGetLastError();
}

extern "C" int uaesk_ghash(void *stream, const uaesk_tables *tb, const uint8_t *H_host,
                           const void *aad, size_t aad_len, const void *ct, size_t ct_len,
                           void *scratch, void *gh_out16)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    GSrc msg;
    msg.aad = (const unsigned char *)aad; msg.aad_len = aad_len;
    msg.ct = (const unsigned char *)ct; msg.ct_len = ct_len;
    msg.has_len = 1; msg.len_aad = aad_len; msg.len_ct = ct_len; msg.rev = 0;
    const u64 nv = ((aad_len + 15) >> 4) + ((ct_len + 15) >> 4) + 1;
    const GPlan pl = plan_for(nv);
    uint4 h;
    memcpy(&h, H_host, 16);
    uaesk_rk dummy_rk;
    memset(&dummy_rk, 0, sizeof dummy_rk);
    int rc = launch_setup<10>(st, tb, &dummy_rk, make_uint4(0, 0, 0, 0), sc, pl, 1, h);
    if (rc) return rc;
    return run_ghash_levels(st, msg, nv, pl, sc, 2, (unsigned char *)gh_out16, nullptr);
}


/* ------------------------------------------------------------------------ */
/* sharded GCM (multi-GPU): weighted partial GHASH of one shard               */
/* ------------------------------------------------------------------------ */
/* out16 <- P * H^e (^ Enc(J0) on the first shard), P = raw GHASH of the shard at
 * scratch+GS_PART, H^(2^k) at scratch+GS_POW64; one wave (square-and-multiply
 * with the wave-cooperative product).                                        */
__global__ __launch_bounds__(64) void k_gcm_weight(const unsigned char *__restrict__ scratch, u64 e,
                                                   u32 add_ej0, u32 zero, unsigned char *__restrict__ out16)
{
    const uint4 p4 = *(const uint4 *)(scratch + GS_PART);
    Gf acc = gf_from_words(p4.x, p4.y, p4.z, p4.w);
    if (zero) { acc.hi = 0; acc.lo = 0; }
    const uint4 *pw = (const uint4 *)(scratch + GS_POW64);
    for (u32 k = 0; k < 64; ++k) {
        if ((e >> k) & 1) {                                   /* wave-uniform */
            const uint4 h = pw[k];
            acc = wave_gfmul(acc, gf_from_words(h.x, h.y, h.z, h.w), threadIdx.x);
        }
    }
    if (threadIdx.x == 0) {
        u32 w[4];
        gf_to_words(acc, w);
        if (add_ej0) {
            const uint4 ej = *(const uint4 *)(scratch + GS_EJ0);
            w[0] ^= ej.x; w[1] ^= ej.y; w[2] ^= ej.z; w[3] ^= ej.w;
        }
        for (u32 i = 0; i < 16; ++i) out16[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
    }
}

/* The GHASH input of a message is the block sequence [AAD][CT][lengths], M
 * blocks; GHASH = sum X_v * H^(M-v).  A shard owning blocks [lo, hi) computes
 * P = sum X_v * H^(hi-v) with the ordinary levels and weights it by H^(M-hi);
 * the tag is the XOR of all shards' results (Enc(J0) rides on the first).
 * Shards are 16-byte aligned slices of the text; the first one also
 * carries the AAD, the last one the length block (with the TOTAL lengths).
 *
 * mode 0: encrypt the shard (in -> out, keystream block J0 + 1 + shard_offset/16 onwards: the pre-increment of
 *         CTR_cipher's CCM_GCM flavour, micro_aes.c:938-941, with incBlock's 56-bit carry, :421-427) and hash what
 *         was written;  mode 1: hash `in` as ciphertext, write nothing (`out` unused);  mode 2: decrypt in -> out
 *         and hash `in` -- the caller owns the N7 decision (micro_aes.c:1200-1208): `out` is written before any tag
 *         is known.  A shard long enough for the striped kernel gets CTR and GHASH in one pass (k_gcm_fused, as a
 *         whole message does); in == out is allowed.                                                            */
extern "C" int uaesk_gcm_shard(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int mode,
                               const uint8_t *nonce12, const void *aad, uint64_t total_aad_len,
                               const void *in, size_t shard_len, uint64_t shard_offset,
                               uint64_t total_len, void *out, void *scratch, void *partial16)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    if (shard_offset % 16 || shard_offset + shard_len > total_len || mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
    const bool first = shard_offset == 0, last = shard_offset + shard_len == total_len;
    if (!last && shard_len % 16) return (int)hipErrorInvalidValue;
    const u64 a_blk = (total_aad_len + 15) >> 4, c_blk = (total_len + 15) >> 4;
    const u64 m_total = a_blk + c_blk + 1;
    const u64 lo = first ? 0 : a_blk + shard_offset / 16;
    const u64 hi = a_blk + ((shard_offset + shard_len + 15) >> 4) + (last ? 1 : 0);
    const u64 nv = hi - lo;

    uint4 j0;
    unsigned char j0b[16];
    memcpy(j0b, nonce12, 12);
    j0b[12] = j0b[13] = j0b[14] = 0; j0b[15] = 1;
    memcpy(&j0, j0b, 16);
    uaesk_ctr c;
    memset(&c, 0, sizeof c);
    memcpy(&c.w0, j0b, 4);
    memcpy(&c.w1, j0b + 4, 4);
    c.b8 = j0b[8];
    {
        uint64_t v = 0;                         /* bytes 9..15 of J0: the reference's 56-bit counter (N2) */
        for (int i = 9; i < 16; ++i) v = (v << 8) | j0b[i];
        c.v0 = (v + 1 + shard_offset / 16) & 0x00FFFFFFFFFFFFFFull;
    }

    GSrc msg;
    msg.aad = first ? (const unsigned char *)aad : nullptr;
    msg.aad_len = first ? total_aad_len : 0;
    msg.ct = (const unsigned char *)(mode == 0 ? out : in); msg.ct_len = shard_len;
    msg.has_len = last ? 1 : 0;
    msg.len_aad = total_aad_len; msg.len_ct = total_len; msg.rev = 0;
    const uint4 z = make_uint4(0, 0, 0, 0);
    int rc;

    if (mode != 1) {                            /* CTR and GHASH of a long shard in one pass: gcm_body's conditions */
        const u64 nfull = shard_len / 16, ablk = (msg.aad_len + 15) >> 4;
        const GcmStripes sp = gcm_stripes(c, shard_len, ablk);
        const int cus = sp.cus;
        const u32 logF = sp.logF;
        const u64 h0 = sp.h0, g_lo = sp.g_lo, n8 = sp.n8;
        if (sp.ok) {
            const u64 h1 = h0 + 2048 * n8;
            GSrc fin;                                   /* [T][tail]([lengths] on the last shard) */
            fin.aad = sc + GS_T; fin.aad_len = 16;
            fin.ct = msg.ct + h1 * 16; fin.ct_len = shard_len - h1 * 16;
            fin.has_len = msg.has_len; fin.len_aad = total_aad_len; fin.len_ct = total_len; fin.rev = 0;
            const u64 nvf = 1 + ((fin.ct_len + 15) >> 4) + (last ? 1 : 0);
            const GPlan plf = plan_for(nvf);
            GSrc front = msg;                           /* [AAD][head] in front of the striped region */
            front.ct_len = h0 * 16; front.has_len = 0;
            GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, plf, 0, z, 1, logF)));
            if (rc) return rc;
            if (mode == 0) {
                rc = launch_fused_nr<false>(nr, st, tb, ek, &c, in, out, (unsigned)cus, g_lo, n8, h1, nfull,
                                            (u32)(shard_len % 16), front, ablk + h0, sc);
                if (rc) return rc;
                rc = run_ghash_levels(st, fin, nvf, plf, sc, 2, sc + GS_PART, nullptr);
            } else {
                /* the fused kernel leaves the tail alone: it is hashed as ciphertext first (in may be out) */
                rc = launch_fused_nr<true>(nr, st, tb, ek, &c, in, out, (unsigned)cus, g_lo, n8, nfull, nfull, 0,
                                           front, ablk + h0, sc);
                if (rc) return rc;
                rc = run_ghash_levels(st, fin, nvf, plf, sc, 2, sc + GS_PART, nullptr);
                if (!rc && shard_len > h1 * 16) {
                    uaesk_ctr ct = c;
                    ct.v0 = (c.v0 + h1) & 0x00FFFFFFFFFFFFFFull;
                    rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &ct, (const unsigned char *)in + h1 * 16,
                                          (unsigned char *)out + h1 * 16, shard_len - h1 * 16, nullptr);
                }
            }
            if (rc) return rc;
            hipLaunchKernelGGL(k_gcm_weight, dim3(1), dim3(64), 0, st, (const unsigned char *)sc, m_total - hi,
                               (u32)first, 0u, (unsigned char *)partial16);
            return (int)hipGetLastError();
        }
    }

    const GPlan pl = plan_for(nv ? nv : 1);
    GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, pl, 0, z, 1)));
    if (rc) return rc;
    if (mode == 0 && shard_len) {
        rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, shard_len, nullptr);
        if (rc) return rc;
    }
    if (nv) {
        rc = run_ghash_levels(st, msg, nv, pl, sc, 2, sc + GS_PART, nullptr);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_gcm_weight, dim3(1), dim3(64), 0, st, (const unsigned char *)sc, m_total - hi,
                       (u32)first, (u32)(nv == 0), (unsigned char *)partial16);
    rc = (int)hipGetLastError();
    if (!rc && mode == 2 && shard_len)          /* behind the hash: the ciphertext may be overwritten now */
        rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, shard_len, nullptr);
    return rc;
}

extern "C" int uaesk_gcm_partial(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                                 const uint8_t *nonce12, const void *aad, uint64_t total_aad_len,
                                 const void *ct_shard, size_t shard_len, uint64_t shard_offset,
                                 uint64_t total_len, void *scratch, void *partial16)
{
    return uaesk_gcm_shard(stream, tb, nr, ek, 1, nonce12, aad, total_aad_len, ct_shard, shard_len, shard_offset,
                           total_len, nullptr, scratch, partial16);
}


/* ------------------------------------------------------------------------ */
/* streamed GCM: the running GHASH value Y lives at scratch+GS_RUN            */
/* ------------------------------------------------------------------------ */
/* Y <- Y * H^m ^ P  (P = raw GHASH of the m blocks just absorbed, at GS_PART):
 * Horner over whole pieces of the message.  One wave.                        */
__global__ __launch_bounds__(64) void k_gcm_fold(unsigned char *__restrict__ scratch, u64 m, u32 reset)
{
    const uint4 y4 = *(const uint4 *)(scratch + GS_RUN);
    Gf acc = gf_from_words(y4.x, y4.y, y4.z, y4.w);
    if (reset) { acc.hi = 0; acc.lo = 0; }
    const uint4 *pw = (const uint4 *)(scratch + GS_POW64);
    for (u32 k = 0; k < 64; ++k) {
        if ((m >> k) & 1) {                                   /* wave-uniform */
            const uint4 h = pw[k];
            acc = wave_gfmul(acc, gf_from_words(h.x, h.y, h.z, h.w), threadIdx.x);
        }
    }
    if (threadIdx.x == 0) {
        u32 w[4];
        gf_to_words(acc, w);
        if (m) {
            const uint4 p4 = *(const uint4 *)(scratch + GS_PART);
            w[0] ^= p4.x; w[1] ^= p4.y; w[2] ^= p4.z; w[3] ^= p4.w;
        }
        *(uint4 *)(scratch + GS_RUN) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

/* tag = Y ^ Enc(J0): write it (mode 0) or compare with tag_io (mode 1) */
__global__ __launch_bounds__(64) void k_gcm_stream_tag(const unsigned char *__restrict__ scratch, int mode,
                                                       unsigned char *tag_io, int *status)
{
    if (threadIdx.x != 0) return;
    const uint4 t = x4(*(const uint4 *)(scratch + GS_RUN), *(const uint4 *)(scratch + GS_EJ0));
    const u32 w[4] = { t.x, t.y, t.z, t.w };
    if (mode == 1) {
        u32 diff = 0;
        for (u32 i = 0; i < 16; ++i) diff |= (u32)tag_io[i] ^ ((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
        *status = diff ? 0x1A : 0;
    } else {
        for (u32 i = 0; i < 16; ++i) tag_io[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
    }
}

/* Absorb one piece of the GHASH input into the running value.  kind 0: the AAD
 * (first piece: Y restarts at 0), kind 1: `len` bytes of ciphertext (a multiple of
 * 16 unless it is the last piece), kind 2: the length block with the totals.   */
extern "C" int uaesk_gcm_stream_absorb(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                                       const uint8_t *nonce12, int kind, const void *data, size_t len,
                                       uint64_t total_aad_len, uint64_t total_ct_len, void *scratch,
                                       unsigned *plan_state)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    uint4 j0;
    unsigned char j0b[16];
    memcpy(j0b, nonce12, 12);
    j0b[12] = j0b[13] = j0b[14] = 0; j0b[15] = 1;
    memcpy(&j0, j0b, 16);

    GSrc msg;
    memset(&msg, 0, sizeof msg);
    if (kind == 0) { msg.aad = (const unsigned char *)data; msg.aad_len = len; }
    else if (kind == 1) { msg.ct = (const unsigned char *)data; msg.ct_len = len; }
    else { msg.has_len = 1; msg.len_aad = total_aad_len; msg.len_ct = total_ct_len; }
    const u64 nv = kind == 2 ? 1 : (len + 15) >> 4;
    const GPlan pl = plan_for(nv ? nv : 1);
    const uint4 z = make_uint4(0, 0, 0, 0);
    int rc = 0;
    /* the tables in this stream's scratch survive between pieces: rebuild them only when the
     * level plan asks for one that is not there (plan_state: bit 31 valid, bit 8 B table, low byte logA) */
    const unsigned have = plan_state ? *plan_state : 0u;
    const bool ok = (have >> 31) && (!pl.logA || (have & 0xffu) == pl.logA) && (!pl.needB || ((have >> 8) & 1u));
    if (!ok) {
        GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, pl, 0, z, 1)));
        if (rc) return rc;
        if (plan_state) *plan_state = 0x80000000u | (*plan_state & 0x200u) | (pl.needB ? 0x100u : 0u) | pl.logA;   /* (bit 9: the striped
                                         * kernel's tables, uaesk_gcm_stream_piece -- another setup leaves them alone) */
    }
    if (nv) {
        rc = run_ghash_levels(st, msg, nv, pl, sc, 2, sc + GS_PART, nullptr);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_gcm_fold, dim3(1), dim3(64), 0, st, sc, nv, (u32)(kind == 0));
    return (int)hipGetLastError();
}

/* One piece of a streamed message, CTR and GHASH in ONE pass when the piece is long enough for the striped kernel
 * (8 MiB on 256 CUs): what uaesk_gcm_shard does for a slice, with the running value folded (k_gcm_fold) instead of a
 * weighted share.  Returns 1 if it was not taken (the caller then runs uaesk_ctr_xcrypt + uaesk_gcm_stream_absorb as
 * before: CTR kernel, GHASH levels, fold -- two passes over the piece, 415 GiB/s for one 256 MiB piece), 0 when the
 * work is enqueued, else a hipError_t.  done_bytes = text absorbed so far (a multiple of 16); a piece of a length
 * that is no multiple of 16 is the message's last.  The stream's scratch must hold uaesk_gcm_scratch_bytes().
 * plan_state bit 9: the striped kernel's tables of this key are in the scratch.                                 */
template <int NR>
static int launch_stream_chunks(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, const uaesk_ctr *c, const GSrc &src,
                                const void *in, void *out, unsigned char *sc, u32 W, u32 steps, int decrypt, const GmcFin &fin,
                                bool hash_only = false)
{
    hipError_t e;
    uint4 *partial = (uint4 *)(sc + GS_ACC1);
    if (hash_only) {                                          /* `in` = the piece's CIPHERTEXT; nothing is written */
        if ((e = uaesk_want_lds((const void *)k_gcm_chunks<NR, 1, true>, (unsigned)(GSM_LDS_TOTAL))) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_chunks<NR, 1, true>), dim3(W + 1u), dim3(GH_T), GSM_LDS_TOTAL, st, *ek, *tb, *c, src,
                           (const uint4 *)in, (uint4 *)out, (const unsigned char *)sc, partial, 0u, steps, fin);
    } else if (decrypt) {
        if ((e = uaesk_want_lds((const void *)k_gcm_chunks<NR, 2, true>, (unsigned)(GSM_LDS_TOTAL))) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_chunks<NR, 2, true>), dim3(W + 1u), dim3(GH_T), GSM_LDS_TOTAL, st, *ek, *tb, *c, src,
                           (const uint4 *)in, (uint4 *)out, (const unsigned char *)sc, partial, 0u, steps, fin);
    } else {
        if ((e = uaesk_want_lds((const void *)k_gcm_chunks<NR, 0, true>, (unsigned)(GSM_LDS_TOTAL))) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_chunks<NR, 0, true>), dim3(W + 1u), dim3(GH_T), GSM_LDS_TOTAL, st, *ek, *tb, *c, src,
                           (const uint4 *)in, (uint4 *)out, (const unsigned char *)sc, partial, 0u, steps, fin);
    }
    return (int)hipGetLastError();
}

/* ... and a piece of 16 KiB .. 8 MiB is ONE launch: the chunk workgroups of the medium-sized one-shot call (k_gcm_chunks,
 * FOLD) with a finisher that folds the piece's raw hash into the running value (GmcFin.mode 2) -- needs done_word (a
 * word that is zero between calls) and the key's nibble tables in the scratch (the stream's first absorb made them). */
extern "C" int uaesk_gcm_stream_piece(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                                      const uint8_t *nonce12, int decrypt, const void *in, size_t len,
                                      uint64_t done_bytes, void *out, void *scratch, unsigned *plan_state,
                                      unsigned *done_word)
{
    hipStream_t st = S(stream);
    unsigned char *sc = (unsigned char *)scratch;
    if (done_bytes % 16 || !plan_state) return (int)hipErrorInvalidValue;
    uint4 j0;
    unsigned char j0b[16];
    memcpy(j0b, nonce12, 12);
    j0b[12] = j0b[13] = j0b[14] = 0; j0b[15] = 1;
    memcpy(&j0, j0b, 16);
    uaesk_ctr c;
    gcm_counter_from_j0(c, j0b, done_bytes / 16);
    /* as for a one-shot call (gcm_plan): up to 16 MiB the chunk workgroups (CTR and GHASH together), from there to the
     * end of their round two phases -- the bulk CTR kernel and the hash-only chunk workgroups over the piece's
     * ciphertext (a decryption hashes first: in may be out) --, beyond that the striped pass; a piece none of them
     * takes goes back to the caller (1: the absorb path) */
    const GcmPlan plan = gcm_plan(decrypt ? 2 : 0, len, 0, c, done_word != nullptr, true, (*plan_state >> 31) != 0);
    const GcmStripes &sp = plan.s;
    const int cus = sp.cus;
    const u64 nfull = len / 16, h0 = sp.h0, g_lo = sp.g_lo, n8 = sp.n8, h1 = sp.h1;
    const u32 logF = sp.logF;
    if (plan.p.arrangement == UAES_ARR_GCM_LEVELS) return 1;
    if (plan.p.arrangement != UAES_ARR_GCM_STRIPED) {
        const bool two = plan.p.arrangement == UAES_ARR_GCM_TWOPHASE;
        const u64 nvp = ((u64)len + 15) >> 4;
        const u32 steps = plan.p.steps, W = plan.p.grid;
        GSrc src;
        memset(&src, 0, sizeof src);
        src.ct = (const unsigned char *)in; src.ct_len = len;          /* the kernel reads the text itself */
        GmcFin fin;
        memset(&fin, 0, sizeof fin);
        fin.done_word = done_word; fin.look_ticks = gcm_look_ticks(); fin.mode = 2; fin.ylog = 10u + log2_u32(steps); fin.fin_build = 1; fin.m = nvp;
        const void *text = in;
        int rc = 0;
        if (two && !decrypt) {
            if ((rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr)) != 0) return rc;
            src.ct = (const unsigned char *)out;
            text = out;
        }
        GCM_NR((launch_stream_chunks<NR>(st, tb, ek, &c, src, text, out, sc, W, steps, decrypt, fin, two)));
        if (!rc && two && decrypt) rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr);
        return rc;
    }
    GSrc msg;
    memset(&msg, 0, sizeof msg);
    msg.ct = (const unsigned char *)(decrypt ? in : out); msg.ct_len = len;
    GSrc fin;                                                   /* [T][tail] */
    memset(&fin, 0, sizeof fin);
    fin.aad = sc + GS_T; fin.aad_len = 16;
    fin.ct = msg.ct + h1 * 16; fin.ct_len = len - h1 * 16;
    const u64 nvf = 1 + ((fin.ct_len + 15) >> 4);
    const GPlan plf = plan_for(nvf);
    GSrc front = msg;                                           /* [head] in front of the striped region */
    front.ct_len = h0 * 16;
    const uint4 z = make_uint4(0, 0, 0, 0);
    int rc;
    if ((*plan_state >> 31) && ((*plan_state >> 9) & 1u) && (!plf.logA || (*plan_state & 0xffu) == plf.logA) &&
        (!plf.needB || ((*plan_state >> 8) & 1u))) {           /* the tables are there: Enc(J0) again and T <- 0 */
        GCM_NR((launch_ej0<NR>(st, tb, ek, j0, sc)));
    } else {
        GCM_NR((launch_setup<NR>(st, tb, ek, j0, sc, plf, 0, z, 1, logF)));
        if (!rc) *plan_state = 0x80000000u | 0x200u | (plf.needB ? 0x100u : 0u) | plf.logA;
    }
    if (rc) return rc;
    if (!decrypt) {
        rc = launch_fused_nr<false>(nr, st, tb, ek, &c, in, out, (unsigned)cus, g_lo, n8, h1, nfull, (u32)(len % 16),
                                    front, h0, sc);
        if (!rc) rc = run_ghash_levels(st, fin, nvf, plf, sc, 2, sc + GS_PART, nullptr);
    } else {
        /* the fused kernel leaves the tail alone: it is hashed as ciphertext first (in may be out) */
        rc = launch_fused_nr<true>(nr, st, tb, ek, &c, in, out, (unsigned)cus, g_lo, n8, nfull, nfull, 0, front, h0, sc);
        if (!rc) rc = run_ghash_levels(st, fin, nvf, plf, sc, 2, sc + GS_PART, nullptr);
        if (!rc && len > h1 * 16) {
            uaesk_ctr ct = c;
            ct.v0 = (c.v0 + h1) & 0x00FFFFFFFFFFFFFFull;
            rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &ct, (const unsigned char *)in + h1 * 16,
                                  (unsigned char *)out + h1 * 16, len - h1 * 16, nullptr);
        }
    }
    if (rc) return rc;
    hipLaunchKernelGGL(k_gcm_fold, dim3(1), dim3(64), 0, st, sc, (u64)((len + 15) >> 4), 0u);
    return (int)hipGetLastError();
}

extern "C" int uaesk_gcm_stream_tag(void *stream, void *scratch, int compare, void *tag_io, int *status)
{
    hipLaunchKernelGGL(k_gcm_stream_tag, dim3(1), dim3(64), 0, S(stream), (const unsigned char *)scratch,
                       compare ? 1 : 0, (unsigned char *)tag_io, status);
    return (int)hipGetLastError();
}


/* ------------------------------------------------------------------------ */
/* long GCM-SIV messages: nothing per-nonce visits the host                    */
/* ------------------------------------------------------------------------ */
/* Through round 4 a message too long for k_siv_small took derive_keys by the ECB kernel (read back: the host expanded
 * the derived key), POLYVAL (read back: the host made the tag's input), the tag by the ECB kernel (read back: the host
 * made the counter) and CTR -- three round trips, 110 us for 4 MiB.  Now the per-nonce values stay in the scratch:
 * k_siv_prep derives the keys under the master key and expands the message-encryption key there (GCM_SIV_init,
 * micro_aes.c:1421-1450; KeyExpansion :144-178) and, for a decryption, turns the received tag into the counter;
 * POLYVAL's setup reads its key from there (h_given = 2); k_siv_tag makes the tag from the raw hash (:1453-1460)
 * and writes it behind the text (or compares it) and the counter; the CTR kernel reads schedule and counter from
 * there (k_ctr_ind).  The launches follow one another on the stream; the host waits once, at the end.            */
template <int NR>
__global__ __launch_bounds__(64) void k_siv_prep(uaesk_rk mk, uaesk_tables tb, uint4 nonce,
                                                 const unsigned char *tag_in, unsigned char *__restrict__ scratch)
{
    u32 *te_plain = (u32 *)uaes_lds;                          /* 1 KiB: an unreplicated Te0 */
    u32 *drv = te_plain + 256;                                /* the derived words */
    for (u32 i = threadIdx.x; i < 256u; i += 64u) te_plain[i] = tb.te0[i];
    __syncthreads();
    constexpr u32 NK = NR - 6, NB = 2 + NK / 2;
    if (threadIdx.x < NB) {                                   /* block i = LE32(i) || nonce under the MASTER key, its low half */
        u32 s1[4] = { threadIdx.x, nonce.x, nonce.y, nonce.z };
        plain_encrypt<NR>(te_plain, mk, s1);
        drv[2 * threadIdx.x] = s1[0];
        drv[2 * threadIdx.x + 1] = s1[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint4 auth = make_uint4(drv[0], drv[1], drv[2], drv[3]);
        u32 hw[4];
        gf_to_words(gf_mul_xk(gf_from4(rev16(auth)), 1), hw);    /* POLYVAL key in GHASH form: mulX(rev(H)) */
        *(uint4 *)(scratch + GS_SIV_HG) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        auto subword = [&](u32 w) -> u32 {
            return ((te_plain[w & 0xffu] >> 8) & 0xffu) | (te_plain[(w >> 8) & 0xffu] & 0xff00u) |
                   ((te_plain[(w >> 16) & 0xffu] & 0xff00u) << 8) | ((te_plain[w >> 24] & 0xff00u) << 16);
        };
        u32 w[4 * (NR + 1)];
#pragma unroll
        for (u32 i = 0; i < NK; ++i) w[i] = drv[4 + i];
        u32 rcon = 1;
#pragma unroll
        for (u32 i = NK; i < 4u * (NR + 1); ++i) {
            u32 t = w[i - 1];
            if (i % NK == 0) {
                t = subword((t >> 8) | (t << 24)) ^ rcon;         /* RotWord on LE words */
                rcon = ((rcon << 1) ^ ((rcon >> 7) * 0x1bu)) & 0xffu;
            } else if (NK == 8 && i % NK == 4) {
                t = subword(t);
            }
            w[i] = w[i - NK] ^ t;
        }
        u32 *rk = (u32 *)(scratch + GS_SIV_RK);
#pragma unroll
        for (u32 i = 0; i < 60u; ++i) rk[i] = i < 4u * (NR + 1) ? w[i] : 0u;
    }
    if (threadIdx.x == 1 && tag_in) {                         /* decrypt: the received tag is the counter (:1500-1502) */
        u32 t[4] = { 0, 0, 0, 0 };
        for (u32 b = 0; b < 16; ++b) t[b >> 2] |= (u32)tag_in[b] << (8 * (b & 3));
        uaesk_ctr *c = (uaesk_ctr *)(scratch + GS_SIV_CTR);
        c->w0 = t[0]; c->w1 = t[1]; c->w2 = t[2]; c->w3 = t[3] | 0x80000000u;     /* c[LAST] |= 0x80 (:936) */
        c->b8 = 0; c->v0 = 0; c->le32 = 1;
    }
}

/* tag = Enc_k((POLYVAL ^ nonce) with the top bit cleared) (GCM_SIVtag :1453-1460) from the raw hash in the scratch.
 * DEC: compared with the 16 bytes at tag_io, *status = 0 / 0x1A; else written there, and the counter made of it. */
template <int NR, bool DEC>
__global__ __launch_bounds__(64) void k_siv_tag(uaesk_tables tb, uint4 nonce, unsigned char *__restrict__ scratch,
                                                unsigned char *tag_io, int *status)
{
    u32 *te_plain = (u32 *)uaes_lds;
    for (u32 i = threadIdx.x; i < 256u; i += 64u) te_plain[i] = tb.te0[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    uaesk_rk rk;
    const u32 *rkw = (const u32 *)(scratch + GS_SIV_RK);
#pragma unroll
    for (int i = 0; i < 4 * (NR + 1); ++i) rk.w[i] = rkw[i];
    const uint4 pv = rev16(*(const uint4 *)(scratch + GS_SIV_PV));
    u32 s1[4] = { pv.x ^ nonce.x, pv.y ^ nonce.y, pv.z ^ nonce.z, pv.w & 0x7fffffffu };
    plain_encrypt<NR>(te_plain, rk, s1);
    if (DEC) {
        u32 diff = 0;
        for (u32 b = 0; b < 16; ++b) diff |= (u32)tag_io[b] ^ ((s1[b >> 2] >> (8 * (b & 3))) & 0xffu);
        *status = diff ? 0x1A : 0;
    } else {
        for (u32 b = 0; b < 16; ++b) tag_io[b] = (unsigned char)(s1[b >> 2] >> (8 * (b & 3)));
        uaesk_ctr *c = (uaesk_ctr *)(scratch + GS_SIV_CTR);
        c->w0 = s1[0]; c->w1 = s1[1]; c->w2 = s1[2]; c->w3 = s1[3] | 0x80000000u;
        c->b8 = 0; c->v0 = 0; c->le32 = 1;
    }
}

template <int NR>
static int siv_long_nr(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *mk, int decrypt, uint4 nn,
                       const void *aad, size_t aad_len, const void *in, size_t len, void *out, unsigned char *sc, int *status,
                       unsigned *done_word)
{
    const unsigned char *tag_in = decrypt ? (const unsigned char *)in + len : nullptr;
    hipLaunchKernelGGL((k_siv_prep<NR>), dim3(1), dim3(64), 1024 + 64, st, *mk, *tb, nn, tag_in, sc);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    const uaesk_rk *d_rk = (const uaesk_rk *)(sc + GS_SIV_RK);
    const uaesk_ctr *d_ctr = (const uaesk_ctr *)(sc + GS_SIV_CTR);
    /* decrypt like the reference: the keystream first (the received tag is its counter), then authenticate */
    if (decrypt && (rc = uaesk_ctr_xcrypt_ind(st, tb, NR, d_rk, d_ctr, in, out, len)) != 0) return rc;
    GSrc msg;                                                    /* POLYVAL over AAD || plaintext || lengths */
    msg.aad = (const unsigned char *)aad; msg.aad_len = aad_len;
    msg.ct = (const unsigned char *)(decrypt ? out : in); msg.ct_len = len;
    msg.has_len = 1; msg.len_aad = aad_len; msg.len_ct = len; msg.rev = 1;
    const u64 nv = ((aad_len + 15) >> 4) + ((len + 15) >> 4) + 1;
    const uaes_plan sp = siv_plan(len, aad_len, done_word != nullptr, true);
    const u32 steps = sp.steps;
    if (sp.arrangement == UAES_ARR_SIV_CHUNKS) {
        /* as far as one round of chunk workgroups reaches (128 MiB on 256 CUs) POLYVAL and the tag are ONE launch:
         * the hash-only chunk workgroups (byte-reversed blocks, the key from the scratch) and the finisher, which
         * also makes the tag and the counter (gcm_combine_body, modes 3 / 4) */
        const u32 W = sp.grid;
        GSrc sm = msg;
        sm.has_len = 0;
        GmcFin fin;
        memset(&fin, 0, sizeof fin);
        fin.j0 = nn; fin.done_word = done_word; fin.look_ticks = gcm_look_ticks(); fin.mode = decrypt ? 4 : 3; fin.ylog = 10u + log2_u32(steps);
        fin.tag_io = decrypt ? (unsigned char *)in + len : (unsigned char *)out + len;
        fin.status = status; fin.len_aad = aad_len; fin.len_ct = len;
        uaesk_rk dummy;
        uaesk_ctr cdummy;
        memset(&dummy, 0, sizeof dummy);
        memset(&cdummy, 0, sizeof cdummy);
        hipError_t e = uaesk_want_lds((const void *)k_gcm_chunks<NR, 1, true>, (unsigned)(GSM_LDS_TOTAL));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_gcm_chunks<NR, 1, true>), dim3(W + 1u), dim3(GH_T), GSM_LDS_TOTAL, st, dummy, *tb, cdummy, sm,
                           (const uint4 *)msg.ct, (uint4 *)nullptr, (const unsigned char *)sc, (uint4 *)(sc + GS_ACC1), 1u, steps, fin);
        if ((rc = (int)hipGetLastError()) != 0) return rc;
        if (!decrypt) rc = uaesk_ctr_xcrypt_ind(st, tb, NR, d_rk, d_ctr, in, out, len);
        return rc;
    }
    const GPlan pl = plan_for(nv);
    uaesk_rk dummy_rk;
    memset(&dummy_rk, 0, sizeof dummy_rk);
    if ((rc = launch_setup<10>(st, tb, &dummy_rk, make_uint4(0, 0, 0, 0), sc, pl, 2, make_uint4(0, 0, 0, 0))) != 0) return rc;
    if ((rc = run_ghash_levels(st, msg, nv, pl, sc, 2, sc + GS_SIV_PV, nullptr)) != 0) return rc;
    if (decrypt) hipLaunchKernelGGL((k_siv_tag<NR, true>), dim3(1), dim3(64), 1024, st, *tb, nn, sc, (unsigned char *)in + len, status);
    else         hipLaunchKernelGGL((k_siv_tag<NR, false>), dim3(1), dim3(64), 1024, st, *tb, nn, sc, (unsigned char *)out + len, nullptr);
    if ((rc = (int)hipGetLastError()) != 0) return rc;
    if (!decrypt) rc = uaesk_ctr_xcrypt_ind(st, tb, NR, d_rk, d_ctr, in, out, len);
    return rc;
}

/* GCM-SIV of a message of any length, six or seven launches and no host round trip (see above).  mk = the schedule
 * of the MASTER key; encrypt: the tag is written at out + len; decrypt: read at in + len, the plaintext is written
 * either way (the reference's order) and *status (device or host-visible) = 0 / 0x1A.  in == out is allowed.
 * With a counter word armed (uaesk_done_word_arm) and up to 128 MiB: three launches (prep, POLYVAL + tag, CTR).  */
extern "C" int uaesk_gcmsiv_long(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *mk, int decrypt,
                                 const uint8_t *nonce12, const void *aad, size_t aad_len,
                                 const void *in, size_t len, void *out, void *scratch, int *status)
{
    uint4 nn = make_uint4(0, 0, 0, 0);
    memcpy(&nn, nonce12, 12);
    unsigned char *sc = (unsigned char *)scratch;
    unsigned *done_word = uaesk_done_word_take();               /* armed by the host layer: a word that is zero between calls */
    switch (nr) {
    case 10: return siv_long_nr<10>(S(stream), tb, mk, decrypt, nn, aad, aad_len, in, len, out, sc, status, done_word);
    case 12: return siv_long_nr<12>(S(stream), tb, mk, decrypt, nn, aad, aad_len, in, len, out, sc, status, done_word);
    case 14: return siv_long_nr<14>(S(stream), tb, mk, decrypt, nn, aad, aad_len, in, len, out, sc, status, done_word);
    default: return (int)hipErrorInvalidValue;
    }
}
