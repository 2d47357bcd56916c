/*
 * uaes_chain.hip -- the feedback modes: CBC (with CS3
 * ciphertext stealing), CFB and OFB.
 *
 *   k_cbc_dec / k_cfb_dec   block-PARALLEL directions:
 *                             CBC  P_i = Dec(C_i) ^ C_{i-1}   (micro_aes.c:746-782)
 *                             CFB  P_i = Enc(C_{i-1}) ^ C_i   (micro_aes.c:799-817, mode 0)
 *                           same launch shape as k_ecb (persistent 1024-thread
 *                           workgroups, 4 blocks per lane, 1 KiB wave accesses).
 *   k_chain_serial          the inherently serial directions -- CBC encrypt
 *                           (:697-744), CFB encrypt (:799-817, mode 1), OFB
 *                           (:861-893) -- and the CTS pair of CBC decrypt: one
 *                           lane walks the chain (north_star: "CBC/CFB/OFB stay
 *                           single-GPU because the chain is serial").
 *
 * The parallel kernels read C_{i-1} from the INPUT buffer, so they must not run
 * in place; the host layer gives them a private copy of the input when
 * in == out (the reference itself works in place, serially).
 */
#include <hip/hip_runtime.h>
#include <type_traits>
#include <string.h>
#include <stdlib.h>
#include "uaes_aes.hip.h"
#include "uaes_device.h"

#define UAES_U 4
static inline hipStream_t S(void *s) { return (hipStream_t)s; }

/* ------------------------------------------------------------------------ */
/* serial directions: one wave, a quad of lanes per block encryption          */
/* ------------------------------------------------------------------------ */
struct Blk {
    u32 w[4];
};

/* byte i of a block held in four registers, without indexing the registers by a run-time value (that
 * would put the block in scratch memory and give the whole kernel a private segment)             */
__device__ __forceinline__ u32 blk_byte(const u32 w0, const u32 w1, const u32 w2, const u32 w3, u32 i)
{
    const u32 w = i < 8 ? (i < 4 ? w0 : w1) : (i < 12 ? w2 : w3);
    return (w >> (8 * (i & 3))) & 0xffu;
}

__device__ __forceinline__ void blk_or_byte(Blk &b, u32 i, u32 v)
{
    const u32 x = v << (8 * (i & 3)), q = i >> 2;
    b.w[0] |= q == 0 ? x : 0;
    b.w[1] |= q == 1 ? x : 0;
    b.w[2] |= q == 2 ? x : 0;
    b.w[3] |= q == 3 ? x : 0;
}

__device__ __forceinline__ Blk ldb(const unsigned char *p, u32 nbytes)        /* zero padded */
{
    Blk b = { { 0, 0, 0, 0 } };
    if (nbytes >= 16 && (((uintptr_t)p) & 15u) == 0) {
        const uint4 v = *(const uint4 *)p;
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        return b;
    }
    for (u32 i = 0; i < (nbytes < 16 ? nbytes : 16u); ++i) blk_or_byte(b, i, (u32)p[i]);
    return b;
}

__device__ __forceinline__ void stb(unsigned char *p, const Blk &b, u32 nbytes)
{
    if (threadIdx.x != 0) return;                  /* every lane holds the same block: one stores */
    if (nbytes >= 16 && (((uintptr_t)p) & 15u) == 0) {
        *(uint4 *)p = make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]);
        return;
    }
    for (u32 i = 0; i < (nbytes < 16 ? nbytes : 16u); ++i) p[i] = (unsigned char)blk_byte(b.w[0], b.w[1], b.w[2], b.w[3], i);
}

__device__ __forceinline__ void xb(Blk &a, const Blk &b)
{
    a.w[0] ^= b.w[0]; a.w[1] ^= b.w[1]; a.w[2] ^= b.w[2]; a.w[3] ^= b.w[3];
}

/* one block, shared by the four lanes of a quad (quad_encrypt, uaes_aes.hip.h); every lane of
 * the wave holds the same block before and after                                          */
template <int NR>
__device__ __forceinline__ void encb(Blk &b, const uaesk_rk &rk, const LaneConst &lc)
{
    quad_encrypt<NR>(b.w, rk, lc);
}

template <int NR>
__device__ __forceinline__ void decb(Blk &b, const uaesk_rk &dk, const LaneConst &lc)
{
    u32 s[1][4] = { { b.w[0], b.w[1], b.w[2], b.w[3] } };
    dec_blocks<NR, 1>(s, dk, lc);
    b.w[0] = s[0][0]; b.w[1] = s[0][1]; b.w[2] = s[0][2]; b.w[3] = s[0][3];
}

/* first r bytes of a, the rest of b */
__device__ __forceinline__ Blk splice(const Blk &a, const Blk &b, u32 r)
{
    Blk o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        u32 v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 i = 4 * w + k;
            v |= ((((i < r) ? a.w[w] : b.w[w]) >> (8 * k)) & 0xffu) << (8 * k);
        }
        o.w[w] = v;
    }
    return o;
}

/* the last two blocks {X full, Z r bytes} of AES_CBC_decrypt (:770-778); rk = decryption keys, iv = the block
 * before X; one lane on the full inverse tables                                                              */
template <int NR>
__device__ __forceinline__ void cbc_dec_cts_pair(const uaesk_rk &rk, const LaneConst &lc, const Blk &iv,
                                                 const unsigned char *in, unsigned char *out, u32 r)
{
    const Blk x = ldb(in, 16), z = ldb(in + 16, r);
    Blk y = x;
    decb<NR>(y, rk, lc);                              /* Y = Dec(X) */
    Blk p2 = y;
    xb(p2, z);                                        /* P2 = Y ^ Z (first r bytes) */
    Blk c = splice(z, y, r);                          /* Z | tail of Y */
    decb<NR>(c, rk, lc);
    xb(c, iv);
    stb(out, c, 16);
    stb(out + 16, p2, r);
}

/* ------------------------------------------------------------------------ */
/* parallel decrypt directions                                                */
/* ------------------------------------------------------------------------ */
/* CFB=false: CBC decrypt of blocks [0, n): out_i = Dec(in_i) ^ prev_i
 * CFB=true : CFB decrypt of blocks [0, n): out_i = Enc(prev_i) ^ in_i, plus
 *            `rem` tail bytes out = Enc(prev_n)[0..rem) ^ in   (mixThenXor, :816)
 * prev_0 = iv, prev_i = in_{i-1}.  CBC with cts_r != 0: blocks n and n+1 are the stolen pair.   */
template <int NR, bool CFB, int U>                 /* U blocks per lane: 4 for bulk texts, 1 for short ones */
__global__ __launch_bounds__(UAES_WG) void k_fb_dec(uaesk_rk rk, uaesk_tables tb, uint4 iv,
                                                    const uint4 *__restrict__ in, uint4 *__restrict__ out,
                                                    u64 n, u32 rem, u32 cts_r)
{
    if (CFB) fill_enc_tables(tb.te0); else fill_dec_tables(tb.td0);
    const LaneConst lc = make_lane_const();
    const u64 stride = (u64)gridDim.x * UAES_WG * U;
    for (u64 base = (u64)blockIdx.x * UAES_WG * U; base < n; base += stride) {
        u32 s[U][4];
        uint4 x[U], prev[U];
        u64 idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = base + (u64)u * UAES_WG + threadIdx.x;
            const u64 i = idx[u] < n ? idx[u] : n - 1;
            x[u] = in[i];
            prev[u] = i ? in[i - 1] : iv;
            const uint4 src = CFB ? prev[u] : x[u];
            s[u][0] = src.x; s[u][1] = src.y; s[u][2] = src.z; s[u][3] = src.w;
        }
        if (!CFB) {
            dec_blocks<NR, U>(s, rk, lc);
        } else if (U == 4) {                       /* two pairs, each half a round out of phase */
            enc_blocks_skewed<NR>(s[0], s[1], rk, lc);
            enc_blocks_skewed<NR>(s[2 % U], s[3 % U], rk, lc);
        } else {
            enc_blocks<NR, U>(s, rk, lc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 m = CFB ? x[u] : prev[u];
            if (idx[u] < n) out[idx[u]] = make_uint4(s[u][0] ^ m.x, s[u][1] ^ m.y, s[u][2] ^ m.z, s[u][3] ^ m.w);
        }
    }
    /* CBC: the last two blocks {X full, Z cts_r bytes} behind the n parallel ones (AES_CBC_decrypt :770-778), one
     * lane of the same launch; the block before X is C_{n-1} or the IV                                       */
    if (!CFB && cts_r && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint4 pv = n ? in[n - 1] : iv;
        const Blk ivb = { { pv.x, pv.y, pv.z, pv.w } };
        cbc_dec_cts_pair<NR>(rk, lc, ivb, (const unsigned char *)(in + n), (unsigned char *)(out + n), cts_r);
    }
    if (CFB && rem && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint4 p = n ? in[n - 1] : iv;
        u32 s1[1][4] = { { p.x, p.y, p.z, p.w } };
        enc_blocks<NR, 1>(s1, rk, lc);
        const unsigned char *src = (const unsigned char *)(in + n);
        unsigned char *dst = (unsigned char *)(out + n);
        for (u32 i = 0; i < rem; ++i) dst[i] = src[i] ^ (unsigned char)blk_byte(s1[0][0], s1[0][1], s1[0][2], s1[0][3], i);
    }
}

enum { CH_CBC_ENC = 0, CH_CFB_ENC = 1, CH_OFB = 2, CH_CBC_DEC_CTS = 3, CH_CBC_ENC_PAD = 4 };

/* OP = CH_CBC_ENC     whole message, CS3 ciphertext stealing (AES_CBC_encrypt :697-744); len >= 16
 *      CH_CFB_ENC     whole message (:799-817 mode 1)
 *      CH_OFB         whole message (:861-886)
 *      CH_CBC_DEC_CTS the last two blocks {X full, Z r bytes} of AES_CBC_decrypt (:770-778):
 *                     in/out point at X; iv = the block before X (or the IV)
 *      CH_CBC_ENC_PAD whole message of a build with CTS 0 (:704-733): no stealing, the last chunk padded by
 *                     padBlock (:610-621) with aux = AES_PADDING (0: zeros behind a partial chunk only; 1 PKCS#7 and
 *                     2 ISO 7816-4 always append); any len, writes 16 * (len / 16 + (len % 16 || aux)) bytes       */
template <int NR, int OP>
__global__ __launch_bounds__(UAES_WG) void k_chain_serial(uaesk_rk rk, uaesk_tables tb, uint4 iv4,
                                                          const uint4 *__restrict__ iv_dev,   /* overrides iv4 if set */
                                                          const unsigned char *in,
                                                          unsigned char *out, u64 len, u32 aux)
{
    /* encrypt-direction chains: ONE wave; the sixteen lanes of a DPP row share each block encryption
     * (row_encrypt, uaes_aes.hip.h: one state byte and one lookup per lane and round), the four rows run
     * redundantly, the chain value stays a column word per lane and lanes 0/4/8/12 store.  The two-block CTS
     * decrypt keeps the one-lane inverse cipher on the full tables (launched with 1024 threads to fill them). */
    if (OP == CH_CBC_DEC_CTS) {
        fill_dec_tables(tb.td0);
        if (threadIdx.x != 0) return;
        const LaneConst lc = make_lane_const();
        if (iv_dev) iv4 = *iv_dev;
        const Blk iv = { { iv4.x, iv4.y, iv4.z, iv4.w } };
        cbc_dec_cts_pair<NR>(rk, lc, iv, in, out, (u32)len);      /* rk = decryption keys */
        return;
    }
    row_fill_tables(tb.te0, rk);
    const RowLane<NR> L = row_lane<NR>();
    if (iv_dev) iv4 = *iv_dev;
    const u32 ivw[4] = { iv4.x, iv4.y, iv4.z, iv4.w };
    u32 m = row_pick(ivw, L.c);                               /* the chain value: this lane's column */

    const bool a4 = ((((uintptr_t)in) | ((uintptr_t)out)) & 3u) == 0;
    auto body = [&](auto A4T) {
        constexpr bool A4 = decltype(A4T)::value;
        if (OP == CH_CBC_ENC) {
            u64 n = len / 16;
            u32 r = (u32)(len % 16);
            if (n > 1 && !r) { --n; r = 16; }                 /* CS3: always swap the last two (:706) */
            /* the last (short or swapped) chunk, zero padded; read before anything is written (in place) */
            const u32 l0 = r ? row_load(in + 16 * n, r, L.c) : 0u;
            row_walk<A4>(in, n, L.c, [&](u64 i, u32 x) {
                m = row_encrypt<NR>(m ^ x, L);
                row_store_full<A4>(out + 16 * i, m, L.c);
            });
            if (r) {                                          /* m == C_{n-1} here */
                row_store(out + 16 * n, m, r);                /* its place takes the head of C_{n-1} ("stolen") */
                m = row_encrypt<NR>(m ^ l0, L);
                row_store(out + 16 * (n - 1), m, 16);
            }
        } else if (OP == CH_CBC_ENC_PAD) {
            const u64 n = len / 16;
            const u32 r = (u32)(len % 16);
            /* the padded last chunk as this lane's column word: text bytes below r, then the padding */
            u32 l0 = r ? row_load(in + 16 * n, r, L.c) : 0u;
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                const u32 i = 4u * L.c + k;
                const u32 pb = aux == 1 ? 16u - r : (aux == 2 && i == r ? 0x80u : 0u);
                if (i >= r) l0 |= pb << (8u * k);
            }
            row_walk<A4>(in, n, L.c, [&](u64 i, u32 x) {
                m = row_encrypt<NR>(m ^ x, L);
                row_store_full<A4>(out + 16 * i, m, L.c);
            });
            if (r || aux) row_store(out + 16 * n, row_encrypt<NR>(m ^ l0, L), 16);
        } else if (OP == CH_CFB_ENC) {
            const u64 n = len / 16;
            const u32 r = (u32)(len % 16);
            const u32 l0 = r ? row_load(in + 16 * n, r, L.c) : 0u;
            row_walk<A4>(in, n, L.c, [&](u64 i, u32 x) {
                m = row_encrypt<NR>(m, L) ^ x;                /* C_i = Enc(C_{i-1}) ^ P_i = next feedback */
                row_store_full<A4>(out + 16 * i, m, L.c);
            });
            if (r) row_store(out + 16 * n, row_encrypt<NR>(m, L) ^ l0, r);
        } else {                                              /* CH_OFB */
            const u64 n = len / 16;
            const u32 r = (u32)(len % 16);
            const u32 l0 = r ? row_load(in + 16 * n, r, L.c) : 0u;
            row_walk<A4>(in, n, L.c, [&](u64 i, u32 x) {
                m = row_encrypt<NR>(m, L);                    /* O_i = Enc(O_{i-1}) */
                row_store_full<A4>(out + 16 * i, m ^ x, L.c);
            });
            if (r) row_store(out + 16 * n, row_encrypt<NR>(m, L) ^ l0, r);
        }
    };
    if (a4) body(std::true_type{}); else body(std::false_type{});
}

/* ------------------------------------------------------------------------ */
/* batches of independent chains: one lane per message                         */
/* ------------------------------------------------------------------------ */
/* A chain is serial, a batch of chains is not: nmsg messages of msg_blocks whole blocks each
 * (message m at byte m * msg_blocks * 16, its IV at ivs + 16 m), every lane walks one of them
 * through the full tables.  MAC = false: AES_CBC_encrypt of every message (:697-744; a
 * message of two or more blocks ends in the CS3 swap of its last two ciphertext blocks, as a
 * single call on it does).  MAC = true: AES_CMAC of every message (:1108-1118), msg_bytes
 * need not be a multiple of 16; macs receive 16 bytes each.                               */
template <int NR, bool MAC>
__global__ __launch_bounds__(UAES_WG) void k_chain_batch(uaesk_rk rk, uaesk_tables tb, const uint4 *__restrict__ ivs,
                                                         u64 nmsg, u64 msg_bytes,
                                                         const unsigned char *in, unsigned char *out)
{
    fill_enc_tables(tb.te0);
    const LaneConst lc = make_lane_const();
    const u64 nb = msg_bytes / 16;
    Blk k1 = { { 0, 0, 0, 0 } }, k2 = k1;
    if (MAC) {                                               /* subkeys K1 = 2L, K2 = 4L (getSubkeys :593-605) */
        u32 s[1][4] = { { 0, 0, 0, 0 } };
        enc_blocks<NR, 1>(s, rk, lc);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u64 hi = ((u64)bswap32(s[0][0]) << 32) | bswap32(s[0][1]), lo = ((u64)bswap32(s[0][2]) << 32) | bswap32(s[0][3]);
            const u64 carry = hi >> 63;
            hi = (hi << 1) | (lo >> 63);
            lo = (lo << 1) ^ (carry ? 0x87ull : 0ull);
            s[0][0] = bswap32((u32)(hi >> 32)); s[0][1] = bswap32((u32)hi);
            s[0][2] = bswap32((u32)(lo >> 32)); s[0][3] = bswap32((u32)lo);
            Blk &k = q ? k2 : k1;
            k.w[0] = s[0][0]; k.w[1] = s[0][1]; k.w[2] = s[0][2]; k.w[3] = s[0][3];
        }
    }
    for (u64 m = (u64)blockIdx.x * UAES_WG + threadIdx.x; m < nmsg; m += (u64)gridDim.x * UAES_WG) {
        const unsigned char *src = in + m * msg_bytes;
        u32 s[1][4] = { { 0, 0, 0, 0 } };
        if (!MAC) {
            const uint4 iv = ivs[m];
            s[0][0] = iv.x; s[0][1] = iv.y; s[0][2] = iv.z; s[0][3] = iv.w;
            uint4 *dst = (uint4 *)(out + m * msg_bytes);
            uint4 prev = iv;
            /* whole groups of four blocks in front of the last two: the next group's text is requested before this
             * group's four dependent encryptions start (a lane's loads are 16 bytes from a line of its own: every block
             * used to wait for memory), and the four results leave together -- 64 contiguous bytes per lane; written
             * one block per trip, a line was evicted half-written with 2^18 lanes at work (2^18 messages of 1 KiB: 524 ->
             * 1122 GiB/s) */
            u64 i0 = 0;
            if (nb >= 6) {
                const u64 ngrp = (nb - 2) / 4;
                uint4 cur[4], nxt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = ((const uint4 *)src)[j];
                for (u64 g = 0; g < ngrp; ++g) {
                    const u64 gn = g + 1 < ngrp ? g + 1 : g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = ((const uint4 *)src)[4 * gn + j];
                    uint4 c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s[0][0] ^= cur[j].x; s[0][1] ^= cur[j].y; s[0][2] ^= cur[j].z; s[0][3] ^= cur[j].w;
                        enc_blocks<NR, 1>(s, rk, lc);
                        c[j] = make_uint4(s[0][0], s[0][1], s[0][2], s[0][3]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dst[4 * g + j] = c[j]; cur[j] = nxt[j]; }
                    prev = c[3];
                }
                i0 = 4 * ngrp;
            }
            for (u64 i = i0; i < nb; ++i) {
                const uint4 x = ((const uint4 *)src)[i];
                s[0][0] ^= x.x; s[0][1] ^= x.y; s[0][2] ^= x.z; s[0][3] ^= x.w;
                enc_blocks<NR, 1>(s, rk, lc);
                const uint4 c = make_uint4(s[0][0], s[0][1], s[0][2], s[0][3]);
                if (nb > 1 && i == nb - 1) { dst[i - 1] = c; dst[i] = prev; }      /* CS3: the last two swap (:706, :738-742) */
                else if (nb == 1 || i < nb - 2) dst[i] = c;
                prev = c;
            }
        } else {
            const u32 last = msg_bytes ? (u32)((msg_bytes - 1) % 16) + 1 : 0;     /* size of the last block */
            const u64 full = (msg_bytes - last) / 16;
            u64 i0 = 0;
            if (full >= 4 && (((uintptr_t)src) & 15u) == 0) {          /* groups of four, the next one requested ahead */
                const u64 ngrp = full / 4;
                uint4 cur[4], nxt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = ((const uint4 *)src)[j];
                for (u64 g = 0; g < ngrp; ++g) {
                    const u64 gn = g + 1 < ngrp ? g + 1 : g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = ((const uint4 *)src)[4 * gn + j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s[0][0] ^= cur[j].x; s[0][1] ^= cur[j].y; s[0][2] ^= cur[j].z; s[0][3] ^= cur[j].w;
                        enc_blocks<NR, 1>(s, rk, lc);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
                }
                i0 = 4 * ngrp;
            }
            for (u64 i = i0; i < full; ++i) {
                const Blk x = ldb(src + 16 * i, 16);
                s[0][0] ^= x.w[0]; s[0][1] ^= x.w[1]; s[0][2] ^= x.w[2]; s[0][3] ^= x.w[3];
                enc_blocks<NR, 1>(s, rk, lc);
            }
            Blk l = ldb(src + 16 * full, last);
            if (last < 16) { blk_or_byte(l, last, 0x80u); xb(l, k2); } else xb(l, k1);   /* the byte there is padding: 0 */
            s[0][0] ^= l.w[0]; s[0][1] ^= l.w[1]; s[0][2] ^= l.w[2]; s[0][3] ^= l.w[3];
            enc_blocks<NR, 1>(s, rk, lc);
            ((uint4 *)out)[m] = make_uint4(s[0][0], s[0][1], s[0][2], s[0][3]);
        }
    }
}

/* The same batch with SIXTEEN LANES PER MESSAGE (round 5): a lane per message fills the GPU only from 2^18 messages
 * on -- 4096 messages of 64 KiB ran on 64 waves, 23 GiB/s, every chain at the 5.8 MiB/s a lane makes of it.  Here
 * the four DPP rows of a wave walk four messages (row_encrypt: one state byte and one lookup per lane and round;
 * row4_fill_tables gives every row banks of its own), a chain runs at the pace of the serial-chain kernel and
 * 16384 messages already occupy every SIMD four waves deep.  Same results byte for byte; launch_batch picks the
 * arrangement by the number of messages.  Messages and IVs as k_chain_batch; A4: every message 4-byte aligned.  */
template <int NR, bool MAC, bool A4>
__global__ __launch_bounds__(UAES_WG) void k_chain_batch_row(uaesk_rk rk, uaesk_tables tb, const uint4 *__restrict__ ivs,
                                                             u64 nmsg, u64 msg_bytes,
                                                             const unsigned char *in, unsigned char *out)
{
    row4_fill_tables(tb.te0, rk);
    const RowLane<NR> L = row4_lane<NR>();
    u32 k1c = 0, k2c = 0;                                     /* this lane's column of the subkeys */
    if (MAC) {                                                /* K1 = 2L, K2 = 4L, L = Enc(0) (getSubkeys :593-605) */
        u32 b[4];
        row_spread(row_encrypt<NR>(0u, L), b);                /* (row 0's copy: all rows hold the same) */
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u64 hi = ((u64)bswap32(b[0]) << 32) | bswap32(b[1]), lo = ((u64)bswap32(b[2]) << 32) | bswap32(b[3]);
            const u64 carry = hi >> 63;
            hi = (hi << 1) | (lo >> 63);
            lo = (lo << 1) ^ (carry ? 0x87ull : 0ull);
            b[0] = bswap32((u32)(hi >> 32)); b[1] = bswap32((u32)hi);
            b[2] = bswap32((u32)(lo >> 32)); b[3] = bswap32((u32)lo);
            if (q) k2c = row_pick(b, L.c); else k1c = row_pick(b, L.c);
        }
    }
    const u64 rows = blockDim.x >> 4;
    for (u64 m = (u64)blockIdx.x * rows + (threadIdx.x >> 4); m < nmsg; m += (u64)gridDim.x * rows) {
        const unsigned char *src = in + m * msg_bytes;
        if (!MAC) {
            unsigned char *dst = out + m * msg_bytes;
            u32 v = ((const u32 *)(ivs + m))[L.c];            /* the chain value: this lane's column */
            u64 n = msg_bytes / 16;
            const bool swap = n > 1;                          /* CS3: the last two blocks swap (:706, :738-742) */
            if (swap) --n;
            const u32 l0 = swap ? row_load_full<A4>(src + 16 * n, L.c) : 0u;   /* read before anything is written */
            row_walk<A4>(src, n, L.c, [&](u64 i, u32 x) {
                v = row_encrypt<NR>(v ^ x, L);
                row_store_full<A4>(dst + 16 * i, v, L.c);
            });
            if (swap) {                                       /* v == C_{n-1} here */
                row_store_full<A4>(dst + 16 * n, v, L.c);
                v = row_encrypt<NR>(v ^ l0, L);
                row_store_full<A4>(dst + 16 * (n - 1), v, L.c);
            }
        } else {
            const u32 last = msg_bytes ? (u32)((msg_bytes - 1) % 16) + 1 : 0;     /* size of the last block */
            const u64 full = (msg_bytes - last) / 16;
            u32 v = 0;
            row_walk<A4>(src, full, L.c, [&](u64, u32 x) { v = row_encrypt<NR>(v ^ x, L); });
            u32 l = last ? row_load(src + 16 * full, last, L.c) : 0u;
            if (last < 16) {                                  /* 10* padding, then K2 */
                if ((last >> 2) == L.c) l |= 0x80u << (8u * (last & 3u));
                l ^= k2c;
            } else {
                l ^= k1c;
            }
            v = row_encrypt<NR>(v ^ l, L);
            ((u32 *)out)[4 * m + L.c] = v;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* launchers                                                                  */
/* ------------------------------------------------------------------------ */
#define DISPATCH_NR(nr, CALL)                         \
    switch (nr) {                                     \
    case 10: { constexpr int NR = 10; CALL; } break;  \
    case 12: { constexpr int NR = 12; CALL; } break;  \
    case 14: { constexpr int NR = 14; CALL; } break;  \
    default: return (int)hipErrorInvalidValue;        \
    }

static unsigned cu_count()
{
    static int cus = 0;
    if (!cus) uaesk_device_info(&cus, nullptr);
    return cus > 0 ? (unsigned)cus : 256u;
}

template <int NR, bool CFB, int U>
static int launch_fb_dec_u(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k, uint4 iv,
                           const void *in, void *out, u64 n, u32 rem, u32 cts_r)
{
    const unsigned lds = CFB ? UAES_LDS_ENC : UAES_LDS_DEC;
    hipError_t e = uaesk_want_lds((const void *)k_fb_dec<NR, CFB, U>, (unsigned)(lds));
    if (e != hipSuccess) return (int)e;
    u64 want = (n + (u64)UAES_WG * U - 1) / ((u64)UAES_WG * U);
    if (!want) want = 1;
    const unsigned grid = (unsigned)(want < cu_count() ? want : cu_count());
    hipLaunchKernelGGL((k_fb_dec<NR, CFB, U>), dim3(grid), dim3(UAES_WG), lds, st, *k, *tb, iv,
                       (const uint4 *)in, (uint4 *)out, n, rem, cts_r);
    return (int)hipGetLastError();
}

/* short texts take one block per lane so that up to four times as many CUs take part */
template <int NR, bool CFB>
static int launch_fb_dec(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k, uint4 iv,
                         const void *in, void *out, u64 n, u32 rem, u32 cts_r = 0)
{
    const u64 wgs4 = (n + (u64)UAES_WG * UAES_U - 1) / ((u64)UAES_WG * UAES_U);
    if (wgs4 * 2 <= cu_count()) return launch_fb_dec_u<NR, CFB, 1>(st, tb, k, iv, in, out, n, rem, cts_r);
    return launch_fb_dec_u<NR, CFB, UAES_U>(st, tb, k, iv, in, out, n, rem, cts_r);
}

template <int NR, int OP>
static int launch_serial(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k, uint4 iv,
                         const void *in, void *out, u64 len, const uint4 *iv_dev = nullptr, u32 aux = 0)
{
    const unsigned lds = OP == CH_CBC_DEC_CTS ? UAES_LDS_DEC : UAES_LDS_ROW;
    const unsigned threads = OP == CH_CBC_DEC_CTS ? UAES_WG : 64u;
    hipError_t e = uaesk_want_lds((const void *)k_chain_serial<NR, OP>, (unsigned)(lds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_chain_serial<NR, OP>), dim3(1), dim3(threads), lds, st, *k, *tb, iv, iv_dev,
                       (const unsigned char *)in, (unsigned char *)out, len, aux);
    return (int)hipGetLastError();
}

/* a lane per message from 81 920 messages on (where the two arrangements cross), sixteen lanes below:
 * tools/batch_rate.py, profiles/r05_batch_rate.log; UAES_BATCH_ROW_MAX moves the switch */
template <int NR, bool MAC>
static int launch_batch(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *k, const void *ivs,
                        u64 nmsg, u64 msg_bytes, const void *in, void *out)
{
    const int row_max = 81919;
    if (nmsg <= (u64)row_max) {
        const bool a4 = ((((uintptr_t)in) | ((uintptr_t)out)) & 3u) == 0 && msg_bytes % 4 == 0;
        /* 64 messages per 16-wave workgroup; few messages: 4-wave workgroups, so that they spread over the CUs */
        const unsigned wg = (nmsg + 63) / 64 * 2 <= cu_count() ? 256u : UAES_WG;
        const u64 want = (nmsg + wg / 16 - 1) / (wg / 16);
        const unsigned grid = (unsigned)(want < cu_count() ? want : cu_count());
        hipError_t e;
        if (a4) {
            if ((e = uaesk_want_lds((const void *)k_chain_batch_row<NR, MAC, true>, (unsigned)(UAES_LDS_ROW4))) != hipSuccess) return (int)e;
            hipLaunchKernelGGL((k_chain_batch_row<NR, MAC, true>), dim3(grid), dim3(wg), UAES_LDS_ROW4, st, *k, *tb,
                               (const uint4 *)ivs, nmsg, msg_bytes, (const unsigned char *)in, (unsigned char *)out);
        } else {
            if ((e = uaesk_want_lds((const void *)k_chain_batch_row<NR, MAC, false>, (unsigned)(UAES_LDS_ROW4))) != hipSuccess) return (int)e;
            hipLaunchKernelGGL((k_chain_batch_row<NR, MAC, false>), dim3(grid), dim3(wg), UAES_LDS_ROW4, st, *k, *tb,
                               (const uint4 *)ivs, nmsg, msg_bytes, (const unsigned char *)in, (unsigned char *)out);
        }
        return (int)hipGetLastError();
    }
    hipError_t e = uaesk_want_lds((const void *)k_chain_batch<NR, MAC>, (unsigned)(UAES_LDS_ENC));
    if (e != hipSuccess) return (int)e;
    u64 want = (nmsg + UAES_WG - 1) / UAES_WG;
    const unsigned grid = (unsigned)(want < cu_count() ? (want ? want : 1) : cu_count());
    hipLaunchKernelGGL((k_chain_batch<NR, MAC>), dim3(grid), dim3(UAES_WG), UAES_LDS_ENC, st, *k, *tb,
                       (const uint4 *)ivs, nmsg, msg_bytes, (const unsigned char *)in, (unsigned char *)out);
    return (int)hipGetLastError();
}

/* mac == 0: CBC encrypt of nmsg messages of msg_bytes (a multiple of 16, >= 16) each, IVs at ivs (device);
 * mac != 0: CMAC of nmsg messages of msg_bytes each into out (16 bytes per message)          */
extern "C" int uaesk_chain_batch(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int mac,
                                 const void *ivs, size_t nmsg, size_t msg_bytes, const void *in, void *out)
{
    if (nmsg == 0) return 0;
    if (!mac && (msg_bytes < 16 || msg_bytes % 16)) return (int)hipErrorInvalidValue;
    if (mac) { DISPATCH_NR(nr, return (launch_batch<NR, true>(S(stream), tb, ek, nullptr, nmsg, msg_bytes, in, out))); }
    else     { DISPATCH_NR(nr, return (launch_batch<NR, false>(S(stream), tb, ek, ivs, nmsg, msg_bytes, in, out))); }
    return 0;
}

/* mode: 0 CBC encrypt, 1 CBC decrypt, 2 CFB encrypt, 3 CFB decrypt, 4 OFB;
 *       5 + p: CBC encrypt of a build with CTS 0 and AES_PADDING p (0..2), 8: its CBC decrypt (whole blocks).
 * ek / dk: encryption / equivalent-inverse keys; iv16: host pointer.
 * Parallel directions (1, 3, 8) require in != out.  CBC with stealing (0, 1) needs len >= 16.        */
extern "C" int uaesk_feedback(void *stream, const uaesk_tables *tb, int nr,
                              const uaesk_rk *ek, const uaesk_rk *dk, int mode, const uint8_t *iv16,
                              const void *in, size_t len, void *out)
{
    hipStream_t st = S(stream);
    uint4 iv;
    memcpy(&iv, iv16, 16);
    if (len == 0 && mode != 6 && mode != 7) return 0;
    switch (mode) {
    case 5: case 6: case 7:
        DISPATCH_NR(nr, return (launch_serial<NR, CH_CBC_ENC_PAD>(st, tb, ek, iv, in, out, len, nullptr, (u32)(mode - 5)))); break;
    case 8:
        if (len % 16) return (int)hipErrorInvalidValue;
        DISPATCH_NR(nr, return (launch_fb_dec<NR, false>(st, tb, dk, iv, in, out, len / 16, 0, 0))); break;
    case 0: DISPATCH_NR(nr, return (launch_serial<NR, CH_CBC_ENC>(st, tb, ek, iv, in, out, len))); break;
    case 2: DISPATCH_NR(nr, return (launch_serial<NR, CH_CFB_ENC>(st, tb, ek, iv, in, out, len))); break;
    case 4: DISPATCH_NR(nr, return (launch_serial<NR, CH_OFB>(st, tb, ek, iv, in, out, len))); break;
    case 3: DISPATCH_NR(nr, return (launch_fb_dec<NR, true>(st, tb, ek, iv, in, out, len / 16, (u32)(len % 16)))); break;
    case 1: {
        u64 n = len / 16;
        u32 r = (u32)(len % 16);
        if (n > 1 && !r) { --n; r = 16; }                 /* CS3 (:756) */
        if (r) --n;                                       /* hold the last two blocks (:764) */
        /* the n parallel blocks and the stolen pair behind them in ONE launch (k_fb_dec: one lane takes the pair) */
        DISPATCH_NR(nr, return (launch_fb_dec<NR, false>(st, tb, dk, iv, in, out, n, 0, r)));
    } break;
    default: return (int)hipErrorInvalidValue;
    }
    return 0;
}
