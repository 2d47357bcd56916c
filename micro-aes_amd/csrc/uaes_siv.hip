/*
 * uaes_siv.hip -- a short GCM-SIV message (RFC 8452; GCM_SIV_encrypt / _decrypt, micro_aes.c:1421-1515) in one workgroup:
 * key derivation, POLYVAL, the tag and the keystream in one launch (k_siv_small).  Longer messages: uaesk_gcmsiv_long in
 * uaes_gcm.hip (they run on the chunk workgroups).  GHASH machinery: uaes_ghash.hip.h.
 */
#include "uaes_ghash.hip.h"

int uaesk_plan_siv(int dir, size_t len, size_t aad_len, unsigned flags, uaes_plan *p);     /* uaes_gcm.hip */

/* ------------------------------------------------------------------------ */
/* short GCM-SIV messages (RFC 8452; GCM_SIV_encrypt/decrypt, micro_aes.c:1421-1515) in one workgroup */
/* ------------------------------------------------------------------------ */
/* The same shape as k_gcm_small with the order of GCM-SIV: the tag comes from the PLAINTEXT
 * (POLYVAL = the GHASH levels on byte-reversed blocks under the key mulX(rev(H)), which the host passes as
 * `hg`), and the counter of the keystream is the tag.  Encrypt: POLYVAL over AAD || plaintext || lengths,
 * tag = Enc((rev(S) ^ nonce) with the top bit cleared), then every thread encrypts the counter blocks of its
 * text positions.  Decrypt: keystream from the RECEIVED tag first, the plaintext is written (the reference
 * releases it before it authenticates, :1500-1511), then the same POLYVAL and the comparison.  The key's
 * nibble tables are made in the kernel (gcm_build_nibble_tables; the message-authentication key is per NONCE
 * here, so there is nothing to keep), and so are the per-nonce keys themselves (derive_keys under the master
 * key + KeyExpansion of the derived key).  One launch instead of key-derivation ECB + POLYVAL setup / levels +
 * tag ECB + CTR with three host round trips in between: a 4 KiB call 100 -> 30 us.                        */
#define SIV_LDS_KEYS   GSM_LDS_TOTAL                    /* 8 derived + 60 schedule words, then a plain copy of Te0 */
#define SIV_LDS_TE     (SIV_LDS_KEYS + 512u)
#define SIV_LDS_TOTAL  (SIV_LDS_TE + 1024u)

template <int NR, bool DEC>
__global__ __launch_bounds__(GH_T) void k_siv_small(uaesk_rk mk, uaesk_tables tb, uint4 nonce,
                                                    GSrc src, const uint4 *in, uint4 *out,
                                                    unsigned char *tag_io, int *status)
{
    uint4 *TC = (uint4 *)(uaes_lds + GSM_LDS_TAB);
    uint4 *buf = TC + GT_NTAB * 512u;
    u32 *drv = (u32 *)(uaes_lds + SIV_LDS_KEYS);              /* [0..11] derived words, [16..75] the schedule */
    u32 *ekl = drv + 16;
    u32 *te_plain = (u32 *)(uaes_lds + SIV_LDS_TE);
    if (threadIdx.x < 256) te_plain[threadIdx.x] = tb.te0[threadIdx.x];
    fill_tables64(tb.te0, 0);                                 /* ends with a barrier */
    const LaneConst2 lc = make_lane_const2(0);

    /* derive_keys (RFC 8452 sec. 4; GCM_SIV_init, micro_aes.c:1421-1450) in the kernel: blocks LE32(i) || nonce
     * under the MASTER key, i < 2 + keybits/64, one lane each; their low halves are the message-authentication
     * key (blocks 0, 1) and the message-encryption key (blocks 2..), whose schedule thread 0 expands here
     * (KeyExpansion :144-178; S[x] = byte 1 of Te0[x]).  The host only expands the master key.            */
    constexpr u32 NK = NR - 6, NB = 2 + NK / 2;
    if (threadIdx.x < 64) {
        u32 s1[1][4] = { { threadIdx.x, nonce.x, nonce.y, nonce.z } };
        enc_blocks<NR, 1>(s1, mk, lc);
        if (threadIdx.x < NB) { drv[2 * threadIdx.x] = s1[0][0]; drv[2 * threadIdx.x + 1] = s1[0][1]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint4 auth = make_uint4(drv[0], drv[1], drv[2], drv[3]);
        u32 hw[4];
        gf_to_words(gf_mul_xk(gf_from4(rev16(auth)), 1), hw);    /* POLYVAL key in GHASH form: mulX(rev(H)) */
        buf[GT_BUF - 3] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        auto subword = [&](u32 w) -> u32 {
            return ((te_plain[w & 0xffu] >> 8) & 0xffu) | (te_plain[(w >> 8) & 0xffu] & 0xff00u) |
                   ((te_plain[(w >> 16) & 0xffu] & 0xff00u) << 8) | ((te_plain[w >> 24] & 0xff00u) << 16);
        };
        /* the schedule grows in REGISTERS (fully unrolled: constant indices) and goes to LDS at the end: with ekl[] as
         * the working array every word waited for an LDS store and two loads, ~200 cycles x 40..52 words = 3.7 us */
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
#pragma unroll
        for (u32 i = 0; i < 4u * (NR + 1); ++i) ekl[i] = w[i];
    }
    __syncthreads();
    uaesk_rk rk;
#pragma unroll
    for (int i = 0; i < 4 * (NR + 1); ++i) rk.w[i] = (u32)__builtin_amdgcn_readfirstlane((int)ekl[i]);
    gcm_build_nibble_tables(TC, buf, tb.frob);

    const u64 len = src.ct_len;
    const u64 ablk = (src.aad_len + 15) >> 4, cblk = (len + 15) >> 4, nv = ablk + cblk + 1;
    const u32 steps = nv + 2 > GH_T ? 2u : 1u;
    const u64 pad = (u64)steps * GH_T - nv;
    GSrc rest = src;                                          /* AAD blocks and the length block */
    rest.ct_len = 0;

    uaesk_ctr ctr;                                            /* the keystream counter = the tag, byte 15 |= 0x80, LE32 in bytes 0..3 (:935-938) */
    ctr.le32 = 1; ctr.v0 = 0; ctr.b8 = 0;
    if (DEC) {
        u32 w[4] = { 0, 0, 0, 0 };
        for (u32 b = 0; b < 16; ++b) w[b >> 2] |= (u32)tag_io[b] << (8 * (b & 3));
        ctr.w0 = w[0]; ctr.w1 = w[1]; ctr.w2 = w[2]; ctr.w3 = w[3] | 0x80000000u;
    }

    /* one text block: out = in ^ Enc(counter i); returns the PLAINTEXT block, zero padded */
    auto crypt_block = [&](u64 i) -> uint4 {
        const u64 avail = len - 16 * i;
        const u32 nb = avail < 16 ? (u32)avail : 16u;
        u32 s1[1][4];
        ctr_words(ctr, i, s1[0]);
        enc_blocks<NR, 1>(s1, rk, lc);
        const uint4 d = nb == 16 ? in[i] : load_bytes_padded((const unsigned char *)(in + i), nb);
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
        return DEC ? make_uint4(o[0], o[1], o[2], o[3]) : d;
    };

    /* POLYVAL input of this thread's positions (decrypt: produced by decrypting them) */
    uint4 xk[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
    for (u32 k = 0; k < 2; ++k) {
        if (k >= steps) break;
        const u64 u = (u64)k * GH_T + threadIdx.x;
        if (u < pad) continue;
        const u64 v = u - pad;
        if (v >= ablk && v < ablk + cblk) {
            const u64 i = v - ablk;
            uint4 p;
            if (DEC) {
                p = crypt_block(i);
            } else {
                const u64 avail = len - 16 * i;
                p = avail >= 16 ? in[i] : load_bytes_padded((const unsigned char *)(in + i), (u32)avail);
            }
            xk[k] = rev16(p);
        } else {
            xk[k] = load_vblock(rest, v < ablk ? v : ablk);
        }
    }
    uint4 acc = xk[0];
    if (steps == 2) acc = x4(tabmul4(TC, acc), xk[1]);
    acc = gh_tree<true>(buf, TC, acc, steps == 1 ? (u32)nv : GH_T);
    if (threadIdx.x == 0) {                                   /* S = POLYVAL ^ nonce, top bit cleared (GCM_SIVtag :1453-1460) */
        uint4 sv = rev16(acc);
        sv.x ^= nonce.x; sv.y ^= nonce.y; sv.z ^= nonce.z;
        sv.w &= 0x7fffffffu;
        buf[GT_BUF - 2] = sv;
    }
    __syncthreads();
    u32 t1[1][4];
    {
        const uint4 sv = buf[GT_BUF - 2];
        t1[0][0] = sv.x; t1[0][1] = sv.y; t1[0][2] = sv.z; t1[0][3] = sv.w;
    }
    enc_blocks<NR, 1>(t1, rk, lc);                            /* every thread: the tag */
    if (DEC) {
        if (threadIdx.x == 0) {
            u32 diff = 0;
            for (u32 b = 0; b < 16; ++b) diff |= (u32)tag_io[b] ^ ((t1[0][b >> 2] >> (8 * (b & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
        }
        return;
    }
    ctr.w0 = t1[0][0]; ctr.w1 = t1[0][1]; ctr.w2 = t1[0][2]; ctr.w3 = t1[0][3] | 0x80000000u;
#pragma unroll                                                /* constant indices: the per-step arrays stay in registers */
    for (u32 k = 0; k < 2; ++k) {
        if (k >= steps) break;
        const u64 u = (u64)k * GH_T + threadIdx.x;
        if (u < pad) continue;
        const u64 v = u - pad;
        if (v >= ablk && v < ablk + cblk) (void)crypt_block(v - ablk);
    }
    if (threadIdx.x == 0)
        for (u32 b = 0; b < 16; ++b) tag_io[b] = (unsigned char)(t1[0][b >> 2] >> (8 * (b & 3)));
}

/* GCM-SIV of a short message in one launch (k_siv_small), key derivation included: mk = the schedule of the MASTER
 * key.  encrypt: tag written at out + len; decrypt: tag read at in + len, *status = 0 / 0x1A, the plaintext is
 * written either way.  Returns -1 if the message is too long for this arrangement (siv_plan): the caller then takes
 * uaesk_gcmsiv_long.                                                                                          */
extern "C" int uaesk_gcmsiv_small(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *mk, int decrypt,
                                  const uint8_t *nonce12,
                                  const void *aad, size_t aad_len, const void *in, size_t len, void *out, int *status)
{
    uaes_plan sp;
    if (uaesk_plan_siv(decrypt ? 1 : 0, len, aad_len, 0, &sp) != 0 || sp.arrangement != UAES_ARR_SIV_SMALL) return -1;
    GSrc src;
    src.aad = (const unsigned char *)aad; src.aad_len = aad_len;
    src.ct = (const unsigned char *)in; src.ct_len = len;
    src.has_len = 1; src.len_aad = aad_len; src.len_ct = len; src.rev = 1;
    uint4 nn = make_uint4(0, 0, 0, 0);
    memcpy(&nn, nonce12, 12);
    hipStream_t st = S(stream);
    hipError_t e;
#define SIV_LAUNCH(NRV, D)                                                                                          \
    do {                                                                                                            \
        e = uaesk_want_lds((const void *)k_siv_small<NRV, D>, (unsigned)(SIV_LDS_TOTAL));                           \
        if (e != hipSuccess) return (int)e;                                                                         \
        hipLaunchKernelGGL((k_siv_small<NRV, D>), dim3(1), dim3(GH_T), SIV_LDS_TOTAL, st, *mk, *tb, nn, src,        \
                           (const uint4 *)in, (uint4 *)out,                                                         \
                           (D) ? (unsigned char *)in + len : (unsigned char *)out + len, status);                   \
    } while (0)
    switch (nr) {
    case 10: if (decrypt) SIV_LAUNCH(10, true); else SIV_LAUNCH(10, false); break;
    case 12: if (decrypt) SIV_LAUNCH(12, true); else SIV_LAUNCH(12, false); break;
    case 14: if (decrypt) SIV_LAUNCH(14, true); else SIV_LAUNCH(14, false); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef SIV_LAUNCH
    return (int)hipGetLastError();
}
