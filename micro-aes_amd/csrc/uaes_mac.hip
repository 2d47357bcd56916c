/*
 * uaes_mac.hip -- the CBC-MAC based modes of the remaining NIST .rsp files
 * (SURVEY.md section 8f-1): CMAC and the authentication half of CCM.
 *
 *   k_cmac     <- AES_CMAC :1108-1118, cMac :576-590, getSubkeys :593-605,
 *                 doubleBblock :434-444
 *   k_ccm_tag  <- CCMtag :1222-1256 (used by AES_CCM_encrypt/decrypt :1268-1314;
 *                 the CTR half is k_ctr with the CCM/GCM pre-increment)
 *
 * A CBC-MAC is a strictly serial chain (M <- Enc(M ^ X_i)): what counts is the latency
 * of one block.  One wave walks the message; the sixteen lanes of a DPP row share each block
 * encryption (row_encrypt, uaes_aes.hip.h: one state byte per lane, DPP reduction).
 * Independent messages belong in the batched entry points (uaes_chain.hip), one lane each.
 */
#include <hip/hip_runtime.h>
#include <string.h>
#include "uaes_ctr.hip.h"
#include "uaes_device.h"

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

struct Blk {
    u32 w[4];
};

/* 16 bytes at p, zero padded after `avail` bytes */
__device__ __forceinline__ Blk ld_block(const unsigned char *p, u64 avail)
{
    Blk b = { { 0, 0, 0, 0 } };
    if (avail >= 16 && (((uintptr_t)p) & 15u) == 0) {
        const uint4 v = *(const uint4 *)p;
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        return b;
    }
    const u32 n = avail < 16 ? (u32)avail : 16u;
#pragma unroll                                                /* constant word indices: b stays in registers (a run-time
                                                                 index puts it into scratch memory, a round trip per access) */
    for (u32 i = 0; i < 16; ++i)
        if (i < n) b.w[i >> 2] |= (u32)p[i] << (8 * (i & 3));
    return b;
}

/* big-endian doubling in GF(2^128) (doubleBblock): <<1, carry -> ^0x87 in the last byte */
__device__ __forceinline__ Blk dbl_be(Blk b)
{
    u64 hi = ((u64)bswap32(b.w[0]) << 32) | bswap32(b.w[1]);
    u64 lo = ((u64)bswap32(b.w[2]) << 32) | bswap32(b.w[3]);
    const u64 carry = hi >> 63;
    hi = (hi << 1) | (lo >> 63);
    lo = (lo << 1) ^ (carry ? 0x87ull : 0ull);
    Blk r;
    r.w[0] = bswap32((u32)(hi >> 32)); r.w[1] = bswap32((u32)hi);
    r.w[2] = bswap32((u32)(lo >> 32)); r.w[3] = bswap32((u32)lo);
    return r;
}

__device__ __forceinline__ void xor_blk(Blk &a, const Blk &b)
{
    a.w[0] ^= b.w[0]; a.w[1] ^= b.w[1]; a.w[2] ^= b.w[2]; a.w[3] ^= b.w[3];
}

__device__ __forceinline__ void st_bytes(unsigned char *dst, const Blk &b)
{
    if (threadIdx.x != 0) return;                  /* every lane holds the same block: one stores */
    for (u32 i = 0; i < 16; ++i) dst[i] = (unsigned char)(b.w[i >> 2] >> (8 * (i & 3)));
}

/* M <- Enc(M ^ X_i) over the zero-padded 16-byte blocks of [p, p+len)   (xMac :551-570);
 * m = this lane's column word of M (row_encrypt, uaes_aes.hip.h: sixteen lanes per block) */
template <int NR>
__device__ __forceinline__ void cbcmac_absorb(u32 &m, const unsigned char *p, u64 len, const RowLane<NR> &L)
{
    if ((((uintptr_t)p) & 3u) == 0)
        row_walk<true>(p, len / 16, L.c, [&](u64, u32 x) { m = row_encrypt<NR>(m ^ x, L); });
    else
        row_walk<false>(p, len / 16, L.c, [&](u64, u32 x) { m = row_encrypt<NR>(m ^ x, L); });
    if (len % 16) m = row_encrypt<NR>(m ^ row_load(p + (len & ~(u64)15), len % 16, L.c), L);
}

/* a block every lane holds -> its encryption, likewise (the few odd blocks of a MAC: L, B0, Enc(iv)) */
template <int NR>
__device__ __forceinline__ void enc1(Blk &b, const RowLane<NR> &L)
{
    row_spread(row_encrypt<NR>(row_pick(b.w, L.c), L), b.w);
}

template <int NR>
__global__ __launch_bounds__(64) void k_cmac(uaesk_rk rk, uaesk_tables tb,
                                                  const unsigned char *__restrict__ data, u64 len,
                                                  unsigned char *__restrict__ mac)
{
    row_fill_tables(tb.te0, rk);                 /* one wave: sixteen lanes per block, four rows redundantly */
    const RowLane<NR> L = row_lane<NR>();
    Blk k1 = { { 0, 0, 0, 0 } };
    enc1<NR>(k1, L);                             /* L = Enc(0)                   */
    k1 = dbl_be(k1);                             /* K1 = 2L                      */
    const Blk k2 = dbl_be(k1);                   /* K2 = 4L                      */
    const u32 s = len ? (u32)((len - 1) % 16) + 1 : 0;      /* size of the last block */
    u32 m = 0;
    cbcmac_absorb<NR>(m, data, len - s, L);
    Blk last = ld_block(data + (len - s), s);
    if (s < 16) {
#pragma unroll
        for (u32 q = 0; q < 4; ++q)                           /* 10* padding, then K2 */
            last.w[q] ^= q == (s >> 2) ? 0x80u << (8 * (s & 3)) : 0u;
        xor_blk(last, k2);
    } else {
        xor_blk(last, k1);
    }
    m = row_encrypt<NR>(m ^ row_pick(last.w, L.c), L);
    row_store(mac, m, 16);
}

/* mode 0: write the first tag_len bytes of the tag to tag_io; mode 1: compare them with tag_io, *status = 0 / 0x1A
 * (tag_len = the reference's CCM_TAG_LEN, micro_aes.h:104: even, 4..16)                                       */
template <int NR>
__global__ __launch_bounds__(64) void k_ccm_tag(uaesk_rk rk, uaesk_tables tb, uint4 iv4,
                                                     const unsigned char *__restrict__ aad, u64 aad_len,
                                                     const unsigned char *__restrict__ pt, u64 pt_len,
                                                     int mode, unsigned char *tag_io, int *status, u32 tag_len)
{
    row_fill_tables(tb.te0, rk);                 /* one wave: sixteen lanes per block, four rows redundantly */
    const RowLane<NR> L = row_lane<NR>();
    const Blk iv = { { iv4.x, iv4.y, iv4.z, iv4.w } };
    /* B0 and the first AAD block are put together in words (byte arrays indexed at run time live in scratch memory) */
    Blk mb = iv;
    mb.w[0] |= (tag_len - 2) << 2;                        /* (CCM_TAG_LEN - 2) << 2 into byte 0, :1229 */
    mb.w[2] ^= bswap32((u32)((u64)pt_len >> 32));         /* xorBEint(M, ptextLen, LAST): big-endian, ending at byte 15 */
    mb.w[3] ^= bswap32((u32)pt_len);
    Blk ab = { { 0, 0, 0, 0 } };
    u64 s = 0;
    if (aad_len) {
        mb.w[0] |= 0x40u;
        enc1<NR>(mb, L);
        /* the length of the AAD in front of it: two bytes, or ff fe + four (the reference's p = 1 / 5, :1236-1241;
         * xorBEint keeps going while bits are left, so a length of 2^32 and more spills into the ff fe bytes) */
        const u32 hdr = aad_len > 0xFEFFull ? 6u : 2u;
        if (hdr == 6) {
            ab.w[0] = (0xFFu ^ ((u32)(aad_len >> 40) & 0xffu)) | (0xFEu ^ ((u32)(aad_len >> 32) & 0xffu)) << 8 |
                      ((u32)(aad_len >> 24) & 0xffu) << 16 | ((u32)(aad_len >> 16) & 0xffu) << 24;
            ab.w[1] = ((u32)(aad_len >> 8) & 0xffu) | ((u32)aad_len & 0xffu) << 8;
        } else {
            ab.w[0] = ((u32)(aad_len >> 8) & 0xffu) | ((u32)aad_len & 0xffu) << 8;
        }
        s = 16 - hdr;
        const u32 take = aad_len < s ? (u32)aad_len : (u32)s;
#pragma unroll
        for (u32 i = 2; i < 16; ++i)
            if (i >= hdr && i - hdr < take) ab.w[i >> 2] |= (u32)aad[i - hdr] << (8 * (i & 3));
    }
    xor_blk(mb, ab);                                      /* xMac(A, 16): also encrypts B0 when there is no AAD */
    u32 m = row_encrypt<NR>(row_pick(mb.w, L.c), L);
    if (aad_len > s) cbcmac_absorb<NR>(m, aad + s, aad_len - s, L);
    cbcmac_absorb<NR>(m, pt, pt_len, L);
    m ^= row_encrypt<NR>(row_pick(iv.w, L.c), L);         /* tag = Enc(iv) ^ CBC-MAC           */
    if (mode == 0) {
        row_store(tag_io, m, tag_len);
    } else {
        Blk t;
        row_spread(m, t.w);
        if (threadIdx.x == 0) {
            u32 diff = 0;
#pragma unroll
            for (u32 i = 0; i < 16; ++i)
                if (i < tag_len) diff |= (u32)tag_io[i] ^ ((t.w[i >> 2] >> (8 * (i & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
        }
    }
}

/* swap a value with lane ^ 16: DPP rows 0 <-> 1 and 2 <-> 3 (ds_swizzle, bit-mask mode: and 0x1f, or 0, xor 0x10) */
__device__ __forceinline__ u32 swap_rows(u32 v)
{
    return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401f);
}

/* the column word of the counter block of text block i (CTR_cipher with the CCM_GCM pre-increment: `ctr` starts at iv + 1) */
__device__ __forceinline__ u32 ccm_ctr_word(const uaesk_ctr &ctr, u64 i, u32 c)
{
    u32 w[4];
    ctr_words(ctr, i, w);
    return row_pick(w, c);
}

/* The text of a CCM message: CBC-MAC over the PLAINTEXT and CTR over the same blocks, in ONE wave.  row_encrypt runs
 * a block on sixteen lanes and the wave's other three rows used to repeat it; here rows 0 and 2 walk the MAC chain
 * and rows 1 and 3 encrypt the counter block -- the same instructions on different data -- so the CTR half costs no
 * time at all and a CCM call needs no k_ctr launch next to the MAC kernel (16 B: 18.7 -> 13.6 us, decrypt 30.5 -> 16.8).
 * Encrypt: the counter rows store C_i = P_i ^ KS_i.  Decrypt: the chain needs P_i = C_i ^ KS_i before it can go on,
 * so the counter rows run one block ahead and hand the keystream across (swap_rows); the chain rows store P_i. */
template <int NR, bool DEC, bool A4>
__device__ __forceinline__ void ccm_text(u32 &m, const uaesk_ctr &ctr, const unsigned char *in, unsigned char *out, u64 len,
                                         const RowLane<NR> &L)
{
    const bool ksrow = ((threadIdx.x >> 4) & 1u) != 0;
    const u64 nfull = len >> 4;
    const u32 rem = (u32)(len & 15u);
    u32 ksn = 0;                                   /* decrypt: the keystream of the next block to open (chain rows) */
    if (DEC) ksn = row_encrypt<NR>(ccm_ctr_word(ctr, 0, L.c), L);
    row_walk<A4>(in, nfull, L.c, [&](u64 i, u32 x) {
        if (!DEC) {
            const u32 w = row_encrypt<NR>(ksrow ? ccm_ctr_word(ctr, i, L.c) : (m ^ x), L);
            if (ksrow) row_store_full<A4>(out + 16 * i, x ^ w, L.c);
            else m = w;
        } else {
            const u32 p = x ^ ksn;
            const u32 w = row_encrypt<NR>(ksrow ? ccm_ctr_word(ctr, i + 1, L.c) : (m ^ p), L);
            if (!ksrow) { m = w; row_store_full<A4>(out + 16 * i, p, L.c); }
            ksn = swap_rows(w);                    /* the chain rows now hold KS_(i+1) */
        }
    });
    if (rem) {                                     /* the partial last block: zero padded into the MAC (xMac :551-570), cut in CTR (N3) */
        const u32 x = row_load(in + 16 * nfull, rem, L.c);
        const u32 keep = rem >= 4 * L.c + 4 ? 0xffffffffu : rem <= 4 * L.c ? 0u : (1u << (8 * (rem - 4 * L.c))) - 1u;
        u32 o;                                     /* this lane's column word of the output block */
        bool mine;                                 /* ... in the rows that hold it */
        if (!DEC) {
            const u32 w = row_encrypt<NR>(ksrow ? ccm_ctr_word(ctr, nfull, L.c) : (m ^ x), L);
            if (!ksrow) m = w;
            o = x ^ w;
            mine = ksrow;
        } else {
            o = (x ^ ksn) & keep;
            m = row_encrypt<NR>(m ^ o, L);
            mine = !ksrow;
        }
        if (mine && (threadIdx.x & 3u) == 0 && (threadIdx.x >> 5) == 0) {      /* one lane per column, rows 0 / 1 only */
            unsigned char *q = out + 16 * nfull + 4 * L.c;
#pragma unroll
            for (u32 k = 0; k < 4; ++k)
                if (4 * L.c + k < rem) q[k] = (unsigned char)(o >> (8 * k));
        }
    }
}

/* CCM in one launch of one wave (CCMtag :1222-1256 + CTR_cipher, AES_CCM_encrypt/decrypt :1268-1314).
 * encrypt: out = ciphertext, the first tag_len bytes of the tag to tag_io; decrypt: out = plaintext (the reference
 * writes it before it authenticates, :1304), the tag compared with tag_io, *status = 0 / 0x1A
 * (tag_len = the reference's CCM_TAG_LEN, micro_aes.h:104: even, 4..16)                                       */
template <int NR, bool DEC>
__global__ __launch_bounds__(64) void k_ccm(uaesk_rk rk, uaesk_tables tb, uint4 iv4, uaesk_ctr ctr,
                                                 const unsigned char *__restrict__ aad, u64 aad_len,
                                                 const unsigned char *in, u64 len, unsigned char *out,
                                                 unsigned char *tag_io, int *status, u32 tag_len)
{
    row_fill_tables(tb.te0, rk);                 /* one wave: sixteen lanes per block */
    const RowLane<NR> L = row_lane<NR>();
    const Blk iv = { { iv4.x, iv4.y, iv4.z, iv4.w } };
    /* B0 and the first AAD block are put together in words (byte arrays indexed at run time live in scratch memory) */
    Blk mb = iv;
    mb.w[0] |= (tag_len - 2) << 2;                        /* (CCM_TAG_LEN - 2) << 2 into byte 0, :1229 */
    mb.w[2] ^= bswap32((u32)((u64)len >> 32));            /* xorBEint(M, ptextLen, LAST): big-endian, ending at byte 15 */
    mb.w[3] ^= bswap32((u32)len);
    Blk ab = { { 0, 0, 0, 0 } };
    u64 s = 0;
    if (aad_len) {
        mb.w[0] |= 0x40u;
        enc1<NR>(mb, L);
        /* the length of the AAD in front of it: two bytes, or ff fe + four (the reference's p = 1 / 5, :1236-1241;
         * xorBEint keeps going while bits are left, so a length of 2^32 and more spills into the ff fe bytes) */
        const u32 hdr = aad_len > 0xFEFFull ? 6u : 2u;
        if (hdr == 6) {
            ab.w[0] = (0xFFu ^ ((u32)(aad_len >> 40) & 0xffu)) | (0xFEu ^ ((u32)(aad_len >> 32) & 0xffu)) << 8 |
                      ((u32)(aad_len >> 24) & 0xffu) << 16 | ((u32)(aad_len >> 16) & 0xffu) << 24;
            ab.w[1] = ((u32)(aad_len >> 8) & 0xffu) | ((u32)aad_len & 0xffu) << 8;
        } else {
            ab.w[0] = ((u32)(aad_len >> 8) & 0xffu) | ((u32)aad_len & 0xffu) << 8;
        }
        s = 16 - hdr;
        const u32 take = aad_len < s ? (u32)aad_len : (u32)s;
#pragma unroll
        for (u32 i = 2; i < 16; ++i)
            if (i >= hdr && i - hdr < take) ab.w[i >> 2] |= (u32)aad[i - hdr] << (8 * (i & 3));
    }
    xor_blk(mb, ab);                                      /* xMac(A, 16): also encrypts B0 when there is no AAD */
    u32 m = row_encrypt<NR>(row_pick(mb.w, L.c), L);
    if (aad_len > s) cbcmac_absorb<NR>(m, aad + s, aad_len - s, L);
    if (((((uintptr_t)in) | ((uintptr_t)out)) & 3u) == 0) ccm_text<NR, DEC, true>(m, ctr, in, out, len, L);
    else ccm_text<NR, DEC, false>(m, ctr, in, out, len, L);
    /* the chain lives in rows 0 and 2 from here on; what rows 1 and 3 make of it is never looked at */
    m ^= row_encrypt<NR>(row_pick(iv.w, L.c), L);         /* tag = Enc(iv) ^ CBC-MAC           */
    if (!DEC) {
        row_store(tag_io, m, tag_len);
    } else {
        Blk t;
        row_spread(m, t.w);
        if (threadIdx.x == 0) {
            u32 diff = 0;
#pragma unroll
            for (u32 i = 0; i < 16; ++i)
                if (i < tag_len) diff |= (u32)tag_io[i] ^ ((t.w[i >> 2] >> (8 * (i & 3))) & 0xffu);
            *status = diff ? 0x1A : 0;
        }
    }
}

#define DISPATCH_NR(nr, CALL)                         \
    switch (nr) {                                     \
    case 10: { constexpr int NR = 10; CALL; } break;  \
    case 12: { constexpr int NR = 12; CALL; } break;  \
    case 14: { constexpr int NR = 14; CALL; } break;  \
    default: return (int)hipErrorInvalidValue;        \
    }

template <int NR>
static int launch_cmac(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek,
                       const void *data, size_t len, void *mac)
{
    hipError_t e = uaesk_want_lds((const void *)k_cmac<NR>, (unsigned)(UAES_LDS_ROW));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_cmac<NR>), dim3(1), dim3(64), UAES_LDS_ROW, st, *ek, *tb,
                       (const unsigned char *)data, (u64)len, (unsigned char *)mac);
    return (int)hipGetLastError();
}

extern "C" int uaesk_cmac(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                          const void *data, size_t len, void *mac16)
{
    DISPATCH_NR(nr, return (launch_cmac<NR>(S(stream), tb, ek, data, len, mac16)));
    return 0;
}

#define CCM_FUSED_MAX 256u

template <int NR>
static int launch_ccm_tag(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, uint4 iv,
                          const void *aad, size_t aad_len, const void *pt, size_t pt_len,
                          int mode, void *tag_io, int *status, u32 tag_len)
{
    hipError_t e = uaesk_want_lds((const void *)k_ccm_tag<NR>, (unsigned)(UAES_LDS_ROW));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((k_ccm_tag<NR>), dim3(1), dim3(64), UAES_LDS_ROW, st, *ek, *tb, iv,
                       (const unsigned char *)aad, (u64)aad_len, (const unsigned char *)pt, (u64)pt_len,
                       mode, (unsigned char *)tag_io, status, tag_len);
    return (int)hipGetLastError();
}
template <int NR>
static int launch_ccm(hipStream_t st, const uaesk_tables *tb, const uaesk_rk *ek, uint4 iv, const uaesk_ctr &c, int decrypt,
                      const void *aad, size_t aad_len, const void *in, size_t len, void *out,
                      void *tag_io, int *status, u32 tag_len)
{
    hipError_t e;
    if (decrypt) {
        e = uaesk_want_lds((const void *)k_ccm<NR, true>, (unsigned)(UAES_LDS_ROW));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_ccm<NR, true>), dim3(1), dim3(64), UAES_LDS_ROW, st, *ek, *tb, iv, c,
                           (const unsigned char *)aad, (u64)aad_len, (const unsigned char *)in, (u64)len,
                           (unsigned char *)out, (unsigned char *)tag_io, status, tag_len);
    } else {
        e = uaesk_want_lds((const void *)k_ccm<NR, false>, (unsigned)(UAES_LDS_ROW));
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((k_ccm<NR, false>), dim3(1), dim3(64), UAES_LDS_ROW, st, *ek, *tb, iv, c,
                           (const unsigned char *)aad, (u64)aad_len, (const unsigned char *)in, (u64)len,
                           (unsigned char *)out, (unsigned char *)tag_io, status, tag_len);
    }
    return (int)hipGetLastError();
}

/* nonce is a host pointer; everything else device memory.  encrypt: tag over
 * `in` (the plaintext) written at out+len, then CTR in -> out.  decrypt: CTR
 * in -> out first, then the tag over the decrypted text is compared with the
 * tag_len bytes at in+len: like the reference (:1304-1312, SABOTAGE is a no-op in
 * its default build) the plaintext stays written even when *status = 0x1A.
 * nonce_len / tag_len = the reference's CCM_NONCE_LEN (7..13) / CCM_TAG_LEN (even, 4..16).  Whatever the
 * nonce length, the keystream counter is the 56-bit big-endian integer in bytes 9..15 (incBlock, :421-427).  */
extern "C" int uaesk_ccm(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                         int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                         const void *aad, size_t aad_len,
                         const void *in, size_t len, void *out, int *status)
{
    if (nonce_len < 7 || nonce_len > 13 || tag_len < 4 || tag_len > 16 || (tag_len & 1)) return (int)hipErrorInvalidValue;
    unsigned char ivb[16] = { 0 };
    ivb[0] = (unsigned char)(14 - nonce_len);            /* iv = { 14 - CCM_NONCE_LEN, nonce, 0... } (:1273) */
    memcpy(ivb + 1, nonce, nonce_len);
    uint4 iv;
    memcpy(&iv, ivb, 16);
    uaesk_ctr c;
    memset(&c, 0, sizeof c);
    memcpy(&c.w0, ivb, 4);
    memcpy(&c.w1, ivb + 4, 4);
    c.b8 = ivb[8];
    {
        uint64_t v = 0;
        for (int i = 9; i < 16; ++i) v = (v << 8) | ivb[i];
        c.v0 = (v + 1) & 0x00FFFFFFFFFFFFFFull;          /* pre-increment */
    }
    const u32 tl = (u32)tag_len;
    int rc;
    if (len <= CCM_FUSED_MAX) {
        /* a short message (CCM's usual diet): ONE launch, the MAC chain and the counter blocks share the wave
         * (ccm_text); its chain step is a third longer than the plain MAC's, so longer texts keep the two kernels */
        DISPATCH_NR(nr, rc = (launch_ccm<NR>(S(stream), tb, ek, iv, c, decrypt, aad, aad_len, in, len, out,
                                             decrypt ? (void *)((unsigned char *)in + len) : (void *)((unsigned char *)out + len),
                                             status, tl)));
        return rc;
    }
    if (!decrypt) {
        DISPATCH_NR(nr, rc = (launch_ccm_tag<NR>(S(stream), tb, ek, iv, aad, aad_len, in, len, 0,
                                                 (unsigned char *)out + len, nullptr, tl)));
        if (rc) return rc;
        return uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr);
    }
    rc = uaesk_ctr_xcrypt(stream, tb, nr, ek, &c, in, out, len, nullptr);
    if (rc) return rc;
    DISPATCH_NR(nr, rc = (launch_ccm_tag<NR>(S(stream), tb, ek, iv, aad, aad_len, out, len, 1,
                                             (unsigned char *)in + len, status, tl)));
    return rc;
}
