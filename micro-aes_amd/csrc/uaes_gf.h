/*
 * uaes_gf.h -- GF(2^128) helpers shared by the GHASH kernels (uaes_gcm.hip)
 * and their CPU unit test (tests/csrc/gf_check.cpp).  Pure integer code, no
 * memory access: compiles as host C++ (g++) and as gfx950 device code.
 *
 * Conventions (GCM, SP 800-38D; reference mulGF128/divideBblock,
 * micro_aes.c:464-493): a block is the polynomial whose x^0 coefficient is
 * the MSB of byte 0.  Held as two big-endian 64-bit halves:
 *     hi = bytes 0..7, lo = bytes 8..15;  coefficient of x^q = bit (127-q).
 * Multiplying by x is a 128-bit right shift; the bit that falls off re-enters
 * as R = 1 + x + x^2 + x^7, i.e. 0xE1 << 120.
 */
#ifndef UAES_GF_H_
#define UAES_GF_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define UAES_HD __host__ __device__ __forceinline__
#else
#define UAES_HD static inline
#endif

struct Gf {
    uint64_t hi, lo;
};

UAES_HD uint32_t gf_bswap32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24);
}

/* memory-order words (little-endian loads of bytes 0..3, 4..7, ...) <-> Gf */
UAES_HD Gf gf_from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    Gf g;
    g.hi = ((uint64_t)gf_bswap32(w0) << 32) | gf_bswap32(w1);
    g.lo = ((uint64_t)gf_bswap32(w2) << 32) | gf_bswap32(w3);
    return g;
}

UAES_HD void gf_to_words(Gf g, uint32_t w[4])
{
    w[0] = gf_bswap32((uint32_t)(g.hi >> 32));
    w[1] = gf_bswap32((uint32_t)g.hi);
    w[2] = gf_bswap32((uint32_t)(g.lo >> 32));
    w[3] = gf_bswap32((uint32_t)g.lo);
}

UAES_HD Gf gf_xor(Gf a, Gf b)
{
    Gf r;
    r.hi = a.hi ^ b.hi;
    r.lo = a.lo ^ b.lo;
    return r;
}

/* (o * R): o holds a polynomial of degree < 64 with x^0 at bit 63 */
UAES_HD Gf gf_fold(uint64_t o)
{
    Gf r;
    r.hi = o ^ (o >> 1) ^ (o >> 2) ^ (o >> 7);
    r.lo = (o << 63) ^ (o << 62) ^ (o << 57);
    return r;
}

/* v * x^k, 0 <= k <= 63 */
UAES_HD Gf gf_mul_xk(Gf v, uint32_t k)
{
    if (k == 0) return v;
    Gf r, f = gf_fold(v.lo << (64 - k));       /* the k bits that fall off */
    r.hi = (v.hi >> k) ^ f.hi;
    r.lo = ((v.lo >> k) | (v.hi << (64 - k))) ^ f.lo;
    return r;
}

/* v * x^64 */
UAES_HD Gf gf_mul_x64(Gf v)
{
    Gf r, f = gf_fold(v.lo);
    r.hi = f.hi;
    r.lo = v.hi ^ f.lo;
    return r;
}

/* coefficient of x^q in v */
UAES_HD uint32_t gf_coeff(Gf v, uint32_t q)
{
    return q < 64 ? (uint32_t)(v.hi >> (63 - q)) & 1u : (uint32_t)(v.lo >> (127 - q)) & 1u;
}

/* The slice of x*y that "lane" l (0..63) of a wave contributes in the
 * cooperative multiply: coefficients q = l and q = l + 64 of x.  XOR of the
 * 64 slices is the product.  y64 = gf_mul_x64(y).                          */
UAES_HD Gf gf_mul_slice(Gf x, Gf y, Gf y64, uint32_t l)
{
    Gf a = gf_mul_xk(y, l), b = gf_mul_xk(y64, l), r;
    const uint64_t ma = 0 - (uint64_t)gf_coeff(x, l), mb = 0 - (uint64_t)gf_coeff(x, l + 64);
    r.hi = (a.hi & ma) ^ (b.hi & mb);
    r.lo = (a.lo & ma) ^ (b.lo & mb);
    return r;
}

/* m * x^q for 0 <= q < 128 (the generators of the multiplication tables) */
UAES_HD Gf gf_mul_xq128(Gf m, uint32_t q)
{
    return q < 64 ? gf_mul_xk(m, q) : gf_mul_xk(gf_mul_x64(m), q - 64);
}

/* --- XTS tweak arithmetic (SP 800-38E; reference doubleLblock,
 * micro_aes.c:449-458): the block is a 128-bit LITTLE-endian integer
 * (lo = bytes 0..7); alpha multiplies by a left shift, overflow * 0x87.     */
struct Tw {
    uint64_t lo, hi;
};

UAES_HD void tw_fold(uint64_t o, uint64_t *flo, uint64_t *fhi)
{
    *flo = o ^ (o << 1) ^ (o << 2) ^ (o << 7);
    *fhi = (o >> 63) ^ (o >> 62) ^ (o >> 57);
}

/* t * alpha^k, 0 <= k <= 63 */
UAES_HD Tw tw_mul_pow(Tw t, uint32_t k)
{
    if (k == 0) return t;
    uint64_t flo, fhi;
    Tw r;
    tw_fold(t.hi >> (64 - k), &flo, &fhi);
    r.lo = (t.lo << k) ^ flo;
    r.hi = ((t.hi << k) | (t.lo >> (64 - k))) ^ fhi;
    return r;
}

/* t * alpha^64 */
UAES_HD Tw tw_mul_pow64(Tw t)
{
    uint64_t flo, fhi;
    Tw r;
    tw_fold(t.hi, &flo, &fhi);
    r.lo = flo;
    r.hi = t.lo ^ fhi;
    return r;
}

/* t * alpha^(256 l), 0 <= l <= 63: lane l of a wave that expands a data unit's chunk tweaks needs exactly this
 * (a chunk is 256 blocks).  Walking there by alpha^64 steps costs 4 l of them -- 252 for the last lane, and the wave
 * waits for that lane.  alpha^(256 2^s) mod P is SPARSE for small s (squaring in GF(2)[x] keeps the number of terms
 * until the exponents pass 128):
 *   s = 0: x^14+x^4+x^2+1   1: x^28+x^8+x^4+1   2: x^56+x^16+x^8+1   3: x^112+x^32+x^16+1
 *   s = 4: x^103+x^98+x^97+x^96+x^64+x^32+1     5: fourteen terms (listed below)
 * so the product is six conditional sums of 4, 4, 4, 4, 7 and 14 shifted copies: 37 shifts instead of up to 252
 * (checked against repeated doubling in tests/test_gf_helpers.py).                                          */
UAES_HD Tw tw_xor(Tw a, Tw b)
{
    Tw r;
    r.lo = a.lo ^ b.lo;
    r.hi = a.hi ^ b.hi;
    return r;
}

UAES_HD Tw tw_mul_a256(Tw t, uint32_t l)
{
#define TW_STEP(bit, EXPR)                                          \
    {                                                               \
        const Tw n_ = (EXPR);                                       \
        const uint64_t m_ = 0 - (uint64_t)((l >> (bit)) & 1u);      \
        t.lo ^= (t.lo ^ n_.lo) & m_;                                \
        t.hi ^= (t.hi ^ n_.hi) & m_;                                \
    }
    TW_STEP(0, tw_xor(tw_xor(t, tw_mul_pow(t, 2)), tw_xor(tw_mul_pow(t, 4), tw_mul_pow(t, 14))))
    TW_STEP(1, tw_xor(tw_xor(t, tw_mul_pow(t, 4)), tw_xor(tw_mul_pow(t, 8), tw_mul_pow(t, 28))))
    TW_STEP(2, tw_xor(tw_xor(t, tw_mul_pow(t, 8)), tw_xor(tw_mul_pow(t, 16), tw_mul_pow(t, 56))))
    {
        const Tw h = tw_mul_pow64(t);                                                    /* t * x^64 */
        TW_STEP(3, tw_xor(tw_xor(t, tw_mul_pow(t, 16)), tw_xor(tw_mul_pow(t, 32), tw_mul_pow(h, 48))))
    }
    {
        const Tw h = tw_mul_pow64(t);
        const Tw h32 = tw_mul_pow(h, 32);                                                /* t * x^96 */
        TW_STEP(4, tw_xor(tw_xor(tw_xor(t, tw_mul_pow(t, 32)), tw_xor(h, h32)),
                          tw_xor(tw_xor(tw_mul_pow(h32, 1), tw_mul_pow(h32, 2)), tw_mul_pow(h32, 7))))
    }
    {
        /* x^1+x^2+x^7 + x^65+x^67+x^69+x^70+x^71+x^73+x^75+x^78+x^79+x^80+x^85 */
        const Tw h = tw_mul_pow(tw_mul_pow64(t), 1);                                     /* t * x^65 */
        Tw a = tw_xor(tw_xor(tw_mul_pow(t, 1), tw_mul_pow(t, 2)), tw_mul_pow(t, 7));
        a = tw_xor(a, tw_xor(tw_xor(h, tw_mul_pow(h, 2)), tw_xor(tw_mul_pow(h, 4), tw_mul_pow(h, 5))));
        a = tw_xor(a, tw_xor(tw_xor(tw_mul_pow(h, 6), tw_mul_pow(h, 8)), tw_xor(tw_mul_pow(h, 10), tw_mul_pow(h, 13))));
        a = tw_xor(a, tw_xor(tw_xor(tw_mul_pow(h, 14), tw_mul_pow(h, 15)), tw_mul_pow(h, 20)));
        TW_STEP(5, a)
    }
#undef TW_STEP
    return t;
}

#endif
