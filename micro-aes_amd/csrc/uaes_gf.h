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

#endif
