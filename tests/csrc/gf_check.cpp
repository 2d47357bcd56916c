// CPU harness around micro-aes_amd/csrc/uaes_gf.h (the pure-integer GF(2^128)
// helpers the GHASH and XTS kernels are built from).  Compiled with g++ by
// tests/test_gf_helpers.py and checked against the oracle: the same source
// lines that run on the GPU are exercised here lane by lane.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../micro-aes_amd/csrc/uaes_gf.h"

static Gf load(const uint8_t b[16])
{
    uint32_t w[4];
    memcpy(w, b, 16);
    return gf_from_words(w[0], w[1], w[2], w[3]);
}

static void store(Gf g, uint8_t b[16])
{
    uint32_t w[4];
    gf_to_words(g, w);
    memcpy(b, w, 16);
}

extern "C" {

// x*y the way one wave does it: 64 lane slices XORed together
void gfc_wave_mul(const uint8_t x[16], const uint8_t y[16], uint8_t out[16])
{
    Gf X = load(x), Y = load(y), Y64 = gf_mul_x64(Y), z = { 0, 0 };
    for (uint32_t l = 0; l < 64; ++l) z = gf_xor(z, gf_mul_slice(X, Y, Y64, l));
    store(z, out);
}

// byte-indexed table of m (4096 x 16 B) exactly as k_gcm_setup builds it
void gfc_table8(const uint8_t m[16], uint8_t *table)
{
    Gf M = load(m), gen[128];
    for (uint32_t q = 0; q < 128; ++q) gen[q] = gf_mul_xq128(M, q);
    for (uint32_t j = 0; j < 16; ++j)
        for (uint32_t v = 0; v < 256; ++v) {
            Gf e = { 0, 0 };
            for (uint32_t i = 0; i < 8; ++i)
                if ((v >> (7 - i)) & 1) e = gf_xor(e, gen[8 * j + i]);
            store(e, table + (j * 256 + v) * 16);
        }
}

void gfc_table4(const uint8_t m[16], uint8_t *table)
{
    Gf M = load(m), gen[128];
    for (uint32_t q = 0; q < 128; ++q) gen[q] = gf_mul_xq128(M, q);
    for (uint32_t p = 0; p < 32; ++p)
        for (uint32_t v = 0; v < 16; ++v) {
            Gf e = { 0, 0 };
            for (uint32_t i = 0; i < 4; ++i)
                if ((v >> (3 - i)) & 1) e = gf_xor(e, gen[4 * p + i]);
            store(e, table + (p * 16 + v) * 16);
        }
}

static void tabmul8(const uint8_t *T, const uint8_t a[16], uint8_t z[16])
{
    uint8_t r[16] = { 0 };
    for (int j = 0; j < 16; ++j)
        for (int k = 0; k < 16; ++k) r[k] ^= T[(j * 256 + a[j]) * 16 + k];
    memcpy(z, r, 16);
}

static void tabmul4(const uint8_t *T, const uint8_t a[16], uint8_t z[16])
{
    uint8_t r[16] = { 0 };
    for (int j = 0; j < 16; ++j)
        for (int k = 0; k < 16; ++k)
            r[k] ^= T[((2 * j) * 16 + (a[j] >> 4)) * 16 + k] ^ T[((2 * j + 1) * 16 + (a[j] & 15)) * 16 + k];
    memcpy(z, r, 16);
}

void gfc_tabmul8(const uint8_t *T, const uint8_t a[16], uint8_t z[16]) { tabmul8(T, a, z); }
void gfc_tabmul4(const uint8_t *T, const uint8_t a[16], uint8_t z[16]) { tabmul4(T, a, z); }

// one strided-Horner level over n blocks (front padded), stride S, table of H^S:
// emulates k_ghash_pass / the stages of k_ghash_final.  accs: S x 16 bytes.
void gfc_level(const uint8_t *T, int nibble, const uint8_t *blocks, uint64_t n, uint64_t S, uint8_t *accs)
{
    const uint64_t steps = (n + S - 1) / S, pad = steps * S - n;
    for (uint64_t j = 0; j < S; ++j) {
        uint8_t acc[16] = { 0 };
        for (uint64_t k = 0; k < steps; ++k) {
            const uint64_t u = k * S + j;
            uint8_t t[16];
            if (nibble) tabmul4(T, acc, t); else tabmul8(T, acc, t);
            for (int b = 0; b < 16; ++b) acc[b] = t[b] ^ (u >= pad ? blocks[(u - pad) * 16 + b] : 0);
        }
        memcpy(accs + j * 16, acc, 16);
    }
}

// plain last level: acc <- (acc ^ X) * H through the nibble table of H
void gfc_last(const uint8_t *T4, const uint8_t *blocks, uint64_t n, uint8_t out[16])
{
    uint8_t acc[16] = { 0 };
    for (uint64_t k = 0; k < n; ++k) {
        for (int b = 0; b < 16; ++b) acc[b] ^= blocks[k * 16 + b];
        tabmul4(T4, acc, acc);
    }
    memcpy(out, acc, 16);
}

// t * alpha^(256 l), l < 64 (the chunk tweak of lane l: tw_mul_a256)
void gfc_tw_a256(const uint8_t t[16], uint32_t l, uint8_t out[16])
{
    Tw a;
    memcpy(&a.lo, t, 8); memcpy(&a.hi, t + 8, 8);
    Tw r = tw_mul_a256(a, l);
    memcpy(out, &r.lo, 8); memcpy(out + 8, &r.hi, 8);
}

// XTS tweak helpers: t * alpha^k (k < 64) and t * alpha^64, 16-byte LE blocks
void gfc_tw_pow(const uint8_t t[16], uint32_t k, uint8_t out[16])
{
    Tw a;
    memcpy(&a.lo, t, 8); memcpy(&a.hi, t + 8, 8);
    Tw r = k == 64 ? tw_mul_pow64(a) : tw_mul_pow(a, k);
    memcpy(out, &r.lo, 8); memcpy(out + 8, &r.hi, 8);
}

}
