/*
 * uaes_host.c -- the engine's OWN host data path: portable C, table-driven, re-entrant, run-time key size.
 *
 * NOT the default path and not a silent fallback: every entry point of libuaes_hip.so runs on the GPU and fails loudly
 * without one unless the deployer has switched this path on (uaes_set_host_policy / UAES_HOST_MAX, UAES_HOST_CHAINS,
 * UAES_HOST_FALLBACK; include/uaes_hip.h).  It exists for the three places where a GPU cannot serve a caller of the
 * reference well (SURVEY.md 8b "Errors"): host-pointer calls too short to amortise a launch, ONE serial
 * chain (CBC / CFB encryption, OFB, CMAC, CCM's CBC-MAC: a latency-bound single wave on the GPU), and a `void` function
 * of the drop-in API on a box whose GPU has gone away.
 *
 * Product code, parity-tested like a kernel (tests/test_host_path.py runs the reference-held vectors and the golden
 * fixtures through it); it shares nothing with oracle/ (test infrastructure).  The arithmetic is the classic 32-bit
 * T-table formulation on the tables the engine derives from GF(2^8) anyway (uaes_engine.c, build_host_tables) and a
 * 4-bit-table GHASH; the semantics are the reference's, cited function by function.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "uaes_host.h"

/* ---- tables ------------------------------------------------------------ */
static uint32_t TE[4][256], TD[4][256];
static uint8_t  SB[256], ISB[256];

static uint32_t rol32(uint32_t v, unsigned n) { return n ? (v << n) | (v >> (32 - n)) : v; }

/* te0[x] = bytes {2S, S, S, 3S}[x], td0[x] = bytes {14Si, 9Si, 13Si, 11Si}[x] (little-endian words: byte 0 lowest).
 * Te_k = rotl(Te0, 8k) serves the byte of row k; S and Si fall out of the tables themselves.                       */
void uaesh_tables_init(const uint32_t te0[256], const uint32_t td0[256])
{
    unsigned x, k;
    for (x = 0; x < 256; ++x) {
        for (k = 0; k < 4; ++k) {
            TE[k][x] = rol32(te0[x], 8 * k);
            TD[k][x] = rol32(td0[x], 8 * k);
        }
        SB[x] = (uint8_t)(te0[x] >> 8);
    }
    for (x = 0; x < 256; ++x) ISB[SB[x]] = (uint8_t)x;
}

static uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static void st32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* ---- the block cipher (rijndaelEncrypt / rijndaelDecrypt, micro_aes.c:242-259, :315-332) ------------------------
 * State = four little-endian column words (byte r of word c = state[c][r], FIPS-197 order, a1 of SURVEY.md 8a).  One
 * round = SubBytes + ShiftRows + MixColumns as four lookups per column + AddRoundKey; the last round has no MixColumns. */
void uaesh_encrypt(const uint32_t *ek, int nr, const uint8_t in[16], uint8_t out[16])
{
    uint32_t s0 = ld32(in) ^ ek[0], s1 = ld32(in + 4) ^ ek[1], s2 = ld32(in + 8) ^ ek[2], s3 = ld32(in + 12) ^ ek[3];
    uint32_t t0, t1, t2, t3;
    int r;
    for (r = 1; r < nr; ++r) {
        const uint32_t *k = ek + 4 * r;
        t0 = TE[0][s0 & 255] ^ TE[1][(s1 >> 8) & 255] ^ TE[2][(s2 >> 16) & 255] ^ TE[3][s3 >> 24] ^ k[0];
        t1 = TE[0][s1 & 255] ^ TE[1][(s2 >> 8) & 255] ^ TE[2][(s3 >> 16) & 255] ^ TE[3][s0 >> 24] ^ k[1];
        t2 = TE[0][s2 & 255] ^ TE[1][(s3 >> 8) & 255] ^ TE[2][(s0 >> 16) & 255] ^ TE[3][s1 >> 24] ^ k[2];
        t3 = TE[0][s3 & 255] ^ TE[1][(s0 >> 8) & 255] ^ TE[2][(s1 >> 16) & 255] ^ TE[3][s2 >> 24] ^ k[3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    {
        const uint32_t *k = ek + 4 * nr;
        t0 = ((uint32_t)SB[s0 & 255] | (uint32_t)SB[(s1 >> 8) & 255] << 8 | (uint32_t)SB[(s2 >> 16) & 255] << 16 | (uint32_t)SB[s3 >> 24] << 24) ^ k[0];
        t1 = ((uint32_t)SB[s1 & 255] | (uint32_t)SB[(s2 >> 8) & 255] << 8 | (uint32_t)SB[(s3 >> 16) & 255] << 16 | (uint32_t)SB[s0 >> 24] << 24) ^ k[1];
        t2 = ((uint32_t)SB[s2 & 255] | (uint32_t)SB[(s3 >> 8) & 255] << 8 | (uint32_t)SB[(s0 >> 16) & 255] << 16 | (uint32_t)SB[s1 >> 24] << 24) ^ k[2];
        t3 = ((uint32_t)SB[s3 & 255] | (uint32_t)SB[(s0 >> 8) & 255] << 8 | (uint32_t)SB[(s1 >> 16) & 255] << 16 | (uint32_t)SB[s2 >> 24] << 24) ^ k[3];
    }
    st32(out, t0); st32(out + 4, t1); st32(out + 8, t2); st32(out + 12, t3);
}

/* the equivalent inverse cipher (FIPS-197 5.3.5) on dk[0] = ek[nr], dk[i] = InvMixColumns(ek[nr - i]), dk[nr] = ek[0]:
 * column c takes row r from column (c - r) mod 4                                                                    */
void uaesh_decrypt(const uint32_t *dk, int nr, const uint8_t in[16], uint8_t out[16])
{
    uint32_t s0 = ld32(in) ^ dk[0], s1 = ld32(in + 4) ^ dk[1], s2 = ld32(in + 8) ^ dk[2], s3 = ld32(in + 12) ^ dk[3];
    uint32_t t0, t1, t2, t3;
    int r;
    for (r = 1; r < nr; ++r) {
        const uint32_t *k = dk + 4 * r;
        t0 = TD[0][s0 & 255] ^ TD[1][(s3 >> 8) & 255] ^ TD[2][(s2 >> 16) & 255] ^ TD[3][s1 >> 24] ^ k[0];
        t1 = TD[0][s1 & 255] ^ TD[1][(s0 >> 8) & 255] ^ TD[2][(s3 >> 16) & 255] ^ TD[3][s2 >> 24] ^ k[1];
        t2 = TD[0][s2 & 255] ^ TD[1][(s1 >> 8) & 255] ^ TD[2][(s0 >> 16) & 255] ^ TD[3][s3 >> 24] ^ k[2];
        t3 = TD[0][s3 & 255] ^ TD[1][(s2 >> 8) & 255] ^ TD[2][(s1 >> 16) & 255] ^ TD[3][s0 >> 24] ^ k[3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    {
        const uint32_t *k = dk + 4 * nr;
        t0 = ((uint32_t)ISB[s0 & 255] | (uint32_t)ISB[(s3 >> 8) & 255] << 8 | (uint32_t)ISB[(s2 >> 16) & 255] << 16 | (uint32_t)ISB[s1 >> 24] << 24) ^ k[0];
        t1 = ((uint32_t)ISB[s1 & 255] | (uint32_t)ISB[(s0 >> 8) & 255] << 8 | (uint32_t)ISB[(s3 >> 16) & 255] << 16 | (uint32_t)ISB[s2 >> 24] << 24) ^ k[1];
        t2 = ((uint32_t)ISB[s2 & 255] | (uint32_t)ISB[(s1 >> 8) & 255] << 8 | (uint32_t)ISB[(s0 >> 16) & 255] << 16 | (uint32_t)ISB[s3 >> 24] << 24) ^ k[2];
        t3 = ((uint32_t)ISB[s3 & 255] | (uint32_t)ISB[(s2 >> 8) & 255] << 8 | (uint32_t)ISB[(s1 >> 16) & 255] << 16 | (uint32_t)ISB[s0 >> 24] << 24) ^ k[3];
    }
    st32(out, t0); st32(out + 4, t1); st32(out + 8, t2); st32(out + 12, t3);
}

static void xor16(uint8_t *d, const uint8_t *a, const uint8_t *b)
{
    int i;
    for (i = 0; i < 16; ++i) d[i] = a[i] ^ b[i];
}

/* ---- ECB (AES_ECB_encrypt / _decrypt, micro_aes.c:636-680; padBlock :610-621; N1) ------------------------------- */
void uaesh_ecb_encrypt(const uaesh_key *k, int padding, const uint8_t *in, size_t len, uint8_t *out)
{
    const size_t nfull = len / 16, rem = len % 16;
    size_t i;
    for (i = 0; i < nfull; ++i) uaesh_encrypt(k->ek, k->nr, in + 16 * i, out + 16 * i);
    if (rem || padding) {                           /* zeros behind a partial block; PKCS#7 / ISO 7816-4 always append */
        uint8_t b[16];
        const uint8_t n = (uint8_t)(16 - rem);
        memcpy(b, in + 16 * nfull, rem);
        memset(b + rem, padding == 1 ? n : 0, n);
        if (padding == 2) b[rem] = 0x80;
        uaesh_encrypt(k->ek, k->nr, b, out + 16 * nfull);
    }
}

void uaesh_ecb_decrypt(const uaesh_key *k, const uint8_t *in, size_t len, uint8_t *out)
{
    const size_t nfull = len / 16;
    size_t i;
    for (i = 0; i < nfull; ++i) uaesh_decrypt(k->dk, k->nr, in + 16 * i, out + 16 * i);
    if (len % 16 && out != in) memmove(out + 16 * nfull, in + 16 * nfull, len % 16);   /* the tail passes through (:664) */
}

/* ---- CTR (CTR_cipher micro_aes.c:919-950; incBlock :421-427; N2, N3) --------------------------------------------- */
/* counter block of stream block i: bytes 0..8 of ctr0 fixed, bytes 9..15 a 56-bit big-endian integer + block_offset + i */
static void ctr_block(uint8_t c[16], const uint8_t ctr0[16], uint64_t v)
{
    int i;
    memcpy(c, ctr0, 9);
    for (i = 15; i >= 9; --i, v >>= 8) c[i] = (uint8_t)v;
}

void uaesh_ctr(const uaesh_key *k, const uint8_t ctr0[16], uint64_t block_offset, const uint8_t *in, size_t len, uint8_t *out)
{
    uint64_t v = 0;
    uint8_t c[16], ks[16];
    size_t off, i;
    for (i = 9; i < 16; ++i) v = (v << 8) | ctr0[i];
    v += block_offset;
    for (off = 0; off < len; off += 16, ++v) {
        const size_t n = len - off < 16 ? len - off : 16;
        ctr_block(c, ctr0, v & 0x00FFFFFFFFFFFFFFull);
        uaesh_encrypt(k->ek, k->nr, c, ks);
        for (i = 0; i < n; ++i) out[off + i] = in[off + i] ^ ks[i];
    }
}

/* ---- XTS (XTS_cipher micro_aes.c:1008-1055; doubleLblock :449-458; N5) -------------------------------------------- */
static void xts_double(uint8_t t[16])
{
    unsigned c = 0;
    int i;
    for (i = 0; i < 16; ++i) {
        c |= (unsigned)t[i] << 1;
        t[i] = (uint8_t)c;
        c >>= 8;
    }
    t[0] ^= (uint8_t)(c * 0x87);
}

static void xex(const uaesh_key *k1, int encrypt, const uint8_t t[16], const uint8_t *in, uint8_t *out)
{
    uint8_t b[16];
    xor16(b, in, t);
    if (encrypt) uaesh_encrypt(k1->ek, k1->nr, b, b); else uaesh_decrypt(k1->dk, k1->nr, b, b);
    xor16(out, b, t);
}

/* one data unit, len >= 16; tweak = the sixteen raw bytes of the unit's identifier */
void uaesh_xts_unit(const uaesh_key *k1, const uaesh_key *k2, int encrypt, const uint8_t tweak[16],
                    const uint8_t *in, size_t len, uint8_t *out)
{
    const size_t r = len % 16, n = len / 16 - (r ? 1 : 0);
    uint8_t t[16];
    size_t i;
    uaesh_encrypt(k2->ek, k2->nr, tweak, t);
    for (i = 0; i < n; ++i) {
        xex(k1, encrypt, t, in + 16 * i, out + 16 * i);
        xts_double(t);
    }
    if (r) {                                        /* ciphertext stealing: decryption uses the last two tweaks swapped */
        uint8_t l[16], x[16], last[16];
        const uint8_t *px = in + 16 * n;
        uint8_t *py = out + 16 * n;
        memcpy(l, t, 16);
        xts_double(encrypt ? t : l);                /* encrypt: L = T_n, T = T_{n+1}; decrypt: L = T_{n+1}, T = T_n */
        xex(k1, encrypt, l, px, x);                 /* the last full block under L */
        memcpy(last, px + 16, r);                   /* the ragged chunk ... */
        memcpy(last + r, x + r, 16 - r);            /* ... completed with the tail it 'steals' */
        memcpy(py + 16, x, r);                      /* head of X becomes the final partial chunk */
        xex(k1, encrypt, t, last, py);
    }
}

void uaesh_xts_sectors(const uaesh_key *k1, const uaesh_key *k2, int encrypt, uint64_t first_sector, size_t sector_bytes,
                       size_t nsectors, const uint8_t *in, uint8_t *out)
{
    size_t s;
    for (s = 0; s < nsectors; ++s) {
        uint8_t tw[16] = { 0 };
        uint64_t id = first_sector + s;
        int i;
        for (i = 0; i < 8; ++i, id >>= 8) tw[i] = (uint8_t)id;          /* LE128(sector id): copyLint, :399-404 */
        uaesh_xts_unit(k1, k2, encrypt, tw, in + s * sector_bytes, sector_bytes, out + s * sector_bytes);
    }
}

/* ---- GHASH (mulGF128 micro_aes.c:476-493, xMac :551-570, gHash :1127-1137; N6) ------------------------------------
 * GCM's field: bit 0 of byte 0 is the coefficient of x^0, so "times x" is a right shift of the big-endian 128-bit
 * value, folding 0xE1 << 120 in when a bit falls off.  4-bit tables: M[n] = n(x) * H for the sixteen nibble values
 * (nibble bit 8 = the lowest power); Y * H = Horner over the 32 nibbles from the last one down, Z <- Z * x^4 ^ M[n].   */
typedef struct { uint64_t hi, lo; } gf128;
typedef struct { gf128 m[16]; uint64_t red[16]; } ghash_key;

static gf128 gf_mulx(gf128 v)
{
    const uint64_t carry = v.lo & 1;
    v.lo = (v.lo >> 1) | (v.hi << 63);
    v.hi = (v.hi >> 1) ^ (carry ? 0xE100000000000000ull : 0);
    return v;
}

static uint64_t be64(const uint8_t *p)
{
    uint64_t v = 0;
    int i;
    for (i = 0; i < 8; ++i) v = (v << 8) | p[i];
    return v;
}

static void put_be64(uint8_t *p, uint64_t v)
{
    int i;
    for (i = 7; i >= 0; --i, v >>= 8) p[i] = (uint8_t)v;
}

static void ghash_setup(ghash_key *g, const uint8_t h[16])
{
    gf128 v;
    unsigned n, b;
    v.hi = be64(h); v.lo = be64(h + 8);
    memset(g, 0, sizeof *g);
    g->m[8] = v;                                    /* x^0 */
    v = gf_mulx(v); g->m[4] = v;
    v = gf_mulx(v); g->m[2] = v;
    v = gf_mulx(v); g->m[1] = v;
    for (n = 3; n < 16; ++n) {
        if ((n & (n - 1)) == 0) continue;
        for (b = 1; b < 16; b <<= 1)
            if (n & b) { g->m[n].hi ^= g->m[b].hi; g->m[n].lo ^= g->m[b].lo; }
    }
    for (n = 0; n < 16; ++n) {                      /* what four bits falling off the low end fold back in */
        gf128 r;
        r.hi = 0; r.lo = n;
        r = gf_mulx(gf_mulx(gf_mulx(gf_mulx(r))));
        g->red[n] = r.hi;
    }
}

static void ghash_mul(const ghash_key *g, uint8_t y[16])
{
    gf128 z;
    int i;
    z = g->m[y[15] & 15];
    for (i = 15; i >= 0; --i) {
        unsigned rem;
        if (i != 15) {
            rem = (unsigned)z.lo & 15;
            z.lo = (z.lo >> 4) | (z.hi << 60);
            z.hi = (z.hi >> 4) ^ g->red[rem];
            z.hi ^= g->m[y[i] & 15].hi; z.lo ^= g->m[y[i] & 15].lo;
        }
        rem = (unsigned)z.lo & 15;
        z.lo = (z.lo >> 4) | (z.hi << 60);
        z.hi = (z.hi >> 4) ^ g->red[rem];
        z.hi ^= g->m[y[i] >> 4].hi; z.lo ^= g->m[y[i] >> 4].lo;
    }
    put_be64(y, z.hi); put_be64(y + 8, z.lo);
}

/* absorb: every 16 bytes Y <- (Y ^ X) * H; a partial last block is zero padded (xMac) */
static void ghash_absorb(const ghash_key *g, uint8_t y[16], const uint8_t *x, size_t len)
{
    size_t i;
    while (len >= 16) {
        for (i = 0; i < 16; ++i) y[i] ^= x[i];
        ghash_mul(g, y);
        x += 16; len -= 16;
    }
    if (len) {
        for (i = 0; i < len; ++i) y[i] ^= x[i];
        ghash_mul(g, y);
    }
}

static void ghash_lengths(const ghash_key *g, uint8_t y[16], uint64_t aad_len, uint64_t ct_len)
{
    uint8_t l[16];
    put_be64(l, aad_len * 8);
    put_be64(l + 8, ct_len * 8);
    ghash_absorb(g, y, l, 16);
}

void uaesh_ghash(const uint8_t h[16], const uint8_t *aad, size_t aad_len, const uint8_t *ct, size_t ct_len, uint8_t out[16])
{
    ghash_key g;
    ghash_setup(&g, h);
    memset(out, 0, 16);
    ghash_absorb(&g, out, aad, aad_len);
    ghash_absorb(&g, out, ct, ct_len);
    ghash_lengths(&g, out, aad_len, ct_len);
    memset(&g, 0, sizeof g);
}

static int differ(const uint8_t *a, const uint8_t *b, size_t n)         /* constant time */
{
    unsigned d = 0;
    size_t i;
    for (i = 0; i < n; ++i) d |= (unsigned)(a[i] ^ b[i]);
    return d != 0;
}

/* ---- GCM (GCMsetup micro_aes.c:1140-1152, AES_GCM_encrypt :1164-1179, _decrypt :1192-1212; N4, N7) ----------------
 * J0 = nonce || 00000001 for a 12-byte nonce, else GHASH(nonce); keystream block i under J0 + 1 + i with the reference's
 * 56-bit increment; tag = Enc(J0) ^ GHASH(aad, ct), tag_len bytes of it.  decrypt: in = ct || tag; returns 0x1A and
 * writes nothing when the tag does not match.                                                                       */
int uaesh_gcm(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
              const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out)
{
    ghash_key g;
    uint8_t h[16] = { 0 }, j0[16] = { 0 }, ej0[16], y[16] = { 0 };
    int rc = 0;
    uaesh_encrypt(k->ek, k->nr, h, h);
    ghash_setup(&g, h);
    if (nonce_len == 12) {
        memcpy(j0, nonce, 12);
        j0[15] = 1;
    } else {
        ghash_absorb(&g, j0, nonce, nonce_len);
        ghash_lengths(&g, j0, 0, nonce_len);
    }
    uaesh_encrypt(k->ek, k->nr, j0, ej0);
    if (!decrypt) {
        uaesh_ctr(k, j0, 1, in, len, out);
        ghash_absorb(&g, y, aad, aad_len);
        ghash_absorb(&g, y, out, len);
        ghash_lengths(&g, y, aad_len, len);
        xor16(y, y, ej0);
        memcpy(out + len, y, tag_len);
    } else {
        ghash_absorb(&g, y, aad, aad_len);
        ghash_absorb(&g, y, in, len);
        ghash_lengths(&g, y, aad_len, len);
        xor16(y, y, ej0);
        if (differ(y, in + len, tag_len)) rc = 0x1A;
        else uaesh_ctr(k, j0, 1, in, len, out);
    }
    memset(&g, 0, sizeof g); memset(h, 0, 16); memset(ej0, 0, 16);
    return rc;
}

/* ---- CBC (AES_CBC_encrypt micro_aes.c:697-744, _decrypt :746-782) ------------------------------------------------- */
/* cts != 0: CS3 ciphertext stealing -- the last two blocks are always swapped, a text shorter than one block is
 * M_DATALENGTH_ERROR (1).  cts == 0: the CTS 0 build, last chunk padded like ECB's (padding = AES_PADDING).          */
int uaesh_cbc_encrypt(const uaesh_key *k, const uint8_t iv[16], int cts, int padding, const uint8_t *in, size_t len, uint8_t *out)
{
    size_t n = len / 16, r = len % 16, i;
    uint8_t chain[16], b[16];
    if (cts) {
        if (n > 1 && r == 0) { --n; r = 16; }
        if (n == 0) return 1;
    }
    memcpy(chain, iv, 16);
    for (i = 0; i < n; ++i) {
        xor16(b, in + 16 * i, chain);
        uaesh_encrypt(k->ek, k->nr, b, chain);
        if (!(cts && r && i == n - 1)) memcpy(out + 16 * i, chain, 16);
    }
    if (cts) {
        if (r) {                                    /* C_{n-1} = chain: its head goes LAST, Enc(C_{n-1} ^ pad0(P_n)) takes its place */
            uint8_t last[16] = { 0 };
            memcpy(last, in + 16 * n, r);
            xor16(b, last, chain);
            uaesh_encrypt(k->ek, k->nr, b, b);
            memcpy(last, chain, r);                 /* (in may be out: everything of block n has been read) */
            memcpy(out + 16 * (n - 1), b, 16);
            memcpy(out + 16 * n, last, r);
        }
    } else if (r || padding) {
        const uint8_t pad = (uint8_t)(16 - r);
        memcpy(b, in + 16 * n, r);
        memset(b + r, padding == 1 ? pad : 0, pad);
        if (padding == 2) b[r] = 0x80;
        xor16(b, b, chain);
        uaesh_encrypt(k->ek, k->nr, b, out + 16 * n);
    }
    return 0;
}

int uaesh_cbc_decrypt(const uaesh_key *k, const uint8_t iv[16], int cts, const uint8_t *in, size_t len, uint8_t *out)
{
    size_t n = len / 16, r = len % 16, i;
    uint8_t chain[16], c[16], b[16];
    if (cts) {
        if (n > 1 && r == 0) { --n; r = 16; }
        if (n == 0) return 1;
    } else if (r) {
        return 1;
    }
    if (r) --n;                                     /* the last two blocks are the stolen pair */
    memcpy(chain, iv, 16);
    for (i = 0; i < n; ++i) {
        memcpy(c, in + 16 * i, 16);
        uaesh_decrypt(k->dk, k->nr, c, b);
        xor16(out + 16 * i, b, chain);
        memcpy(chain, c, 16);
    }
    if (r) {                                        /* {X full, Z r bytes}: Y = Dec(X); P_n = Z ^ Y; P_{n-1} = Dec(Z | tail of Y) ^ chain */
        uint8_t x[16], z[16], y[16], p2[16];
        memcpy(x, in + 16 * n, 16);
        memcpy(z, in + 16 * n + 16, r);
        uaesh_decrypt(k->dk, k->nr, x, y);
        for (i = 0; i < r; ++i) p2[i] = y[i] ^ z[i];
        memcpy(z + r, y + r, 16 - r);
        uaesh_decrypt(k->dk, k->nr, z, b);
        xor16(out + 16 * n, b, chain);
        memcpy(out + 16 * n + 16, p2, r);
    }
    return 0;
}

/* ---- CFB (CFB_cipher micro_aes.c:799-817) and OFB (AES_OFB_encrypt :861-885) -------------------------------------- */
void uaesh_cfb(const uaesh_key *k, const uint8_t iv[16], int encrypt, const uint8_t *in, size_t len, uint8_t *out)
{
    uint8_t chain[16], ks[16], c[16];
    size_t off, i;
    memcpy(chain, iv, 16);
    for (off = 0; off < len; off += 16) {
        const size_t n = len - off < 16 ? len - off : 16;
        uaesh_encrypt(k->ek, k->nr, chain, ks);
        memcpy(c, in + off, n);
        for (i = 0; i < n; ++i) out[off + i] = c[i] ^ ks[i];
        memcpy(chain, encrypt ? out + off : c, n);  /* the next input of the cipher is the CIPHERTEXT block */
    }
}

void uaesh_ofb(const uaesh_key *k, const uint8_t iv[16], const uint8_t *in, size_t len, uint8_t *out)
{
    uint8_t chain[16];
    size_t off, i;
    memcpy(chain, iv, 16);
    for (off = 0; off < len; off += 16) {
        const size_t n = len - off < 16 ? len - off : 16;
        uaesh_encrypt(k->ek, k->nr, chain, chain);
        for (i = 0; i < n; ++i) out[off + i] = in[off + i] ^ chain[i];
    }
}

/* ---- CMAC (AES_CMAC micro_aes.c:1108-1118, cMac :576-591, getSubkeys :594-605, doubleBblock :432-443) ------------- */
static void cmac_double(uint8_t b[16])
{
    unsigned c = 0;
    int i;
    for (i = 15; i >= 0; --i) {
        c |= (unsigned)b[i] << 1;
        b[i] = (uint8_t)c;
        c >>= 8;
    }
    b[15] ^= (uint8_t)(c * 0x87);
}

void uaesh_cmac(const uaesh_key *k, const uint8_t *data, size_t len, uint8_t mac[16])
{
    uint8_t k1[16] = { 0 }, k2[16], last[16] = { 0 };
    const size_t s = len ? (len - 1) % 16 + 1 : 0;  /* bytes of the last block: 1..16, 0 for the empty message */
    size_t off, i;
    uaesh_encrypt(k->ek, k->nr, k1, k1);
    cmac_double(k1);
    memcpy(k2, k1, 16);
    cmac_double(k2);
    memset(mac, 0, 16);
    for (off = 0; off + s < len; off += 16) {
        for (i = 0; i < 16; ++i) mac[i] ^= data[off + i];
        uaesh_encrypt(k->ek, k->nr, mac, mac);
    }
    memcpy(last, data + (len - s), s);
    if (s < 16) last[s] = 0x80;
    for (i = 0; i < 16; ++i) mac[i] ^= last[i] ^ (s < 16 ? k2[i] : k1[i]);
    uaesh_encrypt(k->ek, k->nr, mac, mac);
    memset(k1, 0, 16); memset(k2, 0, 16);
}

/* ---- CCM (CCMtag micro_aes.c:1226-1262, AES_CCM_encrypt :1268-1282, _decrypt :1294-1314) -------------------------- */
static void cbcmac_absorb(const uaesh_key *k, uint8_t m[16], const uint8_t *x, size_t len)
{
    size_t i;
    while (len) {
        const size_t n = len < 16 ? len : 16;
        for (i = 0; i < n; ++i) m[i] ^= x[i];
        uaesh_encrypt(k->ek, k->nr, m, m);
        x += n; len -= n;
    }
}

static void ccm_tag(const uaesh_key *k, const uint8_t iv[16], size_t tag_len, const uint8_t *aad, size_t aad_len,
                    const uint8_t *text, size_t len, uint8_t tag[16])
{
    uint8_t m[16], a[16] = { 0 }, s0[16];
    size_t v = len;
    int pos = 15;
    memcpy(m, iv, 16);
    m[0] |= (uint8_t)((tag_len - 2) << 2);
    do m[pos--] ^= (uint8_t)v; while ((v >>= 8) != 0);                  /* the text length, big-endian, ending at byte 15 */
    if (aad_len) {
        size_t p = 1, take;
        m[0] |= 0x40;
        uaesh_encrypt(k->ek, k->nr, m, m);
        if (aad_len > 0xFEFFu) { p = 5; a[0] = 0xFF; a[1] = 0xFE; }
        v = aad_len; pos = (int)p;
        do a[pos--] ^= (uint8_t)v; while ((v >>= 8) != 0);              /* the AAD length in front of the AAD */
        ++p;
        take = aad_len < 16 - p ? aad_len : 16 - p;
        memcpy(a + p, aad, take);
        cbcmac_absorb(k, m, a, 16);
        cbcmac_absorb(k, m, aad + take, aad_len - take);
    } else {
        cbcmac_absorb(k, m, a, 16);                                     /* the reference digests its zero A block (xMac of 16 bytes) */
    }
    cbcmac_absorb(k, m, text, len);
    uaesh_encrypt(k->ek, k->nr, iv, s0);
    xor16(tag, m, s0);
}

/* decrypt: in = ct || tag; like the reference it decrypts first and authenticates the result: 0x1A leaves the
 * unauthenticated text in `out` (the engine's wipe switch is the caller's business)                              */
int uaesh_ccm(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
              const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out)
{
    uint8_t iv[16] = { 0 }, tag[16], given[16];
    iv[0] = (uint8_t)(14 - nonce_len);
    memcpy(iv + 1, nonce, nonce_len);
    if (!decrypt) {
        ccm_tag(k, iv, tag_len, aad, aad_len, in, len, tag);            /* over the plaintext, before in may be overwritten */
        uaesh_ctr(k, iv, 1, in, len, out);
        memcpy(out + len, tag, tag_len);
        return 0;
    }
    memcpy(given, in + len, tag_len);
    uaesh_ctr(k, iv, 1, in, len, out);
    ccm_tag(k, iv, tag_len, aad, aad_len, out, len, tag);
    return differ(tag, given, tag_len) ? 0x1A : 0;
}

/* ---- GCM-SIV (RFC 8452; GCM_SIVsetup micro_aes.c:1434-1449, polyval :1421-1432, GCM_SIVtag :1452-1459,
 *      GCM_SIV_encrypt :1473-1485, _decrypt :1494-1515; CTR flavour SIVGCM_CTR :935-938) --------------------------------
 * POLYVAL runs through the GHASH tables above: POLYVAL(H, X_1..X_n) = ByteReverse(GHASH(mulX(ByteReverse(H)),
 * ByteReverse(X_1) .. ByteReverse(X_n)))  (RFC 8452 appendix A).                                                      */
static void rev16(uint8_t d[16], const uint8_t s[16])
{
    uint8_t t[16];
    int i;
    for (i = 0; i < 16; ++i) t[i] = s[15 - i];
    memcpy(d, t, 16);
}

static void polyval_absorb(const ghash_key *g, uint8_t y[16], const uint8_t *x, size_t len)
{
    uint8_t b[16];
    while (len) {
        const size_t n = len < 16 ? len : 16;
        memset(b, 0, 16);
        memcpy(b, x, n);
        rev16(b, b);
        ghash_absorb(g, y, b, 16);
        x += n; len -= n;
    }
}

static void put_le64(uint8_t *p, uint64_t v)
{
    int i;
    for (i = 0; i < 8; ++i, v >>= 8) p[i] = (uint8_t)v;
}

/* the 32-bit little-endian counter of the GCM-SIV keystream: bytes 0..3 of the block, wrapping mod 2^32 */
static void siv_ctr(const uint32_t *ek, int nr, const uint8_t tag[16], const uint8_t *in, size_t len, uint8_t *out)
{
    uint8_t c[16], ks[16];
    uint32_t v;
    size_t off, i;
    memcpy(c, tag, 16);
    c[15] |= 0x80;
    v = ld32(c);
    for (off = 0; off < len; off += 16, ++v) {
        const size_t n = len - off < 16 ? len - off : 16;
        st32(c, v);
        uaesh_encrypt(ek, nr, c, ks);
        for (i = 0; i < n; ++i) out[off + i] = in[off + i] ^ ks[i];
    }
}

/* expand: the engine's key schedule for the derived message-encryption key (ek words out) */
int uaesh_gcmsiv(const uaesh_key *master, int keybits, int decrypt, const uint8_t nonce[12],
                 const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out,
                 int (*expand)(int keybits, const uint8_t *key, uint32_t ek[60], uint32_t dk[60]))
{
    uint8_t blk[16], derived[16 + 32], hg[16], y[16] = { 0 }, pv[16], tag[16], lens[16];
    uint32_t ek[60], dk[60];
    ghash_key g;
    const unsigned nblk = 2u + (unsigned)keybits / 64u;     /* two for the authentication key, keybits / 64 for the cipher key */
    unsigned i;
    int nr, rc = 0;
    for (i = 0; i < nblk; ++i) {                    /* Enc_master(LE32(i) || nonce), the first eight bytes of each */
        uint8_t e[16];
        st32(blk, i);
        memcpy(blk + 4, nonce, 12);
        uaesh_encrypt(master->ek, master->nr, blk, e);
        memcpy(derived + 8 * i, e, 8);
    }
    nr = expand(keybits, derived + 16, ek, dk);
    if (nr < 0) return nr;
    /* GHASH key of POLYVAL's H: mulX(ByteReverse(H)) */
    {
        gf128 v;
        rev16(hg, derived);
        v.hi = be64(hg); v.lo = be64(hg + 8);
        v = gf_mulx(v);
        put_be64(hg, v.hi); put_be64(hg + 8, v.lo);
        ghash_setup(&g, hg);
    }
    if (decrypt) {                                  /* the tag is the counter: decrypt first, authenticate the result */
        memcpy(tag, in + len, 16);
        siv_ctr(ek, nr, tag, in, len, out);
    }
    polyval_absorb(&g, y, aad, aad_len);
    polyval_absorb(&g, y, decrypt ? out : in, len);
    put_le64(lens, (uint64_t)aad_len * 8);
    put_le64(lens + 8, (uint64_t)len * 8);
    polyval_absorb(&g, y, lens, 16);
    rev16(pv, y);
    for (i = 0; i < 12; ++i) pv[i] ^= nonce[i];
    pv[15] &= 0x7F;
    uaesh_encrypt(ek, nr, pv, pv);
    if (decrypt) {
        rc = differ(pv, tag, 16) ? 0x1A : 0;
    } else {
        siv_ctr(ek, nr, pv, in, len, out);
        memcpy(out + len, pv, 16);
    }
    memset(ek, 0, sizeof ek); memset(dk, 0, sizeof dk); memset(derived, 0, sizeof derived); memset(&g, 0, sizeof g);
    return rc;
}

/* ---- OCB (RFC 7253; OCB_cipher micro_aes.c:1693-1762, getDelta :1662-1680, AES_OCB_encrypt :1774-1784,
 *      _decrypt :1797-1811) -------------------------------------------------------------------------------------------
 * Offsets are chained here (Offset_i = Offset_{i-1} ^ L_ntz(i)); L_* = Enc(0), L_$ = 2 L_*, L_0 = 2 L_$, L_{k+1} = 2 L_k.  */
typedef struct { uint8_t star[16], dollar[16], l[64][16]; int have; } ocb_l;

static const uint8_t *ocb_lk(ocb_l *t, unsigned k)
{
    while (t->have <= (int)k) {
        memcpy(t->l[t->have], t->have ? t->l[t->have - 1] : t->dollar, 16);
        cmac_double(t->l[t->have]);
        t->have++;
    }
    return t->l[k];
}

static unsigned ntz64(uint64_t v)
{
    unsigned n = 0;
    while (!(v & 1)) { v >>= 1; ++n; }
    return n;
}

static void ocb_hash(const uaesh_key *k, ocb_l *t, const uint8_t *aad, size_t aad_len, uint8_t sum[16])
{
    uint8_t off[16] = { 0 }, b[16];
    uint64_t i;
    size_t j;
    memset(sum, 0, 16);
    for (i = 1; aad_len >= 16; ++i, aad += 16, aad_len -= 16) {
        xor16(off, off, ocb_lk(t, ntz64(i)));
        xor16(b, aad, off);
        uaesh_encrypt(k->ek, k->nr, b, b);
        xor16(sum, sum, b);
    }
    if (aad_len) {
        xor16(off, off, t->star);
        memset(b, 0, 16);
        memcpy(b, aad, aad_len);
        b[aad_len] = 0x80;
        for (j = 0; j < 16; ++j) b[j] ^= off[j];
        uaesh_encrypt(k->ek, k->nr, b, b);
        xor16(sum, sum, b);
    }
}

/* decrypt: in = ct || tag; like the reference the text is written either way and 0x1A reports a mismatch */
int uaesh_ocb(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
              const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out)
{
    ocb_l *t = (ocb_l *)calloc(1, sizeof *t);
    uint8_t kt[24], off[16], sum[16] = { 0 }, b[16], tag[16], given[16], hash[16];
    const unsigned bottom = nonce[nonce_len - 1] & 63u;
    size_t nfull = len / 16, rem = len % 16, j;
    uint64_t i;
    int rc = 0;
    if (!t) return -1;
    uaesh_encrypt(k->ek, k->nr, t->star, t->star);                       /* L_* = Enc(0) */
    memcpy(t->dollar, t->star, 16);
    cmac_double(t->dollar);
    /* Nonce block: taglen (bits) mod 128 in the top seven bits, zeros, a one bit, the nonce; its last six bits cleared */
    memset(kt, 0, sizeof kt);
    memcpy(kt + 16 - nonce_len, nonce, nonce_len);
    kt[0] |= (uint8_t)(tag_len << 4);
    kt[15 - nonce_len] |= 1;
    kt[15] &= 0xC0;
    uaesh_encrypt(k->ek, k->nr, kt, kt);                                  /* Ktop */
    for (j = 0; j < 8; ++j) kt[16 + j] = kt[j] ^ kt[j + 1];               /* Stretch = Ktop || (Ktop[0..7] ^ Ktop[1..8]) */
    for (j = 0; j < 16; ++j) {                                            /* Offset_0 = Stretch << bottom, top 128 bits */
        const unsigned v = ((unsigned)kt[j + bottom / 8] << 8) | kt[j + bottom / 8 + 1];
        off[j] = (uint8_t)(v >> (8 - bottom % 8));
    }
    if (decrypt) memcpy(given, in + len, tag_len);
    for (i = 1; i <= nfull; ++i, in += 16, out += 16) {
        xor16(off, off, ocb_lk(t, ntz64(i)));
        if (!decrypt) {
            xor16(sum, sum, in);
            xor16(b, in, off);
            uaesh_encrypt(k->ek, k->nr, b, b);
            xor16(out, b, off);
        } else {
            xor16(b, in, off);
            uaesh_decrypt(k->dk, k->nr, b, b);
            xor16(out, b, off);
            xor16(sum, sum, out);
        }
    }
    if (rem) {
        uint8_t pad[16], p[16] = { 0 };
        xor16(off, off, t->star);
        uaesh_encrypt(k->ek, k->nr, off, pad);
        if (!decrypt) memcpy(p, in, rem);
        for (j = 0; j < rem; ++j) out[j] = in[j] ^ pad[j];
        if (decrypt) memcpy(p, out, rem);
        p[rem] = 0x80;
        xor16(sum, sum, p);
        out += rem;
    }
    for (j = 0; j < 16; ++j) b[j] = sum[j] ^ off[j] ^ t->dollar[j];
    uaesh_encrypt(k->ek, k->nr, b, tag);
    ocb_hash(k, t, aad, aad_len, hash);
    xor16(tag, tag, hash);
    if (decrypt) rc = differ(tag, given, tag_len) ? 0x1A : 0;
    else memcpy(out, tag, tag_len);
    memset(t, 0, sizeof *t);
    free(t);
    return rc;
}
