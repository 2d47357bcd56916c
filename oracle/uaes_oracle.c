/*
 * uaes_oracle.c -- CPU oracle for the uAES hot path (TEST INFRASTRUCTURE ONLY;
 * see uaes_oracle.h for the parity-pinning statement and the usage rule).
 *
 * Written from FIPS-197 / SP 800-38A,D,E following the behaviour of the
 * reference's functions cited at each definition (paths relative to the
 * reference checkout, e.g. micro_aes.c:242).  Byte-oriented and deliberately
 * simple: clarity over speed.
 */
#include "uaes_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------------ */
/* GF(2^8) and the S-box (reference: literal tables, micro_aes.c:41-65).      */
/* Here the tables are derived: S(x) = affine(x^-1) per FIPS-197 sec. 5.1.1.  */
/* ------------------------------------------------------------------------ */
static uint8_t SB[256], ISB[256];
static int tables_ready;

static uint8_t gf_double(uint8_t a)           /* micro_aes.c:115 xtime */
{
    return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0));
}

static uint8_t gf_mul(uint8_t a, uint8_t b)
{
    uint8_t p = 0;
    while (b) {
        if (b & 1) p ^= a;
        a = gf_double(a);
        b >>= 1;
    }
    return p;
}

static void make_tables(void)
{
    int x;
    if (tables_ready) return;
    for (x = 0; x < 256; ++x) {
        uint8_t inv = 0, s;
        int y;
        if (x)
            for (y = 1; y < 256; ++y)
                if (gf_mul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        s = inv;
        s ^= (uint8_t)((inv << 1) | (inv >> 7));
        s ^= (uint8_t)((inv << 2) | (inv >> 6));
        s ^= (uint8_t)((inv << 3) | (inv >> 5));
        s ^= (uint8_t)((inv << 4) | (inv >> 4));
        s ^= 0x63;
        SB[x] = s;
        ISB[s] = (uint8_t)x;
    }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------ */
/* Key schedule -- micro_aes.c:144-178                                        */
/* ------------------------------------------------------------------------ */
int orc_setkey(orc_key *ks, const uint8_t *key, int keybits)
{
    int nk, nwords, i;
    uint8_t rcon = 1;
    uint8_t *w = ks->rk;

    if (keybits != 128 && keybits != 192 && keybits != 256) return -1;
    make_tables();
    nk = keybits / 32;
    ks->nr = nk + 6;
    nwords = 4 * (ks->nr + 1);
    memcpy(w, key, (size_t)(4 * nk));
    for (i = nk; i < nwords; ++i) {
        uint8_t t[4];
        memcpy(t, w + 4 * (i - 1), 4);
        if (i % nk == 0) {                     /* RotWord, SubWord, Rcon      */
            uint8_t t0 = t[0];
            t[0] = (uint8_t)(SB[t[1]] ^ rcon);
            t[1] = SB[t[2]];
            t[2] = SB[t[3]];
            t[3] = SB[t0];
            rcon = gf_double(rcon);            /* wraps 0x80 -> 0x1b (:155)   */
        } else if (nk == 8 && i % nk == 4) {   /* AES-256 extra SubWord (:165)*/
            t[0] = SB[t[0]]; t[1] = SB[t[1]]; t[2] = SB[t[2]]; t[3] = SB[t[3]];
        }
        w[4 * i + 0] = (uint8_t)(w[4 * (i - nk) + 0] ^ t[0]);
        w[4 * i + 1] = (uint8_t)(w[4 * (i - nk) + 1] ^ t[1]);
        w[4 * i + 2] = (uint8_t)(w[4 * (i - nk) + 2] ^ t[2]);
        w[4 * i + 3] = (uint8_t)(w[4 * (i - nk) + 3] ^ t[3]);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Block cipher.  State byte i = column i/4, row i%4 (micro_aes.c:74-77).     */
/* ------------------------------------------------------------------------ */
static void xor16(uint8_t *dst, const uint8_t *src)   /* micro_aes.c:105 */
{
    int i;
    for (i = 0; i < 16; ++i) dst[i] ^= src[i];
}

/* micro_aes.c:242-259: ARK(r) SubBytes ShiftRows [MixColumns | ARK(last)] */
void orc_encrypt_block(const orc_key *ks, const uint8_t in[16], uint8_t out[16])
{
    uint8_t s[16], t[16];
    int r, c;
    memcpy(s, in, 16);
    for (r = 0; r < ks->nr; ++r) {
        xor16(s, ks->rk + 16 * r);
        /* SubBytes (:187) fused with ShiftRows (:198): row k moves left by k */
        for (c = 0; c < 4; ++c) {
            t[4 * c + 0] = SB[s[4 * c + 0]];
            t[4 * c + 1] = SB[s[4 * ((c + 1) & 3) + 1]];
            t[4 * c + 2] = SB[s[4 * ((c + 2) & 3) + 2]];
            t[4 * c + 3] = SB[s[4 * ((c + 3) & 3) + 3]];
        }
        if (r + 1 < ks->nr) {
            for (c = 0; c < 4; ++c) {          /* MixColumns (:221)           */
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1];
                uint8_t a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                uint8_t all = (uint8_t)(a0 ^ a1 ^ a2 ^ a3);
                s[4 * c + 0] = (uint8_t)(a0 ^ all ^ gf_double((uint8_t)(a0 ^ a1)));
                s[4 * c + 1] = (uint8_t)(a1 ^ all ^ gf_double((uint8_t)(a1 ^ a2)));
                s[4 * c + 2] = (uint8_t)(a2 ^ all ^ gf_double((uint8_t)(a2 ^ a3)));
                s[4 * c + 3] = (uint8_t)(a3 ^ all ^ gf_double((uint8_t)(a3 ^ a0)));
            }
        } else {
            memcpy(s, t, 16);
        }
    }
    xor16(s, ks->rk + 16 * ks->nr);
    memcpy(out, s, 16);
}

/* micro_aes.c:315-332: straightforward inverse cipher, same keys reversed */
void orc_decrypt_block(const orc_key *ks, const uint8_t in[16], uint8_t out[16])
{
    uint8_t s[16], t[16];
    int r, c;
    memcpy(s, in, 16);
    xor16(s, ks->rk + 16 * ks->nr);
    for (r = ks->nr - 1; r >= 0; --r) {
        /* InvShiftRows (:278) + InvSubBytes (:268): row k moves right by k */
        for (c = 0; c < 4; ++c) {
            t[4 * c + 0] = ISB[s[4 * c + 0]];
            t[4 * c + 1] = ISB[s[4 * ((c + 3) & 3) + 1]];
            t[4 * c + 2] = ISB[s[4 * ((c + 2) & 3) + 2]];
            t[4 * c + 3] = ISB[s[4 * ((c + 1) & 3) + 3]];
        }
        xor16(t, ks->rk + 16 * r);
        if (r > 0) {
            for (c = 0; c < 4; ++c) {          /* InvMixColumns (:301)        */
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1];
                uint8_t a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c + 0] = (uint8_t)(gf_mul(a0, 14) ^ gf_mul(a1, 11) ^ gf_mul(a2, 13) ^ gf_mul(a3, 9));
                s[4 * c + 1] = (uint8_t)(gf_mul(a0, 9) ^ gf_mul(a1, 14) ^ gf_mul(a2, 11) ^ gf_mul(a3, 13));
                s[4 * c + 2] = (uint8_t)(gf_mul(a0, 13) ^ gf_mul(a1, 9) ^ gf_mul(a2, 14) ^ gf_mul(a3, 11));
                s[4 * c + 3] = (uint8_t)(gf_mul(a0, 11) ^ gf_mul(a1, 13) ^ gf_mul(a2, 9) ^ gf_mul(a3, 14));
            }
        } else {
            memcpy(s, t, 16);
        }
    }
    memcpy(out, s, 16);
}

/* ------------------------------------------------------------------------ */
/* ECB -- micro_aes.c:636-680 (N1)                                            */
/* ------------------------------------------------------------------------ */
/* padding = the reference's compile-time AES_PADDING (micro_aes.h:79), padBlock
 * micro_aes.c:610-621: 0 -> a ragged tail is zero padded, nothing is added to whole
 * blocks; 1 -> PKCS#7: the 16 - rem missing bytes all hold the value 16 - rem;
 * 2 -> ISO/IEC 7816-4: one 0x80 byte, then zeros.  With 1 and 2 padBlock returns
 * true even for rem == 0, so a whole extra block is encrypted (:648-651).         */
void orc_ecb_encrypt_padded(int keybits, const uint8_t *key, int padding,
                            const void *pt, size_t len, void *ct)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)pt;
    uint8_t *y = (uint8_t *)ct;
    size_t n = len / 16, rem = len % 16, i;
    if (orc_setkey(&ks, key, keybits)) return;
    for (i = 0; i < n; ++i)
        orc_encrypt_block(&ks, x + 16 * i, y + 16 * i);
    if (rem || padding) {
        uint8_t last[16];
        memset(last, padding == 1 ? (int)(16 - rem) : 0, sizeof last);
        if (rem) memcpy(last, x + 16 * n, rem);
        if (padding == 2) last[rem] = 0x80;
        orc_encrypt_block(&ks, last, y + 16 * n);
    }
}

void orc_ecb_encrypt(int keybits, const uint8_t *key,
                     const void *pt, size_t len, void *ct)
{
    orc_ecb_encrypt_padded(keybits, key, 0, pt, len, ct);
}

char orc_ecb_decrypt(int keybits, const uint8_t *key,
                     const void *ct, size_t len, void *pt)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)ct;
    uint8_t *y = (uint8_t *)pt;
    size_t n = len / 16, rem = len % 16, i;
    if (orc_setkey(&ks, key, keybits)) return ORC_E_DECRYPT;
    for (i = 0; i < n; ++i)
        orc_decrypt_block(&ks, x + 16 * i, y + 16 * i);
    if (rem) {                                 /* reference copies the tail   */
        memmove(y + 16 * n, x + 16 * n, rem);  /* through untouched (:664)    */
        return ORC_E_DECRYPT;                  /* :679                        */
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* CTR -- micro_aes.c:919-990 with incBlock :421-427 (N2, N3)                 */
/* ------------------------------------------------------------------------ */
/* add `n` to the 56-bit big-endian counter in bytes 9..15; byte 8 and below
 * never change (the carry chain of incBlock stops after byte 9)             */
static void ctr56_add(uint8_t c[16], uint64_t n)
{
    uint64_t v = 0;
    int i;
    for (i = 9; i < 16; ++i) v = (v << 8) | c[i];
    v = (v + n) & 0x00FFFFFFFFFFFFFFull;
    for (i = 15; i >= 9; --i) { c[i] = (uint8_t)v; v >>= 8; }
}

static void ctr_stream(const orc_key *ks, uint8_t c[16],
                       const uint8_t *x, size_t len, uint8_t *y)
{
    uint8_t e[16];
    size_t n = len / 16, rem = len % 16, i, k;
    for (i = 0; i < n; ++i) {
        orc_encrypt_block(ks, c, e);
        for (k = 0; k < 16; ++k) y[16 * i + k] = (uint8_t)(x[16 * i + k] ^ e[k]);
        ctr56_add(c, 1);
    }
    if (rem) {                                 /* mixThenXor, :534-544        */
        orc_encrypt_block(ks, c, e);
        for (k = 0; k < rem; ++k) y[16 * n + k] = (uint8_t)(x[16 * n + k] ^ e[k]);
    }
}

void orc_ctr_xcrypt_at(int keybits, const uint8_t *key, const uint8_t ctr0[16],
                       uint64_t block_offset,
                       const void *in, size_t len, void *out)
{
    orc_key ks;
    uint8_t c[16];
    if (orc_setkey(&ks, key, keybits)) return;
    memcpy(c, ctr0, 16);
    ctr56_add(c, block_offset);
    ctr_stream(&ks, c, (const uint8_t *)in, len, (uint8_t *)out);
}

/* a build with other CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99): the counter block is
 * the IV's first iv_len bytes, zeros behind them, and the start value XORed in as a big-endian
 * integer ending at byte 15 (xorBEint :410-415: at least one byte, then while bytes remain)     */
void orc_ctr_encrypt_iv(int keybits, const uint8_t *key, const uint8_t *iv, size_t iv_len, uint64_t start,
                        const void *in, size_t len, void *out)
{
    uint8_t c[16] = { 0 };
    int pos = 15;
    memcpy(c, iv, iv_len > 16 ? 16 : iv_len);  /* CTR_IV_LENGTH, :968-971     */
    do c[pos--] ^= (uint8_t)start; while ((start >>= 8) != 0);
    orc_ctr_xcrypt_at(keybits, key, c, 0, in, len, out);
}

void orc_ctr_encrypt(int keybits, const uint8_t *key, const uint8_t *iv,
                     const void *in, size_t len, void *out)
{
    orc_ctr_encrypt_iv(keybits, key, iv, 12, 1, in, len, out);   /* CTR_IV_LENGTH 12, CTR_START_VALUE 1 */
}

/* ------------------------------------------------------------------------ */
/* XTS -- micro_aes.c:1008-1093, doubleLblock :449-458 (N5)                   */
/* ------------------------------------------------------------------------ */
static void xts_double(uint8_t t[16])
{
    int i;
    uint8_t carry = (uint8_t)(t[15] >> 7);
    for (i = 15; i > 0; --i) t[i] = (uint8_t)((t[i] << 1) | (t[i - 1] >> 7));
    t[0] = (uint8_t)((t[0] << 1) ^ (carry ? 0x87 : 0));
}

static void xex_block(const orc_key *k1, int enc, const uint8_t T[16], uint8_t *y)
{
    xor16(y, T);
    if (enc) orc_encrypt_block(k1, y, y); else orc_decrypt_block(k1, y, y);
    xor16(y, T);
}

static char xts_unit(int keybits, const uint8_t *keys, const uint8_t T0[16],
                     const void *in, size_t len, void *out, int enc)
{
    orc_key k1, k2;
    uint8_t T[16], *y = (uint8_t *)out;
    size_t rem = len % 16, n, i;
    if (len < 16) return ORC_E_DATALENGTH;     /* :1069, output untouched     */
    if (orc_setkey(&k1, keys, keybits)) return ORC_E_ENCRYPT;
    orc_setkey(&k2, keys + keybits / 8, keybits);
    if (out != in) memmove(out, in, len);
    n = len / 16 - (rem ? 1 : 0);
    orc_encrypt_block(&k2, T0, T);             /* :1026-1027                  */
    for (i = 0; i < n; ++i, y += 16) {
        xex_block(&k1, enc, T, y);
        xts_double(T);
    }
    if (rem) {                                 /* ciphertext stealing :1037   */
        uint8_t L[16], tmp[16];
        memcpy(L, T, 16);
        if (enc) xts_double(T); else xts_double(L);   /* swapped on decrypt   */
        xex_block(&k1, enc, L, y);
        memcpy(tmp, y, 16);
        memcpy(y, y + 16, rem);
        memcpy(y + 16, tmp, rem);
        xex_block(&k1, enc, T, y);
    }
    return ORC_OK;
}

static char xts_api(int keybits, const uint8_t *keys, const uint8_t *tweak,
                    const void *in, size_t len, void *out, int enc)
{
    uint8_t T0[16] = { 0 };                    /* NULL tweak == sector 0      */
    if (tweak) memcpy(T0, tweak, 16);
    return xts_unit(keybits, keys, T0, in, len, out, enc);
}

char orc_xts_encrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *pt, size_t len, void *ct)
{
    return xts_api(keybits, keys, tweak, pt, len, ct, 1);
}

char orc_xts_decrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *ct, size_t len, void *pt)
{
    return xts_api(keybits, keys, tweak, ct, len, pt, 0);
}

char orc_xts_sectors(int keybits, const uint8_t *keys, uint64_t first_sector,
                     size_t sector_bytes, size_t nsectors,
                     const void *in, void *out, int encrypt)
{
    size_t s;
    int b;
    for (s = 0; s < nsectors; ++s) {
        uint8_t T0[16] = { 0 };
        uint64_t id = first_sector + s;        /* copyLint, :399-404          */
        char rc;
        for (b = 0; b < 8; ++b) T0[b] = (uint8_t)(id >> (8 * b));
        rc = xts_unit(keybits, keys, T0,
                      (const uint8_t *)in + s * sector_bytes, sector_bytes,
                      (uint8_t *)out + s * sector_bytes, encrypt);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* GCM -- mulGF128 :476-493, xMac :551-570, gHash :1127-1137, API :1164-1212  */
/* ------------------------------------------------------------------------ */
void orc_gf128_mul(const uint8_t x[16], uint8_t y[16])
{
    uint8_t z[16] = { 0 }, v[16];
    int i, b, k;
    memcpy(v, y, 16);
    for (i = 0; i < 16; ++i) {
        for (b = 7; b >= 0; --b) {
            uint8_t lsb;
            if ((x[i] >> b) & 1) xor16(z, v);
            lsb = (uint8_t)(v[15] & 1);        /* v <- v * x  (:464-473)      */
            for (k = 15; k > 0; --k) v[k] = (uint8_t)((v[k] >> 1) | (v[k - 1] << 7));
            v[0] >>= 1;
            if (lsb) v[0] ^= 0xe1;
        }
    }
    memcpy(y, z, 16);
}

static void ghash_absorb(const uint8_t H[16], const uint8_t *x, size_t len,
                         uint8_t acc[16])
{
    size_t n = len / 16, rem = len % 16, i, k;
    for (i = 0; i < n; ++i) {
        xor16(acc, x + 16 * i);
        orc_gf128_mul(H, acc);
    }
    if (rem) {                                 /* zero padded partial block   */
        for (k = 0; k < rem; ++k) acc[k] ^= x[16 * n + k];
        orc_gf128_mul(H, acc);
    }
}

void orc_ghash(const uint8_t H[16], const void *aad, size_t aad_len,
               const void *ct, size_t ct_len, uint8_t gh[16])
{
    uint8_t lens[16];
    uint64_t abits = (uint64_t)aad_len * 8, cbits = (uint64_t)ct_len * 8;
    int i;
    for (i = 0; i < 8; ++i) {                  /* N6: two BE 64-bit lengths   */
        lens[7 - i] = (uint8_t)(abits >> (8 * i));
        lens[15 - i] = (uint8_t)(cbits >> (8 * i));
    }
    ghash_absorb(H, (const uint8_t *)aad, aad_len, gh);
    ghash_absorb(H, (const uint8_t *)ct, ct_len, gh);
    ghash_absorb(H, lens, 16, gh);
}

/* nonce_len = the reference's compile-time GCM_NONCE_LEN (micro_aes.h:108).  12 is the
 * default; any other length takes GCMsetup's first branch (:1145-1149): the initial
 * counter block is gHash(H, no AAD, the nonce) -- i.e. GHASH of the zero-padded nonce and
 * the length block [0]_64 || [8 * nonce_len]_64, as SP 800-38D prescribes -- and the
 * keystream then steps it with the SAME 56-bit incBlock as before (N2), not with inc32.   */
static void gcm_setup_iv(orc_key *ks, int keybits, const uint8_t *key,
                         const uint8_t *nonce, size_t nonce_len, uint8_t H[16], uint8_t J0[16])
{
    orc_setkey(ks, key, keybits);
    memset(H, 0, 16);
    orc_encrypt_block(ks, H, H);               /* :1144                       */
    if (nonce_len != 12) {
        memset(J0, 0, 16);
        orc_ghash(H, NULL, 0, nonce, nonce_len, J0);     /* :1147 */
        return;
    }
    memcpy(J0, nonce, 12);                     /* GCM_NONCE_LEN == 12, :1150  */
    J0[12] = J0[13] = J0[14] = 0;
    J0[15] = 1;
}

/* tag_len = the reference's compile-time GCM_TAG_LEN (micro_aes.h:109): the first tag_len bytes of the
 * tag are written (:1178) / compared (:1204)                                                        */
void orc_gcm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag)
{
    orc_key ks;
    uint8_t H[16], J0[16], c[16], G[16] = { 0 }, *out = (uint8_t *)ct_and_tag;
    gcm_setup_iv(&ks, keybits, key, nonce, nonce_len, H, J0);
    memcpy(c, J0, 16);
    ctr56_add(c, 1);                           /* N4: pre-increment, :939-941 */
    ctr_stream(&ks, c, (const uint8_t *)pt, len, out);
    orc_encrypt_block(&ks, J0, J0);            /* tag mask, :1173             */
    orc_ghash(H, aad, aad_len, out, len, G);
    xor16(G, J0);
    memcpy(out + len, G, tag_len);
}

void orc_gcm_encrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag)
{
    orc_gcm_encrypt_ex(keybits, key, nonce, nonce_len, 16, aad, aad_len, pt, len, ct_and_tag);
}

void orc_gcm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag)
{
    orc_gcm_encrypt_iv(keybits, key, nonce, 12, aad, aad_len, pt, len, ct_and_tag);
}

char orc_gcm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt)
{
    orc_key ks;
    uint8_t H[16], J0[16], E[16], c[16], G[16] = { 0 };
    const uint8_t *in = (const uint8_t *)ct_and_tag;
    gcm_setup_iv(&ks, keybits, key, nonce, nonce_len, H, J0);
    orc_ghash(H, aad, aad_len, in, len, G);    /* N7: authenticate first      */
    orc_encrypt_block(&ks, J0, E);
    xor16(G, E);
    if (memcmp(G, in + len, tag_len)) return ORC_E_AUTH;   /* pt left untouched */
    memcpy(c, J0, 16);
    ctr56_add(c, 1);
    ctr_stream(&ks, c, in, len, (uint8_t *)pt);
    return ORC_OK;
}

char orc_gcm_decrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt)
{
    return orc_gcm_decrypt_ex(keybits, key, nonce, nonce_len, 16, aad, aad_len, ct_and_tag, len, pt);
}

char orc_gcm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt)
{
    return orc_gcm_decrypt_iv(keybits, key, nonce, 12, aad, aad_len, ct_and_tag, len, pt);
}

/* ------------------------------------------------------------------------ */
/* CBC with CS3 ciphertext stealing -- micro_aes.c:697-782 (CTS 1)             */
/* ------------------------------------------------------------------------ */
char orc_cbc_encrypt(int keybits, const uint8_t *key, const uint8_t iv[16],
                     const void *pt, size_t len, void *ct)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)pt;
    uint8_t *y = (uint8_t *)ct, chain[16];
    size_t n = len / 16, r = len % 16, i;
    if (n > 1 && !r) { --n; r = 16; }          /* the last two blocks are always swapped */
    if (n == 0) return ORC_E_DATALENGTH;
    if (orc_setkey(&ks, key, keybits)) return ORC_E_ENCRYPT;
    memcpy(chain, iv, 16);
    for (i = 0; i < n; ++i) {
        uint8_t b[16];
        memcpy(b, x + 16 * i, 16);
        xor16(b, chain);
        orc_encrypt_block(&ks, b, chain);
        memcpy(y + 16 * i, chain, 16);
    }
    if (r) {
        uint8_t last[16] = { 0 };
        memcpy(last, x + 16 * n, r);           /* P_n, zero padded                       */
        memcpy(y + 16 * n, chain, r);          /* its slot gets the head of C_{n-1}      */
        xor16(last, chain);
        orc_encrypt_block(&ks, last, y + 16 * (n - 1));
    }
    return ORC_OK;
}

char orc_cbc_decrypt(int keybits, const uint8_t *key, const uint8_t iv[16],
                     const void *ct, size_t len, void *pt)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)ct;
    uint8_t *y = (uint8_t *)pt, prev[16], cur[16];
    size_t n = len / 16, r = len % 16, i;
    if (n > 1 && !r) { --n; r = 16; }
    if (n == 0) return ORC_E_DATALENGTH;
    if (r) --n;                                /* hold the last two blocks               */
    if (orc_setkey(&ks, key, keybits)) return ORC_E_DECRYPT;
    memcpy(prev, iv, 16);
    for (i = 0; i < n; ++i) {
        memcpy(cur, x + 16 * i, 16);
        orc_decrypt_block(&ks, cur, y + 16 * i);
        xor16(y + 16 * i, prev);
        memcpy(prev, cur, 16);
    }
    if (r) {
        uint8_t xx[16], z[16] = { 0 }, yy[16], c[16];
        size_t k;
        memcpy(xx, x + 16 * n, 16);
        memcpy(z, x + 16 * n + 16, r);
        orc_decrypt_block(&ks, xx, yy);        /* Y = Dec(X)                             */
        memcpy(c, yy, 16);
        memcpy(c, z, r);                       /* Z | tail of Y                          */
        for (k = 0; k < r; ++k) y[16 * n + 16 + k] = (uint8_t)(yy[k] ^ z[k]);
        orc_decrypt_block(&ks, c, c);
        xor16(c, prev);
        memcpy(y + 16 * n, c, 16);
    }
    return ORC_OK;
}

/* CBC of a reference build with CTS 0 (micro_aes.c:704-708, :727-733, :753-761): no stealing; the
 * last chunk is padded like ECB's (padBlock :610-621: zeros only if it is partial; PKCS#7 and
 * ISO 7816-4 ALWAYS append, so ct holds (len / 16 + 1) * 16 bytes then), no minimum length;
 * decryption wants whole blocks (:761) and does not strip the padding.  *out_len = bytes written. */
char orc_cbc_encrypt_nocts(int keybits, const uint8_t *key, const uint8_t iv[16], int padding,
                           const void *pt, size_t len, void *ct, size_t *out_len)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)pt;
    uint8_t *y = (uint8_t *)ct, chain[16];
    const size_t n = len / 16, r = len % 16;
    size_t i;
    if (orc_setkey(&ks, key, keybits)) return ORC_E_ENCRYPT;
    memcpy(chain, iv, 16);
    for (i = 0; i < n; ++i) {
        uint8_t b[16];
        memcpy(b, x + 16 * i, 16);
        xor16(b, chain);
        orc_encrypt_block(&ks, b, chain);
        memcpy(y + 16 * i, chain, 16);
    }
    if (r || padding) {
        uint8_t last[16] = { 0 };
        if (r) memcpy(last, x + 16 * n, r);
        if (padding == 1) memset(last + r, (int)(16 - r), 16 - r);
        else if (padding == 2) last[r] = 0x80;
        xor16(last, chain);
        orc_encrypt_block(&ks, last, y + 16 * n);
    }
    if (out_len) *out_len = 16 * (n + ((r || padding) ? 1 : 0));
    return ORC_OK;
}

char orc_cbc_decrypt_nocts(int keybits, const uint8_t *key, const uint8_t iv[16],
                           const void *ct, size_t len, void *pt)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)ct;
    uint8_t *y = (uint8_t *)pt, prev[16], cur[16];
    size_t i;
    if (len % 16) return ORC_E_DATALENGTH;
    if (orc_setkey(&ks, key, keybits)) return ORC_E_DECRYPT;
    memcpy(prev, iv, 16);
    for (i = 0; i < len / 16; ++i) {
        memcpy(cur, x + 16 * i, 16);
        orc_decrypt_block(&ks, cur, y + 16 * i);
        xor16(y + 16 * i, prev);
        memcpy(prev, cur, 16);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* CFB -- micro_aes.c:799-845;  OFB -- micro_aes.c:861-893                     */
/* ------------------------------------------------------------------------ */
void orc_cfb(int keybits, const uint8_t *key, const uint8_t iv[16], int encrypt,
             const void *in, size_t len, void *out)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)in;
    uint8_t *y = (uint8_t *)out, fb[16], e[16];
    size_t off, k;
    if (orc_setkey(&ks, key, keybits)) return;
    memcpy(fb, iv, 16);
    for (off = 0; off < len; off += 16) {
        size_t n = len - off < 16 ? len - off : 16;
        uint8_t c[16] = { 0 };
        orc_encrypt_block(&ks, fb, e);
        memcpy(c, x + off, n);                 /* input block (ciphertext when decrypting) */
        for (k = 0; k < n; ++k) y[off + k] = (uint8_t)(e[k] ^ c[k]);
        if (n == 16) memcpy(fb, encrypt ? y + off : c, 16);      /* feedback = ciphertext */
    }
}

void orc_ofb(int keybits, const uint8_t *key, const uint8_t iv[16],
             const void *in, size_t len, void *out)
{
    orc_key ks;
    const uint8_t *x = (const uint8_t *)in;
    uint8_t *y = (uint8_t *)out, o[16];
    size_t off, k;
    if (orc_setkey(&ks, key, keybits)) return;
    memcpy(o, iv, 16);
    for (off = 0; off < len; off += 16) {
        size_t n = len - off < 16 ? len - off : 16;
        orc_encrypt_block(&ks, o, o);
        for (k = 0; k < n; ++k) y[off + k] = (uint8_t)(x[off + k] ^ o[k]);
    }
}

/* ------------------------------------------------------------------------ */
/* CMAC -- micro_aes.c:1108-1118, cMac :576-590, getSubkeys :593-605,         */
/*         doubleBblock :434-444                                              */
/* ------------------------------------------------------------------------ */
static void double_be(uint8_t b[16])
{
    int i;
    uint8_t carry = (uint8_t)(b[0] >> 7);
    for (i = 0; i < 15; ++i) b[i] = (uint8_t)((b[i] << 1) | (b[i + 1] >> 7));
    b[15] = (uint8_t)((b[15] << 1) ^ (carry ? 0x87 : 0));
}

/* M <- Enc(M ^ X_i) over zero padded blocks (xMac with mix = rijndaelEncrypt) */
static void cbcmac_absorb(const orc_key *ks, const uint8_t *x, size_t len, uint8_t m[16])
{
    size_t off, k;
    for (off = 0; off < len; off += 16) {
        size_t n = len - off < 16 ? len - off : 16;
        for (k = 0; k < n; ++k) m[k] ^= x[off + k];
        orc_encrypt_block(ks, m, m);
    }
}

void orc_cmac(int keybits, const uint8_t *key, const void *data, size_t len, uint8_t mac[16])
{
    orc_key ks;
    uint8_t k1[16] = { 0 }, k2[16], last[16] = { 0 };
    const uint8_t *x = (const uint8_t *)data;
    size_t s = len ? (len - 1) % 16 + 1 : 0;
    if (orc_setkey(&ks, key, keybits)) return;
    orc_encrypt_block(&ks, k1, k1);
    double_be(k1);
    memcpy(k2, k1, 16);
    double_be(k2);
    memset(mac, 0, 16);
    cbcmac_absorb(&ks, x, len - s, mac);
    if (s) memcpy(last, x + len - s, s);
    if (s < 16) { last[s] ^= 0x80; xor16(last, k2); } else { xor16(last, k1); }
    xor16(mac, last);
    orc_encrypt_block(&ks, mac, mac);
}

/* ------------------------------------------------------------------------ */
/* CCM -- CCMtag :1222-1256, AES_CCM_encrypt/decrypt :1268-1314               */
/* 11-byte nonce (CCM_NONCE_LEN), 16-byte tag (CCM_TAG_LEN)                   */
/* ------------------------------------------------------------------------ */
static void be_xor(uint8_t *buf, size_t num, int pos)          /* xorBEint :410 */
{
    do buf[pos--] ^= (uint8_t)num; while (num >>= 8);
}

static void ccm_tag(const orc_key *ks, const uint8_t iv[16], size_t tag_len, const uint8_t *aad, size_t alen,
                    const uint8_t *pt, size_t plen, uint8_t m[16])
{
    uint8_t a[16] = { 0 }, e[16];
    size_t s = 0;
    int p = 1;
    memcpy(m, iv, 16);
    m[0] |= (uint8_t)((tag_len - 2) << 2);     /* :1229 */
    be_xor(m, plen, 15);
    if (alen) {
        m[0] |= 0x40;
        orc_encrypt_block(ks, m, m);
        if (alen > 0xFEFF) { p += 4; a[0] = 0xFF; a[1] = 0xFE; }
        be_xor(a, alen, p);
        ++p;
        s = (size_t)(16 - p);
        memcpy(a + p, aad, alen < s ? alen : s);
    }
    cbcmac_absorb(ks, a, 16, m);
    if (alen > s) cbcmac_absorb(ks, aad + s, alen - s, m);
    cbcmac_absorb(ks, pt, plen, m);
    orc_encrypt_block(ks, iv, e);
    xor16(m, e);
}

static void ccm_iv(const uint8_t *nonce, size_t nonce_len, uint8_t iv[16])
{
    memset(iv, 0, 16);
    iv[0] = (uint8_t)(14 - nonce_len);        /* :1273 */
    memcpy(iv + 1, nonce, nonce_len);
}

/* nonce_len / tag_len = the reference's compile-time CCM_NONCE_LEN (7..13) and CCM_TAG_LEN (even, 4..16),
 * micro_aes.h:103-104.  Whatever the nonce length, the keystream counter is stepped by the same 56-bit
 * incBlock (bytes 15..9, N2) and the text length is XORed in from byte 15 downwards (xorBEint, :1230).  */
void orc_ccm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag)
{
    orc_key ks;
    uint8_t iv[16], c[16], tag[16], *out = (uint8_t *)ct_and_tag;
    if (orc_setkey(&ks, key, keybits)) return;
    ccm_iv(nonce, nonce_len, iv);
    ccm_tag(&ks, iv, tag_len, (const uint8_t *)aad, aad_len, (const uint8_t *)pt, len, tag);
    memcpy(c, iv, 16);
    ctr56_add(c, 1);
    ctr_stream(&ks, c, (const uint8_t *)pt, len, out);
    memcpy(out + len, tag, tag_len);
}

void orc_ccm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag)
{
    orc_ccm_encrypt_ex(keybits, key, nonce, 11, 16, aad, aad_len, pt, len, ct_and_tag);
}

char orc_ccm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt)
{
    orc_key ks;
    uint8_t iv[16], c[16], tag[16];
    const uint8_t *in = (const uint8_t *)ct_and_tag;
    if (orc_setkey(&ks, key, keybits)) return ORC_E_DECRYPT;
    ccm_iv(nonce, nonce_len, iv);
    memcpy(c, iv, 16);
    ctr56_add(c, 1);
    ctr_stream(&ks, c, in, len, (uint8_t *)pt);        /* decrypt first (:1304) */
    ccm_tag(&ks, iv, tag_len, (const uint8_t *)aad, aad_len, (const uint8_t *)pt, len, tag);
    return memcmp(tag, in + len, tag_len) ? ORC_E_AUTH : ORC_OK;    /* text stays (SABOTAGE no-op) */
}

char orc_ccm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt)
{
    return orc_ccm_decrypt_ex(keybits, key, nonce, 11, 16, aad, aad_len, ct_and_tag, len, pt);
}

/* ------------------------------------------------------------------------ */
/* GCM-SIV -- RFC 8452; divideLblock/dotGF128 :499-529, polyval :1421-1432,    */
/*            GCM_SIVsetup :1435-1450, GCM_SIVtag :1453-1460, API :1473-1515   */
/* ------------------------------------------------------------------------ */
/* POLYVAL's field: blocks are 128-bit LITTLE-endian integers, bit j = x^j.
 * dot(a, b) = a * b * x^-128: walk the bits of a from x^127 down to x^0, dividing
 * b by x before each test (a right shift; a dropped x^0 re-enters as 0xe1<<120). */
static void polyval_dot(const uint8_t a[16], uint8_t b[16])
{
    uint64_t lo = 0, hi = 0, rlo = 0, rhi = 0;
    int i, j;
    for (i = 0; i < 8; ++i) { lo |= (uint64_t)b[i] << (8 * i); hi |= (uint64_t)b[8 + i] << (8 * i); }
    for (j = 127; j >= 0; --j) {
        const uint64_t drop = lo & 1;
        lo = (lo >> 1) | (hi << 63);
        hi >>= 1;
        if (drop) hi ^= (uint64_t)0xe1 << 56;
        if ((a[j >> 3] >> (j & 7)) & 1) { rlo ^= lo; rhi ^= hi; }
    }
    for (i = 0; i < 8; ++i) { b[i] = (uint8_t)(rlo >> (8 * i)); b[8 + i] = (uint8_t)(rhi >> (8 * i)); }
}

static void polyval_absorb(const uint8_t H[16], const uint8_t *x, size_t len, uint8_t acc[16])
{
    size_t off, k;
    for (off = 0; off < len; off += 16) {
        size_t n = len - off < 16 ? len - off : 16;
        for (k = 0; k < n; ++k) acc[k] ^= x[off + k];
        polyval_dot(H, acc);
    }
}

static void gcmsiv_keys(int keybits, const uint8_t *key, const uint8_t *nonce,
                        uint8_t auth[16], orc_key *enc)
{
    orc_key master;
    uint8_t blk[16], out[16], derived[48];
    int i, n = 2 + keybits / 64;
    orc_setkey(&master, key, keybits);
    for (i = 0; i < n; ++i) {
        memset(blk, 0, 16);
        blk[0] = (uint8_t)i;
        memcpy(blk + 4, nonce, 12);
        orc_encrypt_block(&master, blk, out);
        memcpy(derived + 8 * i, out, 8);
    }
    memcpy(auth, derived, 16);
    orc_setkey(enc, derived + 8 * n - keybits / 8, keybits);
}

static void gcmsiv_tag(const orc_key *enc, const uint8_t auth[16], const uint8_t *nonce,
                       const uint8_t *aad, size_t alen, const uint8_t *pt, size_t plen, uint8_t tag[16])
{
    uint8_t lens[16], s[16] = { 0 };
    uint64_t abits = (uint64_t)alen * 8, pbits = (uint64_t)plen * 8;
    int i;
    for (i = 0; i < 8; ++i) { lens[i] = (uint8_t)(abits >> (8 * i)); lens[8 + i] = (uint8_t)(pbits >> (8 * i)); }
    polyval_absorb(auth, aad, alen, s);
    polyval_absorb(auth, pt, plen, s);
    polyval_absorb(auth, lens, 16, s);
    for (i = 0; i < 12; ++i) s[i] ^= nonce[i];
    s[15] &= 0x7F;
    orc_encrypt_block(enc, s, tag);
}

/* CTR with a 32-bit little-endian counter in bytes 0..3 (SIVGCM_CTR, :935-938) */
static void ctr32le_stream(const orc_key *ks, const uint8_t tag[16], const uint8_t *x, size_t len, uint8_t *y)
{
    uint8_t c[16], e[16];
    size_t off, k;
    uint32_t ctr;
    memcpy(c, tag, 16);
    c[15] |= 0x80;
    ctr = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
    for (off = 0; off < len; off += 16, ++ctr) {
        size_t n = len - off < 16 ? len - off : 16;
        c[0] = (uint8_t)ctr; c[1] = (uint8_t)(ctr >> 8); c[2] = (uint8_t)(ctr >> 16); c[3] = (uint8_t)(ctr >> 24);
        orc_encrypt_block(ks, c, e);
        for (k = 0; k < n; ++k) y[off + k] = (uint8_t)(x[off + k] ^ e[k]);
    }
}

void orc_gcmsiv_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag)
{
    orc_key enc;
    uint8_t auth[16], tag[16], *out = (uint8_t *)ct_and_tag;
    gcmsiv_keys(keybits, key, nonce, auth, &enc);
    gcmsiv_tag(&enc, auth, nonce, (const uint8_t *)aad, aad_len, (const uint8_t *)pt, len, tag);
    ctr32le_stream(&enc, tag, (const uint8_t *)pt, len, out);
    memcpy(out + len, tag, 16);
}

char orc_gcmsiv_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt)
{
    orc_key enc;
    uint8_t auth[16], tag[16], given[16];
    const uint8_t *in = (const uint8_t *)ct_and_tag;
    gcmsiv_keys(keybits, key, nonce, auth, &enc);
    memcpy(given, in + len, 16);
    ctr32le_stream(&enc, given, in, len, (uint8_t *)pt);          /* decrypt first (:1500) */
    gcmsiv_tag(&enc, auth, nonce, (const uint8_t *)aad, aad_len, (const uint8_t *)pt, len, tag);
    return memcmp(tag, given, 16) ? ORC_E_AUTH : ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* OCB -- RFC 7253 as the reference fixes it (12-byte nonce, 16-byte tag):      */
/*        getDelta :1662-1680, OCB_cipher :1693-1762, API :1774-1811,          */
/*        doubleBblock :434-443                                               */
/* ------------------------------------------------------------------------ */
static void be_double(uint8_t b[16])
{
    int i, carry = b[0] >> 7;
    for (i = 0; i < 15; ++i) b[i] = (uint8_t)((b[i] << 1) | (b[i + 1] >> 7));
    b[15] = (uint8_t)((b[15] << 1) ^ (carry ? 0x87 : 0));
}

typedef struct {
    orc_key k;
    uint8_t Lstar[16], Ldollar[16], L[64][16];
} ocb_ctx;

static void ocb_setup(ocb_ctx *o, int keybits, const uint8_t *key)
{
    int j;
    orc_setkey(&o->k, key, keybits);
    memset(o->Lstar, 0, 16);
    orc_encrypt_block(&o->k, o->Lstar, o->Lstar);         /* L_* = Enc(0)          */
    memcpy(o->Ldollar, o->Lstar, 16); be_double(o->Ldollar);
    memcpy(o->L[0], o->Ldollar, 16);  be_double(o->L[0]);
    for (j = 1; j < 64; ++j) { memcpy(o->L[j], o->L[j - 1], 16); be_double(o->L[j]); }
}

static int ntz64(uint64_t i) { int n = 0; while (!(i & 1)) { i >>= 1; ++n; } return n; }

/* Offset_0 from the nonce: Ktop = Enc(0^7 1-padded nonce with the low 6 bits cleared),
 * Stretch = Ktop || (Ktop[0..7] ^ Ktop[1..8]), Offset_0 = Stretch[bottom .. bottom+127] */
static void ocb_offset0(const ocb_ctx *o, const uint8_t *nonce, size_t nonce_len, size_t tag_len, uint8_t off[16])
{
    uint8_t kt[24];
    int i, bottom = nonce[nonce_len - 1] & 63, sh = bottom & 7, by = bottom >> 3;     /* :1703 */
    memset(kt, 0, sizeof kt);
    memcpy(kt + 16 - nonce_len, nonce, nonce_len);                                    /* :1706 */
    kt[0] |= (uint8_t)(tag_len << 4);         /* :1707, taglen 128 mod 128 = 0 in the top 7 bits */
    kt[15 - nonce_len] |= 1;                  /* :1708 */
    kt[15] &= 0xC0;
    orc_encrypt_block(&o->k, kt, kt);
    for (i = 0; i < 8; ++i) kt[16 + i] = (uint8_t)(kt[i] ^ kt[i + 1]);
    for (i = 0; i < 16; ++i)
        off[i] = (uint8_t)(((kt[by + i] << 8 | kt[by + i + 1]) >> (8 - sh)) & 0xff);
}

static void ocb_hash(const ocb_ctx *o, const uint8_t *a, size_t alen, uint8_t sum[16])
{
    uint8_t off[16] = { 0 }, t[16];
    uint64_t i, m = alen / 16;
    size_t r = alen % 16;
    memset(sum, 0, 16);
    for (i = 1; i <= m; ++i, a += 16) {
        xor16(off, o->L[ntz64(i)]);
        memcpy(t, a, 16); xor16(t, off);
        orc_encrypt_block(&o->k, t, t);
        xor16(sum, t);
    }
    if (r) {
        xor16(off, o->Lstar);
        memset(t, 0, 16); memcpy(t, a, r); t[r] = 0x80;
        xor16(t, off);
        orc_encrypt_block(&o->k, t, t);
        xor16(sum, t);
    }
}

static void ocb_crypt(const ocb_ctx *o, const uint8_t *nonce, size_t nonce_len, size_t tag_len, int decrypt,
                      const uint8_t *aad, size_t alen, const uint8_t *in, size_t len,
                      uint8_t *out, uint8_t tag[16])
{
    uint8_t off[16], sum[16] = { 0 }, t[16], h[16];
    uint64_t i, m = len / 16;
    size_t r = len % 16, k;
    ocb_offset0(o, nonce, nonce_len, tag_len, off);
    for (i = 1; i <= m; ++i, in += 16, out += 16) {
        xor16(off, o->L[ntz64(i)]);
        memcpy(t, in, 16);
        if (!decrypt) xor16(sum, t);
        xor16(t, off);
        if (decrypt) orc_decrypt_block(&o->k, t, t); else orc_encrypt_block(&o->k, t, t);
        xor16(t, off);
        if (decrypt) xor16(sum, t);
        memcpy(out, t, 16);
    }
    if (r) {
        xor16(off, o->Lstar);
        orc_encrypt_block(&o->k, off, t);                  /* Pad = Enc(Offset_*)   */
        for (k = 0; k < r; ++k) {
            const uint8_t x = in[k], y = (uint8_t)(x ^ t[k]);
            sum[k] ^= decrypt ? y : x;
            out[k] = y;
        }
        sum[r] ^= 0x80;
    }
    xor16(sum, off);
    xor16(sum, o->Ldollar);
    orc_encrypt_block(&o->k, sum, tag);
    ocb_hash(o, aad, alen, h);
    xor16(tag, h);
}

/* nonce_len / tag_len = the reference's compile-time OCB_NONCE_LEN (1..15) and OCB_TAG_LEN (1..16),
 * micro_aes.h:115-116                                                                             */
void orc_ocb_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag)
{
    ocb_ctx o;
    uint8_t tag[16];
    ocb_setup(&o, keybits, key);
    ocb_crypt(&o, nonce, nonce_len, tag_len, 0, (const uint8_t *)aad, aad_len, (const uint8_t *)pt, len,
              (uint8_t *)ct_and_tag, tag);
    memcpy((uint8_t *)ct_and_tag + len, tag, tag_len);
}

void orc_ocb_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag)
{
    orc_ocb_encrypt_ex(keybits, key, nonce, 12, 16, aad, aad_len, pt, len, ct_and_tag);
}

/* like the reference, the text is decrypted before the tag is known to be good and stays
 * in pt on a mismatch (SABOTAGE is a no-op, :382)                                      */
char orc_ocb_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt)
{
    ocb_ctx o;
    uint8_t tag[16], given[16];
    memcpy(given, (const uint8_t *)ct_and_tag + len, tag_len);
    ocb_setup(&o, keybits, key);
    ocb_crypt(&o, nonce, nonce_len, tag_len, 1, (const uint8_t *)aad, aad_len, (const uint8_t *)ct_and_tag, len,
              (uint8_t *)pt, tag);
    return memcmp(tag, given, tag_len) ? ORC_E_AUTH : ORC_OK;
}

char orc_ocb_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt)
{
    return orc_ocb_decrypt_ex(keybits, key, nonce, 12, 16, aad, aad_len, ct_and_tag, len, pt);
}

/* ------------------------------------------------------------------------ */
/* Synthetic input of SURVEY.md section 8d                                    */
/* ------------------------------------------------------------------------ */
void orc_fill_splitmix(uint64_t seed, uint64_t word0, size_t nwords, void *dst)
{
    uint8_t *p = (uint8_t *)dst;
    size_t i;
    int b;
    for (i = 0; i < nwords; ++i) {
        uint64_t z = seed + (word0 + i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        for (b = 0; b < 8; ++b) p[8 * i + b] = (uint8_t)(z >> (8 * b));
    }
}
