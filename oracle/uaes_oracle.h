/*
 * uaes_oracle.h -- CPU oracle for the uAES hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the algorithms on the hot path of the
 * reference (polfosol/micro-AES v11): the Rijndael block primitive and the
 * ECB / CTR / XTS / GCM mode drivers, with the reference's exact edge
 * semantics (notes N1..N8).  It exists so that the HIP
 * engine can be checked bit-for-bit on a box that has no copy of the
 * reference.  Nothing in the product library links, loads or calls it: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this code against
 *   - the NIST CAVP files the reference's harness consumes
 *     (GcmEncryptExtIV{128,192,256}.rsp, XTSGenAES{128,256}.rsp,
 *     CMACGenAES{128,192,256}.rsp, VNT{128,192,256}.rsp; the same
 *     case filters as testvectors/aes_testvectors_GCM.h:86 and _XTS.h:84),
 *   - the known answers of the reference's main.c (main.c:16-34,49-50,58-60),
 *   - FIPS-197 appendix C,
 *   - digests produced by the compiled reference on the BASELINE configs
 *     (SURVEY.md section 8d), and, when oracle/_ref is built, the reference
 *     itself on random inputs.
 *
 * Unlike the reference (compile-time AES___ macro, micro_aes.h:17, and one
 * global RoundKey, micro_aes.c:72) the key size is a run-time argument and
 * all state lives on the caller's stack, so the oracle is re-entrant.
 */
#ifndef UAES_ORACLE_H_
#define UAES_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes, values as in micro_aes.h:469-476 */
#define ORC_OK            0
#define ORC_E_DATALENGTH  1
#define ORC_E_AUTH        0x1A
#define ORC_E_DECRYPT     0x1D
#define ORC_E_ENCRYPT     0x1E

typedef struct {
    int     nr;            /* 10 / 12 / 14 */
    uint8_t rk[15 * 16];   /* (nr+1) round keys, FIPS byte order */
} orc_key;

/* micro_aes.c:144-178 KeyExpansion.  keybits in {128,192,256}; returns 0 / -1 */
int  orc_setkey(orc_key *ks, const uint8_t *key, int keybits);
/* micro_aes.c:242-259 rijndaelEncrypt / :315-332 rijndaelDecrypt (aliasing ok) */
void orc_encrypt_block(const orc_key *ks, const uint8_t in[16], uint8_t out[16]);
void orc_decrypt_block(const orc_key *ks, const uint8_t in[16], uint8_t out[16]);

/* micro_aes.c:636-680.  encrypt writes ceil(len/16)*16 bytes (zero padding, N1) */
void orc_ecb_encrypt_padded(int keybits, const uint8_t *key, int padding,
                            const void *pt, size_t len, void *ct);
void orc_ecb_encrypt(int keybits, const uint8_t *key,
                     const void *pt, size_t len, void *ct);
char orc_ecb_decrypt(int keybits, const uint8_t *key,
                     const void *ct, size_t len, void *pt);

/* micro_aes.c:962-990.  iv = 12 bytes; counter = iv || 00000001 (N2, N3) */
void orc_ctr_encrypt(int keybits, const uint8_t *key, const uint8_t *iv,
                     const void *in, size_t len, void *out);
/* CTR_IV_LENGTH = iv_len (<= 16), CTR_START_VALUE = start (micro_aes.h:98-99, micro_aes.c:968-971) */
void orc_ctr_encrypt_iv(int keybits, const uint8_t *key, const uint8_t *iv, size_t iv_len, uint64_t start,
                        const void *in, size_t len, void *out);
/* extension used by sharded CTR: 16-byte initial counter block plus a block
 * offset that is added with the reference's 56-bit big-endian carry (N2)   */
void orc_ctr_xcrypt_at(int keybits, const uint8_t *key, const uint8_t ctr0[16],
                       uint64_t block_offset,
                       const void *in, size_t len, void *out);

/* micro_aes.c:1008-1093.  keys = key1 || key2; tweak = 16 raw bytes or NULL */
char orc_xts_encrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *pt, size_t len, void *ct);
char orc_xts_decrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *ct, size_t len, void *pt);
/* batched data units: unit i uses tweak LE128(first_sector + i), the
 * reference's own sectid convention (micro_aes.c:1017-1021)                */
char orc_xts_sectors(int keybits, const uint8_t *keys, uint64_t first_sector,
                     size_t sector_bytes, size_t nsectors,
                     const void *in, void *out, int encrypt);

/* micro_aes.c:476-493 mulGF128: y <- x*y in GCM's GF(2^128) */
void orc_gf128_mul(const uint8_t x[16], uint8_t y[16]);
/* micro_aes.c:1127-1137 gHash (result xored into gh, which starts at 0) */
void orc_ghash(const uint8_t H[16], const void *aad, size_t aad_len,
               const void *ct, size_t ct_len, uint8_t gh[16]);
/* micro_aes.c:1164-1212.  12-byte nonce, 16-byte tag appended at ct+len */
/* nonce_len: the reference's compile-time GCM_NONCE_LEN; != 12 -> J0 = GHASH(nonce), micro_aes.c:1145-1149 */
void orc_gcm_encrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag);
char orc_gcm_decrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt);
/* tag_len: the reference's compile-time GCM_TAG_LEN (micro_aes.h:109): tag_len bytes appended / compared */
void orc_gcm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag);
char orc_gcm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt);
void orc_gcm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag);
char orc_gcm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt);

/* micro_aes.c:697-782 CBC with CS3 ciphertext stealing (len < 16 -> 1) */
char orc_cbc_encrypt(int keybits, const uint8_t *key, const uint8_t iv[16],
                     const void *pt, size_t len, void *ct);
char orc_cbc_decrypt(int keybits, const uint8_t *key, const uint8_t iv[16],
                     const void *ct, size_t len, void *pt);
/* the same of a reference build with CTS 0 (micro_aes.c:704-733, :753-761); padding = AES_PADDING */
char orc_cbc_encrypt_nocts(int keybits, const uint8_t *key, const uint8_t iv[16], int padding,
                           const void *pt, size_t len, void *ct, size_t *out_len);
char orc_cbc_decrypt_nocts(int keybits, const uint8_t *key, const uint8_t iv[16],
                           const void *ct, size_t len, void *pt);
/* micro_aes.c:799-845 CFB (encrypt != 0 / decrypt), :861-893 OFB */
void orc_cfb(int keybits, const uint8_t *key, const uint8_t iv[16], int encrypt,
             const void *in, size_t len, void *out);
void orc_ofb(int keybits, const uint8_t *key, const uint8_t iv[16],
             const void *in, size_t len, void *out);

/* micro_aes.c:1108-1118 AES_CMAC */
void orc_cmac(int keybits, const uint8_t *key, const void *data, size_t len, uint8_t mac[16]);
/* micro_aes.c:1268-1314.  11-byte nonce, 16-byte tag; decrypt runs CTR first and
 * leaves the text in place on a tag mismatch (SABOTAGE is a no-op, :382)     */
void orc_ccm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag);
char orc_ccm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt);

/* nonce_len / tag_len: the reference's compile-time CCM_NONCE_LEN (7..13) / CCM_TAG_LEN (even, 4..16), micro_aes.h:103-104 */
void orc_ccm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag);
char orc_ccm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt);

/* micro_aes.c:1473-1515 GCM_SIV_encrypt/decrypt (RFC 8452); 12-byte nonce */
void orc_gcmsiv_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag);
char orc_gcmsiv_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt);

/* micro_aes.c:1774-1811 AES_OCB_encrypt/decrypt (RFC 7253); 12-byte nonce, 16-byte tag */
void orc_ocb_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *pt, size_t len, void *ct_and_tag);
char orc_ocb_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aad, size_t aad_len,
                     const void *ct_and_tag, size_t len, void *pt);

/* nonce_len / tag_len: the reference's compile-time OCB_NONCE_LEN (1..15) / OCB_TAG_LEN (1..16), micro_aes.h:115-116 */
void orc_ocb_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *pt, size_t len, void *ct_and_tag);
char orc_ocb_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
                        const void *aad, size_t aad_len,
                        const void *ct_and_tag, size_t len, void *pt);

/* SURVEY.md section 8d synthetic input: 64-bit LE word w of the stream is
 * splitmix64(seed + (w+1)*0x9E3779B97F4A7C15); fills [word0, word0+nwords) */
void orc_fill_splitmix(uint64_t seed, uint64_t word0, size_t nwords, void *dst);

#ifdef __cplusplus
}
#endif
#endif
