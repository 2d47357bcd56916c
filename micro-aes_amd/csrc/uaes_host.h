/*
 * uaes_host.h -- the engine's own host data path (uaes_host.c): what uaes_engine.c calls when the deployer has
 * switched it on (uaes_set_host_policy, include/uaes_hip.h).  Plain C, host memory only, re-entrant; key schedules are
 * the engine's (little-endian words of the FIPS-197 byte stream: ek = encryption round keys, dk = the equivalent
 * inverse cipher's).  Every function states the reference function it follows (paths relative to the reference).
 */
#ifndef UAES_HOST_H_
#define UAES_HOST_H_

#include <stddef.h>
#include <stdint.h>

typedef struct {
    const uint32_t *ek, *dk;
    int nr;                                         /* 10 / 12 / 14 */
} uaesh_key;

/* once per process, from the engine's GF(2^8)-derived Te0 / Td0 */
void uaesh_tables_init(const uint32_t te0[256], const uint32_t td0[256]);

void uaesh_encrypt(const uint32_t *ek, int nr, const uint8_t in[16], uint8_t out[16]);          /* micro_aes.c:242-259 */
void uaesh_decrypt(const uint32_t *dk, int nr, const uint8_t in[16], uint8_t out[16]);          /* :315-332 */

void uaesh_ecb_encrypt(const uaesh_key *k, int padding, const uint8_t *in, size_t len, uint8_t *out);      /* :636-652 */
void uaesh_ecb_decrypt(const uaesh_key *k, const uint8_t *in, size_t len, uint8_t *out);                   /* :663-680 */
void uaesh_ctr(const uaesh_key *k, const uint8_t ctr0[16], uint64_t block_offset,
               const uint8_t *in, size_t len, uint8_t *out);                                                /* :919-950 */
void uaesh_xts_unit(const uaesh_key *k1, const uaesh_key *k2, int encrypt, const uint8_t tweak[16],
                    const uint8_t *in, size_t len, uint8_t *out);                                           /* :1008-1055 */
void uaesh_xts_sectors(const uaesh_key *k1, const uaesh_key *k2, int encrypt, uint64_t first_sector,
                       size_t sector_bytes, size_t nsectors, const uint8_t *in, uint8_t *out);
void uaesh_ghash(const uint8_t h[16], const uint8_t *aad, size_t aad_len, const uint8_t *ct, size_t ct_len,
                 uint8_t out[16]);                                                                          /* :1127-1137 */
int  uaesh_gcm(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
               const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out);            /* :1140-1212 */
int  uaesh_cbc_encrypt(const uaesh_key *k, const uint8_t iv[16], int cts, int padding,
                       const uint8_t *in, size_t len, uint8_t *out);                                        /* :697-744 */
int  uaesh_cbc_decrypt(const uaesh_key *k, const uint8_t iv[16], int cts,
                       const uint8_t *in, size_t len, uint8_t *out);                                        /* :746-782 */
void uaesh_cfb(const uaesh_key *k, const uint8_t iv[16], int encrypt, const uint8_t *in, size_t len, uint8_t *out);   /* :799-817 */
void uaesh_ofb(const uaesh_key *k, const uint8_t iv[16], const uint8_t *in, size_t len, uint8_t *out);     /* :861-885 */
void uaesh_cmac(const uaesh_key *k, const uint8_t *data, size_t len, uint8_t mac[16]);                      /* :1108-1118 */
int  uaesh_ccm(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
               const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out);            /* :1226-1314 */

/* expand = the engine's key schedule (uaes_expand_key): GCM-SIV derives its message key per nonce */
int  uaesh_gcmsiv(const uaesh_key *master, int keybits, int decrypt, const uint8_t nonce[12],
                  const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out,
                  int (*expand)(int keybits, const uint8_t *key, uint32_t ek[60], uint32_t dk[60]));      /* :1418-1516 */
int  uaesh_ocb(const uaesh_key *k, int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
               const uint8_t *aad, size_t aad_len, const uint8_t *in, size_t len, uint8_t *out);            /* :1693-1811 */

#endif
