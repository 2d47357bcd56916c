/*
 * uaes_device.h -- interface between the C host layer (uaes_engine.c) and the
 * HIP kernels (uaes_kernels.hip).  Plain C types only; every entry point is a
 * thin launcher that enqueues gfx950 kernels on the given stream and returns
 * the hipError_t value (0 = success).  Nothing here touches host data.
 */
#ifndef UAES_DEVICE_H_
#define UAES_DEVICE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Expanded key as little-endian words of the FIPS-197 byte stream.
 * ek = encryption round keys; dk = equivalent-inverse-cipher round keys
 * (dk[0] = ek[nr], dk[i] = InvMixColumns(ek[nr-i]), dk[nr] = ek[0]).      */
typedef struct {
    uint32_t w[60];
} uaesk_rk;

/* Read-only device tables uploaded once per context by the host layer. */
typedef struct {
    const uint32_t *te0;   /* 256 words: bytes {2S,S,S,3S}[x]              */
    const uint32_t *td0;   /* 256 words: bytes {14Si,9Si,13Si,11Si}[x] (their XOR is Si[x]) */
    /* GHASH: x -> x^(2^k) in GF(2^128) is GF(2)-linear with a matrix that depends on the field
     * only.  frob[(k-1)*256 + 2p .. +1] = the (hi, lo) input mask of output bit p (p < 64: bit p of
     * the big-endian high half, else bit p-64 of the low half), k = 1..63.  May be NULL (the
     * setup kernel then squares k times).                                                     */
    const uint64_t *frob;
} uaesk_tables;

/* 56-bit big-endian counter description (reference: incBlock with index 15
 * carries through bytes 15..9 only, micro_aes.c:421-427).                 */
typedef struct {
    uint32_t w0, w1;       /* counter-block bytes 0..7 as LE words         */
    uint32_t b8;           /* counter-block byte 8 (never changes)         */
    uint64_t v0;           /* bytes 9..15 as a 56-bit big-endian integer   */
    /* le32 != 0: the GCM-SIV flavour of CTR_cipher (SIVGCM_CTR, micro_aes.c:935-938):
     * a 32-bit LITTLE-endian counter in bytes 0..3 (w0, wraps mod 2^32), the other
     * twelve bytes (w1, w2, w3) fixed.  b8 / v0 are ignored.                 */
    uint32_t le32, w2, w3;
} uaesk_ctr;

int uaesk_device_info(int *cu_count, int *lds_bytes);

/* ECB: nfull whole blocks, plus (enc only) one padded tail block built from the
 * `rem` trailing bytes (reference N1, micro_aes.c:648-651).  padding = the
 * reference's AES_PADDING (padBlock, micro_aes.c:610-621): 0 zeros and only when
 * rem != 0, 1 PKCS#7, 2 ISO/IEC 7816-4 -- the latter two always add a block.  */
int uaesk_ecb(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *keys,
              int decrypt, const void *in, void *out, size_t nfull, unsigned rem, unsigned padding);

/* CTR keystream xor over len bytes (whole blocks + byte-granular tail).
 * If gate != NULL the kernel does nothing unless *gate == 0 (used by GCM
 * decrypt: authenticate first, micro_aes.c:1204-1208).                     */
int uaesk_ctr_xcrypt(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                     const uaesk_ctr *ctr, const void *in, void *out, size_t len,
                     const int *gate);
/* the generic CTR kernel with its key schedule and counter description in DEVICE memory (made by an earlier kernel
 * of the same stream) */
int uaesk_ctr_xcrypt_ind(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *d_rk,
                         const uaesk_ctr *d_ctr, const void *in, void *out, size_t len);

/* XTS over nsectors data units of sector_bytes each (>= 16).  Unit i has the
 * tweak block LE128(first_sector+i), or, if tweak16 != NULL (single unit),
 * the raw 16 bytes at tweak16 (a HOST pointer, copied into the launch).
 * scratch must hold uaesk_xts_scratch_bytes().                            */
size_t uaesk_xts_scratch_bytes(size_t sector_bytes, size_t nsectors);
int uaesk_xts(void *stream, const uaesk_tables *tb, int nr,
              const uaesk_rk *k1, const uaesk_rk *k2_enc, int decrypt,
              const uint8_t *tweak16, uint64_t first_sector,
              size_t sector_bytes, size_t nsectors,
              const void *in, void *out, void *scratch);

/* GCM.  All pointers are device pointers except j0_16 (host): the initial counter block J0 =
 * nonce || 00000001 for a 12-byte nonce, else the 16 bytes uaesk_gcm_j0 computed (GHASH of the
 * nonce, micro_aes.c:1145-1149).
 * scratch must hold uaesk_gcm_scratch_bytes().  encrypt: CTR then GHASH, tag
 * written at out+len.  decrypt: GHASH over in[0..len), compare with the tag
 * at in+len, *status = 0 / 0x1A, CTR gated on *status.                     */
size_t uaesk_gcm_scratch_bytes(void);
size_t uaesk_gcm_stream_scratch_bytes(void);     /* what the streamed API needs (no fused-pass buffers) */
int uaesk_gcm_j0(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                 const void *d_iv, size_t iv_len, void *scratch, void *j0_out16);
/* decrypt: 0 = encrypt; 1 = decrypt, nothing written unless the tag matches (two passes over the text);
 * 2 = decrypt, long texts in one pass: `out` is written before the tag is known and zeroed if it
 * turns out wrong (for callers that accept a wiped buffer, or whose `out` is private staging) */
int uaesk_gcm(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
              int decrypt, const uint8_t *j0_16,
              const void *aad, size_t aad_len,
              const void *in, size_t len, void *out,
              void *scratch, int *status);

/* decrypt == 3 (uaesk_gcm only): just the tag of (aad, `in` as ciphertext), 16 bytes written at `status`; the
 * host layer compares a truncated tag (GCM_TAG_LEN < 16) itself and then calls uaesk_gcm_ctr for the text */
int uaesk_gcm_ctr(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                  const uint8_t *j0_16, const void *in, size_t len, void *out);

/* Key context: uaesk_gcm_key_tables fills a scratch buffer (uaesk_gcm_scratch_bytes()) with every table
 * that depends on the key only; uaesk_gcm_keyed is uaesk_gcm on such a buffer -- per message it computes
 * just Enc(J0), unless the text needs a size-dependent bulk table.  One call at a time per buffer.  */
int uaesk_gcm_key_tables(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, void *key_scratch);
int uaesk_gcm_keyed(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                    int decrypt, const uint8_t *j0_16,
                    const void *aad, size_t aad_len,
                    const void *in, size_t len, void *out,
                    void *key_scratch, int *status);

/* Many short messages under one key context in one launch (k_gcm_records): record r = rec_len bytes at
 * in + r * in_stride under nonce nonces12 + 12 r with the AAD at aad + r * aad_stride (stride 0: shared);
 * encrypt writes text || tag at out + r * out_stride; decrypt reads text || tag, writes the text of the records
 * whose tag matches, verdicts[r] = 0 / 0x1A (may be NULL) and ORs 0x1A into *status (zero it first).
 * in, out and the strides are multiples of 16; rec_len <= uaesk_gcm_record_max(aad_len).  Reads the key
 * context's tables only: any number of these calls may share one context.                               */
size_t uaesk_gcm_record_max(size_t aad_len);
int uaesk_gcm_records(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int decrypt,
                      const void *nonces12, const void *aad, size_t aad_len, size_t aad_stride,
                      const void *in, size_t rec_len, size_t in_stride, void *out, size_t out_stride,
                      size_t nrec, const void *key_scratch, unsigned char *verdicts, int *status,
                      const void *lens /* NULL: every record rec_len bytes; else uint32 lens[nrec], record r = min(lens[r], rec_len) bytes */);

/* Sharded GCM: the weighted partial GHASH of one 16-byte-aligned ciphertext
 * shard (first shard: + AAD and Enc(J0); last shard: + length block).  The tag
 * of the whole message is the XOR of all shards' 16-byte results.           */
int uaesk_gcm_partial(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                      const uint8_t *nonce12, const void *aad, uint64_t total_aad_len,
                      const void *ct_shard, size_t shard_len, uint64_t shard_offset,
                      uint64_t total_len, void *scratch, void *partial16);
/* ... and the shard's CTR pass with it (one process driving several GPUs, uaes_mgpu_gcm_*): mode 0 encrypts
 * in -> out (keystream from J0 + 1 + shard_offset / 16) and hashes `out`; mode 1 = uaesk_gcm_partial (`out` unused);
 * mode 2 decrypts in -> out and hashes `in` -- `out` is written before any tag is known, the caller owns N7.
 * Every shard but the last is a multiple of 16 bytes.  A long shard runs CTR and GHASH in one pass.          */
int uaesk_gcm_shard(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int mode,
                    const uint8_t *nonce12, const void *aad, uint64_t total_aad_len,
                    const void *in, size_t shard_len, uint64_t shard_offset,
                    uint64_t total_len, void *out, void *scratch, void *partial16);

/* Streamed GCM: the running GHASH value stays in `scratch` (one
 * scratch buffer per stream).  absorb kind 0 = AAD (restarts the hash), 1 = a piece of
 * ciphertext (multiple of 16 bytes unless last), 2 = the length block.  tag:
 * write Y ^ Enc(J0) to tag_io, or compare with it (*status = 0 / 0x1A).       */
int uaesk_gcm_stream_absorb(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                            const uint8_t *nonce12, int kind, const void *data, size_t len,
                            uint64_t total_aad_len, uint64_t total_ct_len, void *scratch,
                            unsigned *plan_state);   /* per stream, starts at 0: which tables scratch holds */
/* one piece of the text, CTR + GHASH in one pass: the striped kernel when it is long enough, one launch of chunk
 * workgroups + finisher from 16 KiB (needs done_word: zero between calls); returns 1 when neither applies (run
 * uaesk_ctr_xcrypt + uaesk_gcm_stream_absorb instead); the scratch must hold uaesk_gcm_scratch_bytes() */
int uaesk_gcm_stream_piece(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
                           const uint8_t *nonce12, int decrypt, const void *in, size_t len,
                           uint64_t done_bytes, void *out, void *scratch, unsigned *plan_state, unsigned *done_word);
int uaesk_gcm_stream_tag(void *stream, void *scratch, int compare, void *tag_io, int *status);

/* short GCM-SIV message in one launch; -1 = not applicable (too long): take the general path */
int uaesk_gcmsiv_small(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *master_ek, int decrypt,
                       const uint8_t *nonce12,
                       const void *aad, size_t aad_len, const void *in, size_t len, void *out, int *status);

/* GCM-SIV of a message of any length without a host round trip: key derivation, key expansion, tag and counter stay
 * in `scratch` (uaesk_gcm_scratch_bytes()); *status (decrypt) = 0 / 0x1A, the plaintext is written either way */
int uaesk_gcmsiv_long(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *master_ek, int decrypt,
                      const uint8_t *nonce12, const void *aad, size_t aad_len,
                      const void *in, size_t len, void *out, void *scratch, int *status);

/* GHASH only: gh = GHASH_H(aad, ct) with H given (device), for tests.      */
int uaesk_ghash(void *stream, const uaesk_tables *tb, const uint8_t *H_host,
                const void *aad, size_t aad_len, const void *ct, size_t ct_len,
                void *scratch, void *gh_out16);

/* CMAC (AES_CMAC, micro_aes.c:1108): mac16 <- CMAC_K(data); all device pointers. */
int uaesk_cmac(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
               const void *data, size_t len, void *mac16);

/* CCM (AES_CCM_encrypt/decrypt, micro_aes.c:1268-1314), nonce (host) of nonce_len = 7..13 bytes,
 * tag_len (even, 4..16) bytes of tag at out+len / in+len; decrypt writes *status = 0 / 0x1A.   */
int uaesk_ccm(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek,
              int decrypt, const uint8_t *nonce, size_t nonce_len, size_t tag_len,
              const void *aad, size_t aad_len,
              const void *in, size_t len, void *out, int *status);

/* Feedback modes (micro_aes.c:697-893).  mode: 0 CBC encrypt (CS3 stealing),
 * 1 CBC decrypt, 2 CFB encrypt, 3 CFB decrypt, 4 OFB.  iv16 is a host pointer.
 * The block-parallel directions (1, 3) need in != out.                       */
int uaesk_feedback(void *stream, const uaesk_tables *tb, int nr,
                   const uaesk_rk *ek, const uaesk_rk *dk, int mode, const uint8_t *iv16,
                   const void *in, size_t len, void *out);

/* Batches of independent serial chains, one lane per message (all device pointers).
 * mac == 0: AES_CBC_encrypt (CS3) of nmsg messages of msg_bytes (multiple of 16) each, message m
 * at in + m*msg_bytes with IV ivs[m]; mac != 0: AES_CMAC of each message into out + 16 m.    */
int uaesk_chain_batch(void *stream, const uaesk_tables *tb, int nr, const uaesk_rk *ek, int mac,
                      const void *ivs, size_t nmsg, size_t msg_bytes, const void *in, void *out);

/* OCB (AES_OCB_encrypt/decrypt, micro_aes.c:1693-1811): nonce (host) of nonce_len = 1..15 bytes, tag_len
 * (1..16) bytes of tag at out+len (encrypt) / read at in+len (decrypt, *status = 0 / 0x1A; the text
 * is written either way, as in the reference).  dk = equivalent-inverse keys.  ONE launch; done_word = a device
 * word that is zero between calls and that nothing else writes (the launch's workgroups count themselves in on it
 * and the last one to arrive computes the tag and puts the zero back).  */
size_t uaesk_ocb_scratch_bytes(void);
int uaesk_ocb(void *stream, const uaesk_tables *tb, int nr,
              const uaesk_rk *ek, const uaesk_rk *dk, int decrypt, const uint8_t *nonce,
              size_t nonce_len, size_t tag_len,
              const void *aad, size_t aad_len, const void *in, size_t len, void *out,
              void *scratch, unsigned *done_word, int *status);

/* A ticket can also ride on the call's ONLY kernel (saves the second launch: 8.9 -> 7 us for an empty kernel,
 * tools/ubench/threadfloor.hip).  The host layer arms one for the calling thread right before a kernel-level call
 * whose last launch may take it (ECB, generic CTR, a one-launch XTS unit, a one-launch GCM encryption); the launcher
 * that does take it makes the kernel release `seq` to *flag when its last workgroup is done (d_count: a zeroed
 * device word for that count).  uaesk_ticket_disarm() = 1 if nobody took it (the host then sends k_ticket).   */
typedef struct {
    unsigned *flag;                 /* pinned host word the host spins on; NULL = no ticket */
    unsigned *count;                /* device word, zero between launches                    */
    unsigned  seq;
} uaesk_done;
void uaesk_ticket_arm(void *pinned_flag, void *d_count, unsigned seq);
/* a device word that is zero between calls and that nothing else writes, for the calling thread's NEXT kernel-level
 * call (uaesk_gcm / uaesk_gcm_keyed take it: with it a medium-sized text is one launch); NULL disarms */
void uaesk_done_word_arm(unsigned *w);
/* test hooks: the preparing workgroup's bounded look (100 MHz ticks; 0 = count in without looking) and the number of
 * folds a CHUNK workgroup has done on the current device (uaes_gcm.hip, "FOLD") */
void uaesk_debug_gcm_look(unsigned long long ticks);
int uaesk_debug_gcm_chunk_folds(unsigned *out);
int  uaesk_ticket_disarm(void);

/* Completion ticket of a synchronous call: a one-wave kernel behind the call's kernels copies nbytes (a multiple
 * of 4, <= 64) from d_src to pinned_dst and then stores seq to *pinned_flag with system-scope release.   */
int uaesk_ticket(void *stream, void *pinned_flag, unsigned seq, const void *d_src, void *pinned_dst, unsigned nbytes);

/* measurement: one wave spins for ticks_100mhz periods of the constant 100 MHz counter and writes
 * { shader cycles, reference ticks } (two 64-bit words) to d_out16                               */
int uaesk_clock_probe(void *stream, void *d_out16, unsigned long long ticks_100mhz);

/* Device self-test of the primitives; writes a bitmask of failures.        */
int uaesk_selftest(void *stream, const uaesk_tables *tb, const uaesk_rk *ek128,
                   const uaesk_rk *dk128, unsigned *d_result);

#ifdef __cplusplus
}
#endif
#endif
