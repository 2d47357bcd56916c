/*
 * uaes_hip.h -- C ABI of the MI355X-native AES engine (libuaes_hip.so).
 *
 * This is the drop-in boundary for the hot path of polfosol/micro-AES: the
 * block-parallel mode drivers ECB / CTR / XTS / GCM.  Every entry point below
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  Differences from the reference API, all of them additive:
 *
 *   - the key size is a run-time argument (`keybits` = 128/192/256) instead of
 *     the compile-time macro AES___ (micro_aes.h:17).  include/micro_aes.h
 *     restores the macro API on top of these functions;
 *   - no global state: the reference keeps one static RoundKey
 *     (micro_aes.c:72) and is not re-entrant; this library is thread-safe AND concurrent:
 *     every host thread runs its synchronous calls on its own HIP stream with its own staging,
 *     so calls from different threads overlap on the GPU (no lock is held while it works);
 *   - data pointers may be HOST or DEVICE (HIP) pointers.  Host buffers are
 *     staged through device memory; device buffers are used in place.
 *     Keys, IVs, nonces and tweaks are always host pointers (<= 64 bytes);
 *   - `in == out` is allowed everywhere (the reference memcpy()s in -> out and
 *     works in place, micro_aes.h:520-526); partial overlap is undefined;
 *   - *_dev variants enqueue on a caller-supplied hipStream_t and return
 *     without synchronising: the form bench.py and multi-GPU sharding use.
 *
 * By DEFAULT every call runs on the GPU and there is no CPU fallback: if no HIP
 * device is usable every call fails loudly (UAES_E_HIP, message in
 * uaes_last_error()).  A deployer may switch on the engine's own host data path
 * for short host-pointer calls, single serial chains and GPU-less boxes:
 * uaes_set_host_policy() below -- opt-in, never silent.
 *
 * Return values: 0 on success; the reference's codes (micro_aes.h:469-476)
 * for the reference's error conditions; negative for engine failures.
 */
#ifndef UAES_HIP_H_
#define UAES_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAES_OK                0
#define UAES_E_DATALENGTH      1      /* M_DATALENGTH_ERROR     (0x1L)  */
#define UAES_E_AUTHENTICATION  0x1A   /* M_AUTHENTICATION_ERROR         */
#define UAES_E_DECRYPTION      0x1D   /* M_DECRYPTION_ERROR             */
#define UAES_E_ENCRYPTION      0x1E   /* M_ENCRYPTION_ERROR             */
#define UAES_E_HIP            (-1)    /* HIP runtime / device failure   */
#define UAES_E_ARG            (-2)    /* bad keybits / NULL / alignment */

/* ---- housekeeping ------------------------------------------------------ */
/* Initialise the engine on the calling thread's current HIP device (idempotent;
 * every other call does this lazily).                                        */
int         uaes_init(void);
/* Give back everything the library holds on every device: the per-thread lanes of the synchronous
 * API (stream, staging, pinned bounce buffers, scratch), the slice pipeline's streams and buffers,
 * the per-stream scratch of the *_dev API and the tables.  Waits for the devices' outstanding work.
 * No other call may be in flight; the next call after it sets the library up again.  Key contexts
 * and GCM streams are the caller's objects and are not touched.                               */
int         uaes_shutdown(void);
/* Run the on-device primitive self test (FIPS-197 C.1 both directions, byte
 * permute semantics, tweak arithmetic).  0 = pass, >0 = failure bitmask.     */
int         uaes_selftest(void);
/* Measurement aid: one wave on `stream` spins for spin_us microseconds of the constant 100 MHz counter and writes
 * { shader cycles elapsed, 100 MHz ticks elapsed } (two uint64) to d_out16 -- the shader clock the chip really runs
 * at while other streams load it (bench.py reports it beside the roofline).  Enqueue only.                    */
int         uaes_clock_probe_dev(void *d_out16, unsigned spin_us, void *stream);
/* Thread-local description of the last negative return value.               */
const char *uaes_last_error(void);
/* "uaes-hip <version> gfx950"                                               */
const char *uaes_version(void);
/* What CCM / GCM-SIV / OCB decryption leaves in pntxt when the tag is wrong.  These three
 * modes decrypt before they authenticate; the reference's default build then returns 0x1A
 * WITH the unauthenticated text in place (SABOTAGE is a no-op, micro_aes.c:1306-1312,
 * :1500-1511, :1803-1810) and that is this library's default too, so results are
 * bit-identical.  on != 0 selects the reference's INCREASE_SECURITY behaviour instead: the
 * buffer is zeroed.  Process-wide; returns the previous setting.  (The *_dev calls of those
 * three modes report through d_status and leave the decision to the caller.)
 * GCM never releases unauthenticated text (N7) and is not affected by this switch.               */
int         uaes_set_wipe_on_auth_failure(int on);
/* GCM decryption into a caller's DEVICE buffer reads the ciphertext twice by default (GHASH, tag check, then
 * CTR: 1.5x the traffic), because nothing may be written before the tag is known (N7, micro_aes.c:1204-1210).
 * on != 0: a long text is decrypted and hashed in ONE pass (8 % faster at 1 GiB) and, when the tag turns out
 * wrong, everything written is zeroed before 0x1A / a non-zero *d_status is reported -- between the launch and that
 * moment the buffer holds unauthenticated plaintext, which other streams could observe: that is what the caller
 * accepts by switching this on.  Process-wide; returns the previous setting.  Host-memory callers get the one-pass
 * kernel regardless: their plaintext is produced in a private staging buffer that is copied out only after the
 * tag has been verified.                                                                                   */
int         uaes_set_gcm_one_pass_decrypt(int on);
/* The *_dev calls keep a device scratch buffer (GHASH tables, XTS chunk tweaks) per
 * hipStream_t they have been used with (8 per device; beyond that the least recently
 * used one is recycled after a device-wide drain).  Call this when a stream will not
 * be used with the library any more -- before or after hipStreamDestroy -- to give
 * its buffer back.  Waits for the device's outstanding work.                       */
int         uaes_stream_release(void *stream);

/* The engine's own HOST data path (micro-aes_amd/csrc/uaes_host.c: portable table-driven C, re-entrant, bit-identical to
 * the kernels and parity-tested like them).  OFF by default -- max_bytes 0, chains 0, fallback 0: every call runs on the
 * GPU and fails loudly without one.  Three independent, process-wide switches for the cases a GPU serves badly:
 *   max_bytes  calls whose data pointers are HOST memory and whose text is at most this long run on the host: a
 *              launch costs 12-20 us whatever the size, one host core needs that long for ~1 KiB (GCM ~200 B);
 *   chains     ONE serial chain in host memory -- CBC / CFB encryption, OFB, CMAC, CCM -- runs on the host whatever
 *              its length (a chain is a latency-bound single wave on the GPU: 36 MiB/s; uaes_*_batch are the GPU's form);
 *   fallback   with NO usable HIP device the calls below run on the host instead of returning UAES_E_HIP -- the
 *              reference's `void` functions cannot report an error.
 * Covered: every synchronous one-message call of this header -- ECB, CTR, XTS (unit and sectors), GCM (any nonce / tag
 * length), CBC (CTS and CTS-0 forms), CFB, OFB, CMAC, CCM, GCM-SIV, OCB.  Not covered (always GPU): the *_dev / *_batch
 * / record / key-context / stream / mgpu calls, uaes_ghash, and any call that is handed a device pointer.
 * Environment, read at first use: UAES_HOST_MAX=<bytes>, UAES_HOST_CHAINS=1, UAES_HOST_FALLBACK=1; or
 * UAES_HOST_POLICY=recommended = (4096, 1, 1), the measured crossover on the builder's hosts (profiles/r05_host_policy.md),
 * which the three variables then refine.                                                                           */
int uaes_set_host_policy(size_t max_bytes, int chains, int fallback);
int uaes_get_host_policy(size_t *max_bytes, int *chains, int *fallback);

/* Host-side key schedule (KeyExpansion, micro_aes.c:144-178) as the kernels
 * receive it: (nr+1)*4 little-endian words of encryption round keys and of
 * the equivalent-inverse-cipher keys.  Returns nr (10/12/14) or a negative
 * error.  Diagnostic: needs no GPU.                                         */
int uaes_expand_key(int keybits, const uint8_t *key,
                    uint32_t enc_words[60], uint32_t dec_words[60]);

/* ---- ECB: replaces AES_ECB_encrypt / AES_ECB_decrypt --------------------
 * micro_aes.h:173-181, micro_aes.c:636-680.  encrypt writes ceil(len/16)*16
 * bytes (a trailing partial block is zero padded, AES_PADDING 0); decrypt
 * processes floor(len/16) blocks, copies a ragged tail through unchanged and
 * returns UAES_E_DECRYPTION if len % 16 != 0.                               */
int uaes_ecb_encrypt(int keybits, const uint8_t *key,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_ecb_decrypt(int keybits, const uint8_t *key,
                     const void *crtxt, size_t crtxtLen, void *pntxt);
/* padding = the reference's compile-time AES_PADDING (micro_aes.h:79; padBlock,
 * micro_aes.c:610-621): 0 as above; 1 PKCS#7 (n bytes of value n), 2 ISO/IEC 7816-4
 * (0x80, zeros).  1 and 2 always append a block: (ptextLen / 16 + 1) * 16 bytes are
 * written.  Decryption does not strip padding (nor does the reference's, :676-679). */
int uaes_ecb_encrypt_padded(int keybits, const uint8_t *key, int padding,
                            const void *pntxt, size_t ptextLen, void *crtxt);

/* ---- CTR: replaces AES_CTR_encrypt / AES_CTR_decrypt --------------------
 * micro_aes.h:256-266, micro_aes.c:962-990.  iv = 12 bytes (CTR_IV_LENGTH);
 * counter block = iv || 00000001 (CTR_START_VALUE); the counter is the
 * reference's 56-bit big-endian integer in bytes 9..15 (incBlock,
 * micro_aes.c:421-427).  Decrypt is the same function.                      */
int uaes_ctr_xcrypt(int keybits, const uint8_t *key, const uint8_t *iv,
                    const void *in, size_t len, void *out);
/* the same of a reference build with other CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99): counter block =
 * iv[0..ivLen) || zeros, startValue XORed in big-endian ending at byte 15 (micro_aes.c:968-971); ivLen <= 16     */
int uaes_ctr_xcrypt_iv(int keybits, const uint8_t *key, const uint8_t *iv, size_t ivLen, uint64_t startValue,
                       const void *in, size_t len, void *out);
/* Sharding extension: full 16-byte initial counter block (what the reference
 * builds internally, or takes directly when PRESET_COUNTER is 1,
 * micro_aes.h:100) plus a block offset added with the same 56-bit carry.
 * Rank g of a sharded stream passes block_offset = g * blocks_per_shard.    */
int uaes_ctr_xcrypt_at(int keybits, const uint8_t *key, const uint8_t ctr0[16],
                       uint64_t block_offset,
                       const void *in, size_t len, void *out);

/* ---- XTS: replaces AES_XTS_encrypt / AES_XTS_decrypt --------------------
 * micro_aes.h:239-249, micro_aes.c:1008-1093.  keys = key1 || key2
 * (2*keybits/8 bytes; key2 encrypts the tweak).  tweak = 16 raw bytes, or
 * NULL for data unit 0.  len < 16 -> UAES_E_DATALENGTH, output untouched.
 * Ciphertext stealing when len % 16 != 0.                                   */
int uaes_xts_encrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_xts_decrypt(int keybits, const uint8_t *keys, const uint8_t *tweak,
                     const void *crtxt, size_t crtxtLen, void *pntxt);
/* Batch extension: nsectors data units of sector_bytes each in one call; unit
 * i uses the tweak block LE128(first_sector + i) -- the reference's own
 * `sectid` convention (XTS_cipher, micro_aes.c:1017-1021).                  */
int uaes_xts_sectors(int keybits, const uint8_t *keys, uint64_t first_sector,
                     size_t sector_bytes, size_t nsectors,
                     const void *in, void *out, int encrypt);

/* ---- GCM: replaces AES_GCM_encrypt / AES_GCM_decrypt --------------------
 * micro_aes.h:294-308, micro_aes.c:1164-1212.  12-byte nonce (GCM_NONCE_LEN),
 * 16-byte tag (GCM_TAG_LEN) appended at crtxt + ptextLen.  decrypt takes
 * CT || tag with crtxtLen excluding the tag, authenticates BEFORE decrypting
 * and on mismatch returns UAES_E_AUTHENTICATION leaving pntxt untouched.     */
int uaes_gcm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_gcm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *crtxt, size_t crtxtLen, void *pntxt);
/* The same with a nonce of any length >= 1 -- the reference's compile-time GCM_NONCE_LEN
 * (micro_aes.h:108): 12 is the default above; otherwise J0 = GHASH_H(nonce) (GCMsetup,
 * micro_aes.c:1145-1149), and the keystream steps it with the reference's 56-bit incBlock.
 * J0 is computed on the GPU and read back (16 bytes), so these are synchronous calls.      */
int uaes_gcm_encrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen,
                        const void *aData, size_t aDataLen,
                        const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_gcm_decrypt_iv(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen,
                        const void *aData, size_t aDataLen,
                        const void *crtxt, size_t crtxtLen, void *pntxt);
/* ... and with the reference's compile-time GCM_TAG_LEN (micro_aes.h:109) as tagLen = 1..16: tagLen bytes of the
 * tag are appended at crtxt + ptextLen / compared (micro_aes.c:1178, :1204).  A truncated tag is compared on the
 * host (constant time) after the device has produced the full one; nothing is decrypted before that (N7).    */
int uaes_gcm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_gcm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *crtxt, size_t crtxtLen, void *pntxt);
/* GHASH_H(aData, crtxt) of micro_aes.c:1127-1137 with an explicit H (test
 * hook for the carry-less-multiply kernels); gh receives 16 bytes.           */
int uaes_ghash(const uint8_t H[16], const void *aData, size_t aDataLen,
               const void *crtxt, size_t crtxtLen, uint8_t gh[16]);

/* ---- CMAC: replaces AES_CMAC ----------------------------------------------
 * micro_aes.h (CMAC section), micro_aes.c:1108-1118.  A CBC-MAC chain is serial:
 * one GPU lane walks it (kept on the device so that no cipher code runs on the
 * host); ~36 MiB/s for one message (latency-bound); uaes_cmac_batch for many.          */
int uaes_cmac(int keybits, const uint8_t *key,
              const void *data, size_t dataSize, uint8_t mac[16]);

/* ---- CCM: replaces AES_CCM_encrypt / AES_CCM_decrypt ----------------------
 * micro_aes.c:1268-1314.  11-byte nonce (CCM_NONCE_LEN), 16-byte tag
 * (CCM_TAG_LEN) appended at crtxt + ptextLen.  Like the reference, decrypt
 * runs CTR first and authenticates the result: on a mismatch it returns
 * UAES_E_AUTHENTICATION and the (unauthenticated) text is left in pntxt.     */
int uaes_ccm_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_ccm_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *crtxt, size_t crtxtLen, void *pntxt);

/* The same with the reference's compile-time CCM_NONCE_LEN (7..13) and CCM_TAG_LEN (even, 4..16), micro_aes.h:103-104 */
int uaes_ccm_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_ccm_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *crtxt, size_t crtxtLen, void *pntxt);

/* ---- CBC / CFB / OFB: replace AES_CBC_*, AES_CFB_*, AES_OFB_* -------------
 * micro_aes.c:697-782 (CBC with CS3 ciphertext stealing, CTS 1: the last two
 * blocks are always swapped, len < 16 -> UAES_E_DATALENGTH), :799-845 (CFB),
 * :861-893 (OFB; decrypt is the same function).  iVec = 16 bytes.  The
 * decrypt directions of CBC and CFB are block-parallel kernels; the encrypt
 * directions and OFB are serial chains: one wave walks the chain, the sixteen lanes of a
 * DPP row share each block encryption (~0.42 us per AES-128 block, 36 MiB/s, latency-bound: a single
 * stream reaches 0.8x of the reference's CPU loop -- use the *_batch calls below when there
 * are many independent messages).                                                */
int uaes_cbc_encrypt(int keybits, const uint8_t *key, const uint8_t *iVec,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_cbc_decrypt(int keybits, const uint8_t *key, const uint8_t *iVec,
                     const void *crtxt, size_t crtxtLen, void *pntxt);
/* CBC as a reference build with CTS 0 does it (micro_aes.h:56; micro_aes.c:704-733, :753-761): no stealing, no
 * minimum length; the last chunk is padded like ECB's -- padding = AES_PADDING: 0 zeros behind a partial chunk,
 * 1 PKCS#7 / 2 ISO 7816-4 always append -- so crtxt receives 16 * (ptextLen / 16 + (ptextLen % 16 || padding))
 * bytes; decryption wants whole blocks (else UAES_E_DATALENGTH), is block-parallel, and leaves the padding.    */
int uaes_cbc_encrypt_padded(int keybits, const uint8_t *key, const uint8_t *iVec, int padding,
                            const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_cbc_decrypt_blocks(int keybits, const uint8_t *key, const uint8_t *iVec,
                            const void *crtxt, size_t crtxtLen, void *pntxt);
int uaes_cfb_encrypt(int keybits, const uint8_t *key, const uint8_t *iVec,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_cfb_decrypt(int keybits, const uint8_t *key, const uint8_t *iVec,
                     const void *crtxt, size_t crtxtLen, void *pntxt);
int uaes_ofb_xcrypt(int keybits, const uint8_t *key, const uint8_t *iVec,
                    const void *in, size_t len, void *out);

/* Batches of INDEPENDENT chains, sixteen GPU lanes per message (up to 81 919 messages) or one (above) -- where
 * a GPU serves the serial modes well.  nmsg messages of msg_bytes each, stored back to back (message m at
 * m * msg_bytes).  uaes_cbc_encrypt_batch: crtxt[m] = AES_CBC_encrypt(key, ivs + 16 m, message
 * m) bit for bit, CS3 swap included; msg_bytes must be a multiple of 16.  uaes_cmac_batch:
 * macs + 16 m = AES_CMAC(key, message m), any msg_bytes.  Data pointers host or device; ivs
 * host, or 16-byte aligned device memory.                                              */
int uaes_cbc_encrypt_batch(int keybits, const uint8_t *key, const uint8_t *ivs, size_t nmsg,
                           size_t msg_bytes, const void *pntxt, void *crtxt);
int uaes_cmac_batch(int keybits, const uint8_t *key, size_t nmsg, size_t msg_bytes,
                    const void *data, uint8_t *macs);

/* ---- GCM-SIV: replaces GCM_SIV_encrypt / GCM_SIV_decrypt ---------------------
 * RFC 8452; micro_aes.c:1418-1516.  12-byte nonce, 16-byte tag appended.  Per-
 * nonce keys are derived with AES (GCM_SIVsetup), the tag is AES over POLYVAL
 * (run through the GHASH kernels with byte-reversed blocks), the text is CTR
 * with a 32-bit little-endian counter seeded by the tag.  decrypt returns
 * UAES_E_AUTHENTICATION on a mismatch and, like the reference, leaves the
 * decrypted text in pntxt.                                                   */
int uaes_gcmsiv_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aData, size_t aDataLen,
                        const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_gcmsiv_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                        const void *aData, size_t aDataLen,
                        const void *crtxt, size_t crtxtLen, void *pntxt);

/* ---- OCB: replaces AES_OCB_encrypt / AES_OCB_decrypt --------------------------
 * RFC 7253; micro_aes.c:1693-1811.  12-byte nonce, 16-byte tag appended.  Block-
 * parallel in both directions: Offset_i is computed per lane from the Gray code of
 * i instead of being chained.  decrypt returns UAES_E_AUTHENTICATION on a mismatch
 * and, like the reference, leaves the decrypted text in pntxt.                  */
int uaes_ocb_encrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_ocb_decrypt(int keybits, const uint8_t *key, const uint8_t *nonce,
                     const void *aData, size_t aDataLen,
                     const void *crtxt, size_t crtxtLen, void *pntxt);

/* The same with the reference's compile-time OCB_NONCE_LEN (1..15) and OCB_TAG_LEN (1..16), micro_aes.h:115-116 */
int uaes_ocb_encrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_ocb_decrypt_ex(int keybits, const uint8_t *key, const uint8_t *nonce, size_t nonceLen, size_t tagLen,
                        const void *aData, size_t aDataLen,
                        const void *crtxt, size_t crtxtLen, void *pntxt);

/* ---- streamed GCM: one message fed in pieces -----------------
 * For texts larger than device (or host) memory: state = key schedule, block
 * position of the counter and the running GHASH value, which stays on the GPU
 * (Y <- Y * H^m ^ GHASH(piece), m = blocks in the piece).  The result is
 * bit-identical to one AES_GCM_encrypt / AES_GCM_decrypt call over the whole
 * text (micro_aes.c:1164-1212).  Every piece but the last must be a multiple of
 * 16 bytes; buffers may be host or device memory.  A decrypting stream releases
 * text before the tag is checked -- unlike AES_GCM_decrypt (N7) -- so the caller
 * must discard it if finish() returns UAES_E_AUTHENTICATION.  finish() and abort()
 * free the stream.                                                           */
typedef struct uaes_gcm_stream uaes_gcm_stream;
int  uaes_gcm_stream_begin(uaes_gcm_stream **s, int keybits, const uint8_t *key, const uint8_t *nonce,
                           const void *aData, size_t aDataLen, int decrypt);
int  uaes_gcm_stream_update(uaes_gcm_stream *s, const void *in, size_t len, void *out);
/* encrypting stream: tag <- the 16-byte tag; decrypting stream: tag = the received tag */
int  uaes_gcm_stream_finish(uaes_gcm_stream *s, uint8_t tag[16]);
void uaes_gcm_stream_abort(uaes_gcm_stream *s);

/* A synchronous call that is handed DEVICE memory first waits for the work the caller may have in
 * flight on the default stream (hipStreamSynchronize(NULL)), then runs on the calling thread's own
 * stream and returns when it is done -- the ordering a default-stream launch used to give.        */

/* ... or, when the caller's device data is produced on a stream of its own (one made with hipStreamNonBlocking does
 * not synchronise with the default stream), for the stream named here.  Thread-local; NULL = the default stream.  */
int uaes_set_producer_stream(void *stream);

/* ---- asynchronous, device-resident variants -----------------------------
 * All data pointers are device pointers, 16-byte aligned; `stream` is a
 * hipStream_t (NULL = default stream).  The call only enqueues work.  Scratch
 * (GHASH tables, XTS chunk tweaks, OCB offsets) is kept per stream, so calls on
 * different streams may overlap; up to 8 streams per device without a drain.
 * Calls that target the SAME stream must not be issued from two threads at once
 * (a call enqueues several dependent kernels).                                  */
int uaes_ecb_dev(int keybits, const uint8_t *key, int decrypt,
                 const void *d_in, size_t len, void *d_out, void *stream);
int uaes_ctr_xcrypt_at_dev(int keybits, const uint8_t *key, const uint8_t ctr0[16],
                           uint64_t block_offset,
                           const void *d_in, size_t len, void *d_out, void *stream);
int uaes_xts_sectors_dev(int keybits, const uint8_t *keys, uint64_t first_sector,
                         size_t sector_bytes, size_t nsectors,
                         const void *d_in, void *d_out, int encrypt, void *stream);
int uaes_gcm_encrypt_dev(int keybits, const uint8_t *key, const uint8_t *nonce,
                         const void *d_aad, size_t aad_len,
                         const void *d_in, size_t len, void *d_out, void *stream);
/* d_status (device int) receives 0 or UAES_E_AUTHENTICATION                 */
int uaes_gcm_decrypt_dev(int keybits, const uint8_t *key, const uint8_t *nonce,
                         const void *d_aad, size_t aad_len,
                         const void *d_in, size_t len, void *d_out,
                         int *d_status, void *stream);

/* decrypt != 0: d_in = CT || tag, d_status (device int) receives 0 or 0x1A    */
int uaes_ocb_dev(int keybits, const uint8_t *key, const uint8_t *nonce, int decrypt,
                 const void *d_aad, size_t aad_len,
                 const void *d_in, size_t len, void *d_out, int *d_status, void *stream);

/* ---- GCM key context -----------------------------------------------------------
 * The reference redoes GCMsetup in every call (micro_aes.c:1140-1152) and so do the one-shot
 * functions above: key schedule, H = Enc(0), the GHASH multiplication tables of H (a 20 us
 * kernel).  A caller that sends many messages under one key builds them ONCE:
 *   uaes_gcm_key_new    key schedule + every key-dependent table, on the current device
 *   uaes_gcm_key_encrypt / _decrypt        = uaes_gcm_encrypt / _decrypt, bit for bit (12-byte
 *                                            nonce; host or device pointers; synchronous)
 *   uaes_gcm_key_encrypt_dev / _decrypt_dev  enqueue on a stream
 * Per message only Enc(J0) is computed (a one-wave kernel).  The context holds per-message
 * state too: ONE call at a time per context (calls on one stream are ordered; use one context
 * per stream for concurrency).  uaes_gcm_key_free wipes the tables.                          */
typedef struct uaes_gcm_key uaes_gcm_key;
int  uaes_gcm_key_new(uaes_gcm_key **out, int keybits, const uint8_t *key);
void uaes_gcm_key_free(uaes_gcm_key *k);
int  uaes_gcm_key_encrypt(uaes_gcm_key *k, const uint8_t *nonce, const void *aData, size_t aDataLen,
                          const void *pntxt, size_t ptextLen, void *crtxt);
int  uaes_gcm_key_decrypt(uaes_gcm_key *k, const uint8_t *nonce, const void *aData, size_t aDataLen,
                          const void *crtxt, size_t crtxtLen, void *pntxt);
int  uaes_gcm_key_encrypt_dev(uaes_gcm_key *k, const uint8_t *nonce, const void *d_aad, size_t aad_len,
                              const void *d_in, size_t len, void *d_out, void *stream);
int  uaes_gcm_key_decrypt_dev(uaes_gcm_key *k, const uint8_t *nonce, const void *d_aad, size_t aad_len,
                              const void *d_in, size_t len, void *d_out, int *d_status, void *stream);

/* Many short messages under one key context in ONE launch (the GCM counterpart of uaes_xts_encrypt_sectors;
 * the reference has a message per call, AES_GCM_encrypt micro_aes.c:1164-1179, and a call costs ~12 us on a GPU
 * whatever its size).  Record r: rec_len bytes at in + r * in_stride, 12-byte nonce nonces + 12 r, AAD
 * aad + r * aad_stride (aad_stride 0: the same aad_len bytes for every record).
 *   encrypt: out + r * out_stride receives ciphertext || 16-byte tag   (out_stride >= rec_len + 16)
 *   decrypt: the record is ciphertext || tag (in_stride >= rec_len + 16), out + r * out_stride receives the
 *            plaintext of every record whose tag matches; the others are left untouched (N7), verdicts[r]
 *            (may be NULL) = 0 / 0x1A, and the call returns UAES_E_AUTHENTICATION if any record failed.
 * Byte for byte what uaes_gcm_key_encrypt / _decrypt give record by record.  Strides are multiples of 16,
 * in / out 16-byte aligned (device flavour), rec_len <= uaes_gcm_record_max(aad_len) (32 KiB - 48 without AAD).
 * The *_dev flavour takes device pointers for everything, enqueues on `stream`, and reports through
 * d_verdicts / *d_status (zeroed by the call, 0x1A ORed in); any number of record calls may share one context. */
size_t uaes_gcm_record_max(size_t aad_len);
int  uaes_gcm_key_encrypt_records(uaes_gcm_key *k, size_t nrec, const uint8_t *nonces,
                                  const void *aad, size_t aad_len, size_t aad_stride,
                                  const void *in, size_t rec_len, size_t in_stride, void *out, size_t out_stride);
int  uaes_gcm_key_decrypt_records(uaes_gcm_key *k, size_t nrec, const uint8_t *nonces,
                                  const void *aad, size_t aad_len, size_t aad_stride,
                                  const void *in, size_t rec_len, size_t in_stride, void *out, size_t out_stride,
                                  uint8_t *verdicts);
int  uaes_gcm_key_encrypt_records_dev(uaes_gcm_key *k, size_t nrec, const uint8_t *d_nonces,
                                      const void *d_aad, size_t aad_len, size_t aad_stride,
                                      const void *d_in, size_t rec_len, size_t in_stride,
                                      void *d_out, size_t out_stride, void *stream);
int  uaes_gcm_key_decrypt_records_dev(uaes_gcm_key *k, size_t nrec, const uint8_t *d_nonces,
                                      const void *d_aad, size_t aad_len, size_t aad_stride,
                                      const void *d_in, size_t rec_len, size_t in_stride,
                                      void *d_out, size_t out_stride, uint8_t *d_verdicts, int *d_status, void *stream);
/* ... with records of DIFFERENT lengths in slots of one size (packet buffers): record r is lens[r] <= max_len bytes at
 * the start of slot r (uint32 lengths; the device flavour clamps a longer one to max_len), its tag follows its own text;
 * strides >= max_len (+ 16 where the tag is).  The other bytes of a slot's first max_len + 16 output bytes are
 * unspecified afterwards.  The arrangement of the launch is the one the longest record needs.                   */
int  uaes_gcm_key_encrypt_records_v(uaes_gcm_key *k, size_t nrec, const uint8_t *nonces,
                                    const void *aad, size_t aad_len, size_t aad_stride,
                                    const void *in, const uint32_t *lens, size_t max_len, size_t in_stride,
                                    void *out, size_t out_stride);
int  uaes_gcm_key_decrypt_records_v(uaes_gcm_key *k, size_t nrec, const uint8_t *nonces,
                                    const void *aad, size_t aad_len, size_t aad_stride,
                                    const void *in, const uint32_t *lens, size_t max_len, size_t in_stride,
                                    void *out, size_t out_stride, uint8_t *verdicts);
int  uaes_gcm_key_encrypt_records_v_dev(uaes_gcm_key *k, size_t nrec, const uint8_t *d_nonces,
                                        const void *d_aad, size_t aad_len, size_t aad_stride,
                                        const void *d_in, const uint32_t *d_lens, size_t max_len, size_t in_stride,
                                        void *d_out, size_t out_stride, void *stream);
int  uaes_gcm_key_decrypt_records_v_dev(uaes_gcm_key *k, size_t nrec, const uint8_t *d_nonces,
                                        const void *d_aad, size_t aad_len, size_t aad_stride,
                                        const void *d_in, const uint32_t *d_lens, size_t max_len, size_t in_stride,
                                        void *d_out, size_t out_stride, uint8_t *d_verdicts, int *d_status, void *stream);

/* (The uaes_mgpu_* calls below start one worker thread per device ordinal the first time it is named; the workers live
 * until the process ends -- they hold nothing but their lane -- so a process that has made such a call must not
 * dlclose() the library.)                                                                                          */
/* ---- one process, several GPUs -------------------------------------------------
 * The text is cut into aligned slices, one per device; one host thread per device
 * runs the single-device call on its slice with the counter / sector offset
 * advanced (the 56-bit add of incBlock, micro_aes.c:421-427; the sectid convention
 * of XTS_cipher, :1017-1021), so the result is the single-device result.  devices
 * = ndev HIP device ordinals, or NULL for 0..ndev-1.  With host buffers every
 * slice travels over its own device's PCIe link.  Pointers may be host memory
 * or memory any of the devices can reach.                                      */
int uaes_mgpu_ctr_xcrypt_at(int ndev, const int *devices, int keybits, const uint8_t *key,
                            const uint8_t ctr0[16], uint64_t block_offset,
                            const void *in, size_t len, void *out);
int uaes_mgpu_xts_sectors(int ndev, const int *devices, int keybits, const uint8_t *keys,
                          uint64_t first_sector, size_t sector_bytes, size_t nsectors,
                          const void *in, void *out, int encrypt);

/* ECB over several GPUs (AES_ECB_encrypt / AES_ECB_decrypt, micro_aes.c:636-680: any partition of the blocks).  The
 * whole blocks are dealt out evenly; the last device also takes the ragged tail and the padding (padding = the
 * reference's AES_PADDING as in uaes_ecb_encrypt_padded), so the bytes written, and decryption's UAES_E_DECRYPTION
 * for a length that is no multiple of 16, are those of the one-device call.                                     */
int uaes_mgpu_ecb_encrypt(int ndev, const int *devices, int keybits, const uint8_t *key, int padding,
                          const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_mgpu_ecb_decrypt(int ndev, const int *devices, int keybits, const uint8_t *key,
                          const void *crtxt, size_t crtxtLen, void *pntxt);

/* GCM over several GPUs (AES_GCM_encrypt / AES_GCM_decrypt, micro_aes.c:1164-1212; gHash :1127-1137), the C host's
 * form of the sharded GCM described further down: device i runs CTR over its 16-byte aligned slice at keystream block
 * J0 + 1 + slice_start / 16 FUSED with the slice's weighted share of Enc(J0) ^ GHASH (one pass over the text; the
 * first slice carries aData and Enc(J0), the last one the length block); the host XORs the ndev 16-byte shares -- the
 * only exchange GCM needs, and it needs no collective in one process.  12-byte nonce, 16-byte tag at crtxt + ptextLen
 * / read at crtxt + crtxtLen, bit for bit what uaes_gcm_encrypt / _decrypt give on one device.
 * decrypt keeps N7 across devices: every device hashes its slice of the INPUT, the host compares the tag, and only
 * then is anything written; a forgery returns UAES_E_AUTHENTICATION with every slice of pntxt untouched.  (Host
 * buffers: each slice is decrypted inside a private device buffer during the hashing pass and copied out after the
 * verdict, so the text crosses each device's link once per direction.  Device buffers: two passes, or -- under
 * uaes_set_gcm_one_pass_decrypt -- one pass and zeroed slices on a forgery.)                                      */
int uaes_mgpu_gcm_encrypt(int ndev, const int *devices, int keybits, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, size_t aDataLen, const void *pntxt, size_t ptextLen, void *crtxt);
int uaes_mgpu_gcm_decrypt(int ndev, const int *devices, int keybits, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, size_t aDataLen, const void *crtxt, size_t crtxtLen, void *pntxt);

/* The same split WITHOUT a change in the caller: with a device list configured -- here, or by the environment
 * variable UAES_DEVICES=all | 0,1,2,... read at first use -- the synchronous uaes_ecb_* / uaes_ctr_xcrypt* /
 * uaes_xts_sectors / uaes_gcm_encrypt / uaes_gcm_decrypt calls (and so the AES_* symbols of include/micro_aes.h) hand
 * a text that lies in HOST memory and is at least min_bytes long (UAES_DEVICES_MIN_MIB, default 64 MiB) to the
 * uaes_mgpu_* functions above, every device staging its slice over its own PCIe link; the bytes produced are the
 * one-device call's.  ndev = 0 switches it off (the default), -1 = every visible device, devices NULL = 0..ndev-1,
 * min_bytes 0 = keep the threshold.  Device-pointer calls and short texts are never split.                       */
int uaes_set_devices(int ndev, const int *devices, size_t min_bytes);

/* BASELINE configs[4] in one call: the plaintext lies sharded over ndev GPUs (d_in[i] on devices[i] holds the slice
 * uaes_mgpu_ctr_xcrypt_at would give device i: blocks [B*i/ndev, B*(i+1)/ndev) of the B = ceil(len/16) blocks), every
 * device encrypts its slice with the counter advanced by its block offset (incBlock's 56-bit add, micro_aes.c:421-427;
 * CTR_cipher :943-949 -- the reference has no collective), and the ciphertext slices are gathered into
 * d_full_on_root (len bytes on devices[root]) by RCCL over xGMI: grouped ncclSend / ncclRecv, one per peer, each slice
 * over its own link.  d_out[i] = device i's own ciphertext shard buffer (required for every device other than the root's;
 * on the root's device NULL means "encrypt straight into place").  RCCL is loaded with dlopen on first use with ndev > 1 (librccl.so.1,
 * or $UAES_RCCL_LIB): without it the call fails with UAES_E_HIP and uaes_last_error() says why; ndev = 1 needs no RCCL.
 * The communicators of the last device list are cached for the life of the process (ncclCommInitAll costs seconds). */
int uaes_mgpu_ctr_encrypt_gather(int ndev, const int *devices, int keybits, const uint8_t *key,
                                 const uint8_t ctr0[16], uint64_t block_offset,
                                 const void *const *d_in, size_t len, void *const *d_out,
                                 int root, void *d_full_on_root);

/* Test hooks of the gather (the one-GPU evidence that the RCCL code runs; tests/test_gpu_robustness.py, bench.py
 * --force-collective).  Environment, read on every call:
 *   UAES_GATHER_FORCE_RCCL=1   slices on the root's OWN device that were encrypted into shard buffers (d_out[i] given)
 *                              also travel by ncclSend / ncclRecv -- rank r to rank r on the cached communicator, which
 *                              RCCL allows inside a group -- instead of hipMemcpy; ndev = 1 then loads RCCL too.
 *   UAES_GATHER_FAIL_SEND=k    the k-th ncclSend of the call names a peer the communicator does not have: RCCL itself
 *                              refuses it with the group open.  The call returns UAES_E_HIP, every gather stream is
 *                              drained before it returns, and the communicators are aborted and rebuilt on the next call.
 * out[0..4] = ncclSend calls accepted, ncclRecv calls accepted, groups opened, ncclCommInitAll runs, failed gathers --
 * since the library was loaded. */
void uaes_debug_gather_stats(unsigned long out[5]);

/* ---- the table of arrangements, as data ------------------------------------------------------------------------
 * Which kernels a call runs is decided in ONE place per mode (csrc/uaes_plan.h holds the table and the names; the
 * launchers switch on the same functions).  uaes_debug_plan() returns that decision without running anything:
 *   mode   0 ECB, 1 CTR, 2 XTS, 3 GCM, 4 OCB, 5 GCM-SIV
 *   dir    0 encrypt, 1 decrypt (GCM: tag first, N7), 2 GCM decrypt in one pass, 3 GCM tag only
 *   a, b   bytes of text, bytes of associated data (XTS: bytes per data unit, number of units)
 *   flags  bit 0 key context, bit 1 explicit XTS tweak, bit 2 no counter word (two-launch forms), bit 3 LE32 counter
 *   out    [0] arrangement id (uaes_debug_arrangement_name), [1] kernel launches, [2] workgroups of the main kernel,
 *          [3] GHASH positions per thread (chunk arrangements)
 * Works without a device (answers for a 256-CU MI355X).  uaes_debug_plan_disable(mask): arrangements whose bit
 * (1u << id) is set are passed over wherever another one can take the call (measurement and tests; environment
 * UAES_PLAN_DISABLE gives the initial mask).  tests/test_gpu_plan.py derives its parity cases from this table. */
int uaes_debug_plan(int mode, int dir, size_t a, size_t b, unsigned flags, int out[4]);
const char *uaes_debug_arrangement_name(int id);
void uaes_debug_plan_disable(unsigned mask);

/* Test hooks of the one-launch GCM / GCM-SIV / streamed-piece arrangements (chunk workgroups + one preparing workgroup
 * in ONE launch; whoever of them arrives last on a counter word folds the chunk hashes and makes the tag -- nobody
 * waits for anybody, DESIGN.md "GCM").  The preparing workgroup looks at the counter for a bounded time before it
 * counts itself in (default 1 ms; environment UAES_GCM_LOOK_TICKS): uaes_debug_gcm_look(0) makes it count in at
 * once, so that a chunk workgroup is usually the last and the fold runs THERE.  uaes_debug_gcm_chunk_folds: how many
 * folds a chunk workgroup has done on the current device.  UAES_GCM_FOLD=0 in the environment switches the
 * one-launch arrangements off altogether (chunks, then k_gcm_combine as a second launch). */
void uaes_debug_gcm_look(unsigned long long ticks_100mhz);
int uaes_debug_gcm_chunk_folds(unsigned *out);

/* ---- sharded GCM (multi-GPU) ------------------------------------------------
 * One message, cut into 16-byte aligned ciphertext shards, one per GPU.  Each
 * rank encrypts its shard with uaes_ctr_xcrypt_at_dev(ctr0 = nonce || 00000001,
 * block_offset = 1 + shard_offset/16) -- the CCM_GCM pre-increment of CTR_cipher
 * (micro_aes.c:939-941) -- and calls this function on the resulting ciphertext.
 * It returns 16 bytes: the shard's share of  Enc(J0) ^ GHASH(aData, crtxt)
 * (gHash, micro_aes.c:1127-1137; the first shard carries aData and Enc(J0), the
 * last one the length block).  The tag of AES_GCM_encrypt is the XOR of all
 * shards' shares -- a 16-byte-per-GPU exchange (RCCL all-gather), the only
 * collective GCM needs.  All data pointers are device pointers.               */
int uaes_gcm_partial_dev(int keybits, const uint8_t *key, const uint8_t *nonce,
                         const void *d_aad, uint64_t total_aad_len,
                         const void *d_ct_shard, size_t shard_len, uint64_t shard_offset,
                         uint64_t total_len, void *d_partial16, void *stream);

/* ... and the shard's CTR pass with it, in ONE pass over the text (what uaes_mgpu_gcm_* run per device): mode 0
 * encrypts d_in -> d_out (keystream block J0 + 1 + shard_offset / 16 onwards) and hashes d_out; mode 1 = the call
 * above (d_out unused); mode 2 decrypts d_in -> d_out and hashes d_in -- d_out is written before any tag is known: a
 * caller that must keep N7 (micro_aes.c:1200-1208) runs mode 1 on every shard first, compares the XOR of the shares
 * with the tag, and only then decrypts (uaes_ctr_xcrypt_at_dev at block offset 1 + shard_offset / 16).  Every shard
 * but the last is a multiple of 16 bytes; d_in == d_out is allowed.  Enqueue only.                               */
int uaes_gcm_shard_dev(int keybits, const uint8_t *key, const uint8_t *nonce, int mode,
                       const void *d_aad, uint64_t total_aad_len,
                       const void *d_in, size_t shard_len, uint64_t shard_offset, uint64_t total_len,
                       void *d_out, void *d_partial16, void *stream);

#ifdef __cplusplus
}
#endif
#endif
