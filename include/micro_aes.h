/*
 * micro_aes.h -- drop-in replacement for the reference's public header, for
 * the modes served by the MI355X engine (ECB, CTR, XTS, GCM on the
 * block-parallel hot path; CMAC and CCM for the remaining NIST .rsp files;
 * CBC, CFB, OFB with their parallel decrypt directions).
 *
 * Same function names, argument order, return types and configuration macros
 * as polfosol/micro-AES v11 (micro_aes.h:17-129, :173-181, :239-249, :256-266,
 * :294-308, :469-476), so that callers written against the reference -- its
 * main.c and its testvectors/ harness -- compile unchanged and link against
 * libmicro_aes_hip_<bits>.so instead of micro_aes.c.  Modes the engine does
 * not implement have their macro set to 0, which is how the reference itself
 * switches a mode (and the harness's test for it) off.
 *
 * The AES_* functions below are thin wrappers (uaes_compat.c) around the
 * run-time-key-size C ABI in uaes_hip.h; buffers may be host or HIP device
 * memory.  On an engine failure (no GPU, HIP error) the void functions print a
 * diagnostic and abort() -- they never return unencrypted data -- and the
 * char functions return M_ENCRYPTION_ERROR / M_DECRYPTION_ERROR.
 */
#ifndef MICRO_AES_H_
#define MICRO_AES_H_

#ifndef AES___
#define AES___ 128 /* or 192 / 256 (also settable with -DAES___=...); must match the library linked */
#endif

#define BLOCKCIPHERS 1
#define AEAD_MODES   1

#define ECB      1
#define CTR      1
#define CTR_NA   1
#define XEX      1
#define XTS      1
#define GCM      1
#define CMAC     1          /* serial CBC-MAC chains: one GPU lane (uaes_mac.hip) */
#define CCM      1
#define GCM_SIV  1          /* RFC 8452: POLYVAL through the GHASH kernels, LE32 counter */

#define CBC      1          /* feedback modes: decrypt of CBC/CFB is block-   */
#define CFB      1          /* parallel, the rest one GPU lane (uaes_chain.hip) */
#define OFB      1
#ifndef CTS                  /* micro_aes.h:56 -- CBC ciphertext stealing (CS3).  A caller built with -DCTS=0 gets   */
#define CTS      1          /* the reference's other CBC: last chunk padded (AES_PADDING), whole-block decrypt   */
#endif
#define KWA      0
#define FPE      0
#define EAX      0
#define EAXP     0
#define SIV      0
#define OCB      1          /* RFC 7253: block-parallel, offsets from the Gray code (uaes_ocb.hip) */
#define POLY1305 0
#define MICRO_RJNDL 0

#ifndef AES_PADDING          /* micro_aes.h:79 -- 0 zeros, 1 PKCS#7, 2 ISO/IEC 7816-4; a caller */
#define AES_PADDING     0   /* built with -DAES_PADDING=1|2 gets that ECB padding (see below)   */
#endif
#define DECRYPTION      1
#ifndef PRESET_COUNTER       /* micro_aes.h:100 -- a caller built with -DPRESET_COUNTER=1 hands AES_CTR_* the    */
#define PRESET_COUNTER  0   /* whole 16-byte counter block instead of a 12-byte IV (micro_aes.c:965-966)     */
#endif

/* The reference fixes these lengths as enum constants that a user edits in its header (micro_aes.h:103-116).  Here a
 * caller built with -DCCM_NONCE_LEN=n (7..13), -DCCM_TAG_LEN=n (even, 4..16), -DGCM_NONCE_LEN=n (>= 1),
 * -DGCM_TAG_LEN=n (1..16), -DOCB_NONCE_LEN=n (1..15) or -DOCB_TAG_LEN=n (1..16) gets that build: the macro stands in
 * for the enum constant and the AES_CCM_* / AES_GCM_* / AES_OCB_* names bind to the general entry points below.    */
enum constant_parameters_of_modes
{
#ifndef CTR_START_VALUE    /* micro_aes.h:98-99; -DCTR_START_VALUE=n / -DCTR_IV_LENGTH=n (<= 16) bind AES_CTR_* to  */
    CTR_START_VALUE = 1,   /* the general entry point below (micro_aes.c:968-971)                                */
#endif
#ifndef CTR_IV_LENGTH
    CTR_IV_LENGTH   = 12,
#endif
#ifndef CCM_NONCE_LEN
    CCM_NONCE_LEN   = 11,
#endif
#ifndef CCM_TAG_LEN
    CCM_TAG_LEN     = 16,
#endif
#ifndef GCM_NONCE_LEN      /* micro_aes.h:108; any value but 12: J0 = GHASH(nonce) (micro_aes.c:1145-1149) */
    GCM_NONCE_LEN   = 12,
#endif
#ifndef GCM_TAG_LEN
    GCM_TAG_LEN     = 16,
#endif
#ifndef OCB_NONCE_LEN
    OCB_NONCE_LEN   = 12,
#endif
#ifndef OCB_TAG_LEN
    OCB_TAG_LEN     = 16,
#endif
    SIVGCM_NONCE_LEN = 12,
    SIVGCM_TAG_LEN  = 16,
#if AES___ == 256 || AES___ == 192
    AES_KEYLENGTH   = AES___ / 8
#else
    AES_KEYLENGTH   = 16
#endif
};
typedef char uaes_ccm_lengths_ok[(CCM_NONCE_LEN >= 7 && CCM_NONCE_LEN <= 13 && CCM_TAG_LEN >= 4 && CCM_TAG_LEN <= 16 &&
                                  CCM_TAG_LEN % 2 == 0) ? 1 : -1];
typedef char uaes_ctr_lengths_ok[(CTR_IV_LENGTH >= 0 && CTR_IV_LENGTH <= 16) ? 1 : -1];
typedef char uaes_gcm_lengths_ok[(GCM_NONCE_LEN >= 1 && GCM_TAG_LEN >= 1 && GCM_TAG_LEN <= 16) ? 1 : -1];
typedef char uaes_ocb_lengths_ok[(OCB_NONCE_LEN >= 1 && OCB_NONCE_LEN <= 15 && OCB_TAG_LEN >= 1 && OCB_TAG_LEN <= 16) ? 1 : -1];

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* real functions behind OBJECT-like macros, so that `&AES_GCM_encrypt` / `&AES_CTR_encrypt` or a dispatch-table entry binds to
 * the lengths of this build as well as a direct call does                                              */
#if defined(__cplusplus) || (defined(__STDC_VERSION__) && __STDC_VERSION__ >= 199901L)
#define UAES_STATIC_INLINE static inline
#elif defined(__GNUC__)
#define UAES_STATIC_INLINE static __inline__ __attribute__((unused))
#else
#define UAES_STATIC_INLINE static
#endif

void AES_ECB_encrypt(const uint8_t *key,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
/* the same with padBlock's other two paddings (micro_aes.c:610-621): both ALWAYS append a
 * block, so crtxt must hold (ptextLen / 16 + 1) * 16 bytes                              */
void AES_ECB_encrypt_pkcs7(const uint8_t *key,
                           const void *pntxt, const size_t ptextLen, void *crtxt);
void AES_ECB_encrypt_iso7816(const uint8_t *key,
                             const void *pntxt, const size_t ptextLen, void *crtxt);
#if AES_PADDING == 1
#define AES_ECB_encrypt AES_ECB_encrypt_pkcs7
#elif AES_PADDING == 2
#define AES_ECB_encrypt AES_ECB_encrypt_iso7816
#endif
char AES_ECB_decrypt(const uint8_t *key,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

void AES_CTR_encrypt(const uint8_t *key, const uint8_t *iv,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
void AES_CTR_decrypt(const uint8_t *key, const uint8_t *iv,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);
/* PRESET_COUNTER is a compile-time switch of the CALLER's build in the reference: `iv` is then the
 * pre-initialised 16-byte counter block (bytes 9..15 = the 56-bit big-endian counter of incBlock,
 * micro_aes.c:421-427) -- the form in which a shard of a longer stream is encrypted.  Every library
 * exports both forms; a caller built with -DPRESET_COUNTER=1 is bound to these:                 */
void AES_CTR_encrypt_preset(const uint8_t *key, const uint8_t *counter16,
                            const void *pntxt, const size_t ptextLen, void *crtxt);
void AES_CTR_decrypt_preset(const uint8_t *key, const uint8_t *counter16,
                            const void *crtxt, const size_t crtxtLen, void *pntxt);
/* CTR_IV_LENGTH / CTR_START_VALUE other than 12 / 1 (micro_aes.h:98-99): counter block = iv[0..CTR_IV_LENGTH) ||
 * zeros with CTR_START_VALUE XORed in big-endian at its end (micro_aes.c:968-971).  Decrypt = encrypt (:986-990). */
void AES_CTR_encrypt_iv(const size_t ivLen, const size_t startValue, const uint8_t *key, const uint8_t *iv,
                        const void *pntxt, const size_t ptextLen, void *crtxt);
#if PRESET_COUNTER
#define AES_CTR_encrypt AES_CTR_encrypt_preset
#define AES_CTR_decrypt AES_CTR_decrypt_preset
#elif defined(CTR_IV_LENGTH) || defined(CTR_START_VALUE)
UAES_STATIC_INLINE void AES_CTR_encrypt_nl(const uint8_t *key, const uint8_t *iv,
                                        const void *pntxt, const size_t ptextLen, void *crtxt)
{
    AES_CTR_encrypt_iv(CTR_IV_LENGTH, CTR_START_VALUE, key, iv, pntxt, ptextLen, crtxt);
}
#define AES_CTR_encrypt AES_CTR_encrypt_nl
#define AES_CTR_decrypt AES_CTR_encrypt_nl
#endif

char AES_XTS_encrypt(const uint8_t *keys, const uint8_t *tweak,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_XTS_decrypt(const uint8_t *keys, const uint8_t *tweak,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

void AES_GCM_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_GCM_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);
/* GCM_NONCE_LEN / GCM_TAG_LEN are compile-time constants of the CALLER's build in the reference; with any nonce
 * length but 12 GCMsetup derives J0 = GHASH(nonce) (micro_aes.c:1145-1149), and GCM_TAG_LEN bytes of the tag are
 * appended / compared (:1178, :1204).  Every library exports the general entry points and a caller built with
 * -DGCM_NONCE_LEN=n and / or -DGCM_TAG_LEN=n is bound to them:                                                */
void AES_GCM_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_GCM_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt);
/* (the round-2 names: the same with a 16-byte tag) */
void AES_GCM_encrypt_ivlen(const size_t nonceLen, const uint8_t *key, const uint8_t *nonce,
                           const void *aData, const size_t aDataLen,
                           const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_GCM_decrypt_ivlen(const size_t nonceLen, const uint8_t *key, const uint8_t *nonce,
                           const void *aData, const size_t aDataLen,
                           const void *crtxt, const size_t crtxtLen, void *pntxt);

#if defined(GCM_NONCE_LEN) || defined(GCM_TAG_LEN)
UAES_STATIC_INLINE void AES_GCM_encrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *pntxt, const size_t ptextLen, void *crtxt)
{
    AES_GCM_encrypt_lens(GCM_NONCE_LEN, GCM_TAG_LEN, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt);
}
UAES_STATIC_INLINE char AES_GCM_decrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return AES_GCM_decrypt_lens(GCM_NONCE_LEN, GCM_TAG_LEN, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt);
}
#define AES_GCM_encrypt AES_GCM_encrypt_nl
#define AES_GCM_decrypt AES_GCM_decrypt_nl
#endif

char AES_CBC_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CBC_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt);
/* CTS 0 (micro_aes.h:56): no stealing -- AES_CBC_encrypt pads its last chunk with padBlock like ECB (micro_aes.c:
 * 727-733; zeros behind a partial chunk, or PKCS#7 / ISO 7816-4 which always append: crtxt then holds
 * (ptextLen / 16 + 1) * 16 bytes), any length is accepted (:704-708), and AES_CBC_decrypt wants whole blocks
 * (:761, M_DATALENGTH_ERROR otherwise) and leaves the padding in place.  Every library exports both families:    */
char AES_CBC_encrypt_nocts(const uint8_t *key, const uint8_t iVec[16],
                           const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CBC_encrypt_nocts_pkcs7(const uint8_t *key, const uint8_t iVec[16],
                                 const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CBC_encrypt_nocts_iso7816(const uint8_t *key, const uint8_t iVec[16],
                                   const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CBC_decrypt_nocts(const uint8_t *key, const uint8_t iVec[16],
                           const void *crtxt, const size_t crtxtLen, void *pntxt);
#if !CTS
#if AES_PADDING == 1
#define AES_CBC_encrypt AES_CBC_encrypt_nocts_pkcs7
#elif AES_PADDING == 2
#define AES_CBC_encrypt AES_CBC_encrypt_nocts_iso7816
#else
#define AES_CBC_encrypt AES_CBC_encrypt_nocts
#endif
#define AES_CBC_decrypt AES_CBC_decrypt_nocts
#endif

void AES_CFB_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt);
void AES_CFB_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

void AES_OFB_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt);
void AES_OFB_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

void AES_CCM_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CCM_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

/* CCM_NONCE_LEN (7..13) / CCM_TAG_LEN (even, 4..16), micro_aes.h:103-104: iv = { 14 - N, nonce, 0.. } (:1273), the
 * flags byte carries (T - 2) << 2 (:1229), T bytes of tag are appended / compared (:1281, :1308)                 */
void AES_CCM_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_CCM_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt);
#if defined(CCM_NONCE_LEN) || defined(CCM_TAG_LEN)
UAES_STATIC_INLINE void AES_CCM_encrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *pntxt, const size_t ptextLen, void *crtxt)
{
    AES_CCM_encrypt_lens(CCM_NONCE_LEN, CCM_TAG_LEN, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt);
}
UAES_STATIC_INLINE char AES_CCM_decrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return AES_CCM_decrypt_lens(CCM_NONCE_LEN, CCM_TAG_LEN, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt);
}
#define AES_CCM_encrypt AES_CCM_encrypt_nl
#define AES_CCM_decrypt AES_CCM_decrypt_nl
#endif

void AES_OCB_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_OCB_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);
/* OCB_NONCE_LEN (1..15) / OCB_TAG_LEN (1..16), micro_aes.h:115-116: the nonce block of OCB_cipher (:1703-1709) */
void AES_OCB_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt);
char AES_OCB_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt);
#if defined(OCB_NONCE_LEN) || defined(OCB_TAG_LEN)
UAES_STATIC_INLINE void AES_OCB_encrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *pntxt, const size_t ptextLen, void *crtxt)
{
    AES_OCB_encrypt_lens(OCB_NONCE_LEN, OCB_TAG_LEN, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt);
}
UAES_STATIC_INLINE char AES_OCB_decrypt_nl(const uint8_t *key, const uint8_t *nonce,
                                           const void *aData, const size_t aDataLen,
                                           const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return AES_OCB_decrypt_lens(OCB_NONCE_LEN, OCB_TAG_LEN, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt);
}
#define AES_OCB_encrypt AES_OCB_encrypt_nl
#define AES_OCB_decrypt AES_OCB_decrypt_nl
#endif

void GCM_SIV_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt);
char GCM_SIV_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt);

void AES_CMAC(const uint8_t *key,
              const void *data, const size_t dataSize, uint8_t mac[16]);

/* Not in the reference: what happens when a `void` function above cannot do its work
 * (no usable HIP device, an allocation or launch failure -- there is no CPU path).  The
 * default handler prints the reason and abort()s; install another to unwind instead.
 * Returns the previous handler (NULL = the default).  fn = API name, rc = engine code
 * (include/uaes_hip.h), msg = uaes_last_error().                                      */
typedef void (*uaes_failure_handler)(const char *fn, int rc, const char *msg);
uaes_failure_handler uaes_compat_set_failure_handler(uaes_failure_handler h);

/* Not in the reference either: DEVICE pointers are accepted wherever the reference takes a buffer.  A call first waits
 * for the work the caller may have in flight on the default stream; a caller that produces its data on a stream of
 * its own (hipStreamNonBlocking streams do not synchronise with the default one) names that stream here, per thread
 * (a hipStream_t; NULL = the default stream again).  Long HOST buffers can be spread over several GPUs without a
 * change in the caller: environment variable UAES_DEVICES=all | 0,1,... (include/uaes_hip.h, uaes_set_devices).   */
void uaes_compat_set_producer_stream(void *stream);

#ifdef __cplusplus
}
#endif

enum function_result_codes
{
    M_ENCRYPTION_ERROR     = 0x1E,
    M_DECRYPTION_ERROR     = 0x1D,
    M_AUTHENTICATION_ERROR = 0x1A,
    M_DATALENGTH_ERROR     = 0x1L,
    M_RESULT_SUCCESS       = 0
};

#endif /* MICRO_AES_H_ */
