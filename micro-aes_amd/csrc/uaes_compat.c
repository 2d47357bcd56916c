/*
 * uaes_compat.c -- the reference's compile-time-key-size API (micro_aes.h:173-
 * 181, :239-249, :256-266, :294-308) on top of the run-time engine.  Built
 * once per key size (-DAES___=128|192|256) into libmicro_aes_hip_<bits>.so.
 */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/micro_aes.h"
#include "../../include/uaes_hip.h"

#define KB (AES_KEYLENGTH * 8)

/* The reference's `void` functions cannot report failure, and there is no CPU path
 * to fall back to.  An engine failure (no device, hipMalloc, a launch error) goes to
 * a process-wide handler; the default prints and abort()s rather than hand back a
 * buffer that was never encrypted.  A host application can install its own with
 * uaes_compat_set_failure_handler (longjmp out, raise a language-level exception,
 * mark a connection dead ...).  If a handler RETURNS, the call returns to its caller
 * with the output buffer in an unspecified state: the handler has taken
 * responsibility for not using it.                                               */
static void default_failure(const char *fn, int rc, const char *msg)
{
    fprintf(stderr, "uaes-hip: %s failed (%d): %s\n", fn, rc, msg);
    abort();
}

static uaes_failure_handler g_on_failure = default_failure;

uaes_failure_handler uaes_compat_set_failure_handler(uaes_failure_handler h)
{
    uaes_failure_handler old = __atomic_exchange_n(&g_on_failure, h ? h : default_failure, __ATOMIC_ACQ_REL);
    return old == default_failure ? NULL : old;
}

void uaes_compat_set_producer_stream(void *stream)
{
    (void)uaes_set_producer_stream(stream);
}

static void must(const char *fn, int rc)
{
    if (rc == 0) return;
    __atomic_load_n(&g_on_failure, __ATOMIC_ACQUIRE)(fn, rc, uaes_last_error());
}

static char soft(const char *fn, int rc, char engine_code)
{
    if (rc >= 0) return (char)rc;              /* 0 or one of the reference's codes */
    fprintf(stderr, "uaes-hip: %s failed (%d): %s\n", fn, rc, uaes_last_error());
    return engine_code;
}

/* AES_PADDING (micro_aes.h:79, padBlock micro_aes.c:610-621) is a compile-time choice of the
 * CALLER's build in the reference; here every library carries all three and
 * include/micro_aes.h maps AES_ECB_encrypt onto the one the caller's AES_PADDING names. */
#undef AES_ECB_encrypt
void AES_ECB_encrypt(const uint8_t *key, const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_ECB_encrypt", uaes_ecb_encrypt_padded(KB, key, 0, pntxt, ptextLen, crtxt));
}

void AES_ECB_encrypt_pkcs7(const uint8_t *key, const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_ECB_encrypt", uaes_ecb_encrypt_padded(KB, key, 1, pntxt, ptextLen, crtxt));
}

void AES_ECB_encrypt_iso7816(const uint8_t *key, const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_ECB_encrypt", uaes_ecb_encrypt_padded(KB, key, 2, pntxt, ptextLen, crtxt));
}

char AES_ECB_decrypt(const uint8_t *key, const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_ECB_decrypt", uaes_ecb_decrypt(KB, key, crtxt, crtxtLen, pntxt), M_DECRYPTION_ERROR);
}

/* PRESET_COUNTER (micro_aes.h:100, micro_aes.c:965-966) is likewise the caller's compile-time choice:
 * both forms are exported and include/micro_aes.h binds AES_CTR_* to the one the caller's build names. */
#undef AES_CTR_encrypt
#undef AES_CTR_decrypt
void AES_CTR_encrypt_preset(const uint8_t *key, const uint8_t *counter16,
                            const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CTR_encrypt", uaes_ctr_xcrypt_at(KB, key, counter16, 0, pntxt, ptextLen, crtxt));
}

void AES_CTR_decrypt_preset(const uint8_t *key, const uint8_t *counter16,
                            const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    must("AES_CTR_decrypt", uaes_ctr_xcrypt_at(KB, key, counter16, 0, crtxt, crtxtLen, pntxt));
}

/* CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99, micro_aes.c:968-971) are the caller's compile-time
 * constants as well: one general entry point, to which include/micro_aes.h binds a build that defines them */
void AES_CTR_encrypt_iv(const size_t ivLen, const size_t startValue, const uint8_t *key, const uint8_t *iv,
                        const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CTR_encrypt", uaes_ctr_xcrypt_iv(KB, key, iv, ivLen, startValue, pntxt, ptextLen, crtxt));
}

void AES_CTR_encrypt(const uint8_t *key, const uint8_t *iv,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CTR_encrypt", uaes_ctr_xcrypt(KB, key, iv, pntxt, ptextLen, crtxt));
}

void AES_CTR_decrypt(const uint8_t *key, const uint8_t *iv,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    must("AES_CTR_decrypt", uaes_ctr_xcrypt(KB, key, iv, crtxt, crtxtLen, pntxt));
}

char AES_XTS_encrypt(const uint8_t *keys, const uint8_t *tweak,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    return soft("AES_XTS_encrypt", uaes_xts_encrypt(KB, keys, tweak, pntxt, ptextLen, crtxt), M_ENCRYPTION_ERROR);
}

char AES_XTS_decrypt(const uint8_t *keys, const uint8_t *tweak,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_XTS_decrypt", uaes_xts_decrypt(KB, keys, tweak, crtxt, crtxtLen, pntxt), M_DECRYPTION_ERROR);
}

#undef AES_GCM_encrypt
#undef AES_GCM_decrypt
void AES_GCM_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_GCM_encrypt", uaes_gcm_encrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_GCM_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_GCM_decrypt", uaes_gcm_decrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

void AES_GCM_encrypt_ivlen(const size_t nonceLen, const uint8_t *key, const uint8_t *nonce,
                           const void *aData, const size_t aDataLen,
                           const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_GCM_encrypt", uaes_gcm_encrypt_iv(KB, key, nonce, nonceLen, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_GCM_decrypt_ivlen(const size_t nonceLen, const uint8_t *key, const uint8_t *nonce,
                           const void *aData, const size_t aDataLen,
                           const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_GCM_decrypt", uaes_gcm_decrypt_iv(KB, key, nonce, nonceLen, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

void AES_GCM_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_GCM_encrypt", uaes_gcm_encrypt(KB, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_GCM_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_GCM_decrypt", uaes_gcm_decrypt(KB, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

#undef AES_CCM_encrypt
#undef AES_CCM_decrypt
void AES_CCM_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CCM_encrypt", uaes_ccm_encrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_CCM_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_CCM_decrypt", uaes_ccm_decrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

void AES_CCM_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CCM_encrypt", uaes_ccm_encrypt(KB, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_CCM_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_CCM_decrypt", uaes_ccm_decrypt(KB, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

void AES_CMAC(const uint8_t *key, const void *data, const size_t dataSize, uint8_t mac[16])
{
    must("AES_CMAC", uaes_cmac(KB, key, data, dataSize, mac));
}

/* CTS (micro_aes.h:56) is the caller's compile-time choice too: with CTS 0 the reference's CBC pads its last
 * chunk like ECB (AES_PADDING) instead of stealing (micro_aes.c:704-733) and its decryption wants whole blocks
 * (:761).  Every library exports both families; include/micro_aes.h binds AES_CBC_* to the caller's.      */
#undef AES_CBC_encrypt
#undef AES_CBC_decrypt
char AES_CBC_encrypt_nocts(const uint8_t *key, const uint8_t iVec[16],
                           const void *pntxt, const size_t ptextLen, void *crtxt)
{
    return soft("AES_CBC_encrypt", uaes_cbc_encrypt_padded(KB, key, iVec, 0, pntxt, ptextLen, crtxt), M_ENCRYPTION_ERROR);
}

char AES_CBC_encrypt_nocts_pkcs7(const uint8_t *key, const uint8_t iVec[16],
                                 const void *pntxt, const size_t ptextLen, void *crtxt)
{
    return soft("AES_CBC_encrypt", uaes_cbc_encrypt_padded(KB, key, iVec, 1, pntxt, ptextLen, crtxt), M_ENCRYPTION_ERROR);
}

char AES_CBC_encrypt_nocts_iso7816(const uint8_t *key, const uint8_t iVec[16],
                                   const void *pntxt, const size_t ptextLen, void *crtxt)
{
    return soft("AES_CBC_encrypt", uaes_cbc_encrypt_padded(KB, key, iVec, 2, pntxt, ptextLen, crtxt), M_ENCRYPTION_ERROR);
}

char AES_CBC_decrypt_nocts(const uint8_t *key, const uint8_t iVec[16],
                           const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_CBC_decrypt", uaes_cbc_decrypt_blocks(KB, key, iVec, crtxt, crtxtLen, pntxt), M_DECRYPTION_ERROR);
}

char AES_CBC_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    return soft("AES_CBC_encrypt", uaes_cbc_encrypt(KB, key, iVec, pntxt, ptextLen, crtxt), M_ENCRYPTION_ERROR);
}

char AES_CBC_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_CBC_decrypt", uaes_cbc_decrypt(KB, key, iVec, crtxt, crtxtLen, pntxt), M_DECRYPTION_ERROR);
}

void AES_CFB_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_CFB_encrypt", uaes_cfb_encrypt(KB, key, iVec, pntxt, ptextLen, crtxt));
}

void AES_CFB_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    must("AES_CFB_decrypt", uaes_cfb_decrypt(KB, key, iVec, crtxt, crtxtLen, pntxt));
}

void AES_OFB_encrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_OFB_encrypt", uaes_ofb_xcrypt(KB, key, iVec, pntxt, ptextLen, crtxt));
}

void AES_OFB_decrypt(const uint8_t *key, const uint8_t iVec[16],
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    must("AES_OFB_decrypt", uaes_ofb_xcrypt(KB, key, iVec, crtxt, crtxtLen, pntxt));
}

void GCM_SIV_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("GCM_SIV_encrypt", uaes_gcmsiv_encrypt(KB, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char GCM_SIV_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("GCM_SIV_decrypt", uaes_gcmsiv_decrypt(KB, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

#undef AES_OCB_encrypt
#undef AES_OCB_decrypt
void AES_OCB_encrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_OCB_encrypt", uaes_ocb_encrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_OCB_decrypt_lens(const size_t nonceLen, const size_t tagLen, const uint8_t *key, const uint8_t *nonce,
                          const void *aData, const size_t aDataLen,
                          const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_OCB_decrypt", uaes_ocb_decrypt_ex(KB, key, nonce, nonceLen, tagLen, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}

void AES_OCB_encrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *pntxt, const size_t ptextLen, void *crtxt)
{
    must("AES_OCB_encrypt", uaes_ocb_encrypt(KB, key, nonce, aData, aDataLen, pntxt, ptextLen, crtxt));
}

char AES_OCB_decrypt(const uint8_t *key, const uint8_t *nonce,
                     const void *aData, const size_t aDataLen,
                     const void *crtxt, const size_t crtxtLen, void *pntxt)
{
    return soft("AES_OCB_decrypt", uaes_ocb_decrypt(KB, key, nonce, aData, aDataLen, crtxt, crtxtLen, pntxt),
                M_DECRYPTION_ERROR);
}
