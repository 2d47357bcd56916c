"""micro-aes_amd -- Python host mirror of the MI355X AES engine.

The product is ``lib/libuaes_hip.so`` (C ABI in ``include/uaes_hip.h``: hand
written gfx950 kernels behind plain-C entry points).  This package is the thin
host-side mirror of the reference's operator interface for the hot path --
``AES_ECB_encrypt`` ... ``AES_GCM_decrypt`` with the reference's argument order
(micro_aes.h:173-181, :239-249, :256-266, :294-308), minus the explicit lengths
-- plus the device-resident entry points that bench.py and the multi-GPU
sharding layer drive with ``torch`` tensors' raw pointers.  torch is only
plumbing here (device memory, streams, torch.distributed).

There is NO fallback: if the shared library is missing or no HIP device is
usable, importing ``engine()`` / calling any function raises.  (The engine's own
host data path, csrc/uaes_host.c, is opt-in: ``host_policy()`` below.)

The directory name carries a hyphen (it is the name the project was given), so
import it through the alias module at the repo root::

    import micro_aes_amd as uaes
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")
CSRC = os.path.join(HERE, "csrc")

M_RESULT_SUCCESS = 0
M_DATALENGTH_ERROR = 1
M_AUTHENTICATION_ERROR = 0x1A
M_DECRYPTION_ERROR = 0x1D
M_ENCRYPTION_ERROR = 0x1E

EXPORTS = [
    "uaes_init", "uaes_shutdown", "uaes_selftest", "uaes_last_error", "uaes_version", "uaes_expand_key", "uaes_stream_release",
    "uaes_set_wipe_on_auth_failure", "uaes_set_gcm_one_pass_decrypt", "uaes_clock_probe_dev",
    "uaes_ecb_encrypt", "uaes_ecb_encrypt_padded", "uaes_ecb_decrypt", "uaes_ctr_xcrypt", "uaes_ctr_xcrypt_iv", "uaes_ctr_xcrypt_at",
    "uaes_xts_encrypt", "uaes_xts_decrypt", "uaes_xts_sectors",
    "uaes_gcm_encrypt", "uaes_gcm_decrypt", "uaes_gcm_encrypt_iv", "uaes_gcm_decrypt_iv", "uaes_ghash",
    "uaes_gcm_encrypt_ex", "uaes_gcm_decrypt_ex", "uaes_ccm_encrypt_ex", "uaes_ccm_decrypt_ex",
    "uaes_ocb_encrypt_ex", "uaes_ocb_decrypt_ex",
    "uaes_cmac", "uaes_ccm_encrypt", "uaes_ccm_decrypt", "uaes_gcmsiv_encrypt", "uaes_gcmsiv_decrypt",
    "uaes_ocb_encrypt", "uaes_ocb_decrypt", "uaes_ocb_dev",
    "uaes_mgpu_ctr_xcrypt_at", "uaes_mgpu_xts_sectors", "uaes_mgpu_ctr_encrypt_gather", "uaes_debug_gather_stats", "uaes_debug_gcm_look", "uaes_debug_gcm_chunk_folds",
    "uaes_debug_plan", "uaes_debug_arrangement_name", "uaes_debug_plan_disable",
    "uaes_mgpu_ecb_encrypt", "uaes_mgpu_ecb_decrypt", "uaes_mgpu_gcm_encrypt", "uaes_mgpu_gcm_decrypt",
    "uaes_set_devices", "uaes_set_producer_stream", "uaes_set_host_policy", "uaes_get_host_policy",
    "uaes_gcm_key_new", "uaes_gcm_key_free", "uaes_gcm_key_encrypt", "uaes_gcm_key_decrypt",
    "uaes_gcm_key_encrypt_dev", "uaes_gcm_key_decrypt_dev",
    "uaes_gcm_record_max", "uaes_gcm_key_encrypt_records", "uaes_gcm_key_decrypt_records",
    "uaes_gcm_key_encrypt_records_dev", "uaes_gcm_key_decrypt_records_dev",
    "uaes_gcm_key_encrypt_records_v", "uaes_gcm_key_decrypt_records_v",
    "uaes_gcm_key_encrypt_records_v_dev", "uaes_gcm_key_decrypt_records_v_dev",
    "uaes_gcm_stream_begin", "uaes_gcm_stream_update", "uaes_gcm_stream_finish", "uaes_gcm_stream_abort",
    "uaes_cbc_encrypt_batch", "uaes_cmac_batch", "uaes_cbc_encrypt", "uaes_cbc_decrypt", "uaes_cbc_encrypt_padded", "uaes_cbc_decrypt_blocks", "uaes_cfb_encrypt", "uaes_cfb_decrypt", "uaes_ofb_xcrypt",
    "uaes_ecb_dev", "uaes_ctr_xcrypt_at_dev", "uaes_xts_sectors_dev",
    "uaes_gcm_encrypt_dev", "uaes_gcm_decrypt_dev", "uaes_gcm_partial_dev", "uaes_gcm_shard_dev",
]
COMPAT_EXPORTS = [
    "AES_ECB_encrypt", "AES_ECB_encrypt_pkcs7", "AES_ECB_encrypt_iso7816", "AES_ECB_decrypt",
    "AES_CTR_encrypt", "AES_CTR_decrypt", "AES_CTR_encrypt_preset", "AES_CTR_decrypt_preset", "AES_CTR_encrypt_iv",
    "AES_CBC_encrypt_nocts", "AES_CBC_encrypt_nocts_pkcs7", "AES_CBC_encrypt_nocts_iso7816", "AES_CBC_decrypt_nocts",
    "uaes_compat_set_failure_handler", "uaes_compat_set_producer_stream",
    "AES_XTS_encrypt", "AES_XTS_decrypt", "AES_GCM_encrypt", "AES_GCM_decrypt",
    "AES_GCM_encrypt_ivlen", "AES_GCM_decrypt_ivlen", "AES_GCM_encrypt_lens", "AES_GCM_decrypt_lens",
    "AES_CCM_encrypt_lens", "AES_CCM_decrypt_lens", "AES_OCB_encrypt_lens", "AES_OCB_decrypt_lens",
    "AES_CCM_encrypt", "AES_CCM_decrypt", "AES_CMAC", "GCM_SIV_encrypt", "GCM_SIV_decrypt",
    "AES_OCB_encrypt", "AES_OCB_decrypt",
    "AES_CBC_encrypt", "AES_CBC_decrypt", "AES_CFB_encrypt", "AES_CFB_decrypt", "AES_OFB_encrypt", "AES_OFB_decrypt",
]


class EngineError(RuntimeError):
    pass


def build(quiet=True):
    """Compile the HIP kernels and the C host layer for gfx950, in-tree."""
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib_path(name="libuaes_hip.so"):
    return os.path.join(LIB_DIR, name)


_lib = None


def engine():
    """The loaded C-ABI library (ctypes.CDLL), prototypes attached."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, same
    # SONAME as /opt/rocm's).  Two HIP runtimes in one process do not work
    # ("No HIP GPUs are available" from whichever initialises second), so when
    # torch is installed let it load its runtime FIRST; our library then binds
    # to that same copy.  A plain C caller simply uses the system ROCm.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise EngineError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (there is no CPU fallback)" % path)
    L = C.CDLL(path)
    sz, i, vp, u64 = C.c_size_t, C.c_int, C.c_void_p, C.c_uint64
    L.uaes_init.restype = i
    L.uaes_selftest.restype = i
    L.uaes_last_error.restype = C.c_char_p
    L.uaes_version.restype = C.c_char_p
    for n in ("uaes_ecb_encrypt", "uaes_ecb_decrypt"):
        getattr(L, n).argtypes = [i, vp, vp, sz, vp]
    L.uaes_stream_release.argtypes = [vp]
    L.uaes_clock_probe_dev.argtypes = [vp, C.c_uint, vp]
    L.uaes_cbc_encrypt_batch.argtypes = [i, vp, vp, sz, sz, vp, vp]
    L.uaes_cmac_batch.argtypes = [i, vp, sz, sz, vp, vp]
    L.uaes_ecb_encrypt_padded.argtypes = [i, vp, i, vp, sz, vp]
    L.uaes_ctr_xcrypt.argtypes = [i, vp, vp, vp, sz, vp]
    L.uaes_ctr_xcrypt_at.argtypes = [i, vp, vp, u64, vp, sz, vp]
    L.uaes_ctr_xcrypt_iv.argtypes = [i, vp, vp, sz, u64, vp, sz, vp]
    L.uaes_cbc_encrypt_padded.argtypes = [i, vp, vp, i, vp, sz, vp]
    L.uaes_cbc_decrypt_blocks.argtypes = [i, vp, vp, vp, sz, vp]
    for n in ("uaes_xts_encrypt", "uaes_xts_decrypt"):
        getattr(L, n).argtypes = [i, vp, vp, vp, sz, vp]
    L.uaes_xts_sectors.argtypes = [i, vp, u64, sz, sz, vp, vp, i]
    for n in ("uaes_gcm_encrypt", "uaes_gcm_decrypt"):
        getattr(L, n).argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
    for n in ("uaes_gcm_encrypt_iv", "uaes_gcm_decrypt_iv"):
        getattr(L, n).argtypes = [i, vp, vp, sz, vp, sz, vp, sz, vp]
    for n in ("uaes_gcm_encrypt_ex", "uaes_gcm_decrypt_ex", "uaes_ccm_encrypt_ex", "uaes_ccm_decrypt_ex",
              "uaes_ocb_encrypt_ex", "uaes_ocb_decrypt_ex"):
        getattr(L, n).argtypes = [i, vp, vp, sz, sz, vp, sz, vp, sz, vp]
    L.uaes_ghash.argtypes = [vp, vp, sz, vp, sz, vp]
    L.uaes_cmac.argtypes = [i, vp, vp, sz, vp]
    for n in ("uaes_cbc_encrypt", "uaes_cbc_decrypt", "uaes_cbc_decrypt_blocks", "uaes_cfb_encrypt", "uaes_cfb_decrypt", "uaes_ofb_xcrypt"):
        getattr(L, n).argtypes = [i, vp, vp, vp, sz, vp]
    for n in ("uaes_ccm_encrypt", "uaes_ccm_decrypt", "uaes_gcmsiv_encrypt", "uaes_gcmsiv_decrypt",
              "uaes_ocb_encrypt", "uaes_ocb_decrypt"):
        getattr(L, n).argtypes = [i, vp, vp, vp, sz, vp, sz, vp]
    L.uaes_mgpu_ctr_xcrypt_at.argtypes = [i, C.POINTER(C.c_int), i, vp, vp, u64, vp, sz, vp]
    L.uaes_mgpu_xts_sectors.argtypes = [i, C.POINTER(C.c_int), i, vp, u64, sz, sz, vp, vp, i]
    L.uaes_mgpu_ctr_encrypt_gather.argtypes = [i, C.POINTER(C.c_int), i, vp, vp, u64, C.POINTER(vp), sz, C.POINTER(vp), i, vp]
    if hasattr(L, "uaes_debug_gcm_look"):              # (an older build loaded for an A/B run has no test hooks)
        L.uaes_debug_gather_stats.argtypes = [C.POINTER(C.c_ulong)]
        L.uaes_debug_gather_stats.restype = None
        L.uaes_debug_gcm_look.argtypes = [C.c_ulonglong]
        L.uaes_debug_gcm_look.restype = None
        L.uaes_debug_gcm_chunk_folds.argtypes = [C.POINTER(C.c_uint)]
    if hasattr(L, "uaes_debug_plan"):
        L.uaes_debug_plan.argtypes = [i, i, sz, sz, C.c_uint, C.POINTER(C.c_int)]
        L.uaes_debug_arrangement_name.argtypes = [i]
        L.uaes_debug_arrangement_name.restype = C.c_char_p
        L.uaes_debug_plan_disable.argtypes = [C.c_uint]
        L.uaes_debug_plan_disable.restype = None
    L.uaes_mgpu_ecb_encrypt.argtypes = [i, C.POINTER(C.c_int), i, vp, i, vp, sz, vp]
    L.uaes_mgpu_ecb_decrypt.argtypes = [i, C.POINTER(C.c_int), i, vp, vp, sz, vp]
    for n in ("uaes_mgpu_gcm_encrypt", "uaes_mgpu_gcm_decrypt"):
        getattr(L, n).argtypes = [i, C.POINTER(C.c_int), i, vp, vp, vp, sz, vp, sz, vp]
    L.uaes_set_devices.argtypes = [i, C.POINTER(C.c_int), sz]
    L.uaes_set_host_policy.argtypes = [sz, i, i]
    L.uaes_get_host_policy.argtypes = [C.POINTER(sz), C.POINTER(i), C.POINTER(i)]
    L.uaes_set_producer_stream.argtypes = [vp]
    L.uaes_gcm_key_new.argtypes = [C.POINTER(vp), i, vp]
    L.uaes_gcm_key_free.argtypes = [vp]
    L.uaes_gcm_key_free.restype = None
    L.uaes_gcm_key_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]
    L.uaes_gcm_key_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]
    L.uaes_gcm_key_encrypt_dev.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
    L.uaes_gcm_key_decrypt_dev.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.uaes_gcm_record_max.argtypes = [sz]
    L.uaes_gcm_record_max.restype = sz
    L.uaes_gcm_key_encrypt_records.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, vp, sz]
    L.uaes_gcm_key_decrypt_records.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, vp, sz, vp]
    L.uaes_gcm_key_encrypt_records_dev.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, vp, sz, vp]
    L.uaes_gcm_key_decrypt_records_dev.argtypes = [vp, sz, vp, vp, sz, sz, vp, sz, sz, vp, sz, vp, vp, vp]
    L.uaes_gcm_key_encrypt_records_v.argtypes = [vp, sz, vp, vp, sz, sz, vp, vp, sz, sz, vp, sz]
    L.uaes_gcm_key_decrypt_records_v.argtypes = [vp, sz, vp, vp, sz, sz, vp, vp, sz, sz, vp, sz, vp]
    L.uaes_gcm_key_encrypt_records_v_dev.argtypes = [vp, sz, vp, vp, sz, sz, vp, vp, sz, sz, vp, sz, vp]
    L.uaes_gcm_key_decrypt_records_v_dev.argtypes = [vp, sz, vp, vp, sz, sz, vp, vp, sz, sz, vp, sz, vp, vp, vp]
    L.uaes_gcm_stream_begin.argtypes = [C.POINTER(vp), i, vp, vp, vp, sz, i]
    L.uaes_gcm_stream_update.argtypes = [vp, vp, sz, vp]
    L.uaes_gcm_stream_finish.argtypes = [vp, vp]
    L.uaes_gcm_stream_abort.argtypes = [vp]
    L.uaes_gcm_stream_abort.restype = None
    L.uaes_ocb_dev.argtypes = [i, vp, vp, i, vp, sz, vp, sz, vp, vp, vp]
    L.uaes_ecb_dev.argtypes = [i, vp, i, vp, sz, vp, vp]
    L.uaes_ctr_xcrypt_at_dev.argtypes = [i, vp, vp, u64, vp, sz, vp, vp]
    L.uaes_xts_sectors_dev.argtypes = [i, vp, u64, sz, sz, vp, vp, i, vp]
    L.uaes_gcm_encrypt_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp, vp]
    L.uaes_gcm_decrypt_dev.argtypes = [i, vp, vp, vp, sz, vp, sz, vp, vp, vp]
    L.uaes_gcm_partial_dev.argtypes = [i, vp, vp, vp, u64, vp, sz, u64, u64, vp, vp]
    L.uaes_gcm_shard_dev.argtypes = [i, vp, vp, i, vp, u64, vp, sz, u64, u64, vp, vp, vp]
    L.uaes_expand_key.argtypes = [i, vp, vp, vp]
    for n in EXPORTS:
        if n.startswith("uaes_debug_") and not hasattr(L, n):
            continue
        if n not in ("uaes_last_error", "uaes_version", "uaes_gcm_key_free", "uaes_gcm_stream_abort", "uaes_debug_gather_stats", "uaes_debug_gcm_look",
                     "uaes_debug_arrangement_name", "uaes_debug_plan_disable"):
            getattr(L, n).restype = i
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, engine().uaes_last_error().decode()))
    return rc


def _in(b):
    b = bytes(b)
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


def _fixed(b, n, what):
    """a fixed-size argument (iv, nonce, tweak, counter block): the C side copies exactly n bytes"""
    b = bytes(b)
    if len(b) != n:
        raise ValueError("%s must be exactly %d bytes (got %d)" % (what, n, len(b)))
    return (C.c_uint8 * n).from_buffer_copy(b)


def _fixed_bytes(b, n):
    b = bytes(b)
    if len(b) != n:
        raise ValueError("expected exactly %d bytes (got %d)" % (n, len(b)))
    return b


def _out(n, fill=0):
    buf = (C.c_uint8 * max(n, 1))()
    if fill:
        C.memset(buf, fill, max(n, 1))
    return buf


def _bits(key, mult=1):
    bits = len(key) * 8 // mult
    if bits not in (128, 192, 256):
        raise ValueError("key must be %s bytes" % "/".join(str(mult * k) for k in (16, 24, 32)))
    return bits


# ---------------------------------------------------------------------------
# host-buffer API: same names and argument meaning as the reference
# ---------------------------------------------------------------------------
def AES_ECB_encrypt(key, pntxt, padding=0):
    """micro_aes.c:636.  padding = the reference's AES_PADDING (micro_aes.h:79): 0 returns
    ceil(len/16)*16 bytes (zero padded tail), 1 (PKCS#7) / 2 (ISO 7816-4) always add a block."""
    if padding not in (0, 1, 2):
        raise ValueError("padding must be 0, 1 or 2")
    n = (len(pntxt) // 16 + 1) * 16 if padding else (len(pntxt) + 15) // 16 * 16
    o = _out(n)
    _check(engine().uaes_ecb_encrypt_padded(_bits(key), _in(key), padding, _in(pntxt), len(pntxt), o),
           "AES_ECB_encrypt")
    return bytes(o)[:n]


def AES_ECB_decrypt(key, crtxt):
    """micro_aes.c:663.  Returns (code, plaintext); code 0x1D on a ragged length."""
    o = _out(len(crtxt))
    rc = _check(engine().uaes_ecb_decrypt(_bits(key), _in(key), _in(crtxt), len(crtxt), o), "AES_ECB_decrypt")
    return rc, bytes(o)[: len(crtxt)]


def AES_CTR_encrypt(key, iv, pntxt, iv_length=12, start_value=1):
    """micro_aes.c:962.  iv: 12 bytes; counter block = iv || 00000001.  iv_length / start_value = the reference's
    compile-time CTR_IV_LENGTH (<= 16) / CTR_START_VALUE (micro_aes.h:98-99; 12 / 1 = the default build)."""
    o = _out(len(pntxt))
    if not 0 <= iv_length <= 16 or not 0 <= start_value < 1 << 64:
        raise ValueError("iv_length: 0..16; start_value: a 64-bit unsigned integer")
    if iv_length == 12 and start_value == 1:
        _check(engine().uaes_ctr_xcrypt(_bits(key), _in(key), _fixed(iv, 12, "iv"), _in(pntxt), len(pntxt), o), "AES_CTR_encrypt")
    else:
        _check(engine().uaes_ctr_xcrypt_iv(_bits(key), _in(key), _fixed(iv, iv_length, "iv"), iv_length, start_value,
                                           _in(pntxt), len(pntxt), o), "AES_CTR_encrypt")
    return bytes(o)[: len(pntxt)]


AES_CTR_decrypt = AES_CTR_encrypt            # micro_aes.c:986-990


def ctr_xcrypt_at(key, ctr0, block_offset, data):
    """Sharding extension: explicit 16-byte counter block + 56-bit block offset."""
    o = _out(len(data))
    _check(engine().uaes_ctr_xcrypt_at(_bits(key), _in(key), _fixed(ctr0, 16, "ctr0"), block_offset, _in(data), len(data), o),
           "uaes_ctr_xcrypt_at")
    return bytes(o)[: len(data)]


def _xts(fn, name, keys, tweak, data, prefill):
    o = _out(len(data), prefill)
    rc = _check(fn(_bits(keys, 2), _in(keys), None if tweak is None else _fixed(tweak, 16, "tweak"), _in(data), len(data), o), name)
    return rc, bytes(o)[: len(data)]


def AES_XTS_encrypt(keys, tweak, pntxt, prefill=0):
    """micro_aes.c:1066.  Returns (code, ciphertext); code 1 if len < 16 (output untouched)."""
    return _xts(engine().uaes_xts_encrypt, "AES_XTS_encrypt", keys, tweak, pntxt, prefill)


def AES_XTS_decrypt(keys, tweak, crtxt, prefill=0):
    """micro_aes.c:1085."""
    return _xts(engine().uaes_xts_decrypt, "AES_XTS_decrypt", keys, tweak, crtxt, prefill)


def xts_sectors(keys, first_sector, sector_bytes, data, encrypt=True):
    """Batch extension: data unit i has tweak LE128(first_sector + i)."""
    assert sector_bytes > 0 and len(data) % sector_bytes == 0
    o = _out(len(data))
    rc = _check(engine().uaes_xts_sectors(_bits(keys, 2), _in(keys), first_sector, sector_bytes,
                                          len(data) // sector_bytes, _in(data), o, 1 if encrypt else 0),
                "uaes_xts_sectors")
    return rc, bytes(o)[: len(data)]


def _taglen(tag_len, lo, hi, even, what):
    if not (lo <= tag_len <= hi) or (even and tag_len % 2):
        raise ValueError("%s tag length %d (%s%d..%d)" % (what, tag_len, "even, " if even else "", lo, hi))
    return tag_len


def AES_GCM_encrypt(key, nonce, aData, pntxt, tag_len=16):
    """micro_aes.c:1164.  Returns ciphertext || tag.  len(nonce) / tag_len are the reference's GCM_NONCE_LEN /
    GCM_TAG_LEN: 12 / 16 by default, any other nonce length >= 1 derives J0 = GHASH(nonce) (:1145-1149)."""
    if len(nonce) < 1:
        raise ValueError("empty nonce")
    _taglen(tag_len, 1, 16, False, "GCM")
    o = _out(len(pntxt) + 16)
    _check(engine().uaes_gcm_encrypt_ex(_bits(key), _in(key), _in(nonce), len(nonce), tag_len, _in(aData), len(aData),
                                        _in(pntxt), len(pntxt), o), "AES_GCM_encrypt")
    return bytes(o)[: len(pntxt) + tag_len]


def AES_GCM_decrypt(key, nonce, aData, crtxt_and_tag, prefill=0, tag_len=16):
    """micro_aes.c:1192.  Returns (code, plaintext); code 0x1A and an untouched
    (prefilled) buffer when authentication fails."""
    _taglen(tag_len, 1, 16, False, "GCM")
    n = len(crtxt_and_tag) - tag_len
    o = _out(n, prefill)
    if len(nonce) < 1:
        raise ValueError("empty nonce")
    rc = _check(engine().uaes_gcm_decrypt_ex(_bits(key), _in(key), _in(nonce), len(nonce), tag_len, _in(aData), len(aData),
                                             _in(crtxt_and_tag), n, o), "AES_GCM_decrypt")
    return rc, bytes(o)[:n]


def _fb(fn, name, key, iVec, data, prefill=0):
    o = _out(len(data), prefill)
    rc = _check(fn(_bits(key), _in(key), _fixed(iVec, 16, "iVec"), _in(data), len(data), o), name)
    return rc, bytes(o)[: len(data)]


def AES_CBC_encrypt(key, iVec, pntxt, prefill=0, cts=True, padding=0):
    """micro_aes.c:697 (CS3 ciphertext stealing).  Returns (code, ciphertext); code 1 if len < 16.
    cts=False: a build with CTS 0 (micro_aes.h:56) -- no stealing, any length, the last chunk padded like ECB's
    with padding = AES_PADDING (micro_aes.c:727-733): 16 * (len // 16 + (len % 16 or padding != 0)) bytes."""
    if cts:
        return _fb(engine().uaes_cbc_encrypt, "AES_CBC_encrypt", key, iVec, pntxt, prefill)
    if padding not in (0, 1, 2):
        raise ValueError("padding must be 0, 1 or 2")
    n = len(pntxt) // 16 * 16 + (16 if (len(pntxt) % 16 or padding) else 0)
    o = _out(n, prefill)
    rc = _check(engine().uaes_cbc_encrypt_padded(_bits(key), _in(key), _fixed(iVec, 16, "iVec"), padding,
                                                 _in(pntxt), len(pntxt), o), "AES_CBC_encrypt")
    return rc, bytes(o)[:n]


def AES_CBC_decrypt(key, iVec, crtxt, prefill=0, cts=True):
    """micro_aes.c:746.  cts=False: the CTS 0 build -- whole blocks only (code 1 otherwise, :761), padding left in place."""
    return _fb(engine().uaes_cbc_decrypt if cts else engine().uaes_cbc_decrypt_blocks, "AES_CBC_decrypt", key, iVec, crtxt, prefill)


def cbc_encrypt_batch(key, ivs, messages):
    """Independent CBC chains, one GPU lane each: == [AES_CBC_encrypt(key, iv, m)[1] for iv, m in ...]."""
    n = len(messages)
    if n == 0:
        return []
    size = len(messages[0])
    if any(len(m) != size for m in messages) or len(ivs) != n or any(len(v) != 16 for v in ivs):
        raise ValueError("equal-sized messages and one 16-byte IV per message")
    o = _out(n * size)
    _check(engine().uaes_cbc_encrypt_batch(_bits(key), _in(key), _in(b"".join(ivs)), n, size,
                                           _in(b"".join(messages)), o), "uaes_cbc_encrypt_batch")
    raw = bytes(o)
    return [raw[i * size:(i + 1) * size] for i in range(n)]


def cmac_batch(key, messages):
    """Independent CMACs, one GPU lane each: == [AES_CMAC(key, m) for m in messages]."""
    n = len(messages)
    if n == 0:
        return []
    size = len(messages[0])
    if any(len(m) != size for m in messages):
        raise ValueError("equal-sized messages")
    o = _out(n * 16)
    _check(engine().uaes_cmac_batch(_bits(key), _in(key), n, size, _in(b"".join(messages)), o), "uaes_cmac_batch")
    raw = bytes(o)
    return [raw[16 * i:16 * i + 16] for i in range(n)]


def AES_CFB_encrypt(key, iVec, pntxt):
    """micro_aes.c:825."""
    return _fb(engine().uaes_cfb_encrypt, "AES_CFB_encrypt", key, iVec, pntxt)[1]


def AES_CFB_decrypt(key, iVec, crtxt):
    """micro_aes.c:839."""
    return _fb(engine().uaes_cfb_decrypt, "AES_CFB_decrypt", key, iVec, crtxt)[1]


def AES_OFB_encrypt(key, iVec, data):
    """micro_aes.c:861; decrypt is the same function (:887)."""
    return _fb(engine().uaes_ofb_xcrypt, "AES_OFB_encrypt", key, iVec, data)[1]


AES_OFB_decrypt = AES_OFB_encrypt


def AES_CMAC(key, data):
    """micro_aes.c:1108.  Returns the 16-byte CMAC."""
    o = _out(16)
    _check(engine().uaes_cmac(_bits(key), _in(key), _in(data), len(data), o), "AES_CMAC")
    return bytes(o)


def _ccm_nonce(nonce):
    if not 7 <= len(nonce) <= 13:
        raise ValueError("CCM nonce of %d bytes (7..13)" % len(nonce))
    return _in(nonce)


def AES_CCM_encrypt(key, nonce, aData, pntxt, tag_len=16):
    """micro_aes.c:1268.  len(nonce) / tag_len are the reference's CCM_NONCE_LEN (7..13; 11 by default) and
    CCM_TAG_LEN (even, 4..16); returns ciphertext || tag."""
    _taglen(tag_len, 4, 16, True, "CCM")
    o = _out(len(pntxt) + 16)
    _check(engine().uaes_ccm_encrypt_ex(_bits(key), _in(key), _ccm_nonce(nonce), len(nonce), tag_len, _in(aData), len(aData),
                                        _in(pntxt), len(pntxt), o), "AES_CCM_encrypt")
    return bytes(o)[: len(pntxt) + tag_len]


def AES_CCM_decrypt(key, nonce, aData, crtxt_and_tag, prefill=0, tag_len=16):
    """micro_aes.c:1294.  Returns (code, text); like the reference the decrypted
    text is returned even when the code is 0x1A."""
    _taglen(tag_len, 4, 16, True, "CCM")
    n = len(crtxt_and_tag) - tag_len
    o = _out(n, prefill)
    rc = _check(engine().uaes_ccm_decrypt_ex(_bits(key), _in(key), _ccm_nonce(nonce), len(nonce), tag_len, _in(aData),
                                             len(aData), _in(crtxt_and_tag), n, o), "AES_CCM_decrypt")
    return rc, bytes(o)[:n]


def GCM_SIV_encrypt(key, nonce, aData, pntxt):
    """micro_aes.c:1473 (RFC 8452).  Returns ciphertext || 16-byte tag."""
    o = _out(len(pntxt) + 16)
    _check(engine().uaes_gcmsiv_encrypt(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                        _in(pntxt), len(pntxt), o), "GCM_SIV_encrypt")
    return bytes(o)[: len(pntxt) + 16]


def GCM_SIV_decrypt(key, nonce, aData, crtxt_and_tag, prefill=0):
    """micro_aes.c:1494.  Returns (code, text)."""
    n = len(crtxt_and_tag) - 16
    o = _out(n, prefill)
    rc = _check(engine().uaes_gcmsiv_decrypt(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                             _in(crtxt_and_tag), n, o), "GCM_SIV_decrypt")
    return rc, bytes(o)[:n]


def _ocb_nonce(nonce):
    if not 1 <= len(nonce) <= 15:
        raise ValueError("OCB nonce of %d bytes (1..15)" % len(nonce))
    return _in(nonce)


def AES_OCB_encrypt(key, nonce, aData, pntxt, tag_len=16):
    """micro_aes.c:1774 (RFC 7253).  len(nonce) / tag_len are the reference's OCB_NONCE_LEN (1..15; 12 by default)
    and OCB_TAG_LEN (1..16).  Returns ciphertext || tag."""
    _taglen(tag_len, 1, 16, False, "OCB")
    o = _out(len(pntxt) + 16)
    _check(engine().uaes_ocb_encrypt_ex(_bits(key), _in(key), _ocb_nonce(nonce), len(nonce), tag_len, _in(aData), len(aData),
                                        _in(pntxt), len(pntxt), o), "AES_OCB_encrypt")
    return bytes(o)[: len(pntxt) + tag_len]


def AES_OCB_decrypt(key, nonce, aData, crtxt_and_tag, prefill=0, tag_len=16):
    """micro_aes.c:1797.  Returns (code, text); the text is written even on 0x1A."""
    _taglen(tag_len, 1, 16, False, "OCB")
    n = len(crtxt_and_tag) - tag_len
    o = _out(n, prefill)
    rc = _check(engine().uaes_ocb_decrypt_ex(_bits(key), _in(key), _ocb_nonce(nonce), len(nonce), tag_len, _in(aData),
                                             len(aData), _in(crtxt_and_tag), n, o), "AES_OCB_decrypt")
    return rc, bytes(o)[:n]


# ---------------------------------------------------------------------------
# one process, several GPUs (uaes_mgpu_*): `devices` = list of HIP ordinals (an ordinal may repeat)
# ---------------------------------------------------------------------------
def _devs(devices):
    devices = list(devices)
    return len(devices), (C.c_int * len(devices))(*devices)


def _buf(x):
    """host bytes -> ctypes copy; an int is taken as a raw (device) address"""
    return C.c_void_p(x) if isinstance(x, int) else _in(x)


def mgpu_ecb_encrypt(devices, key, pntxt, padding=0):
    n, d = _devs(devices)
    m = (len(pntxt) // 16 + 1) * 16 if padding else (len(pntxt) + 15) // 16 * 16
    o = _out(m)
    _check(engine().uaes_mgpu_ecb_encrypt(n, d, _bits(key), _in(key), padding, _in(pntxt), len(pntxt), o), "uaes_mgpu_ecb_encrypt")
    return bytes(o)[:m]


def mgpu_ecb_decrypt(devices, key, crtxt, prefill=0):
    n, d = _devs(devices)
    o = _out(len(crtxt), prefill)
    rc = _check(engine().uaes_mgpu_ecb_decrypt(n, d, _bits(key), _in(key), _in(crtxt), len(crtxt), o), "uaes_mgpu_ecb_decrypt")
    return rc, bytes(o)[: len(crtxt)]


def mgpu_gcm_encrypt(devices, key, nonce, aData, pntxt):
    """== AES_GCM_encrypt, the text cut into one 16-byte aligned slice per device (uaes_mgpu_gcm_encrypt)."""
    n, d = _devs(devices)
    o = _out(len(pntxt) + 16)
    _check(engine().uaes_mgpu_gcm_encrypt(n, d, _bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                          _in(pntxt), len(pntxt), o), "uaes_mgpu_gcm_encrypt")
    return bytes(o)[: len(pntxt) + 16]


def mgpu_gcm_decrypt(devices, key, nonce, aData, crtxt_and_tag, prefill=0):
    """== AES_GCM_decrypt: (code, plaintext); 0x1A and an untouched (prefilled) buffer on a forgery (N7 across devices)."""
    n, d = _devs(devices)
    m = len(crtxt_and_tag) - 16
    o = _out(m, prefill)
    rc = _check(engine().uaes_mgpu_gcm_decrypt(n, d, _bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                               _in(crtxt_and_tag), m, o), "uaes_mgpu_gcm_decrypt")
    return rc, bytes(o)[:m]


def mgpu_gcm_dev(devices, key, nonce, aad, src_ptr, nbytes, dst_ptr, decrypt=False):
    """the same on raw device addresses (ints): encrypt writes nbytes + 16 at dst_ptr; decrypt reads nbytes + 16"""
    n, d = _devs(devices)
    fn = engine().uaes_mgpu_gcm_decrypt if decrypt else engine().uaes_mgpu_gcm_encrypt
    return _check(fn(n, d, _bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aad or b""), len(aad or b""),
                     C.c_void_p(src_ptr), nbytes, C.c_void_p(dst_ptr)), "uaes_mgpu_gcm_*")


class GcmKey:
    """GCM key context (uaes_gcm_key_*): the key's GHASH tables are built once; encrypt / decrypt then equal
    AES_GCM_encrypt / AES_GCM_decrypt bit for bit.  One call at a time per object."""

    def __init__(self, key):
        self._h = C.c_void_p()
        _check(engine().uaes_gcm_key_new(C.byref(self._h), _bits(key), _in(key)), "uaes_gcm_key_new")

    def encrypt(self, nonce, aData, pntxt):
        o = _out(len(pntxt) + 16)
        _check(engine().uaes_gcm_key_encrypt(self._h, _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                             _in(pntxt), len(pntxt), o), "uaes_gcm_key_encrypt")
        return bytes(o)[: len(pntxt) + 16]

    def decrypt(self, nonce, aData, crtxt_and_tag, prefill=0):
        n = len(crtxt_and_tag) - 16
        o = _out(n, prefill)
        rc = _check(engine().uaes_gcm_key_decrypt(self._h, _fixed(nonce, 12, "nonce"), _in(aData), len(aData),
                                                  _in(crtxt_and_tag), n, o), "uaes_gcm_key_decrypt")
        return rc, bytes(o)[:n]

    def encrypt_dev(self, nonce, aad, src, nbytes, dst, stream=None):
        _check(engine().uaes_gcm_key_encrypt_dev(self._h, _fixed(nonce, 12, "nonce"), _ptr(aad),
                                                 0 if aad is None else aad.numel(), _ptr(src), nbytes, _ptr(dst),
                                                 _stream(stream)), "uaes_gcm_key_encrypt_dev")

    def decrypt_dev(self, nonce, aad, src, nbytes, dst, status, stream=None):
        _check(engine().uaes_gcm_key_decrypt_dev(self._h, _fixed(nonce, 12, "nonce"), _ptr(aad),
                                                 0 if aad is None else aad.numel(), _ptr(src), nbytes, _ptr(dst),
                                                 _ptr(status), _stream(stream)), "uaes_gcm_key_decrypt_dev")

    @staticmethod
    def record_max(aad_len=0):
        """longest record (bytes) the record calls take with aad_len bytes of AAD per record"""
        return int(engine().uaes_gcm_record_max(aad_len))

    def encrypt_records(self, nonces, aads, records, stride=None):
        """Many equally long messages in one launch (uaes_gcm_key_encrypt_records).  nonces: list of 12-byte
        values; aads: one bytes object for all records or a list of equally long ones; records: list of equally
        long plaintexts.  Returns the list of ciphertext || tag, each equal to encrypt() of that record."""
        n = len(records)
        if n == 0:
            return []
        rec_len = len(records[0])
        if any(len(r) != rec_len for r in records) or len(nonces) != n:
            raise ValueError("records must be equally long and have one nonce each")
        stride = stride or (rec_len + 16 + 15) // 16 * 16
        aad, aad_len, aad_stride = self._pack_aads(aads, n)
        src = bytearray(stride * n)
        for r, rec in enumerate(records):
            src[r * stride: r * stride + rec_len] = rec
        dst = _out(stride * n)
        _check(engine().uaes_gcm_key_encrypt_records(self._h, n, _in(b"".join(_fixed_bytes(x, 12) for x in nonces)),
                                                     _in(aad), aad_len, aad_stride, _in(bytes(src)), rec_len, stride,
                                                     dst, stride), "uaes_gcm_key_encrypt_records")
        b = bytes(dst)
        return [b[r * stride: r * stride + rec_len + 16] for r in range(n)]

    def decrypt_records(self, nonces, aads, records, prefill=0, stride=None):
        """records: list of equally long ciphertext || tag.  Returns (code, verdicts, texts): code 0 or 0x1A (some
        record failed), verdicts[r] 0 / 0x1A, texts[r] the plaintext -- `prefill` bytes where the tag was wrong (N7)."""
        n = len(records)
        if n == 0:
            return 0, [], []
        rec_len = len(records[0]) - 16
        if rec_len < 0 or any(len(r) != rec_len + 16 for r in records) or len(nonces) != n:
            raise ValueError("records must be equally long (text || tag) and have one nonce each")
        stride = stride or (rec_len + 16 + 15) // 16 * 16
        aad, aad_len, aad_stride = self._pack_aads(aads, n)
        src = bytearray(stride * n)
        for r, rec in enumerate(records):
            src[r * stride: r * stride + rec_len + 16] = rec
        dst = _out(stride * n, prefill)
        ver = _out(n, 0x55)
        rc = _check(engine().uaes_gcm_key_decrypt_records(self._h, n, _in(b"".join(_fixed_bytes(x, 12) for x in nonces)),
                                                          _in(aad), aad_len, aad_stride, _in(bytes(src)), rec_len, stride,
                                                          dst, stride, ver), "uaes_gcm_key_decrypt_records")
        b = bytes(dst)
        return rc, list(bytes(ver)), [b[r * stride: r * stride + rec_len] for r in range(n)]

    def encrypt_records_v(self, nonces, aads, records, max_len=None, stride=None):
        """records of DIFFERENT lengths in equal slots (uaes_gcm_key_encrypt_records_v).  Returns the list of
        ciphertext || tag, each equal to encrypt() of that record."""
        n = len(records)
        if n == 0:
            return []
        max_len = max(len(r) for r in records) if max_len is None else max_len
        stride = stride or (max_len + 16 + 15) // 16 * 16
        aad, aad_len, aad_stride = self._pack_aads(aads, n)
        src = bytearray(stride * n)
        for r, rec in enumerate(records):
            src[r * stride: r * stride + len(rec)] = rec
        lens = (C.c_uint32 * n)(*[len(r) for r in records])
        dst = _out(stride * n)
        _check(engine().uaes_gcm_key_encrypt_records_v(self._h, n, _in(b"".join(_fixed_bytes(x, 12) for x in nonces)),
                                                       _in(aad), aad_len, aad_stride, _in(bytes(src)), lens, max_len, stride,
                                                       dst, stride), "uaes_gcm_key_encrypt_records_v")
        b = bytes(dst)
        return [b[r * stride: r * stride + len(records[r]) + 16] for r in range(n)]

    def decrypt_records_v(self, nonces, aads, records, prefill=0, max_len=None, stride=None):
        """records: list of ciphertext || tag of different lengths.  Returns (code, verdicts, texts)."""
        n = len(records)
        if n == 0:
            return 0, [], []
        tl = [len(r) - 16 for r in records]
        max_len = max(tl) if max_len is None else max_len
        stride = stride or (max_len + 16 + 15) // 16 * 16
        aad, aad_len, aad_stride = self._pack_aads(aads, n)
        src = bytearray(stride * n)
        for r, rec in enumerate(records):
            src[r * stride: r * stride + len(rec)] = rec
        lens = (C.c_uint32 * n)(*tl)
        dst = _out(stride * n, prefill)
        ver = _out(n, 0x55)
        rc = _check(engine().uaes_gcm_key_decrypt_records_v(self._h, n, _in(b"".join(_fixed_bytes(x, 12) for x in nonces)),
                                                            _in(aad), aad_len, aad_stride, _in(bytes(src)), lens, max_len, stride,
                                                            dst, stride, ver), "uaes_gcm_key_decrypt_records_v")
        b = bytes(dst)
        return rc, list(bytes(ver)), [b[r * stride: r * stride + tl[r]] for r in range(n)]

    @staticmethod
    def _pack_aads(aads, n):
        if isinstance(aads, (bytes, bytearray)):
            return bytes(aads), len(aads), 0
        if len(aads) != n or any(len(a) != len(aads[0]) for a in aads):
            raise ValueError("one equally long AAD per record (or one bytes object for all)")
        return b"".join(aads), len(aads[0]), len(aads[0])

    def encrypt_records_dev(self, nrec, nonces, aad, aad_len, aad_stride, src, rec_len, in_stride, dst, out_stride, stream=None):
        """device tensors (uint8): nonces 12 * nrec bytes; enqueues on `stream`"""
        _check(engine().uaes_gcm_key_encrypt_records_dev(self._h, nrec, _ptr(nonces), _ptr(aad), aad_len, aad_stride, _ptr(src),
                                                         rec_len, in_stride, _ptr(dst), out_stride, _stream(stream)),
               "uaes_gcm_key_encrypt_records_dev")

    def decrypt_records_dev(self, nrec, nonces, aad, aad_len, aad_stride, src, rec_len, in_stride, dst, out_stride,
                            verdicts, status, stream=None):
        _check(engine().uaes_gcm_key_decrypt_records_dev(self._h, nrec, _ptr(nonces), _ptr(aad), aad_len, aad_stride, _ptr(src),
                                                         rec_len, in_stride, _ptr(dst), out_stride, _ptr(verdicts), _ptr(status),
                                                         _stream(stream)), "uaes_gcm_key_decrypt_records_dev")

    def close(self):
        if getattr(self, "_h", None):
            engine().uaes_gcm_key_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: the module globals may be gone already
            pass


class GcmStream:
    """One GCM message fed in pieces (uaes_gcm_stream_*): same bytes as AES_GCM_encrypt/decrypt."""

    def __init__(self, key, nonce, aData=b"", decrypt=False):
        self._h = C.c_void_p()
        _check(engine().uaes_gcm_stream_begin(C.byref(self._h), _bits(key), _in(key), _fixed(nonce, 12, "nonce"), _in(aData),
                                              len(aData), 1 if decrypt else 0), "uaes_gcm_stream_begin")
        self._decrypt = decrypt

    def update(self, piece):
        o = _out(len(piece))
        _check(engine().uaes_gcm_stream_update(self._h, _in(piece), len(piece), o), "uaes_gcm_stream_update")
        return bytes(o)[: len(piece)]

    def finish(self, tag=None):
        """encrypt: returns the tag; decrypt: returns 0 or M_AUTHENTICATION_ERROR for the given tag."""
        t = _fixed(tag if self._decrypt else bytes(16), 16, "tag")
        h, self._h = self._h, None
        rc = _check(engine().uaes_gcm_stream_finish(h, t), "uaes_gcm_stream_finish")
        return rc if self._decrypt else bytes(t)

    def __del__(self):
        if getattr(self, "_h", None):
            engine().uaes_gcm_stream_abort(self._h)
            self._h = None


MODES = {"ecb": 0, "ctr": 1, "xts": 2, "gcm": 3, "ocb": 4, "siv": 5}


def plan(mode, a, b=0, direction=0, flags=0):
    """What a call would run (uaes_debug_plan): (arrangement name, launches, workgroups, positions per thread)."""
    out = (C.c_int * 4)()
    _check(engine().uaes_debug_plan(MODES[mode], direction, a, b, flags, out), "uaes_debug_plan")
    return engine().uaes_debug_arrangement_name(out[0]).decode(), out[1], out[2], out[3]


def arrangement_id(name):
    L = engine()
    for k in range(64):
        if L.uaes_debug_arrangement_name(k) == name.encode():
            return k
    raise KeyError(name)


def ghash(H, aData, crtxt):
    """gHash of micro_aes.c:1127 with an explicit subkey (test hook)."""
    o = _out(16)
    _check(engine().uaes_ghash(_fixed(H, 16, "H"), _in(aData), len(aData), _in(crtxt), len(crtxt), o), "uaes_ghash")
    return bytes(o)


def selftest():
    return _check(engine().uaes_selftest(), "uaes_selftest")


def host_policy(max_bytes=None, chains=None, fallback=None):
    """Get / set the opt-in host data path (uaes_set_host_policy).  Returns the policy in force BEFORE the call as
    (max_bytes, chains, fallback); arguments left None keep their value.  Default (0, 0, 0): GPU always."""
    mb, ch, fb = C.c_size_t(), C.c_int(), C.c_int()
    engine().uaes_get_host_policy(C.byref(mb), C.byref(ch), C.byref(fb))
    prev = (mb.value, ch.value, fb.value)
    if max_bytes is not None or chains is not None or fallback is not None:
        engine().uaes_set_host_policy(prev[0] if max_bytes is None else max_bytes,
                                      prev[1] if chains is None else int(chains),
                                      prev[2] if fallback is None else int(fallback))
    return prev


# ---------------------------------------------------------------------------
# device-resident API (torch tensors are only carriers of raw device pointers)
# ---------------------------------------------------------------------------
def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def ctr_xcrypt_dev(key, ctr0, block_offset, src, dst, nbytes=None, stream=None):
    """Enqueue CTR over device tensors (uint8, contiguous).  No synchronisation."""
    n = src.numel() * src.element_size() if nbytes is None else nbytes
    _check(engine().uaes_ctr_xcrypt_at_dev(_bits(key), _in(key), _fixed(ctr0, 16, "ctr0"), block_offset,
                                           _ptr(src), n, _ptr(dst), _stream(stream)), "uaes_ctr_xcrypt_at_dev")


def ecb_dev(key, src, dst, decrypt=False, nbytes=None, stream=None):
    n = src.numel() * src.element_size() if nbytes is None else nbytes
    _check(engine().uaes_ecb_dev(_bits(key), _in(key), 1 if decrypt else 0, _ptr(src), n, _ptr(dst),
                                 _stream(stream)), "uaes_ecb_dev")


def xts_sectors_dev(keys, first_sector, sector_bytes, nsectors, src, dst, encrypt=True, stream=None):
    _check(engine().uaes_xts_sectors_dev(_bits(keys, 2), _in(keys), first_sector, sector_bytes, nsectors,
                                         _ptr(src), _ptr(dst), 1 if encrypt else 0, _stream(stream)),
           "uaes_xts_sectors_dev")


def gcm_encrypt_dev(key, nonce, aad, src, nbytes, dst, stream=None):
    """dst must hold nbytes + 16 (tag appended)."""
    _check(engine().uaes_gcm_encrypt_dev(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), _ptr(aad),
                                         0 if aad is None else aad.numel(), _ptr(src), nbytes, _ptr(dst),
                                         _stream(stream)), "uaes_gcm_encrypt_dev")


def gcm_decrypt_dev(key, nonce, aad, src, nbytes, dst, status, stream=None):
    """src holds nbytes + 16 (ciphertext || tag); status: int32 device tensor."""
    _check(engine().uaes_gcm_decrypt_dev(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), _ptr(aad),
                                         0 if aad is None else aad.numel(), _ptr(src), nbytes, _ptr(dst),
                                         _ptr(status), _stream(stream)), "uaes_gcm_decrypt_dev")


def ocb_dev(key, nonce, aad, src, nbytes, dst, decrypt=False, status=None, stream=None):
    """encrypt: dst holds nbytes + 16; decrypt: src holds nbytes + 16, status = int32 device tensor."""
    _check(engine().uaes_ocb_dev(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), 1 if decrypt else 0, _ptr(aad),
                                 0 if aad is None else aad.numel(), _ptr(src), nbytes, _ptr(dst),
                                 _ptr(status), _stream(stream)), "uaes_ocb_dev")


def gcm_partial_dev(key, nonce, aad, total_aad_len, ct_shard, shard_len, shard_offset, total_len, partial,
                    stream=None):
    """This shard's 16-byte share of the GCM tag (see uaes_gcm_partial_dev)."""
    _check(engine().uaes_gcm_partial_dev(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), _ptr(aad), total_aad_len,
                                         _ptr(ct_shard), shard_len, shard_offset, total_len, _ptr(partial),
                                         _stream(stream)), "uaes_gcm_partial_dev")


def gcm_shard_dev(key, nonce, mode, aad, total_aad_len, src, shard_len, shard_offset, total_len, dst, partial, stream=None):
    """This shard's CTR pass (mode 0 encrypt / 2 decrypt; 1 = hash only) fused with its 16-byte share of the tag
    (uaes_gcm_shard_dev): one pass over the text."""
    _check(engine().uaes_gcm_shard_dev(_bits(key), _in(key), _fixed(nonce, 12, "nonce"), mode, _ptr(aad), total_aad_len,
                                       _ptr(src), shard_len, shard_offset, total_len, _ptr(dst), _ptr(partial),
                                       _stream(stream)), "uaes_gcm_shard_dev")
