"""ctypes bindings for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (micro-aes_amd/) never does.

Two checkers are exposed:

* ``Oracle``      -- our plain-C restatement (oracle/uaes_oracle.c), run-time
                     key size, always available after ``make -C oracle``.
* ``Reference``   -- the REAL reference compiled from /root/reference by
                     oracle/Makefile into oracle/_ref/libmicroaes_ref_<bits>.so
                     (one library per compile-time key size, micro_aes.h:17).
                     Present wherever the prebuilt files travelled.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)


def _buf(b):
    """bytes/bytearray -> ctypes array (copy); None -> NULL."""
    if b is None:
        return None
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))


def _out(n):
    return (C.c_uint8 * max(n, 1))()


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.L = C.CDLL(path)
        sz, i, vp, u64 = C.c_size_t, C.c_int, C.c_void_p, C.c_uint64
        L.orc_ecb_encrypt.argtypes = [i, vp, vp, sz, vp]; L.orc_ecb_encrypt.restype = None
        L.orc_gcm_encrypt_iv.argtypes = [i, vp, vp, sz, vp, sz, vp, sz, vp]; L.orc_gcm_encrypt_iv.restype = None
        L.orc_gcm_decrypt_iv.argtypes = [i, vp, vp, sz, vp, sz, vp, sz, vp]; L.orc_gcm_decrypt_iv.restype = C.c_char
        for m in ("gcm", "ccm", "ocb"):
            f = getattr(L, "orc_%s_encrypt_ex" % m); f.argtypes = [i, vp, vp, sz, sz, vp, sz, vp, sz, vp]; f.restype = None
            f = getattr(L, "orc_%s_decrypt_ex" % m); f.argtypes = [i, vp, vp, sz, sz, vp, sz, vp, sz, vp]; f.restype = C.c_char
        L.orc_ecb_encrypt_padded.argtypes = [i, vp, i, vp, sz, vp]; L.orc_ecb_encrypt_padded.restype = None
        L.orc_ecb_decrypt.argtypes = [i, vp, vp, sz, vp]; L.orc_ecb_decrypt.restype = C.c_char
        L.orc_ctr_encrypt.argtypes = [i, vp, vp, vp, sz, vp]; L.orc_ctr_encrypt.restype = None
        L.orc_ctr_xcrypt_at.argtypes = [i, vp, vp, u64, vp, sz, vp]; L.orc_ctr_xcrypt_at.restype = None
        L.orc_ctr_encrypt_iv.argtypes = [i, vp, vp, sz, u64, vp, sz, vp]; L.orc_ctr_encrypt_iv.restype = None
        L.orc_cbc_encrypt_nocts.argtypes = [i, vp, vp, i, vp, sz, vp, C.POINTER(sz)]; L.orc_cbc_encrypt_nocts.restype = C.c_char
        L.orc_cbc_decrypt_nocts.argtypes = [i, vp, vp, vp, sz, vp]; L.orc_cbc_decrypt_nocts.restype = C.c_char
        for f in (L.orc_xts_encrypt, L.orc_xts_decrypt):
            f.argtypes = [i, vp, vp, vp, sz, vp]; f.restype = C.c_char
        L.orc_xts_sectors.argtypes = [i, vp, u64, sz, sz, vp, vp, i]; L.orc_xts_sectors.restype = C.c_char
        L.orc_gcm_encrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_gcm_encrypt.restype = None
        L.orc_gcm_decrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_gcm_decrypt.restype = C.c_char
        for f in (L.orc_cbc_encrypt, L.orc_cbc_decrypt):
            f.argtypes = [i, vp, vp, vp, sz, vp]; f.restype = C.c_char
        L.orc_cfb.argtypes = [i, vp, vp, i, vp, sz, vp]; L.orc_cfb.restype = None
        L.orc_ofb.argtypes = [i, vp, vp, vp, sz, vp]; L.orc_ofb.restype = None
        L.orc_cmac.argtypes = [i, vp, vp, sz, vp]; L.orc_cmac.restype = None
        L.orc_ccm_encrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_ccm_encrypt.restype = None
        L.orc_ccm_decrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_ccm_decrypt.restype = C.c_char
        L.orc_gcmsiv_encrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_gcmsiv_encrypt.restype = None
        L.orc_gcmsiv_decrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_gcmsiv_decrypt.restype = C.c_char
        L.orc_ocb_encrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_ocb_encrypt.restype = None
        L.orc_ocb_decrypt.argtypes = [i, vp, vp, vp, sz, vp, sz, vp]; L.orc_ocb_decrypt.restype = C.c_char
        L.orc_gf128_mul.argtypes = [vp, vp]; L.orc_gf128_mul.restype = None
        L.orc_ghash.argtypes = [vp, vp, sz, vp, sz, vp]; L.orc_ghash.restype = None
        L.orc_fill_splitmix.argtypes = [u64, u64, sz, vp]; L.orc_fill_splitmix.restype = None
        L.orc_setkey.argtypes = [vp, vp, i]; L.orc_setkey.restype = i
        L.orc_encrypt_block.argtypes = [vp, vp, vp]; L.orc_encrypt_block.restype = None
        L.orc_decrypt_block.argtypes = [vp, vp, vp]; L.orc_decrypt_block.restype = None

    # -- block primitive ----------------------------------------------------
    def encrypt_block(self, key, block, decrypt=False):
        ks = _out(4 + 240)
        assert self.L.orc_setkey(ks, _buf(key), len(key) * 8) == 0
        o = _out(16)
        (self.L.orc_decrypt_block if decrypt else self.L.orc_encrypt_block)(ks, _buf(block), o)
        return bytes(o)

    # -- modes --------------------------------------------------------------
    def ecb_encrypt(self, key, pt, padding=0):
        n = (len(pt) // 16 + 1) * 16 if padding else (len(pt) + 15) // 16 * 16
        o = _out(n)
        self.L.orc_ecb_encrypt_padded(len(key) * 8, _buf(key), padding, _buf(pt), len(pt), o)
        return bytes(o)[:n]

    def ecb_decrypt(self, key, ct):
        o = _out(len(ct))
        rc = self.L.orc_ecb_decrypt(len(key) * 8, _buf(key), _buf(ct), len(ct), o)
        return ord(rc), bytes(o)[: len(ct)]

    def ctr_encrypt(self, key, iv, data):
        o = _out(len(data))
        self.L.orc_ctr_encrypt(len(key) * 8, _buf(key), _buf(iv), _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def ctr_encrypt_iv(self, key, iv, start, data):
        """a build with CTR_IV_LENGTH = len(iv) (<= 16) and CTR_START_VALUE = start (micro_aes.h:98-99)"""
        o = _out(len(data))
        self.L.orc_ctr_encrypt_iv(len(key) * 8, _buf(key), _buf(iv), len(iv), start, _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def ctr_xcrypt_at(self, key, ctr0, block_offset, data):
        o = _out(len(data))
        self.L.orc_ctr_xcrypt_at(len(key) * 8, _buf(key), _buf(ctr0), block_offset,
                                 _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def xts(self, keys, tweak, data, encrypt=True, prefill=0xCC):
        o = (C.c_uint8 * max(len(data), 1))(*([prefill] * max(len(data), 1)))
        f = self.L.orc_xts_encrypt if encrypt else self.L.orc_xts_decrypt
        rc = f(len(keys) * 4, _buf(keys), _buf(tweak), _buf(data), len(data), o)
        return ord(rc), bytes(o)[: len(data)]

    def xts_sectors(self, keys, first_sector, sector_bytes, data, encrypt=True):
        n = len(data) // sector_bytes
        o = _out(len(data))
        rc = self.L.orc_xts_sectors(len(keys) * 4, _buf(keys), first_sector, sector_bytes, n,
                                    _buf(data), o, 1 if encrypt else 0)
        return ord(rc), bytes(o)[: len(data)]

    def gcm_encrypt(self, key, nonce, aad, pt, tag_len=16):
        """len(nonce) / tag_len play the role of the reference's GCM_NONCE_LEN / GCM_TAG_LEN (12 / 16 = default build)"""
        o = _out(len(pt) + 16)
        self.L.orc_gcm_encrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                  _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + tag_len]

    def gcm_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC, tag_len=16):
        n = len(ct_and_tag) - tag_len
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.orc_gcm_decrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                       _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def cbc(self, key, iv, data, encrypt=True, prefill=0xCC):
        o = (C.c_uint8 * max(len(data), 1))(*([prefill] * max(len(data), 1)))
        f = self.L.orc_cbc_encrypt if encrypt else self.L.orc_cbc_decrypt
        rc = f(len(key) * 8, _buf(key), _buf(iv), _buf(data), len(data), o)
        return ord(rc), bytes(o)[: len(data)]

    def cbc_nocts(self, key, iv, data, encrypt=True, padding=0, prefill=0xCC):
        """CBC of a build with CTS 0: encrypt pads like ECB (padding = AES_PADDING) and returns the padded
        length's worth; decrypt wants whole blocks (else 0x1D... the reference's M_DATALENGTH_ERROR = 1)"""
        cap = len(data) + 16
        o = (C.c_uint8 * cap)(*([prefill] * cap))
        if encrypt:
            n = C.c_size_t(0)
            rc = self.L.orc_cbc_encrypt_nocts(len(key) * 8, _buf(key), _buf(iv), padding, _buf(data), len(data), o, C.byref(n))
            return ord(rc), bytes(o)[: n.value]
        rc = self.L.orc_cbc_decrypt_nocts(len(key) * 8, _buf(key), _buf(iv), _buf(data), len(data), o)
        return ord(rc), bytes(o)[: len(data)]

    def cfb(self, key, iv, data, encrypt=True):
        o = _out(len(data))
        self.L.orc_cfb(len(key) * 8, _buf(key), _buf(iv), 1 if encrypt else 0, _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def ofb(self, key, iv, data):
        o = _out(len(data))
        self.L.orc_ofb(len(key) * 8, _buf(key), _buf(iv), _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def cmac(self, key, data):
        o = _out(16)
        self.L.orc_cmac(len(key) * 8, _buf(key), _buf(data), len(data), o)
        return bytes(o)

    def ccm_encrypt(self, key, nonce, aad, pt, tag_len=16):
        """len(nonce) / tag_len = CCM_NONCE_LEN (7..13) / CCM_TAG_LEN (even, 4..16)"""
        o = _out(len(pt) + 16)
        self.L.orc_ccm_encrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                  _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + tag_len]

    def ccm_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC, tag_len=16):
        n = len(ct_and_tag) - tag_len
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.orc_ccm_decrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                       _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def gcmsiv_encrypt(self, key, nonce, aad, pt):
        o = _out(len(pt) + 16)
        self.L.orc_gcmsiv_encrypt(len(key) * 8, _buf(key), _buf(nonce), _buf(aad), len(aad),
                                  _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + 16]

    def gcmsiv_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC):
        n = len(ct_and_tag) - 16
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.orc_gcmsiv_decrypt(len(key) * 8, _buf(key), _buf(nonce), _buf(aad), len(aad),
                                       _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def ocb_encrypt(self, key, nonce, aad, pt, tag_len=16):
        """len(nonce) / tag_len = OCB_NONCE_LEN (1..15) / OCB_TAG_LEN (1..16)"""
        o = _out(len(pt) + 16)
        self.L.orc_ocb_encrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                  _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + tag_len]

    def ocb_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC, tag_len=16):
        n = len(ct_and_tag) - tag_len
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.orc_ocb_decrypt_ex(len(key) * 8, _buf(key), _buf(nonce), len(nonce), tag_len, _buf(aad), len(aad),
                                       _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def gf128_mul(self, x, y):
        yy = _buf(y)
        self.L.orc_gf128_mul(_buf(x), yy)
        return bytes(yy)[:16]

    def ghash(self, H, aad, ct):
        g = _out(16)
        self.L.orc_ghash(_buf(H), _buf(aad), len(aad), _buf(ct), len(ct), g)
        return bytes(g)

    def splitmix(self, seed, nbytes, word0=0):
        """first `nbytes` of the synthetic stream starting at 64-bit word `word0`"""
        nw = (nbytes + 7) // 8
        o = _out(nw * 8)
        self.L.orc_fill_splitmix(seed, word0, nw, o)
        return bytes(o)[:nbytes]

    def splitmix_into(self, seed, array, word0=0):
        """fill a writable, C-contiguous numpy uint8 array (size % 8 == 0) in place"""
        assert array.nbytes % 8 == 0 and array.flags["C_CONTIGUOUS"]
        self.L.orc_fill_splitmix(seed, word0, array.nbytes // 8, C.c_void_p(array.ctypes.data))
        return array


class Reference:
    """The compiled reference; one library per key size (micro_aes.h:17)."""

    PAD_SUFFIX = {0: "", 1: "_pkcs7", 2: "_iso7816"}

    # builds with the other compile-time length constants patched (oracle/Makefile, micro_aes.h:103-116):
    # name -> (key bits, CCM_NONCE_LEN, CCM_TAG_LEN, GCM_TAG_LEN, OCB_NONCE_LEN, OCB_TAG_LEN)
    LENS = {"A": (128, 13, 4, 12, 15, 8), "B": (256, 7, 10, 4, 7, 12)}

    # builds with CTS 0 (micro_aes.h:56; CBC pads like ECB then, AES_PADDING as given) and with other
    # CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99); oracle/Makefile has the same tables:
    # name -> (key bits, AES_PADDING)  /  name -> (key bits, CTR_IV_LENGTH, CTR_START_VALUE)
    NOCTS = {"nocts": (128, 0), "nocts_pkcs7": (256, 1), "nocts_iso7816": (192, 2)}
    CTRV = {"ctrA": (128, 8, 0x01A2B3C4), "ctrB": (256, 16, 2)}

    @classmethod
    def path(cls, bits, padding=0, gcm_nonce_len=12, preset_counter=False, lens=None, variant=None):
        if variant:
            return os.path.join(HERE, "_ref", "libmicroaes_ref_%d_%s.so" % (bits, variant))
        iv = "" if gcm_nonce_len == 12 else "_gcmiv%d" % gcm_nonce_len
        pc = "_presetctr" if preset_counter else ""
        ln = "_lens%s" % lens if lens else ""
        return os.path.join(HERE, "_ref", "libmicroaes_ref_%d%s%s%s%s.so" % (bits, cls.PAD_SUFFIX[padding], iv, pc, ln))

    @classmethod
    def available(cls, bits=128, padding=0, gcm_nonce_len=12, preset_counter=False, lens=None, variant=None):
        return os.path.exists(cls.path(bits, padding, gcm_nonce_len, preset_counter, lens, variant))

    @classmethod
    def of_variant(cls, variant):
        """the build named in NOCTS / CTRV, or None if it did not travel"""
        bits = (cls.NOCTS.get(variant) or cls.CTRV[variant])[0]
        if not cls.available(bits, variant=variant):
            return None
        r = cls(bits, padding=cls.NOCTS[variant][1] if variant in cls.NOCTS else 0, variant=variant)
        return r

    def __init__(self, bits, padding=0, gcm_nonce_len=12, preset_counter=False, lens=None, variant=None):
        """padding / gcm_nonce_len / preset_counter: builds with AES_PADDING (micro_aes.h:79) /
        GCM_NONCE_LEN (:108) / PRESET_COUNTER (:100) patched (oracle/Makefile).  With preset_counter the
        `iv` of ctr_encrypt is the full 16-byte counter block (micro_aes.c:965-966)."""
        self.bits = bits
        self.padding = padding
        self.gcm_nonce_len = gcm_nonce_len
        self.preset_counter = preset_counter
        self.ccm_nonce, self.ccm_tag, self.gcm_tag, self.ocb_nonce, self.ocb_tag = 11, 16, 16, 12, 16
        if lens:
            assert self.LENS[lens][0] == bits
            self.ccm_nonce, self.ccm_tag, self.gcm_tag, self.ocb_nonce, self.ocb_tag = self.LENS[lens][1:]
        self.variant = variant
        self.ctr_iv_len, self.ctr_start = (self.CTRV[variant][1:] if variant in self.CTRV else (12, 1))
        L = self.L = C.CDLL(self.path(bits, padding, gcm_nonce_len, preset_counter, lens, variant))
        sz, vp = C.c_size_t, C.c_void_p
        L.AES_ECB_encrypt.argtypes = [vp, vp, sz, vp]; L.AES_ECB_encrypt.restype = None
        L.AES_ECB_decrypt.argtypes = [vp, vp, sz, vp]; L.AES_ECB_decrypt.restype = C.c_char
        for f in (L.AES_CTR_encrypt, L.AES_CTR_decrypt):
            f.argtypes = [vp, vp, vp, sz, vp]; f.restype = None
        for f in (L.AES_XTS_encrypt, L.AES_XTS_decrypt):
            f.argtypes = [vp, vp, vp, sz, vp]; f.restype = C.c_char
        L.AES_GCM_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_GCM_encrypt.restype = None
        L.AES_GCM_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_GCM_decrypt.restype = C.c_char
        L.AES_CCM_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_CCM_encrypt.restype = None
        L.AES_CCM_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_CCM_decrypt.restype = C.c_char
        L.AES_CMAC.argtypes = [vp, vp, sz, vp]; L.AES_CMAC.restype = None
        L.AES_OCB_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_OCB_encrypt.restype = None
        L.AES_OCB_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.AES_OCB_decrypt.restype = C.c_char
        L.GCM_SIV_encrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.GCM_SIV_encrypt.restype = None
        L.GCM_SIV_decrypt.argtypes = [vp, vp, vp, sz, vp, sz, vp]; L.GCM_SIV_decrypt.restype = C.c_char
        for f in (L.AES_CBC_encrypt, L.AES_CBC_decrypt):
            f.argtypes = [vp, vp, vp, sz, vp]; f.restype = C.c_char
        for f in (L.AES_CFB_encrypt, L.AES_CFB_decrypt, L.AES_OFB_encrypt, L.AES_OFB_decrypt):
            f.argtypes = [vp, vp, vp, sz, vp]; f.restype = None

    def ocb_encrypt(self, key, nonce, aad, pt):
        self._chk(key)
        assert len(nonce) == self.ocb_nonce
        o = _out(len(pt) + 16)
        self.L.AES_OCB_encrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + self.ocb_tag]

    def ocb_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC):
        self._chk(key)
        assert len(nonce) == self.ocb_nonce
        n = len(ct_and_tag) - self.ocb_tag
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.AES_OCB_decrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def gcmsiv_encrypt(self, key, nonce, aad, pt):
        self._chk(key)
        o = _out(len(pt) + 16)
        self.L.GCM_SIV_encrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + 16]

    def gcmsiv_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC):
        self._chk(key)
        n = len(ct_and_tag) - 16
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.GCM_SIV_decrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def cbc(self, key, iv, data, encrypt=True, prefill=0xCC):
        self._chk(key)
        o = (C.c_uint8 * max(len(data), 1))(*([prefill] * max(len(data), 1)))
        f = self.L.AES_CBC_encrypt if encrypt else self.L.AES_CBC_decrypt
        rc = f(_buf(key), _buf(iv), _buf(data), len(data), o)
        return ord(rc), bytes(o)[: len(data)]

    def cbc_nocts(self, key, iv, data, encrypt=True, prefill=0xCC):
        """only on a CTS 0 build (variant in NOCTS): AES_CBC_encrypt pads (micro_aes.c:727-733)"""
        self._chk(key)
        assert self.variant in self.NOCTS
        r = len(data) % 16
        n = len(data) if not encrypt else (len(data) - r + (16 if (r or self.padding) else 0))
        cap = len(data) + 16
        o = (C.c_uint8 * cap)(*([prefill] * cap))
        f = self.L.AES_CBC_encrypt if encrypt else self.L.AES_CBC_decrypt
        rc = f(_buf(key), _buf(iv), _buf(data), len(data), o)
        return ord(rc), bytes(o)[:n]

    def cfb(self, key, iv, data, encrypt=True):
        self._chk(key)
        o = _out(len(data))
        (self.L.AES_CFB_encrypt if encrypt else self.L.AES_CFB_decrypt)(_buf(key), _buf(iv), _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def ofb(self, key, iv, data):
        self._chk(key)
        o = _out(len(data))
        self.L.AES_OFB_encrypt(_buf(key), _buf(iv), _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def cmac(self, key, data):
        self._chk(key)
        o = _out(16)
        self.L.AES_CMAC(_buf(key), _buf(data), len(data), o)
        return bytes(o)

    def ccm_encrypt(self, key, nonce, aad, pt):
        self._chk(key)
        assert len(nonce) == self.ccm_nonce
        o = _out(len(pt) + 16)
        self.L.AES_CCM_encrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + self.ccm_tag]

    def ccm_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC):
        self._chk(key)
        assert len(nonce) == self.ccm_nonce
        n = len(ct_and_tag) - self.ccm_tag
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.AES_CCM_decrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]

    def _chk(self, key, mult=1):
        assert len(key) * 8 == self.bits * mult, "key size does not match this reference build"

    def ecb_encrypt(self, key, pt):
        self._chk(key)
        n = (len(pt) // 16 + 1) * 16 if self.padding else (len(pt) + 15) // 16 * 16
        o = _out(n)
        self.L.AES_ECB_encrypt(_buf(key), _buf(pt), len(pt), o)
        return bytes(o)[:n]

    def ecb_decrypt(self, key, ct):
        self._chk(key)
        o = _out(len(ct))
        rc = self.L.AES_ECB_decrypt(_buf(key), _buf(ct), len(ct), o)
        return ord(rc), bytes(o)[: len(ct)]

    def ctr_encrypt(self, key, iv, data):
        self._chk(key)
        o = _out(len(data))
        self.L.AES_CTR_encrypt(_buf(key), _buf(iv), _buf(data), len(data), o)
        return bytes(o)[: len(data)]

    def xts(self, keys, tweak, data, encrypt=True, prefill=0xCC):
        self._chk(keys, 2)
        o = (C.c_uint8 * max(len(data), 1))(*([prefill] * max(len(data), 1)))
        f = self.L.AES_XTS_encrypt if encrypt else self.L.AES_XTS_decrypt
        rc = f(_buf(keys), _buf(tweak), _buf(data), len(data), o)
        return ord(rc), bytes(o)[: len(data)]

    def gcm_encrypt(self, key, nonce, aad, pt):
        self._chk(key)
        o = _out(len(pt) + 16)
        self.L.AES_GCM_encrypt(_buf(key), _buf(nonce), _buf(aad), len(aad), _buf(pt), len(pt), o)
        return bytes(o)[: len(pt) + self.gcm_tag]

    def gcm_decrypt(self, key, nonce, aad, ct_and_tag, prefill=0xCC):
        self._chk(key)
        n = len(ct_and_tag) - self.gcm_tag
        o = (C.c_uint8 * max(n, 1))(*([prefill] * max(n, 1)))
        rc = self.L.AES_GCM_decrypt(_buf(key), _buf(nonce), _buf(aad), len(aad),
                                    _buf(ct_and_tag), n, o)
        return ord(rc), bytes(o)[:n]
