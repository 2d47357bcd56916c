"""GPU parity tests (need a real MI355X): the HIP engine, driven through its
C ABI, against (1) every golden vector the reference's own tests hold for the
hot path, (2) outputs of the compiled reference (tests/golden/ref_vectors.json,
digests.json) and (3) the CPU oracle on seeded inputs, with the reference's
edge semantics (N1..N8 of SURVEY.md section 8a).  Bit-exact everywhere."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

import micro_aes_amd as uaes
from tests.rsp import ccm_cases, cmac_cases, gcm_cases, gcmsiv_cases, ocb_cases, xts_cases

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def check_out(got, spec):
    if "hex" in spec:
        assert got.hex() == spec["hex"]
    else:
        assert len(got) == spec["len"]
        assert got[:16].hex() == spec["head"] and got[-16:].hex() == spec["tail"]
        assert hashlib.sha256(got).hexdigest() == spec["sha256"]


def test_device_selftest():
    assert uaes.selftest() == 0


# ---- the reference's own vectors, through the reference-compatible AES_* API ----
class Compat:
    """libmicro_aes_hip_<bits>.so: the drop-in AES_* symbols (include/micro_aes.h)"""

    def __init__(self, bits):
        uaes.engine()       # imports torch first: its bundled HIP runtime must be the one both bind to
        L = self.L = C.CDLL(uaes.lib_path("libmicro_aes_hip_%d.so" % bits))
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

    @staticmethod
    def buf(n, fill=0xCC):
        b = (C.c_uint8 * max(n, 1))()
        C.memset(b, fill, max(n, 1))
        return b


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_rsp_through_compat_api(bits):
    cases = gcm_cases(bits)
    assert len(cases) == 375                                   # the harness's own count
    L = Compat(bits)
    for c in cases:
        n = len(c["PT"])
        out = L.buf(n + 16)
        L.L.AES_GCM_encrypt(c["Key"], c["IV"], c["AAD"], len(c["AAD"]), c["PT"], n, out)
        assert bytes(out)[: n + 16] == c["CT"] + c["Tag"], c["Count"]
        pt = L.buf(n)
        rc = L.L.AES_GCM_decrypt(c["Key"], c["IV"], c["AAD"], len(c["AAD"]), c["CT"] + c["Tag"], n, pt)
        assert ord(rc) == 0 and bytes(pt)[:n] == c["PT"], c["Count"]


@pytest.mark.parametrize("bits", [128, 192, 256])
@pytest.mark.parametrize("iv_bytes", [1, 128])
def test_gcm_rsp_other_nonce_lengths_through_compat_api(bits, iv_bytes):
    """GCM_NONCE_LEN != 12 (J0 = GHASH(nonce), micro_aes.c:1145-1149): the NIST [IVlen = 8] and
    [IVlen = 1024] sections through the entry points a caller built with -DGCM_NONCE_LEN=n binds to"""
    cases = gcm_cases(bits, iv_bytes)
    assert len(cases) == 375
    L = Compat(bits)
    for c in cases[::3]:
        n = len(c["PT"])
        out = L.buf(n + 16)
        L.L.AES_GCM_encrypt_ivlen(C.c_size_t(iv_bytes), c["Key"], c["IV"], c["AAD"], C.c_size_t(len(c["AAD"])),
                                  c["PT"], C.c_size_t(n), out)
        assert bytes(out)[: n + 16] == c["CT"] + c["Tag"], c["Count"]
        pt = L.buf(n)
        L.L.AES_GCM_decrypt_ivlen.restype = C.c_char
        rc = L.L.AES_GCM_decrypt_ivlen(C.c_size_t(iv_bytes), c["Key"], c["IV"], c["AAD"], C.c_size_t(len(c["AAD"])),
                                       c["CT"] + c["Tag"], C.c_size_t(n), pt)
        assert ord(rc) == 0 and bytes(pt)[:n] == c["PT"], c["Count"]


def test_gcm_nonce_lengths_vs_oracle(orc):
    """nonces of 1..128 bytes over sizes up to the one-pass kernel's range; a flipped nonce bit fails"""
    rnd = random.Random(2024)
    for nl, n in ((1, 100), (8, 0), (11, 4096), (13, 70001), (16, (9 << 20) + 5), (60, 1 << 20), (128, 333)):
        bits = rnd.choice([128, 192, 256])
        key, nonce, aad = rnd.randbytes(bits // 8), rnd.randbytes(nl), rnd.randbytes(rnd.choice([0, 7, 40]))
        data = orc.splitmix(nl, (n + 7) // 8 * 8)[:n]
        ct = uaes.AES_GCM_encrypt(key, nonce, aad, data)
        assert ct == orc.gcm_encrypt(key, nonce, aad, data), (nl, n)
        assert uaes.AES_GCM_decrypt(key, nonce, aad, ct) == (0, data)
        bad = bytes([nonce[0] ^ 1]) + nonce[1:]
        assert uaes.AES_GCM_decrypt(key, bad, aad, ct, prefill=0xCC) == (0x1A, b"\xcc" * n)


@pytest.mark.parametrize("bits,count", [(128, 800), (256, 600)])
def test_xts_rsp_through_compat_api(bits, count):
    cases = xts_cases(bits)
    assert len(cases) == count
    L = Compat(bits)
    for c in cases:
        n = len(c["PT"])
        out = L.buf(n)
        assert ord(L.L.AES_XTS_encrypt(c["Key"], c["i"], c["PT"], n, out)) == 0
        assert bytes(out)[:n] == c["CT"], c["COUNT"]
        out = L.buf(n)
        assert ord(L.L.AES_XTS_decrypt(c["Key"], c["i"], c["CT"], n, out)) == 0
        assert bytes(out)[:n] == c["PT"], c["COUNT"]


@pytest.mark.parametrize("bits,count", [(128, 96), (192, 144), (256, 96)])
def test_cmac_rsp_through_compat_api(bits, count):
    cases = cmac_cases(bits)
    assert len(cases) == count
    L = Compat(bits)
    for c in cases:
        mac = L.buf(16)
        L.L.AES_CMAC(c["Key"], c["Msg"] or b"\0", len(c["Msg"]), mac)
        assert bytes(mac)[: c["Tlen"]] == c["Mac"], c["Count"]


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ccm_rsp_through_compat_api(bits):
    cases = ccm_cases(bits)
    assert len(cases) == 10
    L = Compat(bits)
    for c in cases:
        n = len(c["Payload"])
        out = L.buf(n + 16)
        L.L.AES_CCM_encrypt(c["Key"], c["Nonce"], c["Adata"], len(c["Adata"]), c["Payload"], n, out)
        assert bytes(out)[: n + 16] == c["CT"], c["Count"]
        pt = L.buf(n)
        rc = L.L.AES_CCM_decrypt(c["Key"], c["Nonce"], c["Adata"], len(c["Adata"]), c["CT"], n, pt)
        assert ord(rc) == 0 and bytes(pt)[:n] == c["Payload"]


class CompatLens(Compat):
    """the general entry points a caller built with -DCCM_NONCE_LEN / -DGCM_TAG_LEN / ... is bound to"""

    def __init__(self, bits):
        super().__init__(bits)
        sz, vp = C.c_size_t, C.c_void_p
        for m in ("GCM", "CCM", "OCB"):
            f = getattr(self.L, "AES_%s_encrypt_lens" % m); f.argtypes = [sz, sz, vp, vp, vp, sz, vp, sz, vp]; f.restype = None
            f = getattr(self.L, "AES_%s_decrypt_lens" % m); f.argtypes = [sz, sz, vp, vp, vp, sz, vp, sz, vp]; f.restype = C.c_char

    def enc(self, mode, key, nonce, tl, aad, pt):
        out = self.buf(len(pt) + 16)
        getattr(self.L, "AES_%s_encrypt_lens" % mode)(len(nonce), tl, key, nonce, aad, len(aad), pt, len(pt), out)
        assert bytes(out)[len(pt) + tl:] == b"\xCC" * (16 - tl) or tl == 16      # nothing behind the tag
        return bytes(out)[: len(pt) + tl]

    def dec(self, mode, key, nonce, tl, aad, ct):
        n = len(ct) - tl
        out = self.buf(n)
        rc = getattr(self.L, "AES_%s_decrypt_lens" % mode)(len(nonce), tl, key, nonce, aad, len(aad), ct, n, out)
        return ord(rc), bytes(out)[:n]


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ccm_rsp_every_nonce_length_through_compat_api(bits):
    """all seven [Nlen = 7..13] sections of the NIST VNT files (the reference's harness runs the one that equals
    its CCM_NONCE_LEN) through the entry point a -DCCM_NONCE_LEN=n caller is bound to"""
    L, total = CompatLens(bits), 0
    for nlen in range(7, 14):
        for c in ccm_cases(bits, nlen):
            assert L.enc("CCM", c["Key"], c["Nonce"], 16, c["Adata"], c["Payload"]) == c["CT"], (nlen, c["Count"])
            assert L.dec("CCM", c["Key"], c["Nonce"], 16, c["Adata"], c["CT"]) == (0, c["Payload"])
            total += 1
    assert total == 70


def test_ocb_vectors_with_other_lengths_through_compat_api():
    L = CompatLens(128)
    for nlen, tlen, count in ((12, 12, 6), (15, 16, 1), (12, 16, 16)):
        cases = ocb_cases(128, nlen, tlen)
        assert len(cases) == count
        for c in cases:
            assert L.enc("OCB", c["key"], c["iv"], tlen, c["aad"], c["pt"]) == c["ct"]
            assert L.dec("OCB", c["key"], c["iv"], tlen, c["aad"], c["ct"]) == (0, c["pt"])
    K, N = bytes.fromhex("0F0E0D0C0B0A09080706050403020100"), bytes.fromhex("BBAA9988776655443322110D")
    A = P = bytes(range(40))
    Cx = bytes.fromhex("1792A4E31E0755FB03E31B22116E6C2DDF9EFD6E33D536F1A0124B0A55BAE884ED93481529C76B6A"
                       "D0C515F4D1CDD4FDAC4F02AA")                                  # RFC 7253 appendix A, TAGLEN 96
    assert L.enc("OCB", K, N, 12, A, P) == Cx and L.dec("OCB", K, N, 12, A, Cx) == (0, P)


def test_length_constant_golden_vectors(golden_dir):
    """outputs of the reference built with the length constants patched (tests/golden/lens_vectors.json)"""
    with open(os.path.join(golden_dir, "lens_vectors.json")) as f:
        fx = json.load(f)
    for name, v in sorted(fx.items()):
        L = CompatLens(v["bits"])
        for c in v["cases"]:
            key, aad, pt = (bytes.fromhex(c[k]) for k in ("key", "aad", "pt"))
            for mode, tl in (("GCM", v["gcm_tag"]), ("CCM", v["ccm_tag"]), ("OCB", v["ocb_tag"])):
                nonce, want = bytes.fromhex(c[mode.lower()]["nonce"]), bytes.fromhex(c[mode.lower()]["out"])
                assert L.enc(mode, key, nonce, tl, aad, pt) == want, (name, mode, len(pt))
                assert L.dec(mode, key, nonce, tl, aad, want) == (0, pt), (name, mode, len(pt))


@pytest.mark.parametrize("bits", [128, 256])
def test_nonce_and_tag_lengths_vs_oracle(orc, bits):
    """every legal length of the three modes on texts that reach the one-launch, chunked and bulk GCM paths;
    a forged truncated tag leaves a GCM output untouched (N7) and returns 0x1A everywhere"""
    rnd = random.Random(bits)
    key = rnd.randbytes(bits // 8)
    sizes = [0, 1, 16, 33, 4096, 70001, (1 << 20) + 5]
    for mode, nonces, tags in (("gcm", (12, 1, 60), (1, 4, 8, 12, 13, 15, 16)),
                               ("ccm", range(7, 14), (4, 6, 8, 10, 12, 14, 16)),
                               ("ocb", (1, 7, 12, 15), (1, 8, 12, 15, 16))):
        enc, dec = getattr(uaes, "AES_%s_encrypt" % mode.upper()), getattr(uaes, "AES_%s_decrypt" % mode.upper())
        oenc = getattr(orc, "%s_encrypt" % mode)
        for i, tl in enumerate(tags):
            for j, nl in enumerate(nonces):
                n = sizes[(i + 2 * j) % len(sizes)]
                if mode != "gcm" and n > 70001:
                    n = 4096                                    # the CBC-MAC of CCM is a serial chain
                nonce, aad, pt = rnd.randbytes(nl), rnd.randbytes(rnd.choice([0, 9, 32, 300])), rnd.randbytes(n)
                want = oenc(key, nonce, aad, pt, tag_len=tl)
                assert enc(key, nonce, aad, pt, tag_len=tl) == want, (mode, nl, tl, n)
                assert dec(key, nonce, aad, want, tag_len=tl) == (0, pt), (mode, nl, tl, n)
                bad = want[:-1] + bytes([want[-1] ^ 1])
                rc, out = dec(key, nonce, aad, bad, prefill=0xA5, tag_len=tl)
                assert rc == 0x1A, (mode, nl, tl, n)
                if mode == "gcm":
                    assert out == b"\xA5" * n


def test_truncated_gcm_tag_with_device_buffers(orc):
    """a caller's device buffer ends GCM_TAG_LEN bytes behind the text: nothing may be written past it"""
    import torch
    key, nonce, aad = bytes(range(16)), bytes(range(12)), b"header"
    L = uaes.engine()
    for n in (100, 5000, 1 << 20):
        pt = bytes((i * 7 + 3) & 0xFF for i in range(n))
        src = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
        for tl in (4, 12):
            dst = torch.full((n + 32,), 0xEE, dtype=torch.uint8, device="cuda")
            assert L.uaes_gcm_encrypt_ex(128, key, nonce, 12, tl, aad, len(aad), C.c_void_p(src.data_ptr()), n,
                                         C.c_void_p(dst.data_ptr())) == 0
            got = bytes(dst.cpu().numpy())
            assert got[: n + tl] == orc.gcm_encrypt(key, nonce, aad, pt, tag_len=tl)
            assert got[n + tl:] == b"\xEE" * (32 - tl)
            back = torch.full((n + 16,), 0x77, dtype=torch.uint8, device="cuda")
            assert L.uaes_gcm_decrypt_ex(128, key, nonce, 12, tl, aad, len(aad), C.c_void_p(dst.data_ptr()), n,
                                         C.c_void_p(back.data_ptr())) == 0
            assert bytes(back.cpu().numpy()) == pt + b"\x77" * 16
            dst[n + tl - 1] ^= 1
            back.fill_(0x77)
            assert L.uaes_gcm_decrypt_ex(128, key, nonce, 12, tl, aad, len(aad), C.c_void_p(dst.data_ptr()), n,
                                         C.c_void_p(back.data_ptr())) == 0x1A
            assert bytes(back.cpu().numpy()) == b"\x77" * (n + 16)


def test_gcmsiv_acvp_vectors():
    cases = gcmsiv_cases(128)
    assert len(cases) == 102
    for c in cases:
        assert uaes.GCM_SIV_encrypt(c["key"], c["iv"], c["aad"], c["pt"]) == c["ct"], c["Count"]
        assert uaes.GCM_SIV_decrypt(c["key"], c["iv"], c["aad"], c["ct"]) == (0, c["pt"])


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcmsiv_vs_oracle(orc, bits):
    rnd = random.Random(bits + 6)
    # up to 2046 POLYVAL blocks (AAD + text + lengths) a message is ONE launch (k_siv_small); the explicit
    # (text, AAD) shapes sit on both sides of its two 1024-position steps and of its upper limit
    shapes = [(n, None) for n in [0, 1, 16, 17, 100, 4096, 65536 + 3, (1 << 20) + 16]] + \
             [(1020 * 16, 16), (1021 * 16, 0), (1021 * 16 + 1, 0), (1022 * 16 - 3, 5), (2044 * 16, 16), (2045 * 16, 0),
              (2045 * 16 + 1, 0), (2046 * 16, 0), (100, 2040 * 16), (16, 2046 * 16), (16384, 13)]
    for n, alen in shapes:
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
        aad = rnd.randbytes(rnd.choice([0, 1, 16, 20, 4097]) if alen is None else alen)
        data = orc.splitmix(n + 2, n)
        ct = uaes.GCM_SIV_encrypt(key, nonce, aad, data)
        assert ct == orc.gcmsiv_encrypt(key, nonce, aad, data), (n, len(aad))
        assert uaes.GCM_SIV_decrypt(key, nonce, aad, ct) == (0, data)
        bad = bytearray(ct)
        bad[0 if n else -1] ^= 0x01
        rc, txt = uaes.GCM_SIV_decrypt(key, nonce, aad, bytes(bad))
        assert (rc, txt) == orc.gcmsiv_decrypt(key, nonce, aad, bytes(bad)) and rc == 0x1A    # text released, as the reference does


def test_gcmsiv_long_messages_stay_on_the_device(orc):
    """A GCM-SIV message too long for the one-workgroup kernel: derive_keys, the derived key's expansion, POLYVAL, the
    tag and the counter made of it run as kernels one behind the other and never visit the host (uaesk_gcmsiv_long:
    k_siv_prep, the hash-only chunk workgroups with a finisher that makes the tag -- 1 .. 32 positions per thread, as
    many workgroups as CUs at 2^k blocks -- and the CTR kernel reading schedule and counter from the scratch; beyond
    one round of workgroups, 128 MiB, the GHASH levels and k_siv_tag).  Device pointers through the C ABI against the
    oracle: whole ciphertext and tag, decryption in place, a forgery (0x1A, the text released as the reference does)."""
    import torch
    L = uaes.engine()
    rnd = random.Random(8452)
    shapes = [(128, (8 << 20) - 16, 0), (128, 8 << 20, 0), (192, (8 << 20) + 16, 33), (256, (16 << 20) + 5, 4096),
              (128, (40 << 20) - 7, 0), (256, (100 << 20) + 3, 20), (128, 128 << 20, 0), (128, (128 << 20) + 16, 1),
              (192, 2047 * 16, 0), (128, 3 << 20, 1 << 20)]
    for bits, n, alen in shapes:
        key, nonce, aad = rnd.randbytes(bits // 8), rnd.randbytes(12), rnd.randbytes(alen)
        data = np.empty((n + 7) // 8 * 8, dtype=np.uint8)
        orc.splitmix_into(n % 1009 + 1, data)
        data = data[:n]
        want = orc.gcmsiv_encrypt(key, nonce, aad, bytes(data))
        src = torch.from_numpy(data.copy()).to("cuda:0")
        dst = torch.full((n + 32,), 0xA5, dtype=torch.uint8, device="cuda:0")
        a = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0") if alen else None
        ap = C.c_void_p(a.data_ptr()) if alen else None
        for _ in range(2):                                             # (the finisher's counter word must be back at zero)
            assert L.uaes_gcmsiv_encrypt(bits, key, nonce, ap, alen, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())) == 0
        got = bytes(dst[: n + 16].cpu().numpy())
        assert got[-16:] == want[-16:], (bits, n, alen)
        assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest(), (bits, n, alen)
        assert int((dst[n + 16:] != 0xA5).sum()) == 0
        work = dst[: n + 16].clone()                                   # in place
        assert L.uaes_gcmsiv_decrypt(bits, key, nonce, ap, alen, C.c_void_p(work.data_ptr()), n, C.c_void_p(work.data_ptr())) == 0
        assert torch.equal(work[:n], src), (bits, n, alen)
        bad = dst[: n + 16].clone()
        bad[rnd.randrange(n)] ^= 0x10
        back = torch.full((n,), 0xCC, dtype=torch.uint8, device="cuda:0")
        assert L.uaes_gcmsiv_decrypt(bits, key, nonce, ap, alen, C.c_void_p(bad.data_ptr()), n, C.c_void_p(back.data_ptr())) == 0x1A
        assert int((back != src).sum()) == 1                           # released: CTR of the forged text under the received tag
        del src, dst, work, bad, back


@pytest.mark.parametrize("bits", [128, 256])
def test_xts_long_data_units(orc, bits):
    """one data unit of many 256-block chunks (the reference API is one unit per call): the chunk
    tweaks come from the parallel expansion, not from walking the unit"""
    rnd = random.Random(bits + 9)
    # (a unit of up to 1024 chunks = 4 MiB is ONE launch, k_xts_small on many workgroups: both sides of that line, of the
    # 64-chunk range of the sparse-shift product and of a workgroup's 1024 blocks, whole and with a stolen tail)
    for n in [9 * 4096, 64 * 4096, 65 * 4096 + 16, 16384 - 16, 16384 + 16, 256 * 1024 - 16, 256 * 1024, 256 * 1024 + 16 + 5,
              (1 << 20) + 33, (4 << 20) - 16, 4 << 20, (4 << 20) + 3, (4 << 20) + 16, (4 << 20) + 4096 * 63 + 17, 16 << 20]:
        keys, tweak = rnd.randbytes(bits // 4), rnd.randbytes(16)
        data = orc.splitmix(n + 13, n)
        rc, ct = uaes.AES_XTS_encrypt(keys, tweak, data)
        assert (rc, ct) == orc.xts(keys, tweak, data, True), n
        assert uaes.AES_XTS_decrypt(keys, tweak, ct) == (0, data), n
    # several long units in one call
    keys = rnd.randbytes(bits // 4)
    sb, ns = 200 * 4096 + 48, 3
    data = orc.splitmix(77, sb * ns)
    rc, ct = uaes.xts_sectors(keys, 1234567, sb, data, encrypt=True)
    assert (rc, ct) == orc.xts_sectors(keys, 1234567, sb, data, True)
    assert uaes.xts_sectors(keys, 1234567, sb, ct, encrypt=False) == (0, data)


@pytest.mark.parametrize("bits", [128, 256])
def test_gcm_stream_equals_one_shot(orc, bits):
    """a message fed in pieces (tiny, odd multiples of 16, > 32768 blocks to reach the bulk GHASH
    levels, ragged last piece) gives the bytes and the tag of one AES_GCM_encrypt call"""
    rnd = random.Random(bits + 8)
    for pieces, aad_len in [([16], 0), ([5], 3), ([16, 16, 7], 20), ([4096, 16, 65536, 1], 0),
                            ([(1 << 20) + 16, 48, (600 << 10), 12345], 4097), ([0, 32, 0, 16], 16),
                            ([40 << 10, 2 << 20, 64, 5 << 20, 2 << 20, 100], 1),    # level plans change back and forth
                            # pieces long enough for the one-pass kernel (8 MiB on 256 CUs) between short ones: its
                            # tables are made once per stream, a later short piece's setup must leave them alone
                            ([9 << 20, 4096, (17 << 20) + 16, 16, (8 << 20) + 5], 33), ([(12 << 20) + 7], 0),
                            # pieces of 16 .. 128 MiB: two phases (bulk CTR kernel + hash-only chunk workgroups), 2^k blocks too
                            ([16 << 20, (33 << 20) + 48, 32, 32 << 20, (20 << 20) + 9], 5)][: 11 if bits == 128 else 10]:
        key, nonce, aad = rnd.randbytes(bits // 8), rnd.randbytes(12), rnd.randbytes(aad_len)
        n = sum(pieces)
        data = orc.splitmix(n + 11, n)
        want = orc.gcm_encrypt(key, nonce, aad, data)
        st = uaes.GcmStream(key, nonce, aad)
        got, off = b"", 0
        for p in pieces:
            got += st.update(data[off:off + p])
            off += p
        tag = st.finish()
        assert got + tag == want, pieces
        st = uaes.GcmStream(key, nonce, aad, decrypt=True)
        back, off = b"", 0
        for p in pieces:
            back += st.update(want[off:off + p])
            off += p
        assert st.finish(want[n:]) == 0 and back == data
        st = uaes.GcmStream(key, nonce, aad, decrypt=True)
        st.update(want[:n])
        assert st.finish(bytes([want[n] ^ 1]) + want[n + 1:]) == 0x1A
    st = uaes.GcmStream(bytes(16), bytes(12))
    st.update(b"abc")
    with pytest.raises(uaes.EngineError):
        st.update(b"x" * 16)                     # a ragged piece closes the stream


def test_ocb_openssl_vectors():
    cases = ocb_cases(128)
    assert len(cases) == 16
    for c in cases:
        assert uaes.AES_OCB_encrypt(c["key"], c["iv"], c["aad"], c["pt"]) == c["ct"]
        assert uaes.AES_OCB_decrypt(c["key"], c["iv"], c["aad"], c["ct"]) == (0, c["pt"])


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ocb_vs_oracle(orc, bits):
    """chunk (256-block) and run (16-chunk) edges of the Gray-code offsets, ragged tails, long AAD"""
    rnd = random.Random(bits + 7)
    # up to 1024 whole blocks and 64 KiB of AAD a call is ONE launch (k_ocb_small); both sides of that limit
    sizes = [0, 1, 15, 16, 17, 63 * 16, 64 * 16 + 1, 255 * 16, 256 * 16, 256 * 16 + 5, 1023 * 16 + 15, 1024 * 16, 1024 * 16 + 15,
             1025 * 16, 4095 * 16, 4096 * 16, 4097 * 16 + 3, (1 << 20) + 16, (3 << 20) + 7, (16 << 20) - 16]
    for n in sizes:
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(11) + bytes([rnd.randrange(256)])
        aad = rnd.randbytes(rnd.choice([0, 1, 16, 20, 4097, 65536, 65537, 70000]))
        data = orc.splitmix(n + 9, n)
        ct = uaes.AES_OCB_encrypt(key, nonce, aad, data)
        assert ct == orc.ocb_encrypt(key, nonce, aad, data), n
        assert uaes.AES_OCB_decrypt(key, nonce, aad, ct) == (0, data), n
        bad = bytearray(ct)
        bad[n // 2 if n else -1] ^= 0x10
        rc, txt = uaes.AES_OCB_decrypt(key, nonce, aad, bytes(bad))
        assert (rc, txt) == orc.ocb_decrypt(key, nonce, aad, bytes(bad)) and rc == 0x1A


@pytest.mark.parametrize("bits", [128, 256])
def test_ocb_one_launch_paths(orc, bits):
    """OCB is one launch per call: every workgroup derives the L table and Offset_0 itself and the LAST workgroup to
    arrive (a counter word that must be back at zero for the next call) hashes the associated data and makes the
    tag; a decrypting launch encrypts through a plain Te0, or brings the encryption tables back for long
    associated data.  Text sizes on both sides of the one-workgroup limit x associated data on both sides of the
    plain-table limit (16 KiB) and of the one-workgroup limit (64 KiB), no text at all, the same lane again and
    again, host and device pointers (the *_dev scratch slots have their own counter word)."""
    import torch
    rnd = random.Random(bits + 77)
    st = torch.cuda.current_stream()
    # (associated data from 8192 whole blocks on is hashed by ALL workgroups as a second walk, the decrypting launch
    # swapping its tables in between: 131072 bytes and more, with and without a ragged last block, text or none)
    for n in (0, 100, 16 * 1024, 16 * 1025 + 3, (5 << 20) + 1, (40 << 20) + 16):
        for alen in (0, 15, 16384, 16385, 65536, 65537, 131071, 131072, 200000, (3 << 20) + 5):
            if n > (5 << 20) and alen not in (0, 16385, (3 << 20) + 5):
                continue
            key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
            aad, data = rnd.randbytes(alen), orc.splitmix(n + alen, n)
            want = orc.ocb_encrypt(key, nonce, aad, data)
            ct = uaes.AES_OCB_encrypt(key, nonce, aad, data)
            assert ct == want, (n, alen)
            assert uaes.AES_OCB_decrypt(key, nonce, aad, ct) == (0, data), (n, alen)
            bad = bytearray(ct)
            bad[-1] ^= 1
            assert uaes.AES_OCB_decrypt(key, nonce, aad, bytes(bad))[0] == 0x1A
            if bits == 128 and n % 16 == 0 and n:
                d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).cuda() if alen else None
                d_out = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
                status = torch.full((1,), -1, dtype=torch.int32, device="cuda")
                for _ in range(2):
                    uaes.ocb_dev(key, nonce, d_aad, d_in, n, d_out, stream=st)
                torch.cuda.synchronize()
                assert bytes(d_out.cpu().numpy()) == want, (n, alen)
                back = torch.zeros(n, dtype=torch.uint8, device="cuda")
                uaes.ocb_dev(key, nonce, d_aad, d_out, n, back, decrypt=True, status=status, stream=st)
                torch.cuda.synchronize()
                assert int(status.item()) == 0 and bytes(back.cpu().numpy()) == data


def _chunk_folds():
    out = C.c_uint(0)
    assert uaes.engine().uaes_debug_gcm_chunk_folds(C.byref(out)) == 0
    return out.value


@pytest.mark.parametrize("look", ["default", "none"])
def test_gcm_medium_texts_one_launch(orc, look):
    """32 KiB .. 8 MiB: the chunk workgroups and a PREPARING workgroup in one launch (k_gcm_chunks<.., FOLD>); whoever of
    them arrives last on a counter word folds the chunk hashes and makes the tag, and puts the word back to zero for the
    next call -- the same lane / stream slot / key context again and again, sizes on both sides of the arrangement's
    limits (one or two positions per thread; as many chunk workgroups as there are CUs), AAD, encryption and both
    decryption orders.  look = "default": the preparing workgroup looks at the counter for up to 1 ms before it counts
    in, so on this idle device it is the last and folds with its own tables (no fold by a chunk workgroup);
    look = "none" (uaes_debug_gcm_look(0)): it counts in at once, a chunk workgroup is usually the last and the whole
    fold -- tables included -- runs there (nobody waits, nothing traps, either order is exact)."""
    import torch
    L = uaes.engine()
    L.uaes_debug_gcm_look(0 if look == "none" else 100000)
    folds0 = _chunk_folds()
    try:
        _medium_texts(orc, torch)
        _siv_and_stream_pieces(orc)
    finally:
        L.uaes_debug_gcm_look(100000)
    folds = _chunk_folds() - folds0
    if look == "none":
        assert folds > 20, "with no look the fold must have run in a chunk workgroup many times (%d)" % folds
    else:
        # a statement about the FAST path, not about correctness (either order is exact): on an idle device the preparing
        # workgroup is the last to arrive -- a handful of exceptions (a box that was briefly busy) are tolerated
        assert folds <= 5, "on an idle device the preparing workgroup is the last to arrive (%d folds elsewhere)" % folds


def _siv_and_stream_pieces(orc):
    """the other users of the one-launch arrangement: long GCM-SIV messages (POLYVAL by hash-only chunk workgroups, the
    tag made by the fold) and the pieces of a streamed GCM message (the fold adds the piece's hash to the running one)"""
    rnd = random.Random(99)
    key, nonce = rnd.randbytes(32), rnd.randbytes(12)
    for n in (70000, (1 << 20) + 5, (6 << 20) + 48):
        data, aad = orc.splitmix(n + 9, n), rnd.randbytes(33)
        ws = orc.gcmsiv_encrypt(key, nonce, aad, data)
        for _ in range(2):
            assert uaes.GCM_SIV_encrypt(key, nonce, aad, data) == ws, n
        assert uaes.GCM_SIV_decrypt(key, nonce, aad, ws) == (0, data), n
    key = rnd.randbytes(16)
    for pieces in ((1 << 20, 1 << 20, 4096 + 5), (262144, 65536, 1 << 21, 16)):
        data = orc.splitmix(sum(pieces), sum(pieces))
        aad = rnd.randbytes(21)
        want = orc.gcm_encrypt(key, nonce, aad, data)
        stm = uaes.GcmStream(key, nonce, aad)
        out, o = bytearray(), 0
        for m in pieces:
            out += stm.update(data[o:o + m])
            o += m
        tag = stm.finish()
        assert bytes(out) + bytes(tag) == want, pieces


def _medium_texts(orc, torch):
    rnd = random.Random(4242)
    key, st = rnd.randbytes(16), torch.cuda.current_stream()
    gk = uaes.GcmKey(key)
    for n in (40000, 65536 + 5, (1 << 20) - 16, (4 << 20) - 32, (4 << 20) - 16, (4 << 20), (4 << 20) + 4096, (8 << 20) - 48, (8 << 20) + 1):
        nonce, aad = rnd.randbytes(12), rnd.randbytes(rnd.choice([0, 20, 4096]))
        data = orc.splitmix(n, n)
        want = orc.gcm_encrypt(key, nonce, aad, data)
        for _ in range(2):
            assert uaes.AES_GCM_encrypt(key, nonce, aad, data) == want, n
            assert gk.encrypt(nonce, aad, data) == want, n
        assert uaes.AES_GCM_decrypt(key, nonce, aad, want) == (0, data), n
        assert gk.decrypt(nonce, aad, want) == (0, data), n
        bad = bytearray(want)
        bad[n // 3] ^= 4
        assert uaes.AES_GCM_decrypt(key, nonce, aad, bytes(bad))[0] == 0x1A
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).cuda() if aad else None
        d_out = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            uaes.gcm_encrypt_dev(key, nonce, d_aad, d_in, n, d_out, stream=st)
        torch.cuda.synchronize()
        assert bytes(d_out.cpu().numpy()) == want, n
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        uaes.gcm_decrypt_dev(key, nonce, d_aad, d_out, n, back, status, stream=st)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and bytes(back.cpu().numpy()) == data


def test_gcm_lane_key_cache(orc):
    """A thread that sends message after message under ONE key through the drop-in calls gets that key's table set
    built in its lane's scratch at the eighth call in a row and runs as on a key context from then on
    (lane_gcm_keyed).  Whatever else writes the scratch in between -- XTS chunk tweaks, OCB rows, GHASH under another
    H, a GCM call under another key, J0 = GHASH(nonce) of another key, the truncated-tag order -- must end that:
    every result against the oracle, sizes of all three GCM arrangements, before and after each disturbance."""
    rnd = random.Random(20260930)
    ka, kb = rnd.randbytes(16), rnd.randbytes(32)
    sizes = [0, 16, 100, 4096, 70000, (1 << 20) + 3, (9 << 20) + 16]

    def check(key, n, nonce_len=12, tag_len=16, alen=5):
        nonce, aad, data = rnd.randbytes(nonce_len), rnd.randbytes(alen), orc.splitmix(n + nonce_len, n)
        want = orc.gcm_encrypt(key, nonce, aad, data, tag_len)
        assert uaes.AES_GCM_encrypt(key, nonce, aad, data, tag_len=tag_len) == want, (n, nonce_len, tag_len)
        assert uaes.AES_GCM_decrypt(key, nonce, aad, want, tag_len=tag_len) == (0, data), (n, nonce_len, tag_len)
        bad = bytearray(want)
        bad[-1] ^= 2
        assert uaes.AES_GCM_decrypt(key, nonce, aad, bytes(bad), tag_len=tag_len)[0] == 0x1A

    for i in range(14):                                       # 28 calls in a row under ka: the tables come at the eighth
        check(ka, sizes[i % len(sizes)])
    disturbances = [
        lambda: uaes.xts_sectors(bytes(range(32)), 7, 512, orc.splitmix(1, 512 * 40), True),
        lambda: uaes.AES_OCB_encrypt(kb[:16], bytes(12), b"", orc.splitmix(2, 70000)),
        lambda: uaes.ghash(rnd.randbytes(16), b"aad", orc.splitmix(3, 50000)),
        lambda: check(kb, 70000),                             # another key, once
        lambda: check(kb, 100, nonce_len=8),                  # ... whose J0 is a GHASH on the lane's scratch
        lambda: check(kb, 4096, tag_len=12),                  # ... the truncated-tag order (tag first, then CTR)
        lambda: uaes.GCM_SIV_encrypt(ka, bytes(12), b"", orc.splitmix(4, 30000)),
    ]
    for d in disturbances:
        d()
        for i in range(10):
            check(ka, sizes[(i + 3) % len(sizes)])
    for i in range(10):                                       # other nonce and tag lengths on the cached key
        check(ka, sizes[i % len(sizes)], nonce_len=rnd.choice([1, 8, 13, 60]), tag_len=rnd.choice([4, 12, 16]))


def test_main_c_kats(golden_dir):
    for k in load(golden_dir, "main_kats.json"):
        key, pt, exp = bytes.fromhex(k["key"]), bytes.fromhex(k["pt"]), bytes.fromhex(k["expect"])
        if k["mode"] == "ecb":
            assert uaes.AES_ECB_encrypt(key, pt) == exp
            rc, back = uaes.AES_ECB_decrypt(key, exp)
            assert rc == 0 and back[: len(pt)] == pt
        elif k["mode"] == "ctr":
            iv = bytes.fromhex(k["iv"])
            assert uaes.AES_CTR_encrypt(key, iv, pt) == exp and uaes.AES_CTR_decrypt(key, iv, exp) == pt
        elif k["mode"] == "xts":
            tw = bytes.fromhex(k["tweak"])
            assert uaes.AES_XTS_encrypt(key, tw, pt) == (0, exp)
            assert uaes.AES_XTS_decrypt(key, tw, exp) == (0, pt)
        elif k["mode"] == "gcm":
            n, a = bytes.fromhex(k["nonce"]), bytes.fromhex(k["aad"])
            assert uaes.AES_GCM_encrypt(key, n, a, pt) == exp
            assert uaes.AES_GCM_decrypt(key, n, a, exp) == (0, pt)
        elif k["mode"] == "cmac":
            assert uaes.AES_CMAC(key, pt) == exp
        elif k["mode"] == "ccm":
            n, a = bytes.fromhex(k["nonce"]), bytes.fromhex(k["aad"])
            assert uaes.AES_CCM_encrypt(key, n, a, pt) == exp
            assert uaes.AES_CCM_decrypt(key, n, a, exp) == (0, pt)
        elif k["mode"] == "ocb":
            n, a = bytes.fromhex(k["nonce"]), bytes.fromhex(k["aad"])
            assert uaes.AES_OCB_encrypt(key, n, a, pt) == exp
            assert uaes.AES_OCB_decrypt(key, n, a, exp) == (0, pt)
        elif k["mode"] == "gcmsiv":
            n, a = bytes.fromhex(k["nonce"]), bytes.fromhex(k["aad"])
            assert uaes.GCM_SIV_encrypt(key, n, a, pt) == exp
            assert uaes.GCM_SIV_decrypt(key, n, a, exp) == (0, pt)
        elif k["mode"] == "cbc":
            assert uaes.AES_CBC_encrypt(key, bytes.fromhex(k["iv"]), pt) == (0, exp)
            assert uaes.AES_CBC_decrypt(key, bytes.fromhex(k["iv"]), exp) == (0, pt)
        elif k["mode"] == "cfb":
            assert uaes.AES_CFB_encrypt(key, bytes.fromhex(k["iv"]), pt) == exp
            assert uaes.AES_CFB_decrypt(key, bytes.fromhex(k["iv"]), exp) == pt
        else:
            assert uaes.AES_OFB_encrypt(key, bytes.fromhex(k["iv"]), pt) == exp
            assert uaes.AES_OFB_decrypt(key, bytes.fromhex(k["iv"]), exp) == pt


def test_reference_generated_vectors(orc, golden_dir):
    for v in load(golden_dir, "ref_vectors.json"):
        n = v["len"]
        data = orc.splitmix(v["seed"], n)
        key = bytes.fromhex(v["key"])
        if v["mode"] == "ecb":
            ct = uaes.AES_ECB_encrypt(key, data)
            check_out(ct, v["out"])
            rc, back = uaes.AES_ECB_decrypt(key, ct[:n] if n % 16 else ct)
            assert rc == v["dec_rc"]
            if n % 16 == 0:
                assert back == data
        elif v["mode"] == "ctr":
            check_out(uaes.AES_CTR_encrypt(key, bytes.fromhex(v["iv"]), data), v["out"])
        elif v["mode"] == "xts":
            rc, ct = uaes.AES_XTS_encrypt(key, bytes.fromhex(v["tweak"]), data, prefill=0xCC)
            assert rc == v["rc"]
            if rc == 0:
                check_out(ct, v["out"])
                assert uaes.AES_XTS_decrypt(key, bytes.fromhex(v["tweak"]), ct) == (0, data)
            else:
                assert ct == b"\xcc" * n                       # N5: untouched
        elif v["mode"] == "gcm":
            nonce, aad = bytes.fromhex(v["nonce"]), bytes.fromhex(v["aad"])
            ct = uaes.AES_GCM_encrypt(key, nonce, aad, data)
            check_out(ct, v["out"])
            assert uaes.AES_GCM_decrypt(key, nonce, aad, ct) == (0, data)
        elif v["mode"] == "cmac":
            check_out(uaes.AES_CMAC(key, data), v["out"])
        elif v["mode"] == "ccm":
            nonce, aad = bytes.fromhex(v["nonce"]), bytes.fromhex(v["aad"])
            ct = uaes.AES_CCM_encrypt(key, nonce, aad, data)
            check_out(ct, v["out"])
            assert uaes.AES_CCM_decrypt(key, nonce, aad, ct) == (0, data)
        elif v["mode"] == "ocb":
            nonce, aad = bytes.fromhex(v["nonce"]), bytes.fromhex(v["aad"])
            ct = uaes.AES_OCB_encrypt(key, nonce, aad, data)
            check_out(ct, v["out"])
            assert uaes.AES_OCB_decrypt(key, nonce, aad, ct) == (0, data)
        elif v["mode"] == "gcmsiv":
            nonce, aad = bytes.fromhex(v["nonce"]), bytes.fromhex(v["aad"])
            ct = uaes.GCM_SIV_encrypt(key, nonce, aad, data)
            check_out(ct, v["out"])
            assert uaes.GCM_SIV_decrypt(key, nonce, aad, ct) == (0, data)
        elif v["mode"] == "cbc":
            rc, ct = uaes.AES_CBC_encrypt(key, bytes.fromhex(v["iv"]), data, prefill=0xCC)
            assert rc == v["rc"]
            if rc == 0:
                check_out(ct, v["out"])
                assert uaes.AES_CBC_decrypt(key, bytes.fromhex(v["iv"]), ct) == (0, data)
            else:
                assert ct == b"\xcc" * n
        elif v["mode"] == "cfb":
            ct = uaes.AES_CFB_encrypt(key, bytes.fromhex(v["iv"]), data)
            check_out(ct, v["out"])
            assert uaes.AES_CFB_decrypt(key, bytes.fromhex(v["iv"]), ct) == data
        else:
            check_out(uaes.AES_OFB_encrypt(key, bytes.fromhex(v["iv"]), data), v["out"])


# ---- against the oracle on seeded inputs -------------------------------------------
SIZES = [0, 1, 15, 16, 17, 100, 4095, 4096, 4097, 65536 + 3, (1 << 20) + 16, (4 << 20) + 9]


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ecb_ctr_vs_oracle(orc, bits):
    rnd = random.Random(bits)
    for n in SIZES:
        key, iv = rnd.randbytes(bits // 8), rnd.randbytes(12)
        data = orc.splitmix(n + bits, n)
        ct = uaes.AES_ECB_encrypt(key, data)
        assert ct == orc.ecb_encrypt(key, data)
        assert uaes.AES_ECB_decrypt(key, ct) == orc.ecb_decrypt(key, ct)
        assert uaes.AES_ECB_decrypt(key, data) == orc.ecb_decrypt(key, data)      # ragged: 0x1D + tail copied
        assert uaes.AES_CTR_encrypt(key, iv, data) == orc.ctr_encrypt(key, iv, data)
    # N2: 56-bit counter carry / wrap, and shard offsets
    key = rnd.randbytes(bits // 8)
    c0 = bytes.fromhex("0011223344556677a8fffffffffffff0")
    data = orc.splitmix(9, 4096 * 3 + 7)
    whole = uaes.ctr_xcrypt_at(key, c0, 0, data)
    assert whole == orc.ctr_xcrypt_at(key, c0, 0, data)
    assert whole[4096:] == uaes.ctr_xcrypt_at(key, c0, 256, data[4096:])
    c1 = bytes.fromhex("00112233445566778800000000ffffff")            # byte-12 carry
    assert uaes.ctr_xcrypt_at(key, c1, 0xFFFFFF00, data) == orc.ctr_xcrypt_at(key, c1, 0xFFFFFF00, data)


def test_ecb_padding_modes(orc, golden_dir):
    """AES_PADDING 1 (PKCS#7) and 2 (ISO/IEC 7816-4) of padBlock, micro_aes.c:610-621: the
    reference-generated vectors (incl. main.c's AES-192 answer), the oracle over edge lengths
    for every key size, and the compat symbols a caller built with -DAES_PADDING=n binds to."""
    for v in load(golden_dir, "ecb_padding_vectors.json"):
        key = bytes.fromhex(v["key"])
        data = bytes.fromhex(v["pt"]) if "pt" in v else orc.splitmix(v["seed"], (v["len"] + 7) // 8 * 8)[: v["len"]]
        check_out(uaes.AES_ECB_encrypt(key, data, padding=v["padding"]),
                  v["out"] if isinstance(v["out"], dict) else {"hex": v["out"]})
    rnd = random.Random(5150)
    for bits in (128, 192, 256):
        lib = C.CDLL(uaes.lib_path("libmicro_aes_hip_%d.so" % bits))
        for padding, sym in ((1, "AES_ECB_encrypt_pkcs7"), (2, "AES_ECB_encrypt_iso7816")):
            for n in (0, 1, 15, 16, 17, 4095, 4096, 65537, (4 << 20) + 5, 4 << 20):
                key, data = rnd.randbytes(bits // 8), rnd.randbytes(n)
                want = orc.ecb_encrypt(key, data, padding=padding)
                assert uaes.AES_ECB_encrypt(key, data, padding=padding) == want
                if n <= 65537:
                    out = (C.c_uint8 * len(want))()
                    getattr(lib, sym)(key, data, C.c_size_t(n), out)
                    assert bytes(out) == want
                    rc, back = uaes.AES_ECB_decrypt(key, want)
                    assert rc == 0 and back[:n] == data and len(back) == len(want)


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_xts_vs_oracle(orc, bits):
    rnd = random.Random(bits + 1)
    # one unit per call: up to 16 KiB is ONE launch (k_xts_small computes Enc_key2(tweak) itself), longer units
    # take the tweak pre-pass; sizes on both sides of its 64-block runs, 256-block chunks and its upper limit
    for n in [16, 17, 31, 32, 33, 100, 1008, 1024, 1025, 1040, 4080, 4096, 4097, 4111, 4112, 8192 + 16, 16368, 16384 - 1,
              16384, 16384 + 1, 16384 + 15, 16384 + 16, 16384 + 17, 65536, 65536 + 1, (1 << 20) + 5]:
        keys, tw = rnd.randbytes(bits // 4), rnd.randbytes(16)
        data = orc.splitmix(n, n)
        rc, ct = uaes.AES_XTS_encrypt(keys, tw, data)
        assert (rc, ct) == orc.xts(keys, tw, data, True)
        assert uaes.AES_XTS_decrypt(keys, tw, ct) == (0, data)
    keys = rnd.randbytes(bits // 4)
    assert uaes.AES_XTS_encrypt(keys, None, b"Q" * 48) == uaes.AES_XTS_encrypt(keys, bytes(16), b"Q" * 48)
    # batched data units, including ciphertext stealing in every unit
    # (up to 4 MiB of whole-block units is ONE launch in which every wave encrypts its unit's tweak itself; beyond
    # that, and for ragged units, the tweak pre-pass: both, on both sides of the limit)
    for sector, count, first in [(4096, 33, 0), (512, 100, (1 << 40) + 7), (528, 9, 5), (16, 70, 1), (25, 40, 2),
                                 (4096, 1, (1 << 63) + 12345), (4100, 1, 77), (16384, 1, 3),       # one unit by sector id
                                 (4096, 1024, (1 << 32) - 5), (4096, 1025, 9), (48, 1300, 0), (1024 + 16, 61, 3),
                                 (65536, 2, 0xFFFFFFFFFFFFFFFE), (1 << 20, 5, 11), (32, 2, 0)]:
        vol = orc.splitmix(sector, sector * count)
        rc, ct = uaes.xts_sectors(keys, first, sector, vol, True)
        assert (rc, ct) == orc.xts_sectors(keys, first, sector, vol, True)
        assert uaes.xts_sectors(keys, first, sector, ct, False) == (0, vol)


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_vs_oracle(orc, bits):
    rnd = random.Random(bits + 2)
    for n in [0, 1, 16, 17, 4096, 65536 + 3, (256 << 10) - 16, (256 << 10), (1 << 20) + 16]:
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
        aad = rnd.randbytes(rnd.choice([0, 1, 16, 20, 4097]))
        data = orc.splitmix(n + 1, n)
        ct = uaes.AES_GCM_encrypt(key, nonce, aad, data)
        assert ct == orc.gcm_encrypt(key, nonce, aad, data), n
        assert uaes.AES_GCM_decrypt(key, nonce, aad, ct) == (0, data)
        bad = bytearray(ct)
        bad[rnd.randrange(len(bad))] ^= 0x40
        assert uaes.AES_GCM_decrypt(key, nonce, aad, bytes(bad), prefill=0xCC) == (0x1A, b"\xcc" * n)   # N7


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_cmac_ccm_vs_oracle(orc, bits):
    rnd = random.Random(bits + 3)
    # (CCM texts up to 256 bytes take the one-launch kernel in which the counter blocks share the MAC's wave, longer
    # ones the MAC kernel + the CTR kernel: both sides of that line, whole and ragged)
    for n in [0, 1, 15, 16, 17, 32, 100, 240, 250, 255, 256, 257, 272, 300, 4096, 65536 + 3]:
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(11)
        data = orc.splitmix(n + 5, n)
        assert uaes.AES_CMAC(key, data) == orc.cmac(key, data), n
        aad = rnd.randbytes(rnd.choice([0, 1, 14, 15, 16, 70000 if n == 100 else 31]))
        ct = uaes.AES_CCM_encrypt(key, nonce, aad, data)
        assert ct == orc.ccm_encrypt(key, nonce, aad, data), n
        assert uaes.AES_CCM_decrypt(key, nonce, aad, ct) == (0, data)
        bad = bytearray(ct)
        bad[-1] ^= 1
        # the reference decrypts first and leaves the text in place on a mismatch
        assert uaes.AES_CCM_decrypt(key, nonce, aad, bytes(bad)) == (0x1A, data)


@pytest.mark.parametrize("total,world,alen", [(0, 2, 5), (16, 2, 0), (1000, 3, 20), ((3 << 20) + 5, 4, 33),
                                                (40 << 20, 8, 0)])
def test_sharded_gcm_partials(orc, total, world, alen):
    """multi-GPU GCM on one device: each 'rank' runs CTR + uaes_gcm_partial_dev on its
    shard; the XOR of the 16-byte shares must be the single-call tag"""
    import torch
    import micro_aes_amd.sharding as sh
    rnd = random.Random(total + world)
    key, nonce, aad = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(alen)
    data = orc.splitmix(total + 3, total)
    want = uaes.AES_GCM_encrypt(key, nonce, aad, data)
    if total <= (4 << 20):
        assert want == orc.gcm_encrypt(key, nonce, aad, data)
    d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0") if alen else None
    shares, pieces = [], []
    for rank in range(world):
        start, n, _ = sh.gcm_shard_roles(total, rank, world)
        src = torch.frombuffer(bytearray(data[start:start + n] + bytes(16)), dtype=torch.uint8).to("cuda:0")
        dst = torch.zeros_like(src)
        tag = sh.gcm_encrypt_sharded(key, nonce, d_aad, alen, total, src, dst, rank, world,
                                     gather=lambda share: shares.append(share) or list(shares))
        torch.cuda.synchronize()
        pieces.append(bytes(dst[:n].cpu().numpy()))
    assert b"".join(pieces) == want[:-16]
    assert tag == want[-16:]


def test_sharded_gcm_partials_on_distinct_devices(orc):
    """the same exchange with every shard on ITS OWN GPU (per-device contexts and scratch, shares gathered
    through the host): needs a multi-GPU box"""
    import torch
    import micro_aes_amd.sharding as sh
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs 2 GPUs: this box has %d" % world)
    rnd = random.Random(77)
    total, alen = (24 << 20) + 16 * 3 + 5, 19
    key, nonce, aad = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(alen)
    data = orc.splitmix(total + 3, total)
    want = uaes.AES_GCM_encrypt(key, nonce, aad, data)
    shares, pieces = [], []
    for rank in range(world):
        dev = torch.device("cuda", rank)
        with torch.cuda.device(dev):
            d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to(dev)
            start, n, _ = sh.gcm_shard_roles(total, rank, world)
            src = torch.frombuffer(bytearray(data[start:start + n] + bytes(16)), dtype=torch.uint8).to(dev)
            dst = torch.zeros_like(src)
            tag = sh.gcm_encrypt_sharded(key, nonce, d_aad, alen, total, src, dst, rank, world,
                                         gather=lambda share: shares.append(share) or list(shares))
            torch.cuda.synchronize(dev)
            pieces.append(bytes(dst[:n].cpu().numpy()))
    assert b"".join(pieces) == want[:-16]
    assert tag == want[-16:]


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_feedback_modes_vs_oracle(orc, bits):
    """CBC (CS3) / CFB decrypt are block-parallel kernels: check them on large inputs too"""
    rnd = random.Random(bits + 4)
    for n in [16, 17, 31, 32, 33, 48, 100, 4096, 4097, (1 << 20) + 5, 4 << 20]:
        key, iv = rnd.randbytes(bits // 8), rnd.randbytes(16)
        ct_in = orc.splitmix(n + 9, n)                 # decrypt directions work on any input
        assert uaes.AES_CBC_decrypt(key, iv, ct_in) == orc.cbc(key, iv, ct_in, False), n
        assert uaes.AES_CFB_decrypt(key, iv, ct_in) == orc.cfb(key, iv, ct_in, False), n
        if n <= 4097:                                   # serial directions: ~1.5 us per block
            assert uaes.AES_CBC_encrypt(key, iv, ct_in) == orc.cbc(key, iv, ct_in, True), n
            assert uaes.AES_CFB_encrypt(key, iv, ct_in) == orc.cfb(key, iv, ct_in, True), n
            assert uaes.AES_OFB_encrypt(key, iv, ct_in) == orc.ofb(key, iv, ct_in), n
    assert uaes.AES_CBC_encrypt(key, iv, b"123456789012345", prefill=0xCC) == (1, b"\xcc" * 15)
    assert uaes.AES_CBC_decrypt(key, iv, b"123456789012345", prefill=0xCC) == (1, b"\xcc" * 15)
    assert uaes.AES_CFB_encrypt(key, iv, b"") == b"" and uaes.AES_OFB_encrypt(key, iv, b"abc") == orc.ofb(key, iv, b"abc")


def test_build_variant_vectors_cbc_without_cts_and_other_ctr_constants(orc, golden_dir):
    """The reference's other compile-time builds of two in-scope functions, through the C ABI and through the
    compat libraries' entry points that include/micro_aes.h binds such a caller to:
    CTS 0 (micro_aes.h:56): AES_CBC_encrypt pads its last chunk like ECB (AES_PADDING 0/1/2, micro_aes.c:727-733),
    any length; AES_CBC_decrypt wants whole blocks (:761) -- block-parallel here -- and keeps the padding;
    CTR_IV_LENGTH / CTR_START_VALUE (micro_aes.h:98-99, micro_aes.c:968-971).
    Against vectors made by reference builds with those switches (incl. main.c's own CTS 0 answer) and the oracle."""
    vecs = load(golden_dir, "build_variant_vectors.json")
    for v in vecs["cbc_nocts"]:
        key, iv = bytes.fromhex(v["key"]), bytes.fromhex(v["iv"])
        data = bytes.fromhex(v["pt"]) if "pt" in v else orc.splitmix(v["seed"], (v["len"] + 7) // 8 * 8)[: v["len"]]
        rc, ct = uaes.AES_CBC_encrypt(key, iv, data, cts=False, padding=v["padding"])
        assert rc == 0
        check_out(ct, v["out"])
        rc, back = uaes.AES_CBC_decrypt(key, iv, ct, cts=False)
        assert rc == 0 and back[: len(data)] == data and back == orc.cbc_nocts(key, iv, ct, False)[1]
        if len(data) % 16:
            assert uaes.AES_CBC_decrypt(key, iv, data, prefill=0xCC, cts=False) == (1, b"\xcc" * len(data))
    for v in vecs["ctr_iv"]:
        key, iv = bytes.fromhex(v["key"]), bytes.fromhex(v["iv"])
        data = orc.splitmix(v["seed"], (v["len"] + 7) // 8 * 8)[: v["len"]]
        check_out(uaes.AES_CTR_encrypt(key, iv, data, iv_length=v["iv_length"], start_value=v["start_value"]), v["out"])
    rnd = random.Random(0xC750)
    sym = {0: "AES_CBC_encrypt_nocts", 1: "AES_CBC_encrypt_nocts_pkcs7", 2: "AES_CBC_encrypt_nocts_iso7816"}
    for bits in (128, 192, 256):
        lib = C.CDLL(uaes.lib_path("libmicro_aes_hip_%d.so" % bits))
        lib.AES_CTR_encrypt_iv.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.AES_CTR_encrypt_iv.restype = None
        for padding in (0, 1, 2):
            f = getattr(lib, sym[padding])
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
            f.restype = C.c_char
            lib.AES_CBC_decrypt_nocts.argtypes = f.argtypes
            lib.AES_CBC_decrypt_nocts.restype = C.c_char
            for n in (0, 5, 16, 47, 48, 4096, 4099, 65537, (1 << 20) + 16):
                key, iv, data = rnd.randbytes(bits // 8), rnd.randbytes(16), rnd.randbytes(n)
                rc, want = orc.cbc_nocts(key, iv, data, True, padding=padding)
                assert rc == 0
                if n <= 65537:                              # the chain itself: 0.42 us per block
                    assert uaes.AES_CBC_encrypt(key, iv, data, cts=False, padding=padding) == (0, want), (bits, padding, n)
                    out = (C.c_uint8 * max(len(want), 1))()
                    assert ord(f(key, iv, data, n, out)) == 0 and bytes(out)[: len(want)] == want
                # decrypt: the parallel kernel, in place as well
                assert uaes.AES_CBC_decrypt(key, iv, want, cts=False) == orc.cbc_nocts(key, iv, want, False), (bits, padding, n)
                if want:
                    buf = (C.c_uint8 * len(want)).from_buffer_copy(want)
                    assert ord(lib.AES_CBC_decrypt_nocts(key, iv, buf, len(want), buf)) == 0
                    assert bytes(buf) == orc.cbc_nocts(key, iv, want, False)[1]
                    assert ord(lib.AES_CBC_decrypt_nocts(key, iv, buf, len(want) - 1, buf)) == 1      # M_DATALENGTH_ERROR
        for ivl, start in ((0, 0), (4, 0xFFFFFFFF), (8, 0x01A2B3C4), (12, 1), (15, 0x0102030405060708), (16, 2)):
            key, iv, data = rnd.randbytes(bits // 8), rnd.randbytes(ivl), rnd.randbytes(rnd.choice([1, 33, 5000]))
            want = orc.ctr_encrypt_iv(key, iv, start, data)
            assert uaes.AES_CTR_encrypt(key, iv, data, iv_length=ivl, start_value=start) == want, (bits, ivl, start)
            out = (C.c_uint8 * len(data))()
            lib.AES_CTR_encrypt_iv(ivl, start, key, iv, data, len(data), out)
            assert bytes(out) == want
    with pytest.raises(ValueError):
        uaes.AES_CTR_encrypt(bytes(16), bytes(17), b"x", iv_length=17)
    assert uaes.engine().uaes_ctr_xcrypt_iv(128, bytes(16), bytes(17), 17, 1, b"x", 1, (C.c_uint8 * 1)()) == -2
    assert uaes.engine().uaes_cbc_encrypt_padded(128, bytes(16), bytes(16), 3, b"x", 1, (C.c_uint8 * 16)()) == -2


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_batched_chains_equal_the_single_calls(orc, bits):
    """uaes_cbc_encrypt_batch / uaes_cmac_batch: N independent messages, a DPP row of sixteen lanes or one lane each, must give
    exactly what N single calls give (AES_CBC_encrypt with its CS3 swap, AES_CMAC) -- checked against
    the oracle, for one block, two blocks, many blocks, and more messages than one workgroup holds"""
    rnd = random.Random(61 + bits)
    key = rnd.randbytes(bits // 8)
    for nmsg, size in ((1, 16), (3, 32), (5, 48), (1500, 16), (1030, 208), (7, 4096)):
        msgs = [rnd.randbytes(size) for _ in range(nmsg)]
        ivs = [rnd.randbytes(16) for _ in range(nmsg)]
        got = uaes.cbc_encrypt_batch(key, ivs, msgs)
        for i in list(range(min(nmsg, 6))) + [nmsg - 1, nmsg // 2]:
            assert got[i] == orc.cbc(key, ivs[i], msgs[i], True)[1], (nmsg, size, i)
        if nmsg <= 5:
            assert got == [uaes.AES_CBC_encrypt(key, iv, m)[1] for iv, m in zip(ivs, msgs)]
    for nmsg, size in ((1, 0), (4, 1), (3, 16), (5, 17), (1200, 33), (1030, 200), (6, 4096 + 5)):
        msgs = [rnd.randbytes(size) for _ in range(nmsg)]
        got = uaes.cmac_batch(key, msgs)
        for i in list(range(min(nmsg, 6))) + [nmsg - 1, nmsg // 2]:
            assert got[i] == orc.cmac(key, msgs[i]), (nmsg, size, i)
    # up to 81919 messages sixteen lanes walk a message (k_chain_batch_row), from 81920 on one lane does
    # (k_chain_batch): the same messages through both, every one of them compared, and spot checks against the oracle
    # (the one-lane arrangement takes whole groups of four blocks in front of the last two: 208 = 13 blocks)
    # (the message count decides the arrangement, not the key size: the 90 000-message part runs for one key size)
    for size in ((48, 208) if bits == 128 else ()):
        nmsg = 90000
        blob = orc.splitmix(size + bits, nmsg * size)
        msgs = [blob[i * size:(i + 1) * size] for i in range(nmsg)]
        ivs = [blob[(i * 7) % (len(blob) - 16):][:16] for i in range(nmsg)]
        lane, row = uaes.cbc_encrypt_batch(key, ivs, msgs), uaes.cbc_encrypt_batch(key, ivs[:60000], msgs[:60000])
        assert lane[:60000] == row
        for i in (0, 1, 59999, 60000, nmsg - 1):
            assert lane[i] == orc.cbc(key, ivs[i], msgs[i], True)[1]
        for cut in (3, 0):
            mm = [m[:size - cut] for m in msgs]
            lane, row = uaes.cmac_batch(key, mm), uaes.cmac_batch(key, mm[:60000])
            assert lane[:60000] == row
            for i in (0, 59999, nmsg - 1):
                assert lane[i] == orc.cmac(key, mm[i])
    L = uaes.engine()
    assert L.uaes_cbc_encrypt_batch(bits, key, bytes(16), 1, 24, bytes(24), (C.c_uint8 * 24)()) == -2


def test_ghash_kernel_levels(orc):
    """every level plan of the GHASH kernels: direct, one bulk level, two bulk levels"""
    rnd = random.Random(77)
    H = rnd.randbytes(16)
    for nblocks, extra, alen in [(0, 0, 0), (1, 0, 5), (300, 7, 0), (1023, 0, 0), (1024, 3, 16), (32766, 0, 16),
                                 (32767, 0, 0), (32768, 1, 3), (70000, 0, 0), (300000, 5, 33), (1500000, 0, 0),
                                 (3000000, 9, 17)]:
        ct = orc.splitmix(nblocks + 3, nblocks * 16 + extra)
        aad = rnd.randbytes(alen)
        assert uaes.ghash(H, aad, ct) == orc.ghash(H, aad, ct), (nblocks, extra, alen)


# ---- BASELINE.json configurations ----------------------------------------------------
def test_baseline_small_digests(orc, golden_dir):
    d = load(golden_dir, "digests.json")
    key16, key64, nonce = bytes(range(16)), bytes(range(64)), bytes(range(0xF0, 0xFC))
    sha = lambda b: hashlib.sha256(b).hexdigest()
    assert sha(uaes.AES_ECB_encrypt(key16, orc.splitmix(1, 4096))) == d["C1_ecb128_4KiB"]["sha256"]
    assert sha(uaes.AES_CTR_encrypt(key16, nonce, orc.splitmix(2, 1 << 20))) == d["ctr128_1MiB_seed2"]["sha256"]
    rc, secs = uaes.xts_sectors(key64, 0, 4096, orc.splitmix(3, 3 * 4096), True)
    assert rc == 0 and sha(secs) == d["xts256_sectors0_2"]["sha256"]
    ct = uaes.AES_GCM_encrypt(key16, nonce, b"", orc.splitmix(4, 1 << 20))
    assert ct[-16:].hex() == d["gcm128_1MiB_seed4"]["tag"] and sha(ct) == d["gcm128_1MiB_seed4"]["sha256_ct_tag"]


def _device_stream(orc, seed, nbytes):
    import torch
    host = np.empty(nbytes, dtype=np.uint8)
    orc.splitmix_into(seed, host)
    return torch.from_numpy(host).to("cuda:0")


def _sha_of(t):
    h = hashlib.sha256()
    step = 1 << 28
    for o in range(0, t.numel(), step):
        h.update(t[o:o + step].cpu().numpy().tobytes())
    return h.hexdigest()


def test_C2_ctr128_1GiB_device_resident(orc, golden_dir):
    """BASELINE configs[1]: bit-exact digest of the whole 1 GiB stream, plus the
    size-independent properties (involution, shard concatenation)."""
    import torch
    d = load(golden_dir, "digests.json")["C2_ctr128_1GiB_seed2"]
    key, ctr0 = bytes(range(16)), bytes(range(0xF0, 0xFC)) + b"\0\0\0\1"
    src = _device_stream(orc, 2, 1 << 30)
    dst = torch.empty_like(src)
    uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst)
    torch.cuda.synchronize()
    assert bytes(dst[-32:].cpu().numpy()).hex() == d["tail"]
    assert _sha_of(dst) == d["sha256"]
    # shards: 8 x 128 MiB with block offsets == one call
    sh = torch.empty_like(src)
    for g in range(8):
        lo, hi = g << 27, (g + 1) << 27
        uaes.ctr_xcrypt_dev(key, ctr0, lo // 16, src[lo:hi], sh[lo:hi])
    torch.cuda.synchronize()
    assert torch.equal(sh, dst)
    # involution, in place
    uaes.ctr_xcrypt_dev(key, ctr0, 0, dst, dst)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


@pytest.mark.parametrize("mib,extra,low7", [(74, 5, "00ffffffffffe"), (64, 0, "00ffffffffffe"), (20, 16 * 777 + 3, "00ffffffffffe"),
                                            (70, 0, "fffffffffffff0"), (130, 9, "00ffffffff0001")])
def test_ctr_split_between_shared_round_and_generic_kernels(orc, mib, extra, low7):
    """sizes whose last round of 256 KiB chunks is thin: whole rounds run on the shared-round kernel,
    the remainder (with its own counter offset) on the generic one; the start counters carry out of
    32 bits, wrap the 56-bit counter (N2) and change counter bits 40..47 inside the launch"""
    import hashlib
    import torch
    n = (mib << 20) + extra
    key, ctr0 = bytes(range(16, 32)), bytes(range(0xE0, 0xE9)) + bytes.fromhex(low7.rjust(14, "0"))
    src = _device_stream(orc, 21, (n + 7) // 8 * 8)[:n]
    dst = torch.empty_like(src)
    uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n)
    torch.cuda.synchronize()
    want = orc.ctr_xcrypt_at(key, ctr0, 0, bytes(src.cpu().numpy()))
    assert hashlib.sha256(bytes(dst.cpu().numpy())).hexdigest() == hashlib.sha256(want).hexdigest()
    uaes.ctr_xcrypt_dev(key, ctr0, 0, dst, dst, nbytes=n)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


def test_C3_xts256_sectors_device_resident(orc, golden_dir):
    """BASELINE configs[2] at 2^18 sectors (1 GiB; the full 2^20-sector digest is checked by
    tests/test_gpu_baseline_full.py): first sectors against the goldens, round trip, and
    equality with per-range calls."""
    import torch
    d = load(golden_dir, "digests.json")
    keys = bytes(range(64))
    nsec = 1 << 18
    src = _device_stream(orc, 3, nsec * 4096)
    dst = torch.empty_like(src)
    uaes.xts_sectors_dev(keys, 0, 4096, nsec, src, dst, encrypt=True)
    torch.cuda.synchronize()
    assert hashlib.sha256(dst[: 3 * 4096].cpu().numpy().tobytes()).hexdigest() == d["xts256_sectors0_2"]["sha256"]
    tail = 1000
    part = torch.empty(tail * 4096, dtype=torch.uint8, device="cuda:0")
    uaes.xts_sectors_dev(keys, nsec - tail, 4096, tail, src[(nsec - tail) * 4096:], part, encrypt=True)
    torch.cuda.synchronize()
    assert torch.equal(part, dst[(nsec - tail) * 4096:])
    rc, want = orc.xts_sectors(keys, nsec - 2, 4096, bytes(src[(nsec - 2) * 4096:].cpu().numpy()), True)
    assert bytes(dst[(nsec - 2) * 4096:].cpu().numpy()) == want
    uaes.xts_sectors_dev(keys, 0, 4096, nsec, dst, dst, encrypt=False)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_fused_encrypt_pass(orc, bits):
    """Long GCM encryptions take the one-pass kernel (CTR + GHASH of the ciphertext in registers,
    k_gcm_fused): sizes around its threshold (one 32 KiB stripe per workgroup + the head),
    stripe counts that do not divide by the grid (workgroups then differ by one stripe and weight
    their shares differently), ragged tails, AAD of every shape (it becomes initial accumulator
    values), against the oracle and against the two-pass decrypt (which re-hashes the ciphertext
    with the separate GHASH levels)."""
    import torch
    rnd = random.Random(900 + bits)
    key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
    S16 = 16 << 19                                     # one round of the grid in bytes (8 MiB)
    cases = [(S16 + 254 * 16, 0), (S16 + 254 * 16 - 16, 0), (S16 + 254 * 16 + 16, 5), (S16 + 5 * 32768 + 4064 + 7, 13),
             (2 * S16 + 5000, 16), (3 * S16 - 1, 4096 + 7), (2 * S16 + S16 // 2, 1 << 20), (5 * S16 + 12345, 0)]
    for i, (n, alen) in enumerate(cases):
        pt = orc.splitmix(1000 + i, (n + 7) // 8 * 8)[:n]
        aad = rnd.randbytes(alen)
        src = torch.frombuffer(bytearray(pt + bytes(16)), dtype=torch.uint8).to("cuda:0")
        dst = torch.zeros(n + 16 + 16, dtype=torch.uint8, device="cuda:0")
        a = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0") if alen else None
        uaes.gcm_encrypt_dev(key, nonce, a, src, n, dst)
        torch.cuda.synchronize()
        got = bytes(dst[: n + 16].cpu().numpy())
        assert int(dst[n + 16:].sum()) == 0
        if i < 4 or bits == 128:
            want = orc.gcm_encrypt(key, nonce, aad, pt)
            assert got[-16:] == want[-16:], (n, alen)
            assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest(), (n, alen)
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        back = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
        uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and bytes(back[:n].cpu().numpy()) == pt, (n, alen)
        # in place
        work = src.clone()
        big = torch.zeros(n + 32, dtype=torch.uint8, device="cuda:0")
        big[:n] = work[:n]
        uaes.gcm_encrypt_dev(key, nonce, a, big, n, big)
        torch.cuda.synchronize()
        assert bytes(big[: n + 16].cpu().numpy()) == got


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_single_launch_sizes(orc, bits):
    """Messages of at most 2047 GHASH blocks (AAD + text + the length block) run as ONE workgroup that does
    CTR, GHASH and the tag (k_gcm_small): the boundaries of its two 1024-position steps, ragged texts, AAD
    of every shape and alignment, both directions, in place, key context, device pointers; forgeries leave
    the output untouched (N7) although the plaintext was computed in the same launch."""
    import torch
    rnd = random.Random(5100 + bits)
    key = rnd.randbytes(bits // 8)
    k = uaes.GcmKey(key)
    # (text bytes, AAD bytes): nv = ceil(a/16) + ceil(n/16) + 1
    shapes = [(0, 0), (1, 0), (0, 1), (15, 17), (16, 16), (1021 * 16, 0), (1022 * 16, 0), (1022 * 16, 1), (1022 * 16 + 1, 0),
              (1023 * 16, 0), (1023 * 16 - 5, 16), (1024 * 16, 0), (1500 * 16 + 3, 300), (16 * 16, 2029 * 16),
              (2045 * 16, 0), (2045 * 16, 16), (2046 * 16 - 1, 0), (2046 * 16, 0), (0, 2046 * 16), (2046 * 16, 1), (2047 * 16, 0),
              (16384, 13), (4096, 0)]
    for n, alen in shapes:
        # the shapes are hand-picked for the INSIDE of the one-workgroup kernel; the table must still send them there
        assert uaes.plan("gcm", n, alen)[0] == ("gcm.small" if (alen + 15) // 16 + (n + 15) // 16 + 1 <= 2046 else "gcm.chunks"), (n, alen)
        nonce = rnd.randbytes(12)
        data, aad = rnd.randbytes(n), rnd.randbytes(alen)
        want = orc.gcm_encrypt(key, nonce, aad, data)
        assert uaes.AES_GCM_encrypt(key, nonce, aad, data) == want, (n, alen)
        assert k.encrypt(nonce, aad, data) == want, (n, alen)
        assert uaes.AES_GCM_decrypt(key, nonce, aad, want) == (0, data), (n, alen)
        assert k.decrypt(nonce, aad, want) == (0, data), (n, alen)
        bad = bytearray(want)
        bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        assert uaes.AES_GCM_decrypt(key, nonce, aad, bytes(bad), prefill=0xCC) == (0x1A, b"\xcc" * n), (n, alen)
        if alen:
            bad_aad = bytearray(aad)
            bad_aad[rnd.randrange(alen)] ^= 0x80
            assert k.decrypt(nonce, bytes(bad_aad), want, prefill=0xCC) == (0x1A, b"\xcc" * n), (n, alen)
        # device pointers: out of place with guard bytes, then decrypt in place
        src = torch.frombuffer(bytearray(data + bytes(16)), dtype=torch.uint8).to("cuda:0")
        dst = torch.full((n + 48,), 0xA5, dtype=torch.uint8, device="cuda:0")
        a = torch.frombuffer(bytearray(aad + bytes(3)), dtype=torch.uint8).to("cuda:0")[:alen] if alen else None
        uaes.gcm_encrypt_dev(key, nonce, a, src, n, dst)
        torch.cuda.synchronize()
        assert bytes(dst[: n + 16].cpu().numpy()) == want and int((dst[n + 16:] != 0xA5).sum()) == 0, (n, alen)
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        work = dst[: n + 16].clone()
        uaes.gcm_decrypt_dev(key, nonce, a, work, n, work, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0 and bytes(work.cpu().numpy()) == data + want[n:], (n, alen)
        forged = torch.frombuffer(bad, dtype=torch.uint8).to("cuda:0")
        back = torch.full((n + 16,), 0xCC, dtype=torch.uint8, device="cuda:0")
        k.decrypt_dev(nonce, a, forged, n, back, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0x1A and int((back != 0xCC).sum()) == 0, (n, alen)
    k.close()


def test_gcm_long_texts_hashed_by_chunk_workgroups(orc):
    """A decryption that authenticates first (N7) hashes its ciphertext with the chunk workgroups + finisher as far as
    ONE round of workgroups reaches (1024 * 2^k positions per workgroup, k <= 7: 512 MiB on 256 CUs), and an
    encryption does up to 16 MiB (two phases up to 128 MiB).  The sizes the oracle finishes in seconds are in test_gcm_chunk_and_combine_kernels;
    here the longer ones are pinned through the OTHER arrangement: the striped one-pass kernel made the tag (the kernel
    of the C4 digest test), the chunk workgroups must accept it, reject one flipped bit anywhere and leave the output
    alone then -- one-shot and key context (its Y tables for chunks of 4096 .. 32768 positions), with and without AAD,
    2^k blocks exactly (as many chunk workgroups as CUs: the finisher comes late) and ragged."""
    import torch
    rnd = random.Random(77001)
    key, nonce = rnd.randbytes(16), rnd.randbytes(12)
    k = uaes.GcmKey(key)
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    try:
        for n, alen in (((24 << 20) + 5, 0), (32 << 20, 0), ((40 << 20) - 16, 7), (64 << 20, 0), ((64 << 20) + 16, 4096),
                        ((100 << 20) + 3, 0), (128 << 20, 0), ((128 << 20) - 4096, 33),
                        ((200 << 20) + 3, 0), (512 << 20, 0)):                  # hash only: 64 and 128 positions per thread
            src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
            a = torch.randint(0, 256, (alen,), dtype=torch.uint8, device="cuda:0") if alen else None
            ct = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
            uaes.gcm_encrypt_dev(key, nonce, a, src, n, ct)                   # striped one-pass kernel (> 16 MiB)
            torch.cuda.synchronize()
            head = 1 << 16                                                     # the first 64 KiB and the tag against the oracle's CTR
            aad = bytes(a.cpu().numpy()) if alen else b""
            assert bytes(ct[:head].cpu().numpy()) == orc.gcm_encrypt(key, nonce, aad, bytes(src[:head].cpu().numpy()))[:head]
            for dec in (lambda *x: uaes.gcm_decrypt_dev(key, nonce, *x), lambda *x: k.decrypt_dev(nonce, *x)):
                back = torch.full((n + 16,), 0xCC, dtype=torch.uint8, device="cuda:0")
                dec(a, ct, n, back, status)
                torch.cuda.synchronize()
                assert int(status.item()) == 0 and torch.equal(back[:n], src) and int((back[n:] != 0xCC).sum()) == 0, (n, alen)
                for pos in (0, n // 2 + 1, n - 1, n + 15):
                    bad = ct.clone()
                    bad[pos] ^= 0x20
                    back.fill_(0xCC)
                    dec(a, bad, n, back, status)
                    torch.cuda.synchronize()
                    assert int(status.item()) == 0x1A and int((back != 0xCC).sum()) == 0, (n, alen, pos)
            del src, ct, back, bad
    finally:
        k.close()


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_chunk_and_combine_kernels(orc, bits):
    """Texts between the single-workgroup kernel and the striped one-pass kernel (2047 GHASH blocks .. 16 MiB;
    32 MiB for a decrypt that authenticates first) run as k_gcm_chunks + k_gcm_combine: chunk counts 1 / 2 / 3 /
    255..257 / 511..513 / 1023..1025, front padding of every size, AAD that spans chunks, ragged texts; encrypt
    against the oracle (tag always, whole text for the small ones), both decrypt orders, forgeries, in place,
    key context."""
    import torch
    L = uaes.engine()
    rnd = random.Random(6100 + bits)
    key = rnd.randbytes(bits // 8)
    k = uaes.GcmKey(key)
    B = 16
    shapes = [(2046 * B, 0), (2047 * B, 0), (2047 * B + 1, 0), (4095 * B, 0), (4095 * B - 7, 32), (100, 4094 * B), (6000 * B + 3, 5000),
              (255 * 2048 * B - B, 0), (256 * 2048 * B - B, 0), (256 * 2048 * B, 0), (256 * 2048 * B + B, 13)]
    shapes += [(0, 2047 * B), (0, (3 << 20) + 5)]             # no text at all (GMAC): hash-only chunk workgroups, both directions
    if bits == 128:
        shapes += [(512 * 2048 * B - B, 0), (512 * 2048 * B, 16), (1024 * 2048 * B - B, 0), (1024 * 2048 * B, 0), (0, 40 << 20)]
    try:
        for n, alen in shapes:
            # hand-picked for the INSIDE of the chunk arrangement (chunk counts, front padding): the table must agree
            assert uaes.plan("gcm", n, alen)[0] in ("gcm.small", "gcm.chunks", "gcm.twophase"), (n, alen)
            nonce, aad = rnd.randbytes(12), rnd.randbytes(alen)
            pt = orc.splitmix(n % 977 + 3, (n + 7) // 8 * 8)[:n]
            want = orc.gcm_encrypt(key, nonce, aad, pt)
            src = torch.frombuffer(bytearray(pt + bytes(16)), dtype=torch.uint8).to("cuda:0")
            dst = torch.full((n + 48,), 0xA5, dtype=torch.uint8, device="cuda:0")
            a = torch.frombuffer(bytearray(aad + bytes(3)), dtype=torch.uint8).to("cuda:0")[:alen] if alen else None
            uaes.gcm_encrypt_dev(key, nonce, a, src, n, dst)
            torch.cuda.synchronize()
            got = bytes(dst[: n + 16].cpu().numpy())
            assert got[-16:] == want[-16:], (n, alen)
            assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest(), (n, alen)
            assert int((dst[n + 16:] != 0xA5).sum()) == 0
            two = torch.full((n + 32,), 0xA5, dtype=torch.uint8, device="cuda:0")
            k.encrypt_dev(nonce, a, src, n, two)
            torch.cuda.synchronize()
            assert torch.equal(two[: n + 16], dst[: n + 16]), (n, alen)
            status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
            for sw in (0, 1):                                            # tag first / one pass
                L.uaes_set_gcm_one_pass_decrypt(sw)
                back = torch.full((n + 16,), 0xCC, dtype=torch.uint8, device="cuda:0")
                uaes.gcm_decrypt_dev(key, nonce, a, dst, n, back, status)
                torch.cuda.synchronize()
                assert int(status.item()) == 0 and torch.equal(back[:n], src[:n]) and int((back[n:] != 0xCC).sum()) == 0, (n, alen, sw)
                bad = dst[: n + 16].clone()
                bad[rnd.randrange(n + 16)] ^= 4
                back.fill_(0xCC)
                k.decrypt_dev(nonce, a, bad, n, back, status)
                torch.cuda.synchronize()
                assert int(status.item()) == 0x1A, (n, alen, sw)
                assert int((back[:n] != (0 if sw else 0xCC)).sum()) == 0 and int((back[n:] != 0xCC).sum()) == 0, (n, alen, sw)
            L.uaes_set_gcm_one_pass_decrypt(0)
            work = dst[: n + 16].clone()                                  # in place
            uaes.gcm_decrypt_dev(key, nonce, a, work, n, work, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(work[:n], src[:n])
    finally:
        L.uaes_set_gcm_one_pass_decrypt(0)
        k.close()


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_one_pass_decrypt(orc, bits):
    """uaes_set_gcm_one_pass_decrypt(1): a long GCM decrypt into the caller's device buffer runs CTR and
    GHASH in one pass (k_gcm_fused<NR, true>: the lane hashes the ciphertext block it has just read, the
    tail is decrypted after the tag check).  Good tags: plaintext identical to the two-pass result, out of
    place and in place, AAD of every shape, ragged tails, key context too.  Bad tag / bad AAD / a flipped
    bit anywhere (head, striped region, tail): non-zero status and an all-zero output.  Host buffers take
    the one-pass kernel regardless of the switch and leave the caller's buffer untouched on 0x1A."""
    import torch
    L = uaes.engine()
    rnd = random.Random(4200 + bits)
    key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
    S16 = 16 << 19
    cases = [(S16 + 254 * 16, 0), (S16 + 254 * 16 + 16, 5), (S16 + 5 * 32768 + 4064 + 7, 13),
             (2 * S16 + 5000, 16), (3 * S16 - 1, 4096 + 7), (5 * S16 + 12345, 0)]
    k = uaes.GcmKey(key)
    try:
        assert L.uaes_set_gcm_one_pass_decrypt(1) == 0
        for i, (n, alen) in enumerate(cases):
            pt = orc.splitmix(2000 + i, (n + 7) // 8 * 8)[:n]
            aad = rnd.randbytes(alen)
            src = torch.frombuffer(bytearray(pt + bytes(16)), dtype=torch.uint8).to("cuda:0")
            ct = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
            a = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to("cuda:0") if alen else None
            uaes.gcm_encrypt_dev(key, nonce, a, src, n, ct)
            status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
            back = torch.full((n + 32,), 0xCC, dtype=torch.uint8, device="cuda:0")
            uaes.gcm_decrypt_dev(key, nonce, a, ct, n, back, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(back[:n], src[:n]), (n, alen)
            assert int((back[n:] != 0xCC).sum()) == 0                     # nothing written past the text
            # key context, in place (CT || tag becomes PT || tag)
            work = ct.clone()
            status.fill_(-1)
            k.decrypt_dev(nonce, a, work, n, work, status)
            torch.cuda.synchronize()
            assert int(status.item()) == 0 and torch.equal(work[:n], src[:n]) and torch.equal(work[n:], ct[n:])
            # forgeries: a bit of the head, of the striped region, of the tail, of the tag; a bit of the AAD
            spots = [3, 255 * 16 + 1, n // 2, n - 1, n + 5]
            for j, pos in enumerate(spots if i < 2 or bits == 128 else spots[2:4]):
                bad = ct.clone()
                bad[pos] ^= 0x40
                status.fill_(-1)
                back.fill_(0xCC)
                uaes.gcm_decrypt_dev(key, nonce, a, bad, n, back, status)
                torch.cuda.synchronize()
                assert int(status.item()) != 0 and int(back[:n].sum()) == 0, (n, alen, pos)
                assert int((back[n:] != 0xCC).sum()) == 0
            if alen:
                a2 = a.clone()
                a2[alen // 2] ^= 1
                status.fill_(-1)
                work = ct.clone()
                uaes.gcm_decrypt_dev(key, nonce, a2, work, n, work, status)        # in place: the ciphertext is gone
                torch.cuda.synchronize()
                assert int(status.item()) != 0 and int(work[:n].sum()) == 0 and torch.equal(work[n:], ct[n:])
        assert L.uaes_set_gcm_one_pass_decrypt(0) == 1
        # default switch: device buffers untouched on a forgery (two passes) ...
        n = S16 + 70000
        pt = orc.splitmix(77, n)
        full = uaes.AES_GCM_encrypt(key, nonce, b"hdr", pt)
        bad = bytearray(full)
        bad[n // 3] ^= 2
        d_bad = torch.frombuffer(bad, dtype=torch.uint8).to("cuda:0")
        d_hdr = torch.frombuffer(bytearray(b"hdr"), dtype=torch.uint8).to("cuda:0")
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        back = torch.full((n,), 0xCC, dtype=torch.uint8, device="cuda:0")
        uaes.gcm_decrypt_dev(key, nonce, d_hdr, d_bad, n, back, status)
        torch.cuda.synchronize()
        assert int(status.item()) != 0 and int((back != 0xCC).sum()) == 0
        # ... and host buffers (one pass in private staging) likewise, both switch settings
        for sw in (0, 1):
            L.uaes_set_gcm_one_pass_decrypt(sw)
            assert uaes.AES_GCM_decrypt(key, nonce, b"hdr", full) == (0, pt)
            rc, text = uaes.AES_GCM_decrypt(key, nonce, b"hdr", bytes(bad), prefill=0xCC)
            assert rc == 0x1A and text == b"\xcc" * n
            assert k.decrypt(nonce, b"hdr", full) == (0, pt)
            assert k.decrypt(nonce, b"hdr", bytes(bad), prefill=0xCC) == (0x1A, b"\xcc" * n)
    finally:
        L.uaes_set_gcm_one_pass_decrypt(0)
        k.close()


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_key_context_equals_the_one_shot_calls(orc, bits):
    """uaes_gcm_key_*: the key's tables are built once, every message then costs Enc(J0) + the data kernels.
    Messages of every path (last levels only, two-pass bulk with its size-dependent table, one-pass kernel),
    interleaved under ONE context, must equal the one-shot calls and the oracle; device-pointer variants too."""
    import torch
    rnd = random.Random(77 + bits)
    key = rnd.randbytes(bits // 8)
    k = uaes.GcmKey(key)
    for n in (0, 1, 16, 4096, 70001, 600 << 10, 33, (9 << 20) + 7, 255, 3 << 20, 100):
        nonce, aad = rnd.randbytes(12), rnd.randbytes(rnd.choice([0, 5, 16, 100]))
        data = orc.splitmix(n + 1, (n + 7) // 8 * 8)[:n]
        ct = k.encrypt(nonce, aad, data)
        assert ct == uaes.AES_GCM_encrypt(key, nonce, aad, data), n
        if n <= 70001:
            assert ct == orc.gcm_encrypt(key, nonce, aad, data), n
        assert k.decrypt(nonce, aad, ct) == (0, data), n
        bad = bytearray(ct)
        bad[rnd.randrange(len(bad))] ^= 0x10
        assert k.decrypt(nonce, aad, bytes(bad), prefill=0xCC) == (0x1A, b"\xcc" * n), n
    n = 1 << 20
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
    one, two = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0"), torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
    nonce = rnd.randbytes(12)
    k.encrypt_dev(nonce, None, src, n, one)
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, two)
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    k.decrypt_dev(nonce, None, one, n, back, status)
    torch.cuda.synchronize()
    assert torch.equal(one, two) and int(status.item()) == 0 and torch.equal(back, src)
    k.close()


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_record_calls_equal_a_call_per_record(orc, bits):
    """uaes_gcm_key_{en,de}crypt_records: many equally long messages in one launch, each with its own nonce (and
    its own or the shared AAD).  Every record must equal the oracle's AES_GCM_encrypt of that record alone; on
    decryption the records with a wrong tag keep what the output held (N7), the others come out, verdicts say which.
    The shapes walk the kernel's six arrangements: 16 / 4 / 1 records per workgroup turn, one or two GHASH positions
    per thread, and their boundaries (63/64, 127/128, 255/256, 511/512, 1023/1024 positions + the Enc(J0) slot)."""
    rnd = random.Random(4100 + bits)
    key = rnd.randbytes(bits // 8)
    k = uaes.GcmKey(key)
    assert k.record_max(0) == (2046 - 1) * 16 and k.record_max(33) == (2046 - 1 - 3) * 16
    try:
        shapes = [(1, 0, 0), (3, 1, 0), (5, 16, 13), (300, 1500, 13), (7, 4096, 0), (600, 100, 5), (2, 16368, 0),
                  (3, k.record_max(40), 40), (9, 16383, 16), (260, 33, 100), (4, 32720, 0), (5, 10000, 20), (11, 3000, 7),
                  (70, 1000, 0), (33, 2016, 0), (17, 8160, 0)]
        for nrec, rec_len, aad_len in shapes:
            nonces = [rnd.randbytes(12) for _ in range(nrec)]
            recs = [rnd.randbytes(rec_len) for _ in range(nrec)]
            for per_record_aad in (False, True):
                aads = [rnd.randbytes(aad_len) for _ in range(nrec)] if per_record_aad else rnd.randbytes(aad_len)
                aad_of = (lambda r: aads[r]) if per_record_aad else (lambda r: aads)
                want = [orc.gcm_encrypt(key, nonces[r], aad_of(r), recs[r]) for r in range(nrec)]
                got = k.encrypt_records(nonces, aads, recs)
                assert got == want, (nrec, rec_len, aad_len, per_record_aad)
                rc, ver, texts = k.decrypt_records(nonces, aads, got, prefill=0xCC)
                assert rc == 0 and ver == [0] * nrec and texts == recs
                # spoil some records: in the text, in the tag, (with AAD) by handing over another record's nonce
                bad = sorted(rnd.sample(range(nrec), min(nrec, 1 + nrec // 7)))
                spoiled = list(got)
                for r in bad:
                    b = bytearray(spoiled[r])
                    b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
                    spoiled[r] = bytes(b)
                rc, ver, texts = k.decrypt_records(nonces, aads, spoiled, prefill=0xCC)
                assert rc == 0x1A
                assert ver == [0x1A if r in bad else 0 for r in range(nrec)]
                for r in range(nrec):
                    assert texts[r] == (b"\xcc" * rec_len if r in bad else recs[r]), (nrec, rec_len, r)
                if aad_len == 0:
                    break
        # a strided layout: the gaps between the records are not written
        nrec, rec_len, stride = 37, 1000, 2048
        nonces = [rnd.randbytes(12) for _ in range(nrec)]
        recs = [rnd.randbytes(rec_len) for _ in range(nrec)]
        got = k.encrypt_records(nonces, b"hdr", recs, stride=stride)
        assert got == [orc.gcm_encrypt(key, nonces[r], b"hdr", recs[r]) for r in range(nrec)]
        L = uaes.engine()
        src = bytearray(stride * nrec)
        for r in range(nrec):
            src[r * stride: r * stride + rec_len + 16] = got[r]
        dst = (C.c_uint8 * (stride * nrec))()
        C.memset(dst, 0xEE, stride * nrec)
        ver = (C.c_uint8 * nrec)()
        rc = L.uaes_gcm_key_decrypt_records(k._h, nrec, (C.c_uint8 * (12 * nrec)).from_buffer_copy(b"".join(nonces)),
                                            (C.c_uint8 * 3).from_buffer_copy(b"hdr"), 3, 0,
                                            (C.c_uint8 * len(src)).from_buffer_copy(bytes(src)), rec_len, stride, dst, stride, ver)
        assert rc == 0
        b = bytes(dst)
        for r in range(nrec):
            assert b[r * stride: r * stride + rec_len] == recs[r]
            assert b[r * stride + rec_len: (r + 1) * stride] == b"\xee" * (stride - rec_len)
        # argument checks
        assert L.uaes_gcm_key_encrypt_records(k._h, 1, dst, None, 0, 0, dst, k.record_max(0) + 1, 65536, dst, 65536) == -2
        assert L.uaes_gcm_key_encrypt_records(k._h, 2, dst, None, 0, 0, dst, 100, 120, dst, 128) == -2     # stride % 16
        assert L.uaes_gcm_key_encrypt_records(k._h, 2, dst, None, 0, 0, dst, 100, 112, dst, 112) == -2     # no room for the tag
        assert L.uaes_gcm_key_encrypt_records(k._h, 0, None, None, 0, 0, None, 100, 112, None, 128) == 0
    finally:
        k.close()


@pytest.mark.parametrize("bits", [128, 256])
def test_gcm_record_calls_with_records_of_different_lengths(orc, bits):
    """uaes_gcm_key_{en,de}crypt_records_v: packet buffers -- equal slots, every record its own length (zero included),
    the launch arranged for the longest.  Every record against the oracle's call of that record alone, in each of the
    kernel's arrangements; forged records are reported one by one and keep the output's prefill."""
    import torch
    rnd = random.Random(4300 + bits)
    key = rnd.randbytes(bits // 8)
    k = uaes.GcmKey(key)
    try:
        for nrec, max_len, aad_len in [(1, 0, 0), (40, 60, 13), (100, 1000, 0), (37, 1500, 5), (19, 4096, 13), (11, 9000, 0),
                                       (5, 20000, 16), (260, 300, 20), (3, k.record_max(0), 0)]:
            lens = [rnd.choice([0, 1, max_len, max_len // 2, rnd.randrange(0, max_len + 1)]) for _ in range(nrec)]
            lens[rnd.randrange(nrec)] = max_len
            nonces = [rnd.randbytes(12) for _ in range(nrec)]
            recs = [rnd.randbytes(n) for n in lens]
            aad = rnd.randbytes(aad_len)
            want = [orc.gcm_encrypt(key, nonces[r], aad, recs[r]) for r in range(nrec)]
            got = k.encrypt_records_v(nonces, aad, recs, max_len=max_len)
            assert got == want, (nrec, max_len, aad_len)
            rc, ver, texts = k.decrypt_records_v(nonces, aad, got, prefill=0xCC, max_len=max_len)
            assert rc == 0 and ver == [0] * nrec and texts == recs
            bad = sorted(rnd.sample(range(nrec), min(nrec, 1 + nrec // 9)))
            spoiled = list(got)
            for r in bad:
                b = bytearray(spoiled[r])
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
                spoiled[r] = bytes(b)
            rc, ver, texts = k.decrypt_records_v(nonces, aad, spoiled, prefill=0xCC, max_len=max_len)
            assert rc == 0x1A and ver == [0x1A if r in bad else 0 for r in range(nrec)]
            for r in range(nrec):
                assert texts[r] == (b"\xcc" * lens[r] if r in bad else recs[r]), (nrec, max_len, r)
        # device flavour: lengths on the device, one of them longer than max_len (clamped, not overrun)
        nrec, max_len, stride = 500, 1440, 1472
        lens = [rnd.randrange(0, max_len + 1) for _ in range(nrec)]
        lens[7] = max_len + 999
        d_lens = torch.tensor(lens, dtype=torch.int32, device="cuda:0")
        nonces = torch.randint(0, 256, (nrec * 12,), dtype=torch.uint8, device="cuda:0")
        src = torch.randint(0, 256, (nrec * stride,), dtype=torch.uint8, device="cuda:0")
        dst = torch.full((nrec * stride,), 0xA5, dtype=torch.uint8, device="cuda:0")
        L = uaes.engine()
        st = torch.cuda.current_stream().cuda_stream
        assert L.uaes_gcm_key_encrypt_records_v_dev(k._h, nrec, nonces.data_ptr(), None, 0, 0, src.data_ptr(), d_lens.data_ptr(),
                                                    max_len, stride, dst.data_ptr(), stride, st) == 0
        torch.cuda.synchronize()
        nb, sb, ob = (bytes(t.cpu().numpy()) for t in (nonces, src, dst))
        for r in range(nrec):
            n = min(lens[r], max_len)
            assert ob[r * stride: r * stride + n + 16] == orc.gcm_encrypt(key, nb[12 * r: 12 * r + 12], b"", sb[r * stride: r * stride + n]), r
    finally:
        k.close()


def test_gcm_record_calls_on_device_buffers(orc):
    """the *_dev flavour: device pointers for nonces, AAD, records; many more records than workgroups; in place;
    two record calls sharing one key context on two streams"""
    import torch
    rnd = random.Random(4200)
    key = rnd.randbytes(16)
    k = uaes.GcmKey(key)
    try:
        nrec, rec_len, stride, aad_len = 3000, 1440, 1472, 13
        nonces = torch.randint(0, 256, (nrec * 12 + 1,), dtype=torch.uint8, device="cuda:0")[1:]     # not word aligned
        aad = torch.randint(0, 256, (nrec * 16,), dtype=torch.uint8, device="cuda:0")
        buf = torch.randint(0, 256, (nrec * stride,), dtype=torch.uint8, device="cuda:0")
        plain = buf.clone()
        out2 = torch.zeros_like(buf)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        aligned = nonces.clone()
        k.encrypt_records_dev(nrec, aligned, aad, aad_len, 16, plain, rec_len, stride, out2, stride, stream=s2)
        k.encrypt_records_dev(nrec, nonces, aad, aad_len, 16, buf, rec_len, stride, buf, stride, stream=s1)   # in place
        torch.cuda.synchronize()
        nb, ab, ob, pb = (bytes(t.cpu().numpy()) for t in (nonces, aad, buf, plain))
        o2 = bytes(out2.cpu().numpy())
        for r in list(range(0, nrec, 97)) + [nrec - 1]:
            want = orc.gcm_encrypt(key, nb[12 * r: 12 * r + 12], ab[16 * r: 16 * r + aad_len], pb[r * stride: r * stride + rec_len])
            assert ob[r * stride: r * stride + rec_len + 16] == want, r
            assert o2[r * stride: r * stride + rec_len + 16] == want, r
            assert ob[r * stride + rec_len + 16: (r + 1) * stride] == pb[r * stride + rec_len + 16: (r + 1) * stride]
        back = torch.full_like(buf, 0xCC)
        ver = torch.full((nrec,), 0x55, dtype=torch.uint8, device="cuda:0")
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        buf[5 * stride + 7] ^= 1
        buf[2999 * stride + rec_len + 15] ^= 0x80
        k.decrypt_records_dev(nrec, nonces, aad, aad_len, 16, buf, rec_len, stride, back, stride, ver, status)
        torch.cuda.synchronize()
        assert int(status.item()) == 0x1A
        v = ver.cpu().numpy()
        assert set(np.nonzero(v)[0].tolist()) == {5, 2999} and int(v[5]) == 0x1A
        bb = bytes(back.cpu().numpy())
        for r in range(nrec):
            want = b"\xcc" * rec_len if r in (5, 2999) else pb[r * stride: r * stride + rec_len]
            assert bb[r * stride: r * stride + rec_len] == want, r
    finally:
        k.close()


def test_C4_gcm128_1GiB_device_resident(orc, golden_dir):
    """BASELINE configs[3]: tag and digest of CT||tag for the 1 GiB message."""
    import torch
    d = load(golden_dir, "digests.json")["C4_gcm128_1GiB_seed4"]
    key, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC))
    n = 1 << 30
    src = _device_stream(orc, 4, n)
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
    torch.cuda.synchronize()
    assert bytes(dst[n:].cpu().numpy()).hex() == d["tag"]
    assert _sha_of(dst) == d["sha256_ct_tag"]
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    back = torch.full((n,), 0xCC, dtype=torch.uint8, device="cuda:0")
    uaes.gcm_decrypt_dev(key, nonce, None, dst, n, back, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(back, src)
    dst[12345] ^= 1
    back.fill_(0xCC)
    uaes.gcm_decrypt_dev(key, nonce, None, dst, n, back, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0x1A and int(back.min().item()) == 0xCC     # untouched (N7)


def test_large_single_calls_against_the_oracle(orc):
    """the largest texts the oracle still finishes in seconds: one XTS data unit of 192 MiB (+ a ragged
    end: 12288 groups of 64 chunk tweaks from the parallel expansion, then ciphertext stealing) and one
    OCB text of 128 MiB (offsets up to L_23, every run length)"""
    import hashlib
    import torch

    def sha(b):
        return hashlib.sha256(b).hexdigest()

    n = (192 << 20) + 16 + 5
    keys, tweak = bytes(range(64)), bytes(range(100, 116))
    src = _device_stream(orc, 31, (n + 7) // 8 * 8)[:n]
    dst = torch.empty_like(src)
    L = uaes.engine()
    assert L.uaes_xts_encrypt(256, keys, tweak, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())) == 0
    rc, want = orc.xts(keys, tweak, bytes(src.cpu().numpy()), True)
    assert rc == 0 and sha(bytes(dst.cpu().numpy())) == sha(want)
    back = torch.empty_like(src)
    assert L.uaes_xts_decrypt(256, keys, tweak, C.c_void_p(dst.data_ptr()), n, C.c_void_p(back.data_ptr())) == 0
    assert torch.equal(back, src)
    del dst, back

    n = (128 << 20) + 16 * 3 + 7
    key, nonce, aad = bytes(range(16)), bytes(range(50, 62)), b"associated data"
    src = src[:n]
    out = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    assert L.uaes_ocb_encrypt(128, key, nonce, aad, len(aad), C.c_void_p(src.data_ptr()), n, C.c_void_p(out.data_ptr())) == 0
    assert sha(bytes(out.cpu().numpy())) == sha(orc.ocb_encrypt(key, nonce, aad, bytes(src.cpu().numpy())))
    pt = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    assert L.uaes_ocb_decrypt(128, key, nonce, aad, len(aad), C.c_void_p(out.data_ptr()), n, C.c_void_p(pt.data_ptr())) == 0
    assert torch.equal(pt, src)


def test_gcm_beyond_4GiB_equals_its_shards():
    """64-bit indexing in CTR and GHASH: one 5 GiB + 40 B GCM message against five shards of it
    (counter offsets above 2^28 blocks, weighted partial tags XORed) -- the oracle is too slow here,
    the two code paths check each other; the head is pinned by the other tests"""
    import torch
    import micro_aes_amd.sharding as sh
    total, world = (5 << 30) + 40, 5
    key, nonce = bytes(range(16)), bytes(range(40, 52))
    aad = torch.arange(23, dtype=torch.uint8, device="cuda:0")
    src = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda:0")
    one = torch.empty(total + 16, dtype=torch.uint8, device="cuda:0")
    uaes.gcm_encrypt_dev(key, nonce, aad, src, total, one)
    torch.cuda.synchronize()
    parts = torch.empty(total + 16, dtype=torch.uint8, device="cuda:0")
    shares = []
    for rank in range(world):
        start, n, _ = sh.gcm_shard_roles(total, rank, world)
        tag = sh.gcm_encrypt_sharded(key, nonce, aad, 23, total, src[start:start + n], parts[start:start + n], rank, world,
                                     gather=lambda share: shares.append(share) or list(shares))
    torch.cuda.synchronize()
    assert torch.equal(one[:total], parts[:total])
    assert tag == bytes(one[total:].cpu().numpy())
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    uaes.gcm_decrypt_dev(key, nonce, aad, one, total, parts, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and torch.equal(parts[:total], src)


def test_gmac_of_a_bulk_text(orc):
    """authentication only (empty plaintext, the text as AAD): aligned device AAD takes whole-block
    loads, a misaligned one the byte path; both against the oracle, sizes around the level plans"""
    import torch
    L = uaes.engine()
    key, nonce = bytes(range(16, 32)), bytes(range(12))
    for n in (1, 16, 4097, (1 << 20) + 5, (6 << 20) + 16):
        data = orc.splitmix(n + 77, n)
        want = orc.gcm_encrypt(key, nonce, data, b"")
        t = torch.zeros(n + 32, dtype=torch.uint8, device="cuda:0")
        for off in (0, 16, 3):
            t[off:off + n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
            tag = (C.c_uint8 * 16)()
            assert L.uaes_gcm_encrypt(128, key, nonce, C.c_void_p(t.data_ptr() + off), n, None, 0, tag) == 0
            assert bytes(tag) == want, (n, off)
            assert L.uaes_gcm_decrypt(128, key, nonce, C.c_void_p(t.data_ptr() + off), n, tag, 0, None) == 0


def test_gcm_records_v_slot_tails_hold_no_earlier_message(orc):
    """the synchronous variable-length record calls copy whole slots back from a per-thread staging buffer
    that is reused across calls.  The bytes of a slot behind lens[r] (+ tag) must not be an EARLIER call's text: after a
    call with long, recognisable records, a call with short ones must return slots whose tails are zeros, in both
    directions (decrypt: only the text is defined, the rest of the slot's max_len bytes)."""
    import ctypes as C
    L = uaes.engine()
    key = bytes(range(16))
    k = uaes.GcmKey(key)
    try:
        nrec, max_len, stride = 64, 1024, 1024 + 16
        marker = b"\xA5SECRET\x5A" * 128
        long_recs = [marker[:max_len] for _ in range(nrec)]
        nonces = [bytes([r]) * 12 for r in range(nrec)]
        ct_long = k.encrypt_records_v(nonces, b"", long_recs, max_len=max_len, stride=stride)
        rc, ver, texts = k.decrypt_records_v(nonces, b"", ct_long, max_len=max_len, stride=stride)      # plaintext now sits in the staging
        assert rc == 0 and texts == long_recs
        short = [bytes([r]) * (r % 7) for r in range(nrec)]
        lens = (C.c_uint32 * nrec)(*[len(s) for s in short])
        src = bytearray(stride * nrec)
        for r, s in enumerate(short):
            src[r * stride: r * stride + len(s)] = s
        dst = (C.c_ubyte * (stride * nrec))(*([0xEE] * (stride * nrec)))
        nb = b"".join(nonces)
        assert L.uaes_gcm_key_encrypt_records_v(k._h, nrec, nb, None, 0, 0, bytes(src), lens, max_len, stride, dst, stride) == 0
        out = bytes(dst)
        for r, s in enumerate(short):
            slot = out[r * stride: (r + 1) * stride]
            assert slot[: len(s) + 16] == orc.gcm_encrypt(key, nonces[r], b"", s)
            assert slot[len(s) + 16: max_len + 16] == bytes(max_len - len(s)), "slot %d leaks staging bytes" % r
        # decrypt direction: the ciphertexts just made, text slots of max_len bytes
        cts = bytearray(stride * nrec)
        for r, s in enumerate(short):
            cts[r * stride: r * stride + len(s) + 16] = out[r * stride: r * stride + len(s) + 16]
        dst2 = (C.c_ubyte * (stride * nrec))(*([0xEE] * (stride * nrec)))
        ver = (C.c_ubyte * nrec)()
        assert L.uaes_gcm_key_decrypt_records_v(k._h, nrec, nb, None, 0, 0, bytes(cts), lens, max_len, stride, dst2, stride, ver) == 0
        out2 = bytes(dst2)
        for r, s in enumerate(short):
            assert out2[r * stride: r * stride + len(s)] == s
            assert marker[:8] not in out2[r * stride: r * stride + max_len], "slot %d leaks an earlier plaintext" % r
            assert out2[r * stride + len(s): r * stride + max_len] == bytes(max_len - len(s))
    finally:
        k.close()


@pytest.mark.parametrize("bits", [128, 256])
def test_ctr_partial_last_round_of_stripes(orc, bits):
    """k_ctr_shared2 deals 32 KiB stripes round-robin over the workgroups; a last round that covers only part of the grid is
    handed to the kernel's edge path (one block per thread, plain rounds) when it is below CTR_TAIL_PCT (80 %) of the grid.
    Sizes on both sides of that rule -- 2.5 rounds, 2 rounds + one stripe, 3 rounds - one stripe, an exact multiple -- with
    ragged ends and a counter that carries, against the oracle."""
    import torch
    key = bytes(range(bits // 8))
    ctr0 = bytes(range(12)) + b"\xff\xff\xff\x80"
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    stripe = 32 << 10
    for n in (cus * stripe * 5 // 2 + 5, 2 * cus * stripe + stripe + 16 * 3, 3 * cus * stripe - stripe - 7, 2 * cus * stripe,
              cus * stripe * 3 // 2 + 4096):
        data = orc.splitmix(n & 0xffff, n)
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
        dst = torch.full((n + 64,), 0xEE, dtype=torch.uint8, device="cuda:0")
        uaes.ctr_xcrypt_dev(key, ctr0, 77, src, dst, nbytes=n)
        torch.cuda.synchronize()
        got = bytes(dst.cpu().numpy())
        assert hashlib.sha256(got[:n]).digest() == hashlib.sha256(orc.ctr_xcrypt_at(key, ctr0, 77, data)).digest(), (bits, n)
        assert got[n:] == b"\xee" * 64


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_xts_units_shorter_than_a_chunk(orc, bits):
    """Data units of 64 B .. 4080 B (a multiple of 64 bytes) beyond the one-launch limit go through the PACKED
    arrangement of k_xts: the text as one flat run of 256-block chunks, a lane on four consecutive blocks of one unit,
    its unit's tweak loaded per lane and shifted to the block's position.  Power-of-two and other unit sizes (the
    division by the unit's block count is a multiplication), a partial last chunk, unit numbers that carry into the
    high word, in place, both directions, against the oracle.  512-byte sectors: 140 -> ~900 GiB/s."""
    import torch
    keys = bytes(range(7, 7 + bits // 4))
    for sector_bytes, nsectors, first in [(512, 8192 + 77, (1 << 32) - 100), (64, 70001, 3), (128, 40000, 0), (192, 30001, 1 << 50),
                                          (1024, 5000, 9), (2048, 4097, (1 << 64) - 3000), (4032, 4100, 77), (576, 9000, 5)]:
        n = sector_bytes * nsectors
        data = orc.splitmix(sector_bytes + nsectors, n)
        rc, want = orc.xts_sectors(keys, first, sector_bytes, data, True)
        assert rc == 0
        buf = torch.full((n + 64,), 0xEE, dtype=torch.uint8, device="cuda:0")
        buf[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
        uaes.xts_sectors_dev(keys, first, sector_bytes, nsectors, buf, buf)               # in place
        torch.cuda.synchronize()
        got = bytes(buf.cpu().numpy())
        assert hashlib.sha256(got[:n]).digest() == hashlib.sha256(want).digest(), (sector_bytes, nsectors)
        assert got[n:] == b"\xee" * 64
        back = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        uaes.xts_sectors_dev(keys, first, sector_bytes, nsectors, buf, back, encrypt=False)
        torch.cuda.synchronize()
        assert bytes(back.cpu().numpy()) == data, (sector_bytes, nsectors)


@pytest.mark.parametrize("sector_bytes,nsectors", [(4096, 4096 + 512), (4096, 2 * 4096 + 1), (512, 8 * 4096 * 2 + 4099), (4096 + 16, 4300),
                                                   (528, 70001)])
def test_xts_partial_last_round_of_chunks(orc, sector_bytes, nsectors):
    """k_xts gives every wave 256-block chunks round-robin; a last round that covers only part of the grid's waves goes by
    QUARTER chunks (64 blocks, one per lane, tweak * alpha^(64 k)) over four times as many waves.  Volumes just past one
    and two rounds (a round = 4096 waves x 4 KiB), short chunks at the end of every unit (4112-byte units: 257 blocks = a
    full chunk + a one-block chunk), both directions, against the oracle."""
    import torch
    keys = bytes(range(64))
    n = sector_bytes * nsectors
    data = orc.splitmix(nsectors & 0xffff, n)
    rc, want = orc.xts_sectors(keys, 1 << 33, sector_bytes, data, True)
    assert rc == 0
    src = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    dst = torch.full((n + 64,), 0xEE, dtype=torch.uint8, device="cuda:0")
    uaes.xts_sectors_dev(keys, 1 << 33, sector_bytes, nsectors, src, dst)
    torch.cuda.synchronize()
    got = bytes(dst.cpu().numpy())
    assert hashlib.sha256(got[:n]).digest() == hashlib.sha256(want).digest()
    assert got[n:] == b"\xee" * 64
    back = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    uaes.xts_sectors_dev(keys, 1 << 33, sector_bytes, nsectors, dst, back, encrypt=False)
    torch.cuda.synchronize()
    assert torch.equal(back, src)


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ecb_partial_last_round_of_tiles(orc, bits):
    """k_ecb's remainder path.  One round = 256 workgroups x 64 KiB tiles = 16 MiB; a last round that
    covers less than 80 % of the grid is handed over as single blocks over all workgroups ([tail_from, nfull)), a fuller
    one runs as whole tiles through the buffer-resource loads with idle workgroups.  Sizes on both sides of that switch,
    each with a trailing partial block and every padding mode, both directions, against the oracle (SHA-256)."""
    import torch
    L = uaes.engine()
    tile, rnd_bytes = 64 << 10, 16 << 20
    key = bytes(range(7, 7 + bits // 8))
    dev = torch.device("cuda", 0)
    for n, padding in (((5 * rnd_bytes) // 2 + 5, 0),                  # 2.5 rounds: the remainder as single blocks
                       (2 * rnd_bytes + tile + 16 * 3 + 11, 1),        # 2 rounds + 1 tile (+ 3 blocks + 11 bytes), PKCS#7
                       (3 * rnd_bytes - tile + 7, 2),                  # 3 rounds - 1 tile: whole tiles, ISO 7816-4
                       (2 * rnd_bytes + 13 * tile, 0),                 # no ragged tail, 80 % boundary side A
                       (2 * rnd_bytes + 204 * tile + 16, 1)):          # just under 80 % of a round: single blocks
        pt = orc.splitmix(n % 251 + bits, n + 8)[:n]
        want = orc.ecb_encrypt(key, pt, padding)
        src = torch.frombuffer(bytearray(pt + bytes(32)), dtype=torch.uint8).to(dev)
        dst = torch.zeros(len(want) + 32, dtype=torch.uint8, device=dev)
        assert L.uaes_ecb_encrypt_padded(bits, key, padding, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())) == 0
        got = dst.cpu().numpy().tobytes()
        assert hashlib.sha256(got[: len(want)]).digest() == hashlib.sha256(want).digest(), (bits, n, padding)
        assert got[len(want):] == bytes(32)                                         # nothing beyond the last block
        back = torch.zeros(len(want), dtype=torch.uint8, device=dev)
        assert L.uaes_ecb_decrypt(bits, key, C.c_void_p(dst.data_ptr()), len(want), C.c_void_p(back.data_ptr())) == 0
        assert back[:n].cpu().numpy().tobytes() == pt, (bits, n, padding)


@pytest.mark.parametrize("bits", [128, 256])
def test_ctr_counter_bits_40_47_move_inside_the_striped_region(orc, bits):
    """The striped CTR kernel makes its lane constants once per launch: they depend on counter bits 40..47, and the
    launcher cuts a text at the (one in 2^40 blocks) place where those move (ctr_stripes_cross_a, launch_ctr_shared).
    Counters whose low 40 bits run out 3 / 40 / 70 MiB into a 96 MiB text -- in the head, in the middle of the stripes
    and near their end -- and the same with the 2^56 wrap, against the oracle (incBlock's carry, micro_aes.c:421-427)."""
    import torch
    key = bytes(range(1, 1 + bits // 8))
    n = (96 << 20) + 16 * 5 + 3
    data = orc.splitmix(4040, n + 5)[:n]
    src = torch.frombuffer(bytearray(data + bytes(16)), dtype=torch.uint8).to("cuda:0")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    for top, into_mib in ((0x12, 3), (0x12, 40), (0xfe, 70), (0xffff, 40), (0xffff, 95)):
        blocks_left = (into_mib << 20) // 16 + 7                      # the carry happens this many blocks into the text
        low40 = (1 << 40) - blocks_left
        v = ((top << 40) | low40) & ((1 << 56) - 1)
        ctr0 = bytes(range(0xA0, 0xA9)) + v.to_bytes(7, "big")
        for off in (0, 11):
            dst.zero_()
            uaes.ctr_xcrypt_dev(key, ctr0, off, src, dst, nbytes=n)
            torch.cuda.synchronize()
            want = orc.ctr_xcrypt_at(key, ctr0, off, data)
            assert hashlib.sha256(dst[:n].cpu().numpy().tobytes()).digest() == hashlib.sha256(want).digest(), (bits, top, into_mib, off)
