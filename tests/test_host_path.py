"""The engine's own HOST data path (micro-aes_amd/csrc/uaes_host.c) parity-tested like a kernel -- on the CPU, no GPU
needed: the very test bodies of tests/test_gpu_parity.py (the reference-held NIST / ACVP / OpenSSL vectors, the outputs
of the compiled reference in tests/golden/*.json, the seeded comparisons with the oracle) are run again with the host
path FORCED through the same C ABI and the same compat libraries, plus the reference's own main.c and test-vector
harness linked to the drop-in libraries on this GPU-less box.

And the other half of the contract: the host path is OPT-IN.  With the default policy (0, 0, 0) nothing ever runs on the
host -- a box without a GPU gets UAES_E_HIP from every call -- and tests/conftest.py checks before every `-m gpu` test
that the policy is still the default, so no GPU parity test can pass on this code.

uaes_host.c shares nothing with oracle/ (test_product_does_not_touch_the_oracle covers the whole product tree)."""
import ctypes as C
import os
import random
import re
import subprocess

import pytest

import micro_aes_amd as uaes
from tests import test_gpu_parity as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture()
def host_forced():
    prev = uaes.host_policy(1 << 62, 1, 1)
    yield
    uaes.host_policy(*prev)


def test_host_path_is_off_by_default_and_a_gpu_less_box_fails_loudly():
    import torch
    assert uaes.host_policy() == (0, 0, 0)
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    with pytest.raises(uaes.EngineError, match="no usable HIP device"):
        uaes.AES_CTR_encrypt(bytes(16), bytes(12), b"x" * 32)
    with pytest.raises(uaes.EngineError, match="no usable HIP device"):
        uaes.AES_CMAC(bytes(16), b"x" * 32)
    # max_bytes alone does not make a long call a host call
    prev = uaes.host_policy(64, 0, 0)
    try:
        assert uaes.AES_CTR_encrypt(bytes(16), bytes(12), b"x" * 64) == uaes.AES_CTR_encrypt(bytes(16), bytes(12), b"x" * 64)
        with pytest.raises(uaes.EngineError, match="no usable HIP device"):
            uaes.AES_CTR_encrypt(bytes(16), bytes(12), b"x" * 65)
        with pytest.raises(uaes.EngineError, match="no usable HIP device"):
            uaes.AES_CMAC(bytes(16), b"x" * 65)                  # a chain, but chains are not switched on
        uaes.host_policy(64, 1, 0)
        assert len(uaes.AES_CMAC(bytes(16), b"x" * 100000)) == 16
        with pytest.raises(uaes.EngineError, match="no usable HIP device"):
            uaes.AES_CBC_decrypt(bytes(16), bytes(16), b"x" * 100000)     # the parallel direction is not a chain
    finally:
        uaes.host_policy(*prev)
    assert uaes.host_policy() == (0, 0, 0)


def test_host_policy_from_the_environment():
    code = ("import micro_aes_amd as u; print(u.host_policy()); "
            "print(u.AES_CTR_encrypt(bytes(16), bytes(12), bytes(40)).hex())")
    env = dict(os.environ, UAES_HOST_MAX="4096", UAES_HOST_CHAINS="1", UAES_HOST_FALLBACK="1", PYTHONPATH=ROOT)
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split()
    assert r.stdout.startswith("(4096, 1, 1)")
    from oracle.pyoracle import Oracle
    assert lines[-1] == Oracle().ctr_encrypt(bytes(16), bytes(12), bytes(40)).hex()
    env = {k: v for k, v in os.environ.items() if not k.startswith("UAES_HOST")}
    env.update(UAES_HOST_POLICY="recommended", PYTHONPATH=ROOT)
    r = subprocess.run(["python", "-c", "import micro_aes_amd as u; print(u.host_policy())"], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=300)
    assert r.stdout.startswith("(4096, 1, 1)"), r.stdout + r.stderr
    r = subprocess.run(["python", "-c", "import micro_aes_amd as u; print(u.host_policy())"], capture_output=True, text=True,
                       env=dict(env, UAES_HOST_MAX="0", UAES_HOST_FALLBACK="0"), cwd=ROOT, timeout=300)
    assert r.stdout.startswith("(0, 1, 0)"), r.stdout + r.stderr


# ---- the reference-held vector files and the compiled reference's outputs, through the compat ABI, host path forced ----
@pytest.mark.parametrize("bits", [128, 192, 256])
def test_gcm_rsp(host_forced, bits):
    P.test_gcm_rsp_through_compat_api(bits)


@pytest.mark.parametrize("bits", [128, 192, 256])
@pytest.mark.parametrize("iv_bytes", [1, 128])
def test_gcm_rsp_other_nonce_lengths(host_forced, bits, iv_bytes):
    P.test_gcm_rsp_other_nonce_lengths_through_compat_api(bits, iv_bytes)


@pytest.mark.parametrize("bits,count", [(128, 800), (256, 600)])
def test_xts_rsp(host_forced, bits, count):
    P.test_xts_rsp_through_compat_api(bits, count)


@pytest.mark.parametrize("bits,count", [(128, 96), (192, 144), (256, 96)])
def test_cmac_rsp(host_forced, bits, count):
    P.test_cmac_rsp_through_compat_api(bits, count)


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_ccm_rsp(host_forced, bits):
    P.test_ccm_rsp_through_compat_api(bits)
    P.test_ccm_rsp_every_nonce_length_through_compat_api(bits)


def test_ocb_and_gcmsiv_vector_files(host_forced):
    P.test_ocb_openssl_vectors()
    P.test_ocb_vectors_with_other_lengths_through_compat_api()
    P.test_gcmsiv_acvp_vectors()


def test_outputs_of_the_compiled_reference(host_forced, orc, golden_dir):
    P.test_main_c_kats(golden_dir)
    P.test_reference_generated_vectors(orc, golden_dir)
    P.test_length_constant_golden_vectors(golden_dir)
    P.test_ecb_padding_modes(orc, golden_dir)
    P.test_build_variant_vectors_cbc_without_cts_and_other_ctr_constants(orc, golden_dir)
    P.test_baseline_small_digests(orc, golden_dir)


# ---- against the oracle on seeded inputs (the same bodies the kernels are held to) ----
@pytest.mark.parametrize("bits", [128, 192, 256])
def test_modes_vs_oracle(host_forced, orc, bits):
    P.test_ecb_ctr_vs_oracle(orc, bits)
    P.test_xts_vs_oracle(orc, bits)
    P.test_gcm_vs_oracle(orc, bits)
    P.test_cmac_ccm_vs_oracle(orc, bits)
    P.test_feedback_modes_vs_oracle(orc, bits)
    P.test_gcmsiv_vs_oracle(orc, bits)
    P.test_ocb_vs_oracle(orc, bits)


@pytest.mark.parametrize("bits", [128, 256])
def test_every_length_constant_vs_oracle(host_forced, orc, bits):
    P.test_nonce_and_tag_lengths_vs_oracle(orc, bits)
    P.test_xts_long_data_units(orc, bits)


def test_gcm_nonce_lengths(host_forced, orc):
    P.test_gcm_nonce_lengths_vs_oracle(orc)


def test_in_place_and_every_small_length(host_forced, orc):
    """in == out (the reference memcpy()s in -> out and works in place, micro_aes.h:520-526) and every length 0..80"""
    L = uaes.engine()
    rnd = random.Random(77)
    for bits in (128, 192, 256):
        key, keys, iv, nonce = rnd.randbytes(bits // 8), rnd.randbytes(bits // 4), rnd.randbytes(16), rnd.randbytes(12)
        for n in range(0, 81):
            pt = rnd.randbytes(n)
            aad = rnd.randbytes(n % 19)
            buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
            assert L.uaes_ctr_xcrypt(bits, key, nonce, buf, n, buf) == 0
            assert bytes(buf)[:n] == orc.ctr_encrypt(key, nonce, pt)
            buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
            assert L.uaes_gcm_encrypt(bits, key, nonce, aad, len(aad), buf, n, buf) == 0
            want = orc.gcm_encrypt(key, nonce, aad, pt)
            assert bytes(buf)[: n + 16] == want
            assert L.uaes_gcm_decrypt(bits, key, nonce, aad, len(aad), buf, n, buf) == 0 and bytes(buf)[:n] == pt
            buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
            assert L.uaes_ocb_encrypt(bits, key, nonce, aad, len(aad), buf, n, buf) == 0
            assert bytes(buf)[: n + 16] == orc.ocb_encrypt(key, nonce, aad, pt)
            assert L.uaes_ocb_decrypt(bits, key, nonce, aad, len(aad), buf, n, buf) == 0 and bytes(buf)[:n] == pt
            if n >= 16:
                for fn, ref in ((L.uaes_xts_encrypt, orc.xts(keys, iv, pt, True)), ):
                    buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
                    assert fn(bits, keys, iv, buf, n, buf) == 0
                    assert bytes(buf)[:n] == (ref[1] if isinstance(ref, tuple) else ref)
                    assert L.uaes_xts_decrypt(bits, keys, iv, buf, n, buf) == 0 and bytes(buf)[:n] == pt
                buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
                assert L.uaes_cbc_encrypt(bits, key, iv, buf, n, buf) == 0
                assert bytes(buf)[:n] == orc.cbc(key, iv, pt, True)[1]
                assert L.uaes_cbc_decrypt(bits, key, iv, buf, n, buf) == 0 and bytes(buf)[:n] == pt
            buf = (C.c_uint8 * (n + 32)).from_buffer_copy(pt + bytes(32))
            assert L.uaes_cfb_encrypt(bits, key, iv, buf, n, buf) == 0
            assert bytes(buf)[:n] == orc.cfb(key, iv, pt, True)
            assert L.uaes_cfb_decrypt(bits, key, iv, buf, n, buf) == 0 and bytes(buf)[:n] == pt


def test_forgeries_on_the_host_path(host_forced, orc):
    """N7 for GCM (nothing written); CCM / GCM-SIV / OCB hand back the text like the reference's default build, or zeros
    under uaes_set_wipe_on_auth_failure"""
    key, nonce, aad, pt = bytes(range(16)), bytes(range(12)), b"hdr", bytes(range(200))
    ct = bytearray(uaes.AES_GCM_encrypt(key, nonce, aad, pt))
    ct[7] ^= 1
    assert uaes.AES_GCM_decrypt(key, nonce, aad, bytes(ct), prefill=0x5A) == (0x1A, b"\x5a" * 200)
    L = uaes.engine()
    for enc, dec, n in ((uaes.AES_CCM_encrypt, uaes.AES_CCM_decrypt, nonce[:11]), (uaes.GCM_SIV_encrypt, uaes.GCM_SIV_decrypt, nonce),
                        (uaes.AES_OCB_encrypt, uaes.AES_OCB_decrypt, nonce)):
        good = enc(key, n, aad, pt)
        bad = good[:-1] + bytes([good[-1] ^ 0x80])
        rc, out = dec(key, n, aad, bad, prefill=0x5A)
        assert rc == 0x1A and out == pt                          # the reference's default: SABOTAGE is a no-op
        prev = L.uaes_set_wipe_on_auth_failure(1)
        try:
            assert dec(key, n, aad, bad, prefill=0x5A) == (0x1A, bytes(200))
        finally:
            L.uaes_set_wipe_on_auth_failure(prev)


# ---- the reference's own callers, linked to the drop-in libraries, on a box WITHOUT a GPU ----
def _gpu_less():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.parametrize("exe,want_passed", [("main_hip_128", 27), ("main_hip_256", 4), ("main_hip_192_pkcs7", 2),
                                             ("main_hip_128_presetctr", 2), ("main_hip_128_nocts", 2)])
def test_reference_main_c_on_the_drop_in_library_without_a_gpu(exe, want_passed):
    """`main_hip_128` on a GPU-less box.  Default policy: the void functions reach the failure handler
    (abort); with UAES_HOST_FALLBACK=1 every known-answer test of the reference's main.c passes."""
    from tests.refbuilt import need
    path = need(exe)
    if not _gpu_less():
        pytest.skip("this box has a GPU")
    r = subprocess.run([path], capture_output=True, text=True, timeout=300, env=dict(os.environ, UAES_HOST_FALLBACK="1"))
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("PASSED") >= want_passed, r.stdout
    env = {k: v for k, v in os.environ.items() if not k.startswith("UAES_HOST")}
    d = subprocess.run([path], capture_output=True, text=True, timeout=300, env=env)
    assert d.returncode != 0 and "no usable HIP device" in d.stderr, (d.returncode, d.stderr[-500:])     # loud, not wrong


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_reference_testvector_harness_on_the_drop_in_library_without_a_gpu(bits, tmp_path, golden_dir):
    """the reference's testvectors/ harness, unchanged, on libmicro_aes_hip_<bits>.so with the host fallback: the same
    case counts as on the GPU (tests/test_gpu_dropin.py)"""
    from tests.test_gpu_dropin import EXPECT
    from tests.refbuilt import need
    exe = need("harness_hip_%d" % bits)
    if not _gpu_less():
        pytest.skip("this box has a GPU")
    for f in os.listdir(golden_dir):
        if f.endswith((".rsp", ".tv")):
            os.symlink(os.path.join(golden_dir, f), tmp_path / f)
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(os.environ, UAES_HOST_FALLBACK="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = dict((m, int(n)) for m, n in re.findall(
        r"Verifying vectors: AES%d-([\w-]+)\s+Nmber of tests:\s*(\d+), All Passed!" % bits, r.stdout))
    assert got == EXPECT[bits], r.stdout
