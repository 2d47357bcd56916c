"""Drop-in proof on the GPU: the reference's OWN callers -- its testvectors/
harness (aes_testvectors.c) and its main.c -- compiled unchanged against
include/micro_aes.h and linked to libmicro_aes_hip_<bits>.so (recipe:
oracle/Makefile, target `dropin`; binaries travel in oracle/_ref/), must report
the same verdicts and case counts as when they are built on micro_aes.c."""
import os
import re
import subprocess

import pytest

from tests.refbuilt import REF_DIR, makefile_targets, missing, need

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
EXPECT = {128: {"CMAC": 96, "GCM": 375, "CCM": 10, "OCB": 16, "GCM-SIV": 102, "XTS": 800}, 192: {"CMAC": 144, "GCM": 375, "CCM": 10},
          256: {"CMAC": 96, "GCM": 375, "CCM": 10, "XTS": 600}}


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_reference_harness_runs_on_the_hip_library(bits, tmp_path, golden_dir):
    exe = need("harness_hip_%d" % bits)
    for f in os.listdir(golden_dir):
        if f.endswith((".rsp", ".tv")):
            os.symlink(os.path.join(golden_dir, f), tmp_path / f)
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict((m, int(n)) for m, n in re.findall(
        r"Verifying vectors: AES%d-([\w-]+)\s+Nmber of tests:\s*(\d+), All Passed!" % bits, r.stdout))
    assert got == EXPECT[bits], r.stdout


@pytest.mark.parametrize("exe,bits,want", [
    ("harness_hip_128_gcmiv1", 128, {"CMAC": 96, "GCM": 375, "CCM": 10, "OCB": 16, "GCM-SIV": 102, "XTS": 800}),
    ("harness_hip_256_gcmiv128", 256, {"CMAC": 96, "GCM": 375, "CCM": 10, "XTS": 600})])
def test_reference_harness_with_other_gcm_nonce_lengths(exe, bits, want, tmp_path, golden_dir):
    """the unchanged harness built with -DGCM_NONCE_LEN=1 / 128 picks the [IVlen = 8] / [IVlen = 1024]
    sections of the NIST GCM files (aes_testvectors_GCM.h:86) and must pass all 375 on the HIP library"""
    path = need(exe)
    for f in os.listdir(golden_dir):
        if f.endswith((".rsp", ".tv")):
            os.symlink(os.path.join(golden_dir, f), tmp_path / f)
    r = subprocess.run([path], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict((m, int(n)) for m, n in re.findall(
        r"Verifying vectors: AES%d-([\w-]+)\s+Nmber of tests:\s*(\d+), All Passed!" % bits, r.stdout))
    assert got == want, r.stdout


@pytest.mark.parametrize("exe,bits,want", [
    ("harness_hip_128_lens1", 128, {"CMAC": 96, "GCM": 375, "CCM": 10, "OCB": 7, "GCM-SIV": 102, "XTS": 800}),
    ("harness_hip_256_lens2", 256, {"CMAC": 96, "GCM": 375, "CCM": 10, "XTS": 600})])
def test_reference_harness_with_other_length_constants(exe, bits, want, tmp_path, golden_dir):
    """the unchanged harness built with -DCCM_NONCE_LEN=13 -DGCM_TAG_LEN=12 -DOCB_TAG_LEN=12 (AES-128) and
    -DCCM_NONCE_LEN=7 -DGCM_TAG_LEN=4 (AES-256) against include/micro_aes.h: it then runs the [Nlen = 13] / [Nlen = 7]
    sections of the VNT files, compares 12 / 4 bytes of every GCM tag and takes the seven 12-byte-tag OCB stanzas (one of them an expected failure)
    (aes_testvectors_CCM.h:84, _GCM.h:24,86, _OCB.h:90) -- all on the HIP library"""
    path = need(exe)
    for f in os.listdir(golden_dir):
        if f.endswith((".rsp", ".tv")):
            os.symlink(os.path.join(golden_dir, f), tmp_path / f)
    r = subprocess.run([path], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = dict((m, int(n)) for m, n in re.findall(
        r"Verifying vectors: AES%d-([\w-]+)\s+Nmber of tests:\s*(\d+), All Passed!" % bits, r.stdout))
    assert got == want, r.stdout


@pytest.mark.parametrize("bits", [128, 192, 256])
def test_reference_main_c_runs_on_the_hip_library(bits):
    exe = need("main_hip_%d" % bits)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    passed = re.findall(r"AES-%d (\w+) \w+: PASSED!" % bits, r.stdout)
    want = {128: ["ECB", "ECB", "CBC", "CBC", "CFB", "CFB", "OFB", "OFB", "CTR", "CTR", "XTS", "XTS", "plaintext",
                  "GCM", "GCM", "CCM", "CCM", "OCB", "OCB", "GCMSIV", "GCMSIV",
                  "OCB", "OCB", "GCMSIV", "GCMSIV", "GCMSIV", "GCMSIV"],
            192: [], 256: ["XTS", "XTS", "GCM", "GCM"]}[bits]
    assert passed == want, r.stdout


def test_reference_main_c_aes192_pkcs7_check_runs_on_the_hip_library():
    """main.c's only AES-192 check on the hot path (main.c:139) is compiled when AES_PADDING is 1:
    the same main.c built with -DAES_PADDING=1 against include/micro_aes.h, which then binds
    AES_ECB_encrypt to the PKCS#7 entry point of libmicro_aes_hip_192.so"""
    exe = need("main_hip_192_pkcs7")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    assert re.findall(r"AES-192 (\w+) \w+: PASSED!", r.stdout) == ["ECB", "ECB"], r.stdout


def test_reference_main_c_preset_counter_runs_on_the_hip_library():
    """main.c built with -DPRESET_COUNTER=1 (micro_aes.h:100) passes its 16-byte iVec to AES_CTR_* as the
    whole counter block and checks its own known answer for that case (main.c:45-47); include/micro_aes.h
    then binds AES_CTR_* to the *_preset entry points (uaes_ctr_xcrypt_at with block offset 0)"""
    exe = need("main_hip_128_presetctr")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    assert re.findall(r"AES-128 (CTR) \w+: PASSED!", r.stdout) == ["CTR", "CTR"], r.stdout


def test_reference_main_c_without_cts_runs_on_the_hip_library():
    """main.c built with -DCTS=0 (micro_aes.h:56) checks CBC against its zero-padded known answer (main.c:36-40,
    :149-150: the whole padded ciphertext, and decryption of it); include/micro_aes.h then binds AES_CBC_* to the
    *_nocts entry points (padded last chunk, block-parallel whole-block decryption)"""
    exe = need("main_hip_128_nocts")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    assert re.findall(r"AES-128 (CBC) \w+: PASSED!", r.stdout) == ["CBC", "CBC"], r.stdout


def test_reference_built_evidence_travelled_to_this_box():
    """on the GPU box every binary the `ref` and `dropin` targets of oracle/Makefile name must be
    under oracle/_ref/ (it is git-ignored and travels only as an untracked directory), and bench.py's CPU baseline must
    really be the compiled reference (`kind: reference`), not the restatement it would quietly fall back to."""
    import bench
    gone = [n for n in makefile_targets() if not os.path.exists(os.path.join(REF_DIR, n))]
    if gone:
        missing("oracle/_ref/{%s}" % ",".join(gone))
    cb = bench.cpu_baseline("ctr", sample=4 << 20, all_cores=False)
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["value"] > 0, cb
