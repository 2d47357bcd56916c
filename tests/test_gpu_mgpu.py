"""uaes_mgpu_gcm_encrypt / _decrypt and uaes_mgpu_ecb_*: the multi-GPU split performed by the C host (one process,
N devices; SURVEY.md 8e, VERDICT r04 row e'').  Reference semantics: AES_GCM_encrypt / _decrypt micro_aes.c:1164-1212
with gHash :1127-1137 (N4, N6, N7), AES_ECB_encrypt / _decrypt :636-680 (N1).

A device list may name a device more than once, so {0}, {0,0,0} and {0} x 8 exercise the slicing, the per-slice
counter offsets, the share exchange and the two-phase decryption on the one GPU of the test box; the variants on
distinct devices skip below two GPUs.  Everything is compared with the oracle, with the single-call result and --
at BASELINE configs[3]'s full size -- with the compiled reference's tag and digest (tests/golden/digests.json)."""
import ctypes as C
import hashlib
import json
import os
import random

import pytest

import micro_aes_amd as uaes

pytestmark = pytest.mark.gpu

GIB = 1 << 30
DEVLISTS = [[0], [0, 0, 0], [0] * 8]


def _distinct_lists():
    import torch
    n = torch.cuda.device_count()
    return [list(range(n)), list(range(n - 1, -1, -1)) + [0]] if n >= 2 else []


# lengths around every boundary of the slicing: empty, shorter than the device list, ragged tails, one block per
# device, a text whose slices take the striped one-pass kernel (>= 256 CUs x 2048 blocks each), ...
LENS = [0, 1, 15, 16, 17, 47, 48, 127, 128, 129, 1000, 4096 + 5, 100003, (1 << 20) + 16 * 3 + 7]


@pytest.mark.parametrize("devlist", DEVLISTS, ids=lambda d: "x".join(map(str, d)))
def test_mgpu_gcm_matches_oracle_and_single_call(orc, devlist):
    rnd = random.Random(500 + len(devlist))
    for bits in (128, 192, 256):
        key, nonce = rnd.randbytes(bits // 8), rnd.randbytes(12)
        for n in LENS:
            for aad in (b"", rnd.randbytes(1 + n % 41)):
                pt = orc.splitmix(n + 7, n)
                want = orc.gcm_encrypt(key, nonce, aad, pt)
                got = uaes.mgpu_gcm_encrypt(devlist, key, nonce, aad, pt)
                assert got == want, (bits, n, len(aad))
                assert got == uaes.AES_GCM_encrypt(key, nonce, aad, pt)
                rc, back = uaes.mgpu_gcm_decrypt(devlist, key, nonce, aad, got, prefill=0xAB)
                assert rc == 0 and back == pt, (bits, n, len(aad))
                # N7 across devices: a forged tag, a forged byte in any slice, forged AAD -> 0x1A, nothing written
                bad = bytearray(got)
                bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
                rc, back = uaes.mgpu_gcm_decrypt(devlist, key, nonce, aad, bytes(bad), prefill=0xAB)
                assert rc == uaes.M_AUTHENTICATION_ERROR and back == b"\xab" * n, (bits, n)
                if aad:
                    rc, back = uaes.mgpu_gcm_decrypt(devlist, key, nonce, aad[:-1] + bytes([aad[-1] ^ 0x80]), got, prefill=0xAB)
                    assert rc == uaes.M_AUTHENTICATION_ERROR and back == b"\xab" * n


def test_mgpu_gcm_nist_vectors_through_three_slices():
    """the reference-held GcmEncryptExtIV*.rsp (the 375 cases per key size its own harness runs: 96-bit IV, 128-bit
    tag, aes_testvectors_GCM.h:86) through uaes_mgpu_gcm_* on {0,0,0}"""
    from tests.rsp import gcm_cases
    for bits in (128, 192, 256):
        cases = gcm_cases(bits)
        assert len(cases) == 375
        for v in cases:
            pt, aad = v.get("PT", b""), v.get("AAD", b"")
            got = uaes.mgpu_gcm_encrypt([0, 0, 0], v["Key"], v["IV"], aad, pt)
            assert got == v.get("CT", b"") + v["Tag"][:16], (bits, v["Count"])
            rc, back = uaes.mgpu_gcm_decrypt([0, 0, 0], v["Key"], v["IV"], aad, got)
            assert rc == 0 and back == pt


def test_mgpu_gcm_in_place_and_device_pointers(orc):
    import torch
    key, nonce, aad = bytes(range(32)), bytes(range(12)), b"associated"
    n = (48 << 20) + 16 * 9 + 3      # {0,0,0}: 16 MiB slices, the striped one-pass kernel; {0} x 8: 6 MiB, the generic path
    pt = orc.splitmix(77, n)
    want = orc.gcm_encrypt(key, nonce, aad, pt)
    # host, in place (the reference memcpy()s in -> out and works there, micro_aes.h:520-526)
    L = uaes.engine()
    buf = (C.c_uint8 * (n + 16)).from_buffer_copy(pt + bytes(16))
    devs = (C.c_int * 3)(0, 0, 0)
    assert L.uaes_mgpu_gcm_encrypt(3, devs, 256, key, nonce, aad, len(aad), buf, n, buf) == 0
    assert bytes(buf) == want
    assert L.uaes_mgpu_gcm_decrypt(3, devs, 256, key, nonce, aad, len(aad), buf, n, buf) == 0
    assert bytes(buf)[:n] == pt
    # device memory, out of place and in place; a forgery leaves the device buffer untouched too
    src = torch.frombuffer(bytearray(pt), dtype=torch.uint8).to("cuda:0")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0")
    for devlist in ([0], [0, 0, 0], [0] * 8):
        dst.zero_()
        uaes.mgpu_gcm_dev(devlist, key, nonce, aad, src.data_ptr(), n, dst.data_ptr())
        assert bytes(dst.cpu().numpy()) == want, devlist
        back = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda:0")
        assert uaes.mgpu_gcm_dev(devlist, key, nonce, aad, dst.data_ptr(), n, back.data_ptr(), decrypt=True) == 0
        assert torch.equal(back, src)
        dst[n // 2] ^= 4
        back.fill_(0xAB)
        assert uaes.mgpu_gcm_dev(devlist, key, nonce, aad, dst.data_ptr(), n, back.data_ptr(), decrypt=True) == uaes.M_AUTHENTICATION_ERROR
        assert int((back != 0xAB).sum()) == 0
    # one-pass decryption of device buffers: same plaintext; a forgery hands back zeros, never text
    prev = L.uaes_set_gcm_one_pass_decrypt(1)
    try:
        dst.zero_()
        uaes.mgpu_gcm_dev([0, 0, 0], key, nonce, aad, src.data_ptr(), n, dst.data_ptr())
        back = torch.full((n,), 0xAB, dtype=torch.uint8, device="cuda:0")
        assert uaes.mgpu_gcm_dev([0, 0, 0], key, nonce, aad, dst.data_ptr(), n, back.data_ptr(), decrypt=True) == 0
        assert torch.equal(back, src)
        dst[5] ^= 1
        assert uaes.mgpu_gcm_dev([0, 0, 0], key, nonce, aad, dst.data_ptr(), n, back.data_ptr(), decrypt=True) == uaes.M_AUTHENTICATION_ERROR
        assert int((back != 0).sum()) == 0
    finally:
        L.uaes_set_gcm_one_pass_decrypt(prev)


def test_mgpu_gcm_C4_1GiB_in_eight_shards_matches_the_reference(golden_dir):
    """BASELINE configs[3] (AES-128-GCM, 1 GiB, seed 4) cut into 8 slices by the C host: the reference's tag and
    SHA-256(CT || tag); then N7 at full size -- a forged byte in the LAST slice leaves all eight untouched"""
    import torch
    import bench
    with open(os.path.join(golden_dir, "digests.json")) as f:
        want = json.load(f)["C4_gcm128_1GiB_seed4"]
    dev = torch.device("cuda", 0)
    key, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC))
    src = bench.splitmix_device(torch, 4, GIB, 0, dev)
    dst = torch.empty(GIB + 16, dtype=torch.uint8, device=dev)
    uaes.mgpu_gcm_dev([0] * 8, key, nonce, b"", src.data_ptr(), GIB, dst.data_ptr())
    assert bytes(dst[GIB:].cpu().numpy()).hex() == want["tag"]
    h = hashlib.sha256()
    for o in range(0, GIB + 16, 1 << 28):
        h.update(dst[o:o + (1 << 28)].cpu().numpy().tobytes())
    assert h.hexdigest() == want["sha256_ct_tag"]
    back = torch.full((GIB,), 0x5A, dtype=torch.uint8, device=dev)
    assert uaes.mgpu_gcm_dev([0] * 8, key, nonce, b"", dst.data_ptr(), GIB, back.data_ptr(), decrypt=True) == 0
    assert torch.equal(back, src)
    back.fill_(0x5A)
    dst[GIB - 3] ^= 0x10
    assert uaes.mgpu_gcm_dev([0] * 8, key, nonce, b"", dst.data_ptr(), GIB, back.data_ptr(), decrypt=True) == uaes.M_AUTHENTICATION_ERROR
    assert int((back != 0x5A).sum()) == 0


@pytest.mark.parametrize("devlist", DEVLISTS, ids=lambda d: "x".join(map(str, d)))
def test_mgpu_ecb_matches_oracle(orc, devlist):
    rnd = random.Random(900 + len(devlist))
    for bits in (128, 192, 256):
        key = rnd.randbytes(bits // 8)
        for n in (0, 1, 16, 20, 47, 48, 16 * 8, 16 * 8 + 5, 4096, 100003, (2 << 20) + 16 * 5 + 9):
            pt = orc.splitmix(n + 3, n)
            for padding in (0, 1, 2):
                want = orc.ecb_encrypt(key, pt, padding)
                got = uaes.mgpu_ecb_encrypt(devlist, key, pt, padding)
                assert got == want, (bits, n, padding)
            ct = orc.ecb_encrypt(key, pt)
            rc, back = uaes.mgpu_ecb_decrypt(devlist, key, ct)
            assert rc == 0 and back[:n] == pt
            if n % 16:                                # N1: floor(len/16) blocks, 0x1D, the ragged tail passed through
                rc, back = uaes.mgpu_ecb_decrypt(devlist, key, ct[:n])
                rc1, back1 = uaes.AES_ECB_decrypt(key, ct[:n])
                assert rc == uaes.M_DECRYPTION_ERROR == rc1 and back == back1


def test_mgpu_argument_checks():
    L = uaes.engine()
    out = (C.c_uint8 * 64)()
    bad = (C.c_int * 1)(99)
    assert L.uaes_mgpu_gcm_encrypt(1, bad, 128, bytes(16), bytes(12), None, 0, b"x" * 16, 16, out) == -2
    assert L.uaes_mgpu_gcm_encrypt(0, None, 128, bytes(16), bytes(12), None, 0, b"x" * 16, 16, out) == -2
    assert L.uaes_mgpu_gcm_encrypt(1, None, 100, bytes(16), bytes(12), None, 0, b"x" * 16, 16, out) == -2
    assert L.uaes_mgpu_gcm_decrypt(1, None, 128, bytes(16), None, None, 0, b"x" * 32, 16, out) == -2
    assert L.uaes_mgpu_ecb_encrypt(1, None, 128, bytes(16), 3, b"x" * 16, 16, out) == -2
    assert L.uaes_mgpu_ecb_decrypt(17, None, 128, bytes(16), b"x" * 16, 16, out) == -2


def test_mgpu_gcm_and_ecb_on_distinct_devices(orc):
    lists = _distinct_lists()
    if not lists:
        pytest.skip("needs 2 GPUs")
    rnd = random.Random(4242)
    key, nonce, aad = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(33)
    for devlist in lists:
        for n in (0, 5, 100003, (40 << 20) + 11):
            pt = orc.splitmix(n + 1, n)
            want = orc.gcm_encrypt(key, nonce, aad, pt)
            assert uaes.mgpu_gcm_encrypt(devlist, key, nonce, aad, pt) == want
            rc, back = uaes.mgpu_gcm_decrypt(devlist, key, nonce, aad, want, prefill=0xAB)
            assert rc == 0 and back == pt
            if n:
                bad = bytearray(want)
                bad[n - 1] ^= 1
                rc, back = uaes.mgpu_gcm_decrypt(devlist, key, nonce, aad, bytes(bad), prefill=0xAB)
                assert rc == uaes.M_AUTHENTICATION_ERROR and back == b"\xab" * n
            assert uaes.mgpu_ecb_encrypt(devlist, key, pt, 1) == orc.ecb_encrypt(key, pt, 1)


# ---- the same split without a change in the caller: UAES_DEVICES / uaes_set_devices ----
def _compat(bits):
    L = C.CDLL(uaes.lib_path("libmicro_aes_hip_%d.so" % bits))
    for name in ("AES_ECB_decrypt", "AES_GCM_decrypt", "AES_XTS_encrypt"):
        getattr(L, name).restype = C.c_char
    return L


def test_drop_in_symbols_spread_long_host_buffers_over_the_configured_devices(orc):
    """uaes_set_devices(3, {0,0,0}): AES_ECB_* / AES_CTR_* / AES_GCM_* (micro_aes.h:173-181, :256-266, :294-308) on a
    256 MiB HOST text go through uaes_mgpu_* -- bit-identical to the default one-device call -- while device pointers
    and texts below the threshold stay on the one-device path"""
    import numpy as np
    L, K = uaes.engine(), _compat(128)
    n = (256 << 20) + 16 * 3 + 5
    key, iv = bytes(range(16)), bytes(range(0xF0, 0xFC))
    src = np.empty(n, dtype=np.uint8)
    orc.splitmix_into(2, src[: n // 8 * 8])
    src[n // 8 * 8:] = 7
    sp = C.c_void_p(src.ctypes.data)

    def run_all():
        res = {}
        out = np.zeros(n + 32, dtype=np.uint8)
        op = C.c_void_p(out.ctypes.data)
        K.AES_CTR_encrypt(key, iv, sp, C.c_size_t(n), op)
        res["ctr"] = hashlib.sha256(out[:n].tobytes()).hexdigest()
        K.AES_ECB_encrypt(key, sp, C.c_size_t(n), op)
        res["ecb"] = hashlib.sha256(out[: (n + 15) // 16 * 16].tobytes()).hexdigest()
        ecb = out[: n // 16 * 16].copy()
        back = np.zeros(n, dtype=np.uint8)
        rc = K.AES_ECB_decrypt(key, C.c_void_p(ecb.ctypes.data), C.c_size_t(ecb.size), C.c_void_p(back.ctypes.data))
        assert ord(rc) == 0 and np.array_equal(back[: ecb.size], src[: ecb.size])
        K.AES_GCM_encrypt(key, iv, b"hdr", C.c_size_t(3), sp, C.c_size_t(n), op)
        res["gcm"] = hashlib.sha256(out[: n + 16].tobytes()).hexdigest()
        back[:] = 0xAB
        rc = K.AES_GCM_decrypt(key, iv, b"hdr", C.c_size_t(3), op, C.c_size_t(n), C.c_void_p(back.ctypes.data))
        assert ord(rc) == 0 and np.array_equal(back, src)
        out[n // 3] ^= 1                                         # N7 through the split: untouched
        back[:] = 0xAB
        rc = K.AES_GCM_decrypt(key, iv, b"hdr", C.c_size_t(3), op, C.c_size_t(n), C.c_void_p(back.ctypes.data))
        assert ord(rc) == uaes.M_AUTHENTICATION_ERROR and int((back != 0xAB).sum()) == 0
        ns = n // 4096
        assert L.uaes_xts_sectors(256, bytes(range(64)), 5, 4096, ns, sp, op, 1) == 0
        res["xts"] = hashlib.sha256(out[: ns * 4096].tobytes()).hexdigest()
        return res

    base = run_all()
    assert base["ctr"] == hashlib.sha256(orc.ctr_encrypt(key, iv, src.tobytes())).hexdigest()
    devs = (C.c_int * 3)(0, 0, 0)
    assert L.uaes_set_devices(3, devs, 1 << 20) == 0
    try:
        assert run_all() == base
        # short texts and device pointers are not split (and still right)
        assert uaes.AES_CTR_encrypt(key, iv, b"x" * 100) == orc.ctr_encrypt(key, iv, b"x" * 100)
    finally:
        assert L.uaes_set_devices(0, None, 64 << 20) == 0
    assert L.uaes_set_devices(2, (C.c_int * 2)(0, 99), 0) == -2


def test_reference_main_c_with_UAES_DEVICES_in_the_environment():
    """the reference's own main.c on the HIP library with UAES_DEVICES=0,0,0 and the threshold at one byte: every host
    call of its known-answer tests -- 57-byte texts -- is cut into three slices, and every verdict stays PASSED"""
    import re
    import subprocess
    from tests.refbuilt import need
    exe = need("main_hip_128")
    base = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    env = dict(os.environ, UAES_DEVICES="0,0,0", UAES_DEVICES_MIN_MIB="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    assert re.findall(r"AES-128 (\w+) \w+: PASSED!", r.stdout) == re.findall(r"AES-128 (\w+) \w+: PASSED!", base.stdout)
    assert r.stdout.count("PASSED") == base.stdout.count("PASSED") >= 27
    bad = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(env, UAES_DEVICES="0,x"))
    assert bad.returncode == 0 and "UAES_DEVICES ignored" in bad.stderr


def test_compat_calls_wait_for_a_named_producer_stream(orc):
    """VERDICT r04 #8c: device-pointer callers whose data is produced on a hipStreamNonBlocking stream (every torch side
    stream is one) name it with uaes_compat_set_producer_stream; the synchronous call then waits for it instead of for
    the default stream.  ~50 ms of work is queued in front of the write the cipher must see."""
    import torch
    K = _compat(128)
    K.uaes_compat_set_producer_stream.argtypes = [C.c_void_p]
    K.uaes_compat_set_producer_stream.restype = None
    key, iv = bytes(range(16)), bytes(range(0xF0, 0xFC))
    n = 64 << 20
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    src = torch.zeros(n, dtype=torch.uint8, device=dev)
    dst = torch.zeros(n, dtype=torch.uint8, device=dev)
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    final = torch.frombuffer(bytearray(orc.splitmix(9, n)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    want = hashlib.sha256(orc.ctr_encrypt(key, iv, bytes(final.cpu().numpy()))).hexdigest()
    K.uaes_compat_set_producer_stream(C.c_void_p(s.cuda_stream))
    try:
        for trial in range(3):
            src.zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                for _ in range(40):
                    big.add_(1)                                 # ~1 ms each: the producer is still busy when the call starts
                src.copy_(final)
            K.AES_CTR_encrypt(key, iv, C.c_void_p(src.data_ptr()), C.c_size_t(n), C.c_void_p(dst.data_ptr()))
            assert hashlib.sha256(dst.cpu().numpy().tobytes()).hexdigest() == want, trial
    finally:
        K.uaes_compat_set_producer_stream(None)


def test_gcm_shard_dev_one_pass_per_rank(orc):
    """uaes_gcm_shard_dev -- what a rank of a one-process-per-GPU job runs on its slice: CTR and the weighted share in
    ONE pass (mode 0), the hash-only share (mode 1) and the one-pass decrypt (mode 2), for 1..5 shards of texts on both
    sides of the striped kernel's minimum; the XOR of the shares is the reference's tag; sharding.gcm_encrypt_sharded /
    gcm_decrypt_sharded drive it with a stand-in exchange"""
    import torch
    from micro_aes_amd import sharding as sh
    dev = torch.device("cuda", 0)
    key, nonce, aad = bytes(range(5, 21)), bytes(range(12)), b"header bytes"
    d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to(dev)
    for total, world in ((0, 2), (100, 3), ((40 << 20) + 37, 2), ((40 << 20) + 37, 5)):
        pt = orc.splitmix(total % 97, total + 8)[:total]
        want = orc.gcm_encrypt(key, nonce, aad, pt)
        src = torch.frombuffer(bytearray(pt + bytes(16)), dtype=torch.uint8).to(dev)
        ct = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        shares = []
        for rank in range(world):
            start, n, takes = sh.gcm_shard_roles(total, rank, world)
            if not takes:
                continue
            p = torch.zeros(16, dtype=torch.uint8, device=dev)
            uaes.gcm_shard_dev(key, nonce, 0, d_aad if rank == 0 else None, len(aad), src[start:start + n], n, start, total,
                               ct[start:start + n], p)
            shares.append(p)
        torch.cuda.synchronize()
        tag = bytes(16)
        for p in shares:
            tag = bytes(a ^ b for a, b in zip(tag, bytes(p.cpu().numpy())))
        assert bytes(ct[:total].cpu().numpy()) == want[:-16] and tag == want[-16:], (total, world)
        # modes 1 and 2 over the ciphertext: the same shares; mode 2 also gives the plaintext back (in place)
        back = ct.clone()
        for mode in (1, 2):
            tag2 = bytes(16)
            for rank in range(world):
                start, n, takes = sh.gcm_shard_roles(total, rank, world)
                if not takes:
                    continue
                p = torch.zeros(16, dtype=torch.uint8, device=dev)
                uaes.gcm_shard_dev(key, nonce, mode, d_aad if rank == 0 else None, len(aad), back[start:start + n], n, start,
                                   total, back[start:start + n] if mode == 2 else None, p)
                tag2 = bytes(a ^ b for a, b in zip(tag2, bytes(p.cpu().numpy())))
            assert tag2 == want[-16:], (total, world, mode)
        assert bytes(back[:total].cpu().numpy()) == pt
        # the sharding module on top of it, rank by rank with a stand-in exchange
        all_shares = [bytes(p.cpu().numpy()) for p in shares]
        for rank in range(world):
            start, n, _ = sh.gcm_shard_roles(total, rank, world)
            dst = torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
            t = sh.gcm_encrypt_sharded(key, nonce, d_aad, len(aad), total, src[start:start + n], dst[:n], rank, world,
                                       gather=lambda share, r=rank: [s if i != r else share for i, s in enumerate(
                                           all_shares + [bytes(16)] * (world - len(all_shares)))])
            assert t == want[-16:] and bytes(dst[:n].cpu().numpy()) == want[start:start + n]


def test_mgpu_gcm_calls_from_several_host_threads(orc):
    """the two phases of uaes_mgpu_gcm_decrypt are two rounds of jobs on the per-device workers, and calls from several
    host threads interleave there: every call's state (shares, the private device copies of host slices) must be its own"""
    import threading
    key, aad = bytes(range(16)), b"thread"
    errors = []

    def work(t):
        try:
            rnd = random.Random(t)
            for i in range(6):
                n = rnd.choice([0, 33, 5000, (1 << 20) + 7, (9 << 20) + 16 * t + i])
                nonce = rnd.randbytes(12)
                pt = orc.splitmix(100 * t + i, n)
                devs = [0] * rnd.choice([1, 2, 3, 5])
                ct = uaes.mgpu_gcm_encrypt(devs, key, nonce, aad, pt)
                assert ct == orc.gcm_encrypt(key, nonce, aad, pt), (t, i, n)
                assert uaes.mgpu_gcm_decrypt(devs, key, nonce, aad, ct, prefill=0xAB) == (0, pt), (t, i, n)
                if n:
                    bad = bytearray(ct)
                    bad[rnd.randrange(n)] ^= 2
                    assert uaes.mgpu_gcm_decrypt(devs, key, nonce, aad, bytes(bad), prefill=0xAB) == (0x1A, b"\xab" * n), (t, i, n)
        except Exception as e:                                   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_long_host_gcm_texts_go_through_the_slice_pipeline(orc):
    """AES_GCM_encrypt on a long HOST text: the slices of the host pipeline are shards of the message (fused CTR +
    weighted share per slice, XORed on the host).  Same bytes and tag as the oracle / the device-resident call, with
    AAD, ragged length, in place; one worker (UAES_PIPE_WORKERS=1 in a child) takes the plain path and agrees"""
    import subprocess
    import sys
    import numpy as np
    L = uaes.engine()
    key, nonce, aad = bytes(range(32)), bytes(range(3, 15)), bytes(range(41))
    for n in ((96 << 20) + 16 * 7 + 9, (32 << 20), (33 << 20) + 1):
        src = np.empty(n + 16, dtype=np.uint8)
        orc.splitmix_into(n % 1000, src[: (n + 15) // 8 * 8] if (n + 15) // 8 * 8 <= n + 16 else src[: n // 8 * 8])
        pt = src[:n].tobytes()
        want = hashlib.sha256(orc.gcm_encrypt(key, nonce, aad, pt)).hexdigest()
        out = np.zeros(n + 16, dtype=np.uint8)
        assert L.uaes_gcm_encrypt(256, key, nonce, aad, len(aad), C.c_void_p(src.ctypes.data), n, C.c_void_p(out.ctypes.data)) == 0
        assert hashlib.sha256(out.tobytes()).hexdigest() == want, n
        buf = np.concatenate([src[:n], np.zeros(16, dtype=np.uint8)])
        assert L.uaes_gcm_encrypt(256, key, nonce, aad, len(aad), C.c_void_p(buf.ctypes.data), n, C.c_void_p(buf.ctypes.data)) == 0
        assert hashlib.sha256(buf.tobytes()).hexdigest() == want, ("in place", n)
        back = np.full(n, 0xAB, dtype=np.uint8)
        assert L.uaes_gcm_decrypt(256, key, nonce, aad, len(aad), C.c_void_p(out.ctypes.data), n, C.c_void_p(back.ctypes.data)) == 0
        assert back.tobytes() == pt
    code = ("import sys, hashlib, ctypes as C; sys.path.insert(0, %r); import numpy as np, micro_aes_amd as u; L = u.engine(); "
            "n = (40 << 20) + 5; src = (np.arange(n, dtype=np.uint32) * 2654435761 >> 7).astype(np.uint8); out = np.zeros(n + 16, dtype=np.uint8); "
            "assert L.uaes_gcm_encrypt(128, bytes(16), bytes(12), None, 0, C.c_void_p(src.ctypes.data), n, C.c_void_p(out.ctypes.data)) == 0; "
            "print(hashlib.sha256(out.tobytes()).hexdigest())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = []
    for workers in ("1", "4"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, UAES_PIPE_WORKERS=workers))
        assert r.returncode == 0, r.stderr[-1500:]
        got.append(r.stdout.split()[-1])
    assert got[0] == got[1]


def test_gcm_shard_whose_counter_bits_40_47_move_takes_the_two_pass_path(orc):
    """the fused CTR + GHASH pass makes its lane constants once per launch; a shard inside which counter bits 40..47
    move (only reachable 64 GiB into a message: nonce byte 11 = 0xff and a shard offset just below 2^36 bytes) must fall
    back to CTR (which cuts there) + the GHASH levels: ciphertext against the oracle's CTR at that offset, share equal
    to the hash-only share over that ciphertext"""
    import torch
    dev = torch.device("cuda", 0)
    key, nonce = bytes(range(16)), bytes(range(11)) + b"\xff"
    n = 64 << 20
    into = (24 << 20) // 16 + 3                                     # the carry: this many blocks into the shard
    off_blocks = (1 << 32) - 2 - into                               # J0's low 40 bits are ff 00 00 00 01; +1 pre-increment
    off, total = off_blocks * 16, off_blocks * 16 + n + (5 << 30)
    pt = orc.splitmix(99, n)
    src = torch.frombuffer(bytearray(pt), dtype=torch.uint8).to(dev)
    ct = torch.zeros(n, dtype=torch.uint8, device=dev)
    p0, p1 = torch.zeros(16, dtype=torch.uint8, device=dev), torch.zeros(16, dtype=torch.uint8, device=dev)
    uaes.gcm_shard_dev(key, nonce, 0, None, 0, src, n, off, total, ct, p0)
    uaes.gcm_shard_dev(key, nonce, 1, None, 0, ct, n, off, total, None, p1)
    torch.cuda.synchronize()
    want = orc.ctr_xcrypt_at(key, nonce + b"\0\0\0\1", 1 + off_blocks, pt)
    assert hashlib.sha256(ct.cpu().numpy().tobytes()).digest() == hashlib.sha256(want).digest()
    assert bytes(p0.cpu().numpy()) == bytes(p1.cpu().numpy())
    # the shard just BEFORE it (no carry inside) takes the fused pass: same two checks
    off2 = off - n
    uaes.gcm_shard_dev(key, nonce, 0, None, 0, src, n, off2, total, ct, p0)
    uaes.gcm_shard_dev(key, nonce, 1, None, 0, ct, n, off2, total, None, p1)
    torch.cuda.synchronize()
    want = orc.ctr_xcrypt_at(key, nonce + b"\0\0\0\1", 1 + off2 // 16, pt)
    assert hashlib.sha256(ct.cpu().numpy().tobytes()).digest() == hashlib.sha256(want).digest()
    assert bytes(p0.cpu().numpy()) == bytes(p1.cpu().numpy())
