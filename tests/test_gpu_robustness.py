"""GPU tests of the boundary's promises beyond plain parity: in-place operation
(the reference works in place, micro_aes.h:520-526), unaligned and mixed
host/device pointers, concurrent host threads (the reference is not re-entrant,
micro_aes.c:72 -- this library must be), and streams > 4 GiB (64-bit indexing)."""
import ctypes as C
import hashlib
import os
import random
import threading

import numpy as np
import pytest

import micro_aes_amd as uaes

pytestmark = pytest.mark.gpu


def test_in_place_host_buffers(orc):
    L = uaes.engine()
    rnd = random.Random(11)
    key, keys, iv = rnd.randbytes(16), rnd.randbytes(64), rnd.randbytes(12)
    for n in (16, 100, 4096, 70001):
        data = orc.splitmix(n, n)
        buf = (C.c_uint8 * (n + 32)).from_buffer_copy(data + bytes(32))
        assert L.uaes_ctr_xcrypt(128, key, iv, buf, n, buf) == 0
        assert bytes(buf)[:n] == orc.ctr_encrypt(key, iv, data)
        buf = (C.c_uint8 * (n + 32)).from_buffer_copy(data + bytes(32))
        assert L.uaes_ecb_encrypt(128, key, buf, n, buf) == 0
        assert bytes(buf)[: (n + 15) // 16 * 16] == orc.ecb_encrypt(key, data)
        buf = (C.c_uint8 * (n + 32)).from_buffer_copy(data + bytes(32))
        assert L.uaes_xts_encrypt(256, keys, None, buf, n, buf) == 0
        assert bytes(buf)[:n] == orc.xts(keys, None, data, True)[1]
        assert L.uaes_xts_decrypt(256, keys, None, buf, n, buf) == 0
        assert bytes(buf)[:n] == data
        buf = (C.c_uint8 * (n + 32)).from_buffer_copy(data + bytes(32))
        assert L.uaes_gcm_encrypt(128, key, iv, b"aad", 3, buf, n, buf) == 0
        assert bytes(buf)[: n + 16] == orc.gcm_encrypt(key, iv, b"aad", data)
        assert L.uaes_gcm_decrypt(128, key, iv, b"aad", 3, buf, n, buf) == 0
        assert bytes(buf)[:n] == data


def test_unaligned_and_mixed_pointers(orc):
    import torch
    L = uaes.engine()
    key, iv = bytes(range(16)), bytes(range(12))
    n = 100003
    data = orc.splitmix(3, n)
    want = orc.ctr_encrypt(key, iv, data)
    dev = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
    dev[3:3 + n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda:0")
    # device -> device, both misaligned by different amounts
    assert L.uaes_ctr_xcrypt(128, key, iv, C.c_void_p(dev.data_ptr() + 3), n, C.c_void_p(out.data_ptr() + 5)) == 0
    torch.cuda.synchronize()
    assert bytes(out[5:5 + n].cpu().numpy()) == want and int(out[:5].sum()) == 0 and int(out[5 + n:].sum()) == 0
    # device -> host and host -> device
    hbuf = (C.c_uint8 * n)()
    assert L.uaes_ctr_xcrypt(128, key, iv, C.c_void_p(dev.data_ptr() + 3), n, hbuf) == 0
    assert bytes(hbuf) == want
    out.zero_()
    assert L.uaes_ctr_xcrypt(128, key, iv, data, n, C.c_void_p(out.data_ptr() + 16)) == 0
    torch.cuda.synchronize()
    assert bytes(out[16:16 + n].cpu().numpy()) == want
    # GCM with device AAD and a host payload
    aad = torch.frombuffer(bytearray(b"device resident header"), dtype=torch.uint8).to("cuda:0")
    ct = (C.c_uint8 * (n + 16))()
    assert L.uaes_gcm_encrypt(128, key, iv, C.c_void_p(aad.data_ptr()), aad.numel(), data, n, ct) == 0
    assert bytes(ct) == orc.gcm_encrypt(key, iv, b"device resident header", data)


def test_concurrent_host_threads(orc):
    rnd = random.Random(5)
    jobs = []
    for t in range(8):
        key, iv = rnd.randbytes(16 + 8 * (t % 3)), rnd.randbytes(12)
        data = orc.splitmix(100 + t, 200000 + 17 * t)
        jobs.append((key, iv, data, orc.ctr_encrypt(key, iv, data), orc.gcm_encrypt(key, iv, b"", data[:5000])))
    errors = []

    def work(job):
        key, iv, data, want_ctr, want_gcm = job
        for _ in range(5):
            if uaes.AES_CTR_encrypt(key, iv, data) != want_ctr:
                errors.append("ctr")
            if uaes.AES_GCM_encrypt(key, iv, b"", data[:5000]) != want_gcm:
                errors.append("gcm")

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_record_calls_from_several_threads_share_one_key_context(orc):
    """uaes_gcm_key_*_records only READS the context's tables: six host threads run record calls of different shapes on
    ONE context at the same time (each on its own lane: stream, stagings), every record against the oracle"""
    rnd = random.Random(61)
    key = rnd.randbytes(16)
    k = uaes.GcmKey(key)
    jobs = []
    for t in range(6):
        nrec, rec_len = [(40, 100), (9, 5000), (300, 48), (3, 20000), (64, 1440), (17, 4096)][t]
        nonces = [rnd.randbytes(12) for _ in range(nrec)]
        recs = [rnd.randbytes(rec_len) for _ in range(nrec)]
        aad = rnd.randbytes(t * 3)
        jobs.append((nonces, aad, recs, [orc.gcm_encrypt(key, nonces[r], aad, recs[r]) for r in range(nrec)]))
    errors = []

    def work(job):
        nonces, aad, recs, want = job
        for _ in range(4):
            got = k.encrypt_records(nonces, aad, recs)
            if got != want:
                errors.append("encrypt")
            rc, ver, back = k.decrypt_records(nonces, aad, got)
            if rc != 0 or any(ver) or back != recs:
                errors.append("decrypt")

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    k.close()
    assert not errors


def test_completion_tickets_and_the_plain_wait_agree(orc):
    """a synchronous call ends with a completion ticket: carried by the call's only kernel (ECB,
    short CTR, a one-launch XTS unit, a one-launch GCM encryption), by the ticket kernel behind a multi-launch call,
    or -- a call that runs longer than the spin window -- by hipStreamSynchronize after all.  All three, and the
    build with UAES_TICKET=0, must hand over the same bytes; many short calls from many threads exercise the
    per-lane sequence numbers"""
    import subprocess
    import sys
    code = r'''
import sys, hashlib, threading
sys.path.insert(0, %r)
import micro_aes_amd as uaes
from oracle.pyoracle import Oracle
orc = Oracle()
key, iv, n12 = bytes(range(16)), bytes(range(16, 28)), bytes(range(12))
h = hashlib.sha256()
for n in (16, 4096, 65536, 300000, 3 << 20):                  # one launch .. the spin window is exceeded
    d = orc.splitmix(77, n)
    for out in (uaes.AES_ECB_encrypt(key, d), uaes.AES_CTR_encrypt(key, iv, d), uaes.AES_GCM_encrypt(key, n12, b"a", d),
                uaes.AES_XTS_encrypt(bytes(range(32)), bytes(16), d)[1], uaes.AES_OCB_encrypt(key, n12, b"", d),
                uaes.AES_GCM_decrypt(key, n12, b"a", uaes.AES_GCM_encrypt(key, n12, b"a", d))[1]):
        h.update(out)
    if n <= 65536:
        h.update(uaes.AES_CMAC(key, d)); h.update(uaes.AES_CBC_encrypt(key, bytes(16), d)[1])
bad = []
def work(t):
    d = orc.splitmix(t, 4096 + 16 * t)
    want = orc.ctr_encrypt(key, iv, d), orc.gcm_encrypt(key, n12, b"", d)
    for _ in range(300):
        if uaes.AES_CTR_encrypt(key, iv, d) != want[0] or uaes.AES_GCM_encrypt(key, n12, b"", d) != want[1]:
            bad.append(t)
ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
[t.start() for t in ts]; [t.join() for t in ts]
print(h.hexdigest(), len(bad))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # (default: the ticket rides up to 512 KiB; never; at every size -- round 4's behaviour; no tickets at all)
    for extra in ({"UAES_TICKET": "1"}, {"UAES_TICKET_RIDE_MAX_KIB": "0"}, {"UAES_TICKET_RIDE_MAX_KIB": "1073741824"},
                  {"UAES_TICKET": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split()[-2:])
    assert all(o == outs[0] for o in outs) and outs[0][1] == "0", outs


def test_ctr_stream_larger_than_4GiB(orc):
    """block indices and byte offsets beyond 2^32 (one 5 GiB call)"""
    import torch
    n = 5 << 30
    key, ctr0 = bytes(range(16)), bytes(range(0xF0, 0xFC)) + b"\xff\xff\xff\xf0"    # low word wraps inside the call
    src = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    piece = np.empty(1 << 28, dtype=np.uint8)
    orc.splitmix_into(9, piece)
    pt = torch.from_numpy(piece).to("cuda:0")
    for o in range(0, n, 1 << 28):
        src[o:o + (1 << 28)] = pt
    dst = torch.empty_like(src)
    uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst)
    torch.cuda.synchronize()
    for off in (0, (1 << 32) - 4096, (1 << 32) + 65536, n - 8192):
        got = bytes(dst[off:off + 8192].cpu().numpy())
        want = orc.ctr_xcrypt_at(key, ctr0, off // 16, bytes(src[off:off + 8192].cpu().numpy()))
        assert got == want, off
    uaes.ctr_xcrypt_dev(key, ctr0, 0, dst, dst)
    torch.cuda.synchronize()
    assert torch.equal(dst[: 1 << 30], src[: 1 << 30]) and torch.equal(dst[-(1 << 30):], src[-(1 << 30):])


def test_dev_calls_on_different_streams_do_not_share_scratch(orc):
    """GCM / OCB / XTS *_dev calls enqueued on four streams at once (each call builds its own GHASH
    tables, offsets or chunk tweaks in device scratch) give the results of running them one by one"""
    import torch
    rnd = random.Random(77)
    n = 3 << 20
    jobs = []
    for i in range(4):
        key, nonce = rnd.randbytes(16), rnd.randbytes(12)
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
        jobs.append(dict(key=key, nonce=nonce, keys2=rnd.randbytes(32), src=src, st=torch.cuda.Stream(),
                         gcm=torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"),
                         ocb=torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"),
                         xts=torch.empty(n, dtype=torch.uint8, device="cuda:0")))
    want = []
    for j in jobs:                                   # reference results, one at a time on the default stream
        g, o, x = torch.empty_like(j["gcm"]), torch.empty_like(j["ocb"]), torch.empty_like(j["xts"])
        uaes.gcm_encrypt_dev(j["key"], j["nonce"], None, j["src"], n, g)
        uaes.ocb_dev(j["key"], j["nonce"], None, j["src"], n, o)
        uaes.xts_sectors_dev(j["keys2"], 9, n, 1, j["src"], x)
        torch.cuda.synchronize()
        want.append((g, o, x))
    for _ in range(10):
        for j in jobs:
            j["gcm"].zero_(); j["ocb"].zero_(); j["xts"].zero_()
        torch.cuda.synchronize()
        for j in jobs:
            uaes.gcm_encrypt_dev(j["key"], j["nonce"], None, j["src"], n, j["gcm"], stream=j["st"])
            uaes.ocb_dev(j["key"], j["nonce"], None, j["src"], n, j["ocb"], stream=j["st"])
            uaes.xts_sectors_dev(j["keys2"], 9, n, 1, j["src"], j["xts"], stream=j["st"])
        torch.cuda.synchronize()
        for j, (g, o, x) in zip(jobs, want):
            assert torch.equal(j["gcm"], g) and torch.equal(j["ocb"], o) and torch.equal(j["xts"], x)
    # and the first one against the oracle's tag
    j = jobs[0]
    head = bytes(j["src"][:4096].cpu().numpy())
    assert bytes(want[0][0][:4096].cpu().numpy()) == orc.gcm_encrypt(j["key"], j["nonce"], b"", head)[:4096]


def test_dev_api_rejects_misaligned_pointers():
    import torch
    t = torch.zeros(4096 + 32, dtype=torch.uint8, device="cuda:0")
    key = bytes(16)
    with pytest.raises(uaes.EngineError, match="16-byte aligned"):
        uaes.ecb_dev(key, t[1:1 + 4096], t[16:16 + 4096], nbytes=4096)
    with pytest.raises(uaes.EngineError, match="16-byte aligned"):
        uaes.ctr_xcrypt_dev(key, bytes(16), 0, t[:4096], t[3:3 + 4096], nbytes=4096)
    with pytest.raises(uaes.EngineError, match="16-byte aligned"):
        uaes.ocb_dev(key, bytes(12), None, t[8:8 + 1024], 1024, t[2048:])


def _mgpu_device_lists():
    """device lists for the uaes_mgpu_* tests: {0,0,0} always (slicing, offsets and worker threads on one
    device), and every visible device once plus a doubled list when the box has more than one GPU"""
    import torch
    n = torch.cuda.device_count()
    lists = [[0, 0, 0]]
    if n >= 2:
        lists += [list(range(n)), list(range(n - 1, -1, -1)) + [0]]
    return lists


def _check_mgpu(orc, devlist):
    L = uaes.engine()
    rnd = random.Random(99 + len(devlist))
    key, keys = rnd.randbytes(32), rnd.randbytes(64)
    ctr0 = rnd.randbytes(12) + b"\xff\xff\xff\xf0"
    nd = len(devlist)
    devs = (C.c_int * nd)(*devlist)
    for n in (0, 5, 16, 47, 100003, (3 << 20) + 9):
        data = orc.splitmix(n + 1, n)
        out = (C.c_uint8 * max(n, 1))()
        assert L.uaes_mgpu_ctr_xcrypt_at(nd, devs, 256, key, ctr0, 7, data, n, out) == 0
        assert bytes(out)[:n] == orc.ctr_xcrypt_at(key, ctr0, 7, data), n
        assert L.uaes_mgpu_ctr_xcrypt_at(1, None, 256, key, ctr0, 7, data, n, out) == 0
        assert bytes(out)[:n] == orc.ctr_xcrypt_at(key, ctr0, 7, data), n
    for sb, ns in ((512, 1), (512, 2), (4096 + 17, 7), (16, 100)):
        data = orc.splitmix(sb + ns, sb * ns)
        out = (C.c_uint8 * (sb * ns))()
        assert L.uaes_mgpu_xts_sectors(nd, devs, 256, keys, (1 << 40) + 5, sb, ns, data, out, 1) == 0
        rc, want = orc.xts_sectors(keys, (1 << 40) + 5, sb, data, True)
        assert rc == 0 and bytes(out) == want, (sb, ns)
        back = (C.c_uint8 * (sb * ns))()
        assert L.uaes_mgpu_xts_sectors(nd, devs, 256, keys, (1 << 40) + 5, sb, ns, out, back, 0) == 0
        assert bytes(back) == data
    # a long host text: every device takes its slice through the pipelined host path
    n = (96 << 20) + 16 * 5
    data = orc.splitmix(12345, n)
    out = (C.c_uint8 * n)()
    assert L.uaes_mgpu_ctr_xcrypt_at(nd, devs, 128, key[:16], ctr0, 3, data, n, out) == 0
    assert hashlib.sha256(bytes(out)).digest() == hashlib.sha256(orc.ctr_xcrypt_at(key[:16], ctr0, 3, data)).digest()


def test_single_process_multi_gpu_entry_points(orc):
    """uaes_mgpu_*: slices of one text on several devices from one process.  The device list {0,0,0}
    exercises the slicing, the counter / sector offsets and the per-device worker threads on any box;
    the result must be the single-call result."""
    L = uaes.engine()
    _check_mgpu(orc, [0, 0, 0])
    bad = (C.c_int * 1)(99)
    out = (C.c_uint8 * 16)()
    assert L.uaes_mgpu_ctr_xcrypt_at(1, bad, 128, bytes(16), bytes(16), 0, b"x" * 16, 16, out) == -2       # UAES_E_ARG
    assert b"not one of" in L.uaes_last_error()


def test_single_process_multi_gpu_on_distinct_devices(orc):
    """the same on DISTINCT devices (per-device contexts, hipSetDevice per worker thread, no peer access
    between them): every visible GPU once, and a list that visits them in reverse and device 0 twice"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs: this box has %d" % torch.cuda.device_count())
    for devlist in _mgpu_device_lists()[1:]:
        _check_mgpu(orc, devlist)


def test_more_streams_than_scratch_slots_from_several_threads(orc):
    """Twelve streams (the engine keeps 8 scratch slots per device) driven by four host threads:
    a slot handed to a caller stays pinned until its launch is issued, so recycling the least
    recently used slot can never take a buffer another thread is about to launch on (round-1
    advisor finding).  Every result must equal the one-at-a-time result; then the slots are
    given back with uaes_stream_release."""
    import torch
    rnd = random.Random(4242)
    n = 1 << 20
    streams = [torch.cuda.Stream() for _ in range(12)]
    jobs = []
    for i, st in enumerate(streams):
        key, nonce, keys2 = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(64)
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
        g, x = torch.empty(n + 16, dtype=torch.uint8, device="cuda:0"), torch.empty(n, dtype=torch.uint8, device="cuda:0")
        uaes.gcm_encrypt_dev(key, nonce, None, src, n, g)
        uaes.xts_sectors_dev(keys2, i, 4096, n // 4096, src, x)
        torch.cuda.synchronize()
        jobs.append(dict(key=key, nonce=nonce, keys2=keys2, src=src, st=st, first=i, want=(g, x),
                         gcm=torch.zeros_like(g), xts=torch.zeros_like(x)))
    errors = []

    def worker(mine):
        try:
            for _ in range(25):
                for j in mine:
                    uaes.gcm_encrypt_dev(j["key"], j["nonce"], None, j["src"], n, j["gcm"], stream=j["st"])
                    uaes.xts_sectors_dev(j["keys2"], j["first"], 4096, n // 4096, j["src"], j["xts"], stream=j["st"])
        except Exception as e:            # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(jobs[k::4],)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for j in jobs:
        assert torch.equal(j["gcm"], j["want"][0]) and torch.equal(j["xts"], j["want"][1])
    L = uaes.engine()
    for st in streams:
        assert L.uaes_stream_release(C.c_void_p(st.cuda_stream)) == 0
    assert L.uaes_stream_release(C.c_void_p(streams[0].cuda_stream)) == 0      # idempotent
    # and the streams still work afterwards
    j = jobs[0]
    j["gcm"].zero_()
    uaes.gcm_encrypt_dev(j["key"], j["nonce"], None, j["src"], n, j["gcm"], stream=j["st"])
    torch.cuda.synchronize()
    assert torch.equal(j["gcm"], j["want"][0])


def test_wipe_on_auth_failure_switch(orc):
    """CCM / GCM-SIV / OCB decrypt before they authenticate.  Default = the reference's default
    build: 0x1A with the text in place.  uaes_set_wipe_on_auth_failure(1) = its INCREASE_SECURITY
    build: 0x1A with zeros.  GCM never writes (N7) either way."""
    L = uaes.engine()
    rnd = random.Random(31)
    key, n12, n11 = rnd.randbytes(16), rnd.randbytes(12), rnd.randbytes(11)
    data = rnd.randbytes(1000)
    cases = [(uaes.AES_CCM_encrypt, uaes.AES_CCM_decrypt, n11), (uaes.GCM_SIV_encrypt, uaes.GCM_SIV_decrypt, n12),
             (uaes.AES_OCB_encrypt, uaes.AES_OCB_decrypt, n12)]
    try:
        for enc, dec, nonce in cases:
            bad = bytearray(enc(key, nonce, b"hdr", data))
            bad[-1] ^= 1
            assert L.uaes_set_wipe_on_auth_failure(0) in (0, 1)
            rc, text = dec(key, nonce, b"hdr", bytes(bad), prefill=0xCC)
            assert rc == 0x1A and text != b"\xcc" * len(data) and text != bytes(len(data))    # released, as the reference does
            assert L.uaes_set_wipe_on_auth_failure(1) == 0
            rc, text = dec(key, nonce, b"hdr", bytes(bad), prefill=0xCC)
            assert rc == 0x1A and text == bytes(len(data))
            good = enc(key, nonce, b"hdr", data)
            assert dec(key, nonce, b"hdr", good) == (0, data)                               # unaffected
            assert L.uaes_set_wipe_on_auth_failure(0) == 1
        bad = bytearray(uaes.AES_GCM_encrypt(key, n12, b"hdr", data))
        bad[3] ^= 1
        assert uaes.AES_GCM_decrypt(key, n12, b"hdr", bytes(bad), prefill=0xCC) == (0x1A, b"\xcc" * len(data))
    finally:
        L.uaes_set_wipe_on_auth_failure(0)


@pytest.mark.parametrize("sector_bytes,nsectors", [(17, 64), (31, 9), (520, 33), (4099, 70), (65537, 3)])
def test_xts_units_of_ragged_size_on_device_pointers(orc, sector_bytes, nsectors):
    """data units whose size is not a multiple of 16: from the second unit on every block sits at
    an odd address (the kernel then uses byte-aligned 16-byte accesses), both directions, in place"""
    import torch
    keys = bytes(range(1, 65))
    total = sector_bytes * nsectors
    data = orc.splitmix(sector_bytes, (total + 7) // 8 * 8)[:total]
    src = torch.frombuffer(bytearray(data + bytes(16)), dtype=torch.uint8).to("cuda:0")
    dst = torch.zeros_like(src)
    uaes.xts_sectors_dev(keys, 1 << 33, sector_bytes, nsectors, src, dst, encrypt=True)
    torch.cuda.synchronize()
    rc, want = orc.xts_sectors(keys, 1 << 33, sector_bytes, data, True)
    assert rc == 0 and bytes(dst[:total].cpu().numpy()) == want and int(dst[total:].sum()) == 0
    uaes.xts_sectors_dev(keys, 1 << 33, sector_bytes, nsectors, dst, dst, encrypt=False)
    torch.cuda.synchronize()
    assert bytes(dst[:total].cpu().numpy()) == data


def test_long_host_texts_go_through_the_slice_pipeline(orc):
    """Host buffers of 32 MiB and more are cut into 16 MiB slices that four worker threads move
    through the GPU concurrently (CTR with per-slice counter offsets, ECB with the ragged/padded
    tail in the last slice, XTS by whole data units); results must equal the oracle's / the
    one-piece device path's, out of place and in place, also from two calling threads at once."""
    import hashlib
    import torch
    L = uaes.engine()
    rnd = random.Random(808)
    key, keys, iv = rnd.randbytes(24), rnd.randbytes(64), rnd.randbytes(12)
    n = (40 << 20) + 21
    data = orc.splitmix(5, (n + 7) // 8 * 8)[:n]
    src = (C.c_uint8 * n).from_buffer_copy(data)
    out = (C.c_uint8 * (n + 32))()
    sha = lambda b: hashlib.sha256(b).digest()
    # CTR, a start counter that carries out of 32 bits inside the text
    ctr0 = iv[:9] + bytes.fromhex("00ffffffff00") + b"\xf0"
    assert L.uaes_ctr_xcrypt_at(192, key, ctr0, 5, src, n, out) == 0
    assert sha(bytes(out)[:n]) == sha(orc.ctr_xcrypt_at(key, ctr0, 5, data))
    # ECB: ragged tail (zero padded) and PKCS#7
    for padding in (0, 1):
        want = orc.ecb_encrypt(key, data, padding=padding)
        assert L.uaes_ecb_encrypt_padded(192, key, padding, src, n, out) == 0
        assert sha(bytes(out)[: len(want)]) == sha(want)
    whole = n // 16 * 16
    back = (C.c_uint8 * whole)()
    assert L.uaes_ecb_decrypt(192, key, out, whole, back) == 0 and bytes(back) == data[:whole]
    # XTS data units of 4 KiB + 16 (slices are whole units), in place
    sb, ns = 4096 + 16, 9000
    buf = (C.c_uint8 * (sb * ns)).from_buffer_copy(orc.splitmix(6, sb * ns))
    plain = bytes(buf)
    assert L.uaes_xts_sectors(256, keys, 1 << 40, sb, ns, buf, buf, 1) == 0
    dsrc = torch.frombuffer(bytearray(plain), dtype=torch.uint8).to("cuda:0")
    ddst = torch.empty_like(dsrc)
    uaes.xts_sectors_dev(keys, 1 << 40, sb, ns, dsrc, ddst)
    torch.cuda.synchronize()
    assert sha(bytes(buf)) == sha(ddst.cpu().numpy().tobytes())
    assert bytes(buf)[-2 * sb:] == orc.xts_sectors(keys, (1 << 40) + ns - 2, sb, plain[-2 * sb:], True)[1]
    assert L.uaes_xts_sectors(256, keys, 1 << 40, sb, ns, buf, buf, 0) == 0 and bytes(buf) == plain
    # two threads at once (the second waits for the pipeline, or takes it first)
    res = {}

    def call(tag, off):
        o = (C.c_uint8 * n)()
        res[tag] = (L.uaes_ctr_xcrypt_at(192, key, ctr0, off, src, n, o), sha(bytes(o)))

    th = [threading.Thread(target=call, args=("a", 5)), threading.Thread(target=call, args=("b", 6))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res["a"] == (0, sha(orc.ctr_xcrypt_at(key, ctr0, 5, data)))
    assert res["b"] == (0, sha(orc.ctr_xcrypt_at(key, ctr0, 6, data)))


def test_lanes_of_short_lived_threads_and_shutdown(orc):
    """Every host thread gets its own lane (stream, staging, pinned buffers, scratch) for the synchronous
    API; a lane goes away with its thread, and uaes_shutdown() returns everything on every device, after
    which the next call sets the library up again -- results unchanged throughout."""
    L = uaes.engine()
    key, iv = bytes(range(16)), bytes(range(12))
    data = orc.splitmix(5, 70001)
    want_ctr = orc.ctr_encrypt(key, iv, data)
    want_gcm = orc.gcm_encrypt(key, iv, b"hdr", data)
    errs = []

    def work():
        try:
            for _ in range(3):
                assert uaes.AES_CTR_encrypt(key, iv, data) == want_ctr
                assert uaes.AES_GCM_encrypt(key, iv, b"hdr", data) == want_gcm
                assert uaes.AES_GCM_decrypt(key, iv, b"hdr", want_gcm) == (0, data)
        except Exception as e:               # noqa: BLE001
            errs.append(repr(e))

    for _ in range(6):                       # 48 threads come and go
        ts = [threading.Thread(target=work) for _ in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errs, errs[:3]
    assert L.uaes_shutdown() == 0
    assert uaes.AES_CTR_encrypt(key, iv, data) == want_ctr          # this thread's emptied lane is rebuilt
    ts = [threading.Thread(target=work) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]
    assert L.uaes_shutdown() == 0 and L.uaes_shutdown() == 0       # idempotent
    assert uaes.AES_XTS_encrypt(bytes(range(32)), bytes(16), data) == orc.xts(bytes(range(32)), bytes(16), data, True, prefill=0)


def test_sync_api_orders_after_the_callers_default_stream_work(orc):
    """a synchronous call handed DEVICE memory runs on the thread's own (non-blocking) stream; it must
    still see what the caller queued on the default stream just before"""
    import torch
    key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
    n = 64 << 20
    Lh = uaes.engine()
    for rep in range(5):
        a = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        for _ in range(4):                   # a queue of default-stream work ending in the fill we need
            a.add_(1)
        a.fill_(rep + 1)
        out = torch.empty_like(a)
        assert Lh.uaes_ctr_xcrypt_at(128, key, ctr0, 0, C.c_void_p(a.data_ptr()), n, C.c_void_p(out.data_ptr())) == 0
        got = bytes(out[:4096].cpu().numpy())
        assert got == orc.ctr_xcrypt_at(key, ctr0, 0, bytes([rep + 1]) * 4096)
        tail = bytes(out[n - 4096:].cpu().numpy())
        assert tail == orc.ctr_xcrypt_at(key, ctr0, (n - 4096) // 16, bytes([rep + 1]) * 4096)


def _gather_check(orc, devlist, root, n, with_own_buffers):
    """uaes_mgpu_ctr_encrypt_gather on device-resident shards: the gathered text is the single-call result"""
    import torch
    L = uaes.engine()
    nd = len(devlist)
    key, ctr0 = bytes(range(32)), bytes(range(12)) + b"\xff\xff\xff\xfe"
    data = orc.splitmix(n + 3, n)
    blocks = (n + 15) // 16
    ins, outs, keep = (C.c_void_p * nd)(), (C.c_void_p * nd)(), []
    for i, d in enumerate(devlist):
        lo, hi = min(n, 16 * (blocks * i // nd)), min(n, 16 * (blocks * (i + 1) // nd))
        t = torch.frombuffer(bytearray(data[lo:hi] or b"\0"), dtype=torch.uint8).to("cuda:%d" % d)
        o = torch.full((max(hi - lo, 1),), 0xEE, dtype=torch.uint8, device="cuda:%d" % d)
        keep += [t, o]
        ins[i] = t.data_ptr()
        outs[i] = o.data_ptr() if (with_own_buffers or d != devlist[root]) else None
    full = torch.full((max(n, 1) + 32,), 0xDD, dtype=torch.uint8, device="cuda:%d" % devlist[root])
    devs = (C.c_int * nd)(*devlist)
    rc = L.uaes_mgpu_ctr_encrypt_gather(nd, devs, 256, key, ctr0, 9, ins, n, outs, root, full.data_ptr())
    assert rc == 0, L.uaes_last_error()
    for d in set(devlist):
        torch.cuda.synchronize(d)
    got = bytes(full.cpu().numpy())
    assert got[:n] == orc.ctr_xcrypt_at(key, ctr0, 9, data), (devlist, root, n)
    assert got[n:n + 32] == b"\xdd" * 32


def test_mgpu_ctr_encrypt_gather_on_one_device(orc):
    """the C-host gather call (BASELINE configs[4] in one call) where every slice lives on the root's device: no RCCL
    involved, the gather degenerates to in-place encryption or device-to-device copies.  Also: a device list that is
    not visible, a root outside the list, and -- when a second device is missing -- that nothing tried to load RCCL."""
    L = uaes.engine()
    for n in (0, 5, 16, 1000, (2 << 20) + 7):
        _gather_check(orc, [0], 0, n, False)
        _gather_check(orc, [0, 0, 0], 1, n, False)
        _gather_check(orc, [0, 0, 0], 2, n, True)
    one = (C.c_void_p * 1)(1)
    assert L.uaes_mgpu_ctr_encrypt_gather(1, (C.c_int * 1)(0), 128, bytes(16), bytes(16), 0, one, 16, one, 3, 1) == -2
    assert b"root" in L.uaes_last_error()
    assert L.uaes_mgpu_ctr_encrypt_gather(1, (C.c_int * 1)(77), 128, bytes(16), bytes(16), 0, one, 16, one, 0, 1) == -2


def test_mgpu_ctr_encrypt_gather_rccl(orc):
    """the real thing: slices on distinct devices, gathered on the root by RCCL send / receive over xGMI"""
    import torch
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("needs 2 GPUs: this box has %d" % nd)
    for n in (16 * nd, 100003, (64 << 20) + 5):
        _gather_check(orc, list(range(nd)), 0, n, True)
        _gather_check(orc, list(range(nd - 1, -1, -1)), nd - 1, n, True)
        _gather_check(orc, [0, 1, 0, 1], 3, n, True)


def _gather_stats():
    out = (C.c_ulong * 5)()
    uaes.engine().uaes_debug_gather_stats(out)
    return dict(zip(("sends", "recvs", "groups", "inits", "failures"), out))


@pytest.fixture
def force_rccl(monkeypatch):
    """UAES_GATHER_FORCE_RCCL=1 for one test (include/uaes_hip.h, "Test hooks of the gather")"""
    monkeypatch.setenv("UAES_GATHER_FORCE_RCCL", "1")
    monkeypatch.delenv("UAES_GATHER_FAIL_SEND", raising=False)
    yield monkeypatch


def test_mgpu_ctr_encrypt_gather_through_rccl_on_one_device(orc, force_rccl):
    """the RCCL leg of the C host's gather (dlopen of librccl, the hand-declared prototypes,
    ncclUint8 = 1, the communicator cache, grouped send / receive, the drain) had never executed on a one-GPU box:
    same-device slices took the hipMemcpy shortcut.  Under UAES_GATHER_FORCE_RCCL every slice that has a shard buffer
    of its own travels rank r -> rank r by ncclSend / ncclRecv on a ONE-rank communicator; the gathered text must
    still be the single-call result (shard arithmetic: micro_aes.c:421-427, :962-976), and the counters must show
    that RCCL carried it."""
    before = _gather_stats()
    for n in (16, 1000, 100003, (2 << 20) + 7):
        _gather_check(orc, [0], 0, n, True)               # one slice, one self send / receive
    mid = _gather_stats()
    assert mid["sends"] - before["sends"] == 4 and mid["recvs"] - before["recvs"] == 4, (before, mid)
    assert mid["groups"] - before["groups"] == 4 and mid["failures"] == before["failures"]
    assert mid["inits"] - before["inits"] <= 1            # the communicator is cached across calls
    for n in (48, 100003, (8 << 20) + 5):
        _gather_check(orc, [0, 0, 0], 1, n, True)         # three slices, three pairs in ONE group on the same rank
    after = _gather_stats()
    assert after["sends"] - mid["sends"] == 9 and after["groups"] - mid["groups"] == 3
    assert after["inits"] == mid["inits"], "the device list {0,0,0} reduces to the cached one-rank communicator"
    # a root slice without a buffer of its own is encrypted straight into place: nothing to send even when forced
    _gather_check(orc, [0], 0, 4096, False)
    assert _gather_stats()["sends"] == after["sends"]


def test_mgpu_gather_c2_digest_through_rccl(orc, force_rccl):
    """the whole C2 text (AES-128-CTR, 1 GiB, seed 2) encrypted as four shards on device 0 and gathered through RCCL
    self send / receive: SHA-256 of the gathered stream == the compiled reference's digest (tests/golden/digests.json)"""
    import hashlib
    import json
    import torch
    import bench
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests.json")) as f:
        gold = json.load(f)["C2_ctr128_1GiB_seed2"]["sha256"]
    L = uaes.engine()
    n, nd = 1 << 30, 4
    dev = torch.device("cuda", 0)
    src = bench.splitmix_device(torch, 2, n, 0, dev)
    shard = n // nd
    outs_t = [torch.empty(shard, dtype=torch.uint8, device=dev) for _ in range(nd)]
    full = torch.zeros(n, dtype=torch.uint8, device=dev)
    ins, outs = (C.c_void_p * nd)(), (C.c_void_p * nd)()
    for i in range(nd):
        ins[i], outs[i] = src.data_ptr() + i * shard, outs_t[i].data_ptr()
    torch.cuda.synchronize()
    before = _gather_stats()
    rc = L.uaes_mgpu_ctr_encrypt_gather(nd, (C.c_int * nd)(*([0] * nd)), 128, bench.KEY16, bench.CTR0, 0, ins, n, outs,
                                        0, C.c_void_p(full.data_ptr()))
    assert rc == 0, L.uaes_last_error()
    torch.cuda.synchronize()
    after = _gather_stats()
    assert after["sends"] - before["sends"] == nd and after["recvs"] - before["recvs"] == nd
    h = hashlib.sha256()
    for o in range(0, n, 1 << 28):
        h.update(full[o:o + (1 << 28)].cpu().numpy().tobytes())
    assert h.hexdigest() == gold


def test_mgpu_gather_failed_send_returns_an_error_and_drains(orc, force_rccl):
    """a send RCCL refuses in the middle of an open group (UAES_GATHER_FAIL_SEND: the k-th ncclSend names a rank the
    communicator does not have): the call returns UAES_E_HIP with RCCL's own message, does not hang, has drained
    every gather stream before returning (the caller may free or reuse the buffers at once -- done here), and the
    NEXT call builds fresh communicators and is correct."""
    import torch
    L = uaes.engine()
    n, nd = (4 << 20) + 16, 3
    key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
    data = orc.splitmix(77, n)
    dev = torch.device("cuda", 0)
    for fail_at in (1, 2, 3):
        force_rccl.setenv("UAES_GATHER_FAIL_SEND", str(fail_at))
        before = _gather_stats()
        blocks = (n + 15) // 16
        ins, outs, keep = (C.c_void_p * nd)(), (C.c_void_p * nd)(), []
        for i in range(nd):
            lo, hi = 16 * (blocks * i // nd), min(n, 16 * (blocks * (i + 1) // nd))
            t = torch.frombuffer(bytearray(data[lo:hi]), dtype=torch.uint8).to(dev)
            o = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
            keep += [t, o]
            ins[i], outs[i] = t.data_ptr(), o.data_ptr()
        full = torch.zeros(n, dtype=torch.uint8, device=dev)
        rc = L.uaes_mgpu_ctr_encrypt_gather(nd, (C.c_int * nd)(0, 0, 0), 128, key, ctr0, 0, ins, n, outs, 0,
                                            C.c_void_p(full.data_ptr()))
        assert rc == -1, (fail_at, rc)                    # UAES_E_HIP
        msg = L.uaes_last_error()
        assert b"ncclSend" in msg or b"ncclGroupEnd" in msg, msg
        after = _gather_stats()
        assert after["failures"] - before["failures"] == 1
        assert after["sends"] - before["sends"] == fail_at - 1, "the sends before the refused one were accepted"
        # drained: the buffers go back to the allocator and are overwritten at once; a transfer still in flight
        # would race with this and, worse, with the next gather
        del keep, full
        torch.cuda.synchronize()
        junk = torch.full(((16 << 20),), 0x5A, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        del junk
        # every shard WAS encrypted (the failure is in the gather): nothing to assert on `full`, it is undefined
        force_rccl.delenv("UAES_GATHER_FAIL_SEND")
        mid = _gather_stats()
        _gather_check(orc, [0, 0, 0], 2, n, True)
        end = _gather_stats()
        assert end["inits"] - mid["inits"] == 1, "the communicators of a failed gather are not reused"
        assert end["failures"] == mid["failures"]


def test_mgpu_gather_says_why_when_rccl_is_missing(orc):
    """RCCL is a soft dependency (dlopen): a library that does not load is UAES_E_HIP with the loader's message,
    in a fresh process so that this process' loaded RCCL is not disturbed"""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import micro_aes_amd as uaes\n"
        "L = uaes.engine()\n"
        "t = torch.zeros(64, dtype=torch.uint8, device='cuda:0'); o = torch.zeros(64, dtype=torch.uint8, device='cuda:0')\n"
        "f = torch.zeros(64, dtype=torch.uint8, device='cuda:0')\n"
        "ins = (C.c_void_p * 1)(t.data_ptr()); outs = (C.c_void_p * 1)(o.data_ptr())\n"
        "rc = L.uaes_mgpu_ctr_encrypt_gather(1, (C.c_int * 1)(0), 128, bytes(16), bytes(16), 0, ins, 64, outs, 0, C.c_void_p(f.data_ptr()))\n"
        "print(rc, L.uaes_last_error().decode())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UAES_GATHER_FORCE_RCCL="1", UAES_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("-1 RCCL is not available"), r.stdout


def test_mgpu_calls_from_several_host_threads_share_the_persistent_workers(orc):
    """the per-device workers of uaes_mgpu_* now live for the life of the process and calls queue up per
    worker.  Six host threads call at once with device lists that all name device 0 several times: every call must
    get its own result (jobs of different calls interleave on the one worker), repeatedly."""
    import threading
    L = uaes.engine()
    key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\7"
    errs = []

    def worker(t):
        try:
            rnd = random.Random(500 + t)
            for rep in range(6):
                n = rnd.choice([0, 16, 1000, 70001, (1 << 20) + 3])
                data = orc.splitmix(1000 * t + rep, n)
                nd = rnd.choice([1, 2, 3, 5])
                devs = (C.c_int * nd)(*([0] * nd))
                out = (C.c_uint8 * max(n, 1))()
                rc = L.uaes_mgpu_ctr_xcrypt_at(nd, devs, 128, key, ctr0, t, data, n, out)
                assert rc == 0, L.uaes_last_error()
                assert bytes(out)[:n] == orc.ctr_xcrypt_at(key, ctr0, t, data), (t, rep, n, nd)
        except Exception as e:                      # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(300)
    assert not errs, errs
    assert all(not th.is_alive() for th in ths)


def test_one_launch_gcm_under_oversubscription(orc):
    """VERDICT r05 weak #5 / ADVICE r05: the one-launch GCM arrangement must not depend on the order or the concurrency
    in which the device runs the workgroups of a launch.  Here the device is kept busy by a bulk CTR stream (every CU
    occupied by 1 GiB passes) while four host threads, each on a stream of its own, send mid-sized GCM calls (chunk
    workgroups + a preparing workgroup in one launch) with the DEFAULT look: whoever of a launch's workgroups arrives
    last folds.  Every tag and ciphertext against the oracle, every decryption accepted, a forgery refused; nothing
    hangs (the whole test is bounded) and the arrival counters are back at zero afterwards (a second round on the same
    streams is exact too)."""
    import torch
    L = uaes.engine()
    dev = torch.device("cuda", 0)
    key, nonce = bytes(range(16)), bytes(range(12))
    big_src = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device=dev)
    big_dst = torch.empty_like(big_src)
    bg = torch.cuda.Stream(device=dev)
    stop, errs = threading.Event(), []
    folds0 = C.c_uint(0)
    assert L.uaes_debug_gcm_chunk_folds(C.byref(folds0)) == 0

    def background():
        try:
            while not stop.is_set():
                for _ in range(8):
                    uaes.ctr_xcrypt_dev(key, nonce + b"\0\0\0\1", 0, big_src, big_dst, nbytes=1 << 30, stream=bg)
                bg.synchronize()
        except Exception as e:                      # noqa: BLE001
            errs.append("bg: %r" % e)

    sizes = [40000, (256 << 10) + 5, (1 << 20) - 16, (3 << 20) + 48, 6 << 20]
    cases = {}
    for n in sizes:
        data, aad = orc.splitmix(n + 11, n), bytes([n & 0xff]) * 19
        cases[n] = (data, aad, orc.gcm_encrypt(key, nonce, aad, data))

    def worker(t):
        try:
            st = torch.cuda.Stream(device=dev)
            status = torch.full((1,), -1, dtype=torch.int32, device=dev)
            for rep in range(2):
                for n in sizes[t % 2:] + sizes[:t % 2]:
                    data, aad, want = cases[n]
                    with torch.cuda.stream(st):
                        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev, non_blocking=False)
                        d_aad = torch.frombuffer(bytearray(aad), dtype=torch.uint8).to(dev)
                        d_out = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
                        back = torch.full((n,), 0xCC, dtype=torch.uint8, device=dev)
                    st.synchronize()
                    for _ in range(6):
                        uaes.gcm_encrypt_dev(key, nonce, d_aad, d_in, n, d_out, stream=st)
                    uaes.gcm_decrypt_dev(key, nonce, d_aad, d_out, n, back, status, stream=st)
                    st.synchronize()
                    assert bytes(d_out.cpu().numpy()) == want, (t, rep, n)
                    assert int(status.item()) == 0 and bytes(back.cpu().numpy()) == data, (t, rep, n)
                    d_out[n // 2] ^= 1
                    back.fill_(0xCC)
                    uaes.gcm_decrypt_dev(key, nonce, d_aad, d_out, n, back, status, stream=st)
                    st.synchronize()
                    assert int(status.item()) == 0x1A and int((back != 0xCC).sum()) == 0, (t, rep, n)
        except Exception as e:                      # noqa: BLE001
            errs.append("worker %d: %r" % (t, e))

    bgt = threading.Thread(target=background)
    bgt.start()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(240)
    stop.set()
    bgt.join(60)
    assert not errs, errs
    assert all(not th.is_alive() for th in ths) and not bgt.is_alive(), "a call did not come back"
    folds1 = C.c_uint(0)
    assert L.uaes_debug_gcm_chunk_folds(C.byref(folds1)) == 0
    # informational: how often a chunk workgroup (not the preparing one) was the last to arrive under this load
    print("folds done by a chunk workgroup under oversubscription: %d" % (folds1.value - folds0.value))


def test_dev_calls_are_capturable_into_a_hip_graph(orc):
    """The *_dev entry points only enqueue kernels on the caller's stream -- no allocation after the first call on a
    stream, no synchronisation, round keys and counters passed by value -- so a launch-bound sequence of small calls can
    be captured into a hipGraph once and replayed (the one-launch GCM arrangement restores its arrival counter itself,
    so a replay finds it at zero).  Sixteen calls of each bulk mode, captured and replayed three times over fresh
    input: every replay's output against the oracle."""
    import torch
    key, keys2, nonce = bytes(range(16)), bytes(range(64)), bytes(range(12))
    ctr0 = nonce + b"\0\0\0\1"
    N, M = 16, 64 << 10
    src = torch.zeros(N, M, dtype=torch.uint8, device="cuda:0")
    out = {m: torch.zeros(N, M + 16, dtype=torch.uint8, device="cuda:0") for m in ("ctr", "xts", "gcm", "ocb")}
    side = torch.cuda.Stream()

    def sequence(st):
        for i in range(N):
            uaes.ctr_xcrypt_dev(key, ctr0, i * (M // 16), src[i], out["ctr"][i, :M], nbytes=M, stream=st)
            uaes.xts_sectors_dev(keys2, i * (M // 4096), 4096, M // 4096, src[i], out["xts"][i, :M], stream=st)
            uaes.gcm_encrypt_dev(key, nonce, None, src[i], M, out["gcm"][i], stream=st)
            uaes.ocb_dev(key, nonce, None, src[i], M, out["ocb"][i], stream=st)

    with torch.cuda.stream(side):
        sequence(side)                                      # warm-up on THIS stream: its scratch slot, LDS attributes
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        sequence(torch.cuda.current_stream())
    for rep in range(3):
        data = orc.splitmix(900 + rep, N * M)
        src.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8).view(N, M))
        for t in out.values():
            t.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for i in (0, 7, N - 1):
            piece = data[i * M:(i + 1) * M]
            assert bytes(out["ctr"][i, :M].cpu().numpy()) == orc.ctr_xcrypt_at(key, ctr0, i * (M // 16), piece), (rep, i)
            assert bytes(out["xts"][i, :M].cpu().numpy()) == orc.xts_sectors(keys2, i * (M // 4096), 4096, piece, True)[1], (rep, i)
            assert bytes(out["gcm"][i].cpu().numpy()) == orc.gcm_encrypt(key, nonce, b"", piece), (rep, i)
            assert bytes(out["ocb"][i].cpu().numpy()) == orc.ocb_encrypt(key, nonce, b"", piece), (rep, i)
