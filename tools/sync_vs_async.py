#!/usr/bin/env python3
"""One long text per call, device pointers: what a SYNCHRONOUS call (uaes_ctr_xcrypt, uaes_gcm_encrypt ...: returns when
the text is done) costs over the same kernel enqueued back to back on a stream (uaes_*_dev).  us per call and the
difference; `few` = the six-call loop of tools/all_modes_rate.py (cold-ish clocks).  UAES_TICKET=0 in the environment
shows hipStreamSynchronize in place of the completion ticket."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
key, nonce = bytes(range(16)), bytes(range(12))
ctr0 = nonce + b"\0\0\0\1"
st = torch.cuda.current_stream()
sizes = [int(x) for x in sys.argv[1:]] or [16, 64, 256, 1024]


def loop(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("# UAES_TICKET=%s" % os.environ.get("UAES_TICKET", "(default 1)"))
print("%6s %-5s %10s %10s %10s %10s" % ("MiB", "mode", "async us", "sync us", "sync-async", "sync few"))
for mib in sizes:
    n = mib << 20
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
    reps = max(60, 6400 // mib)
    for name, fa, fs in (
            ("ctr", lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st),
             lambda: L.uaes_ctr_xcrypt(128, key, ctr0, a, n, b)),
            ("ecb", lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st),
             lambda: L.uaes_ecb_encrypt(128, key, a, n, b)),
            ("gcm", lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st),
             lambda: L.uaes_gcm_encrypt(128, key, nonce, None, 0, a, n, b))):
        ta = loop(fa, reps, reps // 2)
        ts = loop(fs, reps, reps // 2)
        time.sleep(0.05)
        tf = loop(fs, 6, 2)
        print("%6d %-5s %10.1f %10.1f %10.1f %10.1f" % (mib, name, ta, ts, ts - ta, tf), flush=True)
