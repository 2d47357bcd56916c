#!/usr/bin/env python3
"""Per-call latency of the drop-in boundary for small buffers (BASELINE config C1 is a
4 KiB ECB call): host pointers (staged through device memory) and device pointers."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import micro_aes_amd as uaes

L = uaes.engine()
key, iv, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC)), bytes(range(12))
keys2 = bytes(range(32))


def bench(name, fn, reps=300):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (16, 4096, 65536, 1 << 20):
    src = np.random.default_rng(1).integers(0, 256, n + 16, dtype=np.uint8)
    dst = np.empty(n + 32, dtype=np.uint8)
    a, b = C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data)
    dsrc = torch.from_numpy(src).cuda()
    ddst = torch.empty(n + 32, dtype=torch.uint8, device="cuda")
    da, db = C.c_void_p(dsrc.data_ptr()), C.c_void_p(ddst.data_ptr())
    row = []
    for tag, x, y in (("host", a, b), ("dev ", da, db)):
        r = {
            "ecb": bench("ecb", lambda: L.uaes_ecb_encrypt(128, key, x, n, y)),
            "ctr": bench("ctr", lambda: L.uaes_ctr_xcrypt(128, key, iv, x, n, y)),
            "xts": bench("xts", lambda: L.uaes_xts_encrypt(128, keys2, None, x, n, y)),
            "gcm": bench("gcm", lambda: L.uaes_gcm_encrypt(128, key, nonce, None, 0, x, n, y)),
            "ocb": bench("ocb", lambda: L.uaes_ocb_encrypt(128, key, nonce, None, 0, x, n, y)),
        }
        print("%8d B  %s pointers, us per call: " % (n, tag) + "  ".join("%s %7.1f" % kv for kv in r.items()))
