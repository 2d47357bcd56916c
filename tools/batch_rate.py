#!/usr/bin/env python3
"""uaes_cbc_encrypt_batch / uaes_cmac_batch (independent chains, device pointers): GiB/s by message size and count."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

if os.environ.get("UAES_LIB"):
    uaes.lib_path.__defaults__ = (os.environ["UAES_LIB"],)
L = uaes.engine()
key = bytes(range(16))
cap = 256 << 20
src = torch.randint(0, 256, (cap,), dtype=torch.uint8, device="cuda")
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
ivs = torch.randint(0, 256, (16 << 20,), dtype=torch.uint8, device="cuda")
a, b, iv = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(ivs.data_ptr())
print("# %s" % os.environ.get("UAES_LIB", "libuaes_hip.so"))
print("%10s %9s  %12s %12s" % ("msg bytes", "messages", "cbc GiB/s", "cmac GiB/s"))
for msg, cnt in ((64, 1 << 20), (512, 1 << 19), (4096, 65536), (4096, 16384), (4096, 4096), (4096, 1024), (65536, 4096),
                 (65536, 1024), (65536, 256), (1 << 20, 256), (1 << 20, 64)):
    row = []
    for fn in (lambda: L.uaes_cbc_encrypt_batch(128, key, iv, cnt, msg, a, b), lambda: L.uaes_cmac_batch(128, key, cnt, msg, a, b)):
        assert fn() == 0
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        row.append(msg * cnt / ((time.perf_counter() - t0) / reps) / 2**30)
    print("%10d %9d  %12.1f %12.1f" % (msg, cnt, row[0], row[1]), flush=True)
