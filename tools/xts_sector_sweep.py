#!/usr/bin/env python3
"""Bulk XTS over many data units per call (uaes_xts_*_sectors, device-resident): GiB/s by unit (sector) size for a
fixed total, both directions.  UAES_LIB=<file> measures another build of the library."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

if os.environ.get("UAES_LIB"):
    uaes.lib_path.__defaults__ = (os.environ["UAES_LIB"],)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
keys = bytes(range(64))
src = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
print("# %d MiB per call, AES-256-XTS, %s" % (total >> 20, os.environ.get("UAES_LIB", "libuaes_hip.so")))
print("%9s  %12s %12s" % ("unit", "enc GiB/s", "dec GiB/s"))
for unit in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20):
    row = []
    for enc in (True, False):
        fn = lambda: uaes.xts_sectors_dev(keys, 5, unit, total // unit, src, dst, encrypt=enc)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        row.append(total / ((time.perf_counter() - t0) / reps) / 2**30)
    print("%9d  %12.1f %12.1f" % (unit, row[0], row[1]), flush=True)
