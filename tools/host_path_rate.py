#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer API: the
drop-in AES_CTR_encrypt signature hands over HOST buffers, which the engine
stages through device memory (pageable hipMemcpy in, kernel, hipMemcpy out)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import micro_aes_amd as uaes

L = uaes.engine()
key, iv = bytes(range(16)), bytes(range(0xF0, 0xFC))
for mib in (1, 16, 256, 1024):
    n = mib << 20
    src = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    dst = np.empty_like(src)
    a, b = C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data)
    L.uaes_ctr_xcrypt(128, key, iv, a, n, b)
    reps = 5 if mib < 1024 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        assert L.uaes_ctr_xcrypt(128, key, iv, a, n, b) == 0
    dt = (time.perf_counter() - t0) / reps
    print("host->device->host AES-128-CTR %5d MiB: %8.3f ms  %7.2f GiB/s" % (mib, dt * 1e3, n / dt / 2**30))
