#!/usr/bin/env python3
"""Device-resident throughput of the bulk modes against the text size (warm clocks: half as
many untimed launches as timed ones).  Shows the launch-bound region and the quantisation
of the chunked kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

key, nonce, keys2 = bytes(range(16)), bytes(range(12)), bytes(range(64))
ctr0 = nonce + b"\0\0\0\1"
st = torch.cuda.current_stream()
print("%8s  %s" % ("MiB", "  ".join("%-9s" % w for w in ("ctr", "ecb", "xts4k", "gcm", "ocb"))) + "   (GiB/s)")
for mib in (1, 4, 16, 48, 64, 80, 256, 1024):
    n = mib << 20
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    row = []
    for fn in (lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st),
               lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st),
               lambda: uaes.xts_sectors_dev(keys2, 0, 4096, n // 4096, src, dst, stream=st),
               lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st),
               lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st)):
        reps = max(100, 200 * 64 // mib)
        for _ in range(reps // 2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        row.append(n / ((time.perf_counter() - t0) / reps) / 2**30)
    print("%8d  %s" % (mib, "  ".join("%-9.1f" % v for v in row)))
