#!/usr/bin/env python3
"""CTR (and GCM) device-resident throughput against the text size, fine steps: shows whether the
kernel's time follows the size or is quantised.  UAES_CTR_GEO=0 selects the
round-1 dealing of 256 KiB chunks for comparison."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

key, nonce = bytes(range(16)), bytes(range(12))
ctr0 = nonce + b"\0\0\0\1"
st = torch.cuda.current_stream()
sizes = [2, 4, 6, 8, 12, 16, 20, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 192, 256, 384, 512, 1024]
print("UAES_CTR_GEO=%s" % os.environ.get("UAES_CTR_GEO", "(default: stripes)"))
print("%8s  %-9s %-9s  (GiB/s)" % ("MiB", "ctr", "gcm"))
big = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda")
dst = torch.empty((1 << 30) + 16, dtype=torch.uint8, device="cuda")
# clocks: settle once
for _ in range(300):
    uaes.ctr_xcrypt_dev(key, ctr0, 0, big, dst, nbytes=1 << 30, stream=st)
torch.cuda.synchronize()
for mib in sizes:
    n = mib << 20
    row = []
    for fn in (lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, big, dst, nbytes=n, stream=st),
               lambda: uaes.gcm_encrypt_dev(key, nonce, None, big, n, dst, stream=st)):
        reps = max(60, 150 * 64 // mib)
        for _ in range(reps // 2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        row.append(n / ((time.perf_counter() - t0) / reps) / 2**30)
    print("%8d  %-9.1f %-9.1f" % (mib, row[0], row[1]))
