#!/usr/bin/env python3
"""One XTS data unit per call (the reference API's shape, AES_XTS_encrypt), device-resident, back to back on one
stream: us per call across the sizes where the call is one launch (k_xts_small, up to 256 KiB) or the tweak pre-pass
+ bulk kernel; both directions, a ragged size (ciphertext stealing) at the end.  ECB beside it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

if os.environ.get("UAES_LIB"):                      # A/B against another build of the library
    uaes.lib_path.__defaults__ = (os.environ["UAES_LIB"],)
keys = bytes(range(32))
print("%9s  %10s %10s %10s" % ("bytes", "ecb", "xts enc", "xts dec"))
SIZES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
    (16, 4096, 16384, 32768, 65536, 131072, 262144, 262144 + 16, 524288, 1 << 20, 4 << 20, 65536 + 7)
for n in SIZES:
    src = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    row = []
    for fn in (lambda: uaes.ecb_dev(keys[:16], src, dst, nbytes=n - n % 16),
               lambda: uaes.xts_sectors_dev(keys, 0, n, 1, src, dst),
               lambda: uaes.xts_sectors_dev(keys, 0, n, 1, src, dst, encrypt=False)):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        row.append("%7.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
    print("%9d  %s" % (n, " ".join("%10s" % r for r in row)), flush=True)
