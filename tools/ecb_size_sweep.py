#!/usr/bin/env python3
"""AES-128-ECB (encrypt, decrypt) and AES-256-XTS (4 KiB sectors) device-resident GiB/s at sizes around the round quanta
(ECB: 256 workgroups x 64 KiB tiles = 16 MiB per round; XTS: 4096 waves x 4 KiB chunks = 16 MiB per round)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, micro_aes_amd as uaes
key, keys = bytes(range(16)), bytes(range(64))
st = torch.cuda.current_stream()
big = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda"); dst = torch.empty((1 << 30) + 16, dtype=torch.uint8, device="cuda")
for _ in range(300): uaes.ecb_dev(key, big, dst, nbytes=1 << 28, stream=st)
torch.cuda.synchronize()
print("%6s  %8s %8s %8s" % ("MiB", "ecb", "ecb-dec", "xts"))
for mib in (16, 18, 20, 24, 28, 32, 36, 40, 48, 52, 56, 64, 72, 80, 100, 128, 1024):
    n = mib << 20
    row = []
    for fn in (lambda: uaes.ecb_dev(key, big, dst, nbytes=n, stream=st), lambda: uaes.ecb_dev(key, big, dst, decrypt=True, nbytes=n, stream=st),
               lambda: uaes.xts_sectors_dev(keys, 0, 4096, n // 4096, big, dst, stream=st)):
        for _ in range(40): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        row.append(n / dt / 2**30)
    print("%6d  %8.0f %8.0f %8.0f" % (mib, *row), flush=True)
