import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
keys = bytes(range(64))
print("%9s %8s %10s" % ("total", "unit", "us/call"))
for total in (65536, 1 << 20, 4 << 20, 8 << 20, 16 << 20, 64 << 20):
    src = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    for unit in (512, 4096):
        fn = lambda: uaes.xts_sectors_dev(keys, 5, unit, total // unit, src, dst)
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize()
        print("%9d %8d %10.1f" % (total, unit, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
