import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import torch, micro_aes_amd as uaes
L = uaes.engine()
keys = bytes(range(64))
for mib in (16, 256, 1024):
    n = mib << 20
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda')
    dst = torch.empty_like(src)
    for _ in range(2):
        uaes.xts_sectors_dev(keys, 5, n, 1, src, dst)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        uaes.xts_sectors_dev(keys, 5, n, 1, src, dst)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("XTS-256 one data unit of %d MiB: %.3f ms, %.1f GiB/s" % (mib, dt * 1e3, n / dt / 2**30))
