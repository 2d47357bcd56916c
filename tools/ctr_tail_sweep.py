import os, subprocess, sys
CH = r'''
import sys, time
sys.path.insert(0, "/root/repo")
import torch, micro_aes_amd as uaes
key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
st = torch.cuda.current_stream()
big = torch.randint(0, 256, (1 << 28,), dtype=torch.uint8, device="cuda"); dst = torch.empty((1 << 28) + 16, dtype=torch.uint8, device="cuda")
for _ in range(400): uaes.ctr_xcrypt_dev(key, ctr0, 0, big, dst, nbytes=1 << 28, stream=st)
torch.cuda.synchronize()
row = []
for mib in (6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24):
    n = mib << 20
    reps = 400
    for _ in range(50): uaes.ctr_xcrypt_dev(key, ctr0, 0, big, dst, nbytes=n, stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): uaes.ctr_xcrypt_dev(key, ctr0, 0, big, dst, nbytes=n, stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    row.append("%d:%.0f" % (mib, n / dt / 2**30))
print(" ".join(row))
'''
for rnd in range(2):
    for hr in (3, 2):
        r = subprocess.run([sys.executable, "-c", CH], env=dict(os.environ, UAES_CTR_MIN_HALF_ROUNDS=str(hr)), capture_output=True, text=True)
        print("min half rounds %d  " % hr + (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
