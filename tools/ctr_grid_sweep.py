#!/usr/bin/env python3
"""How does AES-128-CTR throughput at the power cap depend on the number of CUs used?  UAES_CTR_GRID workgroups
(one per CU) encrypt 1 GiB steps back to back; the one-wave clock probe reads the shader clock meanwhile.
If throughput were set by clock x CUs x LDS rate, fewer CUs at a higher clock would lose in proportion;
what the sweep shows is power / energy-per-block."""
import os, subprocess, sys
CHILD = r'''
import sys, ctypes as C
sys.path.insert(0, %r)
import torch, micro_aes_amd as uaes
L = uaes.engine()
key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
n = 1 << 30
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
st = torch.cuda.current_stream(); side = torch.cuda.Stream(); out = torch.zeros(2, dtype=torch.int64, device="cuda")
import time
t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    for _ in range(8): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    st.synchronize()
res = []
for i in range(3):
    L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(80): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    e1.record(st); torch.cuda.synchronize()
    c, t = out.tolist(); res.append((e0.elapsed_time(e1) / 80, c / (t / 100.0)))
ms = sum(r[0] for r in res) / 3; mhz = sum(r[1] for r in res) / 3
g = int(%r)
print("%%4d workgroups: %%.4f ms per GiB  %%7.1f GiB/s  sclk %%4.0f MHz  %%.2f clk per block per CU used" %% (g, ms, 1e3 / ms, mhz, g * mhz * 1e6 * ms * 1e-3 / 2**26))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for g in (256, 240, 224, 192, 160, 128, 96, 64):
    r = subprocess.run([sys.executable, "-c", CHILD % (root, g)], env=dict(os.environ, UAES_CTR_GRID=str(g)), capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
