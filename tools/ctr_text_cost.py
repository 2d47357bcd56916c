#!/usr/bin/env python3
"""What do the text loads and stores cost the CTR kernel in CYCLES?  Timing-only builds of the library
(-DUAES_CTR_NOLOAD, -DUAES_CTR_NOSTORE, both: wrong results by construction) against the product build, with
UAES_CTR_GRID=128 -- at 128 CUs the chip is not power-capped (2.38 GHz), so time is cycles -- and at the full grid.
The switches are not in the product source: `git apply tools/experiments/ctr_measurement_switches.patch` first (and
`git apply -R` it afterwards).
    cd micro-aes_amd/csrc; for v in NOLOAD NOSTORE "NOLOAD -DUAES_CTR_NOSTORE"; do touch uaes_ctr.hip.h; make XFLAGS=-DUAES_CTR_$v; cp ../lib/libuaes_hip.so ../lib/libuaes_hip_<name>.so; done"""
import os, subprocess, sys
CHILD = r'''
import sys, ctypes as C
sys.path.insert(0, %r)
import torch, micro_aes_amd as uaes
uaes.lib_path.__defaults__ = (%r,)
L = uaes.engine()
key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
n = 1 << 30
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
st = torch.cuda.current_stream(); side = torch.cuda.Stream(); out = torch.zeros(2, dtype=torch.int64, device="cuda")
import time
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(8): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    st.synchronize()
res = []
for i in range(3):
    L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(60): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    e1.record(st); torch.cuda.synchronize()
    c, t = out.tolist(); res.append((e0.elapsed_time(e1) / 60, c / (t / 100.0)))
ms = sum(r[0] for r in res) / 3; mhz = sum(r[1] for r in res) / 3
g = int(%r)
print("%%-26s %%4d workgroups: %%.4f ms  %%7.1f GiB/s  sclk %%4.0f MHz  %%.2f clk per block per CU" %% (%r, g, ms, 1e3 / ms, mhz, g * mhz * 1e6 * ms * 1e-3 / 2**26))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for g in (128, 256):
    for lib in ("libuaes_hip_base.so", "libuaes_hip_NOLOAD.so", "libuaes_hip_NOSTORE.so", "libuaes_hip_NOLOADNOSTORE.so"):
        r = subprocess.run([sys.executable, "-c", CHILD % (root, lib, g, lib)], env=dict(os.environ, UAES_CTR_GRID=str(g)), capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
