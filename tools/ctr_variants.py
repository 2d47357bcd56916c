#!/usr/bin/env python3
"""A/B of libuaes_hip builds on AES-128-CTR, 1 GiB steps back to back, interleaved over several rounds on ONE box:
per library the time per step, GiB/s, the shader clock under the load (one-wave probe) and the cost in CYCLES
(clk per block per CU) -- at 128 workgroups, where the chip is not power-capped and time = cycles, and at the
product's 256.  Every library's output for a 64 MiB + 5 byte call is hashed and compared with the first library's
(the product build), and with the CPU oracle's for the first MiB.

    tools/build_variants.sh base= v2=-DUAES_ASM_VARIANT=2 ...
    gpurun -- 'python tools/ctr_variants.py base v2 ...'          # names = micro-aes_amd/lib/libuaes_hip_<name>.so
"""
import os, subprocess, sys
CHILD = r'''
import sys, ctypes as C, hashlib
sys.path.insert(0, %(root)r)
import torch, micro_aes_amd as uaes
uaes.lib_path.__defaults__ = (%(lib)r,)
L = uaes.engine()
key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
wl = %(wl)r
n = 1 << 30
torch.manual_seed(1)
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda"); dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream(); side = torch.cuda.Stream(); out = torch.zeros(2, dtype=torch.int64, device="cuda")
keys2 = bytes(range(64))
def step(m=n):
    if wl == "ctr": uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=m, stream=st)
    elif wl == "gcm": uaes.gcm_encrypt_dev(key, ctr0[:12], None, src, m, dst, stream=st)
    elif wl == "ecb": uaes.ecb_dev(key, src, dst, nbytes=m & ~15, stream=st)
    elif wl == "xts": uaes.xts_sectors_dev(keys2, 0, 4096, m // 4096, src, dst, stream=st)
    elif wl == "ctr256": uaes.ctr_xcrypt_dev(keys2[:32], ctr0, 0, src, dst, nbytes=m, stream=st)
m = (64 << 20) + 5
dst.zero_(); step(m); torch.cuda.synchronize()
digest = hashlib.sha256(dst[: m + 16].cpu().numpy().tobytes()).hexdigest()[:16]
import time
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(8): step()
    st.synchronize()
res = []
for i in range(2):
    L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(60): step()
    e1.record(st); torch.cuda.synchronize()
    c, t = out.tolist(); res.append((e0.elapsed_time(e1) / 60, c / (t / 100.0)))
ms = sum(r[0] for r in res) / len(res); mhz = sum(r[1] for r in res) / len(res)
g = int(%(grid)r)
print("%%-14s %%-6s %%4d wg: %%.4f ms %%7.1f GiB/s  sclk %%4.0f MHz  %%.3f clk/block/CU  sha %%s" %% (%(name)r, wl, g, ms, 1e3 / ms, mhz, g * mhz * 1e6 * ms * 1e-3 / 2**26, digest))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = ([a[5:] for a in sys.argv[1:] if a.startswith("--wl=")] or ["ctr"])[0]
rounds = int(([a[9:] for a in sys.argv[1:] if a.startswith("--rounds=")] or ["2"])[0])
grids = [int(x) for x in ([a[8:] for a in sys.argv[1:] if a.startswith("--grids=")] or ["128,256"])[0].split(",")]
first = {}
for rnd in range(rounds):
    for g in grids:
        for name in args:
            code = CHILD % dict(root=root, lib="libuaes_hip_%s.so" % name, grid=g, name=name, wl=wl)
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UAES_CTR_GRID=str(g)), capture_output=True, text=True)
            line = (r.stdout.strip().splitlines() or ["%s: FAILED %s" % (name, r.stderr[-400:])])[-1]
            sha = line.rsplit("sha ", 1)[-1] if "sha " in line else None
            first.setdefault(wl, sha)
            print(line + ("" if sha == first[wl] else "   <-- DIFFERS from %s" % args[0]), flush=True)
