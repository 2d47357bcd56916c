#!/usr/bin/env python3
"""A/B two builds of libuaes_hip.so on the same box, interleaved (box-to-box variance is +-2 %, so
numbers from different gpurun calls must not be compared).

    cp micro-aes_amd/lib/libuaes_hip.so micro-aes_amd/lib/libuaes_hip_A.so    # build A
    ... change the source, make ...                                           # build B = libuaes_hip.so
    gpurun -- 'python tools/ab_libs.py libuaes_hip_A.so libuaes_hip.so gcm 1024'

Each library is loaded in its own child process (one HIP runtime binding per process)."""
import os
import subprocess
import sys

CHILD = r'''
import sys, time, ctypes as C
sys.path.insert(0, %(root)r)
import torch, micro_aes_amd as uaes
uaes.lib_path.__defaults__ = (%(lib)r,)
st = torch.cuda.current_stream()
key, nonce, keys2 = bytes(range(16)), bytes(12), bytes(range(64))
n = %(mib)d << 20
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
status = torch.zeros(1, dtype=torch.int32, device="cuda")
fns = {
    "ctr": lambda: uaes.ctr_xcrypt_dev(key, nonce + b"\0\0\0\1", 0, src, dst, nbytes=n, stream=st),
    "ecb": lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st),
    "ecb-dec": lambda: uaes.ecb_dev(key, src, dst, decrypt=True, nbytes=n, stream=st),
    "xts": lambda: uaes.xts_sectors_dev(keys2, 0, 4096, n // 4096, src, dst, stream=st),
    "gcm": lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st),
    "ocb": lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st),
    "gcm-dec": lambda: uaes.gcm_decrypt_dev(key, nonce, None, dst, n, src, status, stream=st),
    "ocb-dec": lambda: uaes.ocb_dev(key, nonce, None, dst, n, src, decrypt=True, status=status, stream=st),
    "cbc-dec": lambda: uaes.engine().uaes_cbc_decrypt(128, key, bytes(16), C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())),
    "cfb-dec": lambda: uaes.engine().uaes_cfb_decrypt(128, key, bytes(16), C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())),
}
if %(wl)r == "ocb-dec":
    uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st)
    torch.cuda.synchronize()
if %(wl)r == "gcm-dec":
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st)
    torch.cuda.synchronize()
fn = fns[%(wl)r]
reps = max(100, 300 * 64 // %(mib)d)
for _ in range(reps // 2): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): fn()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print("%%-28s %%-8s %%5d MiB  %%.4f ms  %%8.1f GiB/s" %% (%(lib)r, %(wl)r, %(mib)d, dt * 1e3, n / dt / 2**30))
'''

if __name__ == "__main__":
    # two libraries, or a comma-separated list in the first argument and "-" as the second
    a, b, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    mib = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = a.split(",") if b == "-" else [a, b]
    for lib in libs * 3:
        code = CHILD % dict(root=root, lib=lib, wl=wl, mib=mib)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print((out.stdout.strip().splitlines() or [out.stderr.strip()[-300:]])[-1])
