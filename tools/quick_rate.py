#!/usr/bin/env python3
"""Warm-clock device-resident rate of one workload at 1 GiB: quick A/B helper.
usage: quick_rate.py ecb|ecb-dec|ctr|xts|xts-dec|gcm|ocb|ocb-dec|cbc-dec [MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import micro_aes_amd as uaes

key, nonce, keys2 = bytes(range(16)), bytes(range(12)), bytes(range(64))
ctr0 = nonce + b"\0\0\0\1"
st = torch.cuda.current_stream()
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = mib << 20
src = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")
dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
status = torch.zeros(1, dtype=torch.int32, device="cuda")
L = uaes.engine()
iv16 = bytes(range(16))
fns = {
    "ctr": lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st),
    "ecb": lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st),
    "ecb-dec": lambda: uaes.ecb_dev(key, src, dst, decrypt=True, nbytes=n, stream=st),
    "xts": lambda: uaes.xts_sectors_dev(keys2, 0, 4096, n // 4096, src, dst, stream=st),
    "xts-dec": lambda: uaes.xts_sectors_dev(keys2, 0, 4096, n // 4096, src, dst, encrypt=False, stream=st),
    "gcm": lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st),
    "ocb": lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st),
    "ocb-dec": lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, decrypt=True, status=status, stream=st),
    "cbc-dec": lambda: L.uaes_cbc_decrypt(128, key, iv16, C.c_void_p(src.data_ptr()), n, C.c_void_p(dst.data_ptr())),
}
for w in sys.argv[1].split(","):
    fn = fns[w]
    reps = max(100, 200 * 64 // mib)
    for _ in range(reps // 2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-8s %5d MiB  %8.4f ms  %8.1f GiB/s" % (w, mib, dt * 1e3, n / dt / 2**30))
