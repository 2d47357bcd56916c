#!/usr/bin/env python3
"""us per SYNCHRONOUS call (device pointers) by mode and text size: run it under UAES_TICKET_RIDE_MAX_KIB=0 (the
completion ticket never rides: k_ticket behind every call's kernel) and =1073741824 (it rides at every size) to see
where a riding ticket's per-workgroup system-scope release starts to cost more than the second launch."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
key, nonce, keys2 = bytes(range(16)), bytes(range(12)), bytes(range(32))
ctr0 = nonce + b"\0\0\0\1"
sizes_kib = [int(x) for x in sys.argv[1:]] or [4, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536]
top = max(sizes_kib) << 10
src = torch.randint(0, 256, (top + 64,), dtype=torch.uint8, device="cuda")
dst = torch.empty(top + 64, dtype=torch.uint8, device="cuda")
a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())


def loop(fn, reps):
    for _ in range(max(10, reps // 3)):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


print("# UAES_TICKET_RIDE_MAX_KIB=%s" % os.environ.get("UAES_TICKET_RIDE_MAX_KIB", "(default)"))
print("%8s %8s %8s %8s %8s %8s %8s" % ("KiB", "ecb", "ecb dec", "ctr", "xts", "gcm", "ocb"))
for kib in sizes_kib:
    n = kib << 10
    reps = max(30, min(400, (1 << 22) // kib))
    row = [loop(f, reps) for f in (
        lambda: L.uaes_ecb_encrypt(128, key, a, n, b),
        lambda: L.uaes_ecb_decrypt(128, key, a, n, b),
        lambda: L.uaes_ctr_xcrypt(128, key, ctr0, a, n, b),
        lambda: L.uaes_xts_encrypt(128, keys2, ctr0, a, n, b),
        lambda: L.uaes_gcm_encrypt(128, key, nonce, None, 0, a, n, b),
        lambda: L.uaes_ocb_encrypt(128, key, nonce, None, 0, a, n, b))]
    print("%8d " % kib + " ".join("%8.1f" % v for v in row), flush=True)
