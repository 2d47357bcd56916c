#!/usr/bin/env python3
"""uaes_gcm_stream_*: one GCM message fed in pieces (device pointers): GiB/s by piece size, encrypt and decrypt."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
key, nonce = bytes(range(16)), bytes(12)
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
src = torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda")
dst = torch.empty(total, dtype=torch.uint8, device="cuda")
tag = (C.c_uint8 * 16)()
print("%10s %12s %12s" % ("piece", "enc GiB/s", "dec GiB/s"))
for piece in (64 << 10, 1 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20, total):
    row = []
    for dec in (0, 1):
        def run():
            h = C.c_void_p()
            assert L.uaes_gcm_stream_begin(C.byref(h), 128, key, nonce, None, 0, dec) == 0
            for off in range(0, total, piece):
                n = min(piece, total - off)
                assert L.uaes_gcm_stream_update(h, C.c_void_p(src.data_ptr() + off), n, C.c_void_p(dst.data_ptr() + off)) == 0
            return L.uaes_gcm_stream_finish(h, tag)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        row.append(total / (time.perf_counter() - t0) / 2**30)
    print("%10d %12.1f %12.1f" % (piece, row[0], row[1]), flush=True)
