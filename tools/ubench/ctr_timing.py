"""Diagnostic: per-wave loop time and time spent at the U-buffer barrier in k_ctr_shared2
(library built with -DUAES_CTR_TIMING as lib/libuaes_hip_T.so):

    cd micro-aes_amd/csrc && touch uaes_*.hip && make XFLAGS=-DUAES_CTR_TIMING && cp ../lib/libuaes_hip.so ../lib/libuaes_hip_T.so
    touch uaes_*.hip && make                       # back to the product build
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import micro_aes_amd as uaes

uaes.lib_path.__defaults__ = ("libuaes_hip_T.so",)
key, nonce = bytes(range(16)), bytes(12)
n = 1 << 30
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(3):
    print("== call", flush=True)
    uaes.ctr_xcrypt_dev(key, nonce + b"\0\0\0\1", 0, src, dst, nbytes=n)
    torch.cuda.synchronize()
