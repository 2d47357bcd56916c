#!/usr/bin/env python3
"""Per-call latency of the f-row modes through the host-pointer C ABI (device pointers), us per call."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
key, iv16, n12, n11 = bytes(range(16)), bytes(range(16)), bytes(range(12)), bytes(range(11))


def bench(fn, reps=500):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (16, 4096):
    src = torch.randint(0, 256, (n + 32,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
    mac = (C.c_uint8 * 16)()
    r = {
        "cbc-enc": bench(lambda: L.uaes_cbc_encrypt(128, key, iv16, a, n, b)),
        "cbc-dec": bench(lambda: L.uaes_cbc_decrypt(128, key, iv16, a, n, b)),
        "cfb-enc": bench(lambda: L.uaes_cfb_encrypt(128, key, iv16, a, n, b)),
        "cfb-dec": bench(lambda: L.uaes_cfb_decrypt(128, key, iv16, a, n, b)),
        "ofb": bench(lambda: L.uaes_ofb_xcrypt(128, key, iv16, a, n, b)),
        "cmac": bench(lambda: L.uaes_cmac(128, key, a, n, mac)),
        "ccm": bench(lambda: L.uaes_ccm_encrypt(128, key, n11, None, 0, a, n, b)),
        "gcm-siv": bench(lambda: L.uaes_gcmsiv_encrypt(128, key, n12, None, 0, a, n, b)),
    }
    print("%6d B, us per call: " % n + "  ".join("%s %6.1f" % kv for kv in r.items()), flush=True)
