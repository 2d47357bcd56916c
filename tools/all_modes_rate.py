#!/usr/bin/env python3
"""Bulk rate of EVERY mode through the synchronous C ABI with device pointers (GiB/s, one call = the whole text):
a sweep for holes -- shapes that run far below the rate their arithmetic allows.  The serial chains (CBC/CFB encrypt,
OFB, CMAC, CCM's MAC) are latency-bound by construction and get a short text.  [MiB] default 256."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
key16, key32, iv16, n12, n11 = bytes(range(16)), bytes(range(32)), bytes(range(16)), bytes(range(12)), bytes(range(11))
src = torch.randint(0, 256, (total + 64,), dtype=torch.uint8, device="cuda")
dst = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
aad_t = torch.randint(0, 256, (32 << 20,), dtype=torch.uint8, device="cuda")
a, b, ad = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(aad_t.data_ptr())
mac = (C.c_uint8 * 16)()


def rate(name, fn, nbytes, prep=None, reps=6):
    if prep:
        prep()
    for _ in range(2):
        rc = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-34s %9.2f ms %9.1f GiB/s  rc=%s" % (name, dt * 1e3, nbytes / dt / 2**30, rc), flush=True)


n = total
small = 1 << 20
for bits, key in ((128, key16), (256, key32)):
    print("# AES-%d, %d MiB" % (bits, n >> 20))
    rate("ecb enc", lambda: L.uaes_ecb_encrypt(bits, key, a, n, b), n)
    rate("ecb dec", lambda: L.uaes_ecb_decrypt(bits, key, a, n, b), n)
    rate("ctr", lambda: L.uaes_ctr_xcrypt(bits, key, iv16, a, n, b), n)
    rate("cbc dec", lambda: L.uaes_cbc_decrypt(bits, key, iv16, a, n, b), n)
    rate("cfb dec", lambda: L.uaes_cfb_decrypt(bits, key, iv16, a, n, b), n)
    rate("gcm enc", lambda: L.uaes_gcm_encrypt(bits, key, n12, None, 0, a, n, b), n)
    rate("gcm dec (two passes, N7)", lambda: L.uaes_gcm_decrypt(bits, key, n12, None, 0, b, n, a), n,
         prep=lambda: L.uaes_gcm_encrypt(bits, key, n12, None, 0, a, n, b))
    rate("gcm enc + 1 MiB AAD", lambda: L.uaes_gcm_encrypt(bits, key, n12, ad, 1 << 20, a, n, b), n)
    rate("gcm enc + 32 MiB AAD", lambda: L.uaes_gcm_encrypt(bits, key, n12, ad, 32 << 20, a, n, b), n + (32 << 20))
    rate("gcm AAD only (32 MiB, GMAC)", lambda: L.uaes_gcm_encrypt(bits, key, n12, ad, 32 << 20, a, 0, b), 32 << 20)
    rate("ocb enc", lambda: L.uaes_ocb_encrypt(bits, key, n12, None, 0, a, n, b), n)
    rate("ocb dec", lambda: L.uaes_ocb_decrypt(bits, key, n12, None, 0, b, n, a), n,
         prep=lambda: L.uaes_ocb_encrypt(bits, key, n12, None, 0, a, n, b))
    rate("ocb enc + 32 MiB AAD", lambda: L.uaes_ocb_encrypt(bits, key, n12, ad, 32 << 20, a, n, b), n + (32 << 20))
    rate("gcm-siv enc", lambda: L.uaes_gcmsiv_encrypt(bits, key, n12, None, 0, a, n, b), n)
    rate("gcm-siv dec", lambda: L.uaes_gcmsiv_decrypt(bits, key, n12, None, 0, b, n, a), n,
         prep=lambda: L.uaes_gcmsiv_encrypt(bits, key, n12, None, 0, a, n, b))
    rate("xts, one unit", lambda: L.uaes_xts_encrypt(bits, (key + key)[: bits // 4], iv16, a, n, b), n)
    rate("xts, one unit + 7 bytes (stealing)", lambda: L.uaes_xts_encrypt(bits, (key + key)[: bits // 4], iv16, a, n - 9, b), n)
    if bits == 128:
        rate("ccm enc (1 MiB: serial MAC)", lambda: L.uaes_ccm_encrypt(bits, key, n11, None, 0, a, small, b), small, reps=2)
        rate("cmac (1 MiB: serial)", lambda: L.uaes_cmac(bits, key, a, small, mac), small, reps=2)
        rate("cbc enc (1 MiB: serial)", lambda: L.uaes_cbc_encrypt(bits, key, iv16, a, small, b), small, reps=2)
        rate("ofb (1 MiB: serial)", lambda: L.uaes_ofb_xcrypt(bits, key, iv16, a, small, b), small, reps=2)
        nb = max(1, n // 65536)                              # (messages of 64 KiB: as many as the buffers hold)
        rate("cbc enc batch %d x 64 KiB" % nb, lambda: L.uaes_cbc_encrypt_batch(bits, key, ad, nb, 65536, a, b), nb * 65536)
        rate("cmac batch %d x 64 KiB" % nb, lambda: L.uaes_cmac_batch(bits, key, nb, 65536, a, b), nb * 65536)
