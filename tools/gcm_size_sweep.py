#!/usr/bin/env python3
"""GCM across the three size regimes (single workgroup / chunk + combine kernels / striped one-pass kernel):
device-resident, back-to-back on one stream (us per call and GiB/s), encrypt and decrypt, against CTR."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

if os.environ.get("UAES_LIB"):                      # A/B against another build of the library
    uaes.lib_path.__defaults__ = (os.environ["UAES_LIB"],)
    print("# " + os.environ["UAES_LIB"])
key, nonce = bytes(range(16)), bytes(range(12))
ctr0 = nonce + b"\0\0\0\1"
st = torch.cuda.current_stream()
gk = uaes.GcmKey(key)
print("%9s  %-18s %-18s %-18s %-18s %-18s" % ("KiB", "ctr", "gcm encrypt", "gcm decrypt", "keyed encrypt", "keyed decrypt"))
SIZES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (4, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536)
for kib in SIZES:
    n = kib << 10
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st)
    row = []
    for fn in (lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, src, back, nbytes=n, stream=st),
               lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st),
               lambda: uaes.gcm_decrypt_dev(key, nonce, None, dst, n, back, status, stream=st),
               lambda: gk.encrypt_dev(nonce, None, src, n, dst, stream=st),
               lambda: gk.decrypt_dev(nonce, None, dst, n, back, status, stream=st)):
        reps = 300 if kib <= 8192 else 100
        for _ in range(reps // 2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row.append("%7.1f us %7.1f" % (dt * 1e6, n / dt / 2**30))
    assert int(status.item()) == 0 and torch.equal(back, src)
    print("%9d  %s" % (kib, "  ".join("%-18s" % r for r in row)), flush=True)
