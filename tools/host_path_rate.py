#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer API: the drop-in
AES_CTR_encrypt signature hands over HOST buffers.  Long texts are cut into slices that
several worker threads move through the GPU concurrently (UAES_PIPE_WORKERS, default 4;
1 = the plain path: one hipMemcpy in, the kernel, one hipMemcpy out).  Every result is
checked against the digest of the device-resident path."""
import argparse
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import micro_aes_amd as uaes

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1,16,64,256,1024", help="MiB, comma separated")
ap.add_argument("--order", default="ctr,ecb,xts4k", help="the order in which the modes are timed at every size")
ap.add_argument("--per-rep", action="store_true", help="print every repetition's time (ms)")
args = ap.parse_args()
L = uaes.engine()
key, iv, keys2 = bytes(range(16)), bytes(range(0xF0, 0xFC)), bytes(range(64))
print("UAES_PIPE_WORKERS=%s UAES_PIPE_SLICE_MIB=%s" % tuple(
    os.environ.get(k, "(default)") for k in ("UAES_PIPE_WORKERS", "UAES_PIPE_SLICE_MIB")))
for mib in [int(x) for x in args.sizes.split(",")]:
    n = mib << 20
    src = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    dst = np.empty(n + 16, dtype=np.uint8)
    a, b = C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data)
    dsrc = torch.from_numpy(src).cuda()
    ddst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    row = []
    modes = {
        "ctr": (lambda: L.uaes_ctr_xcrypt(128, key, iv, a, n, b),
                lambda: uaes.ctr_xcrypt_dev(key, iv + b"\0\0\0\1", 0, dsrc, ddst, nbytes=n)),
        "ecb": (lambda: L.uaes_ecb_encrypt(128, key, a, n, b), lambda: uaes.ecb_dev(key, dsrc, ddst, nbytes=n)),
        "xts4k": (lambda: L.uaes_xts_sectors(256, keys2, 77, 4096, n // 4096, a, b, 1),
                  lambda: uaes.xts_sectors_dev(keys2, 77, 4096, n // 4096, dsrc, ddst)),
        # GCM encrypt (round 5: slices as shards of the message, shares XORed on the host); the digest covers the text,
        # the tag is compared separately below
        "gcm": (lambda: L.uaes_gcm_encrypt(128, key, iv, None, 0, a, n, b),
                lambda: uaes.gcm_encrypt_dev(key, iv, None, dsrc, n, ddst))}
    for name in args.order.split(","):
        host, dev = modes[name]
        assert host() == 0
        dev()
        torch.cuda.synchronize()
        m = n + 16 if name == "gcm" else n
        ok = hashlib.sha256(dst[:m].tobytes()).digest() == hashlib.sha256(ddst[:m].cpu().numpy().tobytes()).digest()
        reps = 5 if mib < 1024 else 3
        each = []
        for _ in range(reps):
            t0 = time.perf_counter()
            host()
            each.append(time.perf_counter() - t0)
        dt = sum(each) / reps
        row.append("%s %7.2f GiB/s%s" % (name, n / dt / 2**30, "" if ok else " MISMATCH"))
        if args.per_rep:
            row[-1] += " [" + " ".join("%.2f" % (t * 1e3) for t in each) + " ms]"
    print("host->device->host %5d MiB: %s" % (mib, "   ".join(row)))
