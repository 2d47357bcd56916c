#!/usr/bin/env python3
"""What does each arrangement of the table (csrc/uaes_plan.h) buy?  For every arrangement that can be switched off
(uaes_debug_plan_disable), at sizes across its range: us per call with the table as it is against the table without
that arrangement (the next one takes the call), interleaved in ONE process on ONE box, device pointers, calls back to
back on one stream.  An arrangement whose gain stays inside the box-to-box spread (+-3 %) at every size it covers is a
candidate for deletion.

    gpurun -- 'python tools/plan_ab.py > gpurun_out/plan_ab.log'
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

L = uaes.engine()
st = torch.cuda.current_stream()
key, keys, nonce = bytes(range(16)), bytes(range(64)), bytes(range(12))
ctr0 = nonce + b"\0\0\0\1"
KIB, MIB = 1 << 10, 1 << 20


def timed(fn, n):
    reps = max(20, min(2000, int(60e6 / max(n, 4096))))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


def call_for(mode, direction, n, src, dst, status):
    if mode == "ecb":
        return lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st)
    if mode == "ctr":
        return lambda: uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    if mode == "gcm":
        if direction == 0:
            return lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st)
        return lambda: uaes.gcm_decrypt_dev(key, nonce, None, dst, n, src, status, stream=st)
    if mode == "ocb":
        return lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st)
    raise KeyError(mode)


ONLY = set(sys.argv[1:])                                     # e.g. `plan_ab.py xts.sectors gcm.twophase`


def row(arr, mode, direction, n, extra=None):
    if ONLY and arr not in ONLY:
        return None
    src = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    if mode == "xts":
        sector, ns = extra
        fn = lambda: uaes.xts_sectors_dev(keys, 0, sector, ns, src, dst, stream=st)          # noqa: E731
        plan_args = ("xts", sector, ns, 0)
    elif mode == "siv":
        import ctypes as C
        sp, dp = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
        fn = lambda: L.uaes_gcmsiv_encrypt(128, key, nonce, None, 0, sp, n, dp)                # noqa: E731
        plan_args = ("siv", n, 0, 0)
    else:
        if mode == "gcm" and direction:
            uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st)
        fn = call_for(mode, direction, n, src, dst, status)
        plan_args = (mode, n, 0, direction)
    L.uaes_debug_plan_disable(0)
    with_it = uaes.plan(*plan_args)[0]
    if with_it != arr:
        return None
    L.uaes_debug_plan_disable(1 << uaes.arrangement_id(arr))
    without = uaes.plan(*plan_args)[0]
    res = []
    for _ in range(2):                                       # interleaved: on, off, on, off
        L.uaes_debug_plan_disable(0)
        a = timed(fn, n)
        L.uaes_debug_plan_disable(1 << uaes.arrangement_id(arr))
        b = timed(fn, n)
        res.append((a, b))
    L.uaes_debug_plan_disable(0)
    a, b = min(r[0] for r in res), min(r[1] for r in res)
    what = ("%d x %d B" % (extra[1], extra[0])) if mode == "xts" else ("%d KiB" % (n >> 10))
    print("%-13s %-4s dir %d %14s   %9.1f us   without (%-12s) %9.1f us   gain %+6.1f %%"
          % (arr, mode, direction, what, a, without, b, (b / a - 1) * 100), flush=True)
    del src, dst


print("# tools/plan_ab.py: us per call with the table as it is / with ONE arrangement switched off (the next takes the call)")
for n in (4 * KIB, 64 * KIB, MIB, 4 * MIB, 8 * MIB - 16):
    row("ecb.single", "ecb", 0, n)
for n in (4 * KIB, 64 * KIB, MIB, 4 * MIB, 8 * MIB - 16):
    row("ctr.single", "ctr", 0, n)
for n in (9 * MIB, 12 * MIB, 16 * MIB, 20 * MIB, 64 * MIB, 1 << 30):
    row("ctr.striped", "ctr", 0, n)
for sector, ns in ((65536, 1), (MIB, 1), (4 * MIB, 1), (8 * MIB, 1), (512, 64), (4096, 64), (4096, 1024)):
    row("xts.small", "xts", 0, sector * ns, (sector, ns))
for sector, ns in ((512, 4097), (512, 65536), (512, 1 << 20), (1024, 1 << 18)):
    row("xts.packed", "xts", 0, sector * ns, (sector, ns))
for d in (0, 1):
    for n in (KIB, 4 * KIB, 16 * KIB, 31 * KIB):
        row("gcm.small", "gcm", d, n)
    for n in (64 * KIB, MIB, 4 * MIB, 8 * MIB, 16 * MIB) + ((32 * MIB, 64 * MIB, 128 * MIB, 256 * MIB, 512 * MIB) if d else ()):
        row("gcm.chunks", "gcm", d, n)
for n in (17 * MIB, 24 * MIB, 32 * MIB, 64 * MIB, 96 * MIB, 128 * MIB):
    row("gcm.twophase", "gcm", 0, n)
for n in (129 * MIB, 256 * MIB, 1 << 30):
    row("gcm.striped", "gcm", 0, n)
for n in (KIB, 4 * KIB, 16 * KIB):
    row("ocb.small", "ocb", 0, n)
for n in (KIB, 16 * KIB, 31 * KIB):
    row("siv.small", "siv", 0, n)
for n in (64 * KIB, MIB, 16 * MIB, 128 * MIB):
    row("siv.chunks", "siv", 0, n)
