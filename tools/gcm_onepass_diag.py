#!/usr/bin/env python3
"""Diagnostic: the one-pass GCM decrypt (k_gcm_fused<NR, true>) against the two-pass one, for several sizes and libraries:
where do the outputs differ?   python tools/gcm_onepass_diag.py <lib name> ..."""
import os, subprocess, sys
CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import torch, micro_aes_amd as uaes
uaes.lib_path.__defaults__ = (%(lib)r,)
L = uaes.engine()
key, nonce = bytes(range(32)), bytes(range(12))
for n in (31892083, 16 << 20, (16 << 20) + 4096, 100 << 20, 20000003):
    torch.manual_seed(n)
    src = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda:0")
    dst = torch.full((n + 32,), 0xA5, dtype=torch.uint8, device="cuda:0")
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
    res = []
    for one in (0, 1, 1):
        L.uaes_set_gcm_one_pass_decrypt(one)
        back = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda:0")
        status.fill_(-1)
        uaes.gcm_decrypt_dev(key, nonce, None, dst, n, back, status)
        torch.cuda.synchronize()
        bad = (back != src[:n]).nonzero().flatten()
        if bad.numel():
            b16 = torch.unique(bad // 16)
            gaps = (b16[1:] - b16[:-1])
            res.append("one_pass=%%d status %%d: %%d bad bytes in %%d blocks, first block %%d last %%d, distinct gaps %%s, sample bytes %%s" %% (
                one, int(status.item()), bad.numel(), b16.numel(), int(b16[0]), int(b16[-1]), torch.unique(gaps)[:6].tolist(),
                back[bad[:4]].tolist()))
        else:
            res.append("one_pass=%%d status %%d ok" %% (one, int(status.item())))
    L.uaes_set_gcm_one_pass_decrypt(0)
    print("%%-10s n=%%d: " %% (%(name)r, n) + " | ".join(res), flush=True)
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=root, lib="libuaes_hip_%s.so" % name, name=name)], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
