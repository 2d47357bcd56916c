"""Diagnostic: phase stamps printed by the k_gcm_setup / k_gcm_fused kernels of a TIMING build of the library.

    cd micro-aes_amd/csrc && touch uaes_*.hip && make XFLAGS=-DUAES_GF_TIMING && cp ../lib/libuaes_hip.so ../lib/libuaes_hip_T.so
    touch uaes_*.hip && make                       # back to the product build
    gpurun -- 'python tools/ubench/gf_timing.py'
"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import micro_aes_amd as uaes
uaes.lib_path.__defaults__ = ("libuaes_hip_T.so",)
key, nonce = bytes(range(16)), bytes(12)
for mib in (9, 64, 1024):
    n = mib << 20
    src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    dst = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    print("== %d MiB" % mib, flush=True)
    for _ in range(3):
        uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
        torch.cuda.synchronize()
