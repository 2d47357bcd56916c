"""Diagnostic: phase stamps printed by the k_ocb_small kernels of a TIMING build of the library.

    cd micro-aes_amd/csrc && touch uaes_*.hip && make XFLAGS=-DUAES_OCB_TIMING && cp ../lib/libuaes_hip.so ../lib/libuaes_hip_T.so
    touch uaes_*.hip && make                       # back to the product build
    gpurun -- 'python tools/ubench/ocb_timing.py'
"""
import sys, os
sys.path.insert(0, os.getcwd())
import micro_aes_amd as uaes
uaes.lib_path.__defaults__ = ("libuaes_hip_T.so",)
key, nonce = bytes(range(16)), bytes(range(12))
for n in (16, 4096, 4096):
    print("n", n, flush=True)
    ct = uaes.AES_OCB_encrypt(key, nonce, b"", bytes(n))
    uaes.AES_OCB_decrypt(key, nonce, b"", ct)
