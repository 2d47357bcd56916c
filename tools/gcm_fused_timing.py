"""Phase stamps of k_gcm_fused (a library built with -DUAES_GF_TIMING as lib/libuaes_hip_T.so): where a mid-sized
one-shot GCM call spends its time.  usage: gcm_fused_timing.py <MiB>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import micro_aes_amd as uaes
uaes.lib_path.__defaults__ = ("libuaes_hip_T.so",)
key, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC))
n = int(sys.argv[1]) << 20
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
ct = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct)
    torch.cuda.synchronize()
    print("--", flush=True)
