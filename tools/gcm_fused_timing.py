"""Phase stamps of k_gcm_setup / k_gcm_fused: where a mid-sized one-shot GCM call spends its time.
Needs a timing build of the library next to the product one (its kernels printf their stamps, x10 ns):

    make -C micro-aes_amd/csrc XFLAGS=-DUAES_GF_TIMING OBJ=$PWD/micro-aes_amd/csrc/build_T OUT=$PWD/micro-aes_amd/lib_T \
         $PWD/micro-aes_amd/lib_T/libuaes_hip.so
    cp micro-aes_amd/lib_T/libuaes_hip.so micro-aes_amd/lib/libuaes_hip_T.so
    gpurun -- 'python tools/gcm_fused_timing.py 16'        # MiB; results: profiles/r03_gcm_setup_split.log
"""
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
