import os, sys, torch
sys.path.insert(0, os.getcwd())
import micro_aes_amd as uaes
key, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC))
n = int(sys.argv[1]) << 20
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
ct = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
for _ in range(30):
    uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct)
torch.cuda.synchronize()
