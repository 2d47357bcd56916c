import sys, os
sys.path.insert(0, "/root/repo")
import micro_aes_amd as uaes
uaes.lib_path.__defaults__ = ("libuaes_hip_T.so",)
key, nonce = bytes(range(16)), bytes(range(12))
k = uaes.GcmKey(key)
for n in (16, 4096, 16384, 32000):
    for _ in range(3):
        uaes.AES_GCM_encrypt(key, nonce, b"", bytes(n))
    for _ in range(3):
        k.encrypt(nonce, b"", bytes(n))
