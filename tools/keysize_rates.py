import sys, time
sys.path.insert(0, '/root/repo')
import torch, micro_aes_amd as uaes
st = torch.cuda.current_stream()
n = 1 << 30
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda'); dst = torch.empty(n + 16, dtype=torch.uint8, device='cuda')
status = torch.zeros(1, dtype=torch.int32, device='cuda')
for bits in (128, 192, 256):
    key = bytes(range(bits // 8)); nonce = bytes(12)
    for name, fn in (("ctr", lambda: uaes.ctr_xcrypt_dev(key, nonce + b"\0\0\0\1", 0, src, dst, nbytes=n, stream=st)),
                     ("ecb", lambda: uaes.ecb_dev(key, src, dst, nbytes=n, stream=st)),
                     ("ecb-dec", lambda: uaes.ecb_dev(key, src, dst, decrypt=True, nbytes=n, stream=st)),
                     ("gcm", lambda: uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst, stream=st)),
                     ("ocb", lambda: uaes.ocb_dev(key, nonce, None, src, n, dst, stream=st))):
        for _ in range(60): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print("AES-%d %-8s %.4f ms %7.1f GiB/s" % (bits, name, dt * 1e3, n / dt / 2**30))
