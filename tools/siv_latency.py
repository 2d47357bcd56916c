import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
L = uaes.engine(); key = bytes(range(16)); n12 = bytes(12)
src = torch.randint(0, 256, (64 << 20,), dtype=torch.uint8, device="cuda"); dst = torch.empty((64 << 20) + 16, dtype=torch.uint8, device="cuda")
a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
for n in (16, 4096, 32768, 65536, 1 << 20, 16 << 20, 64 << 20):
    row = []
    for fn in (lambda: L.uaes_gcmsiv_encrypt(128, key, n12, None, 0, a, n, b), lambda: L.uaes_gcmsiv_decrypt(128, key, n12, None, 0, b, n, a),
               lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, n, b), lambda: L.uaes_ccm_encrypt(128, key, bytes(11), None, 0, a, min(n, 65536), b)):
        for _ in range(5): fn()
        t0 = time.perf_counter()
        for _ in range(30): fn()
        row.append((time.perf_counter() - t0) / 30 * 1e6)
    print("%9d B: siv enc %8.1f us  siv dec %8.1f us   gcm enc %8.1f us   ccm(<=64K) %8.1f us" % (n, *row), flush=True)
