import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
L = uaes.engine(); key = bytes(range(16)); n12 = bytes(12)
src = torch.randint(0, 256, (16 << 20,), dtype=torch.uint8, device="cuda"); dst = torch.empty((16 << 20) + 16, dtype=torch.uint8, device="cuda")
a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
hsrc = bytes(4096); hdst = (C.c_uint8 * 5000)()
print("UAES_GCM_KEY_CACHE=%s: us per synchronous uaes_gcm_encrypt under one key" % os.environ.get("UAES_GCM_KEY_CACHE", "1"))
for name, fn in (("16 B dev", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, 16, b)),
                 ("4 KiB host", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, hsrc, 4096, hdst)),
                 ("4 KiB dev", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, 4096, b)),
                 ("64 KiB dev", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, 65536, b)),
                 ("1 MiB dev", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, 1 << 20, b)),
                 ("16 MiB dev", lambda: L.uaes_gcm_encrypt(128, key, n12, None, 0, a, 16 << 20, b))):
    for _ in range(30): fn()
    t0 = time.perf_counter()
    for _ in range(300): fn()
    print("  %-12s %7.1f" % (name, (time.perf_counter() - t0) / 300 * 1e6), flush=True)
