import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import micro_aes_amd as uaes
L = uaes.engine()
key, iv, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC)), bytes(range(12))
def bench(fn, reps=100):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print("UAES_PIN_KIB=%s UAES_ZEROCOPY_MAX_KIB=%s" % (os.environ.get("UAES_PIN_KIB"), os.environ.get("UAES_ZEROCOPY_MAX_KIB")))
for n in (65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 4 << 20):
    src = np.random.default_rng(1).integers(0, 256, n + 16, dtype=np.uint8)
    dst = np.empty(n + 32, dtype=np.uint8)
    a, b = C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data)
    r = {"ecb": bench(lambda: L.uaes_ecb_encrypt(128, key, a, n, b)),
         "ctr": bench(lambda: L.uaes_ctr_xcrypt(128, key, iv, a, n, b)),
         "gcm": bench(lambda: L.uaes_gcm_encrypt(128, key, nonce, None, 0, a, n, b))}
    print("%8d B host: " % n + "  ".join("%s %7.1f us (%5.2f GiB/s)" % (k, v, n / v / 1073.74) for k, v in r.items()))
