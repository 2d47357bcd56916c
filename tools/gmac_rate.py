import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
L = uaes.engine(); key = bytes(range(16)); n12 = bytes(12)
cap = 1 << 30
aad = torch.randint(0, 256, (cap,), dtype=torch.uint8, device="cuda"); dst = torch.empty(4096 + 64, dtype=torch.uint8, device="cuda")
src = torch.randint(0, 256, (4096,), dtype=torch.uint8, device="cuda")
a, b, ad = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(aad.data_ptr())
for alen in (1 << 20, 4 << 20, 32 << 20, 256 << 20, 1 << 30):
    for tl in (0, 4096):
        fn = lambda: L.uaes_gcm_encrypt(128, key, n12, ad, alen, a, tl, b)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("gmac aad %5d MiB + %4d B text: %8.1f us %8.1f GiB/s" % (alen >> 20, tl, dt * 1e6, alen / dt / 2**30), flush=True)
