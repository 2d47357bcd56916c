import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
L = uaes.engine(); key = bytes(range(16))
cap = 256 << 20
src = torch.randint(0, 256, (cap,), dtype=torch.uint8, device="cuda"); dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
ivs = torch.randint(0, 256, (16 << 20,), dtype=torch.uint8, device="cuda")
a, b, iv = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(ivs.data_ptr())
print("# UAES_BATCH_ROW_MAX=%s" % os.environ.get("UAES_BATCH_ROW_MAX"))
for msg, cnt in ((4096, 32768), (2048, 65536), (4096, 65536), (1024, 98304), (1024, 131072), (2048, 131072), (1024, 196608), (1024, 262144), (512, 262144), (256, 524288)):
    row = []
    for fn in (lambda: L.uaes_cbc_encrypt_batch(128, key, iv, cnt, msg, a, b), lambda: L.uaes_cmac_batch(128, key, cnt, msg, a, b)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); row.append(msg * cnt / ((time.perf_counter() - t0) / 3) / 2**30)
    print("%8d %8d %10.1f %10.1f" % (msg, cnt, row[0], row[1]), flush=True)
