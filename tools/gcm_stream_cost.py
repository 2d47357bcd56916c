import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, micro_aes_amd as uaes
L = uaes.engine(); key = bytes(range(16)); nonce = bytes(12); tag = (C.c_uint8 * 16)()
src = torch.randint(0, 256, (256 << 20,), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
def one(n, pieces=1):
    h = C.c_void_p()
    t0 = time.perf_counter(); L.uaes_gcm_stream_begin(C.byref(h), 128, key, nonce, None, 0, 0); t1 = time.perf_counter()
    for i in range(pieces):
        if n: L.uaes_gcm_stream_update(h, C.c_void_p(src.data_ptr() + i * n), n, C.c_void_p(dst.data_ptr() + i * n))
    t2 = time.perf_counter(); L.uaes_gcm_stream_finish(h, tag); t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
for n, p in ((0, 1), (4096, 1), (1 << 20, 1), (16 << 20, 1), (64 << 20, 1), (256 << 20, 1), (16 << 20, 16)):
    one(n, p); r = [one(n, p) for _ in range(5)]
    print("piece %9d x %2d: begin %7.1f us  updates %8.1f us  finish %7.1f us" % (n, p, min(x[0] for x in r), min(x[1] for x in r), min(x[2] for x in r)))
