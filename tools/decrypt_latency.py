import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, micro_aes_amd as uaes
L = uaes.engine()
key, nonce = bytes(range(16)), bytes(range(12))
for n in (16, 4096, 65536):
    pt = bytes(n)
    row = []
    for name, enc, dec in (("gcm", uaes.AES_GCM_encrypt, L.uaes_gcm_decrypt), ("ocb", uaes.AES_OCB_encrypt, L.uaes_ocb_decrypt),
                           ("ccm", uaes.AES_CCM_encrypt, L.uaes_ccm_decrypt)):
        nn = nonce if name != "ccm" else nonce[:11]
        ct = enc(key, nn, b"", pt)
        out = (C.c_uint8 * (n + 16))()
        for _ in range(200): dec(128, key, nn, None, 0, ct, n, out)
        t0 = time.perf_counter()
        for _ in range(2000): dec(128, key, nn, None, 0, ct, n, out)
        row.append("%s-dec %6.1f" % (name, (time.perf_counter() - t0) / 2000 * 1e6))
    print("%8d B host pointers, us per call: %s" % (n, "  ".join(row)))
