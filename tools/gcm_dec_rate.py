"""GCM decrypt of a device-resident message: the default two-pass order (tag first, N7) against the
one-pass kernel that uaes_set_gcm_one_pass_decrypt(1) allows.  Interleaved rounds, hipEvent timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import micro_aes_amd as uaes  # noqa: E402


def main():
    L = uaes.engine()
    key, nonce = bytes(range(16)), bytes(range(0xF0, 0xFC))
    sizes = [int(a) << 20 for a in sys.argv[1:]] or [16 << 20, 64 << 20, 256 << 20, 1 << 30]
    for n in sizes:
        src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
        ct = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
        out = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        status = torch.full((1,), -1, dtype=torch.int32, device="cuda:0")
        uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct)
        k = uaes.GcmKey(key)
        res = {}
        for rnd in range(3):
            for name, sw, keyed in (("two-pass", 0, False), ("one-pass", 1, False), ("one-pass keyed", 1, True),
                                    ("encrypt", 0, False)):
                L.uaes_set_gcm_one_pass_decrypt(sw)
                reps = max(5, min(200, (8 << 30) // n))

                def call():
                    if name == "encrypt":
                        uaes.gcm_encrypt_dev(key, nonce, None, src, n, ct)
                    elif keyed:
                        k.decrypt_dev(nonce, None, ct, n, out, status)
                    else:
                        uaes.gcm_decrypt_dev(key, nonce, None, ct, n, out, status)
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                res.setdefault(name, []).append(ms)
                if name != "encrypt":
                    assert int(status.item()) == 0 and torch.equal(out, src)
        L.uaes_set_gcm_one_pass_decrypt(0)
        k.close()
        print("%5d MiB  " % (n >> 20) + "  ".join("%s %.4f ms %6.1f GiB/s" % (nm, min(v), n / 2**30 / (min(v) * 1e-3))
                                                   for nm, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
