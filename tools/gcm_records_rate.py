#!/usr/bin/env python3
"""Record streams under one GCM key: ONE record call (uaes_gcm_key_{en,de}crypt_records_dev) against a keyed call per
record, device-resident.  Per record size: records/s and GiB/s of text, both directions; the first and last record of
every run are checked against a call of their own."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_aes_amd as uaes

key = bytes(range(16))
gk = uaes.GcmKey(key)
st = torch.cuda.current_stream()
TOTAL = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 256 << 20
AAD = 13                                            # a TLS 1.2 record header; 5 for TLS 1.3
print("# %d MiB of text per call, %d bytes of AAD per record (shared), AES-128" % (TOTAL >> 20, AAD))
print("%8s %9s | %-30s | %-30s | %-26s" % ("rec B", "records", "one record call: encrypt", "decrypt", "a keyed call per record"))
for rec_len in (64, 256, 1440, 4096, 16384, 32704):
    stride = (rec_len + 16 + 15) // 16 * 16
    nrec = max(1, TOTAL // rec_len)
    nonces = torch.randint(0, 256, (nrec * 12,), dtype=torch.uint8, device="cuda")
    aad = torch.randint(0, 256, (16,), dtype=torch.uint8, device="cuda")
    src = torch.randint(0, 256, (nrec * stride,), dtype=torch.uint8, device="cuda")
    dst = torch.zeros(nrec * stride, dtype=torch.uint8, device="cuda")
    back = torch.zeros(nrec * stride, dtype=torch.uint8, device="cuda")
    ver = torch.zeros(nrec, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    enc = lambda: gk.encrypt_records_dev(nrec, nonces, aad, AAD, 0, src, rec_len, stride, dst, stride, stream=st)
    dec = lambda: gk.decrypt_records_dev(nrec, nonces, aad, AAD, 0, dst, rec_len, stride, back, stride, ver, status, stream=st)
    row = []
    for fn in (enc, dec):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row.append("%8.3f ms %7.2f M/s %7.1f GiB/s" % (dt * 1e3, nrec / dt / 1e6, nrec * rec_len / dt / 2**30))
    assert int(status.item()) == 0 and int(ver.max().item()) == 0
    nb = bytes(nonces.cpu().numpy())
    ab = bytes(aad.cpu().numpy())[:AAD]
    one = torch.zeros(rec_len + 16, dtype=torch.uint8, device="cuda")
    aad_t = aad[:AAD].clone()
    for r in (0, nrec - 1):
        gk.encrypt_dev(nb[12 * r: 12 * r + 12], aad_t, src[r * stride: r * stride + rec_len], rec_len, one, stream=st)
        torch.cuda.synchronize()
        assert torch.equal(one, dst[r * stride: r * stride + rec_len + 16]), (rec_len, r)
        assert torch.equal(back[r * stride: r * stride + rec_len], src[r * stride: r * stride + rec_len])
    calls = min(nrec, 2000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(calls):
        gk.encrypt_dev(nb[12 * r: 12 * r + 12], aad_t, src[r * stride: r * stride + rec_len], rec_len, one, stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    row.append("%7.3f M/s %8.2f GiB/s" % (1 / dt / 1e6, rec_len / dt / 2**30))
    print("%8d %9d | %s" % (rec_len, nrec, " | ".join(row)), flush=True)

# packet buffers: records of different lengths (uniform in 40..1500 bytes) in 1536-byte slots, one call
nrec, max_len, stride = 1 << 18, 1500, 1536
L = uaes.engine()
lens = torch.randint(40, max_len + 1, (nrec,), dtype=torch.int32, device="cuda")
nonces = torch.randint(0, 256, (nrec * 12,), dtype=torch.uint8, device="cuda")
aad = torch.randint(0, 256, (16,), dtype=torch.uint8, device="cuda")
src = torch.randint(0, 256, (nrec * stride,), dtype=torch.uint8, device="cuda")
dst = torch.zeros(nrec * stride, dtype=torch.uint8, device="cuda")
sp = st.cuda_stream
call = lambda: L.uaes_gcm_key_encrypt_records_v_dev(gk._h, nrec, nonces.data_ptr(), aad.data_ptr(), AAD, 0, src.data_ptr(),
                                                    lens.data_ptr(), max_len, stride, dst.data_ptr(), stride, sp)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
total = int(lens.sum().item())
print("mixed 40..1500-byte records in 1536-byte slots, %d records: %.3f ms  %.2f M records/s  %.1f GiB/s of text"
      % (nrec, dt * 1e3, nrec / dt / 1e6, total / dt / 2**30))
r = 12345
n = int(lens[r].item())
one = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
gk.encrypt_dev(bytes(nonces[12 * r: 12 * r + 12].cpu().numpy()), aad[:AAD].clone(), src[r * stride: r * stride + n], n, one, stream=st)
torch.cuda.synchronize()
assert torch.equal(one, dst[r * stride: r * stride + n + 16])
