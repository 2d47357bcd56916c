#!/bin/bash
# tools/siv_trace.sh -- run on the GPU box: rocprofv3 kernel trace of long GCM-SIV calls (device pointers, synchronous):
# the kernels of the last calls, their durations and the gaps between them (tools/kernel_gaps.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/siv_trace; mkdir -p $OUT
cat > $OUT/run.py <<'P'
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
import micro_aes_amd as uaes
L = uaes.engine()
n = int(sys.argv[1]) << 10
key, nonce = bytes(range(16)), bytes(range(12))
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
dst = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
a, b = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
for _ in range(10):
    L.uaes_gcmsiv_encrypt(128, key, nonce, None, 0, a, n, b)
for _ in range(10):
    L.uaes_gcmsiv_decrypt(128, key, nonce, None, 0, b, n, a)
torch.cuda.synchronize()
P
for kib in 4096 16384; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$kib -o kt -- python $OUT/run.py $kib > $OUT/kt_$kib.log 2>&1
  echo "== gcm-siv, $kib KiB per call: kernel, duration us, gap to the previous kernel's end (the last decrypt calls)"
  python tools/kernel_gaps.py $OUT/kt_$kib 16
done
