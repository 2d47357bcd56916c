#!/bin/bash
# tools/mid_call_trace.sh -- run on the GPU box: rocprofv3 kernel traces of mid-sized GCM calls (device pointers, enqueued
# back to back): which kernels a call of 8 / 16 / 64 MiB is made of after round 5's last session -- the chunk workgroups
# with CTR and GHASH together, the two phases past 16 MiB, the tag-first decryption -- with their durations and the gaps
# between them (tools/kernel_gaps.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mid_call_trace; mkdir -p $OUT
cat > $OUT/run.py <<'P'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import micro_aes_amd as uaes
what, n = sys.argv[1], int(sys.argv[2]) << 10
key, nonce = bytes(range(16)), bytes(range(12))
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
dst = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
back = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
status = torch.zeros(1, dtype=torch.int32, device="cuda:0")
uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
for _ in range(12):
    if what == "enc": uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
    else: uaes.gcm_decrypt_dev(key, nonce, None, dst, n, back, status)
torch.cuda.synchronize()
assert what == "enc" or (int(status.item()) == 0 and torch.equal(back, src))
P
for spec in "enc 8192" "enc 16384" "enc 65536" "dec 16384" "dec 65536"; do
  set -- $spec
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$1_$2 -o kt -- python $OUT/run.py $1 $2 > $OUT/kt_$1_$2.log 2>&1
  echo "== gcm $1, $2 KiB per call: kernel, duration us, gap to the previous kernel's end"
  python tools/kernel_gaps.py $OUT/kt_$1_$2 4
done
