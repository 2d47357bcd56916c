#!/bin/bash
# tools/one_launch_trace.sh -- run on the GPU box: rocprofv3 kernel traces of calls that became ONE launch in round 5
# (OCB, GCM 32 KiB .. 4 MiB, XTS of 4 KiB sectors up to 128 MiB, a streamed GCM piece): the kernels of the last calls,
# their durations and the gaps between them (tools/kernel_gaps.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05b/one_launch; mkdir -p $OUT
cat > $OUT/run.py <<'P'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import micro_aes_amd as uaes
what, n = sys.argv[1], int(sys.argv[2]) << 10
key, keys2, nonce = bytes(range(16)), bytes(range(64)), bytes(range(12))
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda:0")
dst = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
for _ in range(20):
    if what == "ocb": uaes.ocb_dev(key, nonce, None, src, n, dst)
    elif what == "gcm": uaes.gcm_encrypt_dev(key, nonce, None, src, n, dst)
    elif what == "xts": uaes.xts_sectors_dev(keys2, 7, 4096, n // 4096, src, dst)
    elif what == "xts512": uaes.xts_sectors_dev(keys2, 7, 512, n // 512, src, dst)
torch.cuda.synchronize()
P
for spec in "ocb 1024" "ocb 16384" "gcm 64" "gcm 1024" "gcm 16384" "xts 16384" "xts 1048576" "xts512 16384"; do
  set -- $spec
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$1_$2 -o kt -- python $OUT/run.py $1 $2 > $OUT/kt_$1_$2.log 2>&1
  echo "== $1, $2 KiB per call: kernel, duration us, gap to the previous kernel's end"
  python tools/kernel_gaps.py $OUT/kt_$1_$2 4
done
