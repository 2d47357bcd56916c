#!/bin/bash
# A/B sweep of the CTR kernel variants (debug knob UAES_CTR_VARIANT), interleaved rounds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  echo -n "$v round$round: "
  UAES_CTR_VARIANT=$v python bench.py --steps 20 --warmup 3 --no-cpu --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'GiB/s  kernel_ms', d['roofline']['kernel_ms'], 'min', d['roofline']['kernel_ms_min'], 'frac', d['roofline']['frac'])"
done; done | tee -a gpurun_out/sweep_ctr.log
