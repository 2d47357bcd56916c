#!/bin/bash
# tools/fuzz_soak.sh <first seed> <seeds> -- run on the GPU box: the randomized differential tests
# (tests/test_gpu_fuzz.py: HIP library against the oracle) under seeds first .. first+seeds-1.
set -u
FIRST=${1:-1}; N=${2:-10}
mkdir -p gpurun_out/r05b
for ((s = FIRST; s < FIRST + N; ++s)); do
  UAES_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/seed $s: /"
done | tee gpurun_out/r05b/fuzz_soak_$FIRST.log
