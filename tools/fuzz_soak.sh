#!/bin/bash
# tools/fuzz_soak.sh <first seed> <seeds> -- run on the GPU box: the randomized differential tests
# (tests/test_gpu_fuzz.py: HIP library against the oracle) under seeds first .. first+seeds-1.  Odd seeds run with
# UAES_GCM_LOOK_TICKS=0 (the one-launch GCM fold then usually runs in a chunk workgroup), every fourth seed with
# UAES_GCM_FOLD=0 (the two-launch forms).
set -u
FIRST=${1:-1}; N=${2:-10}
mkdir -p gpurun_out
for ((s = FIRST; s < FIRST + N; ++s)); do
  extra=""
  (( s % 2 )) && extra="UAES_GCM_LOOK_TICKS=0"
  (( s % 4 == 0 )) && extra="UAES_GCM_FOLD=0"
  env $extra UAES_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/seed $s ($extra): /"
done | tee gpurun_out/fuzz_soak_$FIRST.log
