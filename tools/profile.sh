#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun):
#   1. rocprofv3 --kernel-trace --stats   (per-kernel durations)
#   2. three separate --pmc passes        (SQ/LDS counters, FETCH_SIZE, WRITE_SIZE;
#      never combined with tracing domains other than --kernel-trace)
# Summaries land in gpurun_out/prof_<tag>/; copy what should be judged to profiles/.
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --no-cpu --no-verify --no-traffic --no-clock-probe --sustain-s 0 --no-c-gather --no-other-configs $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH --steps 20 --warmup 3 > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_SALU -d $OUT/pmc_sq2 -o pmc -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_wr -o pmc -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_wr.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
