#!/bin/bash
# tools/r05_profiles.sh -- rocprofv3 summaries of the round's bench commands (tools/profile.sh: --kernel-trace --stats and the
# separate --pmc passes) + the GPU suite's log.   gpurun -- 'bash tools/r05_profiles.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05p
timeout 900 bash tools/profile.sh r05_ctr > /dev/null 2>&1
timeout 900 bash tools/profile.sh r05_gcm --workload gcm > /dev/null 2>&1
timeout 900 bash tools/profile.sh r05_xts --workload xts > /dev/null 2>&1
for t in ctr gcm xts; do
    cp gpurun_out/prof_r05_$t/summary.txt gpurun_out/r05p/r05_${t}_rocprof_summary.txt
    cp gpurun_out/prof_r05_$t/kt/*kernel_stats.csv gpurun_out/r05p/r05_${t}_kernel_stats.csv 2>/dev/null
    rm -rf gpurun_out/prof_r05_$t
done
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05p/r05_gpu_suite.log
tail -4 gpurun_out/r05p/r05_gpu_suite.log
grep -A3 "steady state" gpurun_out/r05p/r05_ctr_rocprof_summary.txt | head -8
