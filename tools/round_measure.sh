#!/bin/bash
# tools/round_measure.sh <tag> -- on the GPU box (via gpurun): every number the round's documents quote, from ONE box:
# the bench line of every workload (cpu_baseline included), per-call latencies, serial chains, and the rocprofv3
# summaries (kernel trace + the PMC passes) of CTR and GCM.  Results under gpurun_out/<tag>/; copy to profiles/.
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout 900 "$@" > $OUT/$name 2> $OUT/$name.err || echo "FAILED $name" >> $OUT/failed.txt; }
run bench_ctr.json   python bench.py
run bench_ecb.json   python bench.py --workload ecb --no-traffic
run bench_xts.json   python bench.py --workload xts --no-traffic
run bench_xts_c3.json python bench.py --workload xts --bytes 4294967296 --no-cpu
run bench_gcm.json   python bench.py --workload gcm
run bench_ocb.json   python bench.py --workload ocb --no-traffic
run bench_cbc_enc.json python bench.py --workload cbc-enc --bytes 4194304 --steps 3 --warmup 1 --settle-ms 0 --sustain-s 0 --no-traffic --no-clock-probe
run bench_cmac.json  python bench.py --workload cmac --bytes 4194304 --steps 3 --warmup 1 --settle-ms 0 --sustain-s 0 --no-traffic --no-clock-probe
run call_latency.log python tools/call_latency.py
run all_modes_rate.log python tools/all_modes_rate.py
run gcm_size_sweep.log python tools/gcm_size_sweep.py
run plan_table.txt   python tools/plan_table.py
run bench_one_rank_force_collective.json python bench.py --gpus 1 --force-collective --no-other-configs
bash tools/profile.sh ${TAG}_ctr > $OUT/profile_ctr.txt 2>&1
bash tools/profile.sh ${TAG}_gcm --workload gcm > $OUT/profile_gcm.txt 2>&1
bash tools/profile.sh ${TAG}_xts --workload xts > $OUT/profile_xts.txt 2>&1
ls -la $OUT
cat $OUT/failed.txt 2>/dev/null
