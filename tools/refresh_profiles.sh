#!/bin/bash
# tools/refresh_profiles.sh <tag> -- run on the GPU box: the round's bench lines, rocprofv3 summaries and sweeps
# into gpurun_out/<tag>/ (copy what should be judged to profiles/).
set -u
TAG=${1:-r03}
O=gpurun_out/$TAG
mkdir -p $O
for w in ctr ecb xts gcm; do timeout 600 python bench.py --workload $w --no-cpu 2>/dev/null | tail -1 > $O/bench_$w.json; done
timeout 600 python bench.py --workload xts --bytes $((4<<30)) --no-cpu 2>/dev/null | tail -1 > $O/bench_xts_c3.json
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_ctr_full.json
timeout 900 bash tools/profile.sh ${TAG}_ctr > /dev/null 2>&1
timeout 900 bash tools/profile.sh ${TAG}_gcm --workload gcm > /dev/null 2>&1
timeout 900 bash tools/profile.sh ${TAG}_xts --workload xts > /dev/null 2>&1
for t in ctr gcm xts; do cp gpurun_out/prof_${TAG}_$t/summary.txt $O/${t}_rocprof_summary.txt; cp gpurun_out/prof_${TAG}_$t/kt/*kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; done
timeout 600 python tools/keysize_rates.py > $O/keysize_rates.log 2>&1
timeout 600 python tools/size_sweep.py > $O/size_sweep.log 2>&1
timeout 600 python tools/serial_rate.py > $O/serial_rate.log 2>&1
timeout 600 python tools/call_latency.py > $O/call_latency.log 2>&1
timeout 600 python tools/threads_rate.py > $O/threads_rate.log 2>&1
timeout 600 python tools/host_path_rate.py --sizes 1,2,4,8,12,16,24,31,32,48,64,256,1024 --per-rep > $O/host_path_rate.log 2>&1
timeout 600 python tools/host_path_rate.py --sizes 8,12,16,24,31 --order ecb,xts4k,ctr --per-rep >> $O/host_path_rate.log 2>&1
timeout 300 python tools/gcm_size_sweep.py > $O/gcm_size_sweep.log 2>&1
timeout 300 python tools/gcm_records_rate.py > $O/gcm_records_rate.log 2>&1
timeout 300 python tools/xts_unit_sweep.py > $O/xts_unit_sweep.log 2>&1
timeout 300 python tools/call_latency_other.py > $O/call_latency_other.log 2>&1
timeout 300 python tools/decrypt_latency.py > $O/decrypt_latency.log 2>&1
cat $O/bench_*.json | cut -c1-260
