#!/bin/bash
# tools/r05_measure.sh -- the round's bench lines (with the CPU leg: the compiled reference on this host), then the tools
# that read them.  gpurun -- 'bash tools/r05_measure.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
for wl in ctr ecb xts gcm ocb; do
    python bench.py --workload $wl > gpurun_out/r05/bench_$wl.log 2>&1
    grep '^{' gpurun_out/r05/bench_$wl.log > profiles/r05_bench_$wl.json
done
python bench.py --workload xts --bytes 4294967296 --no-cpu > gpurun_out/r05/bench_xts_c3.log 2>&1
grep '^{' gpurun_out/r05/bench_xts_c3.log > profiles/r05_bench_xts_c3.json
for wl in cbc-enc cmac; do
    python bench.py --workload $wl --bytes 4194304 --steps 3 --warmup 1 --sustain-s 0 --no-traffic --no-clock-probe > gpurun_out/r05/bench_$wl.log 2>&1
    grep '^{' gpurun_out/r05/bench_$wl.log > profiles/r05_bench_${wl/-/_}.json
done
python tools/host_policy_sweep.py > gpurun_out/r05/host_policy.log 2>&1
bash tools/first_call_latency.sh > gpurun_out/r05/first_call.log 2>&1
python tools/call_latency.py > gpurun_out/r05/call_latency.log 2>&1
cp profiles/r05_*.json profiles/r05_host_policy.md gpurun_out/r05/ 2>/dev/null
tail -3 gpurun_out/r05/first_call.log; head -30 gpurun_out/r05/host_policy.log
