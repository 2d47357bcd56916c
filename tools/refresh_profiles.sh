set -u
mkdir -p gpurun_out/r02b
for w in ctr ecb xts gcm; do python bench.py --workload $w --no-cpu 2>/dev/null | tail -1 > gpurun_out/r02b/bench_$w.json; done
python bench.py --workload xts --bytes $((4<<30)) --no-cpu 2>/dev/null | tail -1 > gpurun_out/r02b/bench_xts_c3.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02b/bench_ctr_full.json
bash tools/profile.sh r02b_ctr > /dev/null 2>&1
bash tools/profile.sh r02b_gcm --workload gcm > /dev/null 2>&1
bash tools/profile.sh r02b_xts --workload xts > /dev/null 2>&1
for t in ctr gcm xts; do cp gpurun_out/prof_r02b_$t/summary.txt gpurun_out/r02b/${t}_rocprof_summary.txt; cp gpurun_out/prof_r02b_$t/kt/*kernel_stats.csv gpurun_out/r02b/${t}_kernel_stats.csv 2>/dev/null; done
python tools/keysize_rates.py > gpurun_out/r02b/keysize_rates.log 2>&1
python tools/size_sweep.py > gpurun_out/r02b/size_sweep.log 2>&1
cat gpurun_out/r02b/bench_*.json | cut -c1-200
