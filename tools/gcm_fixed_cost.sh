#!/bin/bash
# tools/gcm_fixed_cost.sh -- run on the GPU box: kernel trace of mid-size GCM encryptions, to see what the
# size-independent part of a call is made of (setup / fused prologue+epilogue / last levels / gaps).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02/gcm_fixed
mkdir -p $OUT
for MIB in 9 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$MIB -o kt -- python bench.py --no-cpu --no-verify --workload gcm --bytes $((MIB << 20)) --steps 50 --warmup 5 --settle-ms 0 > $OUT/kt$MIB.log 2>&1
  python - $OUT/kt$MIB $MIB <<'P'
import csv, glob, sys
d, mib = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(r['Kernel_Name'].split('(')[0][:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# steady state: last 60 launches of our kernels
ours = [k for k in ks if 'k_g' in k[0]]
tail = ours[-90:]
print("== %s MiB: kernel, duration us, gap to previous us" % mib)
for i, k in enumerate(tail[-9:]):
    prev = tail[len(tail) - 9 + i - 1]
    print("  %-40s %8.2f %8.2f" % (k[0], (k[2] - k[1]) / 1e3, (k[1] - prev[2]) / 1e3))
P
done
