cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05b/gcm_mid; mkdir -p $OUT
for MIB in 1 4 8 16 64; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$MIB -o kt -- python tools/gcm_mid_trace.py $MIB > $OUT/kt$MIB.log 2>&1
  echo "== gcm encrypt one-shot $MIB MiB"; python tools/kernel_gaps.py $OUT/kt$MIB 8
done
