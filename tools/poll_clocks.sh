#!/bin/bash
# poll the shader/memory clocks and power while a long membench variant runs
for mode in 0 3 1 2; do
  ./tools/ubench/membench random $mode 6000 > /tmp/mb_$mode.log 2>&1 &
  pid=$!
  sleep 1.5
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.7
  done
  wait $pid
  cat /tmp/mb_$mode.log | grep -v input | cut -c1-70
done
