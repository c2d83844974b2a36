#!/bin/bash
# sample power / clocks while the lab kernels run (read-only sensors)
for b in build_lab build_lab_noepi; do
  echo "=== $b"
  LAB_ITERS=12000 timeout -k 10 120 tools/$b.bin > gpurun_out/power_$b.out 2>&1 &
  pid=$!
  for i in $(seq 1 14); do
    timeout 10 rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr -s ' ' | tr '\n' '|'
    echo
    kill -0 $pid 2>/dev/null || break
  done
  wait $pid
  cat gpurun_out/power_$b.out
done
