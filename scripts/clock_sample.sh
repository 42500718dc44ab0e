#!/bin/bash
# Samples the GPU's shader clock and power while a command runs: usage scripts/clock_sample.sh OUT.txt -- command...
out=$1; shift; shift
"$@" &
pid=$!
sleep 12
for i in $(seq 1 40); do
  kill -0 $pid 2>/dev/null || break
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ';' >> "$out"
  echo >> "$out"
  sleep 0.5
done
wait $pid
