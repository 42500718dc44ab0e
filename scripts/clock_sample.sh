#!/bin/bash
# Samples the GPU's shader clock and power while a command runs: usage scripts/clock_sample.sh OUT.txt DELAY_S -- command...
# (one line per half second: sclk MHz, socket power W; starts sampling DELAY_S seconds after the command)
out=$1; delay=$2; shift; shift; shift
"$@" &
pid=$!
sleep "$delay"
for i in $(seq 1 120); do
  kill -0 $pid 2>/dev/null || break
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | awk '/sclk/ {gsub(/[()]|Mhz/,"",$NF); s=$NF} /Power/ && /[0-9]/ {p=$NF} END {print "sclk_MHz", s, "power_W", p}' >> "$out"
  sleep 0.5
done
wait $pid
