#!/bin/bash
# throughput of bench.py over host threads x pairs in flight x batch (no side measurements)
for cfg in "8 72 1152" "8 108 1728" "6 144 1728" "16 36 1152" "12 72 1728" "8 72 576" "8 144 2304"; do
  set -- $cfg
  python bench.py --threads $1 --in-flight $2 --batch $3 --steps 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T=$1 inflight=$2 batch=$3: %.0f pairs/s, %.2f ms/step, in-situ frac %.3f, avg launch %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']))"
done
