#!/bin/bash
# throughput of bench.py over host threads x pairs in flight x pairs per launch (no side measurements)
for cfg in "8 36 576 36" "4 72 576 72" "4 144 1152 144" "8 72 1152 72" "2 144 576 144" "2 288 1152 288" "6 96 1152 96"; do
  set -- $cfg
  DVO_AMD_ITEMS_PER_LAUNCH=$4 python bench.py --threads $1 --in-flight $2 --batch $3 --steps 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('T=$1 inflight=$2 batch=$3 per-launch=$4: %.0f pairs/s, %.2f ms/step, in-situ frac %.3f, avg launch %.1f us, isolated %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_isolated_kernel']['frac']))"
done
