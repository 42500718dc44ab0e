#!/bin/bash
# throughput of bench.py over host threads x pairs in flight (no side measurements)
for cfg in "4 36 288" "8 36 576" "8 18 288" "12 36 864" "16 18 576"; do
  set -- $cfg
  python bench.py --threads $1 --in-flight $2 --batch $3 --steps 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('T=$1 inflight=$2 batch=$3: %.0f pairs/s, %.2f ms/step, in-situ frac %.3f, avg launch %.1f us, isolated %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_isolated_kernel']['frac']))"
done
