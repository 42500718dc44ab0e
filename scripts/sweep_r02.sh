#!/bin/bash
# throughput sweep over host threads x resident pairs per tracker (default workload)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
for cfg in "8 72" "8 108" "8 144" "12 72" "16 36" "16 72" "6 144" "4 144" "4 288"; do
  set -- $cfg
  python3 bench.py --steps 8 --warmup 3 --threads $1 --in-flight $2 --no-extras --no-cpu-baseline > $out/sweep_$1_$2.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('$out/sweep_$1_$2.json')); print('threads $1 in_flight $2:', round(d['value']), 'pairs/s')" | tee -a $out/sweep.log
done
