#!/bin/bash
# Resident pairs per launch through the device-resident item table (DVO_AMD_ITEMS_PER_LAUNCH) against the 36 of the kernel arguments.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
for spec in "36 72" "72 144" "72 288" "144 288" "108 216"; do
  set -- $spec
  DVO_AMD_ITEMS_PER_LAUNCH=$1 python3 bench.py --steps 10 --warmup 3 --in-flight $2 --no-extras --no-cpu-baseline > $out/b_$1_$2.json 2>> $out/err.log || exit 1
  python3 -c "import json; d=json.load(open('$out/b_$1_$2.json')); print('items/launch $1 in-flight $2:', round(d['value']), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['alg_bytes_per_launch']/1e6,1))" >> $out/ab.log
done
cat $out/ab.log
