#!/bin/bash
# Interleaved A/B of environment variants on the bench alone (columns: pairs/s, single-pair ms, frac per launch, launch us, concurrent frac).  usage: ab_bench.sh OUT REPEATS "name|ENV=.. ENV=..[|extra bench.py arguments]" ...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; reps=$1; shift; mkdir -p $out
cd $GRAFT_REPO_ROOT
for r in $(seq $reps); do
  for spec in "$@"; do
    IFS='|' read -r name envs extra <<< "$spec"
    env $envs python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $extra > $out/bench_${name}_$r.json 2>> $out/ab.err || exit 1
    python3 -c "import json; d=json.load(open('$out/bench_${name}_$r.json')); print('bench $name run $r', round(d['value']), round(d['single_pair_latency_ms'],4), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['concurrent']['frac'],4))" >> $out/ab.log
  done
done
cat $out/ab.log
