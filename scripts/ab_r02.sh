#!/bin/bash
# A/B of library variants on the isolated kernel (distinct pairs) and a short bench.  usage: ab_r02.sh OUT "name|ENV=.. ENV=.." ...
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $out
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  echo "== $name ($envs)" >> $out/ab.log
  env $envs python3 scripts/kernel_pairs.py 36 20 >> $out/ab.log 2>&1 || exit 1
  env $envs python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $out/bench_$name.json 2>> $out/ab.log || exit 1
  python3 -c "import json; d=json.load(open('$out/bench_$name.json')); print('bench $name', round(d['value']), d['single_pair_latency_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_isolated_kernel']['frac'])" >> $out/ab.log
done
grep -v amdgpu.ids $out/ab.log
