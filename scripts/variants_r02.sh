#!/bin/bash
# Times the k_tick variants (DVO_AMD_ACCUM x DVO_AMD_OCC) on the isolated residual pass and on a short bench run.
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in "mfma16 4" "mfma4 4" "mfma4 3"; do
  set -- $v
  export DVO_AMD_ACCUM=$1 DVO_AMD_OCC=$2
  echo "== ACCUM=$1 OCC=$2" >> $out/variants.log
  for lvl in 0 1 2 3; do python3 scripts/kernel_one.py $lvl 36 0 20 >> $out/variants.log 2>&1 || exit 1; done
  python3 scripts/kernel_one.py 0 36 1 20 >> $out/variants.log 2>&1
  python3 scripts/kernel_one.py 0 36 4 20 >> $out/variants.log 2>&1
  python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $out/bench_$1_$2.json 2>> $out/variants.log || exit 1
  python3 -c "import json,sys; d=json.load(open('$out/bench_$1_$2.json')); print('bench', d['value'], d['single_pair_latency_ms'], d['roofline']['frac'], d['roofline_isolated_kernel']['frac'])" >> $out/variants.log
done
cat $out/variants.log
