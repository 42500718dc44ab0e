#!/bin/bash
# Round-3 sweeps of the streaming bench: usage scripts/sweep_r03.sh OUTDIR name:"ENV=.. ENV=.."|"bench args" ...
# each case: name|env assignments|bench arguments
out=gpurun_out/$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
for c in "$@"; do
  name=${c%%|*}; rest=${c#*|}; envs=${rest%%|*}; args=${rest#*|}
  env $envs python3 "$R/bench.py" --steps 12 --warmup 3 --no-extras --no-cpu-baseline $args > "$R/$out/$name.json" 2> "$R/$out/$name.err" || { echo "$name failed"; tail -3 "$R/$out/$name.err"; exit 1; }
  python3 - "$R/$out/$name.json" "$name" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=d['roofline']
print(f"{sys.argv[2]:28s} {d['value']:9.0f} pairs/s  {d['ms_per_step']:6.2f} ms/step  launch {r['avg_launch_us']:.1f} us frac {r['frac']:.3f} conc {r['concurrent']['frac']:.3f} lat {d['single_pair_latency_ms']:.3f}", flush=True)
PY
done
