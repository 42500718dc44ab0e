#!/bin/bash
# The judged profile set of a round: kernel stats of exactly the driver's command, the PMC traffic passes on the same
# workload, the issue counts.  Usage: scripts/profile_bench.sh OUTDIR [steps: default|pmc|issue|stats|stats1 ...]
out=gpurun_out/$1; shift
steps=${@:-default t1plain big shard pmc issue stats1 stats}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
for s in $steps; do
  case $s in
    default) python3 "$R/bench.py" --steps 20 --warmup 5 > "$R/$out/bench_default.json" 2> "$R/$out/bench_default.err" || exit 1 ;;
    pmc)
      export DVO_AMD_LAUNCH_LOCK=1  # (multi-thread run under the profiler's queue interceptor: see the stats step)
      rocprofv3 --pmc FETCH_SIZE -d "$R/$out/pmc_fetch" -o pmc --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --prime 1 --no-extras --no-cpu-baseline --no-live-counters > "$R/$out/pmc_fetch.json" 2> "$R/$out/pmc_fetch.err" || exit 1
      rocprofv3 --pmc WRITE_SIZE -d "$R/$out/pmc_write" -o pmc --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --prime 1 --no-extras --no-cpu-baseline --no-live-counters > "$R/$out/pmc_write.json" 2> "$R/$out/pmc_write.err" || exit 1
      unset DVO_AMD_LAUNCH_LOCK ;;
    issue)
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA -d "$R/$out/pmc_issue" -o pmc --output-format csv -- python3 "$R/scripts/issue_counts.py" run "$R/$out/issue_run.json" > "$R/$out/issue.log" 2>&1 || exit 1
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA -d "$R/$out/pmc_issue_iso" -o pmc --output-format csv -- python3 "$R/scripts/kernel_one.py" 0 36 0 10 > "$R/$out/issue_iso.log" 2>&1 || exit 1 ;;
    t1plain) python3 "$R/bench.py" --threads 1 --batch 72 --in-flight 36 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-live-counters > "$R/$out/bench_t1_plain.json" 2> "$R/$out/bench_t1_plain.err" || exit 1 ;;
    big) python3 "$R/bench.py" --width 1280 --height 960 --batch 288 --distinct 48 --distinct-refs 6 --steps 8 --warmup 2 --no-extras --no-live-counters --cpu-seconds 8 > "$R/$out/bench_1280x960.json" 2> "$R/$out/bench_1280x960.err" || exit 1 ;;
    shard)
      DVO_AMD_EXCHANGE=peer python3 "$R/bench.py" --tile-shard --steps 8 --warmup 3 > "$R/$out/bench_tileshard_peer.json" 2> "$R/$out/bench_tileshard_peer.err" || exit 1
      DVO_AMD_EXCHANGE=rccl python3 "$R/bench.py" --tile-shard --steps 8 --warmup 3 > "$R/$out/bench_tileshard_rccl.json" 2> "$R/$out/bench_tileshard_rccl.err" || exit 1 ;;
    stats1) rocprofv3 --kernel-trace --stats -d "$R/$out/stats_t1" -o bench --output-format csv -- python3 "$R/bench.py" --threads 1 --batch 72 --in-flight 36 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-live-counters > "$R/$out/bench_t1.json" 2> "$R/$out/bench_t1.err" || exit 1 ;;
    # The driver's 8-thread command under the kernel trace.  DVO_AMD_LAUNCH_LOCK=1: rocprofv3's queue interceptor reads past
    # the end of an AQL ring when two host threads publish packets to one hardware queue across the ring's wrap
    # (profiles/r03_rocprofv3_sigsegv_root_cause.md: the two SIGSEGVs of round 2); under the lock every doorbell finds one packet.
    # DVO_BENCH_MAPS: /proc/self/maps of the run, so that any raw-address stack is attributable.
    stats) export DVO_AMD_LAUNCH_LOCK=1 DVO_BENCH_MAPS="$R/$out/bench_stats.maps"
      rocprofv3 --kernel-trace --stats -d "$R/$out/stats" -o bench --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-live-counters > "$R/$out/bench_stats.json" 2> "$R/$out/bench_stats.err" || exit 1
      unset DVO_AMD_LAUNCH_LOCK DVO_BENCH_MAPS ;;
  esac
done
# the per-dispatch traces are large: keep the stats and drop the traces unless asked
find "$R/$out" -name '*kernel_trace.csv' -size +20M -delete
ls -la "$R/$out"
