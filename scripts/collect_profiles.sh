#!/bin/bash
# Copies the summaries of a scripts/profile_bench.sh run (gpurun_out/<dir>) into profiles/ under the round's names.
# usage: scripts/collect_profiles.sh gpurun_out/r02x r02
src=$1; tag=$2
set -e
cp $src/bench_default.json profiles/${tag}_bench_default.json
cp $src/bench_t1_plain.json profiles/${tag}_bench_1thread.json
cp $src/bench_t1.json profiles/${tag}_bench_1thread_under_rocprof.json
cp $src/stats_t1/bench_kernel_stats.csv profiles/${tag}_bench_1thread_kernel_stats.csv
cp $src/bench_stats.json profiles/${tag}_bench_default_6threads_under_rocprof.json
cp $src/stats/bench_kernel_stats.csv profiles/${tag}_bench_default_6threads_kernel_stats.csv
cp $src/bench_1280x960.json profiles/${tag}_bench_1280x960.json
cp $src/bench_tileshard_peer.json profiles/${tag}_bench_tileshard_1rank_peer_exchange.json
cp $src/bench_tileshard_rccl.json profiles/${tag}_bench_tileshard_1rank_rccl.json
python3 scripts/profile_summary.py traffic $src/pmc_fetch/pmc_counter_collection.csv $src/pmc_write/pmc_counter_collection.csv $src/pmc_fetch.json $src/pmc_write.json > profiles/${tag}_traffic.json
python3 scripts/issue_counts.py fit $src/issue_run.json $src/pmc_issue/pmc_counter_collection.csv $src/pmc_issue_iso/pmc_counter_collection.csv > profiles/${tag}_issue.json
python3 scripts/profile_summary.py agreement $src/stats_t1/bench_kernel_trace.csv $src/bench_t1_plain.json > profiles/${tag}_k_tick_duration_rocprof_vs_bench.txt
cat profiles/${tag}_k_tick_duration_rocprof_vs_bench.txt
head -3 profiles/${tag}_bench_1thread_kernel_stats.csv | cut -c1-120
head -3 profiles/${tag}_bench_default_6threads_kernel_stats.csv | cut -c1-120
