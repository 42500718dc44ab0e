#!/bin/bash
# Prologue / epilogue share of a k_tick block: the isolated residual pass with the full step loop, without the steps (ablation 16)
# and without the epilogue (ablation 32), at the segment lengths a batch uses.  usage: ablate_run.sh OUT   (after scripts/ablate.sh 16 32)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
for lib in default abl16 abl32; do
  for spec in "0 36 1" "0 36 2" "0 36 4" "1 36 1" "1 36 2" "2 36 1"; do
    set -- $spec
    if [ $lib = default ]; then unset DVO_AMD_LIB; else export DVO_AMD_LIB=$GRAFT_REPO_ROOT/dvo_slam_amd/libdvo_amd_$lib.so; fi
    echo -n "$lib steps=$((4 * $3)): " >> $out/ablate.log
    python3 scripts/kernel_one.py $1 $2 $3 20 2>/dev/null | tail -1 >> $out/ablate.log
  done
done
cat $out/ablate.log
