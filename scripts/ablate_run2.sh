#!/bin/bash
# Which part of the k_tick epilogue costs what (after scripts/ablate.sh 32 64 128 256 448).  usage: ablate_run2.sh OUT
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
python3 scripts/kernel_one.py 0 36 1 30 > /dev/null 2>&1   # warm the clocks
for rep in 1 2; do
for lib in default abl64 abl128 abl256 abl448 abl32; do
  for spec in "0 36 1" "1 36 1"; do
    set -- $spec
    if [ $lib = default ]; then unset DVO_AMD_LIB; else export DVO_AMD_LIB=$GRAFT_REPO_ROOT/dvo_slam_amd/libdvo_amd_$lib.so; fi
    echo -n "$lib steps=$((4 * $3)): " >> $out/ablate.log
    python3 scripts/kernel_one.py $1 $2 $3 30 2>/dev/null | tail -1 >> $out/ablate.log
  done
done
done
cat $out/ablate.log
