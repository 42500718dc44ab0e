#!/bin/bash
# The isolated residual pass (36 pairs per launch) at levels 0..3 for a list of library builds: scripts/iso_levels.sh lib.so ...
# (variants from scripts/variant.sh; -DDVO_ABLATE=1 removes the Gram accumulation altogether: what is left is the rest of the step)
for lib in "$@"; do
  for lv in 0 1 2 3; do
    echo -n "$lib "; DVO_AMD_LIB=dvo_slam_amd/$lib python3 scripts/kernel_one.py $lv 36 0 20 2>/dev/null | grep "^level"
  done
done
