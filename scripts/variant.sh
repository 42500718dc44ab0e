#!/bin/bash
# Builds a tuning variant of the library next to the real one: scripts/variant.sh NAME [extra hipcc flags...]
# Use it with DVO_AMD_LIB=dvo_slam_amd/libdvo_amd_var_NAME.so (results of -DDVO_ABLATE builds are wrong by construction).
cd "$(dirname "$0")/.."
name=$1; shift
src=dvo_slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -x hip \
  $src/dvo_kernels.hip $src/dvo_pyramid.cpp $src/dvo_tracker.cpp $src/dvo_sharded.cpp $src/dvo_probes.cpp $src/dvo_validator.cpp \
  $src/dvo_frontend.cpp $src/dvo_tum.cpp -lz \
  -o dvo_slam_amd/libdvo_amd_var_$name.so
