#!/bin/bash
# Builds timing-only ablation variants of the library (results are wrong by construction) next to the real one.
cd "$(dirname "$0")/.."
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DDVO_ABLATE=$m -x hip \
    -fno-slp-vectorize dvo_slam_amd/csrc/dvo_kernels.hip dvo_slam_amd/csrc/dvo_tracker.cpp dvo_slam_amd/csrc/dvo_validator.cpp \
    dvo_slam_amd/csrc/dvo_frontend.cpp dvo_slam_amd/csrc/dvo_tum.cpp -lz -o dvo_slam_amd/libdvo_amd_abl$m.so || exit 1
done
