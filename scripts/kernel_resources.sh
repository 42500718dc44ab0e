#!/bin/bash
# Per-kernel registers / scratch / LDS / kernarg bytes of the gfx950 code object inside libdvo_amd.so (llvm-readelf --notes).
# usage: scripts/kernel_resources.sh [path/to/libdvo_amd.so]
set -e
LIB=${1:-$(dirname "$0")/../dvo_slam_amd/libdvo_amd.so}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/clang-offload-bundler --unbundle --type=o --input="$LIB" --output="$TMP/dev.co" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null || \
  $LLVM/llvm-objcopy --dump-section .hip_fatbin="$TMP/fat.bin" "$LIB" && [ -f "$TMP/dev.co" ] || \
  $LLVM/clang-offload-bundler --unbundle --type=o --input="$TMP/fat.bin" --output="$TMP/dev.co" --targets=hipv4-amdgcn-amd-amdhsa--gfx950
$LLVM/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
print("%-64s %5s %5s %8s %6s %8s %8s" % ("kernel", "vgpr", "sgpr", "scratch", "spill", "lds", "kernarg"))
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    print("%-64s %5s %5s %8s %6s %8s %8s" % (dem[:64], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("kernarg_segment_size")))
'
rm -rf "$TMP"
