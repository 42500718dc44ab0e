#!/bin/bash
# The same SQ counters over (a) the isolated level-0 residual pass (36 pairs per launch) and (b) the k_tick launches of one
# 72-pair rolling batch in timing mode (every launch alone on the GPU): where do the issue cycles of a real batch go?
# Usage: scripts/pmc_batch.sh OUTDIR
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
rocprofv3 --pmc $C -d "$R/$out/iso" -o pmc --output-format csv -- python3 "$R/scripts/kernel_one.py" 0 36 0 10 > "$R/$out/iso.log" 2>&1 || exit 1
rocprofv3 --pmc $C -d "$R/$out/batch" -o pmc --output-format csv -- python3 "$R/scripts/issue_counts.py" run "$R/$out/batch_run.json" > "$R/$out/batch.log" 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d "$R/$out/iso2" -o pmc --output-format csv -- python3 "$R/scripts/kernel_one.py" 0 36 0 10 > "$R/$out/iso2.log" 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d "$R/$out/batch2" -o pmc --output-format csv -- python3 "$R/scripts/issue_counts.py" run "$R/$out/batch_run2.json" > "$R/$out/batch2.log" 2>&1 || exit 1
python3 - "$R/$out" <<'PY' | tee "$R/$out/summary.txt"
import csv, sys, glob, collections
root = sys.argv[1]
def tot(d):
    t = collections.Counter(); n = 0
    for f in glob.glob(f"{root}/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_tick" in r["Kernel_Name"]:
                t[r["Counter_Name"]] += float(r["Counter_Value"])
    return t
for name, a, b in (("isolated level-0 pass, 36 pairs per launch", "iso", "iso2"), ("k_tick launches of a 72-pair rolling batch (36 resident), each alone on the GPU", "batch", "batch2")):
    t, u = tot(a), tot(b)
    valu = u["SQ_INSTS_VALU"] - u["SQ_INSTS_MFMA"]
    issue = 4 * valu + 32 * u["SQ_INSTS_MFMA"]  # cycles a SIMD is held, summed over the SIMDs
    print(name)
    print(f"  VALU {valu:.4g}  MFMA {u['SQ_INSTS_MFMA']:.4g}  issue cycles (4 / 32 per instruction) {issue:.4g}  GRBM_GUI_ACTIVE {u['GRBM_GUI_ACTIVE']:.4g}")
    print(f"  issue cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = {issue / (u['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    print(f"  waves resident per busy SQ cycle: SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = {t['SQ_WAVE_CYCLES'] / t['SQ_BUSY_CYCLES']:.2f}")
    print(f"  share of wave cycles waiting on an instruction: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {t['SQ_WAIT_INST_ANY'] / t['SQ_WAVE_CYCLES']:.3f}")
    print(f"  VALU active: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {t['SQ_ACTIVE_INST_VALU'] / t['SQ_BUSY_CYCLES']:.2f};  MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {t['SQ_VALU_MFMA_BUSY_CYCLES'] / t['SQ_BUSY_CYCLES']:.2f}")
PY
