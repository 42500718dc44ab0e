#!/bin/bash
# Counter passes over the isolated level-0 residual pass (36 pairs, one launch's worth): one rocprofv3 --pmc run per group
# (<= 8 SQ counters or <= 4 TCP/TCC counters per pass).  Usage: scripts/pmc_isolated.sh OUTDIR [level items rounds reps]
out=$1; shift
lvl=${1:-0}; items=${2:-36}; rounds=${3:-0}; reps=${4:-10}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" -d "$R/$out/$name" -o pmc --output-format csv -- python3 "$R/scripts/kernel_one.py" $lvl $items $rounds $reps > "$R/$out/$name.log" 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES &&
run p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES &&
run p3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run p4 TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE &&
run p5 TCC_HIT_sum TCC_MISS_sum &&
run p6 FETCH_SIZE &&
run p7 WRITE_SIZE &&
python3 "$R/scripts/pmc_summary.py" $(find "$R/$out" -name '*counter_collection.csv') > "$R/$out/summary.txt"
