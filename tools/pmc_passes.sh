#!/bin/bash
# Runs the rocprofv3 PMC passes (one run per pass, as the MI355X guide prescribes) on
# tools/prof_step.py and leaves per-dispatch CSVs under gpurun_out/pmc/<pass>/.
# usage: bash tools/pmc_passes.sh [steps] [config] [outdir-name]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-3}
CFG=${2:-sintel}
OUT=$ROOT/gpurun_out/${3:-pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/prof_step.py" "$STEPS" "$CFG" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum &&
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
