#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/pmcw"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python "$ROOT/bench.py" --dominant-only > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
run grbm GRBM_GUI_ACTIVE
cd "$ROOT"
python tools/pmc_summary.py gpurun_out/pmcw rowgemm_win > gpurun_out/pmcw/summary.txt 2>&1
find gpurun_out/pmcw -name "*.db" -delete; find gpurun_out/pmcw -name "*.csv" -size +1M -delete
cat gpurun_out/pmcw/summary.txt
