#!/bin/bash
# PMC passes over the dominant kernel (bench.py --kernel-only).  One counter group per rocprofv3
# run, kernel-trace only (gpurun refuses --pmc together with sys/hip/hsa tracing).
# usage: tools/pmc_kernel.sh <outdir> [extra bench args]
set -u
OUT="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python "$ROOT/bench.py" --kernel-only ${EXTRA:-} > "$OUT/$name.log" 2>&1
}
EXTRA="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES
run sq3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr
ls -R "$OUT" | head -50
