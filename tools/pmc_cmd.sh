#!/bin/bash
# SQ / GRBM counter passes (one group per rocprofv3 run, kernel trace only) over an arbitrary command; prints the mean per
# kernel whose name contains <filter> and the kernel's mean duration.
# usage (GPU box): bash tools/pmc_cmd.sh <tag> <filter> <command ...>      -> gpurun_out/pmc_<tag>/summary.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; FILT="$2"; shift 2
OUT="$ROOT/gpurun_out/pmc_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name="$1"; shift
  ( cd "$ROOT" && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- "${CMD[@]}" > "$OUT/$name.log" 2>&1 )
}
CMD=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
run grbm GRBM_GUI_ACTIVE
cd "$ROOT"
python tools/pmc_summary.py "gpurun_out/pmc_$TAG" "$FILT" > "$OUT/summary.txt" 2>&1
python - "$OUT" "$FILT" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys
out, filt = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/*/**/*kernel_trace.csv", recursive=True))[:1]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"]]
    if d:
        print(f"kernel duration under the profiler: n={len(d)} mean={sum(d)/len(d):.1f} us min={min(d):.1f} max={max(d):.1f}")
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" -size +1M -delete
cat "$OUT/summary.txt"
