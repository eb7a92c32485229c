#!/bin/bash
# PMC passes over radmmm_wgrad_rm at the in_layer shape (12 800 frames, 1024 x 1024, 5 taps, dilation 2): the compiled
# tools/wgrad_rm_probe.hip runs the same kernel structure stand-alone.  One counter group per rocprofv3 run.
# usage: tools/pmc_wgrad_rm.sh <outdir>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$1"; case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac
mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 "$ROOT/tools/wgrad_rm_probe.hip" -o /tmp/wgrad_rm_probe || exit 1
cd /tmp && export TMPDIR=/tmp
run() { local n="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -- /tmp/wgrad_rm_probe > "$OUT/$n.log" 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_LDS
run sq2 SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_rm" in r["Kernel_Name"] and r["Grid_Size"] == "61440":      # 240 workgroups: the 5-tap launch
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d): print("%-28s %.4g  (%d launches)" % (k, sum(d[k]) / len(d[k]), len(d[k])))
PY
