#!/bin/bash
# HBM-side bytes and SQ counters of the spline kernels (tools/spline_kernel_probe.py at configs[4]'s size): FETCH_SIZE and
# WRITE_SIZE in their own rocprofv3 passes (kernel-trace only), then two SQ groups.  FETCH_SIZE x2 for the 16-byte loads per
# the guide's gfx950 correction (MI355X_MICROARCH.md, HBM); both are KiB.
# usage (GPU box): bash tools/pmc_spline.sh   -> gpurun_out/pmc_spline/summary.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/pmc_spline"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name="$1"; shift
  ( cd "$ROOT" && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python tools/spline_kernel_probe.py --reps 3 > "$OUT/$name.log" 2>&1 )
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES
run grbm GRBM_GUI_ACTIVE
cd "$ROOT"
python - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pq_spline" in k and "bins" not in k and "ELi1EE" not in k:
            acc[k.split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows, h, K = 32000, 80, 32
q = rows * h * (2 * K + 1) * 4
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    for c, v in sorted(m.items()):
        print(f"    {c:28s} {v:.5g}")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        bwd = "bwd" in k
        alg = q * (2 if bwd else 1) + rows * h * 12
        tr = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        print(f"    traffic = 2 x FETCH + WRITE = {tr / 1e6:.1f} MB against {alg / 1e6:.1f} MB algorithmic ({tr / alg:.3f} x)")
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        print(f"    VALU active / (4 SIMDs x 256 CUs x GRBM cycles) = {m['SQ_ACTIVE_INST_VALU'] / (m['GRBM_GUI_ACTIVE'] * 1024):.3f}")
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" -size +1M -delete
cat "$OUT/summary.txt"
