#!/bin/bash
# HBM-side bytes per launch of the roofline leg's kernel (bench.py --dominant-only): FETCH_SIZE and WRITE_SIZE in
# their own rocprofv3 passes (kernel-trace only), MFMA-busy in a third; writes <outdir>/pmc_dominant.json in the form
# bench.py quotes (copy it to profiles/pmc_dominant.json).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE
# counts 16-B/lane loads (global_load and buffer_load..lds alike) at half their bytes -> doubled; both are in KiB.
# usage: tools/pmc_dominant.sh <outdir> <tag>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$1"; TAG="${2:-r02}"
case "$OUT" in /*) ;; *) OUT="$ROOT/$OUT" ;; esac        # rocprofv3 runs from /tmp: keep the output path absolute
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python "$ROOT/bench.py" --dominant-only > "$OUT/$c.log" 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- python "$ROOT/bench.py" --dominant-only > "$OUT/mfma.log" 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def mean(counter, sub):
    vals = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ("rowgemm_win_kernel" in r["Kernel_Name"] or "rowgemm_h3d_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
fetch, nf = mean("FETCH_SIZE", "FETCH_SIZE")
write, nw = mean("WRITE_SIZE", "WRITE_SIZE")
busy, _ = mean("SQ_VALU_MFMA_BUSY_CYCLES", "mfma")
sqb, _ = mean("SQ_BUSY_CYCLES", "mfma")
insts, _ = mean("SQ_INSTS_MFMA", "mfma")
gui, _ = mean("GRBM_GUI_ACTIVE", "mfma")
res = {"M": 12800, "kernel": "rowgemm_win_kernel / rowgemm_h3d_kernel: whichever `bench.py --dominant-only` launches (WN in_layer conv fwd, M=12800 N=1024 K=5x1024)",
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches_averaged": [nf, nw],
       "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024 if fetch and write else None,
       "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": sqb, "SQ_INSTS_MFMA": insts, "GRBM_GUI_ACTIVE": gui,
       "source": f"profiles/pmc_dominant.json ({tag}: tools/pmc_dominant.sh = rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, "
                 "own passes, over `bench.py --dominant-only`; FETCH_SIZE x2 per the guide's gfx950 correction)"}
json.dump(res, open(os.path.join(out, "pmc_dominant.json"), "w"), indent=1)
print(json.dumps(res))
PY
