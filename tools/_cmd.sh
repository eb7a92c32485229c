#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r06_pytest_gpu.txt 2>&1
tail -18 gpurun_out/r06_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity_vs_cpu'), d.get('cpu_baseline'))
print(json.dumps(d.get('roofline_mfma'))[:1500])
PY
timeout 600 bash tools/prof_step.sh r06 > /dev/null 2>&1
head -6 gpurun_out/r06_kernel_stats.txt | cut -c1-130
# roctx schema exploration
cd /tmp && export TMPDIR=/tmp
RADMMM_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-trace -d /tmp/rtx -- python $GRAFT_REPO_ROOT/bench.py --step-only --steps 1 --warmup 1 --batch 8 --frames 400 > /tmp/rtx.log 2>&1
tail -2 /tmp/rtx.log | cut -c1-200
DB=$(find /tmp/rtx -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)
python - "$DB" > $GRAFT_REPO_ROOT/gpurun_out/r06_rocpd_schema.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, sql in db.execute("select name, sql from sqlite_master where type in ('table','view')"):
    print("==", name)
    print(sql)
    try:
        n = db.execute(f"select count(*) from {name}").fetchone()[0]
        print("   rows:", n)
        for r in db.execute(f"select * from {name} limit 3"):
            print("   ", r)
    except Exception as e:
        print("   err", e)
PY
wc -l $GRAFT_REPO_ROOT/gpurun_out/r06_rocpd_schema.txt
