#!/usr/bin/env python3
"""Grid / workgroup sizes and durations of the dispatches of the kernels whose name contains SUBSTR, from a rocprofv3 .db:
    kernel_grids.py results.db SUBSTR [SUBSTR ...]     (one line per distinct (kernel, grid) with count and mean duration)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
    wx = "workgroup_size_x" if "workgroup_size_x" in cols else "workgroup_x"
    for sub in sys.argv[2:]:
        rows = db.execute(f"select s.kernel_name, d.{gx}, d.{gx.replace('_x', '_y')}, d.{wx}, count(*), avg(d.end - d.start) from {kd} d "
                          f"join {ks} s on d.kernel_id = s.id where s.kernel_name like ? group by 1, 2, 3, 4 order by 6 desc",
                          (f"%{sub}%",)).fetchall()
        for n, g0, g1, w, c, a in rows:
            print(f"{n.replace('_ZN12_GLOBAL__N_1', '')[:60]:60s} grid {g0 // max(w, 1):6d} x {g1:4d} workgroups of {w:4d}  {c:5d} launches  {a / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
