#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg ms, share) from a rocprofv3 --kernel-trace results .db
(rocpd sqlite schema), for when the --stats CSVs were not merged back.
usage: kernel_stats.py results.db [steps]   -> prints a table; with `steps`, also ms per step."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start) from {kd} d "
                      f"join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}" + ("  ms/step" if steps else ""))
    for n, c, t, a in rows[:40]:
        line = f"{n[:70]:70s} {c:7d} {t / 1e6:10.2f} {a / 1e3:9.1f} {100 * t / tot:6.2f}"
        if steps:
            line += f" {t / 1e6 / steps:8.2f}"
        print(line)
    print(f"{'TOTAL':70s} {sum(r[1] for r in rows):7d} {tot / 1e6:10.2f}")


if __name__ == "__main__":
    main()
