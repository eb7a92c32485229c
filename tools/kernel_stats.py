#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg ms, share) from a rocprofv3 --kernel-trace results .db
(rocpd sqlite schema), for when the --stats CSVs were not merged back.
usage: kernel_stats.py results.db [steps] [--hist SUBSTR]
  -> prints a table; with `steps`, also ms per step; with --hist, the launch-duration clusters of the
     kernels whose name contains SUBSTR (one kernel serves several GEMM shapes: the per-shape average is
     what bench.py's roofline leg times, the all-shapes average is what --stats prints)."""
import sqlite3
import sys


def main():
    argv = list(sys.argv[1:])
    hist = None
    if "--hist" in argv:
        i = argv.index("--hist")
        hist = argv[i + 1]
        del argv[i:i + 2]
    db = sqlite3.connect(argv[0])
    steps = float(argv[1]) if len(argv) > 1 else None
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
    if hist:
        durs = [r[0] / 1e3 for r in db.execute(
            f"select d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like ?",
            (f"%{hist}%",))]
        durs.sort()
        print(f"\nlaunch-duration clusters of *{hist}* ({len(durs)} launches; a new cluster starts at a > 15 % jump):")
        start = 0
        for i in range(1, len(durs) + 1):
            if i == len(durs) or durs[i] > 1.15 * durs[i - 1]:
                c = durs[start:i]
                print(f"  {len(c):6d} launches  {c[0]:8.1f} .. {c[-1]:8.1f} us   mean {sum(c) / len(c):8.1f} us")
                start = i


if __name__ == "__main__":
    main()
