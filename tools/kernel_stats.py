#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg ms, share) from a rocprofv3 --kernel-trace results .db
(rocpd sqlite schema), for when the --stats CSVs were not merged back.
usage: kernel_stats.py results.db [steps] [--hist SUBSTR] [--gaps] [--json FILE] [--by-range]
  -> prints a table; with `steps`, also ms per step; with --hist, the launch-duration clusters of the
     kernels whose name contains SUBSTR (one kernel serves several GEMM shapes: the per-shape average is
     what bench.py's roofline leg times, the all-shapes average is what --stats prints); with --gaps, how much of the
     busiest window (the last `steps` steps: the 60 % of the dispatches at the end of the trace) the GPU spent between
     kernels (end of one dispatch -> start of the next, same device), by gap size; with --by-range (a trace taken with
     `RADMMM_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --hip-trace`), kernel time grouped by the innermost roctx range
     (rad_mmm_amd/_trace.py: flow<i>.fwd, flow<i>.bwd, context.fwd, lstm.bwd, loss) whose host-side launch produced the
     dispatch: dispatch -> its event's stack id -> the HIP launch call with that stack id -> the marker range on the same
     thread that encloses the call's start (backward ranges live on autograd's thread, like their launches)."""
import sqlite3
import sys


def main():
    argv = list(sys.argv[1:])
    gaps = "--gaps" in argv
    if gaps:
        argv.remove("--gaps")
    by_range = "--by-range" in argv
    if by_range:
        argv.remove("--by-range")
    jpath = None
    if "--json" in argv:
        i = argv.index("--json")
        jpath = argv[i + 1]
        del argv[i:i + 2]
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
    if jpath:                                           # machine-readable copy (bench.py quotes it: roofline_hbm)
        import json
        with open(jpath, "w") as f:
            json.dump({"steps": steps, "total_ms": tot / 1e6,
                       "kernels": {n: {"calls": c, "total_ms": t / 1e6, "avg_us": a / 1e3,
                                       "ms_per_step": (t / 1e6 / steps) if steps else None} for n, c, t, a in rows}}, f, indent=1)
    if gaps:
        ev = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
        ev = ev[int(len(ev) * 0.4):]                                  # the timed steps at the end of the run
        span = ev[-1][1] - ev[0][0]
        busy, idle, cur_end, big, prev_name, pairs = 0, [], ev[0][0], [], "", {}
        for s0, e0, name in ev:
            if s0 > cur_end:
                idle.append(s0 - cur_end)
                if s0 - cur_end > 1e5:
                    big.append((s0 - cur_end, prev_name, name))
                if s0 - cur_end > 5e3:
                    k = (prev_name, name)
                    pairs[k] = (pairs.get(k, (0, 0))[0] + 1, pairs.get(k, (0, 0))[1] + s0 - cur_end)
            busy += max(0, e0 - max(s0, cur_end))
            if e0 >= cur_end:
                prev_name = name
            cur_end = max(cur_end, e0)
        print(f"\nlast {len(ev)} dispatches: span {span / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), "
              f"idle {sum(idle) / 1e6:.2f} ms in {len(idle)} gaps")
        for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
            g = [x for x in idle if lo <= x < hi]
            print(f"  gaps {lo / 1e3:6.0f} .. {hi / 1e3 if hi < 1e12 else float('inf'):6.0f} us: {len(g):6d}  total {sum(g) / 1e6:7.3f} ms")
        short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at:")[:48]
        for g, a, b in sorted(big, reverse=True)[:40]:
            print(f"    {g / 1e3:8.0f} us   after {short(a):48s} before {short(b)}")
        print("  gaps > 5 us by (kernel before, kernel after), largest totals:")
        for (a, b), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:30]:
            print(f"    {n:5d} x  total {t / 1e6:7.3f} ms   after {short(a):48s} before {short(b)}")
    if by_range:
        ranges_report(db, tabs, kd, ks, steps)
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


def ranges_report(db, tabs, kd, ks, steps):
    """kernel time per roctx range (see the module docstring)"""
    import bisect
    rg = next(t for t in tabs if t.startswith("rocpd_region"))
    ev = next(t for t in tabs if t.startswith("rocpd_event"))
    st = next(t for t in tabs if t.startswith("rocpd_string"))
    cats = db.execute(f"select s.string, count(*) from {rg} r join {ev} e on e.id = r.event_id join {st} s on s.id = e.category_id "
                      f"group by s.string").fetchall()
    print("\nregion categories in the trace:", ", ".join(f"{c} x {n}" for c, n in cats))
    mk = [c for c, _ in cats if "MARKER" in c.upper() or "ROCTX" in c.upper()]
    if not mk:
        print("no marker regions in this trace (run with RADMMM_ROCTX=1 and rocprofv3 --marker-trace --hip-trace)")
        return
    q = ",".join("?" * len(mk))
    import json
    # (the range's text is the event's extdata {"message": ...}; the region's name is the API function, roctxThreadRangeA)
    marks = []
    for tid, a, b, ext, fn in db.execute(f"select r.tid, r.start, r.end, e.extdata, n.string from {rg} r join {ev} e on e.id = r.event_id "
                                         f"join {st} c on c.id = e.category_id join {st} n on n.id = r.name_id where c.string in ({q}) "
                                         f"order by r.start", mk):
        try:
            msg = json.loads(ext).get("message") if ext else None
        except Exception:
            msg = None
        marks.append((tid, a, b, msg or fn))
    print(f"{len(marks)} marker ranges; names: {sorted({m[3] for m in marks})[:24]}")
    # a dispatch's event carries the stack id of the HIP call that launched it: stack id -> (thread, host start of the call)
    api = {}
    for sid, tid, start in db.execute(f"select e.stack_id, r.tid, r.start from {rg} r join {ev} e on e.id = r.event_id "
                                      f"join {st} c on c.id = e.category_id where c.string like 'HIP_RUNTIME_API%'"):
        api.setdefault(sid, (tid, start))
    per_tid = {}
    for tid, a, b, name in marks:
        per_tid.setdefault(tid, []).append((a, b, name))
    starts = {tid: [m[0] for m in v] for tid, v in per_tid.items()}

    def innermost(tid, t):
        v = per_tid.get(tid)
        if not v:
            return "(no range)"
        i = bisect.bisect_right(starts[tid], t) - 1
        while i >= 0:                                   # ranges nest and are sorted by start: walk back to the first one still open
            if v[i][1] >= t:
                return v[i][2]
            i -= 1
        return "(no range)"
    agg, tot, lost = {}, 0, 0
    for corr, dur, kname in db.execute(f"select e.stack_id, d.end - d.start, s.kernel_name from {kd} d join {ev} e on e.id = d.event_id "
                                       f"join {ks} s on s.id = d.kernel_id"):
        h = api.get(corr)
        if h is None:
            lost += 1
            name = "(launch call not in the trace)"
        else:
            name = innermost(*h)
        a = agg.setdefault(name, [0, 0, {}])
        a[0] += 1
        a[1] += dur
        a[2][kname] = a[2].get(kname, 0) + dur
        tot += dur
    print(f"\n{'range':28s} {'launches':>9s} {'total_ms':>10s} {'%':>6s}" + ("  ms/step" if steps else "") + "   largest kernels")
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at:")[:34]
    for name, (n, t, ks_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        top = ", ".join(f"{short(k)} {v / 1e6:.1f}" for k, v in sorted(ks_.items(), key=lambda kv: -kv[1])[:3])
        print(f"{name[:28]:28s} {n:9d} {t / 1e6:10.2f} {100 * t / max(tot, 1):6.2f}" + (f" {t / 1e6 / steps:8.2f}" if steps else "") + "   " + top)
    if lost:
        print(f"({lost} dispatches without a traced launch call)")


if __name__ == "__main__":
    main()
