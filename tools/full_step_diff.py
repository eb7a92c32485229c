#!/usr/bin/env python3
"""Per-kernel ms/step of a full-training-step kernel trace minus the decoder-only trace (tools/kernel_stats.py --json files):
what the step spends outside the decoder's forward + backward.
    python tools/full_step_diff.py gpurun_out/<tag>_full_step_kernel_stats.json [profiles/r04_kernel_stats.json] [rows]"""
import json
import sys


def main():
    cur = json.load(open(sys.argv[1]))
    dec = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "profiles/r04_kernel_stats.json"))
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows, tot = [], 0.0
    for n, k in cur["kernels"].items():
        base = dec["kernels"].get(n, {}).get("ms_per_step", 0.0)
        rows.append((k["ms_per_step"] - base, n, k["calls"] / cur["steps"], k["ms_per_step"], base))
    for n, k in dec["kernels"].items():
        if n not in cur["kernels"]:
            rows.append((-k["ms_per_step"], n, 0.0, 0.0, k["ms_per_step"]))
    tot = sum(r[0] for r in rows)
    rows.sort(key=lambda r: -abs(r[0]))
    print(f"{'kernel':100s} {'calls':>6s} {'full':>7s} {'decoder':>7s} {'extra':>7s}   (ms per step)")
    for extra, n, calls, ms, base in rows[:top]:
        print(f"{n[:100]:100s} {calls:6.1f} {ms:7.3f} {base:7.3f} {extra:+7.3f}")
    print(f"kernel time per step: full {cur['total_ms'] / cur['steps']:.2f} ms, decoder only {dec['total_ms'] / dec['steps']:.2f} ms, "
          f"difference {tot:+.2f} ms")


if __name__ == "__main__":
    main()
