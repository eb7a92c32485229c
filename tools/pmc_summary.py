#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv) per kernel: mean of each counter
over the dispatches of the kernels whose name contains a filter string."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "rowgemm"
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if filt in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(os.path.relpath(f, root).split(os.sep)[0], "|", k)
        for c, v in d.items():
            print(f"    {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
