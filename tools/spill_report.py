#!/usr/bin/env python3
"""Register spills per kernel instantiation of HIP sources (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU):
    python tools/spill_report.py rad_mmm_amd/csrc/rowgemm_win.hip [more.hip ...] [-- extra hipcc flags]
prints kernel, VGPRs, scratch bytes per lane, spilled VGPRs; exit code 1 if any kernel spills."""
import os
import re
import subprocess
import sys


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    bad = 0
    for f in args:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.basename(f),
                            "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra],
                           cwd=os.path.dirname(os.path.abspath(f)), capture_output=True, text=True)
        name, row = None, {}
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                name, row = m.group(1), {}
                continue
            m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
            if m and name:
                row[m.group(1)] = int(m.group(2))
                if m.group(1) == "VGPRs Spill":
                    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:46]
                    sp, sc = row["VGPRs Spill"], row.get("ScratchSize [bytes/lane]", 0)
                    bad += 1 if (sp or sc) else 0
                    print("%-48s VGPRs %4d  scratch %5d  spills %4d%s" % (short, row.get("VGPRs", -1), sc, sp, "   <-- SPILLS" if (sp or sc) else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
