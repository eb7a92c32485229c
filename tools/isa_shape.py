"""Compact view of a kernel's instruction stream (one letter per instruction) between two line numbers of an ISA listing
(hipcc -S --cuda-device-only): M = f16 MFMA, X = scaled FP8/FP6 MFMA, r = ds_read, w = ds_write, D = LDS-DMA buffer load,
L = other VMEM load, S = VMEM store, v = VALU, s = SALU, W = s_waitcnt, B = s_barrier.  A line break after every MFMA group
makes front-loaded bursts of reads / DMA (a schedule the pinning did not hold) visible at a glance.
usage: python tools/isa_shape.py file.s first_line last_line"""
import re
import sys


def letter(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma_scale") or op.startswith("v_smfmac"):
        return "X"
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "r"
    if op.startswith("ds_"):
        return "w"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "D" if " lds" in ins else "L"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return "S"
    if op == "s_waitcnt":
        return "W"
    if op == "s_barrier":
        return "B"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "J"
    if op.startswith("v_"):
        return "v"
    if op.startswith("s_"):
        return "s"
    return "?"


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = []
    with open(path) as f:
        for n, line in enumerate(f, 1):
            if n < a or n > b:
                continue
            t = line.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            if "sched_barrier" in t and t.startswith(";"):
                continue
            out.append(letter(t))
    s = "".join(out)
    # break lines at transitions from MFMA to non-MFMA
    print(re.sub(r"([MX]+)", r"\1\n", s))
    from collections import Counter
    print(Counter(s))


if __name__ == "__main__":
    main()
