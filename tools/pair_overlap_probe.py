#!/usr/bin/env python3
"""VERDICT r5 item 1a, measured before built: what would a GROUPED launch of res_skip[j] (1x1, K = 1024) with in_layer[j+1]
(5 taps) buy?  Both read only h_{j+1} and do not depend on each other (reference common.py:829-832).  A grouped persistent
grid is, at best, what the hardware's own workgroup dispatcher does with the two launches on two streams: the 24 CUs the
232-workgroup 5-tap grid leaves idle take 1x1 tiles, and the 1x1 tiles' prologue / epilogue run under the big tiles' K loops.
So the probe runs the WN forward chain of one flow step (M = 12 800, 1024 channels, FP8-cross scheme, the step's own
epilogue kinds)

    serial   : in0 res0 in1 res1 in2 res2 in3 res3            (one stream, the product's order)
    paired   : in0 [res0 || in1] [res1 || in2] [res2 || in3] res3   (res0..2 on a side stream behind an event)

and prints the chain's duration both ways (HIP events on the main stream, a 52 MB copy between repetitions as inside the
step).  --mb N forces the 5-tap tile height (RADMMM_H3W_MB under RADMMM_DEBUG): MB = 8 gives 200 big workgroups, i.e. 56 CUs
for the small tiles.

    RADMMM_DEBUG=1 python tools/pair_overlap_probe.py [--reps 24]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=24)
    args = ap.parse_args()
    os.environ.setdefault("RADMMM_DEBUG", "1")
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    dev = torch.device("cuda:0")
    B, T, W = 32, 400, 1024
    N = B * T
    g = torch.Generator().manual_seed(0)
    x = torch.nn.functional.softplus(torch.randn(N, W, generator=g)).to(dev)
    gg = torch.ones(W, 1, 1, device=dev)
    bias = torch.zeros(W, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    Hp = [ops.split_f16(x, W, 1.0, W, 2, ops.X8_ACT_EXP)] + [ops._halves(N, W, like=x) for _ in range(4)]
    W5, W1 = [], []
    for j in range(4):
        v5 = (torch.randn(W, W, 5, generator=g) * 0.02).to(dev)
        v1 = (torch.randn(W, W, 1, generator=g) * 0.03).to(dev)
        W5.append(ops.split_weight(v5, gg, W, nprod=2)[:2])
        W1.append(ops.split_weight(v1, gg, W, nprod=2)[:2])
    R = [torch.empty(N, W, device=dev) for _ in range(4)]
    OUTh, OUTl = ops._halves(N, W, like=x)
    gin = dict(nprod=2, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / ops.W_SCALE, T=T, sat_flag=flag)
    gout = dict(split_fmt=ops.SPLIT_X8A, ch_x8_exp=ops.X8_ACT_EXP, c2h_x8_exp=ops.X8_ACT_EXP)

    def in_layer(j):
        rowgemm_h3(Ah=Hp[j][0], Al=Hp[j][1], lda_h=W, Bh=W5[j][0], Bl=W5[j][1], ldb_h=W, b_tap_stride_h=W5[j][0].stride(0),
                   C=None, ldc=W, M=N, N=W, K=W, taps=5, dil=2 ** j, sign=1, lens=lens, a_mask_mode=1, bias=bias, pconv=1,
                   ratio_taps=5, ratio_dil=2 ** j, postmask=1, act=1, Ch=Hp[j + 1][0], Cl=Hp[j + 1][1], ldch=W, ch_scale=1.0,
                   **gin, **gout)

    def res(j):
        kw = dict(Ah=Hp[j + 1][0], Al=Hp[j + 1][1], lda_h=W, Bh=W1[j][0], Bl=W1[j][1], ldb_h=W, C=R[j], ldc=W, M=N, N=W, K=W,
                  bias=bias, act=1, **gin, **gout)
        if j == 3:
            kw.update(C2=None, ldc2=W, c2_src=R[:3], C2h=OUTh, C2l=OUTl, ldc2h=W, c2h_scale=1.0)
        rowgemm_h3(**kw)

    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()

    def serial():
        for j in range(4):
            in_layer(j)
            res(j)

    def paired():
        evs = []
        for j in range(4):
            in_layer(j)
            if j < 3:
                e = torch.cuda.Event()
                e.record(main_s)
                with torch.cuda.stream(side):
                    side.wait_event(e)
                    res(j)
                    d = torch.cuda.Event()
                    d.record(side)
                evs.append(d)
        for d in evs:
            main_s.wait_event(d)
        res(3)

    def only_in():
        for j in range(4):
            in_layer(j)

    def only_res():
        for j in range(4):
            res(j)

    filler_src, filler_dst = torch.empty(N, W, device=dev), torch.empty(N, W, device=dev)

    def timeit(fn):
        for _ in range(3):
            fn()
            filler_dst.copy_(filler_src)
        ev = []
        torch.cuda.synchronize()
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            filler_dst.copy_(filler_src)
            ev.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return round(us[len(us) // 2], 1), round(us[0], 1)

    for mb in ("default", "8", "6"):
        if mb == "default":
            os.environ.pop("RADMMM_H3W_MB", None)
        else:
            os.environ["RADMMM_H3W_MB"] = mb
        out = {"mb": mb}
        for name, fn in (("serial", serial), ("paired", paired), ("only_in", only_in), ("only_res", only_res)):
            out[name + "_us_median"], out[name + "_us_min"] = timeit(fn)
        out["paired_minus_serial_us"] = round(out["paired_us_median"] - out["serial_us_median"], 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
