#!/usr/bin/env python3
"""Launch-shape probe of the split-f16 row GEMM at the benchmark size (M = 12 800 frames, 1024 channels):
times the launches the flow step makes, with epilogue options switched on/off and the K extent varied, so
that the fixed part (launch + prologue + epilogue) separates from the K loop.  Prints one JSON line per case.

    python tools/gemm_probe.py [--reps 30]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--nprod", type=int, default=3, help="3 split-f16, 2 f16 + FP8 cross terms, 1 single product")
    args = ap.parse_args()
    NPR = args.nprod
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    dev = torch.device("cuda:0")
    B, T = args.batch, args.frames
    N = B * T
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 1024, generator=g).to(dev)
    gy = torch.randn(N, 1024, generator=g).to(dev)
    v5 = (torch.randn(1024, 1024, 5, generator=g) * 0.02).to(dev)
    v1 = (torch.randn(1024, 1024, 1, generator=g) * 0.03).to(dev)
    gg = torch.ones(1024, 1, 1, device=dev)
    bias = torch.zeros(1024, device=dev)
    xh, xl = ops.split_f16(x, 1024, 1.0, 1024, NPR, ops.X8_ACT_EXP)
    W5h, W5l, _ = ops.split_weight(v5, gg, 1024, nprod=NPR)
    W1h, W1l, _ = ops.split_weight(v1, gg, 1024, nprod=NPR)
    y = torch.empty(N, 1024, device=dev)
    y2 = torch.zeros(N, 1024, device=dev)
    add = torch.randn(N, 1024, device=dev)
    yh, yl = torch.empty_like(xh), torch.empty_like(xl)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    inv = 1.0 / ops.W_SCALE

    def conv(taps, **kw):
        Wh, Wl = (W5h, W5l) if taps > 1 else (W1h, W1l)
        base = dict(nprod=NPR, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, split_fmt=ops.fmt_a(NPR), ch_x8_exp=ops.X8_ACT_EXP,
                    Ah=xh, Al=xl, lda_h=1024, Bh=Wh, Bl=Wl, ldb_h=1024, b_tap_stride_h=Wh.stride(0), acc_scale=inv,
                    C=y, ldc=1024, M=N, N=1024, K=1024, taps=taps, dil=2, sign=1, T=T, lens=lens)
        base.update(kw)
        return lambda: rowgemm_h3(**base)

    fwd_epi = dict(a_mask_mode=1, bias=bias, pconv=1, ratio_taps=5, ratio_dil=2, postmask=1, act=1)
    cases = [
        ("in_layer fwd (5 taps, pconv+softplus, C + Ch/Cl)", conv(5, Ch=yh, Cl=yl, ldch=1024, ch_scale=1.0, **fwd_epi), 5),
        ("in_layer fwd, no Ch/Cl", conv(5, **fwd_epi), 5),
        ("in_layer fwd, plain epilogue (C only)", conv(5), 5),
        ("3 taps, pconv+softplus, C + Ch/Cl", conv(3, Ch=yh, Cl=yl, ldch=1024, ch_scale=1.0, **fwd_epi), 3),
        ("1 tap (K=1024), pconv+softplus, C + Ch/Cl", conv(1, Ch=yh, Cl=yl, ldch=1024, ch_scale=1.0, **fwd_epi), 1),
        ("1 tap (K=1024), plain epilogue (C only)", conv(1), 1),
        ("res fwd (K=1024, bias+softplus, C + C2 accumulate)", conv(1, bias=bias, act=1, C2=y2, ldc2=1024, c2_accum=1), 1),
        ("res fwd first layer (C + C2 store)", conv(1, bias=bias, act=1, C2=y2, ldc2=1024, c2_accum=0), 1),
        ("res fwd, no C2", conv(1, bias=bias, act=1), 1),
        ("res dgrad (K=1024, add + dact + rowscale, C + Ch/Cl)",
         conv(1, add=add, ldadd=1024, dact_src=x, lddact=1024, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, Ch=yh, Cl=yl,
              ldch=1024, ch_scale=1.0), 1),
        ("res dgrad, no add", conv(1, dact_src=x, lddact=1024, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, Ch=yh, Cl=yl,
                                   ldch=1024, ch_scale=1.0), 1),
        ("in_layer dgrad (5 taps, sign -1, premask, C only)", conv(5, sign=-1, premask=1), 5),
    ]
    for name, fn, taps in cases:
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        flop = 2.0 * N * 1024 * 1024 * taps
        print(json.dumps({"nprod": NPR, "case": name, "us": round(us, 1), "fp32_equiv_tflops": round(flop / us / 1e6, 1)}), flush=True)
    if NPR != 3:
        return
    # weight gradient incl. / excl. producers
    gy_t = ops.transpose_split_act(gy, 1024, B, T, None, 0, 1.0, "gy")
    x_t = ops.transpose_split_act(x, 1024, B, T, lens, 1, 1.0, "x")
    for name, taps, dil in (("wgrad_h3 in_layer (5 taps)", 5, 2), ("wgrad_h3 res (1 tap)", 1, 1)):
        fn = lambda: ops.wgrad_h3_slabs(gy_t, x_t, 1024, 1024, 1024, taps, dil, 1.0)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        print(json.dumps({"case": name, "us": round(us, 1),
                          "fp32_equiv_tflops": round(2.0 * N * 1024 * 1024 * taps / us / 1e6, 1)}), flush=True)
    for name, fn in (("transpose_split_act gy + colsum", lambda: ops.transpose_split_act(gy, 1024, B, T, None, 0, 1.0, "gy", colsum=(0, None, 1, 1))),
                     ("transpose_split_act x (masked)", lambda: ops.transpose_split_act(x, 1024, B, T, lens, 1, 1.0, "x")),
                     ("weight split+norm in_layer (5 taps)", lambda: ops.split_weight(v5, gg, 1024)),
                     ("weight split+norm in_layer, 8-bit cross", lambda: ops.split_weight(v5, gg, 1024, nprod=2)),
                     ("transpose_split W in_layer", lambda: ops.transpose_split(W5h, W5l, 1024, 1024, 1024))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"case": name, "us": round(e0.elapsed_time(e1) * 1e3 / args.reps, 1)}), flush=True)


if __name__ == "__main__":
    main()
