#!/usr/bin/env python3
"""What is a GEMM launch of the flow step made of?  Times the step's launch shapes (M = 12 800 frames, 1024 channels, FP8-cross
scheme) on whatever library RADMMM_LIB_PATH names -- the product build and the timing-only builds of tools/floor_probe.sh
(-DRADMMM_TIMING=1/2/3: no cross-term MFMAs / no cross-fragment reads either / no MFMA at all; -DRADMMM_TIMING_NOCVT: the
weight gradient without its in-register hi8 conversions).  Each launch is bracketed by its own pair of HIP events and
alternates with a 52 MB device copy (a memory-bound neighbour, as inside the training step: twenty GEMMs back to back run
into the chip's power limit and read 15 % slower); the copy's own time is not part of the figure.

    RADMMM_LIB_PATH=... python tools/floor_probe.py [--reps 24] [--tag name]      -> one JSON line per case
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
    ap.add_argument("--tag", default="product")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    dev = torch.device("cuda:0")
    B, T, W = 32, 400, 1024
    N = B * T
    g = torch.Generator().manual_seed(0)
    x = torch.nn.functional.softplus(torch.randn(N, W, generator=g)).to(dev)
    gy = (torch.randn(N, W, generator=g) * 3e-3).to(dev)
    v5 = (torch.randn(W, W, 5, generator=g) * 0.02).to(dev)
    v1 = (torch.randn(W, W, 1, generator=g) * 0.03).to(dev)
    gg = torch.ones(W, 1, 1, device=dev)
    bias = torch.zeros(W, device=dev)
    SG = 2048.0
    GE = ops.X8_GRAD_EXP
    xh, xl = ops.split_f16(x, W, 1.0, W, 2, ops.X8_ACT_EXP)
    gh2, gl2 = ops._halves(2 * N, W, like=x)                         # [g_conv | gQ] pair of the fused data gradient
    a, b_ = ops.split_f16(gy, W, SG, W, 2, GE)
    gh2[:N], gl2[:N], gh2[N:], gl2[N:] = a, b_, a, b_
    W5h, W5l, _ = ops.split_weight(v5, gg, W, nprod=2)
    W1h, W1l, _ = ops.split_weight(v1, gg, W, nprod=2)
    W6h, W6l = ops._halves(6, W, W, like=x)                          # transposed tap stack with the res_skip slot
    W6h[:5], W6l[:5], W6h[5:], W6l[5:] = W5h, W5l, W1h, W1l
    y = torch.empty(N, W, device=dev)
    y2 = torch.zeros(N, W, device=dev)
    yh, yl = torch.empty_like(xh), torch.empty_like(xl)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    csb = torch.empty(W, device=dev)
    css = torch.empty(int(ops.lib.radmmm_rowgemm_h3_colsum_scratch_floats(N, W)), device=dev)
    filler_src, filler_dst = torch.empty(N, W, device=dev), torch.empty(N, W, device=dev)
    fwd = dict(nprod=2, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / ops.W_SCALE, T=T, sat_flag=flag,
               split_fmt=ops.SPLIT_X8A, ch_x8_exp=ops.X8_ACT_EXP, c2h_x8_exp=ops.X8_ACT_EXP, ldc=W, M=N, N=W, K=W, lens=lens)
    bwd = dict(fwd, a8_exp=GE, ch_x8_exp=GE, acc_scale=1.0 / (SG * ops.W_SCALE))
    in_fwd = dict(Ah=xh, Al=xl, lda_h=W, Bh=W5h, Bl=W5l, ldb_h=W, b_tap_stride_h=W5h.stride(0), taps=5, dil=2, sign=1,
                  a_mask_mode=1, bias=bias, pconv=1, ratio_taps=5, ratio_dil=2, postmask=1, act=1, Ch=yh, Cl=yl, ldch=W, ch_scale=1.0)
    dg = dict(Ah=gh2, Al=gl2, lda_h=W, Bh=W6h, Bl=W6l, ldb_h=W, b_tap_stride_h=W6h.stride(0), taps=5, dil=2, sign=-1,
              a_mask_mode=0, extra_tap=1, extra_a_rows=N, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, Ch=yh, Cl=yl, ldch=W,
              ch_scale=SG, colsum_out=csb, colsum_scratch=css)
    res = dict(Ah=xh, Al=xl, lda_h=W, Bh=W1h, Bl=W1l, ldb_h=W, bias=bias, act=1, C=y, C2=y2, ldc2=W, c2_accum=1)
    pair = dict(dact_h=xh, dact_x=xl, lddact_h=W, dact_x8_exp=ops.X8_ACT_EXP)
    cases = [
        ("5-tap fwd, SPLIT epilogue, C + pair (round 4)", lambda: rowgemm_h3(C=y, **in_fwd, **fwd), 5),
        ("5-tap fwd, SPLIT epilogue, pair only (C = NULL)", lambda: rowgemm_h3(C=None, **in_fwd, **fwd), 5),
        ("fused dgrad (5 taps + extra segment), fp32 dact_src, C + pair (round 4)",
         lambda: rowgemm_h3(C=y, dact_src=x, lddact=W, **dg, **bwd), 6),
        ("fused dgrad, dact from the split pair, pair only (C = NULL)", lambda: rowgemm_h3(C=None, **pair, **dg, **bwd), 6),
        ("fused dgrad, dact from the split pair, C + pair", lambda: rowgemm_h3(C=y, **pair, **dg, **bwd), 6),
        ("1x1 res fwd (bias + softplus, C + C2 accumulate)", lambda: rowgemm_h3(**res, **fwd), 1),
        ("1x1 plain (C only)", lambda: rowgemm_h3(Ah=xh, Al=xl, lda_h=W, Bh=W1h, Bl=W1l, ldb_h=W, C=y, **fwd), 1),
        ("wgrad_rm8 5 taps (in_layer)", lambda: ops.wgrad_rm8_slabs((gh2[:N], gl2[:N]), GE, (xh, xl), ops.X8_ACT_EXP, B, T, W, W, 5, 2,
                                                                  1.0 / SG, lens), 5),
        ("wgrad_rm8 1 tap (res_skip)", lambda: ops.wgrad_rm8_slabs((gh2[:N], gl2[:N]), GE, (xh, xl), ops.X8_ACT_EXP, B, T, W, W, 1, 1,
                                                                 1.0 / SG, None), 1),
    ]
    for name, fn, taps in cases:
        if args.only and args.only not in name:
            continue
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
        us = sorted(a_.elapsed_time(b__) * 1e3 for a_, b__ in ev)
        med = us[len(us) // 2]
        flop = 2.0 * N * W * W * taps
        print(json.dumps({"lib": args.tag, "case": name, "us_median": round(med, 1), "us_min": round(us[0], 1),
                          "algorithmic_tflops": round(flop / med / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
