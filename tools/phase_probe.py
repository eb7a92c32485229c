#!/usr/bin/env python3
"""In-kernel phase times of the wide split GEMM (rowgemm_h3d_kernel) at the benchmark shapes.

Needs a measurement build of the library with -DRADMMM_PHASE_TIMERS:

    RADMMM_OUT=$PWD/rad_mmm_amd/libradmmm_hip_timers.so bash rad_mmm_amd/csrc/build.sh -DRADMMM_PHASE_TIMERS
    RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip_timers.so python tools/phase_probe.py

Every workgroup stamps the 100 MHz wall clock at entry / after the prologue / after the K loop / after the epilogue;
printed: medians over the workgroups in microseconds, the spread of the entry stamps and the launch's span.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nprod", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=400)
    args = ap.parse_args()
    NPR = args.nprod
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3, lib, LIB_PATH
    rd = lib.radmmm_debug_phase_read
    rd.restype = C.c_int
    rd.argtypes = [C.c_void_p, C.c_int]
    dev = torch.device("cuda:0")
    B, T = args.batch, args.frames
    N = B * T
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 1024, generator=g).to(dev)
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
    ylo = torch.empty_like(xh)
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
    split = dict(Ch=yh, Cl=yl, ldch=1024, ch_scale=1.0)
    cases = [
        ("in_layer fwd (5 taps, pconv+softplus, C + Ch/Cl + Clo)", conv(5, Clo=ylo, **split, **fwd_epi)),
        ("in_layer fwd (5 taps, pconv+softplus, C + Ch/Cl)", conv(5, **split, **fwd_epi)),
        ("in_layer fwd, no Ch/Cl", conv(5, **fwd_epi)),
        ("in_layer fwd, plain epilogue (C only)", conv(5)),
        ("1 tap, pconv+softplus, C + Ch/Cl", conv(1, **split, **fwd_epi)),
        ("1 tap, plain epilogue (C only)", conv(1)),
        ("res fwd (bias+softplus, C + C2 accumulate)", conv(1, bias=bias, act=1, C2=y2, ldc2=1024, c2_accum=1)),
        ("res dgrad (add + dact + rowscale, C + Ch/Cl)",
         conv(1, add=add, ldadd=1024, dact_src=x, lddact=1024, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, **split)),
    ]
    nwg = ((N + 223) // 224) * 4                                      # MB = 7 tiles x 4 column tiles
    buf = (C.c_ulonglong * (4 * nwg))()
    print(json.dumps({"lib": LIB_PATH}))
    for name, fn in cases:
        for _ in range(3):
            fn()
        rows = []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            assert rd(buf, nwg) == 0
            t = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.int64)
            ph = np.diff(t, axis=1) / 100.0                       # us
            rows.append([np.median(ph[:, 0]), np.median(ph[:, 1]), np.median(ph[:, 2]), ph[:, 2].max(),
                         (t[:, 0].max() - t[:, 0].min()) / 100.0, (t[:, 3].max() - t[:, 0].min()) / 100.0,
                         (t[:, 3].max() - t[:, 3].min()) / 100.0, e0.elapsed_time(e1) * 1e3])
        r = np.median(np.array(rows), axis=0)
        print(json.dumps({"case": name, "prologue_us": round(r[0], 2), "kloop_us": round(r[1], 2), "epilogue_us_median": round(r[2], 2),
                          "epilogue_us_max": round(r[3], 2), "entry_skew_us": round(r[4], 2), "span_us": round(r[5], 2),
                          "exit_skew_us": round(r[6], 2), "event_us": round(r[7], 1)}), flush=True)


if __name__ == "__main__":
    main()
