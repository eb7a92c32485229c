#!/usr/bin/env python3
"""Does the single-launch LSTM recurrence's step time follow the VOLUME of h / partial gradients that every workgroup pulls
through the fabric each step (round 6)?  The recurrence kernels alone (no input projection, no gradient GEMMs: round 5's
figure for H = 64 included them), forward and backward, over a sweep of hidden sizes at T = 400, B = 32: per direction H / 8
workgroups each read the whole h of their direction ([32 x H] hi + lo fp16 = 128 H bytes) with agent-scope loads, i.e.
2 x (H / 8) x 128 H = 32 H^2 bytes per step cross the fabric (H = 524: 8.8 MB, H = 384: 4.7 MB, H = 64: 0.13 MB).

    python tools/lstm_volume_probe.py [--frames 400] [--batch 32]     -> one JSON line per hidden size
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hidden", default="64,128,192,256,320,384,448,524,576,640,768")
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd._lib import lib, check, ptr, stream
    from rad_mmm_amd.lstm import _scratch
    dev = torch.device("cuda:0")
    B, T = args.batch, args.frames
    g = torch.Generator().manual_seed(1)
    for H in [int(h) for h in args.hidden.split(",")]:
        G0 = (torch.randn(B * T, 8 * H, generator=g) * 0.5).to(dev)
        W_hh = (torch.randn(2, 4 * H, H, generator=g) * (0.5 / H ** 0.5)).to(dev)
        dy = (torch.randn(B * T, 2 * H, generator=g) * 1e-2).to(dev)
        y = torch.empty(B * T, 2 * H, device=dev)
        c = torch.empty(B * T, 2 * H, device=dev)
        wsplit, hsplit = _scratch(B, H, 0, y), _scratch(B, H, 1, y)
        nq = int(lib.radmmm_lstm_hseq_bytes(B, T, H))
        hseq = torch.empty(nq // 4, device=dev) if nq else None
        wtpack, P, dcbuf = _scratch(B, H, 2, y), _scratch(B, H, 3, y), _scratch(B, H, 4, y)
        gscale = torch.ones(1, device=dev) * 64.0
        tf, tb = [], []
        for r in range(args.reps + 2):
            G = G0.clone()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            check(lib.radmmm_lstm_fwd(ptr(G), ptr(W_hh), ptr(y), ptr(c), None, ptr(wsplit), ptr(hsplit), ptr(hseq), B, T, H, stream()), "fwd")
            e[1].record()
            e[2].record()
            check(lib.radmmm_lstm_bwd(ptr(G), ptr(c), ptr(dy), ptr(W_hh), None, ptr(wtpack), ptr(P), ptr(dcbuf), B, T, H, ptr(gscale),
                                      stream()), "bwd")
            e[3].record()
            torch.cuda.synchronize()
            if r >= 2:
                tf.append(e[0].elapsed_time(e[1]))
                tb.append(e[2].elapsed_time(e[3]))
        tf.sort()
        tb.sort()
        f, b = tf[len(tf) // 2], tb[len(tb) // 2]
        print(json.dumps({"H": H, "single_launch": bool(nq), "workgroups": 2 * ((H + 7) // 8), "fabric_MB_per_step_fwd": round(32 * H * H / 1e6, 2),
                          "fwd_ms": round(f, 3), "bwd_ms": round(b, 3), "fwd_us_per_step": round(f * 1e3 / T, 2),
                          "bwd_us_per_step": round(b * 1e3 / T, 2)}), flush=True)


if __name__ == "__main__":
    main()
