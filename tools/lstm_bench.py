#!/usr/bin/env python3
"""Context LSTM alone at the benchmark size (B = 32, T' = 400, I = 1040 -> H = 524, bidirectional): forward and
forward + backward time with the single-launch recurrence (default) and with one launch per step.

    RADMMM_DEBUG=1 python tools/lstm_bench.py [--batch 32] [--frames 400] [--hidden 524] [--inp 1040]
"""
import argparse
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--hidden", type=int, default=524)
    ap.add_argument("--inp", type=int, default=1040)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--modes", default="1,1p,0", help="comma list of 1 (single launch), 1p (+ probe loads), 0 (launch per step)")
    args = ap.parse_args()
    os.environ["RADMMM_DEBUG"] = "1"
    from rad_mmm_amd.lstm import bilstm
    dev = torch.device("cuda:0")
    B, T, I, H = args.batch, args.frames, args.inp, args.hidden
    g = torch.Generator().manual_seed(3)
    lstm = nn.LSTM(I, H, num_layers=1, batch_first=True, bidirectional=True).to(dev)
    x = (torch.randn(B, T, I, generator=g) * 0.5).to(dev).requires_grad_(True)
    lens = torch.tensor(sorted([int(T * (0.6 + 0.4 * i / max(1, B - 1))) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
    gy = (torch.randn(B, T, 2 * H, generator=g) * 1e-3).to(dev)
    outs = {}
    modes = args.modes.split(",")
    for mode in modes:
        os.environ["RADMMM_LSTM_PERSISTENT"] = mode[0]
        os.environ["RADMMM_LSTM_PROBE"] = "3" if mode.endswith("p") else "0"
        for _ in range(2):
            y = bilstm(lstm, x, lens)
            (y * gy).sum().backward()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(args.iters):
            lstm.zero_grad()
            x.grad = None
            e[0].record()
            y = bilstm(lstm, x, lens)
            e[1].record()
            (y * gy).sum().backward()
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        outs[mode] = (y.detach().clone(), x.grad.clone(), lstm.weight_hh_l0.grad.clone())
        print(f"RADMMM_LSTM_PERSISTENT={mode[0]} PROBE={os.environ['RADMMM_LSTM_PROBE']}: forward {tf / args.iters:.3f} ms, backward {tb / args.iters:.3f} ms "
              f"(B={B} T'={T} I={I} H={H}; includes the input projection and the gradient GEMMs)")
    if "1" not in outs or "0" not in outs:
        return
    a, b = outs["1"], outs["0"]
    for n, u, v in zip(("y", "dx", "dW_hh"), a, b):
        print(f"  single launch vs per step, {n}: max |diff| / max |ref| = {float((u - v).abs().max() / v.abs().max()):.2e}")


if __name__ == "__main__":
    main()
