#!/usr/bin/env python3
"""Can the parameter-only preparation of the flow steps (weight norm + split of every conv weight, the transposed copies
for the data gradients: ~1.75 ms of memory-bound launches per step) run UNDER the context LSTM's forward recurrence, now
that the recurrence is ONE cooperative launch on 132 of the 256 CUs (round 2 measured the same idea against 800 dependent
launches: slower)?  Times the LSTM forward alone, the preparation of 8 flows alone, and both on two streams.

    python tools/lstm_prep_overlap_probe.py [--flows 8]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flows", type=int, default=8)
    args = ap.parse_args()
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd.lstm import bilstm
    dev = torch.device("cuda:0")
    B, T, I, H = 32, 400, 1052, 524
    g = torch.Generator().manual_seed(0)
    lstm = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).to(dev)
    x = torch.randn(B, T, I, generator=g).to(dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    v5 = [(torch.randn(1024, 1024, 5, generator=g) * 0.02).to(dev) for _ in range(4)]
    v1 = [(torch.randn(1024, 1024, 1, generator=g) * 0.03).to(dev) for _ in range(4)]
    vs = (torch.randn(1024, 1128, 1, generator=g) * 0.03).to(dev)
    g5 = torch.ones(1024, 1, 1, device=dev)
    s_l, s_p = torch.cuda.Stream(), torch.cuda.Stream()

    def run_lstm():
        with torch.cuda.stream(s_l), torch.no_grad():
            bilstm(lstm, x, lens)

    def run_prep():
        with torch.cuda.stream(s_p):
            for _ in range(args.flows):
                ops.split_weight(vs, g5, 1152, nprod=2)
                for j in range(4):
                    Wh, Wl, _ = ops.split_weight(v5[j], g5, 1024, nprod=2)
                    ops.transpose_split(Wh, Wl, 1024, 1024, 1024, 2)
                    Wh, Wl, _ = ops.split_weight(v1[j], g5, 1024, nprod=2)
                    ops.transpose_split(Wh, Wl, 1024, 1024, 1024, 2)

    def timed(*fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(3):
        timed(run_lstm)
        timed(run_prep)
    res = {"lstm_fwd_alone_ms": min(timed(run_lstm) for _ in range(5)),
           "prep_alone_ms": min(timed(run_prep) for _ in range(5)),
           "both_lstm_first_ms": min(timed(run_lstm, run_prep) for _ in range(5)),
           "both_prep_first_ms": min(timed(run_prep, run_lstm) for _ in range(5)), "flows": args.flows}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
