#!/usr/bin/env python3
"""radmmm_wgrad_rm8 at the benchmark shapes (12 800 frames, 1024 x 1024; 5 taps dil 2 = the in_layer conv, 1 tap = res_skip):
launch time, and -- for A/B builds of the library (RADMMM_LIB_PATH) -- a dump / bit comparison of the results.
    python tools/wgrad8_probe.py [--dump FILE | --compare FILE] [--reps 30] [--loop N]   (--loop: launches only, for rocprofv3 --pmc)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dump")
    ap.add_argument("--compare")
    ap.add_argument("--loop", type=int, default=0)
    args = ap.parse_args()
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    dev = torch.device("cuda:0")
    B, T, C = 32, 400, 1024
    g = torch.Generator().manual_seed(0)
    lens = torch.tensor([T - (7 * b) % 90 for b in range(B)], dtype=torch.int32, device=dev)
    x = torch.nn.functional.softplus(torch.randn(B * T, C, generator=g) * 2).to(dev)
    gy = (torch.randn(B * T, C, generator=g) * 3e-3).to(dev)
    SG = 2048.0
    gh, gx = ops.split_f16(gy, C, SG, C, 2, ops.X8_GRAD_EXP)
    xh, xx = ops.split_f16(x, C, 1.0, C, 2, ops.X8_ACT_EXP)
    cases = [("in_layer (5 taps, dil 2, masked)", 5, 2, lens), ("res_skip (1 tap)", 1, 1, None), ("5 taps, dil 1, masked", 5, 1, lens)]
    outs = {}
    for name, taps, dil, ln in cases:
        fn = lambda: ops.wgrad_rm8_slabs((gh, gx), ops.X8_GRAD_EXP, (xh, xx), ops.X8_ACT_EXP, B, T, C, C, taps, dil, 1.0 / SG, ln)
        if args.loop:
            if taps == 5 and dil == 2:
                for _ in range(args.loop):
                    fn()
                torch.cuda.synchronize()
            continue
        P = fn()
        torch.cuda.synchronize()
        outs[name] = P.sum(0).cpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        print(json.dumps({"case": name, "splits": int(P.shape[0]), "us": round(us, 1),
                          "fp32_equiv_tflops": round(2.0 * B * T * C * C * taps / us / 1e6, 1)}), flush=True)
    if args.dump:
        torch.save(outs, args.dump)
    if args.compare:
        ref = torch.load(args.compare)
        for k, v in outs.items():
            same = torch.equal(v.view(torch.int32), ref[k].view(torch.int32))
            d = float((v - ref[k]).abs().max() / ref[k].abs().max())
            print(f"compare {k}: {'bit-identical' if same else 'DIFFERENT'} (max rel diff {d:.2e})")


if __name__ == "__main__":
    main()
