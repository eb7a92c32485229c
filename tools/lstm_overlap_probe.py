#!/usr/bin/env python3
"""Does the context LSTM (400 dependent launches per direction pass, latency bound, 132 small workgroups each) overlap
with weight-gradient GEMMs (one workgroup per CU, all 256 CUs) when the two run on different streams?
Times the LSTM forward+backward alone, a batch of wgrad_h3 launches alone, and both at once.  One JSON line.

    python tools/lstm_overlap_probe.py [--gemms 12]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemms", type=int, default=12, help="in_layer weight-gradient launches (5 taps) per trial")
    ap.add_argument("--priority", type=int, default=0, help="-1: LSTM stream high priority")
    ap.add_argument("--gemm-cus", type=int, default=0, help="> 0: the GEMM stream may use only this many CUs "
                                                             "(radmmm_stream_create_masked = hipExtStreamCreateWithCUMask)")
    args = ap.parse_args()
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd import ops
    from rad_mmm_amd.lstm import bilstm
    dev = torch.device("cuda:0")
    B, T, I, H = 32, 400, 1052, 524
    g = torch.Generator().manual_seed(0)
    lstm = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).to(dev)
    x = torch.randn(B, T, I, generator=g).to(dev).requires_grad_(True)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    gy = torch.randn(B * T, 1024, generator=g).to(dev)
    xa = torch.randn(B * T, 1024, generator=g).to(dev)
    s_l = torch.cuda.Stream(priority=args.priority)
    s_g = torch.cuda.Stream()
    if args.gemm_cus > 0:
        import ctypes
        from rad_mmm_amd._lib import lib, check
        raw = ctypes.c_void_p()
        check(lib.radmmm_stream_create_masked(args.gemm_cus, ctypes.byref(raw)), "stream_create_masked")
        s_g = torch.cuda.ExternalStream(raw.value)

    def run_lstm():
        with torch.cuda.stream(s_l):
            y = bilstm(lstm, x, lens)
            y.sum().backward()

    def run_gemms():
        with torch.cuda.stream(s_g):
            gy_t = ops.transpose_split_act(gy, 1024, B, T, None, 0, 1.0, "gy")
            x_t = ops.transpose_split_act(xa, 1024, B, T, lens, 1, 1.0, "x")
            for _ in range(args.gemms):
                ops.wgrad_h3_slabs(gy_t, x_t, 1024, 1024, 1024, 5, 2, 1.0)

    def timed(*fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(2):
        timed(run_lstm)
        timed(run_gemms)
    res = {"lstm_alone_ms": min(timed(run_lstm) for _ in range(3)),
           "gemms_alone_ms": min(timed(run_gemms) for _ in range(3)),
           "both_gemms_first_ms": min(timed(run_gemms, run_lstm) for _ in range(3)),
           "both_lstm_first_ms": min(timed(run_lstm, run_gemms) for _ in range(3)), "gemms": args.gemms,
           "lstm_priority": args.priority, "gemm_cus": args.gemm_cus}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
