#!/usr/bin/env python3
"""How much does this GPU gain when two independent training steps share it?  Two decoder replicas run K steps each,
first one after the other on one stream, then concurrently on two streams (one host thread each).  The ratio says what
a two-stream arrangement INSIDE one step (weight-gradient chain on a side stream next to the data-gradient chain)
could at best recover from idle CUs (232 of 256 busy in the dominant launches) and from memory-bound passes
running beside GEMMs.  Prints one JSON line.

    python tools/stream_overlap_probe.py [--steps 6] [--batch 32] [--frames 800]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=800)
    args = ap.parse_args()
    import bench
    import rad_mmm_amd  # noqa: F401
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    import radmmm_synth as O
    dev = torch.device("cuda:0")
    cfg, sd = bench.procedural_state(bench.RADTTS)
    crit = RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)

    def replica(seed):
        dec = RADMMMFlow(use_accent=True, **bench.RADTTS)
        dec.load_state_dict(sd)
        dec = dec.to(dev).train()
        b = O.synthetic_batch(args.batch, args.frames, cfg, seed=seed, ragged=False)
        gb = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in b.items()}
        sl = SequenceLength(gb["lengths"])

        def step():
            for p in dec.parameters():
                p.grad = None
            out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
            loss = crit(out, None, sl, 0)["loss_mel"][0]
            loss.backward()
            return loss
        return step

    steps = [replica(1234), replica(4321)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(i, k):
        with torch.cuda.stream(streams[i]):
            for _ in range(k):
                steps[i]()

    for i in (0, 1):                      # warm-up on the stream each replica keeps (scale state, pools)
        run(i, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0, args.steps)
    torch.cuda.synchronize()
    run(1, args.steps)
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, args.steps)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    par = time.perf_counter() - t0
    print(json.dumps({"steps_each": args.steps, "sequential_ms_per_step": seq * 1e3 / (2 * args.steps),
                      "two_streams_ms_per_step": par * 1e3 / (2 * args.steps), "ratio": par / seq}))


if __name__ == "__main__":
    main()
