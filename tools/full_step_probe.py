#!/usr/bin/env python3
"""Only bench.py's full-step leg (TTSTrainingStep.training_step + backward + clip + FlatRAdam on the benchmark batch), for a
kernel trace of the WHOLE training step (3 warm-up + `steps` timed + 1 sync-counting + 1 section-timing step = steps + 5 steps in the trace):
    rocprofv3 --kernel-trace --stats -d out -- python tools/full_step_probe.py [--steps 7]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--joint", action="store_true", help="BASELINE configs[3]: RADMMM decoder + the four attribute predictors")
    args = ap.parse_args()
    import bench
    import radmmm_synth as O
    from rad_mmm_amd.decoders import RADMMMFlow
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    CFG = bench.CONFIGS["joint" if args.joint else "radtts"]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    B, T = 32, 800
    gb = {k: torch.from_numpy(v).to(dev) for k, v in O.synthetic_batch(B, T, cfg, seed=1234, ragged=False).items()}
    print(json.dumps(bench.full_step_leg(dec, cfg, CFG, gb, B, T, dev, 0.0, steps=args.steps, joint=args.joint)))


if __name__ == "__main__":
    main()
