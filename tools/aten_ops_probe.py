#!/usr/bin/env python3
"""Which stock torch (aten) kernels does one decoder training step launch, how often, for how long, and from which line of this
package?  torch.profiler over one step (after warm-up), grouped by (aten op, innermost rad_mmm_amd frame)."""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from rad_mmm_amd import synthetic as O
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    dev = torch.device("cuda:0")
    CFG = bench.CONFIGS["radtts"]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    B, T = 32, 800
    gb = {k: torch.from_numpy(v).to(dev) for k, v in O.synthetic_batch(B, T, cfg, seed=1234, ragged=False).items()}
    sl = SequenceLength(gb["lengths"])
    crit = RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)
    red = BucketedGradReducer(dec)

    def step():
        red.prepare()
        out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        red.finish()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        dt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
        if not ev.kernels:
            continue
        where = "?"
        for fr in ev.stack or []:
            if "rad_mmm_amd" in fr or "bench.py" in fr or "aten_ops_probe" in fr:
                where = fr.split("rad_mmm_amd/")[-1][:70]
                break
        k = (ev.name, where)
        agg[k][0] += 1
        agg[k][1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"stock torch kernels in one decoder step: {sum(v[0] for _, v in rows)} launches, {tot / 1e3:.2f} ms of device time")
    for (name, where), (n, dt) in rows[:45]:
        print(f"  {name:28s} {n:4d} x  {dt / 1e3:7.3f} ms   {where}")


if __name__ == "__main__":
    main()
