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
    import radmmm_synth as O
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    dev = torch.device("cuda:0")
    name = sys.argv[1] if len(sys.argv) > 1 else "radtts"          # radtts | radmmm | radmmm_splines [frames]
    CFG = bench.CONFIGS[name]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    B, T = 32, int(sys.argv[2]) if len(sys.argv) > 2 else 800
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    rows = prof.key_averages(group_by_stack_n=8)
    out = []
    for r in rows:
        if not r.key.startswith("aten::") or r.device_time_total <= 0 or r.self_device_time_total <= 0:
            continue
        where = "?"
        for fr in r.stack or []:
            if "rad_mmm_amd" in fr or "aten_ops_probe" in fr:
                where = fr.split("rad_mmm_amd/")[-1].split("/root/repo/")[-1][:80]
                break
        out.append((r.self_device_time_total, r.count, r.key, where))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print(f"stock torch kernels in one decoder step: {sum(o[1] for o in out)} calls, {tot / 1e3:.2f} ms of device time")
    for dt, n, name, where in out[:50]:
        print(f"  {name:24s} {n:4d} x  {dt / 1e3:7.3f} ms   {where}")


if __name__ == "__main__":
    main()
