import sys, time, os, torch, numpy as np
sys.path.insert(0, '.')
import radmmm_synth as S
from rad_mmm_amd.common import SequenceLength
from rad_mmm_amd.decoders import RADMMMFlow
from rad_mmm_amd.loss import RADMMMLoss
from rad_mmm_amd.ddp import BucketedGradReducer
import bench
CFG = bench.CONFIGS["radtts"]
cfg, sd = bench.procedural_state(CFG)
dev = "cuda:0"
dec = RADMMMFlow(use_accent=True, **CFG); dec.load_state_dict(sd); dec = dec.to(dev).train()
crit = RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)
gb = {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(32, 800, cfg, seed=1234, ragged=("--ragged" in sys.argv)).items()}
sl = SequenceLength(gb["lengths"]); red = BucketedGradReducer(dec)
def step():
    red.prepare()
    out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
    loss = crit(out, None, sl, 0)["loss_mel"][0]
    loss.backward(); red.finish(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, until idle {1e3*(t2-t0):.1f} ms")
