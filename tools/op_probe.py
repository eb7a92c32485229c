"""List the stock ATen ops of one training step grouped by (name, input shapes): finds the small
launch-bound stragglers between the HIP kernels.  Usage: python tools/op_probe.py [filter]"""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
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
gb = {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(32, 800, cfg, seed=1234, ragged=False).items()}
sl = SequenceLength(gb["lengths"]); red = BucketedGradReducer(dec)
def step():
    red.prepare()
    out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
    loss = crit(out, None, sl, 0)["loss_mel"][0]
    loss.backward(); red.finish(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
flt = sys.argv[1:] or [""]
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::"): continue
    if not any(f in e.name for f in flt): continue
    st = [s for s in (e.stack or []) if "rad_mmm_amd" in s or "bench" in s or "op_probe" in s]
    key = (e.name, str(e.input_shapes)[:80], st[0][-70:] if st else "<autograd engine>")
    r = rows.setdefault(key, [0, 0.0]); r[0] += 1; r[1] += e.device_time
for k, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n:5d} {us:9.1f}us  {k[0]:24s} {k[1]:80s} {k[2]}")
tot = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        r = tot.setdefault(e.name[:60], [0, 0.0]); r[0] += 1; r[1] += e.device_time
stock = [(k, v) for k, v in tot.items() if "radmmm" not in k and "GLOBAL__N" not in k]
print("stock device kernels: %.1f us in %d launches" % (sum(v[1] for _, v in stock), sum(v[0] for _, v in stock)))
for k, v in sorted(stock, key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[0]:5d} {v[1]:9.1f}us  {k}")
