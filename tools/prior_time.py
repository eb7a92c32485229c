import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from rad_mmm_amd import data as D
from oracle import radmmm_oracle as O
rng = np.random.default_rng(3)
out_lens = [800] + list(rng.integers(200, 800, 31)); in_lens = [150] + list(rng.integers(20, 150, 31))
it = D.BetaBinomialInterpolator()
t0 = time.perf_counter(); it.batch(in_lens, out_lens); torch.cuda.synchronize(); print("first batch (builds anchors) %.2f ms" % (1e3*(time.perf_counter()-t0)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(20): it.batch(in_lens, out_lens)
e1.record(); torch.cuda.synchronize()
print("steady batch: %.3f ms device, %.3f ms wall" % (e0.elapsed_time(e1)/20, 1e3*(time.perf_counter()-t0)/20))
t0 = time.perf_counter(); O.attention_prior_batch(in_lens, out_lens); print("oracle (numpy/scipy.special, 1 core, no cache) %.1f ms" % (1e3*(time.perf_counter()-t0)))
mel = torch.randn(32, 80, 800, device="cuda")
for _ in range(3): D.get_energy_average(mel)
e0.record()
for _ in range(20): D.get_energy_average(mel)
e1.record(); torch.cuda.synchronize(); print("energy average [32,80,800]: %.1f us" % (1e3*e0.elapsed_time(e1)/20))
