#!/usr/bin/env python3
"""How many threads does torch-CPU use by default on this box, and what does the CPU oracle cost with them vs with the
threads the process may really use (affinity capped by the cgroup quota)?  (round 4: the GPU test suite's oracle runs)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import host_threads  # noqa: E402
from oracle import radmmm_oracle as O  # noqa: E402


def one(p, cfg, B, T):
    for v in p.values():
        if v.requires_grad:
            v.grad = None
    b = {k: torch.from_numpy(v) for k, v in O.synthetic_batch(B, T, cfg, 4321, False).items()}
    t0 = time.perf_counter()
    out = O.decoder_forward(p, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"])
    lm, _ = O.decoder_loss(out, b["lengths"], cfg.n_group_size)
    lm.backward()
    return time.perf_counter() - t0


def main():
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
              scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True, n_conv_layers_per_step=4, n_flows=8)
    cfg = O.DecoderConfig(**kw)
    sd = {k: torch.from_numpy(v) if not torch.is_tensor(v) else v for k, v in O.procedural_decoder_state(O.decoder_state_shapes(cfg)).items()}
    p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0
             and not k.endswith((".p", "lower_diag", "input_mean")) else v) for k, v in sd.items()}
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        quota = "n/a"
    print(f"os.cpu_count {os.cpu_count()}  affinity {len(os.sched_getaffinity(0))}  cgroup cpu.max {quota}  "
          f"torch default threads {torch.get_num_threads()}  host_threads() {host_threads()}", flush=True)
    one(p, cfg, 1, 100)
    print(f"default threads ({torch.get_num_threads()}): B=4,T=800 fwd+bwd {one(p, cfg, 4, 800):.2f} s", flush=True)
    torch.set_num_threads(host_threads())
    one(p, cfg, 1, 100)
    print(f"host_threads ({torch.get_num_threads()}): B=4,T=800 fwd+bwd {one(p, cfg, 4, 800):.2f} s", flush=True)


if __name__ == "__main__":
    main()
