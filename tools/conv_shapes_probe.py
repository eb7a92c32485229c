#!/usr/bin/env python3
"""Which conv_norm calls does one full training step make outside the decoder's flows, with which shapes and on which GEMM
path, and what does each cost?  (HIP events around every call of ops.conv_norm in the forward; the backward's share shows
in the kernel trace.)  RADMMM_DEBUG=1 RADMMM_CONVNORM_H3_MIN_ROWS=<n> moves the text-rate convs to the split-f16 path."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from rad_mmm_amd import ops
    import radmmm_synth as O
    from rad_mmm_amd.data import BetaBinomialInterpolator
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    dev = torch.device("cuda:0")
    CFG = bench.CONFIGS["radtts"]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    B, T, t_txt = 32, 800, 150
    gb = {k: torch.from_numpy(v).to(dev) for k, v in O.synthetic_batch(B, T, cfg, seed=1234, ragged=False).items()}
    torch.manual_seed(1234)
    model = TTSTrainingStep(Encoder(3, CFG["n_text_dim"], 5), dec, RADMMMLoss(sigma=1.0, kl_loss_start_iter=0), n_speakers=8,
                            n_accents=4, n_text_tokens=185, n_text_dim=CFG["n_text_dim"], n_speaker_dim=CFG["n_speaker_dim"],
                            n_accent_dim=CFG["n_accent_dim"], use_accent=True,
                            use_accent_emb_for_decoder=CFG["use_accent_emb_for_decoder"], binarization_start_iter=0).to(dev).train()
    g = torch.Generator().manual_seed(99)
    in_lens = [t_txt] * B
    batch = {"mel": gb["mel"] * 2 - 5, "speaker_ids": torch.randint(0, 8, (B,), generator=g).to(dev),
             "accent_ids": torch.randint(0, 4, (B,), generator=g).to(dev), "text": torch.randint(0, 185, (B, t_txt), generator=g).to(dev),
             "input_lengths": torch.tensor(in_lens, device=dev), "output_lengths": gb["lengths"],
             "input_lengths_host": torch.tensor(in_lens), "output_lengths_host": gb["lengths"].cpu(),
             "attn_prior": BetaBinomialInterpolator(device=dev).batch(in_lens, [T] * B), "f0": gb["f0"], "energy_avg": gb["energy"]}

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _, _ = model.training_step(batch, global_step=10)
        loss.backward()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    orig = ops.conv_norm
    log = []

    def traced(x, v, g_, bias, lens, B_, T_, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, v, g_, bias, lens, B_, T_, **kw)
        e1.record()
        log.append((tuple(x.shape), tuple(v.shape), type(y.grad_fn).__name__ if y.grad_fn is not None else "-", e0, e1))
        return y
    ops.conv_norm = traced
    import rad_mmm_amd.encoder as E, rad_mmm_amd.attention as A
    loss = step()
    torch.cuda.synchronize()
    ops.conv_norm = orig
    print(f"loss {float(loss):.6f}")
    for xs, vs, fn, e0, e1 in log:
        print(f"  rows x ld {xs}  weight {vs}  {fn:28s} forward {e0.elapsed_time(e1) * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
