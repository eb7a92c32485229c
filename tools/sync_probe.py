#!/usr/bin/env python3
"""Where does one full training step (bench.py's full-step leg) synchronise with the host?  Runs the step under torch's sync
debug mode and prints every warning with the innermost rad_mmm_amd / bench frame that caused it."""
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    import radmmm_synth as O
    from rad_mmm_amd.data import BetaBinomialInterpolator
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
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
    reducer = BucketedGradReducer(model)
    opt = FlatRAdam(model.named_parameters(), lr=1e-6, weight_decay=1e-6, reducer=reducer)

    def step():
        reducer.prepare()
        loss, _, _ = model.training_step(batch, global_step=10)
        loss.backward()
        reducer.finish()
        opt.clip_grad_norm(1.0)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    hits = []
    orig = warnings.showwarning

    def show(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" in str(message).lower():
            fr = [f for f in traceback.extract_stack() if ("rad_mmm_amd" in f.filename or f.filename.endswith("sync_probe.py"))]
            hits.append(" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:])))
    warnings.showwarning = show
    torch.cuda.set_sync_debug_mode("warn")
    warnings.simplefilter("always")
    step()
    torch.cuda.set_sync_debug_mode("default")
    warnings.showwarning = orig
    print(f"{len(hits)} host synchronisations in one step:")
    for h in hits:
        print("  ", h)


if __name__ == "__main__":
    main()
