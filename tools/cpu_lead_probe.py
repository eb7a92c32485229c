#!/usr/bin/env python3
"""Is the training step bound by the GPU or by the host issuing it?  Runs bench.py's full-step leg (and the decoder-only step)
without any synchronisation inside the loop and prints, per step, when the host RETURNED from step() and when the GPU
FINISHED it (HIP event), both relative to the start of the loop: the lead the host has over the device.  A lead near zero
means the device waits for launches."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(step, n=8, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    cpu = []
    ev[0].record()
    t0 = time.perf_counter()
    for i in range(n):
        step()
        ev[i + 1].record()
        cpu.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    gpu = [ev[0].elapsed_time(ev[i + 1]) for i in range(n)]
    for i in range(n):
        print(f"  step {i}: host returned at {cpu[i]:7.1f} ms, device finished at {gpu[i]:7.1f} ms  (lead {gpu[i] - cpu[i]:6.1f} ms)")
    print(f"  host {cpu[-1] / n:.1f} ms/step, device {gpu[-1] / n:.1f} ms/step")


def main():
    import bench
    import radmmm_synth as O
    from rad_mmm_amd.common import SequenceLength
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
    sl = SequenceLength(gb["lengths"])
    crit = RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)
    red = BucketedGradReducer(dec)

    def dec_step():
        red.prepare()
        out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        red.finish()
    print("decoder forward + backward:")
    run(dec_step)
    red.detach()
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
    print("full training step:")
    run(step)
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
