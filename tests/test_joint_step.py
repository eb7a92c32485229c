"""BASELINE configs[3] as ONE step at its shipped size: the RADMMM decoder (8 flows, D = 1056) + text encoder
+ alignment attention + on-device MAS + the f0 / energy / voiced / duration predictors (ConvLSTMLinearDAP at the dims of
configs/RADMMM_{f0,energy,vpred,duration}model_config.yaml), B = 32, T = 800, 150 tokens -- the model and batch that
`python bench.py --config joint` times -- against the CPU oracle's restatement of TTSModel.training_step
(tts_lightning_modules.py:643-750; oracle.tts_joint_step) on the WHOLE batch, dropout off on both sides."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_joint_step_at_the_shipped_size_matches_the_oracle(monkeypatch):
    import bench
    import radmmm_synth as S
    from rad_mmm_amd.decoders import RADMMMFlow
    # the reference takes numpy's float32 log on the host in front of the alignment search (alignment.py:36); with the same
    # log the device search is bit-exact, so every hard alignment must be identical (INTEGRATION.md "MAS: which log")
    monkeypatch.setenv("RADMMM_MAS_LOG", "host")
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    dev = torch.device("cuda:0")
    B, T = 32, 800
    CFG = bench.CONFIGS["joint"]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    gb = {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(B, T, cfg, seed=1234, ragged=True).items()}
    model = bench.build_step_model(dec, CFG, dev, joint=True)
    batch = bench.build_step_batch(gb, B, T, dev, joint=True)
    rep, worst = bench.joint_parity_vs_cpu(model, batch, cfg, B, dev)
    print({k: rep[k] for k in ("summed_loss_hip", "summed_loss_cpu", "summed_loss_rel_diff", "z_rel_err_vs_cpu",
                               "predictor_output_rel_err_vs_cpu", "hard_alignments_identical", "oracle_seconds")})
    print({k: v["rel_diff"] for k, v in rep["loss_terms"].items()})
    assert worst["alignments_identical"] == B
    assert worst["z"] < 1e-4 and worst["loss"] < 1e-4
    assert worst["terms"] < 1e-4, rep["loss_terms"]
    assert worst["pred"] < 1e-4, rep["predictor_output_rel_err_vs_cpu"]
    # ... and the whole thing trains: one real step (dropout on) through the bucket reducer, every predictor gets gradients
    from rad_mmm_amd.ddp import BucketedGradReducer
    red = BucketedGradReducer(model)
    red.prepare()
    loss, losses, _ = model.training_step(batch, global_step=10)
    loss.backward()
    red.finish()
    assert {"f0_loss", "energy_loss", "vpred_loss", "duration_loss", "loss_mel", "loss_ctc", "binarization_loss"} <= set(losses)
    for name in bench.JOINT_PREDICTORS:
        gs = [p.grad for n, p in model.named_parameters() if n.startswith(f"{name}_predictor.")]
        assert gs and all(g is not None and torch.isfinite(g).all() for g in gs) and sum(float(g.abs().max()) > 0 for g in gs) >= len(gs) // 2
    keys = [b["key"] for b in red.buckets]
    assert sum(k.startswith("decoder.flows.") for k in keys) == 16 and "misc" in keys          # two buckets per flow step + the rest
