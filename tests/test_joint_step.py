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


def test_joint_step_backward_at_the_shipped_size_matches_the_oracles_autograd(monkeypatch):
    """the BACKWARD of the joint step at the shipped model size.  Twelve utterances of the bench batch
    (T = 800, 150 tokens; the step's terms are means over utterances, so twelve are the step at B = 12), dropout off on both
    sides: every parameter gradient of the HIP step -- decoder, text encoder, attention, embeddings, the four predictors --
    against the autograd of the oracle's restatement (oracle.tts_joint_step, pinned to the reference's components by
    tests/test_oracle_joint.py), as the relative L2 error of the tensor AND the relative difference of its norm, both held to
    5e-4.  Kink accounting as in tests/test_attribute_predictors.py: the predictors' (leaky) ReLU decisions of the HIP run are
    imposed on the oracle; every differing decision must lie within rounding of 0 and is counted.  The predictors read
    DETACHED inputs (tts_lightning_modules.py:688-727): their loss terms must not reach the decoder / encoder -- checked by
    the oracle's gradients of those parameters being the same with and without the predictor terms, which the comparison of
    every tensor implies, and directly by the gradient of the summed predictor terms w.r.t. the context being absent."""
    import numpy as np
    import torch.nn.functional as F
    import bench
    import radmmm_synth as S
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import attribute_predictors as AP
    from rad_mmm_amd import ops as _ops
    from rad_mmm_amd.decoders import RADMMMFlow
    monkeypatch.setenv("RADMMM_MAS_LOG", "host")
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    dev = torch.device("cuda:0")
    B, T, Bs = 32, 800, 12      # (12 x 400 grouped frames = 4800 rows: the WN stack on the default FP8-cross kernels, >= 4096 rows)
    CFG = bench.CONFIGS["joint"]
    cfg, sd = bench.procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    gb = {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(B, T, cfg, seed=1234, ragged=True).items()}
    model = bench.build_step_model(dec, CFG, dev, joint=True)
    for name in bench.JOINT_PREDICTORS:                      # no further power iteration: both sides see the same u / v
        getattr(model, f"{name}_predictor").feat_pred_fn.bilstm.eval()
    full = bench.build_step_batch(gb, B, T, dev, joint=True)
    sub = {k: (v[:Bs] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in full.items()}
    # ---- the HIP step; the predictors' conv outputs in call order (bottleneck, conv 0 .. 2 per predictor)
    seen, cur, real_conv = {}, [None], _ops.conv_norm
    real_pre = AP.ConvLSTMLinearDAP.forward_pre

    def spy_conv(*a, **k):
        y = real_conv(*a, **k)
        if cur[0] is not None:
            seen.setdefault(cur[0], []).append(y.detach())
        return y

    def spy_pre(self, *a, **k):
        cur[0] = next(n for n in bench.JOINT_PREDICTORS if getattr(model, f"{n}_predictor") is self)
        try:
            return real_pre(self, *a, **k)
        finally:
            cur[0] = None
    monkeypatch.setattr(_ops, "conv_norm", spy_conv)
    monkeypatch.setattr(AP.ConvLSTMLinearDAP, "forward_pre", spy_pre)
    model.zero_grad(set_to_none=True)
    loss, losses, outs = model.training_step(sub, global_step=10)
    loss.backward()
    torch.cuda.synchronize()
    monkeypatch.setattr(_ops, "conv_norm", real_conv)
    monkeypatch.setattr(AP.ConvLSTMLinearDAP, "forward_pre", real_pre)
    assert all(len(seen[n]) == 4 for n in bench.JOINT_PREDICTORS), {n: len(v) for n, v in seen.items()}
    g_hip = {n: q.grad.detach().float().cpu() for n, q in model.named_parameters() if q.grad is not None}
    # ---- the oracle: the same weights (after the HIP forward: spectral norm's u / v as that forward used them)
    p = {n: (v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu()) for n, v in model.state_dict().items()}
    pnames = {n for n, _ in model.named_parameters()}
    for n in p:
        if n in pnames and p[n].is_floating_point():
            p[n].requires_grad_(True)
    cb = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sub.items()}
    specs = {name: dict(n_layers=3, weight=1.0, **spec) for name, spec in bench.JOINT_PREDICTORS.items()}
    rec = {}
    with torch.no_grad():
        O.tts_joint_step(p, cfg, cb, specs, binarize=True, bin_loss=True, dap_record=rec)
    lens_of = {n: (cb["input_lengths"] if n == "duration" else cb["output_lengths"]) for n in specs}
    gates, flipped, total, worst_pre = {}, 0, 0, 0.0
    for name in specs:
        lens = lens_of[name]
        Tn = rec[name]["bottleneck"].shape[2]

        def cl(y, C):                                        # channels-last rows -> [Bs, C, Tn]
            return y[:, :C].reshape(Bs, Tn, C).permute(0, 2, 1).cpu()
        gt = {"bottleneck": cl(seen[name][0], rec[name]["bottleneck"].shape[1]) > 0}
        for i in range(3):
            g_i = cl(seen[name][1 + i], rec[name][(0, i)].shape[1]) > 0
            for b in range(Bs):
                gt[(b, i)] = g_i[b: b + 1, :, : int(lens[b])]
        mask = (torch.arange(Tn)[None, :] < lens[:, None])[:, None]
        for key, gk in gt.items():
            pre = rec[name][key]
            valid = mask.expand_as(pre) if key == "bottleneck" else torch.ones_like(pre, dtype=torch.bool)
            diff = ((pre > 0) != gk) & valid
            total += int(valid.sum())
            if diff.any():
                flipped += int(diff.sum())
                bound = 2e-5 * float(pre[valid].pow(2).mean().sqrt())
                worst_pre = max(worst_pre, float(pre[diff].abs().max()) / bound)
        gates[name] = gt
    print(f"joint step, {Bs} utterances: {flipped} of {total} predictor pre-activations on the other side of the kink; the worst "
          f"one lies at {worst_pre:.2f} x the rounding bound (2e-5 rms)")
    assert worst_pre <= 1.0 and flipped <= max(20, total // 100000)
    ref = O.tts_joint_step(p, cfg, cb, specs, binarize=True, bin_loss=True, dap_gates=gates)
    ref["loss"].backward()
    assert abs(float(loss) - float(ref["loss"])) <= 1e-4 * abs(float(ref["loss"]))
    rows, bad, zeros = [], {}, []
    gmax = {}
    for n in g_hip:
        if p[n].grad is not None:
            k = n.split(".")[0]
            gmax[k] = max(gmax.get(k, 0.0), float(p[n].grad.norm()))
    for n in sorted(g_hip):
        gr = p[n].grad
        if gr is None:
            continue
        nr = float(gr.norm())
        if nr < 1e-5 * gmax[n.split(".")[0]]:
            # analytically zero -- the scale and the bias of a conv in front of an instance norm (text encoder): both sides hold
            # rounding residue 1e-6 and less of the module's other gradients, not comparable in relative terms
            assert float(g_hip[n].norm()) <= 1e-4 * gmax[n.split(".")[0]], (n, float(g_hip[n].norm()))
            zeros.append(n)
            continue
        l2 = float((g_hip[n] - gr).norm()) / nr
        dn = abs(float(g_hip[n].norm()) - nr) / nr
        rows.append((l2, dn, n))
        if not (l2 <= 5e-4 and dn <= 5e-4):
            bad[n] = (l2, dn)
    rows.sort(reverse=True)
    print(f"{len(rows)} parameter gradients compared; worst relative L2 errors:")
    for l2, dn, n in rows[:10]:
        print(f"   {l2:.2e} (norm {dn:.2e})  |g_cpu| {float(p[n].grad.norm()):.3e}  {n}")
    print(f"{len(zeros)} analytically zero gradients (scale / bias in front of an instance norm) held to 1e-4 of their module's largest: {zeros}")
    groups = {}
    for l2, dn, n in rows:
        k = n.split(".")[0]
        groups[k] = max(groups.get(k, 0.0), l2)
    print({k: f"{v:.1e}" for k, v in groups.items()})
    assert len(rows) >= 300 and len(zeros) <= 8 and not bad, bad
    # predictors have parameters with gradients, and all four were compared
    assert all(any(n.startswith(f"{name}_predictor.") for _, _, n in rows) for name in specs)
