"""Attribute predictor.  CPU: oracle restatement vs vectors captured from the reference's
ConvLSTMLinearDAP + AttributeRegressionLoss (eval mode).  GPU: the batched HIP predictor vs the same."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

HERE = os.path.dirname(__file__)


def _load():
    g = np.load(os.path.join(HERE, "golden", "dap_tiny.npz"))
    return {k: torch.from_numpy(np.asarray(g[k])) for k in g.files}


def test_oracle_dap_matches_reference():
    from oracle import radmmm_oracle as O
    g = _load()
    p = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        x = O.dap_tx_data(g["target"], 2.0, 0.5, True)
        x_hat = O.dap_forward(p, "", g["txt"], g["spk"], g["lens"], 2)
        loss = O.attribute_regression_loss(x_hat, x, g["lens"])
    assert rel_err(x, g["x"]) < 1e-6 and rel_err(x_hat, g["x_hat"]) < 3e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


@pytest.mark.gpu
def test_hip_dap_matches_reference():
    from rad_mmm_amd.attribute_predictors import AttributeRegressionLoss, ConvLSTMLinearDAP
    from rad_mmm_amd.common import SequenceLength
    g = _load()
    dev = "cuda:0"
    dap = ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2, n_hidden=16,
                            kernel_size=3, p_dropout=0.25, target_scale=2.0, target_offset=0.5, log_target=True)
    dap.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")})
    dap = dap.to(dev).eval()
    txt = g["txt"].to(dev).requires_grad_(True)
    sl = SequenceLength(g["lens"].to(dev))
    out = dap(g["target"].to(dev), txt, g["spk"].to(dev), sl)
    assert rel_err(out["x_hat"].detach().cpu(), g["x_hat"]) < 1e-4
    loss = AttributeRegressionLoss("f0_", 1.0)(out, None, sl, 0)["f0_loss"][0]
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    assert rel_err(txt.grad.cpu(), g["grad.txt"]) < 2e-4
    scale = max(float(g[k].abs().max()) for k in g if k.startswith("gradp."))
    for n, p in dap.named_parameters():
        ref = g["gradp." + n]
        assert rel_err(p.grad.cpu(), ref) < 3e-4 or float((p.grad.cpu() - ref).abs().max()) < 1e-4 * scale, n


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,k", [(3, 70, 15), (32, 400, 15), (32, 400, 5)])
def test_hip_dap_at_the_shipped_config_sizes(B, T, k, monkeypatch):
    """The f0 / voiced predictor shapes of configs/RAD{TTS,MMM}_f0model_config.yaml (in_dim 512, reduction 16, 3 backbone
    layers of 256 channels, kernel 15 resp. 5, accent embedding on), small batch (fp32 conv path) and the full
    32 x 400 batch (split-f16 conv path with 15 taps): forward and parameter gradients against the oracle."""
    from oracle import radmmm_oracle as O
    from rad_mmm_amd.attribute_predictors import ConvLSTMLinearDAP
    from rad_mmm_amd.common import SequenceLength
    dev = "cuda:0"
    torch.manual_seed(B + k)
    dap = ConvLSTMLinearDAP(n_speaker_dim=16, n_accent_dim=8, in_dim=512, out_dim=1, reduction_factor=16, n_backbone_layers=3,
                            n_hidden=256, kernel_size=k, p_dropout=0.0, target_offset=-5, use_accent_embedding=True)
    dap = dap.to(dev).train()
    for _ in range(20):          # converge spectral norm's power iteration: fresh u / v leave |W_hh| ~ 10, a chaotic recurrence
        for hook in dap.feat_pred_fn.bilstm._forward_pre_hooks.values():
            hook(dap.feat_pred_fn.bilstm, ())
    dap = dap.eval()
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    txt = torch.randn(B, 512, T, generator=g) * 0.5
    spk, acc = torch.randn(B, 16, generator=g), torch.randn(B, 8, generator=g)
    gy = torch.randn(B, 1, T, generator=g)
    sl = SequenceLength(lens.to(dev))
    # the HIP module's activation outputs, in call order (bottleneck, conv 0 .. 2): y > 0 is the side of the (leaky) ReLU's
    # kink its pre-activation fell on
    from rad_mmm_amd import ops as _ops
    seen, real = [], _ops.conv_norm

    def spy(*a, **k):
        y = real(*a, **k)
        seen.append(y.detach())
        return y
    monkeypatch.setattr(_ops, "conv_norm", spy)
    out = dap(None, txt.to(dev), spk.to(dev), sl, accent_emb=acc.to(dev))["x_hat"]
    monkeypatch.setattr(_ops, "conv_norm", real)
    mask = (torch.arange(T)[None, :] < lens[:, None])[:, None]
    ((out * (gy * mask).to(dev)).sum()).backward()
    assert len(seen) == 4
    p = {n: v.detach().cpu().clone() for n, v in dap.state_dict().items()}
    for n in p:
        if p[n].dtype == torch.float32 and not n.endswith(("_u", "_v")) and p[n].dim() > 0:
            p[n].requires_grad_(True)
    spk_acc = torch.cat((spk, acc), 1)                                          # the module appends accent after speaker
    rec = {}
    with torch.no_grad():
        ref0 = O.dap_forward(p, "", txt, spk_acc, lens, 3, record=rec)
    # ---- accounting of the kink: relu' / leaky' are discontinuous at 0.  Every element where the two
    # implementations sit on different sides must have an oracle pre-activation within rounding of 0; the oracle's gradient
    # is then taken WITH THE HIP MODULE'S decisions, and every parameter gradient is held to the common 5e-4.
    def cl(y, C):                                                               # channels-last rows -> [B, C, T]
        return y[:, :C].reshape(B, T, C).permute(0, 2, 1).cpu()
    gates = {"bottleneck": cl(seen[0], rec["bottleneck"].shape[1]) > 0}
    for i in range(3):
        g_i = cl(seen[1 + i], rec[(0, i)].shape[1]) > 0
        for b in range(B):
            gates[(b, i)] = g_i[b: b + 1, :, : int(lens[b])]
    flipped, worst_pre, total = 0, 0.0, 0
    for key, gt in gates.items():
        pre = rec[key]
        valid = mask.expand_as(pre) if key == "bottleneck" else torch.ones_like(pre, dtype=torch.bool)
        diff = ((pre > 0) != gt) & valid
        total += int(valid.sum())
        if diff.any():
            flipped += int(diff.sum())
            # within rounding of zero: a conv output is a sum of K = Cin * k terms of typical size rms(pre) / sqrt(K)
            bound = 2e-5 * float(pre[valid].pow(2).mean().sqrt())
            worst_pre = max(worst_pre, float(pre[diff].abs().max()) / bound)
    print(f"DAP B={B} T={T} k={k}: {flipped} of {total} pre-activations on the other side of the kink; the worst one lies at "
          f"{worst_pre:.2f} x the rounding bound (2e-5 rms)")
    assert worst_pre <= 1.0 and flipped <= max(20, total // 100000)
    ref = O.dap_forward(p, "", txt, spk_acc, lens, 3, gates=gates)
    ((ref * gy[:, :, : ref.shape[2]] * mask[:, :, : ref.shape[2]]).sum()).backward()
    o = out.detach().cpu()[:, :, : ref.shape[2]]
    assert rel_err(o[mask[:, :, : ref.shape[2]]], ref0[mask[:, :, : ref.shape[2]]]) < 1e-4      # (the un-gated oracle)
    errs = {n: rel_err(q.grad.cpu(), p[n].grad) for n, q in dap.named_parameters() if n in p and p[n].grad is not None}
    bad = {n: e for n, e in errs.items() if not e < 5e-4}
    assert not bad, (bad, errs)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(4, 96), (32, 400)])
def test_merged_predictor_lstms_match_separate_calls(B, T, monkeypatch):
    """attribute_predictors.dap_forward_many runs the bi-LSTMs of the f0 / energy / voiced predictors (one shape, same frames) as
    ONE block-diagonal recurrence (lstm.MergedBiLSTMFn): outputs and every parameter gradient must equal the three separate
    launches' (tts_lightning_modules.py:688-717 calls them one after the other) up to the recurrence's fp32 summation order."""
    from rad_mmm_amd.attribute_predictors import ConvLSTMLinearDAP, dap_forward_many
    from rad_mmm_amd.common import SequenceLength
    dev = "cuda:0"
    torch.manual_seed(3)
    daps = [ConvLSTMLinearDAP(n_speaker_dim=16, n_accent_dim=8, in_dim=64, out_dim=1, reduction_factor=4, n_backbone_layers=2,
                              n_hidden=256, kernel_size=5, p_dropout=0.0, use_accent_embedding=True).to(dev).train() for _ in range(3)]
    for d in daps:
        for _ in range(20):
            for hook in d.feat_pred_fn.bilstm._forward_pre_hooks.values():
                hook(d.feat_pred_fn.bilstm, ())
        d.eval()                                             # (no further power iteration: both passes see the same weights)
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    txt = (torch.randn(B, 64, T, generator=g) * 0.5).to(dev)
    spk, acc = torch.randn(B, 16, generator=g).to(dev), torch.randn(B, 8, generator=g).to(dev)
    gys = [torch.randn(B, 1, T, generator=g).to(dev) for _ in daps]
    sl = SequenceLength(lens.to(dev))
    mask = (torch.arange(T)[None, :] < lens[:, None])[:, None].to(dev)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RADMMM_MERGE_DAP_LSTM", mode)
        for d in daps:
            d.zero_grad(set_to_none=True)
        outs = dap_forward_many(daps, [((None, txt, spk, sl), {"accent_emb": acc}) for _ in daps])
        sum(((o["x_hat"] * gy * mask).sum() for o, gy in zip(outs, gys))).backward()
        res[mode] = ([o["x_hat"].detach().clone() for o in outs],
                     [{n: p.grad.detach().clone() for n, p in d.named_parameters() if p.grad is not None} for d in daps])
    for a, b in zip(*[res[m][0] for m in ("0", "1")]):
        assert rel_err((a * mask).cpu(), (b * mask).cpu()) < 2e-5
    worst = 0.0
    for ga, gb in zip(*[res[m][1] for m in ("0", "1")]):
        assert ga.keys() == gb.keys() and len(ga) >= 10
        for n in ga:
            e = rel_err(gb[n].cpu(), ga[n].cpu())
            worst = max(worst, e)
            assert e < 2e-4, (n, e)
    print(f"merged vs separate predictor LSTMs (B={B}, T={T}): worst parameter-gradient rel err {worst:.2e}")
