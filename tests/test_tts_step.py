"""The training step's caller: rad_mmm_amd.tts_step.TTSTrainingStep against loss values,
alignments, contexts and gradient norms captured by running the reference's components (Encoder,
ConvAttention, mas_width1, RADMMMFlow, RADMMMLoss) under TTSModel.training_step's glue
(tests/golden/make_golden.py, section "tts_step"), with soft and with MAS-binarized alignments."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


@pytest.mark.parametrize("tag,step", [("soft", 0), ("hard", 10)])
def test_training_step_matches_reference_components(tag, step, monkeypatch):
    import radmmm_synth as S
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10)
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion")]
    shapes = {n: tuple(model.state_dict()[n].shape) for n in names}
    proc = S.procedural_decoder_state(shapes)
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)      # fixture ran without dropout
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    loss, losses, out = model.training_step(batch, global_step=step)
    loss.backward()
    assert rel_err(out["attn"].detach().cpu(), torch.from_numpy(g[f"{tag}.attn"])) < (1e-4 if tag == "soft" else 1e-12)
    assert rel_err(out["context"].detach().cpu(), torch.from_numpy(g[f"{tag}.context"])) < 1e-4
    for k, (v, w) in losses.items():
        ref = float(g[f"{tag}.{k}"])
        assert abs(float(v) - ref) <= 1e-4 * max(abs(ref), 1e-3), (k, float(v), ref)
    assert abs(float(loss) - float(g[f"{tag}.loss"])) < 1e-4 * abs(float(g[f"{tag}.loss"]))
    worst = (0.0, "")
    for k in g.files:
        if k.startswith(f"{tag}.gradnorm."):
            n = k[len(tag) + 10:]
            p = dict(model.named_parameters())[n]
            ref = float(g[k])
            if ref > 1e-6:
                e = abs(float(p.grad.norm()) - ref) / ref
                worst = max(worst, (e, n))
                assert e < 5e-4, (n, float(p.grad.norm()), ref)          # (measured: 3e-6)
    print(f"tts_step[{tag}]: worst gradient-norm difference {worst[0]:.2e} ({worst[1]})")


def test_joint_step_with_attribute_predictors_through_the_bucket_reducer(monkeypatch):
    """BASELINE configs[3]: decoder + f0 / energy / duration predictors in one step, every module's gradients in the
    reducer's flat buckets (what the N-GPU run all-reduces).  The step's own pieces are pinned elsewhere
    (test above, tests/test_attribute_predictors.py); here: the joint loss is the weighted sum of all parts, the
    predictors receive gradients and do not leak any into the decoder / encoder (detached inputs,
    tts_lightning_modules.py:300-369), and the bucketed gradients equal a plain backward bit for bit."""
    import radmmm_synth as S
    from rad_mmm_amd.attribute_predictors import AttributeRegressionLoss, ConvLSTMLinearDAP
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)

    def build(with_predictors):
        extra = {}
        if with_predictors:
            mk = lambda: ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2,
                                           n_hidden=32, kernel_size=3, p_dropout=0.0)
            extra = dict(f0_predictor=mk(), f0_predictor_loss=AttributeRegressionLoss("f0_", 1.0),
                         energy_predictor=mk(), energy_predictor_loss=AttributeRegressionLoss("energy_", 0.5),
                         duration_predictor=mk(), duration_predictor_loss=AttributeRegressionLoss("duration_", 0.25))
        torch.manual_seed(11)                      # predictors: torch init (no golden for them here), same in every build
        m = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10, **extra)
        names = [n for n in m.state_dict() if not n.startswith("decoder_criterion") and "_predictor" not in n]
        proc = S.procedural_decoder_state({n: tuple(m.state_dict()[n].shape) for n in names})
        m.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
        return m.to(dev).train()

    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    batch["voiced_mask"] = (batch["f0"] > 0).float()
    base = build(False)
    loss0, _, _ = base.training_step(batch, global_step=0)
    loss0.backward()

    grads = {}
    for mode in ("plain", "reducer"):
        model = build(True)
        red = BucketedGradReducer(model) if mode == "reducer" else None
        for it in range(2):
            if red is not None:
                red.prepare()
            else:
                model.zero_grad(set_to_none=True)
            loss, losses, _ = model.training_step(batch, global_step=0)
            loss.backward()
            if red is not None:
                red.finish()
        grads[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        assert {"f0_loss", "energy_loss", "duration_loss", "loss_mel"} <= set(losses)
        total = sum(float(v) * w for v, w in losses.values())
        assert abs(float(loss) - total) < 1e-5 * abs(total)
        part = sum(float(losses[k][0]) * losses[k][1] for k in ("f0_loss", "energy_loss", "duration_loss"))
        assert abs((float(loss) - part) - float(loss0)) < 1e-5 * abs(float(loss0))      # decoder part unchanged
        for pred in ("f0_predictor", "energy_predictor", "duration_predictor"):
            ps = [p for n, p in model.named_parameters() if n.startswith(pred + ".") and p.requires_grad]
            assert ps and all(p.grad is not None for p in ps)
            # (single tensors may be analytically zero: weight_g in front of a scale-invariant norm)
            assert sum(int(float(p.grad.abs().max()) > 0) for p in ps) >= len(ps) // 2, pred
        if red is not None:
            keys = [b["key"] for b in red.buckets]
            assert any(k.startswith("decoder.flows.") or k.startswith("flows.") for k in keys) and len(keys) >= 2, keys
            for n, p in model.named_parameters():
                if p.requires_grad:
                    assert p.grad.data_ptr() == red._views[id(p)].data_ptr(), n
    # predictor inputs are detached: decoder / encoder gradients are the predictor-free step's, bit for bit
    for n, p in base.named_parameters():
        if p.grad is not None:
            assert torch.equal(grads["plain"][n], p.grad), n
    for n in grads["plain"]:
        assert torch.equal(grads["plain"][n], grads["reducer"][n]), n


def test_joint_step_optimizer_update_matches_oracle(monkeypatch):
    """Two full training steps of the joint model (decoder + encoder + attention + an f0 predictor): bucket
    reducer -> global-norm clip 1.0 -> FlatRAdam on the reducer's flats, against the oracle's RAdam / clip applied
    tensor by tensor to the same gradients (direct-write parameters sit first in their buckets: the per-tensor
    mapping of the flat update is what this checks on the real parameter set)."""
    from oracle import radmmm_oracle as O
    import radmmm_synth as S
    from rad_mmm_amd.attribute_predictors import AttributeRegressionLoss, ConvLSTMLinearDAP
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    torch.manual_seed(5)
    pred = ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2, n_hidden=32,
                             kernel_size=3, p_dropout=0.0)
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10, f0_predictor=pred,
                            f0_predictor_loss=AttributeRegressionLoss("f0_", 1.0))
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion") and "_predictor" not in n]
    proc = S.procedural_decoder_state({n: tuple(model.state_dict()[n].shape) for n in names})
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    batch["voiced_mask"] = (batch["f0"] > 0).float()
    red = BucketedGradReducer(model)
    lr, wd = 1e-3, 1e-6
    opt = FlatRAdam(model.named_parameters(), lr=lr, weight_decay=wd, reducer=red)
    train = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    ref_p = {n: p.detach().clone() for n, p in train}
    ref_m = {n: torch.zeros_like(p) for n, p in train}
    ref_v = {n: torch.zeros_like(p) for n, p in train}
    for k in range(2):
        for n, p in train:                                   # both sides start the step from identical parameters
            assert torch.equal(p.detach(), ref_p[n]) or k > 0
        red.prepare()
        loss, _, _ = model.training_step(batch, global_step=0)
        loss.backward()
        red.finish()
        grads = {n: p.grad.detach().clone() for n, p in train}
        with torch.no_grad():                                # oracle continues from the HIP parameters of this step
            for n, p in train:
                ref_p[n] = p.detach().clone()
        total = opt.clip_grad_norm(1.0)
        opt.step()
        ref_total, clipped = O.clip_grad_norm([grads[n] for n, _ in train], 1.0)
        assert abs(float(total) - float(ref_total)) < 1e-5 * float(ref_total)
        for (n, p), gc in zip(train, clipped):
            O.radam_step(ref_p[n], gc, ref_m[n], ref_v[n], k + 1, lr=lr, weight_decay=wd)
            upd = (p.detach() - ref_p[n]).abs().max()
            scale = lr                                        # an RAdam step moves every element by <= ~lr
            assert float(upd) <= 2e-3 * scale + 1e-6 * float(p.detach().abs().max()), (k, n, float(upd))


def test_regularisation_and_bce_losses_match_reference_values():
    """rad_mmm_amd.loss.{VarianceCovarianceEmbeddingRegLoss, AttributeMinCrossCovarianceRegLoss, AttributeBCELoss}
    (configs/RADMMM_model_config.yaml:49-61, loss.py:213-347) against values captured from the reference."""
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd import loss as L
    g = np.load(os.path.join(HERE, "golden", "regloss.npz"))
    dev = "cuda:0"
    spk, acc = torch.nn.Embedding(7, 16).to(dev), torch.nn.Embedding(3, 8).to(dev)
    with torch.no_grad():
        spk.weight.copy_(torch.from_numpy(g["spk"]))
        acc.weight.copy_(torch.from_numpy(g["acc"]))
    sid, aid = torch.from_numpy(g["sid"]).to(dev), torch.from_numpy(g["aid"]).to(dev)
    close = lambda a, ref: abs(float(a) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-7
    d = L.VarianceCovarianceEmbeddingRegLoss("speaker", 0.3, 0.7, gamma=1.0)(spk)
    assert d["loss_speaker_variance"][1] == 0.3 and d["loss_speaker_covariance"][1] == 0.7
    assert close(d["loss_speaker_variance"][0], g["vc.variance"]) and close(d["loss_speaker_covariance"][0], g["vc.covariance"])
    d = L.VarianceCovarianceEmbeddingRegLoss("accent", 1.0, 1.0, gamma=2.0)(acc)
    assert close(d["loss_accent_variance"][0], g["vt.variance"]) and close(d["loss_accent_covariance"][0], g["vt.covariance"])
    cc = L.AttributeMinCrossCovarianceRegLoss("speaker", "accent", 1.0)
    assert close(cc(spk(sid), acc(aid), spk, acc)["loss_speaker-accent_cross_covariance"][0], g["cc.tables"])
    assert close(cc(spk(sid), acc(aid), None, None)["loss_speaker-accent_cross_covariance"][0], g["cc.batch"])
    lens = SequenceLength(torch.from_numpy(g["bce.lens"]).to(dev))
    out = {"x": torch.from_numpy(g["bce.x"]).to(dev), "x_hat": torch.from_numpy(g["bce.x_hat"]).to(dev)}
    assert close(L.AttributeBCELoss("vpred_", 1.0)(out, None, lens, 0)["vpred_loss"][0], g["bce.loss"])


def test_validation_step_equals_the_training_pass_without_gradients(monkeypatch):
    """validation_step (tts_lightning_modules.py:752-860) = the training pass with the criterion at step 100000,
    no gradients, plus the embedding regularisers of the model config."""
    import radmmm_synth as S
    from rad_mmm_amd import loss as L
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), L.RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10,
                            speaker_embed_regularization_loss=L.VarianceCovarianceEmbeddingRegLoss("speaker", 0.0, 0.0, 1.0),
                            speaker_accent_cross_regularization_loss=L.AttributeMinCrossCovarianceRegLoss("speaker", "accent", 1.0))
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion")]
    proc = S.procedural_decoder_state({n: tuple(model.state_dict()[n].shape) for n in names})
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    loss_t, losses_t, _ = model.training_step(batch, global_step=100000)
    assert model.binarize and {"loss_speaker_variance", "loss_speaker_covariance", "loss_speaker-accent_cross_covariance"} <= set(losses_t)
    # the golden "hard" case (step 10) has the same terms switched on: its decoder losses must reappear here
    for k in losses_t:
        if f"hard.{k}" in g.files:
            assert abs(float(losses_t[k][0]) - float(g[f"hard.{k}"])) <= 1e-4 * max(abs(float(g[f"hard.{k}"])), 1e-3), k
    model.eval()
    loss_v, losses_v, out_v = model.validation_step(batch)
    assert not loss_v.requires_grad and set(losses_v) == set(losses_t) and "txt_enc" in out_v
    for k in losses_t:
        assert abs(float(losses_v[k][0]) - float(losses_t[k][0])) <= 1e-5 * max(abs(float(losses_t[k][0])), 1e-3), k
        assert losses_v[k][1] == losses_t[k][1]
    assert abs(float(loss_v) - float(loss_t)) <= 1e-5 * abs(float(loss_t))


def test_training_step_under_autocast_runs_in_fp32_where_it_matters(monkeypatch):
    """Lightning `precision: bf16-mixed`: the caller's stock matmuls (context = txt_enc x attn) autocast to bf16 as
    they do in the reference; everything behind this package's modules stays fp32 (no bf16 tensor may reach the C
    ABI).  The joint loss must stay within bf16 rounding of the fp32 run and backward must work."""
    import radmmm_synth as S
    from rad_mmm_amd.attribute_predictors import AttributeRegressionLoss, ConvLSTMLinearDAP
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    torch.manual_seed(3)
    mk = lambda: ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2, n_hidden=32,
                                   kernel_size=3, p_dropout=0.0)
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10, f0_predictor=mk(),
                            f0_predictor_loss=AttributeRegressionLoss("f0_", 1.0), duration_predictor=mk(),
                            duration_predictor_loss=AttributeRegressionLoss("duration_", 1.0))
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion") and "_predictor" not in n]
    proc = S.procedural_decoder_state({n: tuple(model.state_dict()[n].shape) for n in names})
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    batch["voiced_mask"] = (batch["f0"] > 0).float()
    loss32, losses32, _ = model.training_step(batch, global_step=0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss16, losses16, out = model.training_step(batch, global_step=0)
    loss16.backward()
    assert out["z_mel"].dtype == torch.float32 and torch.isfinite(loss16)
    for k in losses32:
        a, b = float(losses32[k][0]), float(losses16[k][0])
        assert abs(a - b) <= 3e-2 * max(abs(a), 1e-2), (k, a, b)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_autocast_step_past_kl_start_has_a_live_binarisation_term(monkeypatch):
    """the binarisation loss is torch's BCE, which autocast bans on CUDA ("unsafe to autocast").  The autocast test
    above runs at global_step 0 where the term is the constant 0.0; this one runs with the term LIVE (global_step >
    kl_loss_start_iter, MAS binarisation on) under bf16 autocast: it must not raise, must equal the fp32 run's term within bf16
    rounding of the attention, and backward must give finite gradients."""
    import radmmm_synth as S
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import AttentionBinarizationLoss, RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    import torch.nn.functional as F
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=5),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=10)
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion")]
    proc = S.procedural_decoder_state({n: tuple(model.state_dict()[n].shape) for n in names})
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    loss32, losses32, _ = model.training_step(batch, global_step=10)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss16, losses16, out = model.training_step(batch, global_step=10)
    loss16.backward()
    b32, b16 = float(losses32["binarization_loss"][0]), float(losses16["binarization_loss"][0])
    assert b32 > 0 and abs(b32 - float(g["hard.binarization_loss"])) <= 1e-4 * b32       # the term is live and pinned
    assert abs(b16 - b32) <= 3e-2 * b32, (b16, b32)
    assert torch.isfinite(loss16) and all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # the module alone on a bf16 attention (what a caller under autocast may hand over): fp32 arithmetic inside, no raise
    hard = (torch.rand(2, 1, 12, 7, device=dev) > 0.8).float()
    soft = torch.rand(2, 1, 12, 7, device=dev).clamp_min(1e-3).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        v = AttentionBinarizationLoss()(hard, soft.bfloat16())
    ref = -(torch.log(soft.detach().bfloat16().float())[hard == 1]).mean()
    assert v.dtype == torch.float32 and abs(float(v) - float(ref)) <= 1e-5 * abs(float(ref))
    v.backward()
    assert torch.isfinite(soft.grad).all()


def test_training_step_with_host_lengths_never_synchronises():
    """Round 4: with the host copies of the lengths in the batch (`input_lengths_host` /
    `output_lengths_host`, what the collate function has before the batch moves to the device) a whole training step --
    text encoder, attention, on-device MAS (binarisation on), decoder, flow + CTC + binarisation losses, backward through the
    step-wide gradient reducer, global-norm clip, FlatRAdam -- makes NO blocking device -> host read: torch's sync debug mode
    is set to "error" around the third step.  Same loss as the step without the host copies (which reads the two length
    tensors once)."""
    import radmmm_synth as S
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
    from rad_mmm_amd.tts_step import TTSTrainingStep
    g = np.load(os.path.join(HERE, "golden", "tts_step.npz"))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("cfg.")}
    dev = "cuda:0"
    torch.manual_seed(3)
    model = TTSTrainingStep(Encoder(3, 32, 5), RADMMMFlow(use_accent=True, **kw), RADMMMLoss(sigma=1.0, kl_loss_start_iter=0),
                            n_speakers=3, n_accents=2, n_text_tokens=40, n_text_dim=32, n_speaker_dim=16, n_accent_dim=8,
                            use_accent=True, binarization_start_iter=0)
    names = [n for n in model.state_dict() if not n.startswith("decoder_criterion")]
    proc = S.procedural_decoder_state({n: tuple(model.state_dict()[n].shape) for n in names})
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()}, strict=False)
    model = model.to(dev).train()
    batch = {k[6:]: torch.from_numpy(np.asarray(g[k])).to(dev) for k in g.files if k.startswith("batch.")}
    with torch.no_grad():
        loss_plain = float(model.training_step(batch, global_step=10)[0])       # (dropout off the record: eval of the same glue)
    batch["input_lengths_host"] = batch["input_lengths"].cpu()
    batch["output_lengths_host"] = batch["output_lengths"].cpu()
    red = BucketedGradReducer(model)
    opt = FlatRAdam(model.named_parameters(), lr=1e-6, weight_decay=1e-6, reducer=red)

    def step():
        red.prepare()
        loss, _, _ = model.training_step(batch, global_step=10)
        loss.backward()
        red.finish()
        opt.clip_grad_norm(1.0)
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss = step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert np.isfinite(float(loss.detach())) and loss_plain > 0
