"""GPU edge cases: batch of one, odd / tiny lengths, group size 1, data-dependent whitening init,
non-default scaling functions and padding modes -- HIP path vs the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def _run_both(kw, B, Tn, lens, seed=5, initialized=True, wn_scale=None):
    from oracle import radmmm_oracle as O
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    cfg = O.DecoderConfig(**kw)
    sd = T(O.procedural_decoder_state(O.decoder_state_shapes(cfg)))
    if not initialized:
        sd["flows.0.invtbl_conv.initialized"] = torch.tensor(False)
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    b = T(O.synthetic_batch(B, Tn, cfg, seed, ragged=False))
    lens = torch.tensor(lens)
    b["lengths"] = lens
    for i in range(B):
        L = int(lens[i])
        b["mel"][i, :, L:] = 0
        b["context"][i, :, L:] = 0
        b["f0"][i, L:] = 0
        b["energy"][i, L:] = 0
    gb = {k: v.to(DEV) for k, v in b.items()}
    sl = SequenceLength(gb["lengths"])
    out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
    lm = RADMMMLoss(n_group_size=cfg.n_group_size)(out, None, sl, 0)["loss_mel"][0]
    lm.backward()
    # oracle (with the same data-dependent init if requested)
    p = {k: v.clone() for k, v in sd.items()}
    if not initialized:
        z0 = O.squeeze_time(b["mel"], cfg.n_group_size)
        mean, ud, up = O.whiten_initialize(z0, lens // cfg.n_group_size)
        p["flows.0.invtbl_conv.input_mean"], p["flows.0.invtbl_conv.upper_diag"], p["flows.0.invtbl_conv.upper"] = mean, ud, up
    p = {k: (v.requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and not k.endswith((".p", "lower_diag", "input_mean")) else v)
         for k, v in p.items()}
    ro = O.decoder_forward(p, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"])
    lo, _ = O.decoder_loss(ro, b["lengths"], cfg.n_group_size)
    lo.backward()
    return dec, out, lm, p, ro, lo, cfg


BASE = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
            n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
            scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True,
            n_conv_layers_per_step=4, n_flows=3)


def _check(dec, out, lm, p, ro, lo, cfg, lens, tol=1e-4):
    ul = torch.tensor(lens) // cfg.n_group_size
    Tg = ro["z_mel"].shape[2]
    m = (torch.arange(Tg)[None] < ul[:, None])[:, None].expand_as(ro["z_mel"])
    zh = out["z_mel"].detach().cpu()[:, :, :Tg]
    assert rel_err(zh[m], ro["z_mel"].detach()[m]) < tol
    assert abs(float(lm.detach()) - float(lo.detach())) < tol * abs(float(lo.detach()))
    params = dict(dec.named_parameters())
    last = cfg.n_flows - 1
    for n in (f"flows.{last}.coupling_tfn.affine_param_predictor.in_layers.1.conv.weight_v",
              "flows.0.coupling_tfn.affine_param_predictor.start.weight_g",
              "flows.1.invtbl_conv.lower", "context_lstm.weight_hh_l0_reverse",
              "flows.1.coupling_tfn.affine_param_predictor.end.weight"):
        a, b = params[n].grad.cpu(), p[n].grad
        assert rel_err(a, b) < 10 * tol, n


def test_batch_of_one_odd_length():
    lens = [31]
    r = _run_both(BASE, 1, 31, lens)
    assert r[1]["z_mel"].shape == (1, 160, 15)          # trailing frame dropped by the squeeze
    _check(*r, lens)


def test_tiny_item_in_ragged_batch():
    lens = [64, 37, 2]                                   # T' = 32, 18, 1
    r = _run_both(BASE, 3, 64, lens)
    _check(*r, lens)


def test_group_size_one():
    kw = dict(BASE, n_group_size=1)
    lens = [40, 29]
    r = _run_both(kw, 2, 40, lens)
    assert r[1]["z_mel"].shape == (2, 80, 40)
    _check(*r, lens)


def test_data_dependent_whitening_init():
    """flows.0.invtbl_conv initialises itself from the first training batch (common.py:569-591)."""
    lens = [256, 231, 200, 187]
    r = _run_both(BASE, 4, 256, lens, initialized=False)
    dec, p = r[0], r[3]
    c = dec.flows[0].invtbl_conv
    assert bool(c.initialized)
    assert rel_err(c.input_mean.cpu(), p["flows.0.invtbl_conv.input_mean"]) < 1e-5
    assert rel_err(c.upper_diag.detach().cpu(), p["flows.0.invtbl_conv.upper_diag"].detach()) < 2e-4
    _check(*r, lens, tol=5e-4)


@pytest.mark.parametrize("scaling", ["exp", "sigmoid"])
def test_other_scaling_functions(scaling):
    kw = dict(BASE, scaling_fn=scaling, n_flows=2)
    lens = [48, 30]
    r = _run_both(kw, 2, 48, lens)
    _check(*r, lens)


def test_no_partial_padding():
    kw = dict(BASE, use_partial_padding=False, n_flows=2)
    lens = [48, 33]
    r = _run_both(kw, 2, 48, lens)
    _check(*r, lens)


def test_rowgemm_degenerate_shapes():
    from rad_mmm_amd._lib import rowgemm
    g = torch.Generator().manual_seed(2)
    for (M, N, K) in [(1, 1, 16), (5, 3, 16), (17, 130, 32), (1, 1, 4), (33, 7, 20)]:
        A = torch.randn(M, (K + 3) // 4 * 4, generator=g)
        Bm = torch.randn(N, (K + 3) // 4 * 4, generator=g)
        C = torch.full((M, (N + 3) // 4 * 4), float("nan"), device=DEV)
        rowgemm(A=A.to(DEV), lda=A.shape[1], B=Bm.to(DEV), ldb=Bm.shape[1], b_layout=0, C=C, ldc=C.shape[1], M=M, N=N,
                K=K, T=M)
        ref = A[:, :K].double() @ Bm[:, :K].double().t()
        assert rel_err(C[:, :N].cpu().double(), ref) < 2e-6, (M, N, K)


def test_masked_reduce_with_empty_item():
    from rad_mmm_amd import ops
    x = torch.randn(3, 5, 9, device=DEV)
    lens = torch.tensor([9, 0, 4], dtype=torch.int32, device=DEV)
    m = (torch.arange(9, device=DEV)[None] < lens[:, None])[:, None].float()
    assert abs(float(ops.masked_sum(x, lens)) - float((x * m).sum())) < 1e-5
    assert abs(float(ops.masked_sumsq(x, lens)) - float(((x * m) ** 2).sum())) < 1e-4


def test_bucket_reducer_direct_gradients_match_plain_backward():
    """BucketedGradReducer with direct-write parameters (ops.grad_out sinks, .grad adopted by autograd) must give
    the same gradients as a plain backward, two steps in a row (second step: stale bucket contents)."""
    import numpy as np
    import torch
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in S.procedural_decoder_state(S.decoder_state_shapes(cfg)).items()}
    dev = "cuda:0"
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    grads = {}
    for mode in ("plain", "reducer"):
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(sd)
        dec = dec.to(dev).train()
        red = BucketedGradReducer(dec) if mode == "reducer" else None
        for it in range(2):
            b = {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(3, 48, cfg, seed=7 + it, ragged=True).items()}
            sl = SequenceLength(b["lengths"])
            if red is not None:
                red.prepare()
            else:
                dec.zero_grad(set_to_none=True)
            out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
            crit(out, None, sl, 0)["loss_mel"][0].backward()
            if red is not None:
                red.finish()
        grads[mode] = {n: p.grad.detach().clone() for n, p in dec.named_parameters()}
        if red is not None:
            n_direct = sum(1 for n, p in dec.named_parameters() if red._direct[id(p)])
            assert n_direct > 50
            for n, p in dec.named_parameters():       # every .grad lives inside its bucket
                assert p.grad.data_ptr() == red._views[id(p)].data_ptr(), n
    for n in grads["plain"]:
        assert torch.equal(grads["plain"][n], grads["reducer"][n]), n


@pytest.mark.parametrize("c,ldw,off", [(160, 160, 0), (154, 160, 6), (7, 12, 3)])
def test_lu_weight_matches_oracle(c, ldw, off):
    """ops.LUWeightFn (one launch each way) vs the oracle's stock-op W = P (L U), log|det|, and their
    autograd gradients (reference common.py:507-548), incl. the column offset of an early exit."""
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import ops
    g = torch.Generator().manual_seed(c)
    Wr = torch.linalg.qr(torch.randn(c, c, generator=g))[0]
    p, lower, upper = torch.linalg.lu(Wr)
    prm = {"p": p, "lower": torch.tril(lower, -1) + torch.triu(torch.randn(c, c, generator=g)),   # junk above the
           "lower_diag": torch.ones(c), "upper": torch.triu(upper, 1) + torch.tril(torch.randn(c, c, generator=g)),
           "upper_diag": torch.diag(upper).clone()}                                                # diagonals is ignored
    gW = torch.randn(ldw, ldw, generator=g)
    gld = torch.randn((), generator=g)
    ref = {k: v.clone().requires_grad_(k in ("lower", "upper", "upper_diag")) for k, v in prm.items()}
    W_ref = O.lus_weight(ref, "")
    ld_ref = torch.log(torch.abs(ref["upper_diag"])).sum()
    ((W_ref * gW[:c, off:off + c]).sum() + ld_ref * gld).backward()
    dev = {k: v.to(DEV).requires_grad_(k in ("lower", "upper", "upper_diag")) for k, v in prm.items()}
    W, ld = ops.LUWeightFn.apply(dev["p"], dev["lower"], dev["lower_diag"], dev["upper"], dev["upper_diag"], ldw, off)
    ((W * gW.to(DEV)).sum() + ld * gld.to(DEV)).backward()
    W = W.detach().cpu()
    assert rel_err(W[:c, off:off + c], W_ref.detach()) < 1e-5
    blk = torch.zeros(ldw, ldw, dtype=torch.bool)
    blk[:c, off:off + c] = True
    assert torch.all(W[~blk] == 0)
    assert abs(float(ld) - float(ld_ref)) < 1e-5 * max(1.0, abs(float(ld_ref)))
    for k in ("lower", "upper", "upper_diag"):
        assert rel_err(dev[k].grad.cpu(), ref[k].grad) < 1e-5, k


def test_saturation_check_flags_an_overflowing_gradient_scale(monkeypatch):
    """RADMMM_CHECK_SATURATION=1: silent fp16 clamping of the split gradients becomes an error.  A normal step
    passes; a gradient scale 2^30 too large must be reported."""
    from rad_mmm_amd import ops
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=3, ragged=True).items()}
    sl = SequenceLength(b["lengths"])
    monkeypatch.setenv("RADMMM_CHECK_SATURATION", "1")

    def step():
        dec.zero_grad(set_to_none=True)
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()

    step()                                                   # in range: no complaint
    real = ops.grad_scale
    monkeypatch.setattr(ops, "grad_scale", lambda box, g: real(box, g) * 2.0 ** 30)
    with pytest.raises(FloatingPointError, match="saturated"):
        step()


def test_saturation_is_detected_by_default(monkeypatch):
    """Without RADMMM_CHECK_SATURATION: a split producer that has to clamp raises the decoder's device-side flag; the
    error surfaces (a) synchronously from decoder.check_saturation() and (b) without any synchronisation from a later
    training pass.  A step in range raises nothing."""
    import time
    from rad_mmm_amd import ops
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    monkeypatch.delenv("RADMMM_CHECK_SATURATION", raising=False)
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=3, ragged=True).items()}
    sl = SequenceLength(b["lengths"])

    def step():
        dec.zero_grad(set_to_none=True)
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()

    for _ in range(3):
        step()
        dec.check_saturation()                               # in range: no complaint, scale carried between passes
    real = ops.grad_scale
    monkeypatch.setattr(ops, "grad_scale", lambda box, g: real(box, g) * 2.0 ** 30)
    step()                                                   # saturates silently on the device ...
    with pytest.raises(FloatingPointError, match="saturated"):
        dec.check_saturation()                               # ... (a) reported synchronously
    step()                                                   # saturates again; flag published by the next forward
    monkeypatch.setattr(ops, "grad_scale", real)
    before = dec._grad_scale.saturated_passes
    ops.GradScale._warned.discard("f16")
    with pytest.warns(RuntimeWarning, match="saturated"):
        for _ in range(4):                                   # (b) a later pass picks the flag up by polling: a warning
            torch.cuda.synchronize()                         #     (once per process) and a counter, no exception by default
            time.sleep(0.01)
            step()
    assert dec._grad_scale.saturated_passes > before
    step()
    dec.check_saturation()                                   # recovered
    # strict mode: the deferred report raises
    monkeypatch.setattr(ops, "grad_scale", lambda box, g: real(box, g) * 2.0 ** 30)
    step()
    monkeypatch.setattr(ops, "grad_scale", real)
    dec._grad_scale.strict = True
    with pytest.raises(FloatingPointError, match="saturated"):
        for _ in range(4):
            torch.cuda.synchronize()
            time.sleep(0.01)
            step()
    dec._grad_scale.strict = None


def test_nonfinite_upstream_gradient_poisons_the_pass_and_keeps_the_scale(monkeypatch):
    """An fp16-AMP overflow step hands the decoder an Inf / NaN gradient.  The split producers clamp those to finite
    values, so the pass is marked instead: every weight-norm gain gradient becomes NaN (what GradScaler / clip_grad_norm_
    look at), nothing raises -- not one step later either -- and the gradient scale keeps its value for the next pass."""
    import time
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    monkeypatch.delenv("RADMMM_CHECK_SATURATION", raising=False)
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(T(S.procedural_decoder_state(S.decoder_state_shapes(cfg))))
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=3, ragged=True).items()}
    sl = SequenceLength(b["lengths"])

    def step(mult=1.0):
        dec.zero_grad(set_to_none=True)
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        (crit(out, None, sl, 0)["loss_mel"][0] * mult).backward()

    step()
    step()
    torch.cuda.synchronize()
    S0 = dec._grad_scale.S
    gains = [p for n, p in dec.named_parameters() if n.endswith("weight_g") and "affine_param_predictor" in n]
    assert gains and all(torch.isfinite(p.grad).all() for p in gains)
    step(float("inf"))                                       # the scaled loss overflowed
    torch.cuda.synchronize()
    assert all(torch.isnan(p.grad).all() for p in gains)     # visible to GradScaler.unscale_ / clip_grad_norm_
    total = torch.nn.utils.clip_grad_norm_(dec.parameters(), 1.0)
    assert not torch.isfinite(total)
    ops_mod = __import__("rad_mmm_amd.ops", fromlist=["GradScale"])
    ops_mod.GradScale._warned.discard("nonfinite")
    with pytest.warns(RuntimeWarning, match="non-finite"):
        for _ in range(3):                                   # later passes: the report is a warning, the scale is kept
            time.sleep(0.01)
            step()
            torch.cuda.synchronize()
    assert dec._grad_scale.nonfinite_passes == 1 and dec._grad_scale.S == S0
    assert all(torch.isfinite(p.grad).all() for p in gains)


def test_batch_shape_changes_between_steps():
    """Real batches differ in size and length from step to step.  The pooled transposed operand buffers of the
    weight-gradient GEMMs are reused across steps: a step after a DIFFERENT shape (here one with the same padded
    contraction length but a shorter real extent, the case where stale columns would be contracted) must give
    exactly the gradients of the same step run from a clean pool."""
    from rad_mmm_amd import ops
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    shapes = [(3, 96), (2, 138), (3, 96), (1, 40), (2, 138)]          # (3, 96) and (2, 138): both pad to 192 columns

    def grads(B, Tn, seed):
        # full-length batches for the 192-column shape: every column of the buffers then holds data
        b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(B, Tn, cfg, seed=seed, ragged=(B != 3)).items()}
        sl = SequenceLength(b["lengths"])
        dec.zero_grad(set_to_none=True)
        # the power-of-two gradient scale is carried from pass to pass (ops.GradScale): start every run from the same
        # state so that "bit for bit" compares the pooled buffers and nothing else
        dec._grad_scale = None
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        return {n: p.grad.detach().clone() for n, p in dec.named_parameters()}

    seq = [grads(B, Tn, 50 + i) for i, (B, Tn) in enumerate(shapes)]
    for i, (B, Tn) in enumerate(shapes):
        ops._ts_pool.clear()
        ref = grads(B, Tn, 50 + i)
        for n in ref:
            assert torch.equal(seq[i][n], ref[n]), (i, (B, Tn), n)


def test_second_backward_without_rearming_accumulates():
    """The reducer's direct-write sinks are one-shot: backward twice between prepare() and finish() (gradient
    accumulation) must ADD the second gradient, not overwrite the first."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    bs = [{k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=70 + i, ragged=True).items()}
          for i in range(2)]

    def backward(b):
        sl = SequenceLength(b["lengths"])
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()

    single = []
    for b in bs:
        dec.zero_grad(set_to_none=True)
        backward(b)
        single.append({n: p.grad.detach().clone() for n, p in dec.named_parameters()})
    red = BucketedGradReducer(dec)
    red.prepare()
    backward(bs[0])
    backward(bs[1])
    red.finish()
    for n, p in dec.named_parameters():
        want = (single[0][n] + single[1][n]).cpu()
        assert rel_err(p.grad.cpu(), want) < 1e-6, n


def test_autocast_context_does_not_reach_the_kernels():
    """Lightning `precision: bf16-mixed` wraps the step in torch.autocast.  The package computes in fp32 through raw
    pointers, so its Functions switch autocast off inside: decoder forward + backward under autocast(bf16) must
    equal the plain run bit for bit (without the guard the LSTM's input GEMM would hand bf16 to an fp32 kernel)."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=9, ragged=True).items()}
    sl = SequenceLength(b["lengths"])
    res = []
    for amp in (False, True):
        dec.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
            loss = crit(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        assert out["z_mel"].dtype == torch.float32
        res.append((out["z_mel"].detach().clone(), loss.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in dec.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for n in res[0][2]:
        assert torch.equal(res[0][2][n], res[1][2][n]), n


@pytest.mark.parametrize("opt", [dict(use_context_lstm=False),
                                 # (the widths count f0 / energy regardless of the flag, models/radmmm.py:73-78: the
                                 #  flag only works together with zero-width f0 / energy, as in the reference)
                                 dict(context_w_f0_and_energy=False, n_f0_dims=0, n_energy_avg_dims=0),
                                 dict(use_accent_emb_for_decoder=False),
                                 dict(use_context_lstm=False, context_w_f0_and_energy=False, n_f0_dims=0, n_energy_avg_dims=0)])
def test_context_options_of_the_constructor(opt):
    """decoders.py:83-143 / models/radmmm.py:103-148: context without the LSTM, without f0 / energy columns,
    without the accent embedding -- HIP path vs the oracle, outputs, NLL and gradients."""
    kw = dict(BASE, n_flows=2, n_text_dim=64, **opt)
    lens = [64, 38]
    dec, out, lm, p, ro, lo, cfg = _run_both(kw, 2, 64, lens)
    ul = torch.tensor(lens) // cfg.n_group_size
    Tg = ro["z_mel"].shape[2]
    m = (torch.arange(Tg)[None] < ul[:, None])[:, None].expand_as(ro["z_mel"])
    assert rel_err(out["z_mel"].detach().cpu()[:, :, :Tg][m], ro["z_mel"].detach()[m]) < 1e-4
    assert abs(float(lm.detach()) - float(lo.detach())) < 1e-4 * abs(float(lo.detach()))
    assert out["context_w_spkvec"].shape[1] == dec.decoder_cond_dims
    params = dict(dec.named_parameters())
    # every parameter gradient, elementwise against the oracle's autograd (softplus: no kink; was three tensors at 1e-3)
    errs = {n: rel_err(params[n].grad.cpu(), p[n].grad) for n in params
            if params[n].grad is not None and p[n].grad is not None and float(p[n].grad.abs().max()) > 0}
    bad = {n: e for n, e in errs.items() if not e < 5e-4}
    print(f"context options {opt}: {len(errs)} parameter gradients, worst elementwise error {max(errs.values()):.2e}")
    assert len(errs) > 50 and not bad, bad


def test_relu_activation_in_the_wn(monkeypatch):
    """affine_activation='relu' (common.py:776-835 takes either): fused epilogue / activation-gradient kernels vs the oracle, on
    the exact split-f16 products (no shipped config uses relu; the product scheme is not what this test is about).
    relu' is discontinuous: an activation within rounding of 0 flips a whole gradient term.  ACCOUNTED (the
    method of tests/test_attribute_predictors.py): the HIP run's decisions -- its activation outputs > 0, read from the fp32
    outputs of the in_layer / res_skip launches -- are imposed on the oracle (oracle.wn_forward's `kinks`); every decision
    that differs from the oracle's own must belong to a pre-activation within rounding of 0 and is counted; every gradient is
    then held to 5e-4 (was: 2e-3 with nothing counted)."""
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import ops as _ops
    monkeypatch.setenv("RADMMM_PRECISION", "h3")
    kw = dict(BASE, n_flows=2, n_text_dim=64, affine_activation="relu")
    lens = [64, 50]
    outs, real = [], _ops.rowgemm_h3

    def spy(**k):
        real(**k)
        if k.get("act") and k.get("dact") is None and k.get("C") is not None and k.get("M") == 2 * 32:
            outs.append(k["C"])                                     # forward launches with an activation: in_0, res_0, in_1, ...
    monkeypatch.setattr(_ops, "rowgemm_h3", spy)
    dec, out, lm, p, ro, lo, cfg = _run_both(kw, 2, 64, lens)
    monkeypatch.setattr(_ops, "rowgemm_h3", real)
    nl = cfg.n_conv_layers_per_step
    assert len(outs) == 2 * nl * cfg.n_flows, len(outs)
    ul = torch.tensor(lens) // cfg.n_group_size
    Tg = ro["z_mel"].shape[2]
    m = (torch.arange(Tg)[None] < ul[:, None])[:, None].expand_as(ro["z_mel"])
    assert rel_err(out["z_mel"].detach().cpu()[:, :, :Tg][m], ro["z_mel"].detach()[m]) < 1e-4
    assert abs(float(lm.detach()) - float(lo.detach())) < 1e-4 * abs(float(lo.detach()))
    # the oracle's pre-activations, the HIP decisions, the accounting
    b = T(O.synthetic_batch(2, 64, cfg, 5, ragged=False))
    lt = torch.tensor(lens)
    b["lengths"] = lt
    for i in range(2):
        L = int(lt[i])
        b["mel"][i, :, L:] = 0
        b["context"][i, :, L:] = 0
        b["f0"][i, L:] = 0
        b["energy"][i, L:] = 0
    q = {k: v.detach().clone() for k, v in p.items()}
    rec = {}
    with torch.no_grad():
        O.decoder_forward(q, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"],
                          wn_kinks={"record": rec})
    gates, flipped, total, worst = {}, 0, 0, 0.0
    it = iter(outs)
    for f in range(cfg.n_flows):
        pre = f"flows.{f}.coupling_tfn.affine_param_predictor."
        for j in range(nl):
            for kind in ("in", "res"):
                y = next(it).detach().cpu()                         # [B * Tg, Wc] channels-last rows
                r = rec[(pre, j, kind)]                             # [B, Wc, Tg]
                gt = y.reshape(2, Tg, -1).permute(0, 2, 1) > 0
                diff = (r > 0) != gt
                total += gt.numel()
                if diff.any():
                    flipped += int(diff.sum())
                    worst = max(worst, float(r[diff].abs().max()) / (2e-5 * float(r.pow(2).mean().sqrt())))
                gates[(pre, j, kind)] = gt
    print(f"relu WN: {flipped} of {total} activations on the other side of the kink; the worst one lies at {worst:.2f} x the rounding "
          f"bound (2e-5 rms)")
    assert worst <= 1.0 and flipped <= max(20, total // 10000)
    q = {k: (v.requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and not k.endswith((".p", "lower_diag", "input_mean")) else v)
         for k, v in q.items()}
    ro2 = O.decoder_forward(q, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"],
                            wn_kinks={"gates": gates})
    O.decoder_loss(ro2, b["lengths"], cfg.n_group_size)[0].backward()
    params = dict(dec.named_parameters())
    errs = {n: rel_err(params[n].grad.cpu(), q[n].grad) for n in params if q[n].grad is not None and float(q[n].grad.abs().max()) > 0}
    bad = {n: e for n, e in errs.items() if not e < 5e-4}
    print(f"relu WN: {len(errs)} parameter gradients, worst elementwise error {max(errs.values()):.2e}")
    assert len(errs) > 50 and not bad, bad


def test_frozen_whitening_layer_trains_the_rest():
    """freeze_whitening_layer=True (decoders.py:143-145): flow 0's 1x1 conv gets no gradient, the others do, and the
    bucket reducer copes with parameters outside its buckets."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    grads = {}
    for frozen in (False, True):
        dec = RADMMMFlow(use_accent=True, freeze_whitening_layer=frozen, **kw)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).train()
        red = BucketedGradReducer(dec)
        b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 64, cfg, seed=4, ragged=True).items()}
        sl = SequenceLength(b["lengths"])
        red.prepare()
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        RADMMMLoss(sigma=1.0, n_group_size=2)(out, None, sl, 0)["loss_mel"][0].backward()
        red.finish()
        grads[frozen] = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in dec.named_parameters()}
        if frozen:
            assert all(not p.requires_grad and p.grad is None for n, p in dec.named_parameters() if n.startswith("flows.0.invtbl_conv."))
    for n, gfree in grads[False].items():
        if not n.startswith("flows.0.invtbl_conv."):
            assert torch.equal(gfree, grads[True][n]), n


def test_three_training_steps_follow_the_oracle_trajectory():
    """End to end, several steps: decoder forward + NLL + backward -> global-norm clip 1.0 -> RAdam, three times, HIP
    (bucket reducer + FlatRAdam) against the CPU oracle (autograd of the restatement + its RAdam / clip): the loss of
    every step and the parameters after the last one must agree."""
    from oracle import radmmm_oracle as O
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
    kw = dict(BASE, n_flows=2, n_text_dim=64)
    cfg = O.DecoderConfig(**kw)
    sd = T(O.procedural_decoder_state(O.decoder_state_shapes(cfg)))
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    red = BucketedGradReducer(dec)
    lr, wd = 2e-4, 1e-6
    opt = FlatRAdam(dec.named_parameters(), lr=lr, weight_decay=wd, reducer=red)
    crit = RADMMMLoss(n_group_size=cfg.n_group_size)
    trainable = [n for n, _ in dec.named_parameters()]
    p = {k: v.clone() for k, v in sd.items()}
    for n in trainable:
        p[n].requires_grad_(True)
    m = {n: torch.zeros_like(p[n]) for n in trainable}
    v = {n: torch.zeros_like(p[n]) for n in trainable}
    lens = [[64, 40], [64, 64], [52, 64]]
    for k in range(3):
        b = T(O.synthetic_batch(2, 64, cfg, 30 + k, ragged=False))
        b["lengths"] = torch.tensor(lens[k])
        for i in range(2):
            L = lens[k][i]
            b["mel"][i, :, L:] = 0; b["context"][i, :, L:] = 0; b["f0"][i, L:] = 0; b["energy"][i, L:] = 0
        gb = {kk: vv.to(DEV) for kk, vv in b.items()}
        sl = SequenceLength(gb["lengths"])
        red.prepare()
        out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
        loss = crit(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        red.finish()
        total = opt.clip_grad_norm(1.0)
        opt.step()
        # oracle
        for n in trainable:
            p[n].grad = None
        ro = O.decoder_forward(p, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"])
        lo, _ = O.decoder_loss(ro, b["lengths"], cfg.n_group_size)
        lo.backward()
        rt, clipped = O.clip_grad_norm([p[n].grad for n in trainable], 1.0)
        with torch.no_grad():
            for n, gc in zip(trainable, clipped):
                O.radam_step(p[n], gc, m[n], v[n], k + 1, lr=lr, weight_decay=wd)
        assert abs(float(loss) - float(lo)) < 2e-5 * abs(float(lo)), (k, float(loss), float(lo))
        assert abs(float(total) - float(rt)) < 1e-4 * float(rt), (k, float(total), float(rt))
    for n, q in dec.named_parameters():
        d = float((q.detach().cpu() - p[n].detach()).abs().max())
        assert d <= 2e-2 * lr + 1e-6 * float(p[n].detach().abs().max()), (n, d)      # 3 steps of <= lr each; 1 % of a step


def test_sync_masked_batchnorm_two_rank_emulation(monkeypatch):
    """MaskedBatchNorm1d.distributed_sync (maskedbatchnorm1d.py:88-95; switched by toggle_syncbnorm as
    tts_lightning_modules.py:241-243 does): two data-parallel ranks with synchronised statistics must reproduce the single
    process run on the concatenated batch -- outputs per item, and parameter gradients as the rank mean.

    One GPU, so the two ranks are emulated by record / replay: torch.distributed.all_reduce is replaced by a stand-in that
    returns the cross-rank sum for every call index already resolved; each pass over both ranks resolves at least the next
    index (its inputs only depend on earlier, resolved, collectives), forward calls first, backward calls after."""
    import torch.distributed as dist
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.spline_layers import toggle_syncbnorm
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=False, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=3,
              n_splines=2, use_bn=True)
    cfg = S.DecoderConfig(**kw)
    sd = T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    full = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(4, 64, cfg, seed=21, ragged=False).items()}
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)

    def run(batch, sync):
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).train()
        dec.precision_guard_every = 0          # (the FP8-cross guard's own MAX all-reduce is not part of what is counted here)
        toggle_syncbnorm(dec, sync)
        sl = SequenceLength(batch["lengths"])
        out = dec(batch["mel"], batch["spk"], batch["context"], sl, batch["f0"], batch["energy"], batch["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        torch.cuda.synchronize()
        return out["z_mel"].detach().clone(), {n: p.grad.detach().clone() for n, p in dec.named_parameters()}

    z_full, g_full = run(full, False)
    halves = [{k: v[:2] for k, v in full.items()}, {k: v[2:] for k, v in full.items()}]

    known, state = {}, {"idx": 0, "rec": None}

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        i = state["idx"]
        state["idx"] += 1
        state["rec"].append(t.detach().clone())
        if i in known:
            t.copy_(known[i])

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    n_calls, results = None, None
    for _ in range(80):
        recs, results = [], []
        for h in halves:
            state["idx"], state["rec"] = 0, []
            results.append(run(h, True))
            recs.append(state["rec"])
        n_calls = len(recs[0])
        assert len(recs[1]) == n_calls and n_calls == 2 * 2 * 4 * 1          # 2 spline flows x 4 blocks, forward + backward
        nxt = len(known)
        if nxt == n_calls:
            break
        known[nxt] = recs[0][nxt] + recs[1][nxt]      # inputs of call `nxt` were computed from resolved collectives only
    assert len(known) == n_calls
    (zA, gA), (zB, gB) = results
    assert rel_err(torch.cat((zA, zB)).cpu(), z_full.cpu()) < 2e-5
    worst = 0.0
    for n in g_full:
        ref = g_full[n]
        got = 0.5 * (gA[n] + gB[n])
        scale = float(ref.abs().max())
        # the per-channel scale and bias of the conv that feeds a batch-norm have an analytically ZERO gradient (the
        # normalisation removes both): what is left is rounding residue, not comparable in relative terms
        if scale < 1e-9 or n.endswith(("hidden_conv.conv.weight_g", "hidden_conv.conv.bias")):
            continue
        e = float((got - ref).abs().max()) / scale
        worst = max(worst, e)
        assert e < 5e-4, (n, e)
    print(f"sync masked batch-norm, 2 emulated ranks vs concatenated batch: worst parameter-gradient rel err {worst:.2e}")
