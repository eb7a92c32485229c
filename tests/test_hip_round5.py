"""Round-5 GPU tests (all through the C ABI):
  * a training run is BITWISE repeatable: the gradient scale / FP8 exponent statistics of a pass are adopted at a fixed lag
    (ops.GradScale), never "whenever an event poll happened to succeed" -- two runs with very different host timing produce
    identical loss bits and identical parameter bytes (the reference's CPU path is bitwise repeatable);
  * the FP8-cross runtime guard issues its collective on every rank whatever the rank-local batch size;
  * the data-initialised 1x1 conv re-checks `initialized` after an in-place reset;
  * the binarisation loss has finite gradients when the soft attention holds exact zeros."""
import time

import numpy as np
import pytest
import torch

DEV = torch.device("cuda:0")

KW = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
          n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
          scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True, n_conv_layers_per_step=4, n_flows=2)


def _T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def _training_run(perturb_host: bool, steps: int = 6):
    """`steps` training steps (decoder forward + NLL + backward -> bucket reducer -> clip 1.0 -> FlatRAdam) of a 2-flow
    WN-1024 decoder on the wide FP8-cross kernels (4800 rows), a different batch every step and a mel distribution that
    changes mid-run (the gradient maximum moves by > 2x: the scale and the exponent really get re-adopted).
    perturb_host: synchronise + sleep after every step (the host never runs ahead) instead of free-running."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
    cfg = S.DecoderConfig(**KW)
    dec = RADMMMFlow(use_accent=True, **KW)
    dec.load_state_dict(_T(S.procedural_decoder_state(S.decoder_state_shapes(cfg))))
    dec = dec.to(DEV).train()
    assert dec.gemm_precision == "f8x"
    red = BucketedGradReducer(dec)
    opt = FlatRAdam(dec.named_parameters(), lr=2e-4, weight_decay=1e-6, reducer=red)
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    losses, scales = [], []
    for k in range(steps):
        b = {kk: vv.to(DEV) for kk, vv in _T(S.synthetic_batch(12, 800, cfg, 500 + k, ragged=True)).items()}
        if k >= 2:
            b["mel"] = (b["mel"] - 2.5) * 3.0 + 2.5                # 3x the spread from the third step on
        sl = SequenceLength(b["lengths"])
        red.prepare()
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        loss = crit(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        red.finish()
        opt.clip_grad_norm(1.0)
        opt.step()
        losses.append(loss.detach())
        scales.append((dec._grad_scale.S, dec._grad_scale.grad_exp()))
        if perturb_host:
            torch.cuda.synchronize()
            time.sleep(0.05)
    torch.cuda.synchronize()
    params = {n: p.detach().cpu().clone() for n, p in dec.named_parameters()}
    return [float(l) for l in losses], [l.cpu().view(torch.int32).item() for l in losses], params, scales


@pytest.mark.gpu
def test_training_run_is_bitwise_repeatable(monkeypatch):
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    monkeypatch.delenv("RADMMM_CHECK_SATURATION", raising=False)
    la, ba, pa, sa = _training_run(perturb_host=False)
    lb, bb, pb, sb = _training_run(perturb_host=True)
    assert sa == sb, (sa, sb)                       # same (scale, exponent) in the same pass, whatever the host timing
    assert len({s for s, _ in sa}) > 1, sa          # ... and the scale really changed during the run
    assert ba == bb, (la, lb)                       # identical loss BITS in every step
    bad = [n for n in pa if not torch.equal(pa[n].view(torch.int32), pb[n].view(torch.int32))]
    assert not bad, bad[:5]                         # identical parameter bytes after six updates


@pytest.mark.gpu
def test_precision_guard_collective_does_not_depend_on_the_local_batch(monkeypatch):
    """With a process group active, a rank whose batch is below the wide kernels' minimum must still take part in the guard's
    MAX all-reduce (with a measurement of 0): ranks pad to their own longest utterance, so one may be above the minimum and
    the other below, and a rank that skipped would leave the other's collective to pair with a gradient bucket."""
    import radmmm_synth as S
    import torch.distributed as dist
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    kw = dict(KW, n_text_dim=64)
    cfg = S.DecoderConfig(**kw)
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict(_T(S.procedural_decoder_state(S.decoder_state_shapes(cfg))))
    dec = dec.to(DEV).train()
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None, async_op=False: calls.append(t.detach().clone()))
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    for B, T in ((2, 64), (2, 96)):                     # 64 / 96 rows: far below RADMMM_F8X_MIN_ROWS
        dec._guard.update(n=0, pending=False)
        b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(B, T, cfg, 3, ragged=True)).items()}
        n0 = len(calls)
        dec(b["mel"], b["spk"], b["context"], SequenceLength(b["lengths"]), b["f0"], b["energy"], b["accent"])
        guard_calls = [c for c in calls[n0:] if c.numel() == 1 and c.dtype == torch.float32]
        assert len(guard_calls) == 1 and float(guard_calls[0]) == 0.0, (B, T, len(calls) - n0)
        assert dec._guard["pending"]
    assert dec.precision_guard_status()[0] == 0.0


@pytest.mark.gpu
def test_data_initialised_conv_rechecks_after_an_in_place_reset(capsys):
    """decoders.FlowStep caches the one host read of the `initialized` buffer per VERSION of the buffer: an in-place reset
    (fill_ / copy_ / a re-init utility) must be seen by the next training forward, as the reference's per-forward check
    would see it (common.py:575-590)."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    kw = dict(KW, n_text_dim=64)
    cfg = S.DecoderConfig(**kw)
    dec = RADMMMFlow(use_accent=True, **kw)
    sd = _T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    dec.load_state_dict(sd)
    dec = dec.to(DEV).train()
    conv = dec.flows[0].invtbl_conv
    b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(4, 512, cfg, 3, ragged=True)).items()}    # > 160 valid frames: a regular covariance

    def fwd():
        return dec(b["mel"], b["spk"], b["context"], SequenceLength(b["lengths"]), b["f0"], b["energy"], b["accent"])
    conv.initialized.fill_(False)
    fwd()
    assert bool(conv.initialized) and "initialized invertible conv" in capsys.readouterr().out
    fwd()
    assert "initialized invertible conv" not in capsys.readouterr().out          # remembered: no second initialisation
    w0 = conv.weight().detach().clone()
    conv.initialized.fill_(False)                                                 # in-place reset
    with torch.no_grad():
        b["mel"].mul_(1.7)                                                        # other data -> another whitening matrix
    fwd()
    assert bool(conv.initialized) and "initialized invertible conv" in capsys.readouterr().out
    assert not torch.equal(conv.weight().detach(), w0)
    import copy
    twin = copy.deepcopy(dec)
    twin.flows[0].invtbl_conv.initialized.fill_(False)
    twin(b["mel"], b["spk"], b["context"], SequenceLength(b["lengths"]), b["f0"], b["energy"], b["accent"])
    assert bool(twin.flows[0].invtbl_conv.initialized)


def test_binarization_loss_gradient_is_finite_with_exact_zeros():
    """loss.py:143-151 gathers soft[hard == 1]; the masked-sum form must not let an exact 0 at an unselected position
    (every padded text column of the masked softmax) reach log's backward as 0 * inf."""
    from rad_mmm_amd.loss import AttentionBinarizationLoss
    soft = torch.tensor([[[[0.7, 0.3, 0.0], [0.2, 0.8, 0.0], [0.1, 0.9, 0.0]]]], requires_grad=True)
    hard = torch.tensor([[[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0]]]])
    loss = AttentionBinarizationLoss()(hard, soft)
    loss.backward()
    ref = -(torch.log(torch.tensor([0.7, 0.8, 0.9]))).mean()
    assert abs(float(loss) - float(ref)) < 1e-6
    assert torch.isfinite(soft.grad).all(), soft.grad
    want = torch.zeros_like(soft)
    want[0, 0, 0, 0], want[0, 0, 1, 1], want[0, 0, 2, 1] = -1 / 0.7 / 3, -1 / 0.8 / 3, -1 / 0.9 / 3
    assert torch.allclose(soft.grad, want, atol=1e-6)
    # a selected position with probability 0 keeps torch's BCE clamp (-100) and a finite gradient as well
    soft2 = torch.tensor([[[[0.0, 1.0]]]], requires_grad=True)
    l2 = AttentionBinarizationLoss()(torch.tensor([[[[1.0, 0.0]]]]), soft2)
    l2.backward()
    assert float(l2) == 100.0 and torch.isfinite(soft2.grad).all()


def _bits(a):
    return a.view(torch.int16 if a.dtype == torch.float16 else torch.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["win", "win_xt", "one", "h3d", "generic"])
def test_dgrad_epilogue_reads_the_saved_activation_from_its_split_pair(kind, monkeypatch):
    """Round 5 (radmmm_rowgemm_desc.dact_h / dact_x, C == NULL): the data-gradient epilogue's softplus' factor from the 8-bit
    split pair of the hidden state (hi + lo8 * 2^-(11+e)) instead of its fp32 copy, and no fp32 copy of the result.
      * against the fp32 source: the factor 1 - exp(-y) moves by <= 1.2e-5 absolute (y within 2^-15 relative), so the fp32
        result agrees to 4e-5 of its maximum and the bias sums to 2e-5;
      * C == NULL changes nothing else: split pair and bias sums bit-identical to the launch that also writes C.
    Kernels: shared-window 5-tap (with and without the extra K segment), one-tap, per-tap tiles (T below the window kernel's
    minimum) and the generic LDS-parking epilogue (an `add` input)."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    monkeypatch.setenv("RADMMM_H3W_MB", "4" if kind == "h3d" else "7")     # (tile height: 7 = the window / one-tap kernels' shape)
    W = 512
    B, T = (3, 300) if kind != "h3d" else (6, 150)
    lens_l = [T, T - 37, T // 2 + 5] * (B // 3)
    N = B * T
    taps = 1 if kind == "one" else 5
    gen = torch.Generator().manual_seed(11 + len(kind))
    Hs = torch.nn.functional.softplus(torch.randn(N, W, generator=gen) * 2).to(DEV)        # the saved hidden state (softplus output)
    gy = (torch.randn(2 * N, W, generator=gen) * 3e-3).to(DEV)
    w = (torch.randn(W, W, taps + 1, generator=gen) * 0.03).to(DEV)
    addt = (torch.randn(N, W, generator=gen) * 1e-3).to(DEV)
    lens = torch.tensor(lens_l, dtype=torch.int32, device=DEV)
    S, GE = 2048.0, ops.X8_GRAD_EXP
    Ah, Al = ops.split_f16(gy, W, S, W, 2, GE)
    Wh, Wl, _ = ops.split_weight(w, None, W, nprod=2)                                       # [taps + 1][W][W]
    Hh, Hl = ops.split_f16(Hs, W, 1.0, W, 2, ops.X8_ACT_EXP)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    css = torch.empty(int(ops.lib.radmmm_rowgemm_h3_colsum_scratch_floats(N, W)), device=DEV)
    base = dict(nprod=2, a8_exp=GE, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / (S * ops.W_SCALE), T=T, sat_flag=flag, Ah=Ah, Al=Al,
                lda_h=W, Bh=Wh, Bl=Wl, ldb_h=W, b_tap_stride_h=Wh.stride(0), ldc=W, M=N, N=W, K=W, taps=taps, dil=2 if taps > 1 else 1,
                sign=-1, lens=lens, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, ldch=W, ch_scale=S, split_fmt=ops.SPLIT_X8A,
                ch_x8_exp=GE)
    if kind == "win_xt":
        base.update(extra_tap=1, extra_a_rows=N)
    if kind == "generic":
        base.update(add=addt, ldadd=W)
    use_cs = kind != "generic"                                   # (colsum_out excludes an `add` input)

    def run(pair, with_c):
        C = torch.full((N, W), float("nan"), device=DEV) if with_c else None
        Ch, Cl = ops._halves(N, W, like=Hs)
        Ch.fill_(float("nan")), Cl.zero_()
        cs = torch.full((W,), float("nan"), device=DEV)
        src = dict(dact_h=Hh, dact_x=Hl, lddact_h=W, dact_x8_exp=ops.X8_ACT_EXP) if pair else dict(dact_src=Hs, lddact=W)
        extra = dict(colsum_out=cs, colsum_scratch=css) if use_cs else {}
        rowgemm_h3(C=C, Ch=Ch, Cl=Cl, **src, **extra, **base)
        torch.cuda.synchronize()
        return C, Ch, Cl, cs
    c_ref, h_ref, l_ref, s_ref = run(False, True)
    c_p, h_p, l_p, s_p = run(True, True)
    _, h_n, l_n, s_n = run(True, False)
    scale = float(c_ref.abs().max())
    assert scale > 0 and torch.isfinite(c_ref).all()
    assert float((c_p - c_ref).abs().max()) <= 4e-5 * scale
    # reconstruction h + l for comparison of the pairs is implied by C; the pair written WITHOUT C is the pair written with it
    assert torch.equal(_bits(h_n), _bits(h_p)) and torch.equal(l_n.view(torch.int16), l_p.view(torch.int16))
    if use_cs:
        assert torch.equal(_bits(s_n), _bits(s_p))
        assert float((s_p - s_ref).abs().max()) <= 2e-5 * float(s_ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("taps", [1, 5])
def test_forward_epilogue_without_an_fp32_copy_writes_the_same_pair(taps, monkeypatch):
    """C == NULL on the forward launches (start conv, in_layer convs): the split pair that carries the hidden state is
    bit-identical to the one written beside an fp32 copy; and without any output (C and Ch both NULL) the C ABI refuses."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3, RadmmmError
    monkeypatch.setenv("RADMMM_H3W_MB", "7")
    W, B, T = 512, 3, 300
    N = B * T
    gen = torch.Generator().manual_seed(taps)
    x = torch.nn.functional.softplus(torch.randn(N, W, generator=gen)).to(DEV)
    w = (torch.randn(W, W, taps, generator=gen) * 0.03).to(DEV)
    bias = (torch.randn(W, generator=gen) * 0.1).to(DEV)
    lens = torch.tensor([300, 251, 170], dtype=torch.int32, device=DEV)
    Ah, Al = ops.split_f16(x, W, 1.0, W, 2, ops.X8_ACT_EXP)
    Wh, Wl, _ = ops.split_weight(w, None, W, nprod=2)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    base = dict(nprod=2, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / ops.W_SCALE, T=T, sat_flag=flag, Ah=Ah, Al=Al,
                lda_h=W, Bh=Wh, Bl=Wl, ldb_h=W, b_tap_stride_h=Wh.stride(0), ldc=W, M=N, N=W, K=W, taps=taps, dil=2 if taps > 1 else 1,
                sign=1, lens=lens, a_mask_mode=1, bias=bias, pconv=1 if taps > 1 else 0, ratio_taps=taps, ratio_dil=2 if taps > 1 else 1,
                postmask=1, act=1, ldch=W, ch_scale=1.0, split_fmt=ops.SPLIT_X8A, ch_x8_exp=ops.X8_ACT_EXP)
    out = {}
    for with_c in (True, False):
        C = torch.empty(N, W, device=DEV) if with_c else None
        Ch, Cl = ops._halves(N, W, like=x)
        Ch.fill_(float("nan")), Cl.zero_()
        rowgemm_h3(C=C, Ch=Ch, Cl=Cl, **base)
        torch.cuda.synchronize()
        out[with_c] = (Ch, Cl)
    assert torch.equal(_bits(out[True][0]), _bits(out[False][0])) and torch.isfinite(out[False][0].float()).all()
    assert torch.equal(out[True][1].view(torch.int16), out[False][1].view(torch.int16))
    with pytest.raises(RadmmmError, match="C may be NULL only"):
        rowgemm_h3(C=None, **{k: v for k, v in base.items()})


@pytest.mark.gpu
@pytest.mark.parametrize("taps,K,N,mb", [(1, 1088, 512, 8), (1, 512, 1024, 7), (5, 512, 512, 8), (5, 512, 512, 7)])
def test_three_product_launches_on_the_slot_pinned_kernels(taps, K, N, mb, monkeypatch):
    """Round 5: the C-only launches of the three-f16-product scheme (FiLM convs of the spline flows, the context LSTM's
    projection) take rowgemm_one / rowgemm_win with the slot-C MFMA replaced by Al.Bh + Ah.Bl (four f16 MFMAs).  Same operands,
    same products, another fp32 summation order than rowgemm_h3d<*, 3>: equal to 1e-6 of the result's maximum, and both within
    3e-6 of a float64 reference.  Ragged batch, masked input rows, partial-conv ratio, bias, leaky ReLU."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    B, T, lens_l = 3, 300, [300, 251, 170]
    M = B * T
    gen = torch.Generator().manual_seed(taps * 100 + K)
    x = (torch.randn(M, K, generator=gen)).to(DEV)
    w = (torch.randn(N, K, taps, generator=gen) * 0.03).to(DEV)
    bias = (torch.randn(N, generator=gen) * 0.1).to(DEV)
    lens = torch.tensor(lens_l, dtype=torch.int32, device=DEV)
    xh, xl = ops.split_f16(x, K, 1.0, K, 3)
    Wh, Wl, _ = ops.split_weight(w, None, K, nprod=3)
    monkeypatch.setenv("RADMMM_H3W_MB", str(mb))
    dil = 2 if taps > 1 else 1
    base = dict(nprod=3, acc_scale=1.0 / ops.W_SCALE, T=T, Ah=xh, Al=xl, lda_h=K, Bh=Wh, Bl=Wl, ldb_h=K, b_tap_stride_h=Wh.stride(0),
                ldc=N, M=M, N=N, K=K, taps=taps, dil=dil, sign=1, lens=lens, a_mask_mode=1, bias=bias, pconv=1 if taps > 1 else 0,
                ratio_taps=taps, ratio_dil=dil, postmask=1, act=3)
    outs = {}
    for slot in ("0", "1"):
        monkeypatch.setenv("RADMMM_SLOT3", slot)
        C = torch.full((M, N), float("nan"), device=DEV)
        rowgemm_h3(C=C, **base)
        torch.cuda.synchronize()
        outs[slot] = C.cpu()
    scale = float(outs["0"].abs().max())
    assert torch.isfinite(outs["1"]).all() and float((outs["1"] - outs["0"]).abs().max()) <= 1e-6 * scale
    # float64 reference of the same conv (masked rows, partial-conv ratio, bias, leaky ReLU)
    xd = x.double().cpu().view(B, T, K).permute(0, 2, 1)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens_l)[:, None])[:, None].double()
    wd = w.double().cpu()
    pad = dil * (taps - 1) // 2
    raw = torch.nn.functional.conv1d(xd * mask, wd, None, padding=pad, dilation=dil)
    cnt = torch.nn.functional.conv1d(mask, torch.ones(1, 1, taps, dtype=torch.float64), padding=pad, dilation=dil)
    ratio = taps / (cnt + 1e-6) * cnt.clamp(0, 1) if taps > 1 else torch.ones_like(cnt)
    ref = torch.nn.functional.leaky_relu((raw * ratio + bias.double().cpu()[None, :, None]) * mask)
    ref = ref.permute(0, 2, 1).reshape(M, N)
    assert float((outs["1"].double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


@pytest.mark.gpu
def test_utterances_are_independent_and_gradients_add_up_beyond_the_bench_size(monkeypatch):
    """Size-independent properties at 5x the benchmark's rows (B = 32, T = 4000: 64 000 grouped frames, a 2000-step context
    LSTM; the CPU oracle would need minutes here).  The decoder has no cross-utterance term in its forward pass
    (decoders.py:168-205: convs, LSTM and couplings act per utterance, padding is masked), so
      * z of utterances [8k, 8k + 8) in the full batch  ==  z of the same eight utterances run alone (same padded length), and
      * the NLL is a sum over utterances divided by the batch's element count (loss.py:85-110), so the full batch's parameter
        gradients are the chunk gradients weighted by n_k / n:   grad = sum_k (n_k / n) grad_k.
    Both would break on a 32-bit offset overflow, a tile that reads across an utterance boundary, or a mask that depends on
    the position of an utterance inside the batch."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    B, T, CH = 32, 4000, 8
    cfg = S.DecoderConfig(**KW)
    dec = RADMMMFlow(use_accent=True, **KW)
    dec.load_state_dict(_T(S.procedural_decoder_state(S.decoder_state_shapes(cfg))))
    dec = dec.to(DEV).train()
    dec.precision_guard_every = 0
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(B, T, cfg, 77, ragged=True)).items()}
    b["lengths"] = b["lengths"] // 2 * 2          # even lengths: the loss's element count, sum(len) // 2, is then additive over chunks

    def run(sel):
        dec.zero_grad(set_to_none=True)
        sl = SequenceLength(b["lengths"][sel])
        out = dec(b["mel"][sel], b["spk"][sel], b["context"][sel], sl, b["f0"][sel], b["energy"][sel], b["accent"][sel])
        loss = crit(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        torch.cuda.synchronize()
        return out["z_mel"].detach().clone(), float(loss.detach()), {n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None}

    run(slice(0, B))                                                  # (first pass: the gradient scale settles)
    z_full, l_full, g_full = run(slice(0, B))
    assert np.isfinite(l_full) and bool(torch.isfinite(z_full).all())
    n = (b["lengths"] // 2).double()
    acc, l_acc, worst_z = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g_full.items()}, 0.0, 0.0
    for k in range(B // CH):
        sel = slice(k * CH, (k + 1) * CH)
        w = float(n[sel].sum() / n.sum())
        z_k, l_k, g_k = run(sel)
        worst_z = max(worst_z, float((z_k - z_full[sel]).abs().max()))
        l_acc += w * l_k
        for name, g in g_k.items():
            acc[name] += w * g.double()
    assert worst_z == 0.0, worst_z                                    # bit-identical rows, wherever the utterance sits in the batch
    assert abs(l_acc - l_full) <= 2e-6 * abs(l_full), (l_acc, l_full)
    gmax = max(float(v.norm()) for v in g_full.values())
    bad = {}
    for name, g in g_full.items():
        if float(g.norm()) < 1e-6 * gmax:
            continue
        r = float((acc[name] - g.double()).norm() / g.double().norm())
        if r > 3e-4:                                                  # two HIP runs with different gradient scales: FP8-cross noise only
            bad[name] = r
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,T,C", [(4, 3, 200, 1024), (2, 2, 77, 64), (3, 1, 64, 128)])
def test_one_pass_over_the_skip_gradient_writes_what_one_pass_per_layer_wrote(n, B, T, C):
    """radmmm_dact_mul_rows_multi (the res/skip layers' gQ_j = gOUT * softplus'(R_j) from ONE read of gOUT) against n calls
    of radmmm_dact_mul_transposed: identical split pairs (hi, 8-bit cross array), column sums equal up to the order of their
    64-frame partial blocks."""
    from rad_mmm_amd import ops
    from rad_mmm_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    N = B * T
    gout = (torch.randn(N, C, generator=g) * 3e-3).to(DEV)
    saved = [torch.nn.functional.softplus(torch.randn(N, C, generator=g) * 2).to(DEV) for _ in range(n)]
    flag = torch.zeros(4, dtype=torch.int32, device=DEV)
    fmt, ge, SG, act = L.SPLIT_X8A, 6, 256.0, L.ACT["softplus"]
    ref = []
    for j in range(n):
        yh, yl = ops._halves(N, C, like=gout)
        _, s = ops.dact_mul_transposed(gout, saved[j], C, B, T, act, SG, None, yh, yl, fmt, ge, flag)
        ref.append((yh, yl, s))
    dst = [ops._halves(N, C, like=gout) for _ in range(n)]
    sums = ops.dact_mul_rows_multi(gout, saved, C, B, T, act, SG, dst, [None] * n, fmt, ge, flag, [None] * n)
    torch.cuda.synchronize()
    for j in range(n):
        assert torch.equal(dst[j][0].view(torch.int16), ref[j][0].view(torch.int16)), j
        assert torch.equal(dst[j][1].view(torch.int16), ref[j][1].view(torch.int16)), j
        assert float((sums[j] - ref[j][2]).abs().max()) <= 2e-6 * float(ref[j][2].abs().max()), j


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nsrc", [("one", 3), ("one", 1), ("h3d", 2), ("generic", 3)])
def test_last_res_skip_layer_forms_the_skip_sum_from_the_earlier_outputs(kind, nsrc, monkeypatch):
    """radmmm_rowgemm_desc.c2_src (ABI 4): the launch's second output is ((src0 + src1) + src2) + y.  Against the running
    read-modify-write over the same layers (C2 += y per layer): the fp32 sum, its split pair and the layer's own output are
    bit-identical; with C2 == NULL the pair alone is written and is still the same.  One-tap slot-pinned kernel, per-tap
    tiles (MB 4) and the generic LDS-parking epilogue (odd ldc2 alignment is not a case: the flow step's arrays are dense)."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    monkeypatch.setenv("RADMMM_H3W_MB", "4" if kind == "h3d" else "7")
    W, B, T = 512, 3, 300
    N = B * T
    gen = torch.Generator().manual_seed(5 + nsrc)
    extra = dict(add=(torch.randn(N, W, generator=gen) * 1e-3).to(DEV), ldadd=W) if kind == "generic" else {}   # (an `add` input: generic epilogue)
    xs = [torch.nn.functional.softplus(torch.randn(N, W, generator=gen)).to(DEV) for _ in range(nsrc + 1)]
    ws = [(torch.randn(W, W, 1, generator=gen) * 0.03).to(DEV) for _ in range(nsrc + 1)]
    bs = [(torch.randn(W, generator=gen) * 0.1).to(DEV) for _ in range(nsrc + 1)]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)

    def launch(j, **kw):
        Ah, Al = ops.split_f16(xs[j], W, 1.0, W, 2, ops.X8_ACT_EXP)
        Wh, Wl, _ = ops.split_weight(ws[j], None, W, nprod=2)
        rowgemm_h3(nprod=2, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / ops.W_SCALE, T=T, sat_flag=flag, Ah=Ah, Al=Al,
                   lda_h=W, Bh=Wh, Bl=Wl, ldb_h=W, ldc=W, M=N, N=W, K=W, bias=bs[j], act=1, ldc2=W, ldc2h=W, c2h_scale=1.0,
                   split_fmt=ops.SPLIT_X8A, c2h_x8_exp=ops.X8_ACT_EXP, **extra, **kw)
    # reference: the running sum
    OUT = torch.empty(N, W, device=DEV)
    Rr = [torch.empty(N, W, device=DEV) for _ in range(nsrc + 1)]
    oh, ol = ops._halves(N, W, like=OUT)
    for j in range(nsrc + 1):
        last = j == nsrc
        launch(j, C=Rr[j], C2=OUT, c2_accum=1 if j else 0, C2h=oh if last else None, C2l=ol if last else None)
    # sources: the earlier layers write their own output only, the last one adds them up
    Rs = [torch.empty(N, W, device=DEV) for _ in range(nsrc + 1)]
    for j in range(nsrc):
        launch(j, C=Rs[j])
    OUT2 = torch.full((N, W), float("nan"), device=DEV)
    ph, pl = ops._halves(N, W, like=OUT)
    launch(nsrc, C=Rs[nsrc], C2=OUT2, c2_src=Rs[:nsrc], C2h=ph, C2l=pl)
    qh, ql = ops._halves(N, W, like=OUT)
    qh.fill_(float("nan"))
    Rn = torch.empty(N, W, device=DEV)
    launch(nsrc, C=Rn, C2=None, c2_src=Rs[:nsrc], C2h=qh, C2l=ql)
    torch.cuda.synchronize()
    assert torch.isfinite(OUT).all() and float(OUT.abs().max()) > 0
    for j in range(nsrc + 1):
        assert torch.equal(_bits(Rs[j]), _bits(Rr[j])), j
    assert torch.equal(_bits(Rn), _bits(Rr[nsrc]))
    assert torch.equal(_bits(OUT2), _bits(OUT))
    for a, b_ in ((ph, oh), (pl, ol), (qh, oh), (ql, ol)):
        assert torch.equal(a.view(torch.int16), b_.view(torch.int16))


@pytest.mark.gpu
def test_flow_step_without_an_fp32_skip_sum_is_bit_identical(monkeypatch):
    """The whole decoder step with the skip sum formed once (default) against four read-modify-writes of an fp32 OUT
    (RADMMM_RES_SRC=0) and one softplus' pass per res/skip layer (RADMMM_DACT_MULTI=0): identical z, loss and gradients."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    monkeypatch.setenv("RADMMM_DEBUG", "1")
    cfg = S.DecoderConfig(**KW)
    sd = _T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(6, 800, cfg, 9, ragged=True)).items()}
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RADMMM_RES_SRC", mode)
        monkeypatch.setenv("RADMMM_DACT_MULTI", mode)
        dec = RADMMMFlow(use_accent=True, **KW)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).train()
        dec.precision_guard_every = 0
        sl = SequenceLength(b["lengths"])
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        loss = RADMMMLoss(sigma=1.0, n_group_size=2)(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (out["z_mel"].detach().clone(), loss.detach().clone(), {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None})
    assert torch.equal(_bits(res["1"][0]), _bits(res["0"][0])) and torch.equal(_bits(res["1"][1]), _bits(res["0"][1]))
    # (bias gradients of the res/skip layers: the one-pass kernel adds its 64-frame partial blocks in another order)
    for n, g in res["1"][2].items():
        g0 = res["0"][2][n]
        if "res_skip_layers" in n and n.endswith("bias"):
            assert float((g - g0).abs().max()) <= 2e-6 * float(g0.abs().max()), n
        else:
            assert torch.equal(_bits(g), _bits(g0)), n


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,layout,bias,lda,ldc", [(32, 400, 0, True, 160, 160), (32, 400, 1, False, 160, 160), (1, 77, 0, False, 192, 164),
                                                     (3, 50, 1, True, 160, 176)])
def test_channel_mix_kernel_matches_the_generic_fp32_gemm_and_float64(B, T, layout, bias, lda, ldc):
    """rowgemm_mix.hip (the flow steps' 160 x 160 channel mix and its data gradient: weight resident in LDS, 40 k quads without a
    barrier) behind radmmm_rowgemm_f32: against the generic tiling (forced by an all-zero `add` input, which the mix kernel does
    not take) -- same ascending k order into one fp32 accumulator per output, so the results agree to the last bit -- and
    against float64."""
    from rad_mmm_amd._lib import rowgemm
    M, NK = B * T, 160
    g = torch.Generator().manual_seed(B * 1000 + T + layout)
    A = torch.zeros(M, lda)
    A[:, :NK] = torch.randn(M, NK, generator=g)
    W = torch.randn(NK, NK, generator=g) * 0.1
    bv = torch.randn(NK, generator=g) if bias else None
    Ad, Wd = A.to(DEV), W.to(DEV)
    bd = bv.to(DEV) if bias else None
    out = {}
    for kind in ("mix", "generic"):
        C = torch.full((M, ldc), float("nan"), device=DEV)
        extra = dict(add=torch.zeros(M, ldc, device=DEV), ldadd=ldc) if kind == "generic" else {}
        rowgemm(A=Ad, lda=lda, B=Wd, ldb=NK, b_layout=layout, C=C, ldc=ldc, M=M, N=NK, K=NK, T=T, bias=bd, **extra)
        torch.cuda.synchronize()
        out[kind] = C[:, :NK].cpu()
        assert torch.isnan(C[:, NK:]).all()                     # nothing written beyond the N columns
    ref = A[:, :NK].double() @ (W.double().t() if layout == 0 else W.double())
    if bias:
        ref = ref + bv.double()
    assert float((out["mix"].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert torch.equal(out["mix"].view(torch.int32), out["generic"].view(torch.int32))
