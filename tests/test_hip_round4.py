"""Round-4 GPU parity tests (all through the C ABI):
  * the one-tap GEMM kernel (csrc/rowgemm_one.hip, rowgemm_onetap.h: three A stages, wave-private B, slot-pinned K loop)
    against the per-tap-tile kernel (rowgemm_h3d) -- same operands, same MFMAs in the same order per accumulator: the
    outputs must be IDENTICAL bit for bit, for every epilogue kind, tile height and a ragged batch with masked input rows."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _bits(a):
    return a.view(torch.int16 if a.dtype == torch.float16 else torch.int32)


@pytest.mark.parametrize("mb", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("kind,Kc,Nout", [("plain", 512, 512), ("split_masked", 512, 512), ("res", 512, 512), ("res_first", 512, 512),
                                          ("dgrad", 512, 512), ("plain", 1152, 320), ("split_masked", 512, 544)])
def test_one_tap_gemm_is_bit_identical_to_per_tap_tiles(mb, kind, Kc, Nout, monkeypatch):
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    B, T, lens = 3, 300, [300, 251, 170]
    N = B * T
    gen = torch.Generator().manual_seed(mb * 100 + Kc + Nout)
    grad = kind == "dgrad"
    x = ((torch.randn(N, Kc, generator=gen) * 3e-3) if grad else torch.nn.functional.softplus(torch.randn(N, Kc, generator=gen) * 2)).to(DEV)
    w = (torch.randn(Nout, Kc, 1, generator=gen) * 0.03).to(DEV)
    bias = (torch.randn(Nout, generator=gen) * 0.1).to(DEV)
    Hs = (torch.randn(N, Nout, generator=gen) * 2).to(DEV)
    acc0 = torch.randn(N, Nout, generator=gen).to(DEV)
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    S = 2048.0 if grad else 1.0
    xe = ops.X8_GRAD_EXP if grad else ops.X8_ACT_EXP
    Ah, Al = ops.split_f16(x, Kc, S, Kc, 2, xe)
    Wh, Wl, _ = ops.split_weight(w, None, Kc, nprod=2)
    monkeypatch.setenv("RADMMM_H3W_MB", str(mb))
    outs = {}
    for one in ("0", "1"):
        monkeypatch.setenv("RADMMM_ONE", one)
        Cf = torch.full((N, Nout), float("nan"), device=DEV)
        C2 = acc0.clone()
        Ch, Cl = ops._halves(N, Nout, like=x)
        Ch.fill_(float("nan")), Cl.zero_()
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        common = dict(nprod=2, a8_exp=xe, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / (S * ops.W_SCALE), T=T, sat_flag=flag,
                      Ah=Ah, Al=Al, lda_h=Kc, Bh=Wh, Bl=Wl, ldb_h=Kc, b_tap_stride_h=Wh.stride(0), C=Cf, ldc=Nout, M=N, N=Nout,
                      K=Kc, taps=1, dil=1, sign=1, lens=lens_d)
        if kind == "plain":
            rowgemm_h3(**common)
        elif kind == "split_masked":        # masked input rows, bias + softplus, split copy out (the start conv / a standalone ConvNorm)
            rowgemm_h3(a_mask_mode=1, bias=bias, postmask=1, act=1, Ch=Ch, Cl=Cl, ldch=Nout, ch_scale=1.0, split_fmt=ops.SPLIT_X8A,
                       ch_x8_exp=xe, **common)
        elif kind in ("res", "res_first"):  # res_skip forward: bias + softplus, C and the running sum C2 (accumulate / first layer: store)
            rowgemm_h3(bias=bias, act=1, C2=C2, ldc2=Nout, c2_accum=1 if kind == "res" else 0, **common)
        else:                               # res_skip data gradient: softplus' of the hidden state, partial-conv row scale, split copy
            rowgemm_h3(dact_src=Hs, lddact=Nout, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, Ch=Ch, Cl=Cl, ldch=Nout, ch_scale=S,
                       split_fmt=ops.SPLIT_X8A, ch_x8_exp=xe, **common)
        torch.cuda.synchronize()
        outs[one] = (Cf.cpu(), C2.cpu(), Ch.cpu(), Cl.cpu())
    assert bool(torch.isfinite(outs["1"][0]).all()) and float(outs["1"][0].abs().max()) > 0
    for name, a, b in zip(("C", "C2", "Ch", "Cl (8-bit cross array)"), outs["1"], outs["0"]):
        if name == "Ch" and kind in ("plain", "res", "res_first"):
            continue                                                    # (not written by these kinds)
        assert torch.equal(_bits(a), _bits(b)), name


def test_gradient_x8_exponent_adapts_to_saturation_reports(monkeypatch):
    """ops.GradScale: a backward pass whose 8-bit gradient parts saturate e4m3 reports by how many powers of two
    (flag bits 2..7 of the backward producers' word); the exponent of the following passes is lowered by that level and
    the reports stop -- no host synchronisation in between (the flags travel with the scale's event).  Started 10 powers
    of two too high on purpose."""
    import numpy as np
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import ops
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    monkeypatch.setenv("RADMMM_PRECISION", "f8x")
    monkeypatch.setenv("RADMMM_F8X_MIN_ROWS", "0")
    monkeypatch.delenv("RADMMM_CHECK_SATURATION", raising=False)
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, scaling_fn="tanh",
              affine_activation="softplus", use_partial_padding=True, n_conv_layers_per_step=4, n_flows=2)
    cfg = O.DecoderConfig(**kw)
    dec = RADMMMFlow(use_accent=True, **kw)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in O.procedural_decoder_state(O.decoder_state_shapes(cfg)).items()})
    dec = dec.to(DEV).train()
    dec.precision_guard_every = 0
    b = {k: torch.from_numpy(v).to(DEV) for k, v in O.synthetic_batch(4, 256, cfg, 5, ragged=True).items()}
    sl = SequenceLength(b["lengths"])
    crit = RADMMMLoss(n_group_size=2)
    dec._grad_scale = ops.GradScale()
    dec._grad_scale.x8_grad_exp = ops.X8_GRAD_EXP + 10
    exps, grads = [], []
    for it in range(6):
        for p in dec.parameters():
            p.grad = None
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        torch.cuda.synchronize()                              # (so that the next pass finds the published flags)
        exps.append(dec._grad_scale.grad_exp())
        grads.append(dec.flows[1].coupling_tfn.affine_param_predictor.in_layers[0].conv.weight_v.grad.clone())
    gs = dec._grad_scale
    assert exps[0] == ops.X8_GRAD_EXP + 10 and exps[-1] < exps[0], exps
    assert gs.x8_adaptations >= 1 and gs.x8_saturated_passes >= 1
    assert exps[-1] == exps[-2] == exps[-3], exps             # settled: the last passes did not saturate any more
    assert int(gs.flags[1].item()) & 2 == 0
    # the settled passes agree with each other bit for bit and differ from the saturated first pass only by the lost
    # cross terms of the clipped elements
    assert torch.equal(grads[-1], grads[-2])
    d = float((grads[0] - grads[-1]).abs().max() / grads[-1].abs().max())
    assert 0.0 <= d < 5e-3, d
