"""Pin the CPU oracle (oracle/radmmm_oracle.py) to outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py,
which imports /root/reference unmodified in the build container.  These tests
need no GPU and do not read /root/reference.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err, sub
from oracle import radmmm_oracle as O

TOL = 2e-6      # oracle vs reference, same backend (torch CPU fp32): rounding-order only


def T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def test_affine_wn_forward_backward(golden):
    g = golden("affine_tiny.npz")
    sd = T(sub(g, "sd."))
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    z = torch.from_numpy(g["in.z"]).requires_grad_(True)
    ctx = torch.from_numpy(g["in.ctx"]).requires_grad_(True)
    lens = torch.from_numpy(g["in.lens"])
    mask = O.lengths_to_mask(lens)[:, None].float()
    zo, log_s = O.affine_coupling_forward(p, "", z, ctx, mask, 4, "tanh")
    assert rel_err(zo.detach(), g["out.z"]) < TOL
    assert rel_err(log_s.detach(), g["out.log_s"]) < TOL
    scalar = 0.5 * ((zo * mask) ** 2).sum() - (log_s * mask).sum()
    assert abs(float(scalar.detach()) - float(g["out.scalar"])) < 1e-5 * abs(float(g["out.scalar"]))
    scalar.backward()
    assert rel_err(z.grad, g["grad.z"]) < 1e-5
    assert rel_err(ctx.grad, g["grad.ctx"]) < 1e-5
    for n, gr in sub(g, "gradp.").items():
        assert rel_err(p[n].grad, gr) < 2e-5, n
    with torch.no_grad():
        wn = O.wn_forward(sd, "affine_param_predictor.", z[:, :4], ctx, mask, 4)
        assert rel_err(wn, g["out.wn"]) < TOL
        zi = O.affine_coupling_inverse(sd, "", zo, ctx, mask, 4, "tanh")
        assert rel_err(zi, g["out.z_inverse"]) < TOL


def test_inv1x1_lus(golden):
    g = golden("inv1x1_tiny.npz")
    sd = T(sub(g, "lus.sd."))
    p = {k: v.clone().requires_grad_(k in ("lower", "upper", "upper_diag")) for k, v in sd.items()}
    z = torch.from_numpy(g["lus.in.z"]).requires_grad_(True)
    zo, ld = O.inv1x1_lus_forward(p, "", z)
    assert rel_err(zo.detach(), g["lus.out.z"]) < TOL
    assert abs(float(ld.detach()) - float(g["lus.out.logdet"])) < 1e-6
    ((zo * torch.from_numpy(g["lus.cot"])).sum() + 3.0 * ld).backward()
    assert rel_err(z.grad, g["lus.grad.z"]) < 1e-5
    for n in ("lower", "upper", "upper_diag"):
        assert rel_err(p[n].grad, g["lus.gradp." + n]) < 1e-5, n
    W = O.lus_weight(sd, "")
    zi = torch.nn.functional.conv1d(zo.detach(), torch.inverse(W)[..., None])
    assert rel_err(zi, g["lus.out.z_inverse"]) < 1e-5


def test_inv1x1_whiten_init(golden):
    g = golden("inv1x1_tiny.npz")
    z = torch.from_numpy(g["wh.in.z"])
    lens = torch.from_numpy(g["wh.in.lens"])
    mean, ud, up = O.whiten_initialize(z, lens)
    assert rel_err(mean, g["wh.sd.input_mean"]) < TOL
    assert rel_err(ud, g["wh.sd.upper_diag"]) < 1e-5
    assert rel_err(up, g["wh.sd.upper"]) < 1e-5
    assert bool(g["wh.sd.initialized"])
    p = {"input_mean": torch.from_numpy(g["wh.sd.input_mean"]),
         "upper_diag": torch.from_numpy(g["wh.sd.upper_diag"]).requires_grad_(True),
         "upper": torch.from_numpy(g["wh.sd.upper"]).requires_grad_(True)}
    zz = z.clone().requires_grad_(True)
    zo, ld = O.inv1x1_whiten_forward(p, "", zz)
    assert rel_err(zo.detach(), g["wh.out.z"]) < TOL
    assert abs(float(ld.detach()) - float(g["wh.out.logdet"])) < 1e-5
    ((zo * torch.from_numpy(g["wh.cot"])).sum() + 2.0 * ld).backward()
    assert rel_err(zz.grad, g["wh.grad.z"]) < 1e-5
    assert rel_err(p["upper"].grad, g["wh.gradp.upper"]) < 1e-5
    assert rel_err(p["upper_diag"].grad, g["wh.gradp.upper_diag"]) < 1e-5


def test_piecewise_quadratic(golden):
    g = golden("spline_tiny.npz")
    x = torch.from_numpy(g["pq.in.x"]).requires_grad_(True)
    w = torch.from_numpy(g["pq.in.w"]).requires_grad_(True)
    v = torch.from_numpy(g["pq.in.v"]).requires_grad_(True)
    y, lj = O.unbounded_piecewise_quadratic_transform(x, w, v)
    assert rel_err(y.detach(), g["pq.out.y"]) < TOL
    assert np.abs(lj.detach().numpy() - g["pq.out.logj"]).max() < 2e-6
    cot = torch.from_numpy(g["pq.cot"])
    ((y * cot).sum() + (lj * torch.flip(cot, [0])).sum()).backward()
    assert rel_err(x.grad, g["pq.grad.x"]) < 1e-5
    assert rel_err(w.grad, g["pq.grad.w"]) < 1e-5
    assert rel_err(v.grad, g["pq.grad.v"]) < 1e-5
    xi, _ = O.unbounded_piecewise_quadratic_transform(y.detach(), w.detach(), v.detach(), inverse=True)
    assert rel_err(xi, g["pq.out.x_inverse"]) < 1e-5
    # elements outside [0,1) pass through with zero log-jacobian
    outside = (g["pq.in.x"] < 0) | (g["pq.in.x"] >= 1)
    assert outside.any()
    assert np.array_equal(y.detach().numpy()[outside], g["pq.in.x"][outside])
    assert (lj.detach().numpy()[outside] == 0).all()


def test_spline_coupling_layer(golden):
    g = golden("spline_tiny.npz")
    shapes = {k: tuple(int(i) for i in v) for k, v in sub(g, "sp.shape.").items()}
    sd = T(O.procedural_decoder_state(shapes, end_scale=0.05))
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in sd.items()}
    z = torch.from_numpy(g["sp.in.z"]).requires_grad_(True)
    ctx = torch.from_numpy(g["sp.in.ctx"]).requires_grad_(True)
    lens = torch.from_numpy(g["sp.in.lens"])
    mask = O.lengths_to_mask(lens)[:, None].float()
    zo, log_s = O.spline_coupling_forward(p, "", z, ctx, mask, 2, use_bn=True, training=True)
    assert rel_err(zo.detach(), g["sp.out.z"]) < 5e-6
    assert rel_err(log_s.detach(), g["sp.out.log_s"]) < 5e-5
    scalar = 0.5 * ((zo * mask) ** 2).sum() - (log_s * mask).sum()
    scalar.backward()
    assert rel_err(z.grad, g["sp.grad.z"]) < 5e-5
    assert rel_err(ctx.grad, g["sp.grad.ctx"]) < 5e-5
    for n, gr in sub(g, "sp.gradp.").items():
        # (a conv bias feeding batch-norm has an analytically zero gradient: noise only)
        assert np.abs(p[n].grad.numpy() - gr).max() < 1e-4 * np.abs(gr).max() + 1e-5, n
    for n, gn in sub(g, "sp.gradnorm.").items():
        assert abs(float(p[n].grad.norm()) - float(gn)) < 1e-4 * float(gn) + 1e-7, n


def test_flow_loss(golden):
    g = golden("flow_loss.npz")
    lens = torch.from_numpy(g["lens"])
    out = {"z_mel": torch.from_numpy(g["z"]),
           "log_s_list": [torch.from_numpy(g[f"log_s{i}"]) for i in range(3)],
           "log_det_W_list": list(torch.from_numpy(g["ldw"]))}
    lm, lp = O.decoder_loss(out, lens, 2, sigma=0.9)
    assert abs(float(lm) - float(g["loss_mel"])) < 1e-6 * abs(float(g["loss_mel"])) + 1e-7
    assert abs(float(lp) - float(g["loss_prior"])) < 1e-6 * abs(float(g["loss_prior"])) + 1e-7


def test_attention_mas_ctc(golden):
    g = golden("attention_tiny.npz")
    sd = T(sub(g, "sd."))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    q = torch.from_numpy(g["in.q"]).requires_grad_(True)
    k = torch.from_numpy(g["in.k"]).requires_grad_(True)
    prior = torch.from_numpy(g["in.prior"])
    in_lens = torch.from_numpy(g["in.in_lens"])
    out_lens = torch.from_numpy(g["in.out_lens"])
    kmask = ~O.lengths_to_mask(in_lens)[..., None]
    attn, lp = O.conv_attention_forward(p, "", q, k, kmask, prior)
    assert rel_err(attn.detach(), g["out.attn"]) < 5e-6
    assert rel_err(lp.detach(), g["out.logprob"]) < 5e-6
    hard = O.binarize_attention(torch.from_numpy(g["out.attn"]), in_lens, out_lens)
    assert np.array_equal(hard.numpy(), g["out.hard"])          # index work: bit-exact
    ctc = O.attention_ctc_loss(lp, in_lens, out_lens)
    assert abs(float(ctc) - float(g["out.ctc"])) < 1e-5 * abs(float(g["out.ctc"]))
    binl = O.attention_binarization_loss(hard, attn)
    assert abs(float(binl) - float(g["out.bin"])) < 1e-5 * abs(float(g["out.bin"]))
    (ctc + 0.7 * binl).backward()
    assert rel_err(q.grad, g["grad.q"]) < 5e-5
    assert rel_err(k.grad, g["grad.k"]) < 5e-5
    for n, gr in sub(g, "gradp.").items():
        assert rel_err(p[n].grad, gr) < 1e-4, n


def test_mas_cases_bit_exact(golden):
    g = golden("attention_tiny.npz")
    i = 0
    while f"mas.{i}.in" in g:
        out = O.mas_width1(g[f"mas.{i}.in"].copy())
        assert np.array_equal(out, g[f"mas.{i}.out"]), i
        assert (out.sum(1) == 1).all()
        i += 1
    assert i == 5


def test_mas_c_restatement_bit_exact(golden):
    """The plain-C MAS (oracle/mas_ref.c) against the reference's outputs and the numpy oracle."""
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "oracle")])
    g = golden("attention_tiny.npz")
    i = 0
    while f"mas.{i}.in" in g:
        out = O.mas_width1_c(g[f"mas.{i}.in"].copy())
        assert np.array_equal(out, g[f"mas.{i}.out"]), i
        i += 1
    r = np.random.Generator(np.random.PCG64(11))
    for (a, b) in [(200, 61), (97, 97), (800, 150)]:
        m = r.random((a, b)).astype(np.float32) ** 4 + 1e-6
        m /= m.sum(1, keepdims=True)
        assert np.array_equal(O.mas_width1_c(m.copy()), O.mas_width1(m.copy()))


def test_stft_magnitude(golden):
    g = golden("stft_tiny.npz")
    m = O.stft_magnitude(g["audio"], 64, 16, 64)
    assert m.shape == g["mag"].shape
    assert rel_err(m, g["mag"]) < 2e-6
    m2 = O.stft_magnitude(g["audio2"], 1024, 256, 1024)
    assert rel_err(m2, g["mag2"]) < 5e-6


@pytest.mark.parametrize("tag", ["22k", "16k"])
def test_mel_filterbank_matches_an_independent_implementation(golden, tag):
    """librosa (the reference's source of the matrix, audio_processing.py:124-125) is absent, so the pin is one step removed:
    tests/golden/mel_basis_hf.npz holds the Slaney filterbank of Hugging Face transformers' `audio_utils.mel_filter_bank`
    (norm = mel_scale = "slaney": the function Whisper's feature extractor uses to reproduce librosa's filters) for the two
    geometries of the reference's data configs (tests/golden/make_mel_basis.py).  The oracle's restatement AND the
    product's (rad_mmm_amd.audio_processing.mel_filterbank: host-side numpy) must agree with it to float32 rounding."""
    g = golden("mel_basis_hf.npz")
    sr, n_fft, n_mels, fmin, fmax = g[f"args_{tag}"]
    ref = g[f"mel_{tag}"]
    fb = O.mel_filterbank_slaney(int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax))
    assert fb.shape == ref.shape and np.abs(fb - ref).max() <= 1e-7 * np.abs(ref).max()
    from rad_mmm_amd.audio_processing import mel_filterbank
    fp = mel_filterbank(int(sr), int(n_fft), int(n_mels), float(fmin), float(fmax))
    assert fp.shape == ref.shape and np.abs(fp - ref).max() <= 1e-7 * np.abs(ref).max()


def test_mel_filterbank_selfcheck():
    """structural checks of the restated filterbank (its values are pinned by the test above)"""
    fb = O.mel_filterbank_slaney(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    freqs = np.linspace(0, 22050 / 2, 513)
    peak = freqs[fb.argmax(1)]
    assert (np.diff(peak) > 0).all()                      # centres increase
    assert fb[:, freqs > 8000.0 + 22050 / 1024].sum() == 0  # nothing above fmax
    # slaney norm: each triangle integrates to ~1 over Hz -> sum*df ~ 1
    area = fb.sum(1) * (freqs[1] - freqs[0])
    assert np.abs(area - 1.0).max() < 0.08
    # Slaney scale is linear below 1 kHz: first centres are 200/3 Hz-mel apart
    mel_pts = O._mel_to_hz_slaney(np.linspace(O._hz_to_mel_slaney(0.0), O._hz_to_mel_slaney(8000.0), 82))
    assert abs((mel_pts[2] - mel_pts[1]) - (mel_pts[1] - mel_pts[0])) < 1e-6
    # The one PUBLISHED value available offline: the example in librosa.filters.mel's docstring (0.8.x),
    # `melfb = librosa.filters.mel(22050, 2048)` -> `array([[ 0.   ,  0.016, ...,  0.   ,  0.   ], ...` (128 mels, fmax = sr/2,
    # Slaney scale and norm), i.e. melfb[0, 0] = 0.000 and melfb[0, 1] = 0.016 to the three decimals printed there.  A
    # weak pin (two rounded entries), but it does tie the restated
    # scale, the triangle construction and the Slaney area normalisation to librosa's own numbers.
    doc = O.mel_filterbank_slaney(22050, 2048, 128, 0.0, None)
    assert doc.shape == (128, 1025)
    assert round(float(doc[0, 0]), 3) == 0.0 and round(float(doc[0, 1]), 3) == 0.016
    assert float(doc[-1, -1]) == 0.0 and float(doc[1, 0]) == 0.0


@pytest.mark.parametrize("tag,n_flows,n_splines", [("cfg1", 2, 0), ("cfg2_small", 8, 0),
                                                   ("cfg3_small", 8, 0), ("cfg5_small", 4, 2)])
def test_full_decoder_procedural(golden, tag, n_flows, n_splines):
    """Full-width (WN 1024) decoder with procedural weights: oracle forward, NLL and
    gradients vs the reference run."""
    g = golden(f"decoder_{tag}.npz")
    kw = {k: (v.item() if v.shape == () else v) for k, v in sub(g, "cfg.").items()}
    cfg = O.DecoderConfig(**kw)
    assert cfg.n_flows == n_flows and cfg.n_splines == n_splines
    torch.set_num_threads(8)
    sd = T(O.procedural_decoder_state(O.decoder_state_shapes(cfg)))
    p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k
             and not k.endswith((".p", "lower_diag", "input_mean")) else v) for k, v in sd.items()}
    b = T(O.synthetic_batch(int(g["B"]), int(g["T"]), cfg, 1234, bool(g["ragged"])))
    assert np.array_equal(b["lengths"].numpy(), g["lengths"])
    mel = b["mel"].clone().requires_grad_(True)
    ctx = b["context"].clone().requires_grad_(True)
    out = O.decoder_forward(p, cfg, mel, b["spk"], ctx, b["lengths"], b["f0"], b["energy"], b["accent"])
    ul = b["lengths"] // cfg.n_group_size
    mask = O.lengths_to_mask(ul)[:, None].float()
    zm = out["z_mel"].detach().numpy()
    m = mask.numpy().astype(bool)
    m3 = np.broadcast_to(m, zm.shape)
    assert rel_err(zm[m3], g["z_mel"][m3]) < 2e-5
    assert rel_err(zm, g["z_mel"]) < 2e-5                  # padded frames too (deterministic garbage)
    ld = torch.stack(out["log_det_W_list"]).detach().numpy()
    assert np.abs(ld - g["log_det_W"]).max() < 1e-4
    for i, ls in enumerate(out["log_s_list"]):
        s = float((ls * mask).sum())
        assert abs(s - float(g[f"log_s.{i}.masked_sum"])) < 2e-4 * max(1.0, abs(s)), i
        assert rel_err(ls[:, :4, :32].detach(), g[f"log_s.{i}.slice"]) < 5e-5, i
    assert rel_err(out["context_w_spkvec"][:, :8, :16].detach(), g["ctx_w_spkvec.slice"]) < 1e-5
    lm, lp = O.decoder_loss(out, b["lengths"], cfg.n_group_size)
    assert abs(float(lm) - float(g["loss_mel"])) < 1e-5 * abs(float(g["loss_mel"]))
    assert abs(float(lp) - float(g["loss_prior"])) < 1e-5 * abs(float(g["loss_prior"]))
    lm.backward()
    assert rel_err(mel.grad, g["grad.mel"]) < 1e-4
    assert rel_err(ctx.grad[:, :8, :32], g["grad.context.slice"]) < 1e-4
    # (absolute floor: a conv bias feeding batch-norm has an analytically zero gradient)
    for n, gn in sub(g, "gradnorm.").items():
        mine = float(p[n].grad.norm())
        assert abs(mine - float(gn)) < 2e-4 * float(gn) + 1e-8, (n, mine, float(gn))
    for n, gr in sub(g, "gradp.").items():
        assert np.abs(p[n].grad.numpy() - gr).max() < 2e-4 * np.abs(gr).max() + 1e-8, n
    for n, gr in sub(g, "gradslice.").items():
        assert np.abs(p[n].grad[:4, :8].numpy() - gr).max() < 2e-4 * np.abs(gr).max() + 1e-8, n


def test_regularisation_and_bce_losses_match_reference(golden):
    """Embedding regularisers (configs/RADMMM_model_config.yaml:49-61) and the voiced predictor's BCE loss:
    oracle restatement against values captured from the reference's loss.py (make_golden.py --only regloss)."""
    g = golden("regloss.npz")
    spk, acc = torch.from_numpy(g["spk"]), torch.from_numpy(g["acc"])
    sid, aid = torch.from_numpy(g["sid"]), torch.from_numpy(g["aid"])
    v, c = O.variance_covariance_reg(spk, 1.0)
    assert abs(float(v) - float(g["vc.variance"])) < 1e-6 and abs(float(c) - float(g["vc.covariance"])) < 1e-6 * float(g["vc.covariance"])
    v, c = O.variance_covariance_reg(acc, 2.0)
    assert abs(float(v) - float(g["vt.variance"])) < 1e-6 and abs(float(c) - float(g["vt.covariance"])) < 1e-6 * float(g["vt.covariance"])
    assert abs(float(O.min_cross_covariance(spk[sid], acc[aid], spk, acc)) - float(g["cc.tables"])) < 1e-6 * float(g["cc.tables"])
    assert abs(float(O.min_cross_covariance(spk[sid], acc[aid], None, None)) - float(g["cc.batch"])) < 1e-6 * float(g["cc.batch"])
    lens = torch.from_numpy(g["bce.lens"])
    mask = (torch.arange(11)[None, :] < lens[:, None]).unsqueeze(1)
    got = O.attribute_bce_loss(torch.from_numpy(g["bce.x_hat"]), torch.from_numpy(g["bce.x"]), mask)
    assert abs(float(got) - float(g["bce.loss"])) < 1e-6 * float(g["bce.loss"])
