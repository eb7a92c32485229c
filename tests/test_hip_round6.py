"""Round-6 GPU tests (all through the C ABI):
  * the context-gradient accumulation of the affine flow steps survives backward traversals that reach only some steps
    (`inputs=` pruning, a gradient of an early-exit output, a second traversal of a retained graph);
  * dap_forward_many's shared channels-last rows never mix an attached context with its detached twin;
  * the kernel choice behind radmmm_rowgemm_h3_colsum_rows is the launcher's."""
import numpy as np
import pytest
import torch

DEV = torch.device("cuda:0")

KW = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
          n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
          scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True, n_conv_layers_per_step=4, n_flows=4)


def _T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def _bits(a):
    return a.detach().contiguous().view(torch.int32)


@pytest.mark.gpu
def test_context_gradient_survives_partial_backward_traversals():
    """Four affine flow steps, early exit in front of step 2.  A traversal pruned to a late step's parameter (`inputs=`: step 3
    only) leaves a partial context-gradient sum that nobody collects.  On the SAME graph, afterwards: (a) the full context
    gradient equals the one of a fresh forward pass -- the stale partial sum was not added; (b) the gradient of the early-exit
    channels alone (steps 1, 0 only) equals the same request on a fresh pass.  And the partial traversals are VALUES, not only
    repeatable: early + final = total within the split scheme's rounding (linearity).  (One traversal per graph through the
    context LSTM: its backward works in place and refuses a second run.)"""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    cfg = S.DecoderConfig(**KW)
    dec = RADMMMFlow(use_accent=True, **KW)
    dec.load_state_dict(_T(S.procedural_decoder_state(S.decoder_state_shapes(cfg))))
    dec = dec.to(DEV).train()
    assert dec.gemm_precision in ("f8x", "h3")
    b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(3, 96, cfg, 11, ragged=True)).items()}
    sl = SequenceLength(b["lengths"])
    late = dec.flows[3].coupling_tfn.affine_param_predictor.end.weight

    def fwd():
        c = b["context"].clone().requires_grad_()
        out = dec(b["mel"], b["spk"], c, sl, b["f0"], b["energy"], b["accent"])
        z = out["z_mel"]
        return c, z[:, :KW["n_early_size"]].square().sum(), z[:, KW["n_early_size"]:].square().sum()

    def close(a, ref):
        return float((a - ref).abs().max()) <= 2e-6 * float(ref.abs().max())

    c, early, final = fwd()
    g_total, = torch.autograd.grad(early + final, c)                               # fresh, all four steps
    c, early, final = fwd()
    gp, = torch.autograd.grad(early + final, late, retain_graph=True)              # pruned: step 3 only
    assert torch.isfinite(gp).all() and float(gp.abs().max()) > 0
    g2, = torch.autograd.grad(early + final, c)                                    # same graph, behind the partial traversal
    assert close(g2, g_total), float((g2 - g_total).abs().max() / g_total.abs().max())
    c, early, final = fwd()
    g_early_fresh, = torch.autograd.grad(early, c)                                 # steps 1, 0 only
    c, early, final = fwd()
    torch.autograd.grad(early + final, late, retain_graph=True)
    g_early, = torch.autograd.grad(early, c)
    assert close(g_early, g_early_fresh), float((g_early - g_early_fresh).abs().max() / g_early_fresh.abs().max())
    c, early, final = fwd()
    g_final, = torch.autograd.grad(final, c)
    assert float(g_early.abs().max()) > 0 and float(g_final.abs().max()) > 0
    err = float((g_early + g_final - g_total).norm() / g_total.norm())
    print(f"context gradient: |early + final - total| / |total| = {err:.2e}")
    assert err < 5e-4                       # (the traversals scale their split gradients independently)


@pytest.mark.gpu
def test_dap_forward_many_does_not_share_rows_between_a_context_and_its_detached_twin(monkeypatch):
    """the shared channels-last copy of the text encoding was keyed on (data_ptr, shape, stride) -- a tensor and its
    .detach() share that key, so a predictor fed the ATTACHED context next to one fed the DETACHED context would both use
    whichever rows were built first.  Two predictors, one of each: the attached one's loss must reach the context, the
    detached one's must not (its input gradient contribution is exactly what the attached predictor alone produces)."""
    import torch.nn.functional as F
    from rad_mmm_amd.attribute_predictors import ConvLSTMLinearDAP, dap_forward_many
    from rad_mmm_amd.common import SequenceLength
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)
    torch.manual_seed(5)
    mk = lambda: ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2, n_hidden=32,
                                   kernel_size=3, p_dropout=0.0).to(DEV).train()
    pa, pd = mk(), mk()
    for d in (pa, pd):           # converge spectral norm's power iteration, then freeze it: every pass sees the same W_hh
        for _ in range(20):
            for hook in d.feat_pred_fn.bilstm._forward_pre_hooks.values():
                hook(d.feat_pred_fn.bilstm, ())
        d.eval()
    B, T = 3, 40
    lens = SequenceLength(torch.tensor([40, 33, 17], device=DEV))
    spk = torch.randn(B, 16, device=DEV)
    ctx = torch.randn(B, 32, T, device=DEV, requires_grad=True)

    def call(ctx_):
        return ((None, ctx_, spk, lens), {})

    def run(order):
        items = {"a": (pa, call(ctx)), "d": (pd, call(ctx.detach()))}
        outs = dap_forward_many([items[k][0] for k in order], [items[k][1] for k in order])
        res = dict(zip(order, outs))
        (g,) = torch.autograd.grad(sum(o["x_hat"].square().sum() for o in outs), ctx)
        return g, res

    g_alone, = torch.autograd.grad(dap_forward_many([pa], [call(ctx)])[0]["x_hat"].square().sum(), ctx)
    assert float(g_alone.abs().max()) > 0
    for order in (("a", "d"), ("d", "a")):
        g, res = run(order)
        assert res["d"]["x_hat"].requires_grad          # (through the predictor's own parameters only)
        # (the two predictors' bi-LSTMs run as one merged recurrence here, alone as a single one: fp32 summation order; a leak of
        #  the detached predictor's loss would be O(1))
        assert float((g - g_alone).abs().max()) <= 2e-4 * float(g_alone.abs().max()), (order, float((g - g_alone).abs().max()))


@pytest.mark.gpu
def test_colsum_rows_follows_the_launchers_kernel_choice(monkeypatch):
    """radmmm_rowgemm_h3_colsum_rows must report the partial rows of the kernel radmmm_rowgemm_h3 really launches
    (one helper decides for both).  For every shape below: fill the scratch with NaN, launch with deferred column sums, and
    check that exactly the reported rows hold finite partials whose sum is the column sum of the output."""
    from rad_mmm_amd import ops
    from rad_mmm_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    for (B, T, K, N, nprod) in [(32, 400, 1024, 1024, 2), (4, 100, 256, 256, 2), (8, 512, 512, 512, 3), (2, 64, 64, 128, 3)]:
        M = B * T
        x = torch.randn(M, K, generator=g).to(DEV)
        v = (torch.randn(N, K, 1, generator=g) * 0.05).to(DEV)
        gg = torch.ones(N, 1, 1, device=DEV)
        xh, xl = ops.split_f16(x, K, 1.0, K, nprod, ops.X8_ACT_EXP)
        Wh, Wl, _ = ops.split_weight(v, gg, K, nprod=nprod)
        y = torch.empty(M, N, device=DEV)
        floats = int(ops.lib.radmmm_rowgemm_h3_colsum_scratch_floats(M, N))
        scratch = torch.full((floats,), float("nan"), device=DEV)
        lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
        kw = dict(Ah=xh, Al=xl, lda_h=K, Bh=Wh, Bl=Wl, ldb_h=K, C=y, ldc=N, M=M, N=N, K=K, T=T, lens=lens, nprod=nprod,
                  a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / ops.W_SCALE, colsum_scratch=scratch)
        rows = L.rowgemm_h3_colsum_rows(**kw)
        if rows <= 0:
            continue
        L.rowgemm_h3(**kw)
        torch.cuda.synchronize()
        part = scratch[: rows * N].view(rows, N)
        assert torch.isfinite(part).all(), (B, T, K, N, nprod, rows)
        assert not torch.isfinite(scratch[rows * N:]).any() or floats == rows * N
        ref = y.double().sum(0)
        err = float((part.double().sum(0) - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (B, T, K, N, nprod, err)


@pytest.mark.gpu
def test_res_skip_launches_on_a_side_stream_give_the_same_bits(monkeypatch):
    """RADMMM_RES_STREAM=1 (the A/B switch of DESIGN 4.15: res_skip[j] on a side stream beside in_layer[j+1]) launches the same
    kernels on the same operands: outputs and every gradient bit-identical to the one-stream step, two passes in a row (the
    second one re-uses the cached side stream and freshly recycled activation buffers)."""
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    kw = dict(KW, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd = _T(S.procedural_decoder_state(S.decoder_state_shapes(cfg)))
    b = {k: v.to(DEV) for k, v in _T(S.synthetic_batch(12, 800, cfg, 21, ragged=True)).items()}
    sl = SequenceLength(b["lengths"])
    crit = RADMMMLoss(n_group_size=2)
    monkeypatch.setenv("RADMMM_DEBUG", "1")
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RADMMM_RES_STREAM", mode)
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(sd)
        dec = dec.to(DEV).train()
        assert dec.gemm_precision == "f8x"
        for _ in range(2):
            dec.zero_grad(set_to_none=True)
            out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
            crit(out, None, sl, 0)["loss_mel"][0].backward()
        torch.cuda.synchronize()
        res[mode] = (out["z_mel"].detach().clone(), {n: p.grad.detach().clone() for n, p in dec.named_parameters() if p.grad is not None})
    assert torch.equal(_bits(res["0"][0]), _bits(res["1"][0]))
    assert res["0"][1].keys() == res["1"][1].keys() and len(res["0"][1]) > 60
    for n in res["0"][1]:
        assert torch.equal(_bits(res["0"][1][n]), _bits(res["1"][1][n])), n
