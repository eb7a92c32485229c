"""Optimizer step.  CPU: the oracle's RAdam + norm-clip restatement against vectors
captured from the reference's radam.RAdam and torch's clip_grad_norm_ (tests/golden/make_golden.py,
section "radam").  GPU: the fused flat HIP step against the same vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden", "radam_tiny.npz")
N_STEPS, N_PARAMS = 9, 4


def _load():
    g = np.load(GOLD)
    return {k: torch.from_numpy(g[k]) for k in g.files}


def test_oracle_radam_matches_reference_vectors():
    from oracle import radmmm_oracle as O
    g = _load()
    ps = [g[f"p0.{i}"].clone() for i in range(N_PARAMS)]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    branches = set()
    for k in range(N_STEPS):
        grads = [g[f"g.{k}.{i}"] for i in range(N_PARAMS)]
        total, clipped = O.clip_grad_norm(grads, 1.0)
        assert abs(float(total) - float(g[f"norm.{k}"])) < 1e-6 * float(g[f"norm.{k}"])
        branches.add(O.radam_scalars(k + 1, 1e-3, 0.9, 0.999)[0] >= 5)
        for i in range(N_PARAMS):
            O.radam_step(ps[i], clipped[i], ms[i], vs[i], k + 1, lr=1e-3, weight_decay=1e-6)
            assert torch.allclose(ps[i], g[f"p.{k}.{i}"], rtol=1e-6, atol=1e-7), (k, i)
    assert branches == {True, False}, "fixture must cover both the warm-up and the rectified branch"


@pytest.mark.gpu
def test_flat_radam_matches_reference_vectors():
    from rad_mmm_amd.optim import FlatRAdam
    g = _load()
    dev = "cuda:0"
    params = [(f"flows.0.p{i}" if i < 2 else f"other.p{i}", torch.nn.Parameter(g[f"p0.{i}"].clone().to(dev)))
              for i in range(N_PARAMS)]
    opt = FlatRAdam(params, lr=1e-3, weight_decay=1e-6)
    assert len(opt.buckets) == 2
    for k in range(N_STEPS):
        for i, (_, p) in enumerate(params):
            p.grad = g[f"g.{k}.{i}"].to(dev)
        total = opt.clip_grad_norm(1.0)
        assert abs(float(total) - float(g[f"norm.{k}"])) < 1e-5 * float(g[f"norm.{k}"])
        opt.step()
        for i, (_, p) in enumerate(params):
            assert torch.allclose(p.detach().cpu(), g[f"p.{k}.{i}"], rtol=2e-6, atol=2e-7), (k, i)
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["step"] == N_STEPS


@pytest.mark.gpu
def test_flat_radam_with_bucket_reducer_shares_gradient_flats():
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.optim import FlatRAdam
    dev = "cuda:0"
    torch.manual_seed(0)
    mod = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    ref.load_state_dict(mod.state_dict())
    red = BucketedGradReducer(mod)
    opt = FlatRAdam(mod.named_parameters(), lr=1e-2, reducer=red)
    assert all(not b["own_g"] for b in opt.buckets)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import radmmm_oracle as O
    ms = [torch.zeros_like(p) for p in ref.parameters()]
    vs = [torch.zeros_like(p) for p in ref.parameters()]
    x = torch.randn(11, 7, device=dev)
    for k in range(3):
        red.prepare()
        mod(x).square().sum().backward()
        red.finish()
        opt.step()
        ref.zero_grad()
        ref(x).square().sum().backward()
        with torch.no_grad():
            for p, m, v in zip(ref.parameters(), ms, vs):
                O.radam_step(p.data, p.grad, m, v, k + 1, lr=1e-2)
    for p, q in zip(mod.parameters(), ref.parameters()):
        assert rel_err(p.detach().cpu(), q.detach().cpu()) < 1e-5


@pytest.mark.gpu
def test_checkpoint_resume_continues_bit_exactly():
    """Checkpoint / resume (Lightning ModelCheckpoint saves module + optimizer state_dicts): two
    steps, save both, one more step  ==  fresh module + optimizer, load both, the same third step."""
    import io
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.optim import FlatRAdam
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=2)
    cfg = S.DecoderConfig(**kw)
    sd0 = {k: torch.from_numpy(np.asarray(v)) for k, v in S.procedural_decoder_state(S.decoder_state_shapes(cfg)).items()}
    dev = "cuda:0"
    crit = RADMMMLoss(sigma=1.0, n_group_size=2)
    batches = [{k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(3, 48, cfg, seed=20 + i, ragged=True).items()}
               for i in range(3)]

    def make(state=None, opt_state=None):
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(state if state is not None else sd0)
        dec = dec.to(dev).train()
        red = BucketedGradReducer(dec)
        opt = FlatRAdam(dec.named_parameters(), lr=1e-3, weight_decay=1e-6, reducer=red)
        if opt_state is not None:
            opt.load_state_dict(opt_state)
        return dec, red, opt

    def step(dec, red, opt, b):
        sl = SequenceLength(b["lengths"])
        red.prepare()
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        crit(out, None, sl, 0)["loss_mel"][0].backward()
        red.finish()
        opt.clip_grad_norm(1.0)
        opt.step()

    dec, red, opt = make()
    step(dec, red, opt, batches[0])
    step(dec, red, opt, batches[1])
    buf = io.BytesIO()
    torch.save({"model": dec.state_dict(), "optimizer": opt.state_dict()}, buf)       # through serialisation, as a checkpoint
    step(dec, red, opt, batches[2])
    want = {n: p.detach().clone() for n, p in dec.named_parameters()}

    buf.seek(0)
    ck = torch.load(buf, map_location="cpu")
    assert ck["optimizer"]["state"][0]["step"] == 2
    dec2, red2, opt2 = make(ck["model"], ck["optimizer"])
    step(dec2, red2, opt2, batches[2])
    for n, p in dec2.named_parameters():
        assert torch.equal(p.detach(), want[n]), n


@pytest.mark.gpu
def test_standard_loop_with_zero_grad_does_not_accumulate():
    """`opt.zero_grad(); loss.backward(); opt.step()` with NO reducer: autograd accumulates into p.grad, so zero_grad()
    has to reset those tensors -- two iterations must use each iteration's own gradient, not the running sum
    (checked against a second optimizer fed the per-iteration gradients by assignment)."""
    from rad_mmm_amd.optim import FlatRAdam
    dev = "cuda:0"
    torch.manual_seed(5)
    w0 = [torch.randn(7, 5), torch.randn(11)]
    x = [torch.randn(3, 5, device=dev), torch.randn(3, 5, device=dev)]

    def make():
        ps = [("flows.0.a", torch.nn.Parameter(w0[0].clone().to(dev))), ("misc.b", torch.nn.Parameter(w0[1].clone().to(dev)))]
        return ps, FlatRAdam(ps, lr=1e-2)

    def loss_of(ps, xi):
        return ((xi @ ps[0][1].t()) ** 2).sum() + (ps[1][1] ** 3).sum()

    ps_a, opt_a = make()
    ps_b, opt_b = make()
    for it in range(3):
        opt_a.zero_grad()                                    # the loop under test
        loss_of(ps_a, x[it % 2]).backward()
        opt_a.step()
        gs = torch.autograd.grad(loss_of(ps_b, x[it % 2]), [p for _, p in ps_b])   # reference: fresh gradients by assignment
        for (_, p), g in zip(ps_b, gs):
            p.grad = g
        opt_b.step()
        for (_, pa), (_, pb) in zip(ps_a, ps_b):
            assert torch.equal(pa.detach(), pb.detach()), it
    opt_a.zero_grad(set_to_none=False)
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for _, p in ps_a)
