"""Data path: attention prior (beta-binomial anchor + scipy-zoom interpolation) and energy average.
CPU: the oracle restatement against the fixtures captured from the reference (tests/golden/make_golden.py --only prior).
GPU: csrc/prior.hip through rad_mmm_amd.data against the same fixtures and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import radmmm_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prior.npz"))
DEV = "cuda:0"


def _cases(prefix, key):
    i = 0
    while f"{prefix}.{i}.{key}" in G.files:
        yield G[f"{prefix}.{i}.{key}"], G[f"{prefix}.{i}.out"]
        i += 1


def test_oracle_prior_matches_reference_fixtures():
    for (p, m), ref in _cases("interp", "pm"):
        got = O.interpolated_prior(int(p), int(m))
        assert got.shape == ref.shape == (m, p)
        assert np.abs(got - ref).max() <= 1e-12 * ref.max()
    for (p, m, s), ref in _cases("bank", "pms"):
        got = O.beta_binomial_prior(int(p), int(m), float(s))
        assert np.abs(got - ref).max() <= 1e-12 * ref.max()


def test_oracle_energy_matches_reference_fixture():
    mel = torch.from_numpy(G["energy.mel"])
    for sc in (0, 1):
        assert np.array_equal(O.energy_average(mel, bool(sc)).numpy(), G[f"energy.out.{sc}"])


def test_oracle_prior_batch_padding():
    out = O.attention_prior_batch([3, 7], [11, 5])
    assert out.shape == (2, 11, 7) and out.dtype == np.float32
    assert np.all(out[0, :, 3:] == 0) and np.all(out[1, 5:] == 0)
    np.testing.assert_allclose(out[0, :, :3].sum(1), 1.0, rtol=1e-6)


@pytest.mark.gpu
def test_hip_betabinom_bank_matches_reference():
    from rad_mmm_amd import data as D
    for (p, m, s), ref in _cases("bank", "pms"):
        got = D.beta_binomial_prior_distribution(int(p), int(m), float(s)).cpu().numpy()
        # float64 lgamma on the device vs scipy's: 1e-10 of the row maximum (pmf tails reach 1e-30)
        assert np.abs(got - ref).max() <= 1e-10 * ref.max()


@pytest.mark.gpu
def test_hip_interpolated_prior_matches_reference():
    from rad_mmm_amd import data as D
    interp = D.BetaBinomialInterpolator()
    for (p, m), ref in _cases("interp", "pm"):
        got = interp(int(p), int(m)).cpu().numpy()
        assert got.shape == (m, p) and got.dtype == np.float32
        # fp32 output of a float64 computation: half an ulp of the value + the bank's 1e-10
        assert np.all(np.abs(got - ref) <= 6e-8 * np.abs(ref) + 1e-9 * ref.max())


@pytest.mark.gpu
def test_hip_prior_batch_full_size():
    """BASELINE batch shape (32 utterances up to 800 frames / 150 tokens, ragged): against the oracle, plus the
    size-independent properties (rows of the valid block sum to 1, padding is exactly zero)."""
    from rad_mmm_amd import data as D
    rng = np.random.default_rng(3)
    out_lens = [800] + list(rng.integers(200, 800, 31))
    in_lens = [150] + list(rng.integers(20, 150, 31))
    interp = D.BetaBinomialInterpolator()
    got = interp.batch(in_lens, out_lens)
    assert got.shape == (32, 800, 150)
    g = got.cpu().numpy()
    for b in (0, 5, 31):
        ref = O.interpolated_prior(int(in_lens[b]), int(out_lens[b]))
        assert np.all(np.abs(g[b, :out_lens[b], :in_lens[b]] - ref) <= 6e-8 * np.abs(ref) + 1e-9 * ref.max())
    for b in range(32):
        m, p = int(out_lens[b]), int(in_lens[b])
        np.testing.assert_allclose(g[b, :m, :p].sum(1, dtype=np.float64), 1.0, rtol=2e-6)
        assert np.all(g[b, m:] == 0) and np.all(g[b, :, p:] == 0)
    assert len(interp._bank) <= 32 and len(interp._bank) < 32 * 2      # anchors are shared between utterances
    again = interp.batch(in_lens, out_lens)
    assert torch.equal(again, got)


@pytest.mark.gpu
def test_hip_energy_average_matches_reference():
    from rad_mmm_amd import data as D
    mel = torch.from_numpy(G["energy.mel"]).to(DEV)
    for sc in (0, 1):
        got = D.get_energy_average(mel, bool(sc)).cpu().numpy()
        np.testing.assert_allclose(got, G[f"energy.out.{sc}"], rtol=2e-6, atol=1e-6)
    batch = torch.stack((mel, mel * 0.5 - 1.0))
    got = D.get_energy_average(batch).cpu()
    ref = O.energy_average(batch.cpu())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-6, atol=1e-6)
