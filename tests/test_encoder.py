"""Text encoder.  CPU: oracle restatement vs vectors captured from the reference's
common.Encoder (eval mode; plain and spectral-normed LSTM).  GPU: the batched HIP encoder against
the same vectors, outputs and every gradient."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

HERE = os.path.dirname(__file__)


def _load(tag):
    g = np.load(os.path.join(HERE, "golden", f"encoder_{tag}.npz"))
    return {k: torch.from_numpy(np.asarray(g[k])) for k in g.files}


@pytest.mark.parametrize("tag", ["plain", "spectral"])
def test_oracle_encoder_matches_reference(tag):
    from oracle import radmmm_oracle as O
    g = _load(tag)
    p = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        out = O.encoder_forward(p, "", g["x"], g["lens"])
    assert out.shape == g["out"].shape
    assert rel_err(out, g["out"]) < 3e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plain", "spectral"])
def test_hip_encoder_matches_reference(tag):
    from rad_mmm_amd.encoder import Encoder
    g = _load(tag)
    dev = "cuda:0"
    enc = Encoder(3, 32, 5, lstm_norm_fn=None if tag == "plain" else "spectral")
    enc.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")})
    enc = enc.to(dev).eval()
    x = g["x"].to(dev).requires_grad_(True)
    out = enc(x, g["lens"].to(dev))
    assert rel_err(out.detach().cpu(), g["out"]) < 1e-4        # spectral: sigma = u^T W v is a 32x16 fp32 dot product on either side
    (out * g["gw"].to(dev)).sum().backward()
    assert rel_err(x.grad.cpu(), g["grad.x"]) < 1e-4
    # weight_g / conv bias feed an instance norm that cancels them: their gradients are analytically ~0
    # (only the eps in rsqrt(var + eps) leaks through) and come out of cancellation, so they get an absolute
    # floor relative to the largest parameter gradient of the model
    scale = max(float(g[k].abs().max()) for k in g if k.startswith("gradp."))
    for n, p in enc.named_parameters():
        ref = g["gradp." + n]
        assert rel_err(p.grad.cpu(), ref) < 2e-4 or float((p.grad.cpu() - ref).abs().max()) < 1e-4 * scale, n
