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
