"""GPU parity of the HIP bidirectional LSTM recurrence (rad_mmm_amd/lstm.py, csrc/lstm.hip)
against torch.nn.LSTM on the CPU (which IS the reference's context LSTM, models/radmmm.py:141-146),
on fixed-length and packed ragged batches: outputs and every gradient."""
import pytest
import torch
from torch import nn

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(lstm, x, lens):
    T = x.shape[1]
    if lens is None:
        return lstm(x)[0]
    packed = nn.utils.rnn.pack_padded_sequence(x, torch.tensor(lens), batch_first=True, enforce_sorted=False)
    out, _ = lstm(packed)
    return nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)[0]


@pytest.mark.parametrize("B,T,I,H,lens", [(3, 7, 12, 8, None), (5, 19, 40, 21, [19, 7, 1, 12, 19]),
                                          (32, 50, 96, 524, None), (34, 23, 30, 70, [23] * 20 + [5] * 14)])
def test_bilstm_matches_torch(B, T, I, H, lens):
    from rad_mmm_amd.lstm import bilstm
    g = torch.Generator().manual_seed(B * 100 + H)
    lstm = nn.LSTM(I, H, num_layers=1, batch_first=True, bidirectional=True)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.6)
    x = torch.randn(B, T, I, generator=g, requires_grad=True)
    gy = torch.randn(B, T, 2 * H, generator=g) * 1e-3
    y_ref = _ref(lstm, x, lens)
    (y_ref * gy).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in lstm.named_parameters()}
    gx_ref = x.grad.clone()

    dl = nn.LSTM(I, H, num_layers=1, batch_first=True, bidirectional=True).to(DEV)
    dl.load_state_dict(lstm.state_dict())
    xd = x.detach().to(DEV).requires_grad_(True)
    lens_d = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    y = bilstm(dl, xd, lens_d)
    assert rel_err(y.detach().cpu(), y_ref.detach()) < 2e-5
    if lens is not None:
        for b, n in enumerate(lens):
            assert torch.all(y[b, n:] == 0)
    (y * gy.to(DEV)).sum().backward()
    assert rel_err(xd.grad.cpu(), gx_ref) < 5e-5
    for n, p in dl.named_parameters():
        assert rel_err(p.grad.cpu(), ref_grads[n]) < 5e-5, n
