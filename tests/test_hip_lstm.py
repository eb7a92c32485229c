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
                                          (32, 50, 96, 524, None), (34, 23, 30, 70, [23] * 20 + [5] * 14),
                                          # B*T >= 4096: the gradient GEMMs run on the split-f16 kernels (lstm.py)
                                          (16, 256, 64, 36, None), (18, 230, 52, 40, [230] * 9 + [100, 64, 7] * 3)])
@pytest.mark.parametrize("persistent", ["0", "1"])
def test_bilstm_matches_torch(B, T, I, H, lens, persistent, monkeypatch):
    """persistent=1: the opt-in one-launch recurrence (grid barrier, sc1 exchange; csrc/lstm.hip)."""
    from rad_mmm_amd.lstm import bilstm
    monkeypatch.setenv("RADMMM_LSTM_PERSISTENT", persistent)
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


@pytest.mark.parametrize("norm", ["spectral", "weight"])
def test_decoder_context_lstm_with_normed_recurrent_weights(norm, monkeypatch):
    """context_lstm_norm != None: the HIP recurrence (weights materialised from torch's hooks) must
    agree with torch.nn.LSTM (MIOpen) on the same module, outputs and gradients."""
    import numpy as np
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=1,
              context_lstm_norm=norm)
    torch.manual_seed(3)
    dec = RADMMMFlow(use_accent=True, **kw).to(DEV).train()      # MIOpen's RNN backward needs training mode
    sd0 = {k: v.clone() for k, v in dec.state_dict().items()}      # spectral norm's power iteration updates u each call
    cfg = S.DecoderConfig(**{k: v for k, v in kw.items() if k != "context_lstm_norm"})
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(3, 40, cfg, seed=2, ragged=True).items()}
    sl = SequenceLength(b["lengths"])
    outs = {}
    for impl in ("hip", "miopen"):
        dec.lstm_impl = impl
        dec.load_state_dict(sd0)
        dec.zero_grad()
        y = dec.preprocess_context_cl(b["context"], b["spk"], sl, b["f0"], b["energy"], b["accent"])
        (y * y).sum().backward()
        outs[impl] = (y.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in dec.context_lstm.named_parameters()})
    assert rel_err(outs["hip"][0], outs["miopen"][0]) < 2e-5
    for n in outs["hip"][1]:
        assert rel_err(outs["hip"][1][n], outs["miopen"][1][n]) < 1e-4, n


@pytest.mark.parametrize("norm", ["spectral", "weight", None])
def test_remove_norms_keeps_the_context(norm):
    """RADMMMFlow.remove_norms (models/radmmm.py:150-166, "call before inference"): the parametrisation of the
    context LSTM's recurrent weights is stripped, the eval-mode context is unchanged, the state_dict now carries
    plain `weight_hh_l0*`."""
    import numpy as np
    import radmmm_synth as S
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.decoders import RADMMMFlow
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=64, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2, n_flows=1,
              context_lstm_norm=norm)
    torch.manual_seed(4)
    dec = RADMMMFlow(use_accent=True, **kw).to(DEV).eval()
    cfg = S.DecoderConfig(**{k: v for k, v in kw.items() if k != "context_lstm_norm"})
    b = {k: torch.from_numpy(v).to(DEV) for k, v in S.synthetic_batch(2, 40, cfg, seed=2, ragged=True).items()}
    sl = SequenceLength(b["lengths"])
    with torch.no_grad():
        before = dec.preprocess_context_cl(b["context"], b["spk"], sl, b["f0"], b["energy"], b["accent"]).clone()
        dec.remove_norms()
        after = dec.preprocess_context_cl(b["context"], b["spk"], sl, b["f0"], b["energy"], b["accent"])
    names = set(dec.context_lstm.state_dict())
    assert "weight_hh_l0" in names and "weight_hh_l0_reverse" in names
    assert not any(n.endswith(("_orig", "_g", "_v", "_u")) for n in names), names
    assert rel_err(after.cpu(), before.cpu()) < 1e-6
    dec.remove_norms()                                          # idempotent


def test_second_backward_through_the_lstm_is_refused():
    """BiLSTMFn overwrites its saved gate activations with their gradients: retain_graph + a second backward would
    silently compute garbage, so it raises."""
    from rad_mmm_amd.lstm import bilstm
    lstm = nn.LSTM(6, 5, num_layers=1, batch_first=True, bidirectional=True).to(DEV)
    x = torch.randn(2, 4, 6, device=DEV, requires_grad=True)
    y = bilstm(lstm, x, None).sum()
    y.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="ran twice"):
        y.backward()
