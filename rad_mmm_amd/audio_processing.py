"""STFT -> mel -> log front end on the GPU (reference audio_processing.py:119-154, 192-255).

`TacotronSTFT(filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin,
mel_fmax).mel_spectrogram(y)` as in the reference, forward only (the reference runs it without
grad inside DataLoader workers).  The windowed DFT basis is built exactly like the reference's
conv1d weights; the mel filterbank restates librosa 0.8.0 `filters.mel` (Slaney scale, Slaney
norm), which the reference imports -- pinned to an independent librosa-validated implementation
(tests/golden/mel_basis_hf.npz: librosa itself is absent from the reference tree and the image).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import fp32_region


def _hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def windowed_dft_basis(n_fft: int, win_length: int) -> np.ndarray:
    """[2*(n_fft/2+1), n_fft] fp32: rows re then im of the DFT, times the centre-padded periodic
    hann window (audio_processing.py:200-223)."""
    cutoff = n_fft // 2 + 1
    fb = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])]).astype(np.float32)
    w = _hann_periodic(win_length)
    lp = (n_fft - win_length) // 2
    w = np.pad(w, (lp, n_fft - win_length - lp)).astype(np.float32)
    return basis * w[None, :]


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filterbank [n_mels, n_fft/2+1]."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    wts = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        wts[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (wts * enorm[:, None]).astype(np.float32)


class STFT(nn.Module):
    """Magnitude STFT (reference STFT.transform; inverse/griffin-lim are out of scope)."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        assert window == "hann" and win_length <= filter_length
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        self.register_buffer("forward_basis", torch.from_numpy(windowed_dft_basis(filter_length, win_length)))


class TacotronSTFT(nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                 sampling_rate=22050, mel_fmin=0.0, mel_fmax=None):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        self.register_buffer("mel_basis", torch.from_numpy(
            mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)))

    @fp32_region
    def mel_spectrogram(self, y: torch.Tensor) -> torch.Tensor:
        """y [B, S] in [-1, 1] -> [B, n_mel, 1 + S//hop] log-mel (clamp 1e-5)."""
        if not y.is_cuda:
            raise RuntimeError("rad_mmm_amd.audio_processing runs on an MI355X only (no CPU path)")
        assert float(y.min()) >= -1 and float(y.max()) <= 1
        s = self.stft_fn
        return ops.stft_mel(y.float(), s.forward_basis, self.mel_basis, s.filter_length, s.hop_length, 1e-5)
