"""On-device pieces of the reference's data path (reference data.py).

The reference computes the attention prior and the energy average per utterance on CPU dataset
workers (scipy.stats.betabinom + scipy.ndimage.zoom; `mel.mean(0)`), caches priors on disk, and pads
them into the batch in DataCollate.  Here the same quantities come from libradmmm_hip.so
(csrc/prior.hip): anchor priors are built once per rounded size and kept on the device, a whole batch
is interpolated / renormalised / zero-padded in one launch.  Same names and argument meaning as
data.py; results are device tensors (fp32, as DataCollate's FloatTensor batch).  float64 arithmetic
inside, as scipy.  There is no CPU path."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from ._lib import lib, check, ptr, stream, RadmmmError


def _device(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RadmmmError("rad_mmm_amd.data needs a GPU (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def beta_binomial_prior_distribution(phoneme_count: int, mel_count: int, scaling_factor: float = 0.05,
                                     device=None) -> torch.Tensor:
    """data.py:90-102 -> float64 [mel_count, phoneme_count] on the device."""
    out = torch.empty(int(mel_count), int(phoneme_count), device=_device(device), dtype=torch.float64)
    check(lib.radmmm_betabinom_prior(int(phoneme_count), int(mel_count), float(scaling_factor), ptr(out), stream()),
          "betabinom_prior")
    return out


class BetaBinomialInterpolator:
    """data.py:61-88: anchor priors at sizes rounded to (round_mel_len_to, round_text_len_to), bilinear
    zoom (scipy.ndimage.zoom order=1 semantics) to the utterance's size, rows renormalised.  The bank
    (the reference's lru_cache) lives on the device."""

    def __init__(self, round_mel_len_to: int = 100, round_text_len_to: int = 20, scaling_factor: float = 0.05,
                 device=None):
        self.round_mel_len_to = round_mel_len_to
        self.round_text_len_to = round_text_len_to
        self.scaling_factor = scaling_factor
        self.device = _device(device)
        self._bank: Dict[Tuple[int, int], torch.Tensor] = {}

    @staticmethod
    def round(val, to):
        return max(1, int(round((val + 1) / to))) * to            # numpy.round: half to even, as Python's round

    def bank(self, bw: int, bh: int) -> torch.Tensor:
        t = self._bank.get((bw, bh))
        if t is None:
            t = self._bank[(bw, bh)] = beta_binomial_prior_distribution(bw, bh, self.scaling_factor, self.device)
        return t

    def batch(self, in_lens: Sequence[int], out_lens: Sequence[int]) -> torch.Tensor:
        """Padded [B, max(out_lens), max(in_lens)] fp32 prior of a batch (DataCollate, data.py:678-679,737-741);
        in_lens / out_lens are host integers (token and frame counts), as the dataset has them."""
        in_lens = [int(v) for v in in_lens]
        out_lens = [int(v) for v in out_lens]
        if len(in_lens) != len(out_lens) or not in_lens or min(in_lens) < 1 or min(out_lens) < 1:
            raise ValueError("in_lens / out_lens: same non-zero length, all counts >= 1")
        rows = []
        for p, m in zip(in_lens, out_lens):
            bh = self.round(m, self.round_mel_len_to)
            bw = self.round(p, self.round_text_len_to)
            rows.append([self.bank(bw, bh).data_ptr(), bh, bw, m, p])
        items = torch.tensor(rows, dtype=torch.int64).to(self.device)
        B, Tmax, Nmax = len(rows), max(out_lens), max(in_lens)
        out = torch.empty(B, Tmax, Nmax, device=self.device, dtype=torch.float32)
        check(lib.radmmm_prior_zoom_batch(ptr(items), B, ptr(out), Tmax, Nmax, stream()), "prior_zoom_batch")
        return out

    def __call__(self, p_count: int, m_count: int) -> torch.Tensor:
        """[m_count, p_count] fp32 prior of one utterance."""
        return self.batch([p_count], [m_count])[0]


def get_energy_average(mel: torch.Tensor, use_scaled_energy: bool = True) -> torch.Tensor:
    """data.py:363-366 (+ energy_avg_normalize :339-342): mel [n_mel, T] or [B, n_mel, T] fp32 -> [T] / [B, T]."""
    single = mel.dim() == 2
    m = (mel[None] if single else mel).float().contiguous()
    B, n_mel, T = m.shape
    out = torch.empty(B, T, device=m.device, dtype=torch.float32)
    check(lib.radmmm_energy_average(ptr(m), ptr(out), B, n_mel, T, 1 if use_scaled_energy else 0, stream()), "energy_average")
    return out[0] if single else out
