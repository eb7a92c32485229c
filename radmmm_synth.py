"""Synthetic inputs and procedural (closed-form) random-init weights for the flow decoder.

There is no network for datasets or checkpoints, so benchmarks and parity tests run on
random-init weights of the named architecture and synthetic batches of the named shape
(SURVEY.md §8c-d).  Everything here is a pure function of names/shapes/seeds (numpy only), so
every rank, the golden-fixture generator and the tests produce bit-identical tensors.
No arithmetic of the hot path lives here.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class DecoderConfig:
    """Mirror of RADMMMFlow's ctor arguments that matter to the arithmetic
    (decoders.py:83-143; models/radmmm.py:30-101)."""

    def __init__(self, n_speaker_dim=16, n_accent_dim=8, n_text_dim=512, n_group_size=2,
                 n_mel_channels=80, n_f0_dims=1, n_energy_avg_dims=1, n_flows=8,
                 n_conv_layers_per_step=4, n_early_size=2, n_early_every=2,
                 scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True,
                 n_splines=0, use_bn=True, use_accent_emb_for_decoder=True,
                 context_w_f0_and_energy=True, use_context_lstm=True):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @property
    def lstm_in(self):
        n = (self.n_f0_dims + self.n_energy_avg_dims + self.n_text_dim) * self.n_group_size
        n += self.n_speaker_dim
        if self.use_accent_emb_for_decoder:
            n += self.n_accent_dim
        return n

    @property
    def lstm_hidden(self):
        n = self.n_speaker_dim + self.n_text_dim * self.n_group_size
        if self.use_accent_emb_for_decoder:
            n += self.n_accent_dim
        return int(n / 2)

    @property
    def cond_dims(self):
        """decoder_cond_dims (decoders.py:126-134): the bi-LSTM's output width, or its input's without the LSTM"""
        return 2 * self.lstm_hidden if self.use_context_lstm else self.lstm_in

    def flow_channels(self) -> List[int]:
        c = self.n_mel_channels * self.n_group_size
        out = []
        for i in range(self.n_flows):
            if i > 0 and i % self.n_early_every == 0:
                c -= self.n_early_size
            out.append(c)
        return out

    def exit_steps(self) -> List[int]:
        return [i for i in range(self.n_flows) if i > 0 and i % self.n_early_every == 0]


def procedural_tensor(shape: Sequence[int], salt: int, scale: float = 1.0) -> np.ndarray:
    """Closed-form pseudo-random fp32 tensor: a deterministic hash of the flat
    index (no RNG state, identical everywhere).  Values roughly uniform in
    [-scale, scale]."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    x = i + np.uint64((int(salt) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)   # wraps mod 2^64
    x ^= x >> np.uint64(30)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(27)
    x = (x * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return ((2.0 * u - 1.0) * scale).astype(np.float32).reshape(shape)


def _salt(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def procedural_decoder_state(shapes: Dict[str, Tuple[int, ...]], end_scale: float = 0.02
                             ) -> Dict[str, np.ndarray]:
    """Deterministic, well-conditioned values for every tensor of a decoder
    state_dict given only names and shapes.

    * conv / lstm weights ~ U(-a, a) with a = sqrt(3 / fan_in)
    * weight_g = 1 + small, biases small
    * `end` conv (zero-init in the reference, common.py:799-801) gets
      U(-end_scale, end_scale) so the coupling is not the identity
    * LUS factors: unit-ish diagonal, small off-diagonals, identity permutation
      (a valid member of the reference's parameterisation, common.py:529-531)
    * whitening layer: initialised=True, mean 2.5-ish (mel scale), upper as LUS
    """
    out: Dict[str, np.ndarray] = {}
    for name, shp in shapes.items():
        s = _salt(name)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "p":
            out[name] = np.eye(shp[0], dtype=np.float32)
        elif leaf == "lower_diag":
            out[name] = np.ones(shp, dtype=np.float32)
        elif leaf == "initialized":
            out[name] = np.array(True)
        elif leaf == "input_mean":
            out[name] = 2.5 + procedural_tensor(shp, s, 0.2)
        elif leaf == "upper_diag":
            d = 1.0 + procedural_tensor(shp, s, 0.25)
            sign = np.where(procedural_tensor(shp, s + 1, 1.0) > 0.6, -1.0, 1.0)
            out[name] = (d * sign).astype(np.float32)
        elif leaf in ("upper", "lower"):
            out[name] = procedural_tensor(shp, s, 0.5 / math.sqrt(shp[0]))
        elif leaf == "weight_g":
            out[name] = 1.0 + procedural_tensor(shp, s, 0.1)
        elif leaf == "num_batches_tracked":
            out[name] = np.array(0, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = np.zeros(shp, dtype=np.float32)
        elif leaf == "running_var":
            out[name] = np.ones(shp, dtype=np.float32)
        elif ".end." in name or name.startswith("end."):
            out[name] = procedural_tensor(shp, s, end_scale)
        elif ".bn." in name and leaf == "weight":
            out[name] = 1.0 + procedural_tensor(shp, s, 0.1)
        elif leaf.startswith("bias"):
            out[name] = procedural_tensor(shp, s, 0.05)
        else:  # weight_v / weight / lstm weight_*
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
            out[name] = procedural_tensor(shp, s, math.sqrt(3.0 / max(fan_in, 1)))
    return out


def decoder_state_shapes(cfg: DecoderConfig, wn_channels: int = 1024, film_hidden: int = 512
                         ) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of RADMMMFlow.state_dict() for `cfg` (verified
    against the reference by tests/golden/make_golden.py)."""
    sh: Dict[str, Tuple[int, ...]] = {}
    H, I = cfg.lstm_hidden, cfg.lstm_in
    for suf in ("", "_reverse") if cfg.use_context_lstm else ():
        sh[f"context_lstm.weight_ih_l0{suf}"] = (4 * H, I)
        sh[f"context_lstm.weight_hh_l0{suf}"] = (4 * H, H)
        sh[f"context_lstm.bias_ih_l0{suf}"] = (4 * H,)
        sh[f"context_lstm.bias_hh_l0{suf}"] = (4 * H,)
    D = cfg.cond_dims
    L = cfg.n_conv_layers_per_step
    for i, C in enumerate(cfg.flow_channels()):
        pre = f"flows.{i}.invtbl_conv."
        if i == 0:
            sh[pre + "input_mean"] = (C, 1)
            sh[pre + "initialized"] = ()
            sh[pre + "p"] = (C, C)
            sh[pre + "upper_diag"] = (C,)
            sh[pre + "upper"] = (C, C)
        else:
            sh[pre + "p"] = (C, C)
            sh[pre + "lower_diag"] = (C,)
            sh[pre + "lower"] = (C, C)
            sh[pre + "upper_diag"] = (C,)
            sh[pre + "upper"] = (C, C)
        h = C // 2
        if i < cfg.n_splines:
            q = f"flows.{i}.coupling_tfn.param_predictor."
            sh[q + "end.weight"] = (h * 65, film_hidden, 1)
            sh[q + "end.bias"] = (h * 65,)
            for j in range(L):
                cin = h if j == 0 else film_hidden
                for nm, co, ci, k in (("input_conv", film_hidden, cin, 1),
                                      ("cond_conv", 2 * film_hidden, D, 1),
                                      ("hidden_conv", film_hidden, film_hidden, 5)):
                    b = f"{q}in_layers.{j}.{nm}.conv."
                    sh[b + "bias"] = (co,)
                    sh[b + "weight_g"] = (co, 1, 1)
                    sh[b + "weight_v"] = (co, ci, k)
                if cfg.use_bn:
                    b = f"{q}in_layers.{j}.bn."
                    sh[b + "weight"] = (film_hidden,)
                    sh[b + "bias"] = (film_hidden,)
                    sh[b + "running_mean"] = (film_hidden,)
                    sh[b + "running_var"] = (film_hidden,)
                    sh[b + "num_batches_tracked"] = ()
        else:
            q = f"flows.{i}.coupling_tfn.affine_param_predictor."
            W = wn_channels
            sh[q + "start.bias"] = (W,)
            sh[q + "start.weight_g"] = (W, 1, 1)
            sh[q + "start.weight_v"] = (W, h + D, 1)
            sh[q + "end.weight"] = (C, W, 1)
            sh[q + "end.bias"] = (C,)
            for j in range(L):
                b = f"{q}in_layers.{j}.conv."
                sh[b + "bias"] = (W,)
                sh[b + "weight_g"] = (W, 1, 1)
                sh[b + "weight_v"] = (W, W, 5)
                b = f"{q}res_skip_layers.{j}."
                sh[b + "bias"] = (W,)
                sh[b + "weight_g"] = (W, 1, 1)
                sh[b + "weight_v"] = (W, W, 1)
    return sh


def synthetic_batch(B: int, T: int, cfg: DecoderConfig, seed: int = 1234, ragged: bool = False
                    ) -> Dict[str, np.ndarray]:
    """Synthetic decoder inputs: mel ~ N(2.5, 0.5^2) (already
    'scale_mel'-ed), context ~ N(0,1), spk/accent ~ N(0,1), f0 in [0,6) with 30%
    zeros, energy ~ U(0,1).  numpy PCG64(seed)."""
    r = np.random.Generator(np.random.PCG64(seed))
    d = {
        "mel": (2.5 + 0.5 * r.standard_normal((B, cfg.n_mel_channels, T))).astype(np.float32),
        "context": r.standard_normal((B, cfg.n_text_dim, T)).astype(np.float32),
        "spk": r.standard_normal((B, cfg.n_speaker_dim)).astype(np.float32),
        "accent": r.standard_normal((B, cfg.n_accent_dim)).astype(np.float32),
    }
    f0 = (6.0 * r.random((B, T))).astype(np.float32)
    f0[r.random((B, T)) < 0.3] = 0.0
    d["f0"] = f0
    d["energy"] = r.random((B, T)).astype(np.float32)
    if ragged:
        lens = np.sort(r.integers(int(0.6 * T), T + 1, size=B))[::-1].copy()
        lens[0] = T
    else:
        lens = np.full((B,), T)
    d["lengths"] = lens.astype(np.int64)
    # zero-pad like DataCollate (data.py:621-790)
    for b in range(B):
        L = int(lens[b])
        d["mel"][b, :, L:] = 0
        d["context"][b, :, L:] = 0
        d["f0"][b, L:] = 0
        d["energy"][b, L:] = 0
    return d
