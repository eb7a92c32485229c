"""Alignment attention (reference common.ConvAttention, common.py:1188-1277).

Same constructor arguments, parameter names (`key_proj.0.conv.weight_g` ...) and
`forward(queries, keys, query_lens, mask, key_lens, attn_prior)` contract; the projections are
weight-normed convs on the HIP row-GEMM and the distance/softmax core is one fused kernel (no
[B, C, T1, T2] intermediate).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from ._lib import fp32_region
from .common import _WNConv


class _CN(nn.Module):
    """ConvNorm holder: parameters under `.conv` like the reference."""

    def __init__(self, cin, cout, k, gain="linear"):
        super().__init__()
        self.conv = _WNConv(cin, cout, k, w_init_gain=gain)
        self.kernel_size = k


class ConvAttention(nn.Module):
    def __init__(self, n_mel_channels=80, n_text_channels=512, n_att_channels=80, temperature=1.0):
        super().__init__()
        self.temperature = temperature
        # indices 0,2(,4) as in the reference's nn.Sequential (ReLUs sit at 1 and 3)
        self.key_proj = nn.ModuleDict({"0": _CN(n_text_channels, n_text_channels * 2, 3, "relu"),
                                       "2": _CN(n_text_channels * 2, n_att_channels, 1)})
        self.query_proj = nn.ModuleDict({"0": _CN(n_mel_channels, n_mel_channels * 2, 3, "relu"),
                                         "2": _CN(n_mel_channels * 2, n_mel_channels, 1),
                                         "4": _CN(n_mel_channels, n_att_channels, 1)})

    @staticmethod
    def _cl(x):
        """[B, C, T] -> contiguous channels-last rows [B*T, round_up(C,4)]"""
        B, C, T = x.shape
        y = x.float().permute(0, 2, 1)
        pad = (-C) % 4
        if pad:
            y = torch.nn.functional.pad(y, (0, pad))
        return y.reshape(B * T, C + pad).contiguous()

    def _proj(self, layers, x, B, T):
        n = len(layers)
        for i, key in enumerate(sorted(layers.keys(), key=int)):
            c = layers[key].conv
            x = ops.conv_norm(x, c.weight_v, c.weight_g, c.bias, None, B, T, dil=1, partial=False, mask_out=False,
                              act="relu" if i < n - 1 else "none", scale_box=ops.module_scale_box(self))
        return x

    @fp32_region
    def forward(self, queries, keys, query_lens=None, mask=None, key_lens=None, attn_prior=None):
        """queries [B, n_mel, T1], keys [B, n_text, T2]; mask: bool [B, T2, 1] True at PADDED text
        positions (only its lengths matter: it is rebuilt from key_lens, or from the mask itself);
        attn_prior [B, T1, T2].  Returns attn, attn_logprob, both [B, 1, T1, T2]."""
        if not queries.is_cuda:
            raise RuntimeError("rad_mmm_amd.attention.ConvAttention runs on an MI355X only (no CPU path)")
        B, _, T1 = queries.shape
        T2 = keys.shape[2]
        ops.module_scale_box(self, new_forward_on=queries.device)
        k = self._proj(self.key_proj, self._cl(keys), B, T2)
        q = self._proj(self.query_proj, self._cl(queries), B, T1)
        Ca = self.query_proj["4"].conv.weight_v.shape[0]
        q = q[:, :Ca].reshape(B, T1, Ca).contiguous()
        k = k[:, :Ca].reshape(B, T2, Ca).contiguous()
        in_lens = None
        if mask is not None:
            if key_lens is None:
                key_lens = (~mask[:, :, 0]).sum(1)
            in_lens = key_lens.to(torch.int32).contiguous()
        prior = attn_prior.float().contiguous() if attn_prior is not None else None
        attn, logprob = ops.AttentionCoreFn.apply(q, k, prior, in_lens, 0.0005)
        return attn[:, None], logprob[:, None]
