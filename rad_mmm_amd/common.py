"""MI355X-native counterparts of the reference's flow primitives (reference: common.py).

Same class names, constructor arguments, parameter/buffer names and shapes as the
reference so that checkpoints interchange; the arithmetic runs in
libradmmm_hip.so.  The modules exchange CHANNELS-LAST matrices internally
([B*T, ld] fp32, ld = ops.ZLD for flow variables); the reference's [B, C, T] layout
appears only at the decoder boundary (decoders.RADMMMFlow).
"""
from __future__ import annotations

import os

import math
from typing import Optional, Tuple

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from ._lib import ACT, SCALE, debug_env


def get_mask_from_lengths(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """bool [B, max_len]; reference common.py:105-116.  max_len: the largest length when the caller holds it on the host
    (without it one device -> host read)."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device)
    return ids < lengths.unsqueeze(1)


class SequenceLength:
    """Sequence lengths + mask (reference common.py:123-128).  Also keeps a host copy of the
    lengths so later stages do not need another device->host sync."""

    def __init__(self, lengths: torch.Tensor, lengths_host: Optional[torch.Tensor] = None):
        """lengths_host: the same lengths as a CPU tensor when the caller has them (the collate function builds them on the
        host before the batch is moved): saves the device -> host read, i.e. a synchronisation per step."""
        self.lengths = lengths.long()
        self.lengths_host = self.lengths.cpu() if lengths_host is None else lengths_host.long().cpu()
        max_len = int(self.lengths_host.max())
        ids = torch.arange(0, max_len, device=lengths.device)
        self.mask = ids < self.lengths.unsqueeze(1)


def _randn_like_ref(*shape):
    return torch.randn(*shape)


def _orthonormal_lu(c: int):
    """Random orthonormal W with det +1, LU-factored (reference common.py:511-515)."""
    W = torch.linalg.qr(torch.randn(c, c))[0]
    if torch.det(W) < 0:
        W[:, 0] = -W[:, 0]
    p, lower, upper = torch.linalg.lu(W)
    return p, lower, upper


class Invertible1x1ConvLUS(nn.Module):
    """W = P (L U) parameterisation (reference common.py:507-548).  `weight()` returns the
    [c, c] matrix and `log_det()` the log-determinant; the channel mix itself is fused into the
    flow step's first GEMM (ops.AffineFlowStepFn)."""

    def __init__(self, c, cache_inverse=False):
        super().__init__()
        p, lower, upper = _orthonormal_lu(c)
        self.register_buffer("p", p)
        self.register_buffer("lower_diag", torch.ones(c))
        self.lower = nn.Parameter(torch.tril(lower, -1))
        self.upper_diag = nn.Parameter(torch.diag(upper).clone())
        self.upper = nn.Parameter(torch.triu(upper, 1))
        self.cache_inverse = cache_inverse

    def weight(self) -> torch.Tensor:
        U = torch.triu(self.upper, 1) + torch.diag(self.upper_diag)
        Lm = torch.tril(self.lower, -1) + torch.diag(self.lower_diag)
        return self.p @ (Lm @ U)

    def weight_and_log_det(self, ldw: int, col_offset: int):
        """Zero-padded [ldw, ldw] channel-mix matrix reading input columns [col_offset, col_offset + c)
        and log|det W|, one HIP launch each way (ops.LUWeightFn) -- what the training step uses;
        `weight()` / `log_det()` are the stock-op restatement kept for inference-time inverses."""
        from . import ops
        return ops.LUWeightFn.apply(self.p, self.lower, self.lower_diag, self.upper, self.upper_diag, ldw, col_offset)

    def mean(self) -> Optional[torch.Tensor]:
        return None

    def log_det(self) -> torch.Tensor:
        return torch.sum(torch.log(torch.abs(self.upper_diag)))


class DataInitializedInvertible1x1Conv(nn.Module):
    """Upper-triangular whitening layer with data-dependent init (reference common.py:551-617)."""

    def __init__(self, c, cache_inverse=False):
        super().__init__()
        self.register_buffer("input_mean", torch.zeros(c, 1))
        self.register_buffer("initialized", torch.tensor(False))
        p, _, upper = _orthonormal_lu(c)
        self.register_buffer("p", p)
        self.upper_diag = nn.Parameter(torch.diag(upper).clone())
        self.upper = nn.Parameter(torch.triu(upper, 1))
        self.cache_inverse = cache_inverse
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.__dict__.pop("_init_seen", None))

    @torch.no_grad()
    def initialize(self, z_cl: torch.Tensor, lens: SequenceLength, T: int):
        """z_cl [B*T, ld] channels-last, first c columns active (reference common.py:569-591)."""
        c = self.upper_diag.shape[0]
        B = z_cl.shape[0] // T
        m = lens.mask[:, :T].reshape(B * T)
        data = z_cl[m][:, :c].t()                                    # [c, N_valid]
        N = data.shape[1]
        input_mean = data.mean(1, keepdim=True)
        cen = data - input_mean
        covar = (cen @ cen.t()) / N
        wm = torch.linalg.cholesky(torch.inverse(covar), upper=True).contiguous()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.broadcast(wm, 0)
            torch.distributed.broadcast(input_mean, 0)
        self.input_mean.copy_(input_mean)
        self.upper_diag.copy_(torch.diag(wm))
        self.upper.copy_(torch.triu(wm, 1))
        self.initialized.fill_(True)

    def weight(self) -> torch.Tensor:
        return torch.triu(self.upper, 1) + torch.diag(self.upper_diag)

    def mean(self) -> Optional[torch.Tensor]:
        return self.input_mean

    def log_det(self) -> torch.Tensor:
        return torch.sum(torch.log(torch.abs(self.upper_diag)))


class _WNConv(nn.Module):
    """Holder of a weight-normed conv's parameters under the reference's names
    (`weight_g`, `weight_v`, `bias`; torch weight_norm, dim 0)."""

    def __init__(self, cin, cout, k, w_init_gain="linear"):
        super().__init__()
        w = torch.empty(cout, cin, k)
        nn.init.xavier_uniform_(w, gain=nn.init.calculate_gain(w_init_gain))
        self.weight_g = nn.Parameter(w.reshape(cout, -1).norm(dim=1).reshape(cout, 1, 1))
        self.weight_v = nn.Parameter(w)
        bound = 1.0 / math.sqrt(cin * k)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))


class _ConvNormHolder(nn.Module):
    """`in_layers.{i}` of the reference is a ConvNorm whose conv lives under `.conv`."""

    def __init__(self, cin, cout, k, dilation):
        super().__init__()
        self.conv = _WNConv(cin, cout, k)
        self.kernel_size = k
        self.dilation = dilation


class _PlainConv(nn.Module):
    def __init__(self, cin, cout, zero=True):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, 1))
        self.bias = nn.Parameter(torch.zeros(cout))
        if not zero:
            nn.init.xavier_uniform_(self.weight)


class WN(nn.Module):
    """Parameter container of the reference WN (common.py:776-835): start (weight-normed 1x1),
    n_layers x [in_layers.i.conv (weight-normed k5, dilation 2^i, partial conv) +
    res_skip_layers.i (weight-normed 1x1)], end (plain 1x1, zero-initialised).  The forward is
    executed inside ops.AffineFlowStepFn."""

    def __init__(self, n_in_channels, n_context_dim, n_layers, n_channels, kernel_size=5,
                 affine_activation="softplus", use_partial_padding=True, use_dilation=True):
        super().__init__()
        assert kernel_size % 2 == 1 and n_channels % 2 == 0
        self.n_layers = n_layers
        self.n_channels = n_channels
        self.affine_activation = affine_activation
        self.use_partial_padding = use_partial_padding
        self.use_dilation = use_dilation
        self.start = _WNConv(n_in_channels + n_context_dim, n_channels, 1)
        self.end = _PlainConv(n_channels, 2 * n_in_channels, zero=True)
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        for i in range(n_layers):
            dilation = 2 ** i if use_dilation else 1
            self.in_layers.append(_ConvNormHolder(n_channels, n_channels, kernel_size, dilation))
            self.res_skip_layers.append(_WNConv(n_channels, n_channels, 1))

    def flat_params(self):
        ins, res = [], []
        for i in range(self.n_layers):
            c = self.in_layers[i].conv
            ins += [c.weight_v, c.weight_g, c.bias]
            r = self.res_skip_layers[i]
            res += [r.weight_v, r.weight_g, r.bias]
        return ([self.start.weight_v, self.start.weight_g, self.start.bias, self.end.weight, self.end.bias],
                ins + res)


class AffineTransformationLayer(nn.Module):
    """Affine coupling with a WN parameter predictor (reference common.py:1093-1185;
    affine_model='wavenet' is the only predictor any config uses and the only one built here)."""

    def __init__(self, n_mel_channels, n_context_dim, n_layers, affine_model="simple_conv",
                 with_dilation=True, kernel_size=5, scaling_fn="exp", affine_activation="softplus",
                 n_channels=1024, use_partial_padding=False):
        super().__init__()
        if affine_model != "wavenet":
            raise Exception(f"{affine_model} affine model not supported by rad_mmm_amd (configs use 'wavenet')")
        if scaling_fn not in SCALE:
            raise Exception(f"{scaling_fn} scaling fn not supported")
        if affine_activation not in ("softplus", "relu"):
            raise Exception(f"{affine_activation} activation not supported")
        self.affine_model = affine_model
        self.scaling_fn = scaling_fn
        self.n_mel_channels = n_mel_channels
        self.n_context_dim = n_context_dim
        self.affine_param_predictor = WN(
            n_mel_channels // 2, n_context_dim, n_layers=n_layers, n_channels=n_channels,
            kernel_size=kernel_size, affine_activation=affine_activation,
            use_partial_padding=use_partial_padding, use_dilation=with_dilation)
        for i, l in enumerate(self.affine_param_predictor.in_layers):
            assert l.dilation == 2 ** i, "ops.AffineFlowStepFn assumes dilation 2^i"

    def run(self, z_cl, cond_cl, lens32, W_eff, b_eff, B, T, precision="fp32", scale_box=None, ctx_acc=None, ctx_slot=0,
            flow_index=None):
        """Fused [1x1 mix -> WN -> coupling] on channels-last operands.  Returns z_out, log_s.
        precision "fp32": fp32 MFMA GEMMs; "h3": split-f16 GEMMs (fp32-class accuracy, f16 matrix
        cores) when the WN width allows it (multiple of 32); "f8x": the hi.hi product on the f16 cores and both cross
        terms in one block-scaled FP8 MFMA (2/3 of the MFMA time, z within 4e-5); "f16": the same kernels with the hi
        halves only = plain fp16 operands, fp32 accumulate (16-bit throughput mode, NOT within the 1e-4 bar)."""
        wn = self.affine_param_predictor
        head, layers = wn.flat_params()
        nprod = ops.NPROD.get(precision, 3)
        if nprod == 2 and B * T < int(debug_env("RADMMM_F8X_MIN_ROWS", "4096")):
            # below half a round of the one-workgroup-per-CU kernel's smallest tile the split GEMMs run on the 128 x 128
            # two-workgroups-per-CU kernel, which has the three-f16-product scheme only (B = 8, T = 800: 38.7 vs 40.9 ms)
            nprod = 3
        meta = dict(B=B, T=T, C=self.n_mel_channels, D=self.n_context_dim, n_layers=wn.n_layers,
                    act=ACT[wn.affine_activation], scaling=SCALE[self.scaling_fn],
                    partial=bool(wn.use_partial_padding), scale_box=scale_box if scale_box is not None else {},
                    nprod=nprod, ctx_acc=ctx_acc, ctx_slot=ctx_slot, flow_index=flow_index)
        fn = ops.AffineFlowStepH3Fn if (precision in ops.NPROD and wn.n_channels % 32 == 0) else ops.AffineFlowStepFn
        return fn.apply(meta, z_cl, cond_cl, lens32, W_eff, b_eff, *head, *layers)
