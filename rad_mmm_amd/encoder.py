"""Text encoder (reference common.Encoder, common.py:424-493): 3 x
[weight-normed PartialConv1d k5 -> InstanceNorm1d(affine) -> ReLU -> dropout 0.5] and a packed
bi-LSTM.  The reference runs the conv stack in a per-utterance Python loop on slices
(`for b_ind in range(B)`, "TODO: improve speed"); here the padded batch goes through each layer in
one launch: the HIP partial conv masks frames >= len and re-normalises border windows exactly as a
conv over the slice does, a masked instance-norm kernel (csrc/instnorm.hip) takes per-utterance
statistics, and the LSTM is the fused HIP recurrence of rad_mmm_amd/lstm.py.  Same constructor
arguments and state_dict names (`convolutions.{i}.0.conv.{weight_g,weight_v,bias}`,
`convolutions.{i}.1.{weight,bias}`, `lstm.*` incl. the spectral-norm `_orig/_u/_v` entries)."""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from ._lib import lib, check, ptr, stream, amp_fwd, amp_bwd, fp32_region
from .common import _WNConv
from .lstm import bilstm


class InstanceNormReluFn(torch.autograd.Function):
    """y = relu(instance_norm(x) * w + b) over the valid frames of each item; x [B*T, ld] channels-last."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, weight, bias, lens, B, T, C, relu):
        y = torch.empty(B * T, x.shape[1], device=x.device, dtype=torch.float32)
        if x.shape[1] != C:
            y.zero_()
        mean = torch.empty(B, C, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(lib.radmmm_instnorm_fwd(ptr(x), x.shape[1], ptr(weight), ptr(bias), ptr(y), y.shape[1], ptr(mean), ptr(rstd),
                                      ptr(lens), B, T, C, 1e-5, 1 if relu else 0, stream()), "instnorm_fwd")
        ctx.dims = (B, T, C, relu)
        ctx.save_for_backward(x, y, weight, mean, rstd, lens)
        return y

    @staticmethod
    @amp_bwd
    def backward(ctx, gy):
        B, T, C, relu = ctx.dims
        x, y, weight, mean, rstd, lens = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        if x.shape[1] != C:
            gx.zero_()
        dwp = torch.empty(B, C, device=x.device, dtype=torch.float32)
        dbp = torch.empty_like(dwp)
        check(lib.radmmm_instnorm_bwd(ptr(gy), gy.shape[1], ptr(x), x.shape[1], ptr(y), y.shape[1], ptr(weight), ptr(mean),
                                      ptr(rstd), ptr(gx), gx.shape[1], ptr(dwp), ptr(dbp), ptr(lens), B, T, C,
                                      1 if relu else 0, stream()), "instnorm_bwd")
        return gx, dwp.sum(0), dbp.sum(0), None, None, None, None, None


class _ConvHolder(nn.Module):
    def __init__(self, c, k):
        super().__init__()
        self.conv = _WNConv(c, c, k, w_init_gain="relu")
        self.kernel_size = k


class Encoder(nn.Module):
    def __init__(self, encoder_n_convolutions=3, encoder_embedding_dim=512, encoder_kernel_size=5, lstm_norm_fn=None):
        super().__init__()
        assert encoder_embedding_dim % 4 == 0, "channels-last rows need a multiple of 4 channels"
        C = encoder_embedding_dim
        self.convolutions = nn.ModuleList(
            [nn.ModuleList([_ConvHolder(C, encoder_kernel_size), nn.InstanceNorm1d(C, affine=True)])
             for _ in range(encoder_n_convolutions)])
        self.lstm = nn.LSTM(C, C // 2, 1, batch_first=True, bidirectional=True)
        if lstm_norm_fn is not None:
            fn = nn.utils.spectral_norm if "spectral" in lstm_norm_fn else nn.utils.weight_norm
            self.lstm = fn(self.lstm, "weight_hh_l0")
            self.lstm = fn(self.lstm, "weight_hh_l0_reverse")

    def _lstm_weights_ready(self):
        """spectral/weight norm recompute `weight_hh_l0*` in forward pre-hooks of nn.LSTM.__call__;
        the HIP recurrence reads the attributes directly, so run the hooks by hand."""
        for hook in self.lstm._forward_pre_hooks.values():
            hook(self.lstm, ())

    @fp32_region
    def forward(self, x, in_lens, max_len=None):
        """x [B, C, L] padded text embeddings, in_lens [B] -> [B, max(in_lens), C].  max_len = max(in_lens) when the caller
        holds it on the host (otherwise one device -> host read)."""
        if not x.is_cuda:
            raise RuntimeError("rad_mmm_amd.encoder.Encoder runs on an MI355X only (no CPU path)")
        B, C, L = x.shape
        lens32 = in_lens.to(device=x.device, dtype=torch.int32).contiguous()
        ops.module_scale_box(self, new_forward_on=x.device)
        h = x.float().permute(0, 2, 1).reshape(B * L, C).contiguous()
        for holder, inorm in self.convolutions:
            c = holder.conv
            h = ops.conv_norm(h, c.weight_v, c.weight_g, c.bias, lens32, B, L, dil=1, partial=True, mask_out=True,
                              act="none", scale_box=ops.module_scale_box(self))
            h = InstanceNormReluFn.apply(h, inorm.weight, inorm.bias, lens32, B, L, C, True)
            h = F.dropout(h, 0.5, self.training)
        self._lstm_weights_ready()
        y = bilstm(self.lstm, h.view(B, L, C), lens32)
        return y[:, : (int(in_lens.max()) if max_len is None else int(max_len))]

    @fp32_region
    def infer(self, x):
        """single utterance / full-length batch (common.py:495-505)."""
        B, _, L = x.shape
        return self.forward(x, torch.full((B,), L, device=x.device, dtype=torch.long))
