"""Attribute predictors (reference attribute_predictors.py:27-197 + common.ConvLSTMLinear,
common.py:240-333): f0 / energy / duration / voiced predictors of config 4.
The reference runs the conv backbone in a per-utterance Python loop on slices; here the padded
batch goes through each layer in one launch (input masked to the utterance length = the zero
padding a conv over the slice sees), the spectral-normed bi-LSTM runs on the fused HIP recurrence,
only the final Linear and the target transforms are torch.  Same constructor arguments,
state_dict names and output dict {'x_hat', 'x'}."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from ._lib import fp32_region
from .common import SequenceLength, _WNConv
from .lstm import bilstm


class _ConvHolder(nn.Module):
    def __init__(self, cin, cout, k, gain="linear"):
        super().__init__()
        self.conv = _WNConv(cin, cout, k, w_init_gain=gain)


# weight gradients of the predictors' convs on the FP8-cross kernel (ops.conv_norm's wgrad8: a leaf -- outputs and data gradients keep
# three products).  Round 6 A/B: joint step 63.67 / 63.57 -> 63.17 / 63.25 ms; every parameter gradient stays inside the 5e-4
# elementwise bars of tests/test_attribute_predictors.py (B = 32, T = 400) and within 4e-6 (L2) of the CPU restatement's autograd in the joint
# step (tests/test_joint_step.py).  RADMMM_DAP_WGRAD8=0: the three-product radmmm_wgrad_rm.
DAP_WGRAD8 = os.environ.get("RADMMM_DAP_WGRAD8", "1") != "0"
PAD = 32        # channel padding of the conv inputs: Cin % 32 == 0 puts a conv on the split-f16 GEMM kernels (ops.conv_norm); the shipped
                # predictors have in_dim 520 and a 56-channel conv-stack input, which ran on the fp32-MFMA kernels at ~170 us a launch


def _rows(x):
    """[B, C, T] -> channels-last rows [B*T, round_up(C, 32)] (zero padded)"""
    B, C, T = x.shape
    y = x.float().permute(0, 2, 1)
    if C % PAD:
        y = F.pad(y, (0, (-C) % PAD))
    return y.reshape(B * T, -1).contiguous()


def _pad_in(v, cin_p):
    """weight_v [Cout, Cin, k] zero-padded along Cin (the weight norm's row norms do not change)"""
    return v if v.shape[1] == cin_p else F.pad(v, (0, 0, 0, cin_p - v.shape[1]))


class BottleneckLayer(nn.Module):
    def __init__(self, in_dim, reduction_factor=16, norm="weightnorm", non_linearity="leakyrelu", kernel_size=3,
                 use_partial_padding=True):
        super().__init__()
        if norm != "weightnorm":
            raise Exception("BottleneckLayer: only norm='weightnorm' is built (the reference's default)")
        self.reduction_factor = reduction_factor
        self.out_dim = int(in_dim / reduction_factor)
        self.leaky = non_linearity == "leakyrelu"
        if reduction_factor > 1:
            self.projection_fn = _ConvHolder(in_dim, self.out_dim, kernel_size)

    def forward_rows(self, x_rows, lens32, B, T, scale_box=None):
        if self.reduction_factor <= 1:
            return x_rows
        c = self.projection_fn.conv
        # ConvNorm without partial padding: conv of the padded batch as it is, then * mask (common.py:179-191)
        return ops.conv_norm(x_rows, _pad_in(c.weight_v, x_rows.shape[1]), c.weight_g, c.bias, lens32, B, T, dil=1, partial=False,
                             mask_out=True, act="leaky_relu" if self.leaky else "relu", scale_box=scale_box, wgrad8=DAP_WGRAD8)


class ConvLSTMLinear(nn.Module):
    def __init__(self, in_dim=None, out_dim=None, n_layers=2, n_channels=256, kernel_size=3, p_dropout=0.1,
                 lstm_type: Optional[str] = "bilstm", use_linear=True, use_weight_norm=True):
        super().__init__()
        if lstm_type != "bilstm" or not use_linear or not use_weight_norm:
            raise Exception("ConvLSTMLinear: only lstm_type='bilstm', use_linear, use_weight_norm is built")
        self.out_dim = out_dim
        self.p_dropout = p_dropout
        self.convolutions = nn.ModuleList(
            [_ConvHolder(in_dim if i == 0 else n_channels, n_channels, kernel_size, "relu") for i in range(n_layers)])
        self.bilstm = nn.LSTM(n_channels, n_channels // 2, 1, batch_first=True, bidirectional=True)
        self.bilstm = nn.utils.spectral_norm(self.bilstm, "weight_hh_l0")
        self.bilstm = nn.utils.spectral_norm(self.bilstm, "weight_hh_l0_reverse")
        self.dense = nn.Linear(n_channels, out_dim)

    def forward_rows(self, h, lens32, B, T):
        """h [B*T, ld] channels-last context -> x_hat [B, out_dim, T]"""
        return self.post_lstm(bilstm(self.bilstm, self.pre_lstm_rows(h, lens32, B, T, ops.module_scale_box(self, h.device)), lens32))

    def post_lstm(self, y):
        return self.dense(y).transpose(1, 2)

    def pre_lstm_rows(self, h, lens32, B, T, scale_box=None):
        """the conv stack: h [B*T, ld] channels-last context -> the bi-LSTM's input [B, T, C]; materialises the
        spectral-normed recurrent weights (the LSTM's own forward pre-hooks) on the way"""
        valid = (torch.arange(T, device=h.device)[None, :] < lens32[:, None]).reshape(B * T, 1)
        h = h * valid                                  # a conv over x[:, :len] sees zeros beyond len
        for holder in self.convolutions:
            c = holder.conv
            v = _pad_in(c.weight_v, h.shape[1]) if h.shape[1] % PAD == 0 else c.weight_v
            h = ops.conv_norm(h, v, c.weight_g, c.bias, lens32, B, T, dil=1, partial=False, mask_out=True,
                              act="relu", scale_box=scale_box, wgrad8=DAP_WGRAD8)
            h = F.dropout(h, self.p_dropout, self.training)
        for hook in self.bilstm._forward_pre_hooks.values():        # materialise the spectral-normed weight_hh_l0*
            hook(self.bilstm, ())
        C = self.convolutions[-1].conv.weight_v.shape[0]
        return h[:, :C].reshape(B, T, C).contiguous()


class AttributePredictor(nn.Module):
    """target transforms of attribute_predictors.py:54-133"""

    def __init__(self, target_scale=1, target_offset=0, log_target=False, normalize_target=False,
                 normalization_type=None):
        super().__init__()
        self.target_scale, self.target_offset, self.log_target = target_scale, target_offset, log_target
        self.normalize_target, self.normalization_type = normalize_target, normalization_type

    def tx_data(self, x, x_mean=None, x_std=None):
        if self.normalize_target:
            assert self.normalization_type is not None
            if self.normalization_type == "norm_lin_space":
                xr = x - x_mean[:, None].expand(-1, x.shape[1]) / x_std[:, None].expand(-1, x.shape[1])
                return torch.log(xr + 10) / 3
            if self.normalization_type == "norm_log_space":
                xr = (x - x_mean[:, None, None].expand(-1, 1, x.shape[2])) / x_std[:, None, None].expand(-1, 1, x.shape[2])
                return (xr + 5) / 10
            return x
        x = x * self.target_scale + self.target_offset
        return torch.log(x + 1) if self.log_target else x

    def inv_tx_data(self, x, x_mean=None, x_std=None):
        if self.normalize_target:
            if self.normalization_type == "norm_lin_space" and x_mean is not None and x_std is not None:
                return (torch.exp(x * 3) - 10) * x_std + x_mean
            if self.normalization_type == "norm_log_space" and x_mean is not None and x_std is not None:
                x = x * 10 - 5
                return x * x_std[:, None, None].expand(-1, 1, x.shape[2]) + x_mean[:, None, None].expand(-1, 1, x.shape[2])
            return x
        if self.log_target:
            x = torch.exp(x) - 1
        return (x - self.target_offset) / self.target_scale


class ConvLSTMLinearDAP(AttributePredictor):
    def __init__(self, n_speaker_dim=16, n_accent_dim=0, in_dim=512, out_dim=1, reduction_factor=16, n_backbone_layers=2,
                 n_hidden=256, kernel_size=3, p_dropout=0.25, target_scale=1, target_offset=0, log_target=False,
                 lstm_type: Optional[str] = "bilstm", use_speaker_embedding=True, use_accent_embedding=False,
                 normalize_target=False, normalization_type=None):
        super().__init__(target_scale, target_offset, log_target, normalize_target, normalization_type)
        self.use_speaker_embedding = bool(use_speaker_embedding)
        self.use_accent_embedding = bool(use_accent_embedding)
        self.bottleneck_layer = BottleneckLayer(in_dim=in_dim, reduction_factor=reduction_factor)
        d = self.bottleneck_layer.out_dim + (n_speaker_dim if use_speaker_embedding else 0) \
            + (n_accent_dim if use_accent_embedding else 0)
        self.feat_pred_fn = ConvLSTMLinear(in_dim=d, out_dim=out_dim, n_layers=n_backbone_layers, n_channels=n_hidden,
                                           kernel_size=kernel_size, p_dropout=p_dropout, lstm_type=lstm_type)

    @fp32_region
    def forward(self, x_target, text_enc, spk_emb, lens: SequenceLength, x_mean=None, x_std=None, accent_emb=None):
        x_target, xin, lens32 = self.forward_pre(x_target, text_enc, spk_emb, lens, x_mean, x_std, accent_emb)
        return {"x_hat": self.feat_pred_fn.post_lstm(bilstm(self.feat_pred_fn.bilstm, xin, lens32)), "x": x_target}

    def forward_pre(self, x_target, text_enc, spk_emb, lens: SequenceLength, x_mean=None, x_std=None, accent_emb=None, rows=None):
        """everything in front of the bi-LSTM: target transform, bottleneck, embeddings, conv stack
        -> (transformed target, LSTM input [B, T, C], lens32).  rows: _rows(text_enc) when the caller has it already (several
        predictors reading the same context)"""
        if not text_enc.is_cuda:
            raise RuntimeError("rad_mmm_amd.attribute_predictors runs on an MI355X only (no CPU path)")
        if x_target is not None:
            x_target = self.tx_data(x_target, x_mean, x_std)
        B, _, T = text_enc.shape
        lens32 = lens.lengths.to(torch.int32).contiguous()
        # gradient scale / saturation state of this predictor's split-f16 convs (ops.GradScale, one per module: with a
        # throw-away dict every conv's backward would read its gradient's maximum on the host -- 8 synchronisations per joint step)
        box = ops.module_scale_box(self, new_forward_on=text_enc.device)
        h = self.bottleneck_layer.forward_rows(_rows(text_enc) if rows is None else rows, lens32, B, T, box)
        parts = [h[:, : self.bottleneck_layer.out_dim].reshape(B, T, -1)]
        if self.use_speaker_embedding:
            parts.append(spk_emb.float()[:, None, :].expand(-1, T, -1))
        if self.use_accent_embedding:
            parts.append(accent_emb.float()[:, None, :].expand(-1, T, -1))
        ctx = torch.cat(parts, 2)
        if ctx.shape[2] % PAD:
            ctx = F.pad(ctx, (0, (-ctx.shape[2]) % PAD))
        return x_target, self.feat_pred_fn.pre_lstm_rows(ctx.reshape(B * T, -1).contiguous(), lens32, B, T, box), lens32

    @fp32_region
    def infer(self, text_enc, spk_emb, lens: SequenceLength, x_mean=None, x_std=None, accent_emb=None):
        res = self.forward(None, text_enc, spk_emb, lens, accent_emb=accent_emb)
        return self.inv_tx_data(res["x_hat"], x_mean, x_std)


@fp32_region
def dap_forward_many(daps, calls):
    """[dap(*args, **kwargs) for dap, (args, kwargs) in zip(daps, calls)] for predictors that read the same frames (the f0 /
    energy / voiced predictors of TTSModel.training_step, tts_lightning_modules.py:688-717: three ConvLSTMLinearDAP over
    context.detach() and out_lens): their bi-LSTMs -- same shape, T dependent steps each, latency-bound -- run as ONE
    block-diagonal recurrence (lstm.MergedBiLSTMFn) instead of one after the other; conv stacks and output layers stay per
    predictor.  Same values as the separate calls (the recurrence's fp32 summation order apart); predictors whose LSTMs
    cannot be merged (other sizes, other lengths) take their own launch."""
    from .lstm import can_merge, merged_bilstm
    shared = {}                                            # channels-last copy of a context several predictors read: made once

    def rows_of(t):
        # (a tensor and its .detach() share storage, shape and strides but not their place in the autograd graph: the rows are
        #  shared only between callers that hold the SAME graph node -- ADVICE r5)
        node = (id(t.grad_fn) if t.grad_fn is not None else id(t)) if t.requires_grad else 0
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), node)
        if key not in shared:
            shared[key] = _rows(t)
        return shared[key]
    pres = [d.forward_pre(*a, **k, rows=rows_of(a[1])) for d, (a, k) in zip(daps, calls)]
    lstms = [d.feat_pred_fn.bilstm for d in daps]
    xs = [p[1] for p in pres]
    same_lens = all(c[0][3] is calls[0][0][3] for c in calls)                 # the same SequenceLength object
    if same_lens and can_merge(lstms, xs) and os.environ.get("RADMMM_MERGE_DAP_LSTM", "1") != "0":
        ys = merged_bilstm(lstms, xs, pres[0][2])
    else:
        ys = [bilstm(l, x, p[2]) for l, x, p in zip(lstms, xs, pres)]
    return [{"x_hat": d.feat_pred_fn.post_lstm(y), "x": p[0]} for d, y, p in zip(daps, ys, pres)]


class AttributeRegressionLoss(nn.Module):
    """loss.py:233-250: masked MSE -> {prefix + 'loss': (value, weight)}"""

    def __init__(self, prefix: Optional[str] = None, weight=1.0):
        super().__init__()
        self.prefix, self.weight = prefix, weight

    def forward(self, model_output, in_lens, out_lens, global_step, mask=None):
        target, prediction = model_output["x"], model_output["x_hat"]
        if mask is None:
            mask = out_lens.mask.unsqueeze(1)
        # the reference gathers prediction[mask] / target[mask] (a boolean index: a device -> host read of the count per
        # predictor and step); the same sum over the selected positions as a masked sum needs none.  (`where`, not a
        # product: an unselected position may hold anything, e.g. log(0) of a duration target beyond the text)
        mask = mask.bool().expand_as(prediction)
        d = torch.where(mask, prediction - target, torch.zeros_like(prediction))
        loss = (d * d).sum() / mask.sum()
        return {self.prefix + "loss": (loss, self.weight)}
