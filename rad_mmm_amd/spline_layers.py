"""Piecewise-quadratic spline coupling with its FiLM parameter predictor (reference
common.SplineTransformationLayer / FiLMStack / FiLMResBlock, common.py:706-773, 1006-1090, and
maskedbatchnorm1d.MaskedBatchNorm1d).  Same constructor keywords and parameter/buffer names; the
convs run on the HIP row-GEMM (ops.ConvNormFn), the block tail + masked batch-norm and the
spline transform are fused HIP kernels.  Only the forward (training) direction with
`use_quadratic=True` is built: that is what decoders.FlowStep wires (decoders.py:51-61).
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from . import ops
from ._lib import lib, check, ptr, stream, amp_fwd, amp_bwd
from .common import _PlainConv, _WNConv


class MaskedBatchNorm1d(nn.Module):
    """Parameter/buffer holder with the reference's names; the arithmetic is fused into
    ops_film (training-mode statistics over unmasked frames, maskedbatchnorm1d.py:77-109).
    `distributed_sync` (the reference's attribute, set by TTSModel.toggle_syncbnorm, tts_lightning_modules.py:241-243):
    with an initialised process group the masked sums [sum x, sum x^2, n] are all-reduced in forward and the two
    gradient sums in backward (maskedbatchnorm1d.py:88-95: an autograd-aware all_reduce), 6 KB per block each way."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.distributed_sync = False
        self.process_group = None
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _CN(nn.Module):
    def __init__(self, cin, cout, k, dilation=1):
        super().__init__()
        self.conv = _WNConv(cin, cout, k)
        self.dilation = dilation


# product scheme of the FiLM blocks' convs under RADMMM_PRECISION=f8x: 3 = three f16 products (the `end` conv, which emits the
# spline parameters, always runs three).  RADMMM_FILM_NPROD=2 is a MEASURED AND REJECTED experiment: configs[4]
# step 98.4 -> 95.6 ms with outputs still inside 1e-4 (z 2.4e-5, log_s 3.9e-6, NLL 5.4e-6), but the gradients leave their
# bars (d loss / d mel L2 3.2e-3 against 9.7e-4, worst gradient norm 4.9e-3 against 6.9e-4): the spline's log-Jacobian
# amplifies errors of its parameters in backward as well.
FILM_BLOCK_NPROD = int(os.environ.get("RADMMM_FILM_NPROD", "3"))


class FiLMPostFn(torch.autograd.Function):
    """out = 0.5 * (leaky(bn(h2) * (c1a + 1) + c1b) + x1r)  (common.py:728-735)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, h2, c1, x1r, bn_w, bn_b, mean, invstd, lens, T, n_valid, use_bn, sync_group=False):
        """sync_group: False = local statistics; None or a process group = synchronised statistics (mean / invstd /
        n_valid are then the GLOBAL ones and backward all-reduces its two gradient sums over that group)."""
        rows, C = h2.shape[0], x1r.shape[1]
        out = torch.empty(rows, C, device=h2.device, dtype=torch.float32)
        check(lib.radmmm_film_fwd(ptr(h2), h2.shape[1], ptr(c1), c1.shape[1], ptr(x1r), x1r.shape[1], ptr(mean),
                                  ptr(invstd), ptr(bn_w), ptr(bn_b), ptr(out), C, rows, C, 1 if use_bn else 0,
                                  stream()), "film_fwd")
        ctx.save_for_backward(h2, c1, bn_w if use_bn else h2, bn_b if use_bn else h2, mean if use_bn else h2,
                              invstd if use_bn else h2, lens if lens is not None else h2)
        ctx.meta = (T, float(n_valid), use_bn, lens is not None, C)
        ctx.sync_group = sync_group
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, gout):
        h2, c1, bn_w, bn_b, mean, invstd, lens = ctx.saved_tensors
        T, n_valid, use_bn, has_lens, C = ctx.meta
        rows = h2.shape[0]
        gout = gout.contiguous()
        gh2 = torch.zeros_like(h2) if h2.shape[1] != C else torch.empty_like(h2)
        gc1 = torch.zeros_like(c1) if c1.shape[1] != 2 * C else torch.empty_like(c1)
        gx1r = torch.empty(rows, C, device=h2.device, dtype=torch.float32)
        gw = torch.empty(C, device=h2.device) if use_bn else None
        gb = torch.empty(C, device=h2.device) if use_bn else None
        nscr = int(lib.radmmm_film_bwd_scratch_floats(rows, C))
        scratch = torch.empty(nscr, device=h2.device)
        if use_bn and ctx.sync_group is not False:
            import torch.distributed as dist
            check(lib.radmmm_film_bwd_sums(ptr(h2), h2.shape[1], ptr(c1), c1.shape[1], ptr(gout), gout.shape[1], ptr(mean),
                                           ptr(invstd), ptr(bn_w), ptr(bn_b), ptr(gw), ptr(gb), ptr(scratch), rows, C,
                                           stream()), "film_bwd_sums")
            dist.all_reduce(scratch[nscr - 2 * C:], op=dist.ReduceOp.SUM, group=ctx.sync_group)     # S = [2][C], in place
            check(lib.radmmm_film_bwd_apply(ptr(h2), h2.shape[1], ptr(c1), c1.shape[1], ptr(gout), gout.shape[1], ptr(mean),
                                            ptr(invstd), ptr(bn_w), ptr(bn_b), n_valid, T, ptr(lens) if has_lens else None,
                                            ptr(gh2), gh2.shape[1], ptr(gc1), gc1.shape[1], ptr(gx1r), C, ptr(scratch), rows,
                                            C, stream()), "film_bwd_apply")
            return gh2, gc1, gx1r, gw, gb, None, None, None, None, None, None, None
        check(lib.radmmm_film_bwd(ptr(h2), h2.shape[1], ptr(c1), c1.shape[1], ptr(gout), gout.shape[1],
                                  ptr(mean) if use_bn else None, ptr(invstd) if use_bn else None,
                                  ptr(bn_w) if use_bn else None, ptr(bn_b) if use_bn else None, n_valid, T,
                                  ptr(lens) if has_lens else None, ptr(gh2), gh2.shape[1], ptr(gc1), gc1.shape[1],
                                  ptr(gx1r), C, ptr(gw), ptr(gb), ptr(scratch), rows, C, 1 if use_bn else 0,
                                  stream()), "film_bwd")
        return gh2, gc1, gx1r, gw, gb, None, None, None, None, None, None, None


class FiLMResBlock(nn.Module):
    def __init__(self, in_channels, cond_channels, out_channels, kernel_size=1, stride=1, dilation=1, use_bn=True,
                 use_partial_padding=True):
        super().__init__()
        self.out_channels = out_channels
        self.use_partial_padding = use_partial_padding
        self.input_conv = _CN(in_channels, out_channels, 1)
        self.cond_conv = _CN(cond_channels, 2 * out_channels, 1)
        self.hidden_conv = _CN(out_channels, out_channels, kernel_size, dilation)
        self.use_bn = use_bn
        self.bn = MaskedBatchNorm1d(out_channels) if use_bn else None

    def forward_cl(self, x, cond, lens32, B, T, n_valid, scale_box=None):
        pp = self.use_partial_padding
        cn = lambda m, t, act: ops.conv_norm(t, m.conv.weight_v, m.conv.weight_g, m.conv.bias, lens32, B, T,
                                             dil=m.dilation, partial=pp, mask_out=True, act=act, scale_box=scale_box,
                                             nprod=FILM_BLOCK_NPROD, wgrad8=True)
        x1r = cn(self.input_conv, x, "leaky_relu")            # act(x1) is all that is used downstream
        c1 = cn(self.cond_conv, cond, "none")
        h2 = cn(self.hidden_conv, x1r, "none")
        C = self.out_channels
        mean = invstd = None
        sync_group = False
        if self.use_bn:
            bn = self.bn
            if self.training and n_valid > 1:
                with torch.no_grad():
                    s1 = ops.colsum(h2, C, 1, T, lens32)
                    s2 = ops.colsum(h2, C, 1, T, lens32, square=True)
                    if bn.distributed_sync and torch.distributed.is_available() and torch.distributed.is_initialized():
                        # maskedbatchnorm1d.py:88-95: [sum x, sum x^2, n] all-reduced at once; everything downstream uses
                        # the global mean / variance / frame count
                        sync_group = bn.process_group
                        st = torch.stack((s1, s2, torch.full_like(s1, float(n_valid))))
                        torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.SUM, group=sync_group)
                        s1, s2 = st[0], st[1]
                        n_valid = float(st[2, 0])            # one host sync per block and step (opt-in path)
                    mean = s1 / n_valid
                    var = s2 / n_valid - mean * mean
                    invstd = torch.rsqrt(var + bn.eps)
                    bn.num_batches_tracked += 1
                    f = bn.momentum
                    bn.running_mean.mul_(1 - f).add_(f * mean)
                    bn.running_var.mul_(1 - f).add_(f * var * n_valid / (n_valid - 1))
            else:
                # eval mode (maskedbatchnorm1d.py:110-118): running statistics, no update; the fused kernel
                # takes mean / invstd as inputs, so only their source changes (inference path: no autograd
                # through the statistics is needed)
                mean = bn.running_mean
                invstd = torch.rsqrt(bn.running_var + bn.eps)
        return FiLMPostFn.apply(h2, c1, x1r, bn.weight if self.use_bn else None, bn.bias if self.use_bn else None,
                                mean, invstd, lens32, T, n_valid, self.use_bn, sync_group)


class FiLMStack(nn.Module):
    def __init__(self, n_in_channels, n_context_dim, n_hidden_channels, n_out_channels, n_layers, kernel_size=5,
                 use_partial_padding=True, use_dilation=True, use_bn=True):
        super().__init__()
        assert kernel_size % 2 == 1
        self.n_layers = n_layers
        self.end = _PlainConv(n_hidden_channels, n_out_channels, zero=True)
        self.in_layers = nn.ModuleList()
        for i in range(n_layers):
            self.in_layers.append(FiLMResBlock(n_in_channels if i == 0 else n_hidden_channels, n_context_dim,
                                               n_hidden_channels, kernel_size=kernel_size,
                                               dilation=2 ** i if use_dilation else 1, use_bn=use_bn))

    def forward_cl(self, x, cond, lens32, B, T, n_valid, scale_box=None):
        for blk in self.in_layers:
            x = blk.forward_cl(x, cond, lens32, B, T, n_valid, scale_box)
        return ops.conv_norm(x, self.end.weight, None, self.end.bias, None, B, T, dil=1, partial=False, mask_out=False,
                             act="none", scale_box=scale_box)


class PQSplineFn(torch.autograd.Function):
    """Element-wise piecewise-quadratic transform on x [rows, ld] (first h columns, already in
    [0,1) units) with q [rows, h*(2K+1)]; returns y [rows, h] and the per-row sum of log-jacobians."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, q, h, K):
        rows = x.shape[0]
        y = torch.empty(rows, h, device=x.device, dtype=torch.float32)
        lj = torch.empty(rows + rows * h, device=x.device, dtype=torch.float32)
        check(lib.radmmm_pq_spline_fwd(ptr(x), x.shape[1], ptr(q), q.shape[1], ptr(y), h, ptr(lj), rows, h, K,
                                       stream()), "pq_spline_fwd")
        ctx.save_for_backward(x, q)
        ctx.hk = (h, K)
        return y, lj[:rows]

    @staticmethod
    @amp_bwd
    def backward(ctx, gy, glj):
        x, q = ctx.saved_tensors
        h, K = ctx.hk
        rows = x.shape[0]
        gx = torch.zeros_like(x) if x.shape[1] != h else torch.empty_like(x)
        gq = torch.empty_like(q)
        gy = gy.contiguous() if gy is not None else torch.zeros(rows, h, device=x.device)
        glj = glj.contiguous() if glj is not None else None
        check(lib.radmmm_pq_spline_bwd(ptr(x), x.shape[1], ptr(q), q.shape[1], ptr(gy), h, ptr(glj), ptr(gx),
                                       x.shape[1], ptr(gq), q.shape[1], rows, h, K, stream()), "pq_spline_bwd")
        return gx, gq, None, None


class SplineTransformationLayer(nn.Module):
    def __init__(self, n_mel_channels, n_context_dim, n_layers, with_dilation=True, kernel_size=5, scaling_fn="exp",
                 affine_activation="softplus", n_bins=8, left=-4, right=4, bottom=-4, top=4, use_quadratic=False,
                 use_bn=True):
        super().__init__()
        if not use_quadratic:
            raise Exception("only the piecewise-quadratic spline is built (decoders.py:60 hard-codes use_quadratic=True)")
        self.n_mel_channels = n_mel_channels
        self.half_mel_channels = n_mel_channels // 2
        self.left, self.right, self.bottom, self.top = left, right, bottom, top
        self.K = n_bins
        self.n_bins = 2 * n_bins + 1
        self.n_context_dim = n_context_dim
        self.param_predictor = FiLMStack(self.half_mel_channels, n_context_dim, 512,
                                         self.half_mel_channels * self.n_bins, n_layers, use_dilation=with_dilation,
                                         kernel_size=kernel_size, use_bn=use_bn)

    @torch.no_grad()
    def inverse_cl(self, z_cl, cond_cl, lens32, B, T, n_valid):
        """common.py:1040-1090 with inverse=True on channels-last rows [N, C]: z0 passes through,
        z1 -> spline^-1 with parameters predicted from z0."""
        C = self.n_mel_channels
        h = self.half_mel_channels
        z = z_cl[:, :C]
        z0 = torch.nn.functional.pad(z[:, :h], (0, (-h) % 4)).contiguous()
        D = cond_cl.shape[1]
        cond = cond_cl if D % 4 == 0 else torch.nn.functional.pad(cond_cl, (0, (-D) % 4))
        q = self.param_predictor.forward_cl(z0, cond.contiguous(), lens32, B, T, n_valid)
        nb = h * self.n_bins
        q = q[:, :nb].contiguous() if q.shape[1] != nb else q
        y = ((z[:, h: 2 * h] - self.bottom) / (self.top - self.bottom)).contiguous()
        x = torch.empty_like(y)
        check(lib.radmmm_pq_spline_inv(ptr(y), h, ptr(q), q.shape[1], ptr(x), h, y.shape[0], h, self.K, stream()),
              "pq_spline_inv")
        z1 = x * (self.right - self.left) + self.left
        return torch.cat((z[:, :h], z1, z[:, 2 * h:]), 1)

    def run(self, z_cl, cond_cl, lens32, W_eff, b_eff, B, T, n_valid, scale_box=None):
        """[1x1 mix -> FiLM predictor -> spline] on channels-last rows; returns z_out [N, ZLD] and
        log_s [N, 1] (common.py:1040-1090)."""
        ZLD = ops.ZLD
        N = B * T
        h = self.half_mel_channels
        z1 = ops.conv_norm(z_cl, W_eff.view(ZLD, ZLD, 1), None, b_eff, None, B, T, dil=1, partial=False,
                           mask_out=False, act="none")
        z0 = z1[:, : ops.round_up(h, 4)].contiguous() if h % 4 else z1[:, :h].contiguous()
        if h % 4:
            z0[:, h:] = 0
        D = cond_cl.shape[1]
        cond = cond_cl if D % 4 == 0 else torch.nn.functional.pad(cond_cl, (0, (-D) % 4))
        q = self.param_predictor.forward_cl(z0, cond.contiguous(), lens32, B, T, n_valid, scale_box)
        nb = h * self.n_bins
        q = q[:, :nb].contiguous() if q.shape[1] != nb else q
        x = ((z1[:, h: 2 * h] - self.left) / (self.right - self.left)).contiguous()
        y, logj = PQSplineFn.apply(x, q, h, self.K)
        z1o = y * (self.top - self.bottom) + self.bottom
        z_out = torch.cat((z1[:, :h], z1o, z1[:, 2 * h:]), 1)
        log_s = logj[:, None] + h * (math.log(self.top - self.bottom) - math.log(self.right - self.left))
        return z_out, log_s


def toggle_syncbnorm(module: nn.Module, use_syncbnorm: bool = False, process_group=None) -> None:
    """TTSModel.toggle_syncbnorm (tts_lightning_modules.py:241-243): switch every MaskedBatchNorm1d below `module` to
    statistics synchronised over the process group (default group if None)."""
    for md in module.modules():
        if isinstance(md, MaskedBatchNorm1d):
            md.distributed_sync = bool(use_syncbnorm)
            md.process_group = process_group
