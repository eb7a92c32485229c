"""Host-side operators over the C-ABI: allocation + launch sequencing + autograd wiring.

Every numeric step is a kernel of libradmmm_hip.so; torch supplies memory, streams and
the autograd graph.  Internal activation layout is channels-last ([B*T, ld] fp32), see
DESIGN.md.
"""
from __future__ import annotations

import collections
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import (ACT, SCALE, lib, check, ptr, stream, rowgemm, wgrad, f32c, amp_fwd, amp_bwd, split_opts, SPLIT_F16,
                   SPLIT_X8A, SPLIT_X8B)

ZLD = 160            # row pitch of every flow-variable matrix (n_mel*group padded, see decoders.py)


def _empty(*shape, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pick_splits(tiles: int, rows: int, slots: int = 512) -> int:
    """split-K factor for the weight-gradient GEMM.  The grid is tiles*S equal workgroups on
    `slots` = 256 CUs x 2 resident workgroups: pick the smallest S whose last round is >= 90 %
    full (a 1.25-round grid wastes 37 % of the machine), keeping >= 8 K-steps per split."""
    max_s = max(1, min(64, rows // (32 * 8)))
    best, best_eff = 1, 0.0
    for s in range(1, max_s + 1):
        rounds = tiles * s / slots
        eff = rounds / math.ceil(rounds)
        if eff >= 0.9:
            return s
        if eff > best_eff:
            best, best_eff = s, eff
    return best


# ---------------------------------------------------------------------------------------
# gradient sinks
# ---------------------------------------------------------------------------------------
# ddp.BucketedGradReducer keeps the gradients in flat buckets.  A backward node that asks grad_out(param)
# for its output buffer writes the gradient straight into the bucket and returns that view; with
# param.grad == None autograd's AccumulateGrad then adopts the view as .grad (no add kernel, no copy).
GRAD_SINKS = {}          # parameter data_ptr -> destination view (registered per step by the reducer)


# Early notification (round 3): a backward node that has finished writing SOME of its parameters' gradients into their
# sinks (a flow step after its upper WN layers, half a flow step before the node returns) tells the reducer, which may
# start that bucket's all-reduce right away.  Callbacks take a list of parameter data_ptrs.
GRAD_FINAL_HOOKS = []


def notify_grads_final(params) -> None:
    if GRAD_FINAL_HOOKS:
        ptrs = [t.data_ptr() for t in params if t is not None]
        for cb in list(GRAD_FINAL_HOOKS):
            cb(ptrs)


def grad_out(param: torch.Tensor) -> torch.Tensor:
    # one-shot: a second gradient for the same parameter before the reducer re-arms (a parameter used by two
    # nodes, a second backward without prepare()) gets its own tensor and is ADDED by autograd, never overwrites
    t = GRAD_SINKS.pop(param.data_ptr(), None)
    if t is not None and t.shape == param.shape and t.dtype == param.dtype:
        return t.view(t.shape)       # a fresh alias: AccumulateGrad only adopts a tensor nobody else references
    return torch.empty_like(param)


class LUWeightFn(torch.autograd.Function):
    """(W_eff [ldw, ldw], log|det W|) of the LUS invertible 1x1 conv (reference common.py:507-548): W = P (L U)
    placed at columns [col_offset, col_offset + c) of a zero matrix; one launch each way (csrc/lu_weight.hip)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, p, lower, lower_diag, upper, upper_diag, ldw, col_offset):
        c = upper_diag.shape[0]
        p, lower_c, ld, upper_c, ud = f32c(p), f32c(lower), f32c(lower_diag), f32c(upper), f32c(upper_diag)
        W = torch.empty(ldw, ldw, device=p.device, dtype=torch.float32)
        logdet = torch.empty((), device=p.device, dtype=torch.float32)
        check(lib.radmmm_lu_weight_fwd(ptr(p), ptr(lower_c), ptr(ld), ptr(upper_c), ptr(ud), c, ptr(W), ldw, col_offset,
                                       ptr(logdet), stream()), "lu_weight_fwd")
        ctx.save_for_backward(p, lower_c, ld, upper_c, ud)
        ctx.params = (lower, upper, upper_diag)
        ctx.dims = (c, ldw, col_offset)
        return W, logdet

    @staticmethod
    @amp_bwd
    def backward(ctx, gW, glogdet):
        p, lower_c, ld, upper_c, ud = ctx.saved_tensors
        lower, upper, upper_diag = ctx.params
        c, ldw, col_offset = ctx.dims
        gW = f32c(gW) if gW is not None else torch.zeros(ldw, ldw, device=p.device, dtype=torch.float32)
        gl, gu, gd = grad_out(lower), grad_out(upper), grad_out(upper_diag)
        check(lib.radmmm_lu_weight_bwd(ptr(p), ptr(lower_c), ptr(ld), ptr(upper_c), ptr(ud), c, ptr(gW), ldw, col_offset,
                                       ptr(f32c(glogdet)) if glogdet is not None else None, ptr(gl), ptr(gu), ptr(gd),
                                       stream()), "lu_weight_bwd")
        return None, gl, None, gu, gd, None, None


# ---------------------------------------------------------------------------------------
# thin launch helpers
# ---------------------------------------------------------------------------------------
def weightnorm_fwd(v: torch.Tensor, g: torch.Tensor, ldw: Optional[int] = None,
                   perm: Tuple[int, int, int] = (0, 0, 0)) -> Tuple[torch.Tensor, torch.Tensor]:
    """v [Cout, Cin, taps] (checkpoint layout) -> W [taps, Cout, ldw], inv_norm [Cout]."""
    Cout, Cin, taps = v.shape
    ldw = ldw or round_up(Cin, 4)
    W = (torch.zeros if ldw != Cin else torch.empty)(taps, Cout, ldw, device=v.device, dtype=torch.float32)
    inv = _empty(Cout, like=v)
    check(lib.radmmm_weightnorm_fwd(ptr(v), ptr(g), ptr(W), ptr(inv), Cout, Cin, taps, ldw,
                                    perm[0], perm[1], perm[2], stream()), "weightnorm_fwd")
    return W, inv


def weightnorm_bwd(v, g, inv, dW_slabs: torch.Tensor, ldw: int, perm=(0, 0, 0), poison: Optional[torch.Tensor] = None):
    """dW_slabs [S, taps, Cout, ldw] -> (dv like v, dg like g).  poison (device fp32[1], GradScale.poison): 0, or NaN when
    the pass's incoming gradient was not finite -- added to dg so that the non-finite pass stays visible to the caller."""
    Cout, Cin, taps = v.shape
    S = dW_slabs.shape[0]
    dv = grad_out(v)
    dg = grad_out(g)
    check(lib.radmmm_weightnorm_bwd(ptr(v), ptr(g), ptr(inv), ptr(dW_slabs), S,
                                    dW_slabs.stride(0), ptr(dv), ptr(dg), Cout, Cin, taps, ldw,
                                    perm[0], perm[1], perm[2], ptr(poison), stream()), "weightnorm_bwd")
    return dv, dg


def colsum(X: torch.Tensor, cols: int, row_weight: int = 0, T: int = 1,
           lens: Optional[torch.Tensor] = None, taps: int = 1, dil: int = 1, square: bool = False,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, ld = X.shape
    out = out if (out is not None and out.numel() == cols) else _empty(cols, like=X)
    scratch = _empty(int(lib.radmmm_colsum_scratch_floats(rows, cols)), like=X)
    check(lib.radmmm_colsum(ptr(X), ld, ptr(out), ptr(scratch), rows, cols, row_weight, T,
                            ptr(lens), taps, dil, 1 if square else 0, stream()), "colsum")
    return out


def wgrad_slabs(GY: torch.Tensor, Mc: int, X: torch.Tensor, Nc: int, ldp: int, T: int,
                lens: Optional[torch.Tensor], taps: int = 1, dil: int = 1,
                x_mask_mode: int = 0) -> torch.Tensor:
    R = GY.shape[0]
    tiles = -(-Mc // 128) * -(-Nc // 128) * taps
    S = pick_splits(tiles, R)
    P = torch.empty(S, taps, Mc, ldp, device=GY.device, dtype=torch.float32)
    if ldp != Nc:
        P.zero_()
    wgrad(GY=GY, ldgy=GY.shape[1], X=X, ldx=X.shape[1], P=P, ldp=ldp, split_stride=P.stride(0),
          R=R, Mc=Mc, Nc=Nc, taps=taps, dil=dil, T=T, lens=lens, x_mask_mode=x_mask_mode, splits=S)
    return P


# ---------------------------------------------------------------------------------------
# affine flow step: invertible 1x1 channel mix + WN + affine coupling, one autograd node
# ---------------------------------------------------------------------------------------
class AffineFlowStepFn(torch.autograd.Function):
    """FlowStep.forward of the reference (decoders.py:72-80) for the affine/WN coupling.

    inputs : z_in [N, ZLD], ctx [N, D], lens (int32 [B]) , W_eff [ZLD, ZLD], b_eff [ZLD],
             start_v/g/b, end_w/b, then L x (in_v, in_g, in_b), L x (res_v, res_g, res_b)
    outputs: z_out [N, ZLD], log_s [N, h]
    """

    @staticmethod
    @amp_fwd
    def forward(ctx, meta, z_in, cond, lens, W_eff, b_eff, start_v, start_g, start_b, end_w, end_b,
                *layer_params):
        B, T, C, D, nl = meta["B"], meta["T"], meta["C"], meta["D"], meta["n_layers"]
        act, scaling, partial = meta["act"], meta["scaling"], meta["partial"]
        h = C // 2
        N = B * T
        Wc = start_v.shape[0]                       # WN width (1024)
        Kp = round_up(D + h, 32)
        in_p = layer_params[: 3 * nl]
        res_p = layer_params[3 * nl:]
        assert z_in.shape == (N, ZLD) and cond.shape == (N, D) and z_in.is_contiguous() and cond.is_contiguous()

        # 1. invertible 1x1 (common.py:546 / :613-615): z1 = z_in @ W_eff^T + b_eff
        z1 = _empty(N, ZLD, like=z_in)
        rowgemm(A=z_in, lda=ZLD, B=W_eff, ldb=ZLD, b_layout=0, C=z1, ldc=ZLD, M=N, N=ZLD, K=ZLD, T=T,
                bias=b_eff)
        # 2. WN input cat((z0, context)) (common.py:819), K-padded
        X0 = _empty(N, Kp, like=z_in)
        check(lib.radmmm_wn_input_fwd(ptr(cond), D, ptr(z1), ZLD, ptr(X0), Kp, N, D, h, None, None, None, stream()), "wn_input_fwd")
        # 3. weight-norm fold (common.py:791,813,174)
        perm = (h, D, 0)                             # ref cols [z0 | ctx] -> packed [ctx | z0 | 0]
        Ws, inv_s = weightnorm_fwd(start_v, start_g, Kp, perm)
        Wi, inv_i, Wr, inv_r = [], [], [], []
        for j in range(nl):
            w, iv = weightnorm_fwd(in_p[3 * j], in_p[3 * j + 1])
            Wi.append(w); inv_i.append(iv)
            w, iv = weightnorm_fwd(res_p[3 * j], res_p[3 * j + 1])
            Wr.append(w); inv_r.append(iv)
        # 4. start conv
        H = [_empty(N, Wc, like=z_in)]
        rowgemm(A=X0, lda=Kp, B=Ws, ldb=Kp, b_layout=0, C=H[0], ldc=Wc, M=N, N=Wc, K=Kp, T=T, bias=start_b)
        # 5. dilated partial convs + res/skip 1x1 (common.py:829-832)
        OUT = _empty(N, Wc, like=z_in)
        R = []
        for j in range(nl):
            d = 2 ** j
            Hn = _empty(N, Wc, like=z_in)
            kt = in_p[3 * j].shape[2]
            rowgemm(A=H[j], lda=Wc, B=Wi[j], ldb=Wc, b_tap_stride=Wi[j].stride(0), b_layout=0, C=Hn, ldc=Wc,
                    M=N, N=Wc, K=Wc, taps=kt, dil=d, sign=1, T=T, lens=lens,
                    a_mask_mode=1 if partial else 0, bias=in_p[3 * j + 2], pconv=1 if partial else 0,
                    ratio_taps=kt, ratio_dil=d, postmask=1, act=act)
            H.append(Hn)
            Rj = _empty(N, Wc, like=z_in)
            rowgemm(A=Hn, lda=Wc, B=Wr[j], ldb=Wc, b_layout=0, C=Rj, ldc=Wc, M=N, N=Wc, K=Wc, T=T,
                    bias=res_p[3 * j + 2], act=act, C2=OUT, ldc2=Wc, c2_accum=1 if j > 0 else 0)
            R.append(Rj)
        # 6. end conv (plain, zero-init in the reference: common.py:799-802)
        O = _empty(N, ZLD, like=z_in)
        end_w2 = end_w.view(C, Wc)
        rowgemm(A=OUT, lda=Wc, B=end_w2, ldb=Wc, b_layout=0, C=O, ldc=ZLD, M=N, N=C, K=Wc, T=T, bias=end_b)
        # 7. affine coupling (common.py:1174-1185)
        z_out = _empty(N, ZLD, like=z_in)
        log_s = _empty(N, h, like=z_in)
        check(lib.radmmm_affine_coupling_fwd(ptr(O), ZLD, ptr(z1), ZLD, ptr(z_out), ptr(log_s), N, h, scaling,
                                             stream()), "affine_coupling_fwd")
        ctx.meta = meta
        ctx.nl = nl
        ctx.save_for_backward(z_in, z1, X0, OUT, O, lens, W_eff, start_v, start_g, end_w, Ws, inv_s,
                              *H, *R, *Wi, *inv_i, *Wr, *inv_r, *layer_params)
        return z_out, log_s

    @staticmethod
    @amp_bwd
    def backward(ctx, g_zout, g_logs):
        meta, nl = ctx.meta, ctx.nl
        B, T, C, D = meta["B"], meta["T"], meta["C"], meta["D"]
        act, scaling, partial = meta["act"], meta["scaling"], meta["partial"]
        sv = ctx.saved_tensors
        z_in, z1, X0, OUT, O, lens, W_eff, start_v, start_g, end_w, Ws, inv_s = sv[:12]
        p = 12
        H = sv[p: p + nl + 1]; p += nl + 1
        R = sv[p: p + nl]; p += nl
        Wi = sv[p: p + nl]; p += nl
        inv_i = sv[p: p + nl]; p += nl
        Wr = sv[p: p + nl]; p += nl
        inv_r = sv[p: p + nl]; p += nl
        layer_params = sv[p:]
        in_p, res_p = layer_params[: 3 * nl], layer_params[3 * nl:]
        h = C // 2
        N = B * T
        Wc = start_v.shape[0]
        Kp = X0.shape[1]
        g_zout = g_zout.contiguous()
        if g_logs is not None:
            g_logs = g_logs.contiguous()

        # coupling
        gO = _empty(N, ZLD, like=z_in)
        gz1 = _empty(N, ZLD, like=z_in)
        check(lib.radmmm_affine_coupling_bwd(ptr(O), ZLD, ptr(z1), ZLD, ptr(g_zout), ptr(g_logs), ptr(gO), ptr(gz1),
                                             N, h, scaling, stream()), "affine_coupling_bwd")
        # end conv
        g_end_b = colsum(gO, C)
        g_end_w = wgrad_slabs(gO, C, OUT, Wc, Wc, T, None).sum(0).view(C, Wc, 1)
        gOUT = _empty(N, Wc, like=z_in)
        rowgemm(A=gO, lda=ZLD, B=end_w.view(C, Wc), ldb=Wc, b_layout=1, C=gOUT, ldc=Wc, M=N, N=Wc, K=C, T=T)
        g_in: List[Optional[torch.Tensor]] = [None] * (3 * nl)
        g_res: List[Optional[torch.Tensor]] = [None] * (3 * nl)
        G = None
        gQ = _empty(N, Wc, like=z_in)
        for j in range(nl - 1, -1, -1):
            d = 2 ** j
            kt = in_p[3 * j].shape[2]
            # through softplus of the res/skip branch
            check(lib.radmmm_dact_mul(ptr(gOUT), Wc, ptr(R[j]), Wc, ptr(gQ), Wc, N, Wc, act, 0, T, None, 1, 1, None, None, 0, 1.0,
                                      None, stream()), "dact_mul")
            g_res[3 * j + 2] = colsum(gQ, Wc)
            slabs = wgrad_slabs(gQ, Wc, H[j + 1], Wc, Wc, T, None)
            g_res[3 * j], g_res[3 * j + 1] = weightnorm_bwd(res_p[3 * j], res_p[3 * j + 1], inv_r[j], slabs, Wc)
            # dL/dconv_j = (gQ @ Wres + dL/dH_{j+1} via in_layer j+1) * softplus'(H_{j+1}) * mask * ratio_j
            g_conv = _empty(N, Wc, like=z_in)
            rowgemm(A=gQ, lda=Wc, B=Wr[j], ldb=Wc, b_layout=1, C=g_conv, ldc=Wc, M=N, N=Wc, K=Wc, T=T,
                    lens=lens, add=G, ldadd=Wc, dact_src=H[j + 1], lddact=Wc, dact=act,
                    rowscale=2 if partial else 1, ratio_taps=kt, ratio_dil=d)
            g_in[3 * j + 2] = colsum(g_conv, Wc, 2 if partial else 0, T, lens, kt, d)
            slabs = wgrad_slabs(g_conv, Wc, H[j], Wc, Wc, T, lens, taps=kt, dil=d,
                                x_mask_mode=1 if partial else 0)
            g_in[3 * j], g_in[3 * j + 1] = weightnorm_bwd(in_p[3 * j], in_p[3 * j + 1], inv_i[j], slabs, Wc)
            G = _empty(N, Wc, like=z_in)
            rowgemm(A=g_conv, lda=Wc, B=Wi[j], ldb=Wc, b_tap_stride=Wi[j].stride(0), b_layout=1, C=G, ldc=Wc,
                    M=N, N=Wc, K=Wc, taps=kt, dil=d, sign=-1, T=T, lens=lens, a_mask_mode=0,
                    premask=1 if partial else 0)
        # start conv: G is dL/dH_0
        g_start_b = colsum(G, Wc)
        perm = (h, D, 0)
        slabs = wgrad_slabs(G, Wc, X0, Kp, Kp, T, None)
        g_start_v, g_start_g = weightnorm_bwd(start_v, start_g, inv_s, slabs, Kp, perm)
        gX0 = _empty(N, Kp, like=z_in)
        rowgemm(A=G, lda=Wc, B=Ws, ldb=Kp, b_layout=1, C=gX0, ldc=Kp, M=N, N=Kp, K=Wc, T=T)
        g_cond = _empty(N, D, like=z_in)
        check(lib.radmmm_wn_input_bwd(ptr(gX0), Kp, ptr(g_cond), D, 0, ptr(gz1), ZLD, N, D, h, stream()), "wn_input_bwd")
        # invertible 1x1
        g_b_eff = colsum(gz1, ZLD) if ctx.needs_input_grad[5] else None   # LUS conv: constant zero bias
        g_W_eff = wgrad_slabs(gz1, ZLD, z_in, ZLD, ZLD, T, None).sum(0).view(ZLD, ZLD)
        g_zin = _empty(N, ZLD, like=z_in)
        rowgemm(A=gz1, lda=ZLD, B=W_eff, ldb=ZLD, b_layout=1, C=g_zin, ldc=ZLD, M=N, N=ZLD, K=ZLD, T=T)
        return (None, g_zin, g_cond, None, g_W_eff, g_b_eff, g_start_v, g_start_g, g_start_b, g_end_w, g_end_b,
                *g_in, *g_res)


# ---------------------------------------------------------------------------------------
# squeeze (nn.Unfold group g) folded into the assembly of the channels-last operands
# ---------------------------------------------------------------------------------------
def _squeeze_rows_into(x, out2d, B, C, T, g, ld, col0):
    """nn.Unfold's squeeze of x [B, C, T] into columns [col0, col0 + C*g) of the channels-last rows out2d [B*(T//g), ld].
    The kernel's tile covers group sizes dividing 64 (every shipped config: 2); other reference-legal group sizes
    (n_group_size 3, 5, 6 ...) take the same index permutation through stock device ops."""
    if 64 % g == 0:
        check(lib.radmmm_squeeze_rows(ptr(x), ptr(out2d), B, C, T, g, ld, col0, stream()), "squeeze_rows")
        return
    Tg = T // g
    out2d.view(B, Tg, ld)[:, :, col0: col0 + C * g] = x[:, :, : Tg * g].reshape(B, C, Tg, g).permute(0, 2, 1, 3).reshape(B, Tg, C * g)


def _unsqueeze_rows_from(gout2d, gin, B, C, T, g, ld, col0):
    """gradient scatter of _squeeze_rows_into: gin [B, C, T] (frames beyond g * (T // g) get zero)"""
    if 64 % g == 0:
        check(lib.radmmm_unsqueeze_rows(ptr(gout2d), ptr(gin), B, C, T, g, ld, col0, stream()), "unsqueeze_rows")
        return
    Tg = T // g
    gin.zero_()
    gin[:, :, : Tg * g] = gout2d.view(B, Tg, ld)[:, :, col0: col0 + C * g].reshape(B, Tg, C, g).permute(0, 2, 1, 3).reshape(B, C, Tg * g)


class SqueezeRowsFn(torch.autograd.Function):
    """x [B, C, T] (reference layout) -> channels-last rows [B*T', ld], T' = T // g, the C*g squeezed channels (order
    c*g + k, nn.Unfold's) at columns [col0, col0 + C*g), zeros elsewhere (decoders.py:118-122,178)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, g, ld, col0):
        B, C, T = x.shape
        x = f32c(x)
        Tg = T // g
        out = (torch.zeros if ld != C * g else torch.empty)(B * Tg, ld, device=x.device, dtype=torch.float32)
        _squeeze_rows_into(x, out, B, C, T, g, ld, col0)
        ctx.dims = (B, C, T, g, ld, col0)
        return out

    @staticmethod
    @amp_bwd
    def backward(ctx, gout):
        B, C, T, g, ld, col0 = ctx.dims
        gin = torch.empty(B, C, T, device=gout.device, dtype=torch.float32)
        _unsqueeze_rows_from(f32c(gout), gin, B, C, T, g, ld, col0)
        return gin, None, None, None


def squeeze_rows(x: torch.Tensor, g: int, ld: int, col0: int = 0) -> torch.Tensor:
    return SqueezeRowsFn.apply(x, g, ld, col0)


class LstmInputFn(torch.autograd.Function):
    """Input of the context LSTM (models/radmmm.py:114-134): cat(unfold(context), spk, [accent], unfold(f0),
    unfold(energy)) as channels-last rows [B, T', I] written in place -- the squeeze kernel puts the context into its
    columns, the per-utterance vectors are broadcast and the one-channel tracks are plain views (channel c*g + k with
    C = 1 is sample t'*g + k).  No permuted copy, no concatenation."""

    @staticmethod
    @amp_fwd
    def forward(ctx, g, context, spk, accent, f0, energy):
        B, Ct, T = context.shape
        Tg = T // g
        widths = [Ct * g, spk.shape[1], accent.shape[1] if accent is not None else 0, g if f0 is not None else 0,
                  g if energy is not None else 0]
        I = sum(widths)
        x = torch.empty(B, Tg, I, device=context.device, dtype=torch.float32)
        _squeeze_rows_into(f32c(context), x.view(B * Tg, I), B, Ct, T, g, I, 0)
        o = widths[0]
        x[:, :, o: o + widths[1]] = spk[:, None, :]
        o += widths[1]
        if accent is not None:
            x[:, :, o: o + widths[2]] = accent[:, None, :]
            o += widths[2]
        if f0 is not None:
            x[:, :, o: o + g] = f0[:, : Tg * g].reshape(B, Tg, g)
            o += g
        if energy is not None:
            x[:, :, o: o + g] = energy[:, : Tg * g].reshape(B, Tg, g)
        ctx.dims = (B, Ct, T, g, I, widths)
        ctx.flags = (accent is not None, f0 is not None, energy is not None)
        return x

    @staticmethod
    @amp_bwd
    def backward(ctx, gx):
        B, Ct, T, g, I, widths = ctx.dims
        has_acc, has_f0, has_en = ctx.flags
        Tg = T // g
        gx = f32c(gx)
        need = ctx.needs_input_grad
        gctx = None
        if need[1]:
            gctx = torch.empty(B, Ct, T, device=gx.device, dtype=torch.float32)
            _unsqueeze_rows_from(gx.view(B * Tg, I), gctx, B, Ct, T, g, I, 0)
        o = widths[0]
        gspk = gx[:, :, o: o + widths[1]].sum(1) if need[2] else None
        o += widths[1]
        gacc = None
        if has_acc:
            gacc = gx[:, :, o: o + widths[2]].sum(1) if need[3] else None
            o += widths[2]

        def track(off):
            gt = torch.zeros(B, T, device=gx.device, dtype=torch.float32)
            gt[:, : Tg * g] = gx[:, :, off: off + g].reshape(B, Tg * g)
            return gt
        gf0 = gen = None
        if has_f0:
            gf0 = track(o) if need[4] else None
            o += g
        if has_en:
            gen = track(o) if need[5] else None
        return None, gctx, gspk, gacc, gf0, gen


# ---------------------------------------------------------------------------------------
# masked reductions for the flow NLL (loss.py:85-110)
# ---------------------------------------------------------------------------------------
class MaskedReduceFn(torch.autograd.Function):
    """sum over [B,C,T] of x*m (mode 0) or (x*m)^2 (mode 1), m = [t < lens[b]]; x may be any
    dense (permuted) view."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, lens, mode):
        assert x.dim() == 3 and x.dtype == torch.float32
        B, C, T = x.shape
        out = _empty(1, like=x)
        scratch = _empty(int(lib.radmmm_masked_reduce_scratch_floats(B, C, T)), like=x)
        sb, sc, st = x.stride()
        check(lib.radmmm_masked_reduce(ptr(x), B, C, T, sb, sc, st, ptr(lens), mode, ptr(out), ptr(scratch),
                                       stream()), "masked_reduce")
        ctx.save_for_backward(x, lens)
        ctx.mode = mode
        return out.view(())

    @staticmethod
    @amp_bwd
    def backward(ctx, g):
        x, lens = ctx.saved_tensors
        B, C, T = x.shape
        gx = torch.empty_like(x)          # preserves x's (dense) strides
        assert gx.stride() == x.stride()
        coef = g.reshape(1).contiguous().float()
        sb, sc, st = x.stride()
        check(lib.radmmm_masked_reduce_bwd(ptr(x), B, C, T, sb, sc, st, ptr(lens), ctx.mode, ptr(coef), ptr(gx),
                                           stream()), "masked_reduce_bwd")
        return gx, None, None


def masked_sum(x, lens):
    return MaskedReduceFn.apply(x, lens, 0)


def masked_sumsq(x, lens):
    return MaskedReduceFn.apply(x, lens, 1)


def fused_add_tanh_sigmoid_multiply(a: torch.Tensor, b: torch.Tensor, n_channels: int) -> torch.Tensor:
    """common.py:66-73 on channels-last [rows, 2n] operands (forward only; no config uses it)."""
    rows, ld = a.shape
    y = _empty(rows, n_channels, like=a)
    check(lib.radmmm_fused_add_tanh_sigmoid_multiply(ptr(a), ptr(b), ld, ptr(y), n_channels, rows, n_channels,
                                                     stream()), "fused_add_tanh_sigmoid_multiply")
    return y


# ---------------------------------------------------------------------------------------
# generic weight-normed ConvNorm (common.py:152-191) on channels-last rows
# ---------------------------------------------------------------------------------------
class ConvNormFn(torch.autograd.Function):
    """y = act( mask_out * ( pconv_ratio * conv(x * mask_in, weight_norm(v, g)) + bias ) ).

    x [N, ldx] (first Cin columns), v [Cout, Cin, taps], g [Cout,1,1] or None (plain conv: v IS the
    weight), bias [Cout].  meta: B, T, dil, partial (PartialConv1d: input masking + window
    re-normalisation), mask_out (ConvNorm multiplies by the mask again), act code.
    Returns y [N, round_up(Cout, 4)].
    """

    @staticmethod
    @amp_fwd
    def forward(ctx, meta, x, v, g, bias, lens):
        B, T, dil = meta["B"], meta["T"], meta["dil"]
        partial, mask_out, act = meta["partial"], meta["mask_out"], meta["act"]
        Cout, Cin, taps = v.shape
        N = B * T
        assert x.shape[0] == N and x.shape[1] >= Cin and x.shape[1] % 4 == 0 and x.is_contiguous()
        if g is not None:
            W, inv = weightnorm_fwd(v, g)
        else:
            ldw = round_up(Cin, 4)
            W = torch.zeros(taps, Cout, ldw, device=x.device, dtype=torch.float32)
            W[:, :, :Cin] = v.permute(2, 0, 1)
            inv = None
        ldy = round_up(Cout, 4)
        y = torch.zeros(N, ldy, device=x.device, dtype=torch.float32) if ldy != Cout else _empty(N, ldy, like=x)
        rowgemm(A=x, lda=x.shape[1], B=W, ldb=W.shape[2], b_tap_stride=W.stride(0), b_layout=0, C=y, ldc=ldy,
                M=N, N=Cout, K=Cin, taps=taps, dil=dil, sign=1, T=T, lens=lens, a_mask_mode=1 if partial else 0,
                bias=bias, pconv=1 if partial else 0, ratio_taps=taps, ratio_dil=dil,
                postmask=1 if mask_out else 0, act=act)
        ctx.meta = meta
        ctx.has_g = g is not None
        ctx.save_for_backward(x, v, g if g is not None else v, bias if bias is not None else v, lens if lens is not None else v,
                              W, inv if inv is not None else v, y)
        ctx.has_bias = bias is not None
        ctx.has_lens = lens is not None
        return y

    @staticmethod
    @amp_bwd
    def backward(ctx, gy):
        meta = ctx.meta
        B, T, dil = meta["B"], meta["T"], meta["dil"]
        partial, mask_out, act = meta["partial"], meta["mask_out"], meta["act"]
        x, v, g, bias, lens, W, inv, y = ctx.saved_tensors
        lens = lens if ctx.has_lens else None
        Cout, Cin, taps = v.shape
        N = B * T
        gy = gy.contiguous()
        ldy = y.shape[1]
        rowscale = 2 if partial else (1 if mask_out else 0)
        gpre = torch.zeros_like(y) if ldy != Cout else torch.empty_like(y)
        check(lib.radmmm_dact_mul(ptr(gy), ldy, ptr(y), ldy, ptr(gpre), ldy, N, Cout, act, rowscale, T, ptr(lens),
                                  taps, dil, None, None, 0, 1.0, None, stream()), "dact_mul")
        g_bias = colsum(gpre, Cout, 2 if partial else 0, T, lens, taps, dil) if ctx.has_bias else None
        ldw = W.shape[2]
        slabs = wgrad_slabs(gpre, Cout, x, Cin, ldw, T, lens, taps=taps, dil=dil, x_mask_mode=1 if partial else 0)
        if ctx.has_g:
            g_v, g_g = weightnorm_bwd(v, g, inv, slabs, ldw)
        else:
            g_v, g_g = slabs.sum(0)[:, :, :Cin].permute(1, 2, 0).contiguous(), None
        gx = torch.zeros_like(x) if x.shape[1] != Cin else torch.empty_like(x)
        if ctx.needs_input_grad[1]:
            rowgemm(A=gpre, lda=ldy, B=W, ldb=ldw, b_tap_stride=W.stride(0), b_layout=1, C=gx, ldc=x.shape[1], M=N,
                    N=Cin, K=Cout, taps=taps, dil=dil, sign=-1, T=T, lens=lens, a_mask_mode=0,
                    premask=1 if partial else 0)
        else:
            gx = None
        return None, gx, g_v, g_g, g_bias, None


def conv_norm(x, v, g, bias, lens, B, T, dil=1, partial=False, mask_out=False, act="none", scale_box=None, nprod=None,
              wgrad8=False):
    """scale_box: dict shared by the convs of one backward pass (gradient scale of the split-f16 path,
    fixed by the first node that runs; without it every conv's backward syncs once for its own)."""
    meta = dict(B=B, T=T, dil=dil, partial=bool(partial), mask_out=bool(mask_out), act=ACT[act],
                scale_box=scale_box if scale_box is not None else {})
    Cout, Cin, taps = v.shape
    # split-f16 path (three f16 products: 2e-6 of fp32) for the frame-rate convs (FiLM stacks: N = B*T' rows) and, since
    # round 4, the text-rate ones of a batch (text encoder, key projection: B * T_txt = 4800 rows at the benchmark batch --
    # on the fp32-MFMA kernels they were 2.2 ms of the full step, 0.9 ms more than here); single utterances stay on fp32 MFMA
    min_rows = int(debug_env("RADMMM_CONVNORM_H3_MIN_ROWS", "1024"))
    prec = os.environ.get("RADMMM_PRECISION", "f8x")
    # "f8x" (FP8 cross terms) is for the WN stack of the affine flows; the FiLM convs of the spline flows keep the three
    # f16 products: the piecewise-quadratic transform's log-Jacobian amplifies errors of its 65 parameters per element
    # (measured: flow-0 log_s off by > 1e-4 with f8x FiLM convs, tests/test_hip_parity.py cfg5_small)
    meta["nprod"] = 3 if prec == "f8x" else NPROD.get(prec, 3)
    if nprod == 2 and prec == "f8x" and x.shape[0] >= int(debug_env("RADMMM_F8X_MIN_ROWS", "4096")) and v.shape[0] % 32 == 0:
        meta["nprod"] = 2          # a caller that has established the FP8-cross scheme's accuracy for this conv (FiLM blocks)
    # Round 6: a conv that keeps three products for its outputs and data gradients (the FiLM blocks: the spline's log-Jacobian
    # amplifies errors of its parameters in backward as well) can still take the FP8-cross kernel for its WEIGHT
    # gradient -- a leaf: its rounding (~1e-4 of the tensor, the WN convs' own weight-gradient accuracy) goes nowhere else --
    # on the slot-pinned radmmm_wgrad_rm8 instead of the three-product radmmm_wgrad_rm (configs[4]: 6.8 ms per step)
    meta["wgrad8"] = bool(wgrad8 and prec == "f8x" and meta["nprod"] == 3 and x.shape[0] >= int(debug_env("RADMMM_F8X_MIN_ROWS", "4096"))
                          and debug_env("RADMMM_FILM_WGRAD8", "1") != "0")
    h3_ok = (prec in NPROD and (taps // 2) * dil <= 16 and x.shape[0] >= min_rows and
             x.shape[0] * max(round_up(Cin, 32), Cout) < 2 ** 30)
    if h3_ok and Cin % 32 != 0 and x.is_cuda and debug_env("RADMMM_CONVNORM_PAD", "1") != "0":
        # the split-f16 kernels want K % 32 == 0: an odd input width (the RADMMM configs' n_text_dim = 520 in the text encoder,
        # the key projection and the predictors' bottlenecks) is zero-padded -- x by one small copy, the weight by autograd's
        # pad, whose zeros leave the weight norm alone -- instead of running the conv on the fp32-MFMA kernels (joint step:
        # 275 us per launch there).  Gradients come back through the slice / pad nodes.
        padc = (-Cin) % 32
        x = torch.nn.functional.pad(x[:, :Cin], (0, padc))
        v = torch.nn.functional.pad(v, (0, 0, 0, padc))
        Cin += padc
    if h3_ok and Cin % 32 == 0:
        return ConvNormH3Fn.apply(meta, x, v, g, bias, lens)
    return ConvNormFn.apply(meta, x, v, g, bias, lens)


# ---------------------------------------------------------------------------------------
# alignment attention core (common.py:1262-1277)
# ---------------------------------------------------------------------------------------
class AttentionCoreFn(torch.autograd.Function):
    """Q [B,T1,Ca], K [B,T2,Ca] (channels-last, contiguous), prior [B,T1,T2] or None, in_lens int32 [B]
    or None -> attn, attn_logprob [B,T1,T2]."""

    @staticmethod
    @amp_fwd
    def forward(ctx, Q, K, prior, in_lens, temp):
        B, T1, Ca = Q.shape
        T2 = K.shape[1]
        attn = _empty(B, T1, T2, like=Q)
        logprob = _empty(B, T1, T2, like=Q)
        check(lib.radmmm_attn_fwd(ptr(Q), ptr(K), ptr(prior), ptr(in_lens), ptr(attn), ptr(logprob), B, T1, T2, Ca,
                                  temp, stream()), "attn_fwd")
        ctx.save_for_backward(Q, K, prior if prior is not None else Q, in_lens if in_lens is not None else Q, attn, logprob)
        ctx.flags = (prior is not None, in_lens is not None, temp)
        return attn, logprob

    @staticmethod
    @amp_bwd
    def backward(ctx, gattn, glogprob):
        Q, K, prior, in_lens, attn, logprob = ctx.saved_tensors
        has_prior, has_lens, temp = ctx.flags
        B, T1, Ca = Q.shape
        T2 = K.shape[1]
        gQ, gK = torch.empty_like(Q), torch.empty_like(K)
        scratch = _empty(B, T1, T2, like=Q)
        ga = gattn.contiguous() if gattn is not None else None
        gl = glogprob.contiguous() if glogprob is not None else None
        check(lib.radmmm_attn_bwd(ptr(Q), ptr(K), ptr(prior) if has_prior else None, ptr(in_lens) if has_lens else None,
                                  ptr(attn), ptr(logprob), ptr(ga), ptr(gl), ptr(gQ), ptr(gK), ptr(scratch), B, T1, T2,
                                  Ca, temp, stream()), "attn_bwd")
        return gQ, gK, None, None, None


def mas_width1_batch(logp: torch.Tensor, in_lens: torch.Tensor, out_lens: torch.Tensor, prob: bool = False) -> torch.Tensor:
    """logp [B,T1,T2] = log(attn) on the GPU (prob=True: the attention itself, the kernel takes the correctly rounded
    fp32 log) -> 0/1 hard alignment [B,T1,T2] (alignment.py:31-59)."""
    B, T1, T2 = logp.shape
    hard = _empty(B, T1, T2, like=logp)
    scratch = torch.empty(int(lib.radmmm_mas_scratch_bytes(B, T1, T2)), device=logp.device, dtype=torch.uint8)
    fn = lib.radmmm_mas_width1_prob if prob else lib.radmmm_mas_width1
    check(fn(ptr(f32c(logp)), ptr(in_lens), ptr(out_lens), ptr(hard), ptr(scratch), B, T1, T2, stream()), "mas_width1")
    return hard


class CTCMonotonicFn(torch.autograd.Function):
    """nll [B] of torch's CTC loss (blank 0, zero_infinity) for the targets 1 .. L_b, from log-probabilities lp [B, T, C]
    (class 0 = blank); value and gradient come out of one call (radmmm_ctc_monotonic: the alpha and beta chains side by
    side, then an elementwise gradient launch; no host synchronisation), the backward scales the stored gradient."""

    @staticmethod
    @amp_fwd
    def forward(ctx, lp, lens_txt, lens_mel):
        B, T, C = lp.shape
        lp = f32c(lp)
        nll = _empty(B, like=lp)
        grad = torch.empty_like(lp)
        scratch = _empty(int(lib.radmmm_ctc_monotonic_scratch_floats(B, T, C)), like=lp)
        check(lib.radmmm_ctc_monotonic(ptr(lp), ptr(lens_txt), ptr(lens_mel), ptr(nll), ptr(grad), ptr(scratch), B, T, C,
                                       stream()), "ctc_monotonic")
        ctx.save_for_backward(grad)
        return nll

    @staticmethod
    @amp_bwd
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.reshape(-1, 1, 1), None, None


def stft_mel(audio: torch.Tensor, basis: torch.Tensor, mel_basis: torch.Tensor, n_fft: int, hop: int,
             clip: float = 1e-5) -> torch.Tensor:
    """audio [B,S] -> log-mel [B, n_mel, 1+S//hop] (audio_processing.py:137-154)."""
    B, S = audio.shape
    n_mel = mel_basis.shape[0]
    F_ = 1 + S // hop
    mel = _empty(B, n_mel, F_, like=audio)
    scratch = _empty(int(lib.radmmm_stft_mel_scratch_floats(B, S, n_fft, hop, n_mel)), like=audio)
    check(lib.radmmm_stft_mel(ptr(audio.contiguous()), ptr(basis), ptr(mel_basis), ptr(mel), ptr(scratch), B, S, n_fft,
                              hop, n_mel, clip, stream()), "stft_mel")
    return mel


# ---------------------------------------------------------------------------------------
# affine flow step on the split-f16 GEMM path (fp32-class accuracy on the f16 matrix cores)
# ---------------------------------------------------------------------------------------
from ._lib import rowgemm_h3, debug_env  # noqa: E402

W_SCALE = 256.0          # power-of-two scale of the split weights (|w| <= |g| ~ 1 after weight norm)
# RADMMM_PRECISION / gemm_precision -> product scheme of radmmm_rowgemm_h3 ("fp32" uses the fp32-MFMA kernels instead)
NPROD = {"h3": 3, "f8x": 2, "f16": 1}


def _halves(*shape, like, zero=False):
    f = torch.zeros if zero else torch.empty
    return f(*shape, device=like.device, dtype=torch.float16), f(*shape, device=like.device, dtype=torch.float16)


# 8-bit parts of the "FP8 cross terms" scheme (nprod = 2) are written as value * 2^e.  e4m3 saturates at
# 448 and flushes below 2^-9; a saturated element loses its cross-term correction (single-product accuracy, ~5e-4), an
# element far below the range merely has cross terms that are 2^-11 of an already negligible product.
#   activations (softplus outputs, typically 0.01 .. 10): x4 -- never seen to saturate (> 112);
#   weights (x256 already, |w| <= ~1): x1;
#   gradients, scaled by S so that the FIRST backward node's amax is 8 .. 16: the gradients INSIDE the WN are up to ~200x
#     that amax (measured on the benchmark batch, tools/x8_exp_probe.py, profiles/r04_x8_exp_probe.txt: saturation reports
#     stop at e = -3; accuracy is flat from e = 0 down to -6: worst elementwise gradient error 6.2e-5, against 2.2e-4 at
#     round 3's e = 2, which saturated on every step).  X8_GRAD_EXP is the STARTING value: GradScale lowers a decoder's
#     exponent by the level the device flag reports whenever a pass saturates (it never needs to come back up).
X8_ACT_EXP, X8_GRAD_EXP, X8_W_EXP = 2, -4, 0
X8_GRAD_EXP_MIN = -16


def fmt_a(nprod: int) -> int:
    return SPLIT_X8A if nprod == 2 else SPLIT_F16


def fmt_b(nprod: int) -> int:
    return SPLIT_X8B if nprod == 2 else SPLIT_F16


def split_weight(v, g, ldk, perm=(0, 0, 0), nprod=3):
    """v [Cout, Cin, taps] (+ weight-norm g or None) -> split packed W{h,l} [taps, Cout, ldk], inv_norm (nprod 2: Wl is
    the B-role 8-bit cross array, same shape/dtype container)."""
    Cout, Cin, taps = v.shape
    Wh, Wl = _halves(taps, Cout, ldk, like=v, zero=(ldk != Cin or perm != (0, 0, 0)))
    inv = _empty(Cout, like=v) if g is not None else None
    check(lib.radmmm_weightnorm_fwd_h3(ptr(v), ptr(g), ptr(Wh), ptr(Wl), ptr(inv), Cout, Cin, taps, ldk, perm[0], perm[1],
                                       perm[2], W_SCALE, split_opts(fmt_b(nprod), X8_W_EXP), stream()), "weightnorm_fwd_h3")
    return Wh, Wl, inv


def split_weights(specs, nprod=3):
    """split_weight for several tensors -- specs: [(v, g or None, ldk, perm or None), ...] -> [(Wh, Wl, inv_norm), ...].
    Up to 16 per launch go through radmmm_weightnorm_fwd_h3_multi (a flow step's ten conv weights in ONE launch: most are
    4 - 9 MB, where a launch of their own is latency- rather than bandwidth-bound); same arithmetic, same results."""
    from ._lib import WnItem
    out, items = [], []
    for v, g, ldk, perm in specs:
        perm = perm or (0, 0, 0)
        Cout, Cin, taps = v.shape
        Wh, Wl = _halves(taps, Cout, ldk, like=v, zero=(ldk != Cin or perm != (0, 0, 0)))
        inv = _empty(Cout, like=v) if g is not None else None
        out.append((Wh, Wl, inv))
        items.append(WnItem(ptr(v), ptr(g), ptr(Wh), ptr(Wl), ptr(inv), Cout, Cin, taps, ldk, perm[0], perm[1], perm[2]))
    for k in range(0, len(items), 16):
        chunk = items[k: k + 16]
        arr = (WnItem * len(chunk))(*chunk)
        check(lib.radmmm_weightnorm_fwd_h3_multi(arr, len(chunk), W_SCALE, split_opts(fmt_b(nprod), X8_W_EXP), stream()),
              "weightnorm_fwd_h3_multi")
    return out


def transpose_splits(specs, nprod=2):
    """transpose_split for several B-role 8-bit pairs (FP8-cross scheme) in one launch -- specs: [(Wh, Wl, rows, cols, ld_dst,
    out or None), ...] -> [(Th, Tl), ...] with Th [taps][cols][ld_dst]; out = (Th, Tl) views to write into (slices of a
    larger tap stack)."""
    from ._lib import TpItem
    assert nprod == 2
    res, items = [], []
    for Wh, Wl, rows, cols, ld_dst, out in specs:
        taps = Wh.shape[0]
        if out is not None:
            Th, Tl = out
            assert Th.shape == (taps, cols, ld_dst) and Th.is_contiguous() and ld_dst == rows
        else:
            Th, Tl = _halves(taps, cols, ld_dst, like=Wh, zero=(ld_dst != rows))
        res.append((Th, Tl))
        items.append(TpItem(ptr(Wh), ptr(Wl), ptr(Th), ptr(Tl), Wh.stride(0), Th.stride(0), Wh.shape[2], ld_dst, taps, rows, cols))
    for k in range(0, len(items), 16):
        chunk = items[k: k + 16]
        arr = (TpItem * len(chunk))(*chunk)
        check(lib.radmmm_transpose_f16_pair_multi(arr, len(chunk), fmt_b(nprod), X8_W_EXP, stream()), "transpose_f16_pair_multi")
    return res


def transpose_split(Wh, Wl, rows, cols, ld_dst, nprod=3, out=None):
    """[taps][rows][ld] pair -> [taps][cols][ld_dst] (zero padded), i.e. the K-contiguous operand of the
    data-gradient GEMM.  out = (Th, Tl): write into these [taps][cols][ld_dst] views (slices of a larger tap stack)."""
    taps = Wh.shape[0]
    if out is not None:
        Th, Tl = out
        assert Th.shape == (taps, cols, ld_dst) and Th.is_contiguous() and ld_dst == rows
    else:
        Th, Tl = _halves(taps, cols, ld_dst, like=Wh, zero=(ld_dst != rows))
    check(lib.radmmm_transpose_f16_pair(ptr(Wh), ptr(Wl), Wh.shape[2], Wh.stride(0), ptr(Th), ptr(Tl), ld_dst, Th.stride(0),
                                        taps, rows, cols, fmt_b(nprod), X8_W_EXP, stream()), "transpose_f16_pair")
    return Th, Tl


def split_f16(x, cols, scale, ldh=None, nprod=3, x8_exp=0, sat_flag=None, lo16=None):
    """lo16 (nprod 2 only): an fp16 [rows, ldh] tensor that receives the fp16 lo part beside the 8-bit cross array"""
    rows = x.shape[0]
    ldh = ldh or round_up(cols, 32 if nprod == 2 else 8)
    hi, lo = _halves(rows, ldh, like=x)
    check(lib.radmmm_split_f16(ptr(x), x.shape[1], ptr(hi), ptr(lo), ldh, rows, cols, scale,
                               split_opts(fmt_a(nprod), x8_exp, sat_flag, lo16), stream()), "split_f16")
    return hi, lo


_TS_FRONT = 16           # leading zero columns / zero gap between utterances (>= max tap shift 2*8)
_ts_pool = {}


def _ts_buffers(device, C, B, T, role, need_odd):
    """pooled [C, ldk] fp16 buffers of a transposed zero-gapped split copy -> (list of views, Tp, Kt, ldk)"""
    Tp = T + _TS_FRONT
    Kt = round_up(B * Tp, 32)                 # contracted columns [FRONT, FRONT + Kt)
    ldk = Kt + 2 * _TS_FRONT
    # one set of flat buffers per (role, C): batches of a real run differ in B and T, so the [C, ldk] views are cut
    # from storage sized for the largest shape seen, and whenever the shape changes the used region is cleared
    # (the kernel rewrites data and gaps only; front, tail and the round-up columns must read as zeros)
    # (one pool per stream: buffers are reused in launch order, which only a single stream guarantees)
    key = (device, torch.cuda.current_stream(device).cuda_stream, C, role, need_odd)
    ent = _ts_pool.get(key)
    need = C * ldk
    if ent is None or ent["cap"] < need:
        n = 4 if need_odd else 2
        ent = {"flat": [torch.zeros(need, device=device, dtype=torch.float16) for _ in range(n)], "cap": need,
               "shape": (B, Tp)}
        _ts_pool[key] = ent
    elif ent["shape"] != (B, Tp):
        for f in ent["flat"]:
            f[:need].zero_()
        ent["shape"] = (B, Tp)
    return [f[:need].view(C, ldk) for f in ent["flat"]], Tp, Kt, ldk


class ColsumBatch:
    """Deferred finals of partial column sums (bias gradients): the producers of a flow step's backward leave their per-tile
    partial rows in scratch of their own and ONE radmmm_colsum_final_multi launch adds them all up -- nine 5-microsecond
    launches per flow step otherwise (82 per training step).  flush() must run before anybody may read the sums."""

    def __init__(self):
        self.items, self.keep = [], []

    def add(self, part: torch.Tensor, out: torch.Tensor, nparts: int, cols: int) -> None:
        self.items.append(L.CsItem(ptr(part), ptr(out), nparts, cols))
        self.keep.append((part, out))

    def flush(self) -> None:
        if self.items:
            arr = (L.CsItem * len(self.items))(*self.items)
            check(lib.radmmm_colsum_final_multi(arr, len(self.items), stream()), "colsum_final_multi")
        self.items, self.keep = [], []


def dact_mul_transposed(g, saved, C, B, T, act, scale, role, yh, yl, fmt, x8_exp, sat_flag, sum_out=None, ylo16=None, defer=None):
    """y = g * act'(saved) written as the row-major split pair yh/yl (format fmt; ylo16: the fp16 lo part as well when
    yl is an 8-bit cross array), as the transposed zero-gapped split-f16 copy (pool `role`; role None: no transposed copy)
    and as column sums, in one pass and without an fp32 y (radmmm_dact_mul_transposed) -> (transposed-copy tuple, sums [C])."""
    if role is None:
        Tp = T + _TS_FRONT
        Kt, ldk, oh, ol = 0, 0, None, None
    else:
        bufs, Tp, Kt, ldk = _ts_buffers(g.device, C, B, T, role, False)
        oh, ol = bufs[0], bufs[1]
    nparts = B * (-(-Tp // 64))
    part = _empty(nparts, C, like=g)
    sums = sum_out if (sum_out is not None and sum_out.numel() == C) else _empty(C, like=g)
    check(lib.radmmm_dact_mul_transposed(ptr(g), g.shape[1], ptr(saved), saved.shape[1] if saved is not None else 0, C, B, T, Tp,
                                         _TS_FRONT, act, scale, ptr(yh), ptr(yl), yh.shape[1] if yh is not None else 0,
                                         split_opts(fmt, x8_exp, sat_flag, ylo16), ptr(oh), ptr(ol), ldk, ptr(part), stream()),
          "dact_mul_transposed")
    if defer is not None:
        defer.add(part, sums, nparts, C)
    else:
        check(lib.radmmm_colsum_final(ptr(part), ptr(sums), nparts, C, stream()), "colsum_final")
    return (oh, ol, None, None, Kt), sums


def dact_mul_rows_multi(g, saved, C, B, T, act, scale, dst, lo16s, fmt, x8_exp, sat_flag, sum_outs, defer=None):
    """y_j = g * act'(saved[j]) for up to four saved tensors in ONE pass over g (radmmm_dact_mul_rows_multi): the split pairs
    dst[j] = (yh, yl) (+ lo16s[j]) and the column sums -> list of sums [C] (final after defer.flush() when deferred)."""
    n = len(saved)
    nparts = B * (-(-T // 64))
    parts = [_empty(nparts, C, like=g) for _ in range(n)]
    sums = [so if (so is not None and so.numel() == C) else _empty(C, like=g) for so in sum_outs]
    items = (L.DactItem * n)(*[L.DactItem(ptr(saved[j]), ptr(dst[j][0]), ptr(dst[j][1]), ptr(lo16s[j]), ptr(parts[j])) for j in range(n)])
    check(lib.radmmm_dact_mul_rows_multi(ptr(g), g.shape[1], items, n, saved[0].shape[1], C, B, T, act, scale, dst[0][0].shape[1],
                                         split_opts(fmt, x8_exp, sat_flag, None), stream()), "dact_mul_rows_multi")
    for j in range(n):
        if defer is not None:
            defer.add(parts[j], sums[j], nparts, C)
        else:
            check(lib.radmmm_colsum_final(ptr(parts[j]), ptr(sums[j]), nparts, C, stream()), "colsum_final")
    return sums


def transpose_split_act(x, C, B, T, lens, mask_mode, scale, role, need_odd=False, colsum=None, sum_out=None):
    """channels-last fp32 [B*T, ld] -> transposed zero-gapped split copy [C, ldk] (+ advanced copy).
    Buffers come from a small zero-initialised pool keyed by role: the pads are never written, the data
    and the gaps are rewritten on every call (single stream => reuse is ordered).
    colsum = (row_weight, lens, taps, dil): also return the weighted column sums of x (the bias
    gradient) computed in the same pass -> (copy tuple, sums [C])."""
    bufs, Tp, Kt, ldk = _ts_buffers(x.device, C, B, T, role, need_odd)
    oh, ol = bufs[0], bufs[1]
    o1h, o1l = (bufs[2], bufs[3]) if need_odd else (None, None)
    if colsum is None:
        check(lib.radmmm_transpose_split_act(ptr(x), x.shape[1], C, B, T, Tp, _TS_FRONT, ptr(lens), mask_mode, scale, ptr(oh),
                                             ptr(ol), ptr(o1h), ptr(o1l), ldk, stream()), "transpose_split_act")
        return oh, ol, o1h, o1l, Kt
    weight, wlens, taps, dil = colsum
    nparts = B * (-(-Tp // 64))
    part = _empty(nparts, C, like=x)
    sums = sum_out if (sum_out is not None and sum_out.numel() == C) else _empty(C, like=x)
    assert mask_mode == 0 or wlens is lens or wlens is None
    check(lib.radmmm_transpose_split_act_colsum(ptr(x), x.shape[1], C, B, T, Tp, _TS_FRONT, ptr(lens if mask_mode else wlens),
                                                mask_mode, scale, ptr(oh), ptr(ol), ptr(o1h), ptr(o1l), ldk, ptr(part),
                                                weight, taps, dil, stream()), "transpose_split_act_colsum")
    check(lib.radmmm_colsum_final(ptr(part), ptr(sums), nparts, C, stream()), "colsum_final")
    return (oh, ol, o1h, o1l, Kt), sums


def wgrad_h3_slabs(gy_t, x_t, Mc, Nc, ldp, taps, dil, acc_scale, nprod=3):
    """gy_t / x_t: results of transpose_split_act -> P [S, taps, Mc, ldp] fp32 slabs."""
    gh, gl, _, _, Kt = gy_t
    xh, xl, x1h, x1l, Kt2 = x_t
    assert Kt == Kt2
    tiles = int(lib.radmmm_wgrad_h3_tiles(Mc, Nc, taps))
    S = pick_splits(tiles, Kt, slots=int(lib.radmmm_gemm_cu_slots()))            # one workgroup per CU
    if debug_env("RADMMM_WGRAD_SPLITS"):              # experiments
        S = int(debug_env("RADMMM_WGRAD_SPLITS"))
    P = torch.empty(S, taps, Mc, ldp, device=gh.device, dtype=torch.float32)
    if ldp != Nc:
        P.zero_()
    check(lib.radmmm_wgrad_h3(ptr(gh), ptr(gl), ptr(xh), ptr(xl), ptr(x1h), ptr(x1l), Kt + 2 * _TS_FRONT, _TS_FRONT, Kt, ptr(P), ldp, P.stride(0),
                              Mc, Nc, taps, dil, S, acc_scale, nprod, stream()), "wgrad_h3")
    return P


def wgrad_rm_slabs(gy_pair, x_pair, B, T, Mc, Nc, taps, dil, acc_scale, lens=None):
    """Weight gradient from ROW-major split pairs (hi, lo fp16 [B*T, ld]) -> P [S, taps, Mc, Nc] fp32 slabs
    (radmmm_wgrad_rm: contraction over the frames, transposition in the LDS read, no transposed copies).
    lens (int32 [B]): x counts as zero at frames >= length (partial padding)."""
    gh, gl = gy_pair
    xh, xl = x_pair
    R = B * T
    assert gh.shape[0] >= R and xh.shape[0] >= R and gh.dtype == torch.float16 and xl.dtype == torch.float16
    tiles = int(lib.radmmm_wgrad_rm_tiles(Mc, Nc, taps))
    S = pick_splits(tiles, R, slots=int(lib.radmmm_gemm_cu_slots()))              # one workgroup per CU
    P = torch.empty(S, taps, Mc, Nc, device=gh.device, dtype=torch.float32)
    assert gh.stride(1) == 1 and xh.stride(1) == 1 and gl.stride(0) == gh.stride(0) and xl.stride(0) == xh.stride(0)
    check(lib.radmmm_wgrad_rm(ptr(gh), ptr(gl), gh.stride(0), ptr(xh), ptr(xl), xh.stride(0), R, T, ptr(lens),
                              1 if lens is not None else 0, ptr(P), Nc, P.stride(0), Mc, Nc, taps, dil, S, acc_scale, stream()),
          "wgrad_rm")
    return P


def wgrad_rm8_slabs(gy_pair, g8_exp, x_pair, x8_exp, B, T, Mc, Nc, taps, dil, acc_scale, lens=None):
    """Weight gradient under the FP8-cross scheme from ROW-major (hi fp16, 8-bit cross array) pairs [B*T, ld]
    (radmmm_wgrad_rm8: GYh.Xh on the f16 pipe, both cross terms in one block-scaled FP8 MFMA) -> P [S, taps, Mc, Nc]."""
    gh, gx = gy_pair
    xh, xx = x_pair
    R = B * T
    assert gh.shape[0] >= R and xh.shape[0] >= R and gh.dtype == torch.float16 and xh.dtype == torch.float16
    tiles = int(lib.radmmm_wgrad_rm_tiles(Mc, Nc, taps))
    S = pick_splits(tiles, R, slots=int(lib.radmmm_gemm_cu_slots()))              # one workgroup per CU
    P = torch.empty(S, taps, Mc, Nc, device=gh.device, dtype=torch.float32)
    assert gh.stride(1) == 1 and xh.stride(1) == 1 and gx.stride(0) == gh.stride(0) and xx.stride(0) == xh.stride(0)
    check(lib.radmmm_wgrad_rm8(ptr(gh), ptr(gx), gh.stride(0), g8_exp, ptr(xh), ptr(xx), xh.stride(0), x8_exp, R, T, ptr(lens),
                               1 if lens is not None else 0, ptr(P), Nc, P.stride(0), Mc, Nc, taps, dil, S, acc_scale, stream()),
          "wgrad_rm8")
    return P


def _all_reduce_or(flags: torch.Tensor) -> torch.Tensor:
    """bitwise OR of small non-negative int32 flag words over the ranks of the default process group.  RCCL / NCCL have no
    BOR reduction (torch raises "Cannot use ReduceOp.BOR with NCCL"): the low 16 bits go out as 0/1 words under MAX."""
    sh = torch.arange(16, device=flags.device, dtype=flags.dtype)
    bits = (flags.reshape(-1, 1) >> sh) & 1
    torch.distributed.all_reduce(bits, op=torch.distributed.ReduceOp.MAX)
    return (bits << sh).sum(1).to(flags.dtype).reshape(flags.shape)


def scale_lag() -> int:
    """number of training passes between a backward pass and the adoption of the scale statistics it produced (GradScale);
    2 = the pass after next.  RADMMM_SCALE_LAG=1 adopts them in the very next pass (one blocking wait per step)."""
    return int(os.environ.get("RADMMM_SCALE_LAG", "2"))


class GradScale:
    """Scale state of the split GRADIENT tensors of one module (the decoder owns one and hands it to every flow step):
    a power-of-two S with amax * S in [8, 16) (2^12 of fp16 headroom above the first gradient's maximum), the exponent of
    the gradients' 8-bit cross-term parts (`x8_grad_exp`), two device flag words OR-ed by the split producers of the
    forward (flag[0]) and of the backward (flag[1]) -- bit 0: an element exceeded the fp16 range and was clamped; bit 1: an
    element's 8-bit cross-term parts exceeded e4m3's 448 (that element keeps single-fp16-product accuracy, ~5e-4, instead
    of the scheme's 4e-5); bits 2..7, one-hot: by how many powers of two (include/radmmm_hip.h) -- and the bookkeeping that
    keeps all of it off the host's critical path.

    Steady state never waits for the device: the first backward node of pass k queues `amax(|g|)` of its incoming
    gradient; forward k + 1 queues an asynchronous copy of (amax, flags) to pinned memory behind an event and clears the
    flags; forward k + SCALE_LAG (default 2) ADOPTS that publication -- the new S and, when the backward's 8-bit parts
    saturated, an exponent lowered by the reported level (`x8_adaptations` counts them; the exponent of a pass is fixed when
    its first node asks for the scale, so producers and consumers of a pass agree).  The adoption is at a FIXED lag, behind
    `event.synchronize()` on an event that is a whole training step old by then: which pass first runs with a new scale does
    not depend on host timing, so a run is bitwise repeatable and every rank follows the same rule (rounds 2-4 adopted
    "whenever `event.query()` happened to succeed": two identical runs could differ at the 1e-5 level).  The wait is over
    unless the host has queued more than SCALE_LAG - 1 steps ahead of the device; then it holds the host to that lead, which
    still leaves the device a full step of queued work.  Only the very first pass of a module reads a value synchronously.

    What happens to a pass that clamped (`reports`, counters `saturated_passes` / `x8_saturated_passes` /
    `nonfinite_passes`):
      * default: one RuntimeWarning per process and kind, the scale is refreshed, training goes on (a raise one step
        later would kill an AMP loop that is about to skip the step anyway, and under DDP only the affected rank would
        raise and the others hang in the next collective).  A gradient pass whose 8-bit parts saturated is counted, the
        exponent adapts, and the warning is kept for the cases adaptation cannot fix: the forward's activations (fixed
        exponent) or a gradient exponent already at its floor;
      * RADMMM_CHECK_SATURATION=1 (or strict=True): FloatingPointError for the fp16 clamp -- synchronously after every flow
        step (`check()`), and for the deferred report.  With a process group active every decision is rank-independent:
        the flags are OR-all-reduced before they are looked at (in `check()`, and in every training forward, which then
        waits for the previous copy instead of polling it), so every rank raises together and the collective sequence is
        the same on all ranks.
    Non-finite incoming gradients (an fp16-AMP GradScaler overflow step): the split producers clamp NaN / Inf to finite
    values, so the pass's amax is tracked with NaN propagation and `poison` (device fp32[1] = amax * 0: 0, or NaN) is
    added to every weight-norm gain gradient of the pass (radmmm_weightnorm_bwd): GradScaler / clip_grad_norm_ see the
    non-finite step and skip it as they would with the reference; S keeps its previous value."""

    def __init__(self, strict: Optional[bool] = None):
        self.S = None
        self.flags = None         # device int32[2]: forward / backward producers
        self.amax = None          # device fp32[1]
        self.poison = None        # device fp32[1]: 0, or NaN when this pass's incoming gradient is not finite
        self._inflight = collections.deque()   # publications on their way: dict(host pinned fp32[3], event, exp, passno)
        self._free = []           # recycled (host, event) pairs
        self._fwd_id = 0
        self._bwd_id = -1
        self._stats_fwd = -1      # forward id whose backward produced the device stats
        self.strict = strict
        self.x8_grad_exp = X8_GRAD_EXP
        self.saturated_passes = 0
        self.x8_saturated_passes = 0
        self.x8_adaptations = 0
        self.nonfinite_passes = 0

    _warned = set()

    # the state is per process and per device (a CUDA event, pinned memory): copies / pickles of the owning module start
    # fresh (copy.deepcopy(model) for an EMA copy, torch.save(model))
    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def __deepcopy__(self, memo):
        return GradScale(self.strict)

    # dict-like compatibility ("S")
    def get(self, k, d=None):
        return self.S if k == "S" and self.S is not None else d

    def __setitem__(self, k, v):
        assert k == "S"
        self.S = v

    @property
    def flag(self):
        """forward producers' flag word (device int32[1] view)"""
        return None if self.flags is None else self.flags[0:1]

    @property
    def flag_bwd(self):
        return None if self.flags is None else self.flags[1:2]

    def _ensure(self, dev):
        if self.flags is None or self.flags.device != dev:
            self.flags = torch.zeros(2, device=dev, dtype=torch.int32)
            self.amax = torch.zeros(1, device=dev, dtype=torch.float32)
            self.poison = torch.zeros(1, device=dev, dtype=torch.float32)
            self._inflight.clear()
            self._free = []

    def _is_strict(self) -> bool:
        return self.strict if self.strict is not None else os.environ.get("RADMMM_CHECK_SATURATION", "0") == "1"

    @staticmethod
    def _distributed() -> bool:
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    @staticmethod
    def _pow2(amax: float) -> float:
        if not (amax > 0 and math.isfinite(amax)):
            return 1.0
        return float(2.0 ** max(-40, min(40, math.floor(math.log2(16.0 / amax)))))

    @staticmethod
    def _x8_level(flag: int) -> int:
        """highest saturation level (1 .. 6) in a flag word, 0 when its 8-bit parts stayed in range"""
        lv = (flag >> 2) & 0x3f
        return lv.bit_length() if (flag & 2) else 0

    def _report(self, kind: str, msg: str):
        if self._is_strict() and kind == "f16":             # (the e4m3 saturation is a soft loss of accuracy: never an error)
            raise FloatingPointError(msg)
        if kind not in GradScale._warned:
            GradScale._warned.add(kind)
            import warnings
            warnings.warn(msg + "  (reported once per process; RADMMM_CHECK_SATURATION=1 turns it into an error)", RuntimeWarning)

    def _account_x8(self, f_fwd: int, f_bwd: int, ran_with_exp):
        """8-bit saturation of a finished pass: count it, lower the gradient exponent by the reported level, warn about what
        adaptation cannot fix"""
        lv_f, lv_b = self._x8_level(f_fwd), self._x8_level(f_bwd)
        if not (lv_f or lv_b):
            return
        self.x8_saturated_passes += 1
        if lv_b and ran_with_exp is not None:
            new = max(X8_GRAD_EXP_MIN, min(self.x8_grad_exp, ran_with_exp - lv_b))
            if new != self.x8_grad_exp:
                self.x8_grad_exp = new
                self.x8_adaptations += 1
        if lv_f or (lv_b and (ran_with_exp is None or ran_with_exp <= X8_GRAD_EXP_MIN)):
            self._report("x8", "FP8 cross terms saturated in an earlier pass: split-operand elements beyond e4m3's range ("
                               + ("activations above ~%g" % (448.0 / 2.0 ** X8_ACT_EXP) if lv_f else
                                  "gradient elements with the exponent at its floor") +
                               ") lost their cross-term correction (single-fp16-product accuracy, ~5e-4, for those elements); "
                               "RADMMM_PRECISION=h3 has no such limit")

    def _consume(self, pub):
        """host copy is complete: adopt the scale, report what the producers flagged in that earlier pass"""
        host = pub["host"]
        amax, f_fwd, f_bwd = float(host[0]), int(host[1]), int(host[2])
        ran_with = pub["exp"]
        self._free.append((host, pub["event"]))
        if not math.isfinite(amax):
            # non-finite upstream gradient (AMP overflow step): that pass was poisoned (see the class docstring); keep S,
            # and do not report the clamping the NaN / Inf values caused
            self.nonfinite_passes += 1
            self._report("nonfinite", "non-finite gradient entered the split-operand backward: the pass's weight-norm gain "
                                      "gradients were set to NaN, the gradient scale keeps its value")
            return
        if amax > 0:
            self.S = self._pow2(amax)
        self._account_x8(f_fwd, f_bwd, ran_with)
        if (f_fwd | f_bwd) & 1:
            self.saturated_passes += 1
            self._report("f16", "split-f16 gradient saturated in an earlier backward pass: a gradient element exceeded 2^12 x the "
                                "first gradient's maximum and was clamped (that pass's gradients are not exact; the scale has been "
                                "refreshed)")

    def new_forward(self, dev):
        """called by the owning module at the start of a training forward"""
        self._ensure(dev)
        sync_ranks = self._is_strict() and self._distributed()
        if self._stats_fwd == self._fwd_id:
            # publish the stats of the backward that followed the previous forward (pass number _fwd_id), then re-arm the flags
            fl = self.flags
            if sync_ranks:
                fl = _all_reduce_or(self.flags)                                              # every rank raises together
            host, event = self._free.pop() if self._free else (torch.zeros(3, dtype=torch.float32).pin_memory(), torch.cuda.Event())
            host[0:1].copy_(self.amax, non_blocking=True)
            host[1:3].copy_(fl.float(), non_blocking=True)
            event.record()
            self.flags.zero_()
            self._inflight.append({"host": host, "event": event, "exp": self._bwd_exp, "passno": self._fwd_id})
        # adopt, at a fixed lag, what earlier passes published: this forward opens pass _fwd_id + 1 (never a poll: the
        # pass that first runs with the new scale must not depend on host timing -- nor on the rank)
        lag = max(1, scale_lag())
        while self._inflight and (self._fwd_id + 1) - self._inflight[0]["passno"] >= lag:
            pub = self._inflight.popleft()
            pub["event"].synchronize()
            self._consume(pub)
        self._fwd_id += 1

    _bwd_exp = None               # gradient exponent of the most recent backward pass

    def scale(self, g: torch.Tensor) -> float:
        """S for this backward pass; the first node of the pass (re)initialises the pass"""
        self._ensure(g.device)
        if self._bwd_id != self._fwd_id:
            self._bwd_id = self._fwd_id
            if self.S is None:                              # first pass of this module: one synchronisation
                self.S = self._pow2(float(g.abs().max()))
            torch.amax(g.detach().abs().reshape(-1), dim=0, keepdim=True, out=self.amax)     # (propagates NaN)
            torch.mul(self.amax, 0.0, out=self.poison)                                     # 0, or NaN for Inf / NaN
            self._stats_fwd = self._fwd_id
            self._bwd_exp = self.x8_grad_exp                # fixed for the whole pass: producers and consumers agree
        return self.S

    def grad_exp(self) -> int:
        """exponent of the 8-bit cross-term parts of THIS backward pass's gradient tensors (valid after scale())"""
        return self._bwd_exp if self._bwd_exp is not None else self.x8_grad_exp

    def check(self) -> None:
        """synchronous: raise FloatingPointError if a split producer has clamped since the flags were last cleared (fp16
        range; the softer e4m3 saturation of the cross terms is counted, adapts the gradient exponent and is warned about,
        see `reports`).  With a process group active the flags are OR-all-reduced first: every rank raises together."""
        while self._inflight:
            pub = self._inflight.popleft()
            pub["event"].synchronize()
            self._consume(pub)
        if self.flags is not None:
            fl = self.flags
            if self._distributed():
                fl = _all_reduce_or(self.flags)
            f_fwd, f_bwd = (int(v) for v in fl.tolist())
            if f_fwd | f_bwd:
                self.flags.zero_()
            self._account_x8(f_fwd, f_bwd, self._bwd_exp)
            if (f_fwd | f_bwd) & 1:
                raise FloatingPointError("split-f16 gradient saturated: a gradient element exceeds 2^12 x the first gradient's "
                                         "maximum of this backward pass and was clamped")


def module_scale_box(module, new_forward_on=None) -> "GradScale":
    """The GradScale a module outside the decoder (text encoder, alignment attention) keeps for its stand-alone conv_norm
    calls: with a throw-away dict every conv's backward reads its gradient's maximum on the host -- a synchronisation per
    conv and step.  new_forward_on = device: mark the start of a training forward (publishes the previous pass's stats)."""
    if debug_env("RADMMM_CONV_OWN_SCALE", "0") == "1":      # A/B aid: every conv's backward scales by its OWN gradient's maximum
        return {}
    box = module.__dict__.get("_scale_box")
    if box is None:
        box = module.__dict__["_scale_box"] = GradScale()
    if new_forward_on is not None and torch.is_grad_enabled():
        box.new_forward(new_forward_on)
    return box


def grad_scale(box, g: torch.Tensor) -> float:
    """Power-of-two scale of the split GRADIENT tensors for this backward pass (see GradScale); a plain dict `box`
    (stand-alone conv_norm calls) fixes it from the first gradient with one host sync."""
    if isinstance(box, GradScale):
        return box.scale(g)
    if box.get("S") is None:
        box["S"] = GradScale._pow2(float(g.abs().max()))
    return box["S"]


def sat_flag_of(box) -> Optional[torch.Tensor]:
    """flag word of the FORWARD's split producers"""
    return box.flag if isinstance(box, GradScale) else None


def sat_flag_bwd_of(box) -> Optional[torch.Tensor]:
    """flag word of the BACKWARD's split producers (its 8-bit saturation level drives the gradient exponent)"""
    return box.flag_bwd if isinstance(box, GradScale) else None


def grad_x8_exp(box) -> int:
    """exponent of the gradients' 8-bit cross-term parts for the backward pass in progress (call after grad_scale)"""
    return box.grad_exp() if isinstance(box, GradScale) else X8_GRAD_EXP


def poison_of(box) -> Optional[torch.Tensor]:
    return box.poison if isinstance(box, GradScale) else None


def check_saturation(box) -> None:
    """RADMMM_CHECK_SATURATION=1 (debugging aid, one host sync per call): raise as soon as a split tensor hit the fp16
    clamp instead of at the next pass."""
    if os.environ.get("RADMMM_CHECK_SATURATION", "0") == "1" and isinstance(box, GradScale):
        box.check()


def traced_bwd(fn):
    """backward of a flow step inside a roctx range `flow<i>.bwd` (RADMMM_ROCTX=1, rad_mmm_amd/_trace.py); off: one test"""
    import functools
    from . import _trace

    @functools.wraps(fn)
    def wrapped(ctx, *grads):
        if not _trace.ENABLED:
            return fn(ctx, *grads)
        with _trace.trace_range("flow%s.bwd" % getattr(ctx, "meta", {}).get("flow_index", "?")):
            return fn(ctx, *grads)
    return wrapped


_SIDE_STREAMS: dict = {}


def side_stream(dev: torch.device) -> "torch.cuda.Stream":
    """The one side stream per device of the WN forward's pairing (AffineFlowStepH3Fn.forward): res_skip[j] (1x1) runs beside
    in_layer[j+1] (5 taps); created on first use, never synchronised with the host."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def ctx_acc_add(acc: dict, slot: int, alloc, write):
    """The affine flow steps' context gradients accumulate in ONE buffer per backward traversal (RADMMMFlow.forward makes
    `acc`, one per forward pass; `slot` = the step's ordinal among the affine steps).  write(dst, accumulate) launches the
    step's contribution.  Slot 0 -- the first affine step: its output feeds every later one, so autograd runs it LAST whenever it
    runs at all -- hands the sum to autograd and clears the state; the other steps return None.
    The state is keyed on autograd's graph-task id: a traversal that reaches only some steps (`inputs=` pruning,
    torch.autograd.grad on an early-exit output, an exception half way through) leaves a partial sum behind, and the NEXT
    traversal -- another id -- starts a fresh buffer instead of adding to the stale one; a pruned traversal that never reaches
    slot 0 drops a gradient nobody asked for (every path from the context to an input goes through slot 0's node or through a
    step that returns its own share below).  A step whose slot already contributed in this traversal (a node run twice:
    cannot happen in one graph task) raises rather than double-count."""
    task = torch._C._current_graph_task_id()
    if acc.get("task") != task:
        acc["task"], acc["buf"], acc["seen"] = task, None, set()
    if slot in acc["seen"]:
        raise RuntimeError("context-gradient accumulation: affine flow step %d ran twice in one backward traversal" % slot)
    first = acc["buf"] is None
    if first:
        acc["buf"] = alloc()
    buf = acc["buf"]
    write(buf, not first)
    acc["seen"].add(slot)
    if slot != 0:
        return None
    acc["task"], acc["buf"], acc["seen"] = None, None, set()
    return buf


class AffineFlowStepH3Fn(torch.autograd.Function):
    """Same contract as AffineFlowStepFn; the WN convs run on radmmm_rowgemm_h3 (split operands on the f16 / fp8 matrix
    cores, fp32 accumulate).  meta["nprod"]: 3 = split-f16 (three f16 products), 2 = f16 product + FP8 cross terms
    (second array of every row-major split pair is then the 8-bit cross array), 1 = fp16 throughput mode.
    The 160-wide invertible 1x1, all weight gradients (contraction over frames, always split-f16 on transposed copies
    made from the fp32 tensors) and every elementwise / reduction kernel stay fp32."""

    @staticmethod
    @amp_fwd
    def forward(ctx, meta, z_in, cond, lens, W_eff, b_eff, start_v, start_g, start_b, end_w, end_b,
                *layer_params):
        B, T, C, D, nl = meta["B"], meta["T"], meta["C"], meta["D"], meta["n_layers"]
        act, scaling, partial = meta["act"], meta["scaling"], meta["partial"]
        NPR = meta.get("nprod", 3)               # product scheme, see the class docstring
        box = meta["scale_box"]
        flag = sat_flag_of(box)
        fa = fmt_a(NPR)
        h = C // 2
        N = B * T
        Wc = start_v.shape[0]
        Kp = round_up(D + h, 32)
        in_p, res_p = layer_params[: 3 * nl], layer_params[3 * nl:]
        assert Wc % 32 == 0 and z_in.shape == (N, ZLD) and cond.shape == (N, D)
        inv_ws = 1.0 / W_SCALE
        # keyword sets shared by the launches of this pass: operand exponents in, output format out
        gin = dict(nprod=NPR, a8_exp=X8_ACT_EXP, b8_exp=X8_W_EXP, acc_scale=inv_ws, T=T, sat_flag=flag)
        gout = dict(split_fmt=fa, ch_x8_exp=X8_ACT_EXP, c2h_x8_exp=X8_ACT_EXP)

        # Weight gradients contract the ROW-major split pairs (hi, fp16 lo) of the activations / gradients directly
        # (radmmm_wgrad_rm): the pairs of X0 and of every hidden state are kept for backward; under the FP8-cross scheme
        # the GEMMs' second operand array is the 8-bit cross array, so the producers write the fp16 lo part as well.
        use_rm = (NPR in (2, 3) and T >= 32 and B <= 1024 and any(ctx.needs_input_grad) and
                  debug_env("RADMMM_WGRAD_RM", "1") != "0")
        # (NPR 2: radmmm_wgrad_rm8 contracts (hi, cross array) pairs -- no fp16 lo copies; RADMMM_DEBUG + RADMMM_WGRAD_RM8=0
        #  keeps the three-product gradient on (hi, fp16 lo) pairs for A/B runs)
        rm8 = use_rm and NPR == 2 and Wc % 32 == 0 and debug_env("RADMMM_WGRAD_RM8", "1") != "0"
        lo16 = (lambda: torch.empty(N, Wc, device=z_in.device, dtype=torch.float16)) if (use_rm and NPR == 2 and not rm8) else (lambda: None)
        # Round 5: with the FP8-cross weight gradient (rm8) NOTHING reads an fp32 copy of the WN input or of the hidden states
        # any more -- the GEMMs and the weight gradients take the split pairs, and the data gradient's softplus' factor is
        # rebuilt from the pair (radmmm_rowgemm_desc.dact_h / dact_x: hi + lo8 * 2^-(11+e), 2^-15 relative) -- so the
        # producers write the pair only (C = NULL): 8 instead of 16 bytes per hidden-state element leave the forward
        # epilogues, 311 MB per flow step at the benchmark size.  RADMMM_KEEP_FP32=1 (RADMMM_DEBUG) keeps round 4's copies.
        pair_only = bool(rm8 and debug_env("RADMMM_KEEP_FP32", "0") != "1")
        hid = (lambda: None) if pair_only else (lambda: _empty(N, Wc, like=z_in))
        z1 = _empty(N, ZLD, like=z_in)
        rowgemm(A=z_in, lda=ZLD, B=W_eff, ldb=ZLD, b_layout=0, C=z1, ldc=ZLD, M=N, N=ZLD, K=ZLD, T=T, bias=b_eff)
        X0 = None if pair_only else _empty(N, Kp, like=z_in)
        X0h, X0l = _halves(N, Kp, like=z_in)
        X0lo = torch.empty(N, Kp, device=z_in.device, dtype=torch.float16) if (use_rm and NPR == 2 and not rm8) else None
        check(lib.radmmm_wn_input_fwd(ptr(cond), D, ptr(z1), ZLD, ptr(X0), Kp, N, D, h, ptr(X0h), ptr(X0l),
                                      split_opts(fa, X8_ACT_EXP, None, X0lo), stream()), "wn_input_fwd")
        perm = (h, D, 0)
        # all conv weights of the flow step in one launch (round 4)
        specs = [(start_v, start_g, Kp, perm)]
        for j in range(nl):
            specs.append((in_p[3 * j], in_p[3 * j + 1], Wc, None))
            specs.append((res_p[3 * j], res_p[3 * j + 1], Wc, None))
        specs.append((end_w, None, Wc, None))
        sw = split_weights(specs, NPR)
        Wsh, Wsl, inv_s = sw[0]
        Wih, Wil, inv_i = [sw[1 + 2 * j][0] for j in range(nl)], [sw[1 + 2 * j][1] for j in range(nl)], [sw[1 + 2 * j][2] for j in range(nl)]
        Wrh, Wrl, inv_r = [sw[2 + 2 * j][0] for j in range(nl)], [sw[2 + 2 * j][1] for j in range(nl)], [sw[2 + 2 * j][2] for j in range(nl)]
        Weh, Wel, _ = sw[-1]

        H = [hid()]
        Hh, Hl = _halves(N, Wc, like=z_in)
        Hlo = lo16()
        rowgemm_h3(Ah=X0h, Al=X0l, lda_h=Kp, Bh=Wsh, Bl=Wsl, ldb_h=Kp, C=H[0], ldc=Wc, M=N, N=Wc, K=Kp,
                   bias=start_b, Ch=Hh, Cl=Hl, Clo=Hlo, ldch=Wc, ch_scale=1.0, **gin, **gout)
        pairs = [(Hh, Hlo if Hlo is not None else Hl)]
        out_pair = bool(use_rm and (rm8 or NPR == 3) and debug_env("RADMMM_END_WGRAD_RM", "1") != "0")
        res_src = bool(2 <= nl <= 4 and debug_env("RADMMM_RES_SRC", "1") != "0")
        # (no fp32 OUT at all when the backward takes OUT's split pair: the last layer's C2 stores are dropped)
        OUT = None if (res_src and out_pair) else _empty(N, Wc, like=z_in)
        OUTh, OUTl = _halves(N, Wc, like=z_in)
        R = []
        # Round 6 (measured before anything was built: tools/pair_overlap_probe.py,
        # profiles/r06_pair_overlap.txt): res_skip[j] and in_layer[j+1] both read only h_{j+1} (common.py:829-832), so the 1x1
        # launches of layers 0 .. nl-2 can run on a side stream BESIDE the next 5-tap launch -- what a grouped persistent grid
        # would do at best, with the hardware's own workgroup dispatcher.  One flow step's forward chain stand-alone: 1241 ->
        # 1190 us.  In the training step: 40.18 / 40.24 / 40.15 -> 40.13 / 40.16 / 40.16 ms, i.e. NOTHING (the 24 idle CUs and the
        # exposed 1x1 epilogues are not where the step's time is), so the pairing is OFF and this stays as the A/B switch
        # RADMMM_RES_STREAM=1 (RADMMM_DEBUG).  Same launches, same values; every tensor the side stream touches is kept alive
        # by this function until the main stream has joined it in front of the last layer's launch.
        pair_res = bool(res_src and NPR == 2 and nl >= 2 and z_in.is_cuda and debug_env("RADMMM_RES_STREAM", "0") == "1")
        main_st = torch.cuda.current_stream(z_in.device) if pair_res else None
        side_st = side_stream(z_in.device) if pair_res else None
        side_done: List = []
        for j in range(nl):
            d = 2 ** j
            kt = in_p[3 * j].shape[2]
            Hn = hid()
            Hnh, Hnl = _halves(N, Wc, like=z_in)
            Hnlo = lo16()
            rowgemm_h3(Ah=Hh, Al=Hl, lda_h=Wc, Bh=Wih[j], Bl=Wil[j], ldb_h=Wc, b_tap_stride_h=Wih[j].stride(0),
                       C=Hn, ldc=Wc, M=N, N=Wc, K=Wc, taps=kt, dil=d, sign=1, lens=lens,
                       a_mask_mode=1 if partial else 0, bias=in_p[3 * j + 2], pconv=1 if partial else 0, ratio_taps=kt,
                       ratio_dil=d, postmask=1, act=act, Ch=Hnh, Cl=Hnl, Clo=Hnlo, ldch=Wc, ch_scale=1.0, **gin, **gout)
            H.append(Hn)
            Hh, Hl = Hnh, Hnl
            pairs.append((Hnh, Hnlo if Hnlo is not None else Hnl))
            Rj = _empty(N, Wc, like=z_in)
            last = j == nl - 1
            if pair_res and not last:
                # res_skip[j] on the side stream, behind in_layer[j] (it reads h_{j+1} and its own weights only); the main
                # stream goes on to in_layer[j+1] and meets the side stream again in front of the last layer's launch
                ev = torch.cuda.Event()
                ev.record(main_st)
                with torch.cuda.stream(side_st):
                    side_st.wait_event(ev)
                    rowgemm_h3(Ah=Hh, Al=Hl, lda_h=Wc, Bh=Wrh[j], Bl=Wrl[j], ldb_h=Wc, C=Rj, ldc=Wc, M=N, N=Wc, K=Wc,
                               bias=res_p[3 * j + 2], act=act, **gin, **gout)
                    done = torch.cuda.Event()
                    done.record(side_st)
                side_done.append(done)
                R.append(Rj)
                continue
            if pair_res:                                # (last layer: its epilogue sums the earlier layers' outputs)
                for done in side_done:
                    main_st.wait_event(done)
            if res_src:
                # Round 5: the skip sum is formed ONCE, by the last layer's epilogue, from the earlier layers' outputs
                # (radmmm_rowgemm_desc.c2_src: ((R0 + R1) + R2) + R3, the running sum's association and bits) -- the earlier
                # layers write their own output only, and with the end conv's weight gradient on OUT's split pair nobody reads
                # an fp32 OUT: 208 MB less per flow step than four read-modify-writes of it.
                if last:
                    rowgemm_h3(Ah=Hh, Al=Hl, lda_h=Wc, Bh=Wrh[j], Bl=Wrl[j], ldb_h=Wc, C=Rj, ldc=Wc, M=N, N=Wc, K=Wc,
                               bias=res_p[3 * j + 2], act=act, C2=OUT, ldc2=Wc, c2_src=R, C2h=OUTh, C2l=OUTl, ldc2h=Wc,
                               c2h_scale=1.0, **gin, **gout)
                else:
                    rowgemm_h3(Ah=Hh, Al=Hl, lda_h=Wc, Bh=Wrh[j], Bl=Wrl[j], ldb_h=Wc, C=Rj, ldc=Wc, M=N, N=Wc, K=Wc,
                               bias=res_p[3 * j + 2], act=act, **gin, **gout)
            else:
                rowgemm_h3(Ah=Hh, Al=Hl, lda_h=Wc, Bh=Wrh[j], Bl=Wrl[j], ldb_h=Wc, C=Rj, ldc=Wc, M=N, N=Wc,
                           K=Wc, bias=res_p[3 * j + 2], act=act, C2=OUT, ldc2=Wc, c2_accum=1 if j > 0 else 0,
                           C2h=OUTh if last else None, C2l=OUTl if last else None, ldc2h=Wc, c2h_scale=1.0, **gin, **gout)
            R.append(Rj)
        O = _empty(N, ZLD, like=z_in)
        rowgemm_h3(Ah=OUTh, Al=OUTl, lda_h=Wc, Bh=Weh, Bl=Wel, ldb_h=Wc, C=O, ldc=ZLD, M=N, N=C, K=Wc,
                   bias=end_b, **gin)
        z_out = _empty(N, ZLD, like=z_in)
        log_s = _empty(N, h, like=z_in)
        check(lib.radmmm_affine_coupling_fwd(ptr(O), ZLD, ptr(z1), ZLD, ptr(z_out), ptr(log_s), N, h, scaling,
                                             stream()), "affine_coupling_fwd")
        ctx.meta = meta
        ctx.nl = nl
        ctx.use_rm = use_rm
        ctx.rm8 = rm8
        ctx.pair_only = pair_only
        if pair_only:                             # (placeholders keep the saved-tensor layout; never read on this path)
            X0 = z1[:0]
            H = [z1[:0]] * (nl + 1)
        rm_saved = [X0h, X0lo if X0lo is not None else X0l, *[t for pr in pairs for t in pr]] if use_rm else []
        # the end conv's weight gradient contracts gO with OUT: with OUT's split pair kept (52 MB per flow step) it runs on
        # the same row-major kernels as every other weight gradient of the step instead of the fp32-MFMA one (50 -> ~25 us)
        ctx.out_pair = out_pair
        if ctx.out_pair:
            rm_saved += [OUTh, OUTl]
        if OUT is None:
            OUT = z1[:0]                              # (placeholder in the saved-tensor layout; never read with out_pair)
        ctx.save_for_backward(z_in, z1, X0, OUT, O, lens, W_eff, start_v, start_g, end_w, Wsh, Wsl, inv_s, Weh, Wel,
                              start_b, end_b,
                              *H, *R, *Wih, *Wil, *inv_i, *Wrh, *Wrl, *inv_r, *rm_saved, *layer_params)
        return z_out, log_s

    @staticmethod
    @amp_bwd
    @traced_bwd
    def backward(ctx, g_zout, g_logs):
        meta, nl = ctx.meta, ctx.nl
        NPR = meta.get("nprod", 3)
        WPR = 3 if NPR == 2 else NPR             # weight gradients: split-f16 (or the throughput mode's single product)
        B, T, C, D = meta["B"], meta["T"], meta["C"], meta["D"]
        act, scaling, partial = meta["act"], meta["scaling"], meta["partial"]
        sv = ctx.saved_tensors
        z_in, z1, X0, OUT, O, lens, W_eff, start_v, start_g, end_w, Wsh, Wsl, inv_s, Weh, Wel, start_b, end_b = sv[:17]
        p = 17
        H = sv[p: p + nl + 1]; p += nl + 1
        R = sv[p: p + nl]; p += nl
        Wih = sv[p: p + nl]; p += nl
        Wil = sv[p: p + nl]; p += nl
        inv_i = sv[p: p + nl]; p += nl
        Wrh = sv[p: p + nl]; p += nl
        Wrl = sv[p: p + nl]; p += nl
        inv_r = sv[p: p + nl]; p += nl
        use_rm = ctx.use_rm
        if use_rm:                                # row-major split pairs (hi, fp16 lo) of X0 and of the hidden states
            X0pair = (sv[p], sv[p + 1]); p += 2
            Hpair = [(sv[p + 2 * i], sv[p + 2 * i + 1]) for i in range(nl + 1)]; p += 2 * (nl + 1)
        OUTpair = None
        if getattr(ctx, "out_pair", False):
            OUTpair = (sv[p], sv[p + 1]); p += 2
        layer_params = sv[p:]
        in_p, res_p = layer_params[: 3 * nl], layer_params[3 * nl:]
        h = C // 2
        N = B * T
        Wc = start_v.shape[0]
        Kp = Wsh.shape[2]
        g_zout = g_zout.contiguous()
        if g_logs is not None:
            g_logs = g_logs.contiguous()
        box = meta["scale_box"]
        rm8 = ctx.rm8
        pair_only = getattr(ctx, "pair_only", False)
        lo16 = (lambda: torch.empty(N, Wc, device=z_in.device, dtype=torch.float16)) if (use_rm and NPR == 2 and not rm8) else (lambda: None)

        def wg_rm(gpair, xpair_, Mc_, Nc_, taps_, dil_, lens_=None):
            if rm8:
                return wgrad_rm8_slabs(gpair, GE, xpair_, X8_ACT_EXP, B, T, Mc_, Nc_, taps_, dil_, 1.0 / SG, lens_)
            return wgrad_rm_slabs(gpair, xpair_, B, T, Mc_, Nc_, taps_, dil_, 1.0 / SG, lens_)
        SG = grad_scale(box, g_zout)
        GE = grad_x8_exp(box)                      # exponent of this pass's 8-bit gradient parts (adapts to saturation reports)
        flag = sat_flag_bwd_of(box)
        poison = poison_of(box)
        fa = fmt_a(NPR)
        inv_acc = 1.0 / (SG * W_SCALE)
        gin = dict(nprod=NPR, a8_exp=GE, b8_exp=X8_W_EXP, acc_scale=inv_acc, T=T, sat_flag=flag)
        gout = dict(split_fmt=fa, ch_x8_exp=GE)

        gO = torch.zeros(N, ZLD, device=z_in.device, dtype=torch.float32)      # columns >= C stay zero (K = ZLD)
        gz1 = _empty(N, ZLD, like=z_in)
        check(lib.radmmm_affine_coupling_bwd(ptr(O), ZLD, ptr(z1), ZLD, ptr(g_zout), ptr(g_logs), ptr(gO), ptr(gz1),
                                             N, h, scaling, stream()), "affine_coupling_bwd")
        g_end_b = colsum(gO, C, out=grad_out(end_b))
        g_end_w = grad_out(end_w)
        gOh, gOl = split_f16(gO, ZLD, SG, ZLD, NPR, GE, flag)
        if OUTpair is not None:
            torch.sum(wg_rm((gOh, gOl), OUTpair, C, Wc, 1, 1), dim=0, out=g_end_w.view(1, C, Wc))
        else:
            torch.sum(wgrad_slabs(gO, C, OUT, Wc, Wc, T, None), dim=0, out=g_end_w.view(1, C, Wc))
        # the transposed weights of the whole flow step in ONE launch (round 4; FP8-cross scheme): end conv, per layer the
        # in_layer stack (with the free slot of the fused data gradient filled by res_skip j-1's weights) and the start conv
        fuse = debug_env("RADMMM_FUSED_DGRAD", "1") != "0"
        pre = None
        if NPR == 2 and debug_env("RADMMM_MULTI_TRANSPOSE", "1") != "0":
            specs, slot = [(Weh, Wel, C, Wc, ZLD, None)], {"end": 0}
            stacks = {}
            for j in range(nl - 1, -1, -1):
                kt_j = in_p[3 * j].shape[2]
                if fuse and j > 0:                       # keep_pair below: in_layer j's stack with one free slot
                    stacks[j] = _halves(kt_j + 1, Wc, Wc, like=z_in)
                    slot[("in", j)] = len(specs)
                    specs.append((Wih[j], Wil[j], Wc, Wc, Wc, (stacks[j][0][:kt_j], stacks[j][1][:kt_j])))
                else:
                    slot[("in", j)] = len(specs)
                    specs.append((Wih[j], Wil[j], Wc, Wc, Wc, None))
            for j in range(nl - 1, -1, -1):
                slot[("res", j)] = len(specs)
                if fuse and j + 1 < nl:                  # fused below: into the stack of in_layer j + 1
                    kt_n = in_p[3 * (j + 1)].shape[2]
                    specs.append((Wrh[j], Wrl[j], Wc, Wc, Wc, (stacks[j + 1][0][kt_n:], stacks[j + 1][1][kt_n:])))
                else:
                    specs.append((Wrh[j], Wrl[j], Wc, Wc, Wc, None))
            slot["start"] = len(specs)
            specs.append((Wsh, Wsl, Wc, Kp, Wc, None))
            done = transpose_splits(specs, NPR)
            pre = {k: done[i] for k, i in slot.items()}
            pre["stacks"] = stacks
        WeTh, WeTl = pre["end"] if pre else transpose_split(Weh, Wel, C, Wc, ZLD, NPR)      # [1][Wc][ZLD]
        gOUT = _empty(N, Wc, like=z_in)
        rowgemm_h3(Ah=gOh, Al=gOl, lda_h=ZLD, Bh=WeTh, Bl=WeTl, ldb_h=ZLD, C=gOUT, ldc=Wc, M=N, N=Wc, K=ZLD, **gin)
        g_in: List[Optional[torch.Tensor]] = [None] * (3 * nl)
        g_res: List[Optional[torch.Tensor]] = [None] * (3 * nl)
        G = None
        Gh = Gl = Glo = None
        # dL/dH_{j+1} = (k-tap data gradient of in_layer j+1) + (1x1 data gradient of res_skip j): the two GEMMs share their
        # output, so they run as ONE launch -- the 1x1 part is an extra K segment of the k-tap GEMM (radmmm_rowgemm_h3's
        # extra_tap) -- and dL/dH_{j+1} itself never exists in memory.  For that the split copies of g_conv_{j+1} and of
        # gQ_j live in one [2N, Wc] pair (rows [0, N) / [N, 2N)) and the transposed weights of in_layer j+1 and res_skip j
        # in one [taps + 1] tap stack.  RADMMM_FUSED_DGRAD=0 keeps the two-launch arrangement (A/B runs).
        # bias gradients of the in_layer convs and of the start conv: column sums of the data-gradient GEMMs' values before
        # their row scale, taken from the accumulators in the epilogue (radmmm_rowgemm_desc.colsum_out) instead of a pass
        # over the 52 MB output each.  RADMMM_FUSED_COLSUM=0: the separate radmmm_colsum launches (A/B runs).
        fuse_cs = use_rm and debug_env("RADMMM_FUSED_COLSUM", "1") != "0"
        cs_scratch = _empty(int(lib.radmmm_rowgemm_h3_colsum_scratch_floats(N, Wc)), like=z_in) if fuse_cs else None
        # round 5: the finals of all these partial sums (9 per flow step) are deferred into one launch (ColsumBatch);
        # RADMMM_COLSUM_BATCH=0 (RADMMM_DEBUG) keeps one radmmm_colsum_final launch each
        batch = ColsumBatch() if debug_env("RADMMM_COLSUM_BATCH", "1") != "0" else None

        def cs_args(param):
            gb = grad_out(param)
            gb = gb if (gb is not None and gb.numel() == Wc) else _empty(Wc, like=z_in)
            if batch is not None:                   # partial rows stay in scratch of their own until the batch is flushed
                return gb, dict(colsum_scratch=_empty(int(lib.radmmm_rowgemm_h3_colsum_scratch_floats(N, Wc)), like=z_in), _defer=gb)
            return gb, dict(colsum_out=gb, colsum_scratch=cs_scratch)

        def gemm_cs(**kw):
            """rowgemm_h3 whose column sums are deferred into `batch` when the launch can leave partial rows"""
            gb = kw.pop("_defer", None)
            if gb is not None:
                rows = L.rowgemm_h3_colsum_rows(**kw)
                if rows > 0:
                    rowgemm_h3(**kw)
                    batch.add(kw["colsum_scratch"], gb, rows, kw["N"])
                    return
                kw["colsum_out"] = gb
            rowgemm_h3(**kw)
        pair_h = pair_l = None           # [2N, Wc]: g_conv_{j+1} split in the first half, gQ_j goes into the second
        WT_prev = None                   # (WiT stack [kt+1][Wc][Wc] of layer j+1 with its last slot free, kt, dil)
        x_prev = None
        # Round 5: gQ_j = gOUT * act'(R_j) of ALL res/skip layers in one pass over gOUT (radmmm_dact_mul_rows_multi; gOUT is the
        # same 52 MB for the four layers).  Their destinations are the second halves of the layer pairs, which are therefore
        # allocated up front.  RADMMM_DACT_MULTI=0 (RADMMM_DEBUG): one launch per layer as before.
        multi = bool(use_rm and fuse and 2 <= nl <= 4 and act and debug_env("RADMMM_DACT_MULTI", "1") != "0")
        pre_pairs, gq_pre = {}, None
        if multi:
            for j in range(nl - 1, 0, -1):
                pre_pairs[j] = _halves(2 * N, Wc, like=z_in)
            top = _halves(N, Wc, like=z_in)
            gq_dst = [(pre_pairs[j + 1][0][N:], pre_pairs[j + 1][1][N:]) if j < nl - 1 else top for j in range(nl)]
            gq_lo = [lo16() for _ in range(nl)]
            gq_sums = dact_mul_rows_multi(gOUT, [R[j] for j in range(nl)], Wc, B, T, act, SG, gq_dst, gq_lo, fa, GE, flag,
                                          [grad_out(res_p[3 * j + 2]) for j in range(nl)], defer=batch)
            gq_pre = (gq_dst, gq_lo, gq_sums)
        for j in range(nl - 1, -1, -1):
            d = 2 ** j
            kt = in_p[3 * j].shape[2]
            fused = fuse and pair_h is not None
            if gq_pre is not None:
                (gQh, gQl), gQlo, g_res[3 * j + 2] = gq_pre[0][j], gq_pre[1][j], gq_pre[2][j]
                gy_t = None
            else:
                if fused:
                    gQh, gQl = pair_h[N:], pair_l[N:]
                else:
                    gQh, gQl = _halves(N, Wc, like=z_in)
                # through softplus of the res/skip branch: gQ = gOUT * act'(R_j), written as the dgrad GEMM's row-major split
                # operand, as the weight gradient's transposed split operand and as bias sums in one pass (no fp32 gQ)
                gQlo = lo16()
                gy_t, g_res[3 * j + 2] = dact_mul_transposed(gOUT, R[j], Wc, B, T, act, SG, None if use_rm else "gy", gQh, gQl, fa,
                                                             GE, flag, sum_out=grad_out(res_p[3 * j + 2]), ylo16=gQlo, defer=batch)
            if use_rm:
                slabs = wg_rm((gQh, gQlo if gQlo is not None else gQl), Hpair[j + 1], Wc, Wc, 1, 1)
            else:
                # H[j+1]'s transposed copy may still be in the pool from layer j+1's in_layer weight gradient (x_prev, set
                # below only when that length-masked copy is IDENTICAL to the unmasked one this gradient needs)
                x_t = x_prev if x_prev is not None else transpose_split_act(H[j + 1], Wc, B, T, None, 0, 1.0, "x")
                slabs = wgrad_h3_slabs(gy_t, x_t, Wc, Wc, Wc, 1, 1, 1.0 / SG, WPR)
            g_res[3 * j], g_res[3 * j + 1] = weightnorm_bwd(res_p[3 * j], res_p[3 * j + 1], inv_r[j], slabs, Wc, poison=poison)
            keep_pair = fuse and j > 0               # g_conv_j's split copy becomes the first half of the next pair
            nh, nlo = pre_pairs[j] if (keep_pair and j in pre_pairs) else _halves(2 * N if keep_pair else N, Wc, like=z_in)
            gch, gcl = nh[:N], nlo[:N]
            gclo = lo16()
            cs_here = fuse_cs and (fused or G is None)         # (a launch with an `add` input keeps the separate pass)
            # the fp32 copy of the pre-activation gradient is read by nobody when its bias sums come out of the epilogue
            # and the weight gradient contracts the pair (round 5: C = NULL, 52 MB per launch less to write)
            g_conv = None if (pair_only and cs_here and use_rm) else _empty(N, Wc, like=z_in)
            if pair_only:                            # softplus' of the hidden state from its split pair (hi, cross array)
                dsrc = dict(dact_h=Hpair[j + 1][0], dact_x=Hpair[j + 1][1], lddact_h=Wc, dact_x8_exp=X8_ACT_EXP)
            else:
                dsrc = dict(dact_src=H[j + 1], lddact=Wc)
            epi = dict(C=g_conv, ldc=Wc, M=N, N=Wc, K=Wc, lens=lens, dact=act, **dsrc,
                       rowscale=2 if partial else 1, ratio_taps=kt, ratio_dil=d, Ch=gch, Cl=gcl, Clo=gclo, ldch=Wc, ch_scale=SG)
            gb, cs = cs_args(in_p[3 * j + 2]) if cs_here else (None, {})
            if fused:
                WTh, WTl, ktn, dn = WT_prev
                if pre is None:
                    transpose_split(Wrh[j], Wrl[j], Wc, Wc, Wc, NPR, out=(WTh[ktn:], WTl[ktn:]))
                gemm_cs(Ah=pair_h, Al=pair_l, lda_h=Wc, Bh=WTh, Bl=WTl, ldb_h=Wc, b_tap_stride_h=WTh.stride(0), taps=ktn,
                        dil=dn, sign=-1, a_mask_mode=0, extra_tap=1, extra_a_rows=N, **epi, **gin, **gout, **cs)
            else:
                WrTh, WrTl = pre[("res", j)] if pre else transpose_split(Wrh[j], Wrl[j], Wc, Wc, Wc, NPR)
                gemm_cs(Ah=gQh, Al=gQl, lda_h=Wc, Bh=WrTh, Bl=WrTl, ldb_h=Wc, add=G, ldadd=Wc, **epi, **gin, **gout, **cs)
            if use_rm:
                g_in[3 * j + 2] = gb if cs_here else colsum(g_conv, Wc, 2 if partial else 0, T, lens, kt, d,
                                                            out=grad_out(in_p[3 * j + 2]))
                slabs = wg_rm((gch, gclo if gclo is not None else gcl), Hpair[j], Wc, Wc, kt, d, lens if partial else None)
            elif (kt // 2) * d <= _TS_FRONT:
                gy_t, g_in[3 * j + 2] = transpose_split_act(g_conv, Wc, B, T, None, 0, SG, "gy",
                                                            colsum=(2 if partial else 0, lens, kt, d),
                                                            sum_out=grad_out(in_p[3 * j + 2]))
                x_t = transpose_split_act(H[j], Wc, B, T, lens, 1 if partial else 0, 1.0, "x", need_odd=(d % 2 == 1))
                slabs = wgrad_h3_slabs(gy_t, x_t, Wc, Wc, Wc, kt, d, 1.0 / SG, WPR)
                # reusable by layer j-1's res_skip gradient (even dilation: no advanced copy in the way).  The copy was made
                # with the length mask (partial padding), the res_skip gradient wants H[j] unmasked: the two are the same
                # tensor because H[j], j >= 1, is exactly zero at frames >= len (in_layer j-1's epilogue multiplies by the
                # mask, common.py:186-190) -- independent of the upstream gradient, so any loss is handled exactly
                x_prev = x_t if (d % 2 == 0 and j >= 1) else None
            else:
                x_prev = None
                g_in[3 * j + 2] = colsum(g_conv, Wc, 2 if partial else 0, T, lens, kt, d)
                slabs = wgrad_slabs(g_conv, Wc, H[j], Wc, Wc, T, lens, taps=kt, dil=d, x_mask_mode=1 if partial else 0)
            g_in[3 * j], g_in[3 * j + 1] = weightnorm_bwd(in_p[3 * j], in_p[3 * j + 1], inv_i[j], slabs, Wc, poison=poison)
            if keep_pair:
                # in_layer j's data gradient is deferred into layer j-1's fused launch: transposed weights into a tap
                # stack with one free slot for res_skip j-1's
                if pre is not None:
                    WTh, WTl = pre["stacks"][j]
                else:
                    WTh, WTl = _halves(kt + 1, Wc, Wc, like=z_in)
                    transpose_split(Wih[j], Wil[j], Wc, Wc, Wc, NPR, out=(WTh[:kt], WTl[:kt]))
                WT_prev = (WTh, WTl, kt, d)
                pair_h, pair_l = nh, nlo
                G = None
            else:
                WiTh, WiTl = pre[("in", j)] if pre else transpose_split(Wih[j], Wil[j], Wc, Wc, Wc, NPR)   # [taps][ci][co]
                if j == 0:
                    Gh, Gl = _halves(N, Wc, like=z_in)
                    Glo = lo16()
                # (dL/dH_0: consumed as its split pair by the start conv's gradients; fp32 only for an unfused bias sum)
                G = None if (pair_only and fuse_cs and j == 0) else _empty(N, Wc, like=z_in)
                g_start_b, cs = cs_args(start_b) if (fuse_cs and j == 0) else (None, {})
                gemm_cs(Ah=gch, Al=gcl, lda_h=Wc, Bh=WiTh, Bl=WiTl, ldb_h=Wc, b_tap_stride_h=WiTh.stride(0),
                        C=G, ldc=Wc, M=N, N=Wc, K=Wc, taps=kt, dil=d, sign=-1, lens=lens,
                        a_mask_mode=0, premask=1 if partial else 0, Ch=Gh if j == 0 else None, Cl=Gl if j == 0 else None,
                        Clo=Glo if j == 0 else None, ldch=Wc, ch_scale=SG, **gin, **gout, **cs)
                pair_h = pair_l = None
            check_saturation(box)
            if j == 2 and nl >= 3:
                if batch is not None:
                    batch.flush()                    # (the bias gradients announced below must be final)
                # the gradients of the end conv and of WN layers >= 2 are final (their kernels are queued): a gradient
                # reducer may start their bucket's all-reduce now (rad_mmm_amd/ddp.py, bucket '.hi')
                notify_grads_final([end_w, end_b] + [t for jj in range(2, nl) for t in (*in_p[3 * jj: 3 * jj + 3], *res_p[3 * jj: 3 * jj + 3])])
        if batch is not None:
            batch.flush()
        perm = (h, D, 0)
        if use_rm:
            if not fuse_cs:
                g_start_b = colsum(G, Wc, out=grad_out(start_b))
            slabs = wg_rm((Gh, Glo if Glo is not None else Gl), X0pair, Wc, Kp, 1, 1)
        else:
            gy_t, g_start_b = transpose_split_act(G, Wc, B, T, None, 0, SG, "gy", colsum=(0, None, 1, 1),
                                                  sum_out=grad_out(start_b))
            x_t = transpose_split_act(X0, Kp, B, T, None, 0, 1.0, "x0")
            slabs = wgrad_h3_slabs(gy_t, x_t, Wc, Kp, Kp, 1, 1, 1.0 / SG, WPR)
        g_start_v, g_start_g = weightnorm_bwd(start_v, start_g, inv_s, slabs, Kp, perm, poison=poison)
        WsTh, WsTl = pre["start"] if pre else transpose_split(Wsh, Wsl, Wc, Kp, Wc, NPR)      # [1][Kp][Wc]
        gX0 = _empty(N, Kp, like=z_in)
        rowgemm_h3(Ah=Gh, Al=Gl, lda_h=Wc, Bh=WsTh, Bl=WsTl, ldb_h=Wc, C=gX0, ldc=Kp, M=N, N=Kp, K=Wc, **gin)
        # the context gradient: into the decoder's shared buffer (meta["ctx_acc"], one per forward pass; every affine flow
        # step reads the same context), in place; the step that runs last returns the buffer, the others return None
        acc = meta.get("ctx_acc")
        if acc is not None and ctx.needs_input_grad[2]:
            g_cond = ctx_acc_add(acc, meta.get("ctx_slot", 0), lambda: _empty(N, D, like=z_in),
                                 lambda dst, accum: check(lib.radmmm_wn_input_bwd(ptr(gX0), Kp, ptr(dst), D, 1 if accum else 0, ptr(gz1),
                                                                                  ZLD, N, D, h, stream()), "wn_input_bwd"))
        else:
            g_cond = _empty(N, D, like=z_in)
            check(lib.radmmm_wn_input_bwd(ptr(gX0), Kp, ptr(g_cond), D, 0, ptr(gz1), ZLD, N, D, h, stream()), "wn_input_bwd")
        g_b_eff = colsum(gz1, ZLD) if ctx.needs_input_grad[5] else None   # LUS conv: constant zero bias
        if OUTpair is not None and T % 16 != 0:
            # the channel mix's weight gradient (gz1^T z_in, 160 x 160 over all frames): the fp32-MFMA fast path needs K steps
            # of 16 frames inside one utterance (T % 16 == 0); where it does not apply (configs[4]: T' = 1000, 155 us on the
            # generic kernel) two 8-bit split passes + a one-tile launch of the row-major split kernel replace it.  At
            # T % 16 == 0 the same move was measured at 42.34 / 42.35 against 42.31 / 42.32 ms per step: nothing, not taken.
            gzh, gzl = split_f16(gz1, ZLD, SG, ZLD, NPR, GE, flag)
            zh, zl = split_f16(z_in, ZLD, 1.0, ZLD, NPR, X8_ACT_EXP)
            g_W_eff = wg_rm((gzh, gzl), (zh, zl), ZLD, ZLD, 1, 1).sum(0).view(ZLD, ZLD)
        else:
            g_W_eff = wgrad_slabs(gz1, ZLD, z_in, ZLD, ZLD, T, None).sum(0).view(ZLD, ZLD)
        g_zin = _empty(N, ZLD, like=z_in)
        rowgemm(A=gz1, lda=ZLD, B=W_eff, ldb=ZLD, b_layout=1, C=g_zin, ldc=ZLD, M=N, N=ZLD, K=ZLD, T=T)
        return (None, g_zin, g_cond, None, g_W_eff, g_b_eff, g_start_v, g_start_g, g_start_b, g_end_w, g_end_b,
                *g_in, *g_res)


class ConvNormH3Fn(torch.autograd.Function):
    """ConvNormFn on the split-operand GEMM path: same contract, Cin % 32 == 0.  The input is
    split on the fly, the weight gradient runs on transposed zero-gapped split copies, the bias gradient
    comes out of the transposing pass; Cout is padded to a multiple of 32 for the data gradient's K."""

    @staticmethod
    @amp_fwd
    def forward(ctx, meta, x, v, g, bias, lens):
        B, T, dil = meta["B"], meta["T"], meta["dil"]
        partial, mask_out, act = meta["partial"], meta["mask_out"], meta["act"]
        NPR = meta.get("nprod", 3)
        flag = sat_flag_of(meta["scale_box"])
        Cout, Cin, taps = v.shape
        N = B * T
        assert x.shape[0] == N and x.shape[1] >= Cin and x.is_contiguous()
        # K extent of the forward GEMM: a 1x1 conv whose input width is an odd multiple of 32 (the FiLM blocks' cond conv:
        # 1056 = 33 x 32) gets one zero K step more, so that it qualifies for the one-tap kernel, which walks K in pairs of
        # steps (split_f16 / split_weight zero-fill the pad columns; +3 % work for the faster loop)
        Kx = round_up(Cin, 64) if (taps == 1 and Cin % 64) else Cin
        ldy = round_up(Cout, 4)
        # weight gradient on the FP8-cross kernel (conv_norm's wgrad8): ONE split pass writes the input as hi + 8-bit cross array
        # (the weight gradient's pair) AND the fp16 lo part (the three-product GEMM's second operand)
        wg8 = bool(meta.get("wgrad8") and NPR == 3 and T >= 32 and B <= 1024 and Kx % 32 == 0 and Cout % 4 == 0 and ldy % 4 == 0 and
                   debug_env("RADMMM_WGRAD_RM", "1") != "0" and debug_env("RADMMM_DACT_ROWS", "1") != "0")
        if wg8:
            xl = torch.empty(N, Kx, device=x.device, dtype=torch.float16)
            xh, xx = split_f16(x, Cin, 1.0, Kx, 2, X8_ACT_EXP, flag, lo16=xl)
        else:
            xh, xl = split_f16(x, Cin, 1.0, Kx, NPR, X8_ACT_EXP, flag)
        Wh, Wl, inv = split_weight(v, g, Kx, nprod=NPR)
        y = torch.zeros(N, ldy, device=x.device, dtype=torch.float32) if ldy != Cout else _empty(N, ldy, like=x)
        rowgemm_h3(nprod=NPR, a8_exp=X8_ACT_EXP, b8_exp=X8_W_EXP, Ah=xh, Al=xl, lda_h=Kx, Bh=Wh, Bl=Wl, ldb_h=Kx,
                   b_tap_stride_h=Wh.stride(0), acc_scale=1.0 / W_SCALE,
                   C=y, ldc=ldy, M=N, N=Cout, K=Kx, taps=taps, dil=dil, sign=1, T=T, lens=lens,
                   a_mask_mode=1 if partial else 0, bias=bias, pconv=1 if partial else 0, ratio_taps=taps, ratio_dil=dil,
                   postmask=1 if mask_out else 0, act=act)
        ctx.meta = meta
        ctx.has_g, ctx.has_bias, ctx.has_lens = g is not None, bias is not None, lens is not None
        # the row-major split pair of x is the weight gradient's operand (radmmm_wgrad_rm) when it is an fp16 pair
        ctx.has_xpair = bool(wg8 or (NPR in (2, 3) and T >= 32 and B <= 1024 and Cin % (32 if NPR == 2 else 8) == 0 and
                                     debug_env("RADMMM_WGRAD_RM", "1") != "0"))
        ctx.wg8 = wg8
        ctx.save_for_backward(x, v, g if g is not None else v, lens if lens is not None else v, Wh, Wl,
                              inv if inv is not None else v, y, *(((xh, xx) if wg8 else (xh, xl)) if ctx.has_xpair else ()))
        return y

    @staticmethod
    @amp_bwd
    def backward(ctx, gy):
        meta = ctx.meta
        B, T, dil = meta["B"], meta["T"], meta["dil"]
        partial, mask_out, act = meta["partial"], meta["mask_out"], meta["act"]
        x, v, g, lens, Wh, Wl, inv, y = ctx.saved_tensors[:8]
        xpair = tuple(ctx.saved_tensors[8:10]) if ctx.has_xpair else None
        lens = lens if ctx.has_lens else None
        NPR = meta.get("nprod", 3)
        WPR = 3 if NPR == 2 else NPR
        Cout, Cin, taps = v.shape
        N = B * T
        gy = gy.contiguous()
        ldy = y.shape[1]
        box = meta["scale_box"]
        SG = grad_scale(box, gy)
        GE = grad_x8_exp(box)
        flag = sat_flag_bwd_of(box)
        Kp = round_up(Cout, 32)
        rowscale = 2 if partial else (1 if mask_out else 0)
        gph, gpl = _halves(N, Kp, like=y, zero=(Kp != Cout))       # K padding of the data gradient must read as zeros
        # with the row-major weight gradient nothing reads an fp32 copy of the pre-activation gradient: it is written as the
        # split pair only and the bias sums leave the same pass as partials (radmmm_dact_mul_rows: 12 instead of 20 bytes per
        # element over two passes; 32 000-row FiLM convs: dact_mul 77 + colsum 27 us -> one launch)
        fused_rows = (xpair is not None and Cout % 4 == 0 and ldy % 4 == 0 and N == B * T and
                      debug_env("RADMMM_DACT_ROWS", "1") != "0")
        wg8 = bool(getattr(ctx, "wg8", False) and fused_rows)
        gpx = None
        if fused_rows:
            gpre = None
            nparts = B * (-(-T // 64))
            part = _empty(nparts, Cout, like=y)
            if wg8:
                # (hi, 8-bit cross array) for the weight gradient + the fp16 lo part for the three-product data gradient, one pass
                gpx = gpl
                gpl = torch.zeros(N, Kp, device=y.device, dtype=torch.float16) if Kp != Cout else torch.empty(N, Kp, device=y.device,
                                                                                                           dtype=torch.float16)
                so = split_opts(SPLIT_X8A, GE, flag, gpl)
            else:
                so = split_opts(fmt_a(NPR), GE, flag)
            check(lib.radmmm_dact_mul_rows(ptr(gy), ldy, ptr(y), ldy, Cout, B, T, act, rowscale, ptr(lens), taps, dil, SG,
                                           ptr(gph), ptr(gpx if wg8 else gpl), Kp, so, ptr(part), stream()),
                  "dact_mul_rows")
            g_bias = _empty(Cout, like=y)
            check(lib.radmmm_colsum_final(ptr(part), ptr(g_bias), nparts, Cout, stream()), "colsum_final")
        else:
            gpre = torch.zeros_like(y) if ldy != Cout else torch.empty_like(y)
            check(lib.radmmm_dact_mul(ptr(gy), ldy, ptr(y), ldy, ptr(gpre), ldy, N, Cout, act, rowscale, T, ptr(lens),
                                      taps, dil, ptr(gph), ptr(gpl), Kp, SG, split_opts(fmt_a(NPR), GE, flag), stream()),
                  "dact_mul")
        if xpair is not None:
            if not fused_rows:
                g_bias = colsum(gpre, Cout, 2 if partial else 0, T, lens, taps, dil)
            if NPR == 2 or wg8:    # (hi, 8-bit cross array) pairs: the FP8-cross weight gradient (radmmm_wgrad_rm8)
                slabs = wgrad_rm8_slabs((gph, gpx if wg8 else gpl), GE, xpair, X8_ACT_EXP, B, T, Cout, Cin, taps, dil, 1.0 / SG,
                                        lens if partial else None)
            else:
                slabs = wgrad_rm_slabs((gph, gpl), xpair, B, T, Cout, Cin, taps, dil, 1.0 / SG, lens if partial else None)
        else:
            gy_t, g_bias = transpose_split_act(gpre, Cout, B, T, None, 0, SG, "gy",
                                               colsum=(2 if partial else 0, lens, taps, dil))
            x_t = transpose_split_act(x, Cin, B, T, lens, 1 if partial else 0, 1.0, "x", need_odd=(dil % 2 == 1 and taps > 1))
            slabs = wgrad_h3_slabs(gy_t, x_t, Cout, Cin, Cin, taps, dil, 1.0 / SG, WPR)
        if ctx.has_g:
            g_v, g_g = weightnorm_bwd(v, g, inv, slabs, Cin, poison=poison_of(box))
        else:
            g_v, g_g = slabs.sum(0).permute(1, 2, 0).contiguous(), None
        gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.zeros_like(x) if x.shape[1] != Cin else torch.empty_like(x)
            WTh, WTl = transpose_split(Wh, Wl, Cout, Cin, Kp, NPR)                   # [taps][Cin][Kp]
            rowgemm_h3(nprod=NPR, a8_exp=GE, b8_exp=X8_W_EXP, Ah=gph, Al=gpl, lda_h=Kp, Bh=WTh, Bl=WTl, ldb_h=Kp,
                       b_tap_stride_h=WTh.stride(0),
                       acc_scale=1.0 / (SG * W_SCALE), C=gx, ldc=x.shape[1], M=N, N=Cin, K=Kp, taps=taps, dil=dil, sign=-1,
                       T=T, lens=lens, a_mask_mode=0, premask=1 if partial else 0)
        check_saturation(box)
        return None, gx, g_v, g_g, g_bias if ctx.has_bias else None, None
