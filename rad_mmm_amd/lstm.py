"""Bidirectional context LSTM on libradmmm_hip.so (reference: models/radmmm.py:141-146, the
nn.LSTM(bidirectional=True, batch_first=True) applied to a packed batch in
RADMMM.preprocess_context).

The input projection x W_ih^T (+ biases) and the weight / input gradients are single large GEMMs;
the T' sequential recurrent steps run as one fused HIP launch per step for both directions
(csrc/lstm.hip).  Same parameters as torch.nn.LSTM (weight_ih_l0, weight_hh_l0, bias_ih_l0,
bias_hh_l0 and their _reverse twins), same output: y [B, T, 2H] with zeros at frames >= length."""
from __future__ import annotations

import os

import torch

from ._lib import lib, check, ptr, stream, amp_fwd, amp_bwd, rowgemm_h3


def _scratch(B, H, which, like):
    n = int(lib.radmmm_lstm_scratch_bytes(B, H, which))
    return torch.empty((n + 3) // 4, device=like.device, dtype=torch.float32)


class BiLSTMFn(torch.autograd.Function):
    """y = BiLSTM(x) for x [B, T, I] (batch first), lens int32 [B] on the device or None.
    backward consumes the saved gate activations in place (not re-entrant: no retain_graph)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, lens, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        B, T, I = x.shape
        H = w_hh_f.shape[1]
        x2 = x.reshape(B * T, I).contiguous()
        W_ih = torch.cat((w_ih_f, w_ih_r), 0)                    # [8H, I]
        bias = torch.cat((b_ih_f + b_hh_f, b_ih_r + b_hh_r))     # [8H]
        if (8 * H) % 4 == 0 and B * T >= 4096 and os.environ.get("RADMMM_LSTM_PROJ", "hip") != "torch":
            # frame-rate inputs: the input projection on the split-f16 row GEMM (three f16 products, 2e-6): 114 GFLOP at
            # the benchmark size in ~0.4 ms instead of 0.91 ms on the fp32 library GEMM; W_ih [8H, I] is already the
            # K-contiguous B operand.  (The projection's GRADIENT GEMMs stay on the library: moving them too was measured
            # twice and did not pay)
            from . import ops
            Kp = ops.round_up(I, 32)
            xh, xl = ops.split_f16(x2, I, 1.0, Kp)
            Wh, Wl = ops.split_f16(W_ih, I, ops.W_SCALE, Kp)
            G = torch.empty(B * T, 8 * H, device=x.device, dtype=torch.float32)
            rowgemm_h3(nprod=3, Ah=xh, Al=xl, lda_h=Kp, Bh=Wh, Bl=Wl, ldb_h=Kp, acc_scale=1.0 / ops.W_SCALE, C=G, ldc=8 * H,
                       M=B * T, N=8 * H, K=Kp, T=T, bias=bias)
        else:
            G = torch.addmm(bias, x2, W_ih.t())                  # [B*T, 8H]
        W_hh = torch.stack((w_hh_f, w_hh_r)).contiguous()        # [2, 4H, H]
        y = torch.empty(B * T, 2 * H, device=x.device, dtype=torch.float32)
        c = torch.empty(B * T, 2 * H, device=x.device, dtype=torch.float32)
        wsplit = _scratch(B, H, 0, x)
        hsplit = _scratch(B, H, 1, x)
        check(lib.radmmm_lstm_fwd(ptr(G), ptr(W_hh), ptr(y), ptr(c), ptr(lens), ptr(wsplit), ptr(hsplit), B, T, H,
                                  stream()), "lstm_fwd")
        ctx.dims = (B, T, I, H)
        ctx.save_for_backward(x2, G, c, y, W_ih, W_hh, lens if lens is not None else torch.empty(0, device=x.device))
        ctx.has_lens = lens is not None
        return y.view(B, T, 2 * H)

    @staticmethod
    @amp_bwd
    def backward(ctx, dy):
        B, T, I, H = ctx.dims
        if getattr(ctx, "_consumed", False):
            raise RuntimeError("BiLSTMFn.backward ran twice on the same graph: it turns the saved gate activations into "
                               "gradients in place (no retain_graph / double backward)")
        ctx._consumed = True
        x2, G, c, y, W_ih, W_hh, lens = ctx.saved_tensors
        lens = lens if ctx.has_lens else None
        dy2 = dy.contiguous().view(B * T, 2 * H)
        amax = dy2.abs().amax().clamp_min(1e-30)
        gscale = torch.exp2(torch.floor(torch.log2(64.0 / amax))).reshape(1).float().contiguous()
        wtpack = _scratch(B, H, 2, dy2)
        P = _scratch(B, H, 3, dy2)
        dcbuf = _scratch(B, H, 4, dy2)
        check(lib.radmmm_lstm_bwd(ptr(G), ptr(c), ptr(dy2), ptr(W_hh), ptr(lens), ptr(wtpack), ptr(P), ptr(dcbuf), B, T, H,
                                  ptr(gscale), stream()), "lstm_bwd")
        dG = G                                                   # now the pre-activation gradients
        dx = (dG @ W_ih).view(B, T, I) if ctx.needs_input_grad[0] else None
        dW_ih = dG.t() @ x2                                      # [8H, I]
        db = dG.sum(0)
        y3 = y.view(B, T, 2 * H)
        hp = torch.zeros(B, T, 2 * H, device=dy.device, dtype=torch.float32)
        hp[:, 1:, :H] = y3[:, :-1, :H]                           # forward direction: h_{t-1}
        hp[:, :-1, H:] = y3[:, 1:, H:]                           # reverse direction: h_{t+1}
        hp = hp.view(B * T, 2 * H)
        dW_hh_f = dG[:, :4 * H].t() @ hp[:, :H]
        dW_hh_r = dG[:, 4 * H:].t() @ hp[:, H:]
        return (dx, None, dW_ih[:4 * H], dW_hh_f, db[:4 * H], db[:4 * H], dW_ih[4 * H:], dW_hh_r, db[4 * H:], db[4 * H:])


def bilstm(lstm: torch.nn.LSTM, x: torch.Tensor, lens32) -> torch.Tensor:
    """Apply `lstm`'s parameters (single layer, bidirectional, batch_first) with the HIP recurrence."""
    assert lstm.num_layers == 1 and lstm.bidirectional and lstm.batch_first and lstm.proj_size == 0
    return BiLSTMFn.apply(x, lens32, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                          lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                          lstm.bias_hh_l0_reverse)
