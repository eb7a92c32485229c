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

from ._lib import lib, check, ptr, stream, amp_fwd, amp_bwd, rowgemm_h3, debug_env
from ._trace import traced


def _scratch(B, H, which, like):
    n = int(lib.radmmm_lstm_scratch_bytes(B, H, which))
    return torch.empty((n + 3) // 4, device=like.device, dtype=torch.float32)


class BiLSTMFn(torch.autograd.Function):
    """y = BiLSTM(x) for x [B, T, I] (batch first), lens int32 [B] on the device or None.
    backward consumes the saved gate activations in place (not re-entrant: no retain_graph)."""

    @staticmethod
    @amp_fwd
    def forward(ctx, x, lens, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, box=None):
        B, T, I = x.shape
        ctx.box = box
        H = w_hh_f.shape[1]
        x2 = x.reshape(B * T, I).contiguous()
        W_ih = torch.cat((w_ih_f, w_ih_r), 0)                    # [8H, I]
        bias = torch.cat((b_ih_f + b_hh_f, b_ih_r + b_hh_r))     # [8H]
        if (8 * H) % 4 == 0 and B * T >= 4096 and debug_env("RADMMM_LSTM_PROJ", "hip") != "torch":
            # frame-rate inputs: the input projection on the split-f16 row GEMM (three f16 products, 2e-6): 114 GFLOP at
            # the benchmark size in ~0.4 ms instead of 0.91 ms on the fp32 library GEMM; W_ih [8H, I] is already the
            # K-contiguous B operand.  (Its gradient GEMMs: BiLSTMFn.backward.)
            from . import ops
            Kp = ops.round_up(I, 32)
            xh, xl = ops.split_f16(x2, I, 1.0, Kp)
            Wh, Wl = ops.split_f16(W_ih, I, ops.W_SCALE, Kp)
            G = torch.empty(B * T, 8 * H, device=x.device, dtype=torch.float32)
            rowgemm_h3(nprod=3, Ah=xh, Al=xl, lda_h=Kp, Bh=Wh, Bl=Wl, ldb_h=Kp, acc_scale=1.0 / ops.W_SCALE, C=G, ldc=8 * H,
                       M=B * T, N=8 * H, K=Kp, T=T, bias=bias)
            xpair = (xh, xl)                                     # the row-major split pair of x: dW_ih's operand in backward
        else:
            xpair = ()
            G = torch.addmm(bias, x2, W_ih.t())                  # [B*T, 8H]
        W_hh = torch.stack((w_hh_f, w_hh_r)).contiguous()        # [2, 4H, H]
        y = torch.empty(B * T, 2 * H, device=x.device, dtype=torch.float32)
        c = torch.empty(B * T, 2 * H, device=x.device, dtype=torch.float32)
        wsplit = _scratch(B, H, 0, x)
        hsplit = _scratch(B, H, 1, x)
        nq = int(lib.radmmm_lstm_hseq_bytes(B, T, H))            # > 0: all T steps in one launch, one operand slot per step
        hseq = torch.empty(nq // 4, device=x.device, dtype=torch.float32) if nq else None
        check(lib.radmmm_lstm_fwd(ptr(G), ptr(W_hh), ptr(y), ptr(c), ptr(lens), ptr(wsplit), ptr(hsplit), ptr(hseq), B, T, H,
                                  stream()), "lstm_fwd")
        ctx.dims = (B, T, I, H)
        ctx.save_for_backward(x2, G, c, y, W_ih, W_hh, lens if lens is not None else torch.empty(0, device=x.device), *xpair)
        ctx.has_lens = lens is not None
        ctx.has_xpair = len(xpair) == 2
        return y.view(B, T, 2 * H)

    @staticmethod
    @amp_bwd
    @traced("lstm.bwd")
    def backward(ctx, dy):
        B, T, I, H = ctx.dims
        if getattr(ctx, "_consumed", False):
            raise RuntimeError("BiLSTMFn.backward ran twice on the same graph: it turns the saved gate activations into "
                               "gradients in place (no retain_graph / double backward)")
        ctx._consumed = True
        x2, G, c, y, W_ih, W_hh, lens = ctx.saved_tensors[:7]
        xpair = tuple(ctx.saved_tensors[7:9]) if ctx.has_xpair else None
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
        y3 = y.view(B, T, 2 * H)
        hp = torch.zeros(B, T, 2 * H, device=dy.device, dtype=torch.float32)
        hp[:, 1:, :H] = y3[:, :-1, :H]                           # forward direction: h_{t-1}
        hp[:, :-1, H:] = y3[:, 1:, H:]                           # reverse direction: h_{t+1}
        hp = hp.view(B * T, 2 * H)
        if (ctx.box is not None and B * T >= 4096 and (8 * H) % 32 == 0 and I % 4 == 0 and H % 2 == 0 and
                debug_env("RADMMM_LSTM_GRADS", "hip") != "torch"):
            # frame-rate batches: the four gradient GEMMs (226 + 57 GFLOP at the benchmark size, 2.5 ms on the fp32 library
            # GEMMs) on the split-f16 kernels: ONE transposing pass over dG feeds both weight gradients (contraction over
            # frames) and yields the bias gradient as its column sums; the input gradient is a row GEMM on dG's split copy.
            # Scale of the split gradient tensors: the module's GradScale (previous pass's amax, no host sync).
            from . import ops
            box = ctx.box
            SG = ops.grad_scale(box, dy2)
            flag = ops.sat_flag_bwd_of(box)
            gh, gl = ops.split_f16(dG, 8 * H, SG, 8 * H, 3, 0, flag)       # row-major split pair of dG: operand of all four GEMMs
            if (xpair is not None and T >= 32 and B <= 1024 and (4 * H) % 8 == 0 and
                    debug_env("RADMMM_WGRAD_RM", "1") != "0"):
                # weight gradients straight from the row-major pairs (radmmm_wgrad_rm: transposition in the LDS read); the
                # pair of x was made for the forward projection, the pairs of h_prev (one per direction, each with its own
                # 16-byte aligned row pitch) cost what the transposed copy did
                db = ops.colsum(dG, 8 * H)
                Hq = ops.round_up(H, 8)
                hpf = ops.split_f16(hp[:, :H].contiguous(), H, 1.0, Hq)
                hpr = ops.split_f16(hp[:, H:].contiguous(), H, 1.0, Hq)
                dW_ih = ops.wgrad_rm_slabs((gh, gl), xpair, B, T, 8 * H, I, 1, 1, 1.0 / SG).sum(0)[0]
                dW_hh_f = ops.wgrad_rm_slabs((gh[:, :4 * H], gl[:, :4 * H]), hpf, B, T, 4 * H, H, 1, 1, 1.0 / SG).sum(0)[0]
                dW_hh_r = ops.wgrad_rm_slabs((gh[:, 4 * H:], gl[:, 4 * H:]), hpr, B, T, 4 * H, H, 1, 1, 1.0 / SG).sum(0)[0]
            else:
                gy_t, db = ops.transpose_split_act(dG, 8 * H, B, T, None, 0, SG, "lstm_gy", colsum=(0, None, 1, 1))
                x_t = ops.transpose_split_act(x2, I, B, T, None, 0, 1.0, "lstm_x")
                hp_t = ops.transpose_split_act(hp, 2 * H, B, T, None, 0, 1.0, "lstm_h")
                dW_ih = ops.wgrad_h3_slabs(gy_t, x_t, 8 * H, I, I, 1, 1, 1.0 / SG, 3).sum(0)[0]
                rows = lambda tt, a, b: (tt[0][a:b], tt[1][a:b], None, None, tt[4])
                dW_hh_f = ops.wgrad_h3_slabs(rows(gy_t, 0, 4 * H), rows(hp_t, 0, H), 4 * H, H, H, 1, 1, 1.0 / SG, 3).sum(0)[0]
                dW_hh_r = ops.wgrad_h3_slabs(rows(gy_t, 4 * H, 8 * H), rows(hp_t, H, 2 * H), 4 * H, H, H, 1, 1, 1.0 / SG, 3).sum(0)[0]
            dx = None
            if ctx.needs_input_grad[0]:
                Wt = W_ih.t().contiguous()                       # [I, 8H]: the K-contiguous operand of dx = dG W_ih
                Wth, Wtl = ops.split_f16(Wt, 8 * H, ops.W_SCALE, 8 * H)
                dx = torch.empty(B * T, I, device=dy.device, dtype=torch.float32)
                rowgemm_h3(nprod=3, Ah=gh, Al=gl, lda_h=8 * H, Bh=Wth, Bl=Wtl, ldb_h=8 * H, acc_scale=1.0 / (SG * ops.W_SCALE),
                           C=dx, ldc=I, M=B * T, N=I, K=8 * H, T=T)
                dx = dx.view(B, T, I)
            ops.check_saturation(box)
        else:
            dx = (dG @ W_ih).view(B, T, I) if ctx.needs_input_grad[0] else None
            dW_ih = dG.t() @ x2                                  # [8H, I]
            db = dG.sum(0)
            dW_hh_f = dG[:, :4 * H].t() @ hp[:, :H]
            dW_hh_r = dG[:, 4 * H:].t() @ hp[:, H:]
        return (dx, None, dW_ih[:4 * H], dW_hh_f, db[:4 * H], db[:4 * H], dW_ih[4 * H:], dW_hh_r, db[4 * H:], db[4 * H:],
                None)


def bilstm(lstm: torch.nn.LSTM, x: torch.Tensor, lens32) -> torch.Tensor:
    """Apply `lstm`'s parameters (single layer, bidirectional, batch_first) with the HIP recurrence."""
    assert lstm.num_layers == 1 and lstm.bidirectional and lstm.batch_first and lstm.proj_size == 0
    box = None
    if x.is_cuda and torch.is_grad_enabled():
        # scale state of the split gradient tensors (ops.GradScale): one per LSTM module, carried from pass to pass
        from . import ops
        box = lstm.__dict__.get("_radmmm_grad_scale")
        if box is None:
            box = ops.GradScale()
            lstm.__dict__["_radmmm_grad_scale"] = box
        box.new_forward(x.device)
    return BiLSTMFn.apply(x, lens32, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                          lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                          lstm.bias_hh_l0_reverse, box)


class MergedBiLSTMFn(torch.autograd.Function):
    """P independent bidirectional LSTMs of the SAME shape (hidden H, same batch, frames and lengths) as ONE recurrence.

    An LSTM's gates are elementwise per hidden unit, so P LSTMs side by side are one LSTM with hidden P*H whose recurrent
    matrix is BLOCK-DIAGONAL.  The recurrence is latency-bound (T dependent steps of a few microseconds, whatever H is up to
    the kernel's size classes), so the merged launch costs about what ONE of the P costs: the f0 / energy / voiced predictors
    of the joint step (BASELINE configs[3]: three ConvLSTMLinearDAP with a 256 -> 2 x 128 bi-LSTM over the same 800 frames) ran
    3 x (3.5 + 4) ms of recurrences one after the other.  Nothing changes in csrc/lstm.hip: the merged gate pre-activations
    G' [B*T, 2, 4, P, H], the block-diagonal W_hh' [2, 4 P H, P H] and dy' are assembled here, and the input projections /
    weight gradients stay per LSTM (small GEMMs).  The zero blocks cost MFMA work nobody waits for.

    forward(lens, P, box, x_0 .. x_{P-1}, then per LSTM: w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
    -> y_0 .. y_{P-1}, each [B, T, 2H]."""

    @staticmethod
    @amp_fwd
    def forward(ctx, lens, P, box, *args):
        xs, ws = args[:P], args[P:]
        ctx.box = box
        B, T, I = xs[0].shape
        H = ws[1].shape[1]
        HP = P * H
        dev = xs[0].device
        W_hh = torch.zeros(2, 4, P, H, P, H, device=dev, dtype=torch.float32)        # [dir][gate][p][unit] x [p'][unit']
        x2s, W_ihs = [], []
        # Round 6: frame-rate batches run the P input projections as ONE split-f16 GEMM on the concatenated inputs with a
        # block-structured weight (row (dir, gate, p, unit) holds W_ih_p's row in columns [p I, (p + 1) I), zeros elsewhere): the
        # result IS the merged gate layout, and in backward the same pair of operands gives dW_ih / dx / dW_hh on the row-major
        # split kernels (three f16 products, 2e-6) -- instead of 3 x {fp32 library GEMM at 40 TFLOP/s + a strided scatter of 105 MB}
        # each way (joint step: 2.96 ms of `Cijk_*` launches per step, profiles/r05_joint_kernel_stats.txt).  The zero blocks
        # cost 2/3 of 121 GFLOP of MFMA work nobody waits for.  RADMMM_MERGED_LSTM_GEMMS=torch (RADMMM_DEBUG): the library path.
        I = xs[0].shape[2]
        split = bool(box is not None and B * T >= 4096 and all(x.shape[2] == I for x in xs) and (P * I) % 32 == 0 and H % 8 == 0
                     and T >= 32 and B <= 1024 and debug_env("RADMMM_MERGED_LSTM_GEMMS", "hip") != "torch")
        ctx.split = split
        if split:
            from . import ops
            Kc = P * I
            x_cat = torch.cat([x.reshape(B * T, I) for x in xs], 1)                # [B*T, P*I]
            Wb = torch.zeros(2, 4, P, H, P, I, device=dev, dtype=torch.float32)
            bias = torch.empty(2, 4, P, H, device=dev, dtype=torch.float32)
        else:
            G = torch.empty(B * T, 2, 4, P, H, device=dev, dtype=torch.float32)
        for p in range(P):
            w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r = ws[8 * p: 8 * p + 8]
            if split:
                Wb[0, :, p, :, p, :] = w_ih_f.view(4, H, I)
                Wb[1, :, p, :, p, :] = w_ih_r.view(4, H, I)
                bias[0, :, p, :] = (b_ih_f + b_hh_f).view(4, H)
                bias[1, :, p, :] = (b_ih_r + b_hh_r).view(4, H)
            else:
                x2 = xs[p].reshape(B * T, -1).contiguous()
                W_ih = torch.cat((w_ih_f, w_ih_r), 0)                               # [8H, I_p]
                bias_p = torch.cat((b_ih_f + b_hh_f, b_ih_r + b_hh_r))
                G[:, :, :, p, :] = torch.addmm(bias_p, x2, W_ih.t()).view(B * T, 2, 4, H)
                x2s.append(x2)
                W_ihs.append(W_ih)
            W_hh[0, :, p, :, p, :] = w_hh_f.view(4, H, H)
            W_hh[1, :, p, :, p, :] = w_hh_r.view(4, H, H)
        if split:
            Wb = Wb.view(8 * HP, Kc)
            xh, xl = ops.split_f16(x_cat, Kc, 1.0, Kc)
            Wh, Wl = ops.split_f16(Wb, Kc, ops.W_SCALE, Kc)
            G = torch.empty(B * T, 8 * HP, device=dev, dtype=torch.float32)
            rowgemm_h3(nprod=3, Ah=xh, Al=xl, lda_h=Kc, Bh=Wh, Bl=Wl, ldb_h=Kc, acc_scale=1.0 / ops.W_SCALE, C=G, ldc=8 * HP,
                       M=B * T, N=8 * HP, K=Kc, T=T, bias=bias.view(8 * HP))
            x2s, W_ihs = [xh, xl], [Wb]
        G = G.view(B * T, 8 * HP)
        W_hh = W_hh.view(2, 4 * HP, HP)
        y = torch.empty(B * T, 2 * HP, device=dev, dtype=torch.float32)
        c = torch.empty(B * T, 2 * HP, device=dev, dtype=torch.float32)
        wsplit, hsplit = _scratch(B, HP, 0, y), _scratch(B, HP, 1, y)
        nq = int(lib.radmmm_lstm_hseq_bytes(B, T, HP))
        hseq = torch.empty(nq // 4, device=dev, dtype=torch.float32) if nq else None
        check(lib.radmmm_lstm_fwd(ptr(G), ptr(W_hh), ptr(y), ptr(c), ptr(lens), ptr(wsplit), ptr(hsplit), ptr(hseq), B, T, HP,
                                  stream()), "lstm_fwd")
        ctx.dims = (B, T, H, P)
        ctx.has_lens = lens is not None
        ctx.n_x2 = len(x2s)
        ctx.save_for_backward(G, c, y, W_hh, lens if lens is not None else torch.empty(0, device=dev), *x2s, *W_ihs)
        y4 = y.view(B, T, 2, P, H)
        return tuple(y4[:, :, :, p, :].reshape(B, T, 2 * H) for p in range(P))

    @staticmethod
    @amp_bwd
    @traced("lstm_merged.bwd")
    def backward(ctx, *dys):
        B, T, H, P = ctx.dims
        HP = P * H
        if getattr(ctx, "_consumed", False):
            raise RuntimeError("MergedBiLSTMFn.backward ran twice on the same graph (it turns the saved gate activations into "
                               "gradients in place)")
        ctx._consumed = True
        G, c, y, W_hh, lens = ctx.saved_tensors[:5]
        x2s, W_ihs = ctx.saved_tensors[5: 5 + ctx.n_x2], ctx.saved_tensors[5 + ctx.n_x2:]
        lens = lens if ctx.has_lens else None
        dev = G.device
        dy = torch.zeros(B, T, 2, P, H, device=dev, dtype=torch.float32)
        for p in range(P):
            if dys[p] is not None:
                dy[:, :, :, p, :] = dys[p].reshape(B, T, 2, H)
        dy2 = dy.view(B * T, 2 * HP)
        amax = dy2.abs().amax().clamp_min(1e-30)
        gscale = torch.exp2(torch.floor(torch.log2(64.0 / amax))).reshape(1).float().contiguous()
        wtpack, Pbuf, dcbuf = _scratch(B, HP, 2, dy2), _scratch(B, HP, 3, dy2), _scratch(B, HP, 4, dy2)
        check(lib.radmmm_lstm_bwd(ptr(G), ptr(c), ptr(dy2), ptr(W_hh), ptr(lens), ptr(wtpack), ptr(Pbuf), ptr(dcbuf), B, T, HP,
                                  ptr(gscale), stream()), "lstm_bwd")
        grads = [None] * (3 + P + 8 * P)
        if ctx.split:
            # the merged tensors as they are: dG' [B*T, 8 HP] is the row-major operand of all four gradient GEMMs (BiLSTMFn.backward)
            from . import ops
            box = ctx.box
            (xh, xl), Wb = x2s, W_ihs[0]
            I = Wb.shape[1] // P
            Kc = P * I
            SG = ops.grad_scale(box, dy2)
            flag = ops.sat_flag_bwd_of(box)
            gh, gl = ops.split_f16(G, 8 * HP, SG, 8 * HP, 3, 0, flag)
            db = ops.colsum(G, 8 * HP).view(2, 4, P, H)
            dWb = ops.wgrad_rm_slabs((gh, gl), (xh, xl), B, T, 8 * HP, Kc, 1, 1, 1.0 / SG).sum(0)[0].view(2, 4, P, H, P, I)
            y4 = y.view(B, T, 2, HP)
            hp = torch.zeros(2, B, T, HP, device=dev, dtype=torch.float32)
            hp[0, :, 1:] = y4[:, :-1, 0]                                             # forward direction: h_{t-1}
            hp[1, :, :-1] = y4[:, 1:, 1]                                             # reverse direction: h_{t+1}
            dWhh = []
            for d in range(2):
                hpair = ops.split_f16(hp[d].view(B * T, HP), HP, 1.0, HP)
                dWhh.append(ops.wgrad_rm_slabs((gh[:, 4 * HP * d: 4 * HP * (d + 1)], gl[:, 4 * HP * d: 4 * HP * (d + 1)]), hpair, B, T,
                                               4 * HP, HP, 1, 1, 1.0 / SG).sum(0)[0].view(4, P, H, P, H))
            dx = None
            if any(ctx.needs_input_grad[3 + p] for p in range(P)):
                Wt = Wb.t().contiguous()                                             # [P I, 8 HP]: the K-contiguous operand of dx = dG' Wb
                Wth, Wtl = ops.split_f16(Wt, 8 * HP, ops.W_SCALE, 8 * HP)
                dx = torch.empty(B * T, Kc, device=dev, dtype=torch.float32)
                rowgemm_h3(nprod=3, Ah=gh, Al=gl, lda_h=8 * HP, Bh=Wth, Bl=Wtl, ldb_h=8 * HP, acc_scale=1.0 / (SG * ops.W_SCALE),
                           C=dx, ldc=Kc, M=B * T, N=Kc, K=8 * HP, T=T)
            for p in range(P):
                if dx is not None and ctx.needs_input_grad[3 + p]:
                    grads[3 + p] = dx[:, p * I: (p + 1) * I].reshape(B, T, I)
                dbp = db[:, :, p, :]
                base = 3 + P + 8 * p
                grads[base: base + 8] = [dWb[0, :, p, :, p, :].reshape(4 * H, I), dWhh[0][:, p, :, p, :].reshape(4 * H, H),
                                         dbp[0].reshape(4 * H), dbp[0].reshape(4 * H),
                                         dWb[1, :, p, :, p, :].reshape(4 * H, I), dWhh[1][:, p, :, p, :].reshape(4 * H, H),
                                         dbp[1].reshape(4 * H), dbp[1].reshape(4 * H)]
            ops.check_saturation(box)
            return tuple(grads)
        dG5 = G.view(B * T, 2, 4, P, H)                                              # now the pre-activation gradients
        y5 = y.view(B, T, 2, P, H)
        for p in range(P):
            dG = dG5[:, :, :, p, :].reshape(B * T, 8 * H)
            yp = y5[:, :, :, p, :]                                                   # [B, T, 2, H]
            hp = torch.zeros(B, T, 2, H, device=dev, dtype=torch.float32)
            hp[:, 1:, 0] = yp[:, :-1, 0]                                             # forward direction: h_{t-1}
            hp[:, :-1, 1] = yp[:, 1:, 1]                                             # reverse direction: h_{t+1}
            hp = hp.view(B * T, 2, H)
            dW_ih = dG.t() @ x2s[p]
            db = dG.sum(0)
            dW_hh_f = dG[:, : 4 * H].t() @ hp[:, 0]
            dW_hh_r = dG[:, 4 * H:].t() @ hp[:, 1]
            if ctx.needs_input_grad[3 + p]:
                grads[3 + p] = (dG @ W_ihs[p]).view(B, T, -1)
            base = 3 + P + 8 * p
            grads[base: base + 8] = [dW_ih[: 4 * H], dW_hh_f, db[: 4 * H], db[: 4 * H], dW_ih[4 * H:], dW_hh_r, db[4 * H:], db[4 * H:]]
        return tuple(grads)


def merged_bilstm(lstms, xs, lens32):
    """y_p = bilstm(lstms[p], xs[p], lens32) for LSTMs of one shape over the same frames, as ONE recurrence (MergedBiLSTMFn);
    the caller has materialised normed recurrent weights (spectral / weight norm hooks) already."""
    ws = []
    for l in lstms:
        assert l.num_layers == 1 and l.bidirectional and l.batch_first and l.proj_size == 0
        ws += [l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0, l.weight_ih_l0_reverse, l.weight_hh_l0_reverse,
               l.bias_ih_l0_reverse, l.bias_hh_l0_reverse]
    box = None
    if xs[0].is_cuda and torch.is_grad_enabled():
        # scale state of the merged recurrence's split gradient tensors (ops.GradScale, as `bilstm` keeps one per LSTM): kept on
        # the first module of the group, carried from pass to pass
        from . import ops
        box = lstms[0].__dict__.get("_radmmm_merged_grad_scale")
        if box is None:
            box = lstms[0].__dict__["_radmmm_merged_grad_scale"] = ops.GradScale()
        box.new_forward(xs[0].device)
    return MergedBiLSTMFn.apply(lens32, len(lstms), box, *[x.contiguous() for x in xs], *ws)


def can_merge(lstms, xs) -> bool:
    """same hidden size, batch and frames; the merged hidden size within the recurrence kernel's size classes (<= 768)"""
    if len(lstms) < 2:
        return False
    H = lstms[0].hidden_size
    return (all(l.hidden_size == H for l in lstms) and all(x.shape[:2] == xs[0].shape[:2] for x in xs) and
            len(lstms) * H <= 768 and xs[0].is_cuda)
