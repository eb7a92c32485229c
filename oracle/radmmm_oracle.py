"""CPU oracle for the RAD-MMM flow-decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (functional, fp32 or fp64, torch-CPU/numpy) of the
arithmetic the reference performs on the path named by BASELINE.json.  It is the
checker for the HIP kernels; it is never the thing shipped or measured.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product package (`rad_mmm_amd`) must never import from here.

Pinning: the reference has no tests or golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, imported in the
build container by `tests/golden/make_golden.py`; the resulting fixtures live in
`tests/golden/*.npz` and `tests/test_oracle_golden.py` replays them.  One piece
is pinned one step removed: the Slaney mel filterbank (`mel_filterbank_slaney`), which
the reference takes from librosa 0.8.0 (absent here): its fixture comes from an
independent, librosa-validated implementation (Hugging Face transformers'
audio_utils.mel_filter_bank, tests/golden/make_mel_basis.py) -- see DESIGN.md.

All tensors use the reference's layout: activations [B, C, T], conv weights
[C_out, C_in, k].  Parameters are passed as a flat dict keyed by the reference's
state_dict names so that `ref_module.state_dict()` can be fed in directly.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# Synthetic-data / procedural-weight generators and the config mirror are NOT oracle
# arithmetic; they live in a neutral numpy-only module at the repo root (radmmm_synth.py)
# that neither imports the product package nor is part of it: bench.py uses them without
# importing the oracle, and the oracle / tests/golden/make_golden.py use them on a tree
# with no built libradmmm_hip.so.  Re-exported here for the tests' convenience.
import os as _os
import sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
from radmmm_synth import (DecoderConfig, decoder_state_shapes,  # noqa: E402,F401
                          procedural_decoder_state, procedural_tensor, synthetic_batch)

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# masks / squeeze
# --------------------------------------------------------------------------
def lengths_to_mask(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """bool [B, max_len], True where t < len_b.  common.py:105-116."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(max_len, device=lengths.device)
    return ids[None, :] < lengths[:, None]


def squeeze_time(x: Tensor, g: int) -> Tensor:
    """[B,C,T] -> [B,C*g,T//g] with out[b,c*g+k,l] = x[b,c,l*g+k].

    Equals nn.Unfold(kernel=(g,1), stride=g) on x[..., None]
    (decoders.py:118-122,178; models/radmmm.py:114-120).  Trailing T % g frames
    are dropped.
    """
    if g == 1:
        return x
    B, C, T = x.shape
    Tg = T // g
    return x[:, :, :Tg * g].reshape(B, C, Tg, g).permute(0, 1, 3, 2).reshape(B, C * g, Tg)


def unsqueeze_time(x: Tensor, g: int) -> Tensor:
    """Inverse of squeeze_time (decoders.py:150-160, `fold`)."""
    if g == 1:
        return x
    B, Cg, Tg = x.shape
    C = Cg // g
    return x.reshape(B, C, g, Tg).permute(0, 1, 3, 2).reshape(B, C, Tg * g)


# --------------------------------------------------------------------------
# weight norm, partial conv, WN
# --------------------------------------------------------------------------
def weight_norm_fold(v: Tensor, g: Tensor) -> Tensor:
    """w = g * v / ||v||_2, norm over all dims but 0 (torch weight_norm dim=0;
    common.py:173-174,791,813)."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _wn_weight(p: Params, prefix: str) -> Tensor:
    if prefix + "weight_v" in p:
        return weight_norm_fold(p[prefix + "weight_v"], p[prefix + "weight_g"])
    return p[prefix + "weight"]


def partial_conv1d(x: Tensor, mask: Optional[Tensor], w: Tensor, b: Optional[Tensor],
                   dilation: int) -> Tensor:
    """Length-mask-aware conv re-normalisation.  partialconv1d.py:58-94.

    mask: float [B,1,T] or None (None -> all-ones window count, which still
    renormalises the zero-padded borders).
    """
    k = w.shape[-1]
    pad = dilation * (k - 1) // 2
    B, _, T = x.shape
    ones_k = torch.ones(1, 1, k, dtype=x.dtype)
    m = mask if mask is not None else torch.ones(1, 1, T, dtype=x.dtype)
    cnt = F.conv1d(m, ones_k, padding=pad, dilation=dilation)
    ratio = k / (cnt + 1e-6)
    upd = cnt.clamp(0, 1)
    ratio = ratio * upd
    raw = F.conv1d(x * mask if mask is not None else x, w, b, padding=pad, dilation=dilation)
    if b is not None:
        bv = b.view(1, -1, 1)
        return ((raw - bv) * ratio + bv) * upd
    return raw * ratio


def conv_norm(p: Params, prefix: str, x: Tensor, mask: Optional[Tensor], dilation: int = 1,
              partial: bool = True) -> Tensor:
    """ConvNorm.forward without batch-norm.  common.py:179-191."""
    w = _wn_weight(p, prefix + "conv.")
    b = p.get(prefix + "conv.bias")
    if partial:
        y = partial_conv1d(x, mask, w, b, dilation)
    else:
        k = w.shape[-1]
        y = F.conv1d(x, w, b, padding=dilation * (k - 1) // 2, dilation=dilation)
    if mask is not None:
        y = y * mask
    return y


def wn_forward(p: Params, prefix: str, z0: Tensor, ctx: Tensor, mask: Optional[Tensor],
               n_layers: int, activation: str = "softplus", partial: bool = True,
               kinks: Optional[Dict] = None) -> Tensor:
    """WN.forward.  common.py:816-835 (no gated tanh, no residual path).
    Test accounting of relu's kink (not reference behaviour; affine_activation='relu' only): kinks["record"] receives the
    pre-activations under (prefix, layer, "in" | "res"); kinks["gates"] (same keys, bool) overrides the decision pre > 0 of the
    activation and of its derivative, as dap_forward's `gates` does."""
    base = F.softplus if activation == "softplus" else torch.relu
    rec = (kinks or {}).get("record")
    gts = (kinks or {}).get("gates")

    def act(pre, key):
        if rec is not None:
            rec[key] = pre.detach()
        if gts is not None and activation != "softplus":
            return torch.where(gts[key], pre, torch.zeros_like(pre))
        return base(pre)
    h = F.conv1d(torch.cat((z0, ctx), 1), _wn_weight(p, prefix + "start."), p[prefix + "start.bias"])
    out = torch.zeros_like(h)
    for i in range(n_layers):
        h = act(conv_norm(p, f"{prefix}in_layers.{i}.", h, mask, 2 ** i, partial), (prefix, i, "in"))
        r = act(F.conv1d(h, _wn_weight(p, f"{prefix}res_skip_layers.{i}."),
                         p[f"{prefix}res_skip_layers.{i}.bias"]), (prefix, i, "res"))
        out = out + r
    return F.conv1d(out, p[prefix + "end.weight"], p[prefix + "end.bias"])


def fused_add_tanh_sigmoid_multiply(a: Tensor, b: Tensor, n: int) -> Tensor:
    """common.py:66-73 (only WaveNetOriginal uses it; no config instantiates it)."""
    x = a + b
    return torch.tanh(x[:, :n]) * torch.sigmoid(x[:, n:])


# --------------------------------------------------------------------------
# coupling layers
# --------------------------------------------------------------------------
def scaling_and_log(su: Tensor, fn: str) -> Tuple[Tensor, Tensor]:
    """AffineTransformationLayer.get_scaling_and_logs.  common.py:1127-1140."""
    if fn == "tanh":
        s = torch.tanh(su) + 1 + 1e-6
        return s, torch.log(s)
    if fn == "exp":
        return torch.exp(su), su
    if fn == "sigmoid":
        s = torch.sigmoid(su + 10) + 1e-6
        return s, torch.log(s)
    if fn == "translate":
        return torch.exp(su * 0), su * 0
    raise ValueError(fn)


def affine_coupling_forward(p: Params, prefix: str, z: Tensor, ctx: Tensor, mask: Optional[Tensor],
                            n_layers: int, scaling_fn: str = "tanh",
                            activation: str = "softplus", partial: bool = True,
                            kinks: Optional[Dict] = None) -> Tuple[Tensor, Tensor]:
    """AffineTransformationLayer.forward (affine_model='wavenet').  common.py:1163-1185."""
    h = z.shape[1] // 2
    z0, z1 = z[:, :h], z[:, h:]
    o = wn_forward(p, prefix + "affine_param_predictor.", z0, ctx, mask, n_layers, activation, partial, kinks)
    s, log_s = scaling_and_log(o[:, :h], scaling_fn)
    return torch.cat((z0, s * z1 + o[:, h:]), 1), log_s


def affine_coupling_inverse(p: Params, prefix: str, z: Tensor, ctx: Tensor, mask: Optional[Tensor],
                            n_layers: int, scaling_fn: str = "tanh",
                            activation: str = "softplus", partial: bool = True) -> Tensor:
    """common.py:1178-1181."""
    h = z.shape[1] // 2
    z0, z1 = z[:, :h], z[:, h:]
    o = wn_forward(p, prefix + "affine_param_predictor.", z0, ctx, mask, n_layers, activation, partial)
    s, _ = scaling_and_log(o[:, :h], scaling_fn)
    return torch.cat((z0, (z1 - o[:, h:]) / s), 1)


# --------------------------------------------------------------------------
# invertible 1x1 convs
# --------------------------------------------------------------------------
def lus_weight(p: Params, prefix: str) -> Tensor:
    """W = P (L U), L unit-lower, U = triu(upper,1)+diag(upper_diag).  common.py:529-531."""
    U = torch.triu(p[prefix + "upper"], 1) + torch.diag(p[prefix + "upper_diag"])
    L = torch.tril(p[prefix + "lower"], -1) + torch.diag(p[prefix + "lower_diag"])
    return p[prefix + "p"] @ (L @ U)


def inv1x1_lus_forward(p: Params, prefix: str, z: Tensor) -> Tuple[Tensor, Tensor]:
    """Invertible1x1ConvLUS.forward.  common.py:527-548."""
    W = lus_weight(p, prefix)
    return F.conv1d(z, W[..., None]), torch.log(torch.abs(p[prefix + "upper_diag"])).sum()


def whiten_weight(p: Params, prefix: str) -> Tensor:
    """common.py:598."""
    return torch.triu(p[prefix + "upper"], 1) + torch.diag(p[prefix + "upper_diag"])


def inv1x1_whiten_forward(p: Params, prefix: str, z: Tensor) -> Tuple[Tensor, Tensor]:
    """DataInitializedInvertible1x1Conv.forward (already initialised).  common.py:612-617."""
    W = whiten_weight(p, prefix)
    z = z - p[prefix + "input_mean"].unsqueeze(0)
    return F.conv1d(z, W[..., None]), torch.log(torch.abs(p[prefix + "upper_diag"])).sum()


def whiten_initialize(z: Tensor, lengths: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Data-dependent init of flow 0: returns (input_mean [C,1], upper_diag [C], upper [C,C]).

    common.py:569-591: gather the valid frames of every item, covariance / N,
    inverse, upper Cholesky factor.
    """
    cols = [z[b, :, : int(lengths[b])] for b in range(z.shape[0])]
    data = torch.cat(cols, 1)
    N = data.shape[1]
    mean = data.mean(1, keepdim=True)
    cen = data - mean
    cov = (cen @ cen.t()) / N
    wm = torch.linalg.cholesky(torch.inverse(cov), upper=True).contiguous()
    return mean, torch.diag(wm).clone(), torch.triu(wm, 1)


# --------------------------------------------------------------------------
# splines (piecewise quadratic, zunis-style)  splines.py:241-339
# --------------------------------------------------------------------------
def weighted_softmax(v: Tensor, w: Tensor) -> Tensor:
    """splines.py:267-272."""
    v = v - v.max(dim=-1, keepdim=True)[0]
    v = torch.exp(v) + 1e-8
    area = ((v[..., :-1] + v[..., 1:]) / 2 * w).sum(-1, keepdim=True)
    return v / area


def piecewise_quadratic_transform(x: Tensor, w_tilde: Tensor, v_tilde: Tensor,
                                  inverse: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """splines.py:274-339.  x in [0,1), any leading shape; w_tilde [...,K], v_tilde [...,K+1]."""
    eps = torch.finfo(x.dtype).eps
    w = torch.softmax(w_tilde, -1)
    v = weighted_softmax(v_tilde, w)
    wc = torch.cumsum(w, -1)
    wc[..., -1] = 1.0
    wc_shift = F.pad(wc, (1, 0))
    cdf = torch.cumsum((v[..., 1:] + v[..., :-1]) / 2 * w, -1)
    cdf[..., -1] = 1.0
    cdf_shift = F.pad(cdf, (1, 0))
    edges = cdf if inverse else wc
    idx = torch.searchsorted(edges, x.unsqueeze(-1))
    take = lambda t, i: torch.gather(t, -1, i).squeeze(-1)
    w_b, w_l = take(w, idx), take(wc_shift, idx)
    v_b, v_r = take(v, idx), take(v, idx + 1)
    c_l = take(cdf_shift, idx)
    if not inverse:
        a = (x - w_l) / w_b.clamp(min=eps)
        y = a ** 2 / 2 * (v_r - v_b) * w_b + a * v_b * w_b + c_l
        logj = torch.lerp(v_b, v_r, a).clamp(min=eps).log()
        return y.clamp(min=eps, max=1.0 - eps), logj
    qa = (v_r - v_b) * w_b / 2
    qb = v_b * w_b
    qc = c_l - x
    a = (-qb + torch.sqrt(qb ** 2 - 4 * qa * qc)) / (2 * qa)
    return (a * w_b + w_l).clamp(min=eps, max=1.0 - eps), None


def unbounded_piecewise_quadratic_transform(x: Tensor, w_tilde: Tensor, v_tilde: Tensor,
                                            upper: float = 1, lower: float = 0,
                                            inverse: bool = False):
    """Identity outside [lower, upper).  splines.py:241-265.

    Restated with a select instead of boolean-mask gather/scatter (same values:
    the inside branch is evaluated on clamped stand-ins for outside elements
    and discarded).
    """
    rng = upper - lower
    inside = (x >= lower) & (x < upper)
    xs = torch.where(inside, (x - lower) / rng, torch.full_like(x, 0.5))
    y, logj = piecewise_quadratic_transform(xs, w_tilde, v_tilde, inverse)
    out = torch.where(inside, y * rng + lower, x)
    if inverse:
        return out, None
    return out, torch.where(inside, logj, torch.zeros_like(logj))


def masked_batchnorm1d(p: Params, prefix: str, x: Tensor, mask: Tensor, training: bool,
                       eps: float = 1e-5) -> Tensor:
    """MaskedBatchNorm1d.forward, single process.  maskedbatchnorm1d.py:53-118.

    Running-stat updates are not restated (side effect; they do not enter the
    training-mode output)."""
    n = mask.sum()
    me = mask.expand(x.shape)
    if training and n > 1:
        mean = (me * x).sum([0, 2]) / n
        var = (me * x ** 2).sum([0, 2]) / n - mean ** 2
    else:
        mean, var = p[prefix + "running_mean"], p[prefix + "running_var"]
    y = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    return y * p[prefix + "weight"][None, :, None] + p[prefix + "bias"][None, :, None]


def film_stack_forward(p: Params, prefix: str, x: Tensor, ctx: Tensor, mask: Tensor, n_layers: int,
                       use_bn: bool, training: bool = True, record: Optional[Dict] = None) -> Tensor:
    """FiLMStack / FiLMResBlock.  common.py:706-773.  record (test accounting): "leaky_near_zero" / "leaky_total" count the
    leaky-ReLU pre-activations (valid frames) within 2e-5 of their tensor's rms of the kink at 0 -- another implementation's
    rounding may put those on the other side, which changes the slope its gradient sees from 1 to 0.01."""
    def count(pre):
        if record is not None:
            with torch.no_grad():
                v = mask.expand_as(pre) > 0
                rms = float(pre[v].pow(2).mean().sqrt())
                record["leaky_near_zero"] = record.get("leaky_near_zero", 0) + int(((pre.abs() < 2e-5 * rms) & v).sum())
                record["leaky_total"] = record.get("leaky_total", 0) + int(v.sum())
    for i in range(n_layers):
        q = f"{prefix}in_layers.{i}."
        x1 = conv_norm(p, q + "input_conv.", x, mask, 1)
        c1 = conv_norm(p, q + "cond_conv.", ctx, mask, 1)
        n_out = x1.shape[1]
        scale, bias = c1[:, :n_out] + 1, c1[:, n_out:]
        count(x1)
        x1r = F.leaky_relu(x1)
        x2 = conv_norm(p, q + "hidden_conv.", x1r, mask, 2 ** i)
        if use_bn:
            x2 = masked_batchnorm1d(p, q + "bn.", x2, mask, training)
        count(x2 * scale + bias)
        x2 = F.leaky_relu(x2 * scale + bias)
        x = 0.5 * (x2 + x1r)
    return F.conv1d(x, p[prefix + "end.weight"], p[prefix + "end.bias"])


def spline_coupling_forward(p: Params, prefix: str, z: Tensor, ctx: Tensor, mask: Tensor,
                            n_layers: int, n_bins: int = 32, bound: float = 3.0,
                            use_bn: bool = True, training: bool = True, record: Optional[Dict] = None) -> Tuple[Tensor, Tensor]:
    """SplineTransformationLayer.forward, use_quadratic=True.  common.py:1040-1090
    (left=bottom=-bound, right=top=bound as wired by decoders.py:51-61).
    record (test accounting, not reference behaviour): receives "x" [B*T, h] (the transform's argument in [0, 1)) and
    "edges" [B*T, h, K] (the bin edges the search runs on) -- an element within an ulp of an edge may take the
    neighbouring bin in another implementation; the transform is continuous there, its parameter gradient is not."""
    B, C, T = z.shape
    h = C // 2
    z0, z1 = z[:, :h], z[:, h:]
    z1n = (z1 + bound) / (2 * bound)
    q = film_stack_forward(p, prefix + "param_predictor.", z0, ctx, mask, n_layers, use_bn, training,
                           record if (record is not None and "ulps" in record) else None)
    nb = 2 * n_bins + 1
    x = z1n.permute(0, 2, 1).reshape(B * T, h)
    qt = q.permute(0, 2, 1).reshape(B * T, h, nb)
    if record is not None:
        with torch.no_grad():
            wc = torch.cumsum(torch.softmax(qt[:, :, :nb // 2].float(), -1), -1)
            wc[..., -1] = 1.0
            if "ulps" in record:                      # full-size runs: only the verdict, not 330 MB of edges per flow
                xf = x.detach().float()
                dist = (wc - xf.unsqueeze(-1)).abs().min(-1)[0]
                inside = (xf >= 0) & (xf < 1)
                record["near"] = inside & (dist <= record["ulps"] * torch.finfo(torch.float32).eps)
                # the bin the reference's search picks (splines.py:300-306), -1 outside [0, 1): for index-by-index
                # comparison with another implementation's search
                idx = torch.searchsorted(wc, xf.unsqueeze(-1)).squeeze(-1)
                record["bins"] = torch.where(inside, idx, torch.full_like(idx, -1)).to(torch.int32)
                record["edge_dist"] = dist
            else:
                record["x"], record["edges"] = x.detach().float().clone(), wc
    y, logj = unbounded_piecewise_quadratic_transform(
        x.float(), qt[:, :, :nb // 2].float(), qt[:, :, nb // 2:].float())
    z1o = y.reshape(B, T, h).permute(0, 2, 1) * (2 * bound) - bound
    log_s = logj.sum(1).reshape(B, T).unsqueeze(1)  # + h*(log(2b)-log(2b)) == 0
    return torch.cat((z0, z1o), 1), log_s


def spline_coupling_inverse(p: Params, prefix: str, z: Tensor, ctx: Tensor, mask: Tensor,
                            n_layers: int, n_bins: int = 32, bound: float = 3.0,
                            use_bn: bool = True) -> Tensor:
    """SplineTransformationLayer.forward(inverse=True), eval mode.  common.py:1040-1090."""
    B, C, T = z.shape
    h = C // 2
    z0, z1 = z[:, :h], z[:, h:]
    z1n = (z1 + bound) / (2 * bound)
    q = film_stack_forward(p, prefix + "param_predictor.", z0, ctx, mask, n_layers, use_bn, training=False)
    nb = 2 * n_bins + 1
    y = z1n.permute(0, 2, 1).reshape(B * T, h)
    qt = q.permute(0, 2, 1).reshape(B * T, h, nb)
    x, _ = unbounded_piecewise_quadratic_transform(y.float(), qt[:, :, :nb // 2].float(), qt[:, :, nb // 2:].float(),
                                                   inverse=True)
    z1o = x.reshape(B, T, h).permute(0, 2, 1) * (2 * bound) - bound
    return torch.cat((z0, z1o), 1)


# --------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------
def lstm_bidir_packed(p: Params, prefix: str, x: Tensor, lengths: Tensor, hidden: int) -> Tensor:
    """Packed bi-LSTM over [B,T,F] -> [B,T,2H], zeros at padded frames.
    models/radmmm.py:136-146 (torch.nn.LSTM is torch itself, available here)."""
    lstm = torch.nn.LSTM(x.shape[-1], hidden, num_layers=1, batch_first=True, bidirectional=True)
    lstm = lstm.to(x.dtype)
    sd = {k: p[prefix + k] for k in lstm.state_dict().keys()}
    # keep autograd connectivity to p: functional call
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True,
                                                     enforce_sorted=False)
    out, _ = torch.func.functional_call(lstm, sd, (packed,))
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
    return y


def preprocess_context(p: Params, cfg: DecoderConfig, context: Tensor, spk: Tensor, lengths: Tensor,
                       f0: Optional[Tensor], energy: Optional[Tensor],
                       accent: Optional[Tensor]) -> Tensor:
    """RADMMM.preprocess_context.  models/radmmm.py:103-148."""
    g = cfg.n_group_size
    ctx = squeeze_time(context, g)
    T = ctx.shape[2]
    parts = [ctx, spk[:, :, None].expand(-1, -1, T)]
    if cfg.use_accent_emb_for_decoder:
        parts.append(accent[:, :, None].expand(-1, -1, T))
    if cfg.context_w_f0_and_energy:
        if f0 is not None:
            parts.append(squeeze_time(f0[:, None], g))
        if energy is not None:
            parts.append(squeeze_time(energy[:, None], g))
    x = torch.cat(parts, 1)
    if not cfg.use_context_lstm:
        return x
    ul = torch.div(lengths, g, rounding_mode="floor").long()
    y = lstm_bidir_packed(p, "context_lstm.", x.transpose(1, 2), ul, cfg.lstm_hidden)
    return y.transpose(1, 2)


def decoder_forward(p: Params, cfg: DecoderConfig, mel: Tensor, spk: Tensor, context: Tensor,
                    lengths: Tensor, f0: Optional[Tensor] = None, energy: Optional[Tensor] = None,
                    accent: Optional[Tensor] = None, training: bool = True,
                    spline_records: Optional[List[Dict]] = None, wn_kinks: Optional[Dict] = None) -> Dict[str, object]:
    """RADMMMFlow.forward.  decoders.py:168-205.  spline_records (test accounting): one dict per spline flow, handed to
    spline_coupling_forward as `record`; wn_kinks (test accounting): wn_forward's `kinks`."""
    g = cfg.n_group_size
    ctx = preprocess_context(p, cfg, context, spk, lengths, f0, energy, accent)
    z = squeeze_time(mel, g)
    ul = torch.div(lengths, g, rounding_mode="floor").long()
    mask = lengths_to_mask(ul)[:, None].to(mel.dtype)
    exits = cfg.exit_steps()
    z_out, log_s_list, log_det_list = [], [], []
    for i in range(cfg.n_flows):
        if i in exits:
            z_out.append(z[:, : cfg.n_early_size])
            z = z[:, cfg.n_early_size:]
        pre = f"flows.{i}."
        if i == 0:
            z, ld = inv1x1_whiten_forward(p, pre + "invtbl_conv.", z)
        else:
            z, ld = inv1x1_lus_forward(p, pre + "invtbl_conv.", z)
        if i < cfg.n_splines:
            z, ls = spline_coupling_forward(p, pre + "coupling_tfn.", z, ctx, mask,
                                            cfg.n_conv_layers_per_step, use_bn=cfg.use_bn,
                                            training=training,
                                            record=spline_records[i] if spline_records is not None else None)
        else:
            z, ls = affine_coupling_forward(p, pre + "coupling_tfn.", z, ctx, mask,
                                            cfg.n_conv_layers_per_step, cfg.scaling_fn,
                                            cfg.affine_activation, cfg.use_partial_padding, wn_kinks)
        log_s_list.append(ls)
        log_det_list.append(ld)
    z_out.append(z)
    return {"z_mel": torch.cat(z_out, 1), "log_det_W_list": log_det_list,
            "log_s_list": log_s_list, "context_w_spkvec": ctx}


def compute_flow_loss(z: Tensor, log_det_W_list: Sequence[Tensor], log_s_list: Sequence[Tensor],
                      n_elements, n_dims: int, mask: Tensor, sigma: float = 1.0
                      ) -> Tuple[Tensor, Tensor]:
    """loss.py:85-110 (no log 2*pi term)."""
    log_s_total = sum(torch.sum(ls * mask) for ls in log_s_list)
    log_det_total = sum(log_det_W_list) * n_elements if len(log_det_W_list) else 0.0
    zm = z * mask
    prior = torch.sum(zm * zm) / (2 * sigma * sigma)
    denom = n_elements * n_dims
    return (prior - log_s_total - log_det_total) / denom, prior / denom


def decoder_loss(out: Dict[str, object], lengths: Tensor, g: int, sigma: float = 1.0):
    """The flow part of RADMMMLoss.forward.  loss.py:518-532."""
    n_el = torch.div(lengths.sum(), g, rounding_mode="floor")
    mask = lengths_to_mask(torch.div(lengths, g, rounding_mode="floor"))[:, None].float()
    z = out["z_mel"]
    return compute_flow_loss(z, out["log_det_W_list"], out["log_s_list"], n_el, z.shape[1],
                             mask.to(z.dtype), sigma)


# --------------------------------------------------------------------------
# alignment attention, MAS, attention losses
# --------------------------------------------------------------------------
def conv_attention_forward(p: Params, prefix: str, queries: Tensor, keys: Tensor,
                           key_pad_mask: Optional[Tensor], attn_prior: Optional[Tensor]
                           ) -> Tuple[Tensor, Tensor]:
    """ConvAttention.forward.  common.py:1239-1277.

    queries [B,n_mel,T1], keys [B,n_txt,T2], key_pad_mask bool [B,T2,1] True at
    PADDED text positions, attn_prior [B,T1,T2].  Returns (attn, attn_logprob)
    both [B,1,T1,T2].
    """
    def cn(name, x, k):
        w = _wn_weight(p, f"{prefix}{name}.conv.")
        return F.conv1d(x, w, p[f"{prefix}{name}.conv.bias"], padding=(k - 1) // 2)
    kk = cn("key_proj.2", torch.relu(cn("key_proj.0", keys, 3)), 1)
    q = torch.relu(cn("query_proj.0", queries, 3))
    q = torch.relu(cn("query_proj.2", q, 1))
    q = cn("query_proj.4", q, 1)
    d = ((q[:, :, :, None] - kk[:, :, None]) ** 2).sum(1, keepdim=True)
    a = -0.0005 * d
    if attn_prior is not None:
        a = torch.log_softmax(a, 3) + torch.log(attn_prior[:, None] + 1e-8)
    logprob = a.clone()
    if key_pad_mask is not None:
        a = a.masked_fill(key_pad_mask.permute(0, 2, 1).unsqueeze(2), -float("inf"))
    return torch.softmax(a, 3), logprob


def mas_width1(attn_map: np.ndarray) -> np.ndarray:
    """Monotonic alignment search, width 1 (INTEGER/INDEX path, bit-exact).

    alignment.py:31-59: Viterbi over log(attn) [T_mel, T_txt]; first row forced
    to column 0; moves are {stay, +1}; ties go to the diagonal (>=); backtrack
    from the last text index.  Output 0/1 matrix of attn_map's dtype.
    """
    T1, T2 = attn_map.shape
    with np.errstate(divide="ignore"):
        lp = np.log(attn_map)
    lp[0, 1:] = -np.inf
    acc = np.zeros_like(lp)
    acc[0] = lp[0]
    prev = np.zeros((T1, T2), dtype=np.int64)
    cols = np.arange(T2)
    for i in range(1, T1):
        stay = acc[i - 1]
        diag = np.concatenate(([-np.inf], acc[i - 1, :-1])).astype(lp.dtype)
        take_diag = diag >= stay
        take_diag[0] = False
        acc[i] = lp[i] + np.where(take_diag, diag, stay)
        prev[i] = np.where(take_diag, cols - 1, cols)
    opt = np.zeros_like(attn_map)
    j = T2 - 1
    for i in range(T1 - 1, -1, -1):
        opt[i, j] = 1
        j = prev[i, j]
    opt[0, j] = 1
    return opt


def mas_width1_c(attn_map: np.ndarray, logp: np.ndarray = None) -> np.ndarray:
    """Same search through the plain-C restatement (oracle/mas_ref.c, built by oracle/Makefile);
    the log is taken here with numpy exactly as alignment.py:36 does (or `logp` is used as given:
    the search on a caller-supplied float32 log, for comparisons on identical log inputs)."""
    import ctypes
    import os
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmas_ref.so")
    lib = ctypes.CDLL(so)
    T1, T2 = attn_map.shape
    with np.errstate(divide="ignore"):
        lp = np.ascontiguousarray(np.log(attn_map.astype(np.float32)) if logp is None else logp.astype(np.float32))
    opt = np.zeros((T1, T2), dtype=np.float32)
    rc = lib.mas_width1_ref(lp.ctypes.data_as(ctypes.c_void_p), T1, T2, opt.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError(f"mas_width1_ref failed: {rc}")
    return opt.astype(attn_map.dtype)


def binarize_attention(attn: Tensor, in_lens: Tensor, out_lens: Tensor) -> Tensor:
    """TTSModel.binarize_attention.  tts_lightning_modules.py:270-284.
    attn [B,1,T_mel_max,T_txt_max] -> same shape 0/1."""
    a = attn.detach().cpu().numpy()
    out = np.zeros_like(a)
    for b in range(a.shape[0]):
        out[b, 0, : int(out_lens[b]), : int(in_lens[b])] = mas_width1(
            a[b, 0, : int(out_lens[b]), : int(in_lens[b])].copy())
    return torch.from_numpy(out)


def attention_ctc_loss(attn_logprob: Tensor, in_lens: Tensor, out_lens: Tensor,
                       blank_logprob: float = -1) -> Tensor:
    """AttentionCTCLoss.forward.  loss.py:119-141."""
    padded = F.pad(attn_logprob, (1, 0), value=blank_logprob)
    total = 0.0
    B = attn_logprob.shape[0]
    for b in range(B):
        kl, ql = int(in_lens[b]), int(out_lens[b])
        tgt = torch.arange(1, kl + 1).unsqueeze(0)
        lp = padded[b].permute(1, 0, 2)[:ql, :, : kl + 1]
        lp = torch.log_softmax(lp, -1)
        total = total + F.ctc_loss(lp, tgt, input_lengths=out_lens[b:b + 1],
                                   target_lengths=in_lens[b:b + 1], zero_infinity=True)
    return total / B


def attention_binarization_loss(hard: Tensor, soft: Tensor) -> Tensor:
    """AttentionBinarizationLoss.forward.  loss.py:147-151:  mean(-log soft[hard==1])
    (torch BCE clamps log at -100)."""
    sel = soft[hard == 1]
    return F.binary_cross_entropy(sel, torch.ones_like(sel), reduction="mean")


# --------------------------------------------------------------------------
# STFT -> mel
# --------------------------------------------------------------------------
def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True)  (audio_processing.py:212)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_magnitude(audio: np.ndarray, n_fft: int, hop: int, win: int) -> np.ndarray:
    """STFT.transform magnitude.  audio_processing.py:227-255.

    audio [B, S] -> [B, n_fft//2+1, 1 + S//hop].  Reflect-pad n_fft/2 both
    sides, frames of n_fft at stride hop, window = periodic hann(win) centre-
    padded to n_fft, DFT basis in float32 as the reference's conv1d weights.
    """
    B, S = audio.shape
    w = hann_periodic(win)
    lp = (n_fft - win) // 2
    w = np.pad(w, (lp, n_fft - win - lp))
    cutoff = n_fft // 2 + 1
    k = np.arange(cutoff)[:, None]
    n = np.arange(n_fft)[None, :]
    ang = 2.0 * np.pi * k * n / n_fft
    basis_re = (np.cos(ang)).astype(np.float32) * w.astype(np.float32)
    basis_im = (-np.sin(ang)).astype(np.float32) * w.astype(np.float32)
    x = np.pad(audio.astype(np.float32), ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    n_frames = (x.shape[1] - n_fft) // hop + 1
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[:, idx]                                # [B, F, n_fft]
    re = np.einsum("bfn,kn->bkf", frames, basis_re, dtype=np.float32)
    im = np.einsum("bfn,kn->bkf", frames, basis_im, dtype=np.float32)
    return np.sqrt(re * re + im * im).astype(np.float32)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank_slaney(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: Optional[float]
                          ) -> np.ndarray:
    """Published algorithm of librosa 0.8.0 `filters.mel(sr, n_fft, n_mels, fmin, fmax)`
    with its defaults (htk=False -> Slaney scale, norm='slaney'); call site
    audio_processing.py:124-125.  librosa is absent from the reference tree and from this
    image: pinned (bit-equal in float32) to transformers.audio_utils.mel_filter_bank(norm =
    mel_scale = "slaney"), an independent implementation that project validates against
    librosa (tests/golden/mel_basis_hf.npz, tests/test_oracle_golden.py).
    """
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    wts = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        wts[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (wts * enorm[:, None]).astype(np.float32)


def mel_spectrogram(audio: np.ndarray, mel_basis: np.ndarray, n_fft: int, hop: int, win: int
                    ) -> np.ndarray:
    """TacotronSTFT.mel_spectrogram.  audio_processing.py:137-154 + :98-104."""
    mag = stft_magnitude(audio, n_fft, hop, win)
    mel = np.einsum("mk,bkf->bmf", mel_basis.astype(np.float32), mag, dtype=np.float32)
    return np.log(np.maximum(mel, 1e-5)).astype(np.float32)


# --------------------------------------------------------------------------
# optimizer step: RAdam as vendored by the reference + Lightning's norm clip
# --------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[Tensor], max_norm: float) -> Tuple[Tensor, List[Tensor]]:
    """torch.nn.utils.clip_grad_norm_ (what Lightning's gradient_clip_val=1.0 / algorithm 'norm'
    calls, configs/RADMMM_train_config.yaml:7-8): coef = min(1, max_norm / (||g||_2 + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def radam_scalars(step: int, lr: float, beta1: float, beta2: float) -> Tuple[float, float]:
    """(N_sma, step_size) of radam.py:101-123 for the 1-based step count."""
    beta2_t = beta2 ** step
    n_sma_max = 2.0 / (1.0 - beta2) - 1.0
    n_sma = n_sma_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
    if n_sma >= 5:
        step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma
                                   * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** step)
    else:
        step_size = lr / (1 - beta1 ** step)
    return n_sma, step_size


def radam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-3,
               betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
    """One RAdam update of one tensor (radam.py:63-142), in place on p, m, v; `step` is 1-based."""
    beta1, beta2 = betas
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    n_sma, step_size = radam_scalars(step, lr, beta1, beta2)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    if n_sma >= 5:
        p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
    else:
        p.add_(m, alpha=-step_size)
    return p


# --------------------------------------------------------------------------
# decoder.infer: z -> mel, affine flows
# --------------------------------------------------------------------------
def length_regulate(x: Tensor, dur: Tensor) -> Tensor:
    """LengthRegulator.forward (common.py:208-237): x [B, T_txt, C], dur [B, T_txt] ints ->
    [B, max_b sum(dur_b), C], frame i repeated dur[i] times, zero padded."""
    outs = [torch.repeat_interleave(x_i, d_i.long(), dim=0) for x_i, d_i in zip(x, dur)]
    T = max(o.shape[0] for o in outs)
    return torch.stack([F.pad(o, (0, 0, 0, T - o.shape[0])) for o in outs])


def inv1x1_lus_inverse(p: Params, prefix: str, z: Tensor) -> Tensor:
    """common.py:532-541."""
    return F.conv1d(z, torch.inverse(lus_weight(p, prefix).float())[..., None])


def inv1x1_whiten_inverse(p: Params, prefix: str, z: Tensor) -> Tensor:
    """common.py:599-607."""
    z = F.conv1d(z, torch.inverse(whiten_weight(p, prefix).float())[..., None])
    return z + p[prefix + "input_mean"].unsqueeze(0)


def fold_time(z: Tensor, g: int) -> Tensor:
    """inverse of squeeze_time (nn.Fold with kernel (g,1), decoders.py:123-126,245-246):
    [B, C*g, T'] -> [B, C, T'*g] with out[b, c, l*g+k] = z[b, c*g+k, l]."""
    B, Cg, T = z.shape
    return z.reshape(B, Cg // g, g, T).permute(0, 1, 3, 2).reshape(B, Cg // g, T * g)


def decoder_infer(p: Params, cfg: DecoderConfig, spk: Tensor, txt_enc: Tensor, residual: Tensor, dur: Tensor,
                  out_lens: Tensor, f0: Optional[Tensor] = None, energy: Optional[Tensor] = None,
                  accent: Optional[Tensor] = None) -> Tensor:
    """RADMMMFlow.infer (decoders.py:207-248) with the noise `residual` [B, n_mel*g, T'] (already
    multiplied by sigma) supplied by the caller; eval mode (spline flows use the batch-norm running
    statistics)."""
    g = cfg.n_group_size
    ctx_t = length_regulate(txt_enc.transpose(1, 2), dur).transpose(1, 2)
    ctx = preprocess_context(p, cfg, ctx_t, spk, out_lens, f0, energy, accent)
    exits = list(cfg.exit_steps())
    ne = cfg.n_early_size
    mel = residual[:, len(exits) * ne:]
    remaining = residual[:, : len(exits) * ne]
    ul = torch.div(out_lens, g, rounding_mode="floor").long()
    mask = lengths_to_mask(ul)[:, None].to(residual.dtype)
    for i in reversed(range(cfg.n_flows)):
        pre = f"flows.{i}."
        if i < cfg.n_splines:
            mel = spline_coupling_inverse(p, pre + "coupling_tfn.", mel, ctx, mask, cfg.n_conv_layers_per_step,
                                          use_bn=cfg.use_bn)
        else:
            mel = affine_coupling_inverse(p, pre + "coupling_tfn.", mel, ctx, mask, cfg.n_conv_layers_per_step,
                                          cfg.scaling_fn, cfg.affine_activation, cfg.use_partial_padding)
        mel = inv1x1_whiten_inverse(p, pre + "invtbl_conv.", mel) if i == 0 else inv1x1_lus_inverse(p, pre + "invtbl_conv.", mel)
        if exits and i == exits[-1]:
            exits.pop()
            mel = torch.cat((remaining[:, len(exits) * ne:], mel), 1)
            remaining = remaining[:, : len(exits) * ne]
    return fold_time(mel, g) if g > 1 else mel


# --------------------------------------------------------------------------
# text Encoder
# --------------------------------------------------------------------------
def spectral_weight(orig: Tensor, u: Tensor, v: Tensor) -> Tensor:
    """torch.nn.utils.spectral_norm in eval mode (no power iteration): W / sigma, sigma = u^T W v."""
    return orig / torch.dot(u, torch.mv(orig, v))


def encoder_forward(p: Params, prefix: str, x: Tensor, in_lens: Tensor, n_conv: int = 3) -> Tensor:
    """common.Encoder.forward in eval mode (common.py:461-493): per utterance, n_conv x
    [weight-normed PartialConv1d k5 on the valid frames -> InstanceNorm1d(affine) -> ReLU], then the
    packed bi-LSTM over the padded batch.  x [B, C, L] -> [B, max(in_lens), C].  The LSTM's recurrent
    weights may be stored plain or spectrally normalised (`*_orig`, `*_u`, `*_v`)."""
    outs = []
    for b in range(x.shape[0]):
        n = int(in_lens[b])
        cur = x[b: b + 1, :, :n]
        mask = torch.ones(1, 1, n, dtype=x.dtype)
        for i in range(n_conv):
            pre = f"{prefix}convolutions.{i}."
            w = weight_norm_fold(p[pre + "0.conv.weight_v"], p[pre + "0.conv.weight_g"])
            cur = partial_conv1d(cur, mask, w, p[pre + "0.conv.bias"], 1)
            cur = F.instance_norm(cur, weight=p[pre + "1.weight"], bias=p[pre + "1.bias"], eps=1e-5)
            cur = torch.relu(cur)
        outs.append(cur[0].transpose(0, 1))
    xp = torch.nn.utils.rnn.pad_sequence(outs, batch_first=True)
    lp = dict(p)
    for suf in ("", "_reverse"):
        key = f"{prefix}lstm.weight_hh_l0{suf}"
        if key not in lp:
            lp[key] = spectral_weight(p[key + "_orig"], p[key + "_u"], p[key + "_v"])
    C = x.shape[1]
    return lstm_bidir_packed(lp, prefix + "lstm.", xp, in_lens, C // 2)


# --------------------------------------------------------------------------
# attribute predictors
# --------------------------------------------------------------------------
def dap_tx_data(x: Tensor, target_scale: float, target_offset: float, log_target: bool) -> Tensor:
    """AttributePredictor.tx_data without target normalisation (attribute_predictors.py:100-104)."""
    x = x * target_scale + target_offset
    return torch.log(x + 1) if log_target else x


def dap_forward(p: Params, prefix: str, text_enc: Tensor, spk: Tensor, lens: Tensor, n_layers: int,
                gates: Optional[Dict] = None, record: Optional[Dict] = None) -> Tensor:
    """ConvLSTMLinearDAP.forward in eval mode (attribute_predictors.py:172-192, common.py:281-333):
    bottleneck = leaky_relu(mask * weight-normed conv k3 (text_enc)); cat speaker; per utterance
    n_layers x relu(weight-normed conv k3) on the valid frames; packed spectral-normed bi-LSTM;
    linear.  text_enc [B, C, T] -> x_hat [B, out_dim, T'].
    Test accounting of the (leaky) ReLU's kink (not reference behaviour): `record` receives the pre-activations
    ("bottleneck": [B, C, T]; (b, i): [1, C, len_b] of utterance b, layer i); `gates` (same keys, bool) overrides the
    sign decision pre > 0 of the activation AND of its derivative -- an implementation whose pre-activation differs in
    the last bits may sit on the other side of 0 for a handful of elements, which moves upstream gradients by O(1e-3);
    with its decisions imposed the two gradients are comparable at full precision."""
    T = text_enc.shape[2]
    mask = lengths_to_mask(lens, T)[:, None].to(text_enc.dtype)
    pre = prefix + "bottleneck_layer.projection_fn.conv."
    w = weight_norm_fold(p[pre + "weight_v"], p[pre + "weight_g"])
    k = w.shape[-1]
    pb = F.conv1d(text_enc, w, p[pre + "bias"], padding=(k - 1) // 2) * mask
    if record is not None:
        record["bottleneck"] = pb.detach()
    ctx = F.leaky_relu(pb) if gates is None else torch.where(gates["bottleneck"], pb, 0.01 * pb)
    ctx = torch.cat((ctx, spk[:, :, None].expand(-1, -1, T)), 1)
    outs = []
    for b in range(ctx.shape[0]):
        cur = ctx[b: b + 1, :, : int(lens[b])]
        for i in range(n_layers):
            q = f"{prefix}feat_pred_fn.convolutions.{i}.conv."
            w = weight_norm_fold(p[q + "weight_v"], p[q + "weight_g"])
            pc = F.conv1d(cur, w, p[q + "bias"], padding=(w.shape[-1] - 1) // 2)
            if record is not None:
                record[(b, i)] = pc.detach()
            cur = torch.relu(pc) if gates is None else torch.where(gates[(b, i)], pc, torch.zeros_like(pc))
        outs.append(cur[0].transpose(0, 1))
    xp = torch.nn.utils.rnn.pad_sequence(outs, batch_first=True)
    lp = dict(p)
    lpre = prefix + "feat_pred_fn.bilstm."
    for suf in ("", "_reverse"):
        key = f"{lpre}weight_hh_l0{suf}"
        if key not in lp:
            lp[key] = spectral_weight(p[key + "_orig"], p[key + "_u"], p[key + "_v"])
    y = lstm_bidir_packed(lp, lpre, xp, lens, xp.shape[2] // 2)
    return F.linear(y, p[prefix + "feat_pred_fn.dense.weight"], p[prefix + "feat_pred_fn.dense.bias"]).transpose(1, 2)


def tts_joint_step(p: Params, cfg: DecoderConfig, batch: Dict[str, Tensor], predictors: Dict[str, Dict],
                   binarize: bool = True, bin_loss: bool = True, n_enc_conv: int = 3,
                   ctc_loss_weight: float = 0.1, binarization_loss_weight: float = 1.0,
                   dap_gates: Optional[Dict[str, Dict]] = None, dap_record: Optional[Dict[str, Dict]] = None) -> Dict[str, object]:
    """TTSModel.training_step (tts_lightning_modules.py:643-750) with dropout off, on CPU: mel scaling (:543-545), speaker /
    accent / text embeddings (:246-268), text encoder, alignment attention with the prior (:440-475), per-item MAS when
    `binarize` (:270-284), context = txt_enc . attn^T (:669), flow decoder + RADMMMLoss (loss.py:518-537: flow NLL, CTC x
    ctc_loss_weight, binarisation loss when `bin_loss`), then the attribute predictors on DETACHED inputs (:688-727) with
    their AttributeRegressionLoss (loss.py:233-250).  `p` holds the step's state_dict (reference names: text_embeddings.,
    text_encoder., speaker_embeddings., accent_embeddings., attention., decoder., <name>_predictor.);
    `predictors`: {"f0" | "energy" | "voiced" | "duration": dict(n_layers, target_scale, target_offset, log_target, weight,
    prefix)}.  batch: mel [B, 80, T], speaker_ids, accent_ids, text [B, L], input_lengths, output_lengths, attn_prior
    [B, T, L], f0, energy_avg [B, T], voiced_mask [B, T].  Returns losses {name: (value, weight)}, `loss` (their weighted
    sum, :746-749), `pred` {name: x_hat} and the intermediate attn / context.
    Pinned to the reference's own components on tests/golden/tts_step.npz (tests/test_oracle_joint.py).  dap_gates /
    dap_record: {predictor name: dap_forward's `gates` / `record`} -- the test accounting of the predictors' ReLU kinks
    (tests/test_joint_step.py), not reference behaviour."""
    in_lens, out_lens = batch["input_lengths"].long(), batch["output_lengths"].long()
    mel = (batch["mel"] + 5) / 2
    spk = p["speaker_embeddings.weight"][batch["speaker_ids"]]
    acc = p["accent_embeddings.weight"][batch["accent_ids"]]
    txt_emb = p["text_embeddings.weight"][batch["text"]].transpose(1, 2)               # [B, C, L]
    L = int(in_lens.max())
    txt_enc = encoder_forward(p, "text_encoder.", txt_emb, in_lens, n_enc_conv).transpose(1, 2)      # [B, C, L]
    pad_mask = ~lengths_to_mask(in_lens, L)[..., None]
    attn_soft, attn_logprob = conv_attention_forward(p, "attention.", mel, txt_emb[:, :, :L], pad_mask, batch["attn_prior"][:, :, :L])
    attn = attn_soft
    if binarize:
        a = attn_soft.detach().numpy()
        hard = np.zeros_like(a)
        for b in range(a.shape[0]):                     # (the C restatement of the search: bit-equal to mas_width1, tests)
            hard[b, 0, : int(out_lens[b]), : int(in_lens[b])] = mas_width1_c(a[b, 0, : int(out_lens[b]), : int(in_lens[b])].copy())
        # (tts_lightning_modules.py:470-475: the map used for the context is the PLAIN 0/1 alignment -- the straight-through
        #  variant `attn_hard` is computed there and dropped by the caller (:665 `attn, attn_soft, _, attn_logprob`), so the
        #  context carries no gradient into the attention once alignments are binarised.  Round 6: this line was the
        #  straight-through form until tests/test_joint_step.py compared backward passes; forward values are the same)
        attn = torch.from_numpy(hard)
    context = torch.bmm(txt_enc, attn.squeeze(1).transpose(1, 2))
    dp = {k[len("decoder."):]: v for k, v in p.items() if k.startswith("decoder.")}
    out = decoder_forward(dp, cfg, mel, spk, context, out_lens, batch["f0"], batch["energy_avg"], acc)
    losses: Dict[str, Tuple[Tensor, float]] = {}
    lm, lp = decoder_loss(out, out_lens, cfg.n_group_size)
    losses["loss_mel"], losses["loss_prior_mel"] = (lm, 1.0), (lp, 0.0)
    losses["loss_ctc"] = (attention_ctc_loss(attn_logprob, in_lens, out_lens), ctc_loss_weight)
    if bin_loss:
        losses["binarization_loss"] = (attention_binarization_loss(attn, attn_soft), binarization_loss_weight)
    pred = {}
    spk_acc = torch.cat((spk, acc), 1).detach()
    T = mel.shape[2]
    for name, spec in predictors.items():
        pre = f"{name}_predictor."
        if name == "duration":
            target, src, lens, tmask = attn.sum(2).detach(), txt_enc.detach(), in_lens, lengths_to_mask(in_lens, L)[:, None]
        else:
            raw = {"f0": batch["f0"], "energy": batch["energy_avg"], "voiced": batch["voiced_mask"]}[name]
            target, src, lens = raw[:, None], context.detach(), out_lens
            tmask = batch["voiced_mask"][:, None].bool() if name == "f0" else lengths_to_mask(out_lens, T)[:, None]
        x = dap_tx_data(target, spec.get("target_scale", 1.0), spec.get("target_offset", 0.0), spec.get("log_target", False))
        x_hat = dap_forward(p, pre, src, spk_acc, lens, spec["n_layers"], gates=(dap_gates or {}).get(name),
                            record=dap_record.setdefault(name, {}) if dap_record is not None else None)
        w = x_hat.shape[2]
        m = tmask[:, :, :w]
        losses[spec["prefix"] + "loss"] = (F.mse_loss(x_hat[m], x[:, :, :w][m], reduction="sum") / tmask.sum(), spec.get("weight", 1.0))
        pred[name] = x_hat
    total = None
    for v, wgt in losses.values():
        total = v * wgt if total is None else total + v * wgt
    return {"loss": total, "losses": losses, "pred": pred, "attn": attn, "attn_soft": attn_soft, "context": context,
            "txt_enc": txt_enc, "z_mel": out["z_mel"]}


def attribute_regression_loss(x_hat: Tensor, x: Tensor, lens: Tensor) -> Tensor:
    """AttributeRegressionLoss (loss.py:233-250): masked mean squared error."""
    mask = lengths_to_mask(lens, x.shape[2])[:, None]
    return F.mse_loss(x_hat[mask], x[mask], reduction="sum") / mask.sum()


# --------------------------------------------------------------------------
# data path: attention prior and energy average
# --------------------------------------------------------------------------
def beta_binomial_prior(phoneme_count: int, mel_count: int, scaling_factor: float = 0.05):
    """data.py:90-102 (beta_binomial_prior_distribution): row i (1-based frame) is the pmf of
    BetaBinomial(n = P-1, a = s*i, b = s*(M+1-i)) over the P tokens.  The reference calls
    scipy.stats.betabinom (scipy 1.15, the image's pinned version); its published pmf is restated here:
    pmf(k) = C(n, k) B(k + a, n - k + b) / B(a, b), evaluated in log space, float64 -> [M, P]."""
    import numpy as np
    from scipy.special import betaln, gammaln
    P, M = int(phoneme_count), int(mel_count)
    n = P - 1
    k = np.arange(P, dtype=np.float64)[None, :]
    i = np.arange(1, M + 1, dtype=np.float64)[:, None]
    a, b = scaling_factor * i, scaling_factor * (M + 1 - i)
    logc = gammaln(n + 1.0) - gammaln(k + 1.0) - gammaln(n - k + 1.0)
    return np.exp(logc + betaln(k + a, n - k + b) - betaln(a, b))


def prior_bank_shape(p_count: int, m_count: int, round_mel_len_to: int = 100, round_text_len_to: int = 20):
    """data.py:72-79: sizes of the cached anchor prior; numpy's round = round-half-to-even, as Python's."""
    rnd = lambda val, to: max(1, int(round((val + 1) / to))) * to
    return rnd(p_count, round_text_len_to), rnd(m_count, round_mel_len_to)          # (bw, bh)


def zoom_linear(x, out_rows: int, out_cols: int):
    """scipy.ndimage.zoom(x, order=1, mode='nearest', grid_mode=False) as data.py:80 uses it, restated:
    output index o samples the input at o * (n_in - 1) / (n_out - 1) (0 when n_out == 1), bilinear, float64."""
    import numpy as np
    def axis(n_in, n_out):
        z = (n_in - 1) / (n_out - 1) if n_out > 1 else 1.0
        c = np.arange(n_out, dtype=np.float64) * z
        i0 = np.minimum(np.floor(c).astype(np.int64), n_in - 1)
        f = c - i0
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, f
    r0, r1, fr = axis(x.shape[0], out_rows)
    c0, c1, fc = axis(x.shape[1], out_cols)
    top = x[r0][:, c0] * (1 - fc)[None, :] + x[r0][:, c1] * fc[None, :]
    bot = x[r1][:, c0] * (1 - fc)[None, :] + x[r1][:, c1] * fc[None, :]
    return top * (1 - fr)[:, None] + bot * fr[:, None]


def interpolated_prior(p_count: int, m_count: int, scaling_factor: float = 0.05):
    """BetaBinomialInterpolator.__call__ (data.py:76-88): anchor prior at the rounded sizes, bilinear zoom to
    [m_count, p_count], rows renormalised.  float64."""
    bw, bh = prior_bank_shape(p_count, m_count)
    ret = zoom_linear(beta_binomial_prior(bw, bh, scaling_factor), m_count, p_count)
    return ret / ret.sum(1, keepdims=True)


def attention_prior_batch(in_lens, out_lens, scaling_factor: float = 0.05):
    """The padded [B, max_frames, max_tokens] fp32 prior DataCollate builds (data.py:678-679,737-741)."""
    import numpy as np
    B = len(in_lens)
    out = np.zeros((B, int(max(out_lens)), int(max(in_lens))), dtype=np.float32)
    for b in range(B):
        out[b, :int(out_lens[b]), :int(in_lens[b])] = interpolated_prior(int(in_lens[b]), int(out_lens[b]), scaling_factor)
    return out


def energy_average(mel, use_scaled_energy: bool = True):
    """data.py:339-342,363-366: mean over the mel channels of [n_mel, T] (or [B, n_mel, T]), then (x + 20) / 20."""
    e = mel.mean(-2)
    return (e + 20.0) / 20.0 if use_scaled_energy else e


# --------------------------------------------------------------------------
# embedding regularisers and the voiced predictor's loss (caller row a17; configs/RADMMM_model_config.yaml:49-61)
# --------------------------------------------------------------------------
def variance_covariance_reg(embs: Tensor, gamma: float = 1.0) -> Tuple[Tensor, Tensor]:
    """VarianceCovarianceEmbeddingRegLoss.forward, loss.py:325-347 -> (variance loss, covariance loss)."""
    n, d = embs.shape
    std = torch.sqrt(embs.var(dim=0) + 1e-4)
    std_loss = torch.relu(gamma - std).mean()
    cen = embs - embs.mean(dim=0, keepdim=True)
    cov = cen.t() @ cen / (n - 1)
    mask = ~torch.eye(d, dtype=torch.bool)
    return std_loss, (cov[mask] ** 2).sum() / d


def min_cross_covariance(batch1: Tensor, batch2: Tensor, table1: Optional[Tensor], table2: Optional[Tensor]) -> Tensor:
    """AttributeMinCrossCovarianceRegLoss.forward, loss.py:262-296."""
    t1 = batch1 if table1 is None else table1
    t2 = batch2 if table2 is None else table2
    a = batch1 - t1.mean(dim=0, keepdim=True)
    b = batch2 - t2.mean(dim=0, keepdim=True)
    cross = a.t() @ b / (batch1.shape[0] - 1)
    return (cross ** 2).sum() / (t1.shape[1] * t2.shape[1])


def attribute_bce_loss(x_hat: Tensor, x: Tensor, mask: Tensor) -> Tensor:
    """AttributeBCELoss.forward, loss.py:219-230: BCE-with-logits over the masked positions, sum / count."""
    m = mask.bool()
    return F.binary_cross_entropy_with_logits(x_hat[m], x[m], reduction="sum") / m.sum()
