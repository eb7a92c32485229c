#!/usr/bin/env python3
"""Error-vs-MFMA-work table for the split-operand GEMM schemes: the CPU oracle's decoder with
every WN convolution (forward product and data-gradient product) replaced by an emulation of the scheme's rounding:

  h3     Ah.Bh + Ah.Bl + Al.Bh          three f16 MFMA products (the shipped parity mode)          3.00 units
  f8x    Ah.Bh + q8(Ah).q8(Bl) + q8(Al).q8(Bh)   cross terms on the FP8 pipe (e4m3, 2x rate)       2.00 units
  f6x    the same with MXFP6 operands (e2m3, one shared power-of-two scale per 32 channels of K: the block-scaled
         v_mfma_scale_f32_32x32x64_f8f6f4 runs FP6 at the FP4 rate, 4x f16)                           1.50 units
  2pa    Ah.Bh + Al.Bh                  activations exact, weights rounded once to f16               2.00 units
  2pb    Ah.Bh + Ah.Bl                  weights exact, activations rounded once to f16               2.00 units
  1p     Ah.Bh                          the 16-bit throughput mode                                    1.00 unit

A = activations / gradients, B = weights (x256), f16 split x = hi + lo, q8 = OCP e4m3 round-to-nearest with
saturation at 448 (lo parts scaled by 2^11 first).  All split values are exactly representable, so products are exact in
fp32 and only the fp32 accumulation (as on the MFMA) rounds.  Weight gradients stay exact fp32 here (they run on their own
kernel).  Test infrastructure: imports oracle/.

    python tools/emu_products.py [--tag cfg2_small]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import radmmm_oracle as O  # noqa: E402

MODE = "exact"
W_SCALE = 256.0
GRAD_S = [None]


def f16_split(x):
    t = x.clamp(-60000.0, 60000.0)
    hi = t.half().float()
    lo = (t - hi).half().float()
    return hi, lo


def q8(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def q8_lo(lo):
    return q8(lo * 2048.0) / 2048.0


def q6_blocks(x, dim):
    """MXFP6 (e2m3) image of x with one E8M0 scale per block of 32 along `dim` (the contraction axis): scale = 2^(floor(log2
    (block max)) - 2) puts the block maximum into [4, 8); values round to nearest even on e2m3's grid (step 1/8 below 2, 1/4 below
    4, 1/2 up to 7.5; saturating at 7.5)."""
    xm = x.movedim(dim, -1)
    shp = xm.shape
    n = shp[-1]
    pad = (-n) % 32
    if pad:
        xm = F.pad(xm, (0, pad))
    blk = xm.reshape(*xm.shape[:-1], -1, 32)
    amax = blk.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 2.0)
    v = (blk / scale).clamp(-7.5, 7.5)
    e = torch.floor(torch.log2(v.abs().clamp_min(1.0)))
    step = torch.exp2(e - 3.0)
    q = torch.round(v / step) * step                      # (torch.round: half to even)
    q = q.clamp(-7.5, 7.5) * scale
    q = q.reshape(*xm.shape)[..., :n].reshape(shp)
    return q.movedim(-1, dim)


KDIMS = (1, 1)        # contraction axis of (a, b): channels of x and Cin of w forward; channels of gy and Cout of w backward


def product(conv, a, b, sa, sb, kdims=(1, 1)):
    """conv(a_operand, b_operand) under MODE; a scaled by sa, b by sb (powers of two)."""
    global KDIMS
    KDIMS = kdims
    if MODE == "exact":
        return conv(a, b)
    ah, al = f16_split(a * sa)
    bh, bl = f16_split(b * sb)
    if MODE == "h3":
        r = conv(ah, bh) + conv(ah, bl) + conv(al, bh)
    elif MODE == "f8x":
        r = conv(ah, bh) + conv(q8(ah), q8_lo(bl)) + conv(q8_lo(al), q8(bh))
    elif MODE == "f6x":
        ka, kb = KDIMS
        r = conv(ah, bh) + conv(q6_blocks(ah, ka), q6_blocks(bl, kb)) + conv(q6_blocks(al, ka), q6_blocks(bh, kb))
    elif MODE == "2pa":
        r = conv(ah, bh) + conv(al, bh)
    elif MODE == "2pb":
        r = conv(ah, bh) + conv(ah, bl)
    elif MODE == "1p":
        r = conv(ah, bh)
    else:
        raise ValueError(MODE)
    return r / (sa * sb)


class EmuConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, dil, pad):
        ctx.save_for_backward(x, w)
        ctx.dp = (dil, pad)
        return product(lambda a, b: _real_conv1d(a, b, None, 1, pad, dil), x, w, 1.0, W_SCALE)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        dil, pad = ctx.dp
        if GRAD_S[0] is None:
            amax = float(gy.abs().max())
            GRAD_S[0] = 2.0 ** np.floor(np.log2(16.0 / amax)) if amax > 0 else 1.0
        gx = product(lambda a, b: F.conv_transpose1d(a, b, None, 1, pad, 0, 1, dil), gy, w, GRAD_S[0], W_SCALE, (1, 0))
        gw = torch.nn.grad.conv1d_weight(x, w.shape, gy, 1, pad, dil)
        return gx, gw, None, None


def emu_conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if MODE == "exact" or w.shape[0] < 128 and w.shape[1] < 128:     # the 160-wide invertible 1x1 stays fp32 in the product
        return _real_conv1d(x, w, b, stride, padding, dilation, groups)
    y = EmuConv.apply(x, w, dilation if isinstance(dilation, int) else dilation[0],
                      padding if isinstance(padding, int) else padding[0])
    return y if b is None else y + b[None, :, None]


_real_conv1d = F.conv1d


def run(tag):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"decoder_{tag}.npz")))
    kw = {k[4:]: (v.item() if v.shape == () else v) for k, v in g.items() if k.startswith("cfg.")}
    cfg = O.DecoderConfig(**kw)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in O.procedural_decoder_state(O.decoder_state_shapes(cfg)).items()}
    b = {k: torch.from_numpy(v) for k, v in O.synthetic_batch(int(g["B"]), int(g["T"]), cfg, 1234, bool(g["ragged"])).items()}
    res = {}
    for mode in ("exact", "h3", "f8x", "f6x", "2pa", "2pb", "1p"):
        global MODE
        MODE = mode
        GRAD_S[0] = None
        p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k
                 and not k.endswith((".p", "lower_diag", "input_mean")) else v) for k, v in sd.items()}
        mel = b["mel"].clone().requires_grad_(True)
        O.F.conv1d = emu_conv1d
        try:
            out = O.decoder_forward(p, cfg, mel, b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"])
            lm, _ = O.decoder_loss(out, b["lengths"], cfg.n_group_size)
            lm.backward()
        finally:
            O.F.conv1d = _real_conv1d
        res[mode] = dict(z=out["z_mel"].detach(), loss=float(lm), gmel=mel.grad.clone(),
                         gn={k: float(v.grad.norm()) for k, v in p.items() if v.requires_grad and v.grad is not None})
    ul = b["lengths"] // cfg.n_group_size
    m = (torch.arange(res["exact"]["z"].shape[2])[None] < ul[:, None])[:, None]
    ex = res["exact"]
    rows = []
    for mode, units in (("h3", 3.0), ("f8x", 2.0), ("f6x", 1.5), ("2pa", 2.0), ("2pb", 2.0), ("1p", 1.0)):
        r = res[mode]
        zerr = float(((r["z"] - ex["z"]) * m).abs().max() / (ex["z"] * m).abs().max())
        lerr = abs(r["loss"] - ex["loss"]) / abs(ex["loss"])
        gmel = float((r["gmel"] - ex["gmel"]).abs().max() / ex["gmel"].abs().max())
        gn = max(abs(r["gn"][k] - ex["gn"][k]) / (ex["gn"][k] + 1e-12) for k in ex["gn"] if ex["gn"][k] > 1e-7)
        rows.append({"scheme": mode, "mfma_units": units, "z_max_rel": zerr, "loss_rel": lerr, "grad_mel_max_rel": gmel,
                     "worst_param_gradnorm_rel": gn})
        print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="cfg2_small")
    a = ap.parse_args()
    torch.set_num_threads(8)
    run(a.tag)
