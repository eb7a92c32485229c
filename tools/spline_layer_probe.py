#!/usr/bin/env python3
"""One spline coupling layer (FiLM predictor + masked batch-norm + piecewise-quadratic spline) at the decoder's real
dimensions, HIP against the CPU oracle's autograd, at a chosen batch shape: which gradient leaves the 5e-4 band first,
and through which output (z or log_s)?

    RADMMM_DEBUG=1 python tools/spline_layer_probe.py --batch 8 --frames 250 [--bn 0] [--film h3]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=250)       # grouped frames T'
    ap.add_argument("--bn", type=int, default=1)
    ap.add_argument("--film", default="fp32")
    ap.add_argument("--ctx", type=int, default=1056)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--fixed", type=int, default=0, help="1: all utterances full length")
    ap.add_argument("--dilation", type=int, default=1)
    args = ap.parse_args()
    os.environ["RADMMM_DEBUG"] = "1"
    os.environ["RADMMM_PRECISION"] = "h3"
    os.environ["RADMMM_CONVNORM_H3_MIN_ROWS"] = "0" if args.film == "h3" else "1000000000"
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import spline_layers as SL
    from rad_mmm_amd.spline_layers import SplineTransformationLayer
    from rad_mmm_amd.ops import ZLD
    stash = {}
    _bwd = SL.PQSplineFn.backward

    def bwd(ctx_, gy, glj):
        r = _bwd(ctx_, gy, glj)
        stash["gq_hip"], stash["gx_hip"] = r[1].detach().cpu(), r[0].detach().cpu()
        stash["q_hip"] = ctx_.saved_tensors[1].detach().cpu()
        return r
    SL.PQSplineFn.backward = staticmethod(bwd)
    _film = O.film_stack_forward

    def film(*a, **k):
        q = _film(*a, **k)
        q.retain_grad()
        stash["q_or"] = q
        return q
    O.film_stack_forward = film
    dev = torch.device("cuda:0")
    C, D, B, Tn = 160, args.ctx, args.batch, args.frames
    layer = SplineTransformationLayer(C, D, args.layers, scaling_fn="tanh", top=3, bottom=-3, left=-3, right=3, n_bins=32,
                                      use_quadratic=True, use_bn=bool(args.bn), with_dilation=bool(args.dilation))
    shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in O.procedural_decoder_state(shapes, end_scale=0.05).items()}
    layer.load_state_dict(sd)
    layer = layer.to(dev).train()
    g = torch.Generator().manual_seed(11)
    z = torch.randn(B, C, Tn, generator=g) * 1.2
    ctx = torch.randn(B, D, Tn, generator=g) * 0.5
    lens = torch.full((B,), Tn, dtype=torch.long) if args.fixed else torch.tensor(sorted([int(Tn * (0.6 + 0.4 * i / max(1, B - 1))) for i in range(B)], reverse=True))
    mask = (torch.arange(Tn)[None] < lens[:, None]).float()
    wz = torch.randn(B, C, Tn, generator=g)                   # upstream gradients
    wl = torch.randn(B, 1, Tn, generator=g)
    for which in ("z", "log_s", "both"):
        layer.zero_grad()
        zcl = F.pad(z.permute(0, 2, 1).reshape(B * Tn, C), (0, ZLD - C)).contiguous().to(dev).requires_grad_(True)
        ccl = ctx.permute(0, 2, 1).reshape(B * Tn, -1).contiguous().to(dev).requires_grad_(True)
        W_eff, b_eff = torch.eye(ZLD, device=dev), torch.zeros(ZLD, device=dev)
        zo, log_s = layer.run(zcl, ccl, lens.to(torch.int32).to(dev), W_eff, b_eff, B, Tn, int(lens.sum()))
        m_d = mask.reshape(B * Tn, 1).to(dev)
        wz_d = wz.permute(0, 2, 1).reshape(B * Tn, C).to(dev)
        wl_d = wl.permute(0, 2, 1).reshape(B * Tn, 1).to(dev)
        s = 0.0
        if which in ("z", "both"):
            s = s + (zo[:, :C] * wz_d * m_d).sum()
        if which in ("log_s", "both"):
            s = s + (log_s.reshape(B * Tn, 1) * wl_d * m_d).sum()
        s.backward()
        torch.cuda.synchronize()
        if which == "both":                                   # determinism: the same pass again, bit for bit?
            g1 = {n: q.grad.clone() for n, q in layer.named_parameters()}
            gz1, gc1 = zcl.grad.clone(), ccl.grad.clone()
            layer.zero_grad()
            zcl.grad = None
            ccl.grad = None
            zo2, ls2 = layer.run(zcl, ccl, lens.to(torch.int32).to(dev), W_eff, b_eff, B, Tn, int(lens.sum()))
            ((zo2[:, :C] * wz_d * m_d).sum() + (ls2.reshape(B * Tn, 1) * wl_d * m_d).sum()).backward()
            torch.cuda.synchronize()
            nd = sum(int(not torch.equal(g1[n], q.grad)) for n, q in layer.named_parameters())
            print(f"      second identical pass: {nd} parameter gradients differ bitwise; gz equal {torch.equal(gz1, zcl.grad)}, gctx equal {torch.equal(gc1, ccl.grad)}")
        p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v) for k, v in sd.items()}
        oz = z.clone().requires_grad_(True)
        oc = ctx.clone().requires_grad_(True)
        zo_o, ls_o = O.spline_coupling_forward(p, "", oz, oc, mask[:, None], args.layers, use_bn=bool(args.bn), training=True)
        so = 0.0
        if which in ("z", "both"):
            so = so + (zo_o * wz * mask[:, None]).sum()
        if which in ("log_s", "both"):
            so = so + (ls_o * wl * mask[:, None]).sum()
        so.backward()
        mk = mask[:, None]
        fz = float(((zo[:, :C].detach().cpu().reshape(B, Tn, C).permute(0, 2, 1) - zo_o.detach()) * mk).abs().max() / (zo_o.detach() * mk).abs().max())
        fl = float(((log_s.detach().cpu().reshape(B, Tn, 1).permute(0, 2, 1) - ls_o.detach()) * mk).abs().max() / (ls_o.detach() * mk).abs().max())
        gz = zcl.grad[:, :C].cpu().reshape(B, Tn, C).permute(0, 2, 1)
        gc = ccl.grad.cpu().reshape(B, Tn, -1).permute(0, 2, 1)
        dz, dc = gz - oz.grad, gc - oc.grad
        rows = []
        for n, q in layer.named_parameters():
            go = p[n].grad
            if go is None or float(go.abs().max()) < 1e-12:
                continue
            d = q.grad.cpu() - go
            rows.append((float(d.norm() / go.norm()), float(d.abs().max() / go.abs().max()), n))
        rows = [r for r in rows if "hidden_conv.conv.bias" not in r[2] or not args.bn]     # (zero true gradient in front of a batch-norm)
        rows.sort(reverse=True)
        gq_o = stash["q_or"].grad                                  # [B, h*65, T']
        gq_h = stash["gq_hip"][:, : gq_o.shape[1]].reshape(B, Tn, -1).permute(0, 2, 1)
        q_h = stash["q_hip"][:, : gq_o.shape[1]].reshape(B, Tn, -1).permute(0, 2, 1)
        dqq = (q_h - stash["q_or"].detach()) * mask[:, None]
        print(f"      predictor output q: L2 {float(dqq.norm() / (stash['q_or'].detach() * mask[:, None]).norm()):.1e} max {float(dqq.abs().max() / stash['q_or'].detach().abs().max()):.1e}")
        dq = (gq_h - gq_o) * mask[:, None]
        if which == "both" and float(dq.abs().max() / gq_o.abs().max()) > 1e-3:
            # where? (utterance, parameter channel, frame) -> (element channel c = pc // 65, parameter k = pc % 65)
            dd = dq.abs()
            for _ in range(4):
                i = int(dd.argmax())
                bi, pc, ti = i // (dd.shape[1] * Tn), (i // Tn) % dd.shape[1], i % Tn
                c, k = pc // 65, pc % 65
                xv = float((z[bi, 80 + c, ti] + 3) / 6)
                qv = stash["q_or"].detach()[bi, c * 65:(c + 1) * 65, ti]
                w = torch.softmax(qv[:32], -1)
                wc = torch.cumsum(w, -1)
                idx = int(torch.searchsorted(wc, torch.tensor([xv])))
                print(f"        worst gq: utt {bi} frame {ti} (len {int(lens[bi])}) channel {c} param {k} ({'w' if k < 32 else 'v'}{k if k < 32 else k - 32}): hip {float(gq_h[bi, pc, ti]):+.4e} oracle {float(gq_o[bi, pc, ti]):+.4e}; "
                      f"x {xv:.6f} bin {idx} edges {float(wc[idx - 1]) if idx > 0 else 0.0:.6f}..{float(wc[min(idx, 31)]):.6f} w_b {float(w[min(idx, 31)]):.3e} max|q| {float(qv.abs().max()):.2f}")
                dd[bi, c * 65:(c + 1) * 65, ti] = 0
        print(f"      gq (spline backward wrt the predictor's output): L2 {float(dq.norm() / (gq_o * mask[:, None]).norm()):.1e} max {float(dq.abs().max() / gq_o.abs().max()):.1e}")
        print(f"[B={B} T'={Tn} bn={args.bn} film={args.film} fixed={args.fixed}] upstream through {which}: fwd z {fz:.1e} log_s {fl:.1e} | "
              f"gz L2 {float(dz.norm() / oz.grad.norm()):.1e} max {float(dz.abs().max() / oz.grad.abs().max()):.1e} | "
              f"gctx L2 {float(dc.norm() / oc.grad.norm()):.1e} max {float(dc.abs().max() / oc.grad.abs().max()):.1e}")
        for r in rows[:4]:
            print(f"      L2 {r[0]:.1e} max {r[1]:.1e} {r[2]}")


if __name__ == "__main__":
    main()
