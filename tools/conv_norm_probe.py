#!/usr/bin/env python3
"""ops.conv_norm (weight-normed partial conv, k = 5) forward + backward against torch autograd on the oracle's
partial_conv1d, over dilation and utterance length, on the fp32 and the split-f16 kernels."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RADMMM_DEBUG"] = "1"
os.environ["RADMMM_PRECISION"] = "h3"


def main():
    from oracle import radmmm_oracle as O
    from rad_mmm_amd import ops
    dev = torch.device("cuda:0")
    C = 512
    for (B, Tn, dil) in ((2, 32, 8), (2, 250, 4), (2, 250, 8), (2, 256, 8), (8, 250, 8), (2, 100, 8), (2, 64, 8), (2, 250, 16)):
        g = torch.Generator().manual_seed(5)
        lens = torch.tensor(sorted([int(Tn * (0.6 + 0.4 * i / max(1, B - 1))) for i in range(B)], reverse=True))
        mask = (torch.arange(Tn)[None] < lens[:, None]).float()[:, None]
        x = torch.randn(B, C, Tn, generator=g)
        v = torch.randn(C, C, 5, generator=g) * 0.05
        gg = torch.rand(C, 1, 1, generator=g) + 0.5
        b = torch.randn(C, generator=g) * 0.1
        gy = torch.randn(B, C, Tn, generator=g)
        xo, vo, go, bo = (t.clone().requires_grad_(True) for t in (x, v, gg, b))
        w = O.weight_norm_fold(vo, go)
        yo = O.partial_conv1d(xo, mask, w, bo, dil) * mask
        (yo * gy).sum().backward()
        cl = lambda t: t.permute(0, 2, 1).reshape(B * Tn, -1).contiguous()
        for mode, rows in (("fp32", "1000000000"), ("h3", "0")):
            os.environ["RADMMM_CONVNORM_H3_MIN_ROWS"] = rows
            xs = cl(x).to(dev).requires_grad_(True)
            vs, gs, bs = (t.clone().to(dev).requires_grad_(True) for t in (v, gg, b))
            y = ops.conv_norm(xs, vs, gs, bs, lens.to(torch.int32).to(dev), B, Tn, dil=dil, partial=True, mask_out=True, act="none")
            (y * cl(gy).to(dev)).sum().backward()
            r = lambda a, c: float((a.cpu() - c).abs().max() / c.abs().max())
            print(f"B={B} T'={Tn:3d} dil={dil:2d} {mode:4s}: y {r(y.detach(), cl(yo.detach())):.1e}  gx {r(xs.grad, cl(xo.grad)):.1e}  gv {r(vs.grad, vo.grad):.1e}  "
                  f"gg {r(gs.grad, go.grad):.1e}  gb {r(bs.grad, bo.grad):.1e}")


if __name__ == "__main__":
    main()
