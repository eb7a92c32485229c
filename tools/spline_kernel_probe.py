#!/usr/bin/env python3
"""The piecewise-quadratic spline kernels alone at BASELINE configs[4]'s size (32 000 grouped frames x 80 coupled channels x
32 bins: q = 666 MB): the register-resident kernels (csrc/spline.hip, pq_spline_*_reg_kernel<K>) against the runtime-K LDS walk
they replace -- outputs compared element by element, bins counted, launches timed with HIP events, HBM roofline fraction.

    python tools/spline_kernel_probe.py [--rows 32000] [--h 80] [--K 32] [--reps 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RADMMM_DEBUG"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32000)
    ap.add_argument("--h", type=int, default=80)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from rad_mmm_amd._lib import lib, check, ptr, stream
    dev = torch.device("cuda:0")
    rows, h, K = a.rows, a.h, a.K
    nb = 2 * K + 1
    g = torch.Generator().manual_seed(7)
    x = torch.rand(rows, h, generator=g)
    x[::97, ::7] = 1.5
    q = (torch.randn(rows, h * nb, generator=g) * 1.5)
    gy = torch.randn(rows, h, generator=g)
    glj = torch.randn(rows, generator=g)
    xd, qd, gyd, gljd = x.to(dev), q.to(dev), gy.to(dev), glj.to(dev)

    def run(mode):
        if mode == "generic":
            os.environ["RADMMM_SPLINE"] = "generic"
        else:
            os.environ.pop("RADMMM_SPLINE", None)
        y = torch.empty(rows, h, device=dev)
        lj = torch.empty(rows + rows * h, device=dev)
        gx = torch.empty(rows, h, device=dev)
        gq = torch.empty_like(qd)
        bins = torch.empty(rows * h, dtype=torch.int32, device=dev)
        el, er = torch.empty(rows * h, device=dev), torch.empty(rows * h, device=dev)
        fwd = lambda: check(lib.radmmm_pq_spline_fwd(ptr(xd), h, ptr(qd), h * nb, ptr(y), h, ptr(lj), rows, h, K, stream()), "fwd")
        bwd = lambda: check(lib.radmmm_pq_spline_bwd(ptr(xd), h, ptr(qd), h * nb, ptr(gyd), h, ptr(gljd), ptr(gx), h, ptr(gq),
                                                     h * nb, rows, h, K, stream()), "bwd")
        check(lib.radmmm_pq_spline_bins(ptr(xd), h, ptr(qd), h * nb, ptr(bins), ptr(el), ptr(er), rows, h, K, stream()), "bins")
        t = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[name] = e0.elapsed_time(e1) / a.reps * 1e3
        return dict(y=y.cpu(), lj=lj.cpu(), gx=gx.cpu(), gq=gq.cpu(), bins=bins.cpu(), el=el.cpu(), er=er.cpu(), t=t)

    ref = run("generic")
    new = run("reg")
    qbytes = rows * h * nb * 4
    out = {"rows": rows, "h": h, "K": K, "q_MB": qbytes / 1e6}
    for k in ("fwd", "bwd"):
        byt = qbytes * (1 if k == "fwd" else 2) + rows * h * 4 * (3 if k == "fwd" else 3)
        out[k] = {"generic_us": round(ref["t"][k], 1), "reg_us": round(new["t"][k], 1),
                  "reg_GBps": round(byt / new["t"][k] / 1e3, 1), "reg_frac_of_8TBps": round(byt / new["t"][k] / 1e3 / 8000, 3),
                  "generic_frac_of_8TBps": round(byt / ref["t"][k] / 1e3 / 8000, 3)}
    mism = ref["bins"] != new["bins"]
    out["bins_differ"] = int(mism.sum())
    ok = ~mism.view(rows, h)
    out["y_max_abs"] = float((ref["y"] - new["y"])[ok].abs().max())
    out["logj_elem_max_abs"] = float((ref["lj"][rows:].view(rows, h) - new["lj"][rows:].view(rows, h))[ok].abs().max())
    out["logj_sum_max_abs"] = float((ref["lj"][:rows] - new["lj"][:rows]).abs().max())
    out["gx_max_rel"] = float((ref["gx"] - new["gx"])[ok].abs().max() / ref["gx"].abs().max())
    okq = ok.unsqueeze(-1).expand(rows, h, nb).reshape(rows, h * nb)
    out["gq_max_rel_of_tensor_max"] = float((ref["gq"] - new["gq"])[okq].abs().max() / ref["gq"].abs().max())
    out["edge_l_max_abs"] = float((ref["el"] - new["el"])[~mism].abs().max())
    out["edge_r_max_abs"] = float((ref["er"] - new["er"])[~mism].abs().max())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
