import os, sys
os.environ.setdefault("RADMMM_DEBUG", "1")
os.environ["RADMMM_H3W_MB"] = "7"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rad_mmm_amd import ops
from rad_mmm_amd._lib import rowgemm_h3
DEV = torch.device("cuda:0")
W, B, T = 512, 3, 300
N = B * T
gen = torch.Generator().manual_seed(14)
Hs = torch.nn.functional.softplus(torch.randn(N, W, generator=gen) * 2).to(DEV)
gy = (torch.randn(2 * N, W, generator=gen) * 3e-3).to(DEV)
w = (torch.randn(W, W, 2, generator=gen) * 0.03).to(DEV)
lens = torch.tensor([T, T - 37, T // 2 + 5], dtype=torch.int32, device=DEV)
S, GE = 2048.0, ops.X8_GRAD_EXP
Ah, Al = ops.split_f16(gy, W, S, W, 2, GE)
Wh, Wl, _ = ops.split_weight(w, None, W, nprod=2)
Hh, Hl = ops.split_f16(Hs, W, 1.0, W, 2, ops.X8_ACT_EXP)
flag = torch.zeros(1, dtype=torch.int32, device=DEV)
base = dict(nprod=2, a8_exp=GE, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / (S * ops.W_SCALE), T=T, sat_flag=flag, Ah=Ah, Al=Al,
            lda_h=W, Bh=Wh, Bl=Wl, ldb_h=W, b_tap_stride_h=Wh.stride(0), ldc=W, M=N, N=W, K=W, taps=1, dil=1,
            sign=-1, lens=lens, dact=1, rowscale=2, ratio_taps=5, ratio_dil=2, ldch=W, ch_scale=S, split_fmt=ops.SPLIT_X8A, ch_x8_exp=GE)
def run(pair):
    C = torch.full((N, W), float("nan"), device=DEV)
    Ch, Cl = ops._halves(N, W, like=Hs)
    src = dict(dact_h=Hh, dact_x=Hl, lddact_h=W, dact_x8_exp=ops.X8_ACT_EXP) if pair else dict(dact_src=Hs, lddact=W)
    rowgemm_h3(C=C, Ch=Ch, Cl=Cl, **src, **base)
    torch.cuda.synchronize()
    return C
c0, c1 = run(False), run(True)
bad = torch.isnan(c1)
print("nan count", int(bad.sum()), "of", c1.numel(), "ref nan", int(torch.isnan(c0).sum()))
idx = bad.nonzero()
print("first nan idx", idx[:10].tolist(), "rows with nan:", sorted(set((idx[:, 0] // 1).tolist()))[:20], "cols:", sorted(set(idx[:, 1].tolist()))[:40])
# host-side decode of the pair
lo_bytes = Hl.view(torch.uint8).view(N, W // 32, 64)[:, :, 32:].reshape(N, W)
lo = lo_bytes.view(torch.float8_e4m3fn).float()
yp = Hh.float() + lo * 2.0 ** -(11 + ops.X8_ACT_EXP)
print("host decode: max rel err of pair vs fp32", float(((yp - Hs).abs() / Hs.abs().clamp_min(1e-6)).max()), "nan in lo", int(torch.isnan(lo).sum()))
ok = ~bad
print("max diff where finite", float((c1 - c0)[ok].abs().max()), "scale", float(c0.abs().max()))
r = idx[0, 0].item() if len(idx) else 0
print("row", r, "frame", r % T, "len", lens[r // T].item(), "Hs row sample", Hs[r, idx[0, 1]].item() if len(idx) else None)
