import sys, torch
sys.path.insert(0, '.')
from rad_mmm_amd._lib import rowgemm_h3
from rad_mmm_amd import ops
dev = torch.device("cuda:0")
N, T = 12800, 400
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 1024, generator=g).to(dev)
v = (torch.randn(1024, 1024, 5, generator=g) * 0.02).to(dev)
gg = torch.ones(1024, 1, 1, device=dev); b = torch.zeros(1024, device=dev)
xh, xl = ops.split_f16(x, 1024, 1.0); Wh, Wl, _ = ops.split_weight(v, gg, 1024)
y = torch.empty(N, 1024, device=dev); yh, yl = torch.empty_like(xh), torch.empty_like(xl)
lens = torch.full((N // T,), T, dtype=torch.int32, device=dev)
def run(split, act, pconv, taps):
    kw = dict(Ah=xh, Al=xl, lda_h=1024, Bh=Wh, Bl=Wl, ldb_h=1024, b_tap_stride_h=Wh.stride(0), acc_scale=1/256., C=y, ldc=1024,
              M=N, N=1024, K=1024, taps=taps, dil=2, sign=1, T=T, lens=lens, a_mask_mode=1, bias=b, pconv=pconv, ratio_taps=5,
              ratio_dil=2, postmask=1, act=act)
    if split: kw.update(Ch=yh, Cl=yl, ldch=1024, ch_scale=1.0)
    for _ in range(3): rowgemm_h3(**kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): rowgemm_h3(**kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for taps in (5, 1):
    for split, act, pconv in ((1,1,1),(0,1,1),(1,0,1),(0,0,0)):
        print(f"taps={taps} split={split} act={act} pconv={pconv}: {run(split, act, pconv, taps):.4f} ms")
