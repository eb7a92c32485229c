"""Is the dominant split-f16 GEMM power bound?  Same launch (in_layer forward conv, M=12800) on
random operands, on zeros, and with only the hi halves non-zero: identical instruction stream, so any
time difference is the clock the power manager lets the chip hold for that operand activity."""
import sys, torch
sys.path.insert(0, '.')
from rad_mmm_amd._lib import rowgemm_h3
from rad_mmm_amd import ops
N, T = 12800, 400
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(N, 1024, generator=g).to(dev)
v = (torch.randn(1024, 1024, 5, generator=g) * 0.02).to(dev)
gg = torch.ones(1024, 1, 1, device=dev); b = torch.zeros(1024, device=dev)
xh, xl = ops.split_f16(x, 1024, 1.0)
Wh, Wl, _ = ops.split_weight(v, gg, 1024)
y = torch.empty(N, 1024, device=dev); yh, yl = torch.empty_like(xh), torch.empty_like(xl)
lens = torch.full((N // T,), T, dtype=torch.int32, device=dev)
def t(xh, xl, Wh, Wl, reps=60):
    def launch():
        rowgemm_h3(Ah=xh, Al=xl, lda_h=1024, Bh=Wh, Bl=Wl, ldb_h=1024, b_tap_stride_h=Wh.stride(0),
                   acc_scale=1.0 / ops.W_SCALE, C=y, ldc=1024, M=N, N=1024, K=1024, taps=5, dil=2, sign=1, T=T, lens=lens,
                   a_mask_mode=1, bias=b, pconv=1, ratio_taps=5, ratio_dil=2, postmask=1, act=1, Ch=yh, Cl=yl, ldch=1024,
                   ch_scale=1.0)
    for _ in range(10): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
z = lambda a: torch.zeros_like(a)
fl = 3 * 2.0 * N * 1024 * 5 * 1024
def trunc(a, m):
    return (a.view(torch.int16) & torch.tensor(-(1 << (10 - m)), dtype=torch.int16, device=a.device)).view(torch.float16)
extra = tuple((f"lo {m}b", (xh, trunc(xl, m), Wh, trunc(Wl, m))) for m in (6, 4, 2, 0))
for name, args in extra + (("random", (xh, xl, Wh, Wl)), ("zeros", (z(xh), z(xl), z(Wh), z(Wl))), ("hi only", (xh, z(xl), Wh, z(Wl))),
                   ("A zero", (z(xh), z(xl), Wh, Wl)), ("random", (xh, xl, Wh, Wl))):
    ms = t(*args)
    print(f"{name:8s} {ms*1e3:7.1f} us  {fl/ms*1e-9:7.0f} TF executed")
