"""Accuracy of the split-fp16 (hi/lo) GEMM on the f16 matrix cores (radmmm_rowgemm_h3) vs fp64 and vs the
fp32-MFMA kernel, so the numbers in DESIGN.md §4.2 are reproducible."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _split(x, scale):
    from rad_mmm_amd._lib import lib, check, ptr, stream
    rows, cols = x.shape
    ldh = (cols + 7) // 8 * 8
    hi = torch.empty(rows, ldh, dtype=torch.float16, device=DEV)
    lo = torch.empty(rows, ldh, dtype=torch.float16, device=DEV)
    check(lib.radmmm_split_f16(ptr(x), x.shape[1], ptr(hi), ptr(lo), ldh, rows, cols, scale, None, stream()), "split")
    return hi, lo


@pytest.mark.parametrize("M,N,K,wscale", [(256, 128, 64, 256.0), (300, 200, 1024, 256.0), (1280, 1024, 5120, 256.0)])
def test_h3gemm_accuracy(M, N, K, wscale):
    from rad_mmm_amd._lib import rowgemm, rowgemm_h3
    g = torch.Generator().manual_seed(K)
    A = torch.nn.functional.softplus(torch.randn(M, K, generator=g) * 2).to(DEV)       # activations-like
    Bw = (torch.randn(N, K, generator=g) * 0.03).to(DEV)                                # weights-like
    Ah, Al = _split(A, 1.0)
    Bh, Bl = _split(Bw, wscale)
    assert rel_err((Ah.float() + Al.float())[:, :K].cpu(), A.cpu()) < 1e-6
    C = torch.full((M, N), float("nan"), device=DEV)
    rowgemm_h3(Ah=Ah, Al=Al, lda_h=Ah.shape[1], Bh=Bh, Bl=Bl, ldb_h=Bh.shape[1], acc_scale=1.0 / wscale, C=C, ldc=N, M=M, N=N,
               K=K, T=M)
    ref = A.double().cpu() @ Bw.double().cpu().t()
    e_h3 = rel_err(C.cpu().double(), ref)
    C32 = torch.empty(M, N, device=DEV)
    rowgemm(A=A, lda=K, B=Bw, ldb=K, b_layout=0, C=C32, ldc=N, M=M, N=N, K=K, T=M)
    e_f32 = rel_err(C32.cpu().double(), ref)
    print(f"M={M} N={N} K={K}: split-f16 err {e_h3:.2e}, fp32 MFMA err {e_f32:.2e}")
    assert e_h3 < 5e-6


def _decode_cross(cross, rows, K, role_b, x8_exp):
    """8-bit cross array (stored in a half tensor [rows][ld]) -> (hi8, lo8) as fp32 [rows][K], undoing the exponents."""
    by = cross.view(torch.uint8).reshape(rows, -1)[:, : 2 * K].reshape(rows, K // 32, 64)
    first, second = by[..., :32], by[..., 32:]
    hi8, lo8 = (second, first) if role_b else (first, second)
    dec = lambda u: u.contiguous().view(torch.float8_e4m3fn).float().reshape(rows, K)
    return dec(hi8) / 2.0 ** x8_exp, dec(lo8) / 2.0 ** (11 + x8_exp)


@pytest.mark.parametrize("M,N,K", [(300, 200, 1024), (1280, 1024, 1024), (12800, 1024, 1024)])
def test_f8x_gemm_layout_and_accuracy(M, N, K):
    """nprod = 2: C = Ah.Bh on the f16 cores + (Ah8.Bl8 + Al8.Bh8) as one block-scaled FP8 MFMA.  The
    kernel's result must equal that sum formed on the host from the very arrays the producers wrote (decoded e4m3
    bytes, fp64 accumulation): checks the cross-array layout, the lane/k-block/scale mapping of the instruction and the
    exponents end to end.  Against the exact product the error is a few 1e-5 (the split-f16 x3 scheme: 2e-6)."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    g = torch.Generator().manual_seed(K + M)
    A = torch.nn.functional.softplus(torch.randn(M, K, generator=g) * 2).to(DEV)
    Bw = (torch.randn(N, K, 1, generator=g) * 0.03).to(DEV)
    Ah, Ax = ops.split_f16(A, K, 1.0, K, nprod=2, x8_exp=ops.X8_ACT_EXP)
    Bh, Bx, _ = ops.split_weight(Bw, None, K, nprod=2)
    C = torch.full((M, N), float("nan"), device=DEV)
    rowgemm_h3(nprod=2, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, Ah=Ah, Al=Ax, lda_h=K, Bh=Bh, Bl=Bx, ldb_h=K,
               acc_scale=1.0 / ops.W_SCALE, C=C, ldc=N, M=M, N=N, K=K, T=M)
    a_hi8, a_lo8 = _decode_cross(Ax, M, K, False, ops.X8_ACT_EXP)
    b_hi8, b_lo8 = _decode_cross(Bx[0], N, K, True, ops.X8_W_EXP)
    # the 8-bit images are what they should be: hi8 ~ hi within e4m3's 2^-4, lo8 ~ the fp16 rounding residual
    assert rel_err(a_hi8.cpu(), Ah.float().cpu()) < 0.07
    resid = (A - Ah.float())
    assert float((a_lo8 - resid).abs().max()) < 0.07 * float(resid.abs().max()) + 1e-9
    d = torch.float64
    model = (Ah.to(d) @ Bh[0].to(d).t() + a_hi8.to(d) @ b_lo8.to(d).t() + a_lo8.to(d) @ b_hi8.to(d).t()) / ops.W_SCALE
    exact = A.to(d) @ Bw[:, :, 0].to(d).t()
    e_model = rel_err(C.cpu().double(), model.cpu())
    e_exact = rel_err(C.cpu().double(), exact.cpu())
    print(f"M={M} N={N} K={K}: f8x vs its own operand model {e_model:.2e}, vs the exact product {e_exact:.2e}")
    assert e_model < 3e-6
    assert e_exact < 1e-4


@pytest.mark.parametrize("nprod", [3, 2])
def test_extra_k_segment_sums_two_gemms(nprod):
    """radmmm_rowgemm_h3's extra_tap: acc = sum_taps A1[r + shift] . B[tap] + A2[r] . B[taps] in ONE launch (the data
    gradients of a k-tap conv and of a 1x1 conv that reach the same tensor) against the two launches it replaces, with the
    intermediate tensor passed through `add`.  Same operands, same products: only the fp32 accumulation order differs."""
    from rad_mmm_amd import ops
    from rad_mmm_amd._lib import rowgemm_h3
    B, T, Wc, taps, dil = 4, 96, 256, 5, 2
    N = B * T
    gen = torch.Generator().manual_seed(17 + nprod)
    a1 = (torch.randn(N, Wc, generator=gen) * 3e-3).to(DEV)
    a2 = (torch.randn(N, Wc, generator=gen) * 3e-3).to(DEV)
    w1 = (torch.randn(Wc, Wc, taps, generator=gen) * 0.03).to(DEV)
    w2 = (torch.randn(Wc, Wc, 1, generator=gen) * 0.03).to(DEV)
    lens = torch.tensor([96, 80, 96, 57], dtype=torch.int32, device=DEV)
    S = 2048.0
    pair_h, pair_l = ops._halves(2 * N, Wc, like=a1)
    for src, lo in ((a1, 0), (a2, N)):
        h, l = ops.split_f16(src, Wc, S, Wc, nprod, ops.X8_GRAD_EXP)
        pair_h[lo: lo + N], pair_l[lo: lo + N] = h, l
    W1h, W1l, _ = ops.split_weight(w1, None, Wc, nprod=nprod)            # [taps][co][ci]: used as [tap][n][k] directly
    W2h, W2l, _ = ops.split_weight(w2, None, Wc, nprod=nprod)
    stack_h, stack_l = ops._halves(taps + 1, Wc, Wc, like=a1)
    stack_h[:taps], stack_l[:taps], stack_h[taps:], stack_l[taps:] = W1h, W1l, W2h, W2l
    common = dict(nprod=nprod, a8_exp=ops.X8_GRAD_EXP, b8_exp=ops.X8_W_EXP, acc_scale=1.0 / (S * ops.W_SCALE), ldc=Wc, M=N,
                  N=Wc, K=Wc, T=T, lens=lens, lda_h=Wc, ldb_h=Wc)
    # two launches: k-tap part -> G, then the 1x1 part with add = G
    G = torch.empty(N, Wc, device=DEV)
    rowgemm_h3(Ah=pair_h[:N], Al=pair_l[:N], Bh=W1h, Bl=W1l, b_tap_stride_h=W1h.stride(0), C=G, taps=taps, dil=dil, sign=-1,
               a_mask_mode=0, **common)
    ref = torch.empty(N, Wc, device=DEV)
    rowgemm_h3(Ah=pair_h[N:], Al=pair_l[N:], Bh=W2h, Bl=W2l, C=ref, add=G, ldadd=Wc, rowscale=1, **common)
    # one launch
    got = torch.full((N, Wc), float("nan"), device=DEV)
    rowgemm_h3(Ah=pair_h, Al=pair_l, Bh=stack_h, Bl=stack_l, b_tap_stride_h=stack_h.stride(0), C=got, taps=taps, dil=dil,
               sign=-1, a_mask_mode=0, extra_tap=1, extra_a_rows=N, rowscale=1, **common)
    assert rel_err(got.cpu(), ref.cpu()) < 2e-6
    assert float(got.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Mc,Nc,taps,dil", [(5, 50, 256, 256, 3, 2), (3, 97, 320, 288, 5, 1), (4, 64, 160, 1152, 1, 1),
                                                 (2, 400, 1024, 1024, 5, 8)])
def test_wgrad_rm_matches_the_transposed_path(B, T, Mc, Nc, taps, dil):
    """radmmm_wgrad_rm (row-major split operands, transposition in the LDS read) against radmmm_wgrad_h3 on transposed
    zero-gapped copies of the same tensors: same three f16 products of the same hi/lo values, fp32 accumulation in a
    different order -> agreement to fp32 rounding of the sums; and against a float64 convolution-style reference with
    utterance boundaries (no cross-utterance products), partial tiles and odd shifts."""
    from rad_mmm_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    gy = (torch.randn(B * T, Mc, generator=g) * 0.3).to(DEV)
    x = torch.nn.functional.softplus(torch.randn(B * T, Nc, generator=g)).to(DEV)
    SG = 4.0
    ldg, ldx = ops.round_up(Mc, 8), ops.round_up(Nc, 8)
    gh, gl = ops.split_f16(gy, Mc, SG, ldg)
    xh, xl = ops.split_f16(x, Nc, 1.0, ldx)
    P = ops.wgrad_rm_slabs((gh, gl), (xh, xl), B, T, Mc, Nc, taps, dil, 1.0 / SG).sum(0)      # [taps, Mc, Nc]
    # the length mask on x (partial padding): equal to the unmasked gradient of x zeroed at frames >= length
    lens = torch.tensor([max(1, T - 7 * b) for b in range(B)], dtype=torch.int32, device=DEV)
    keep = (torch.arange(T, device=DEV)[None] < lens[:, None]).reshape(B * T, 1)
    xmh, xml = ops.split_f16(x * keep, Nc, 1.0, ldx)
    Pm = ops.wgrad_rm_slabs((gh, gl), (xh, xl), B, T, Mc, Nc, taps, dil, 1.0 / SG, lens).sum(0)
    Pz = ops.wgrad_rm_slabs((gh, gl), (xmh, xml), B, T, Mc, Nc, taps, dil, 1.0 / SG).sum(0)
    assert torch.equal(Pm, Pz)
    # float64 reference from the split values
    gv = (gh.double() + gl.double())[:, :Mc].view(B, T, Mc) / SG
    xv = (xh.double() + xl.double())[:, :Nc].view(B, T, Nc)
    ref = torch.zeros(taps, Mc, Nc, dtype=torch.float64, device=DEV)
    for tp in range(taps):
        s = (tp - taps // 2) * dil
        lo, hi = max(0, -s), min(T, T - s)
        if hi > lo:
            ref[tp] = torch.einsum("btm,btn->mn", gv[:, lo:hi], xv[:, lo + s:hi + s])
    assert rel_err(P.double().cpu(), ref.cpu()) < 3e-6
    if (taps // 2) * dil <= ops._TS_FRONT:
        gy_t = ops.transpose_split_act(gy, Mc, B, T, None, 0, SG, "gy")
        x_t = ops.transpose_split_act(x, Nc, B, T, None, 0, 1.0, "x", need_odd=(dil % 2 == 1 and taps > 1))
        Q = ops.wgrad_h3_slabs(gy_t, x_t, Mc, Nc, Nc, taps, dil, 1.0 / SG).sum(0)
        assert rel_err(P.cpu(), Q.cpu()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Mc,Nc,taps,dil", [(4, 64, 256, 256, 1, 1), (3, 96, 320, 288, 5, 2), (2, 160, 1024, 1152, 1, 1),
                                                (5, 40, 96, 160, 3, 1), (2, 352, 512, 256, 5, 8)])
def test_wgrad_rm8_fp8_cross_terms(B, T, Mc, Nc, taps, dil):
    """radmmm_wgrad_rm8 (hi.hi on the f16 pipe, both cross terms in one block-scaled FP8 MFMA whose hi8 halves are derived
    from the fp16 fragments in registers, lo8 halves read from the cross arrays through ds_read_b64_tr_b8) against a float64
    reference built from the exact fp32 tensors: utterance boundaries, partial tiles, odd shifts, the length mask; and its
    error against the three-product kernel on the same tensors (the scheme's cross terms are e4m3-rounded: ~2^-4 * 2^-11
    per product, averaged over the frames)."""
    from rad_mmm_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    gy = (torch.randn(B * T, Mc, generator=g) * 0.3).to(DEV)
    x = torch.nn.functional.softplus(torch.randn(B * T, Nc, generator=g)).to(DEV)
    SG = 32.0                                                # amax * SG in [8, 16) .. the producers' calibration
    ldg, ldx = ops.round_up(Mc, 32), ops.round_up(Nc, 32)
    gh, gx = ops.split_f16(gy, Mc, SG, ldg, 2, ops.X8_GRAD_EXP)
    xh, xx = ops.split_f16(x, Nc, 1.0, ldx, 2, ops.X8_ACT_EXP)
    P = ops.wgrad_rm8_slabs((gh, gx), ops.X8_GRAD_EXP, (xh, xx), ops.X8_ACT_EXP, B, T, Mc, Nc, taps, dil, 1.0 / SG).sum(0)
    lens = torch.tensor([max(1, T - 7 * b) for b in range(B)], dtype=torch.int32, device=DEV)
    keep = (torch.arange(T, device=DEV)[None] < lens[:, None]).reshape(B * T, 1)
    xmh, xmx = ops.split_f16(x * keep, Nc, 1.0, ldx, 2, ops.X8_ACT_EXP)
    Pm = ops.wgrad_rm8_slabs((gh, gx), ops.X8_GRAD_EXP, (xh, xx), ops.X8_ACT_EXP, B, T, Mc, Nc, taps, dil, 1.0 / SG, lens).sum(0)
    Pz = ops.wgrad_rm8_slabs((gh, gx), ops.X8_GRAD_EXP, (xmh, xmx), ops.X8_ACT_EXP, B, T, Mc, Nc, taps, dil, 1.0 / SG).sum(0)
    assert torch.equal(Pm, Pz)
    gv = gy.double().view(B, T, Mc)
    xv = x.double().view(B, T, Nc)
    ref = torch.zeros(taps, Mc, Nc, dtype=torch.float64, device=DEV)
    for tp in range(taps):
        s = (tp - taps // 2) * dil
        lo, hi = max(0, -s), min(T, T - s)
        if hi > lo:
            ref[tp] = torch.einsum("btm,btn->mn", gv[:, lo:hi], xv[:, lo + s:hi + s])
    err8 = rel_err(P.double().cpu(), ref.cpu())
    g3h, g3l = ops.split_f16(gy, Mc, SG, ldg)
    x3h, x3l = ops.split_f16(x, Nc, 1.0, ldx)
    P3 = ops.wgrad_rm_slabs((g3h, g3l), (x3h, x3l), B, T, Mc, Nc, taps, dil, 1.0 / SG).sum(0)
    err3 = rel_err(P3.double().cpu(), ref.cpu())
    print(f"wgrad_rm8 B={B} T={T} {Mc}x{Nc} taps={taps}: max rel err fp8-cross {err8:.2e}, three products {err3:.2e}")
    assert err8 < 6e-5 and err3 < 3e-6            # (a few hundred frames here: the e4m3 rounding of the cross terms averages down with the frame count)
