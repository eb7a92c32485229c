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
    check(lib.radmmm_split_f16(ptr(x), x.shape[1], ptr(hi), ptr(lo), ldh, rows, cols, scale, stream()), "split")
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
