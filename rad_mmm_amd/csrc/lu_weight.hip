// Channel-mix matrix of the LUS invertible 1x1 conv (reference common.py:507-548):
//   W = P (L U),  L = tril(lower, -1) + diag(lower_diag),  U = triu(upper, 1) + diag(upper_diag),
//   log|det W| = sum log|upper_diag|.
// The reference (and a torch restatement) spends ~16 tiny launches per flow on this in the forward and
// as many in the backward (triu/tril/diag_embed/add/mm/pad/abs/log/sum at 5-30 us each); here it is one
// launch each way, one workgroup per row, and the forward writes the zero-padded [ldw][ldw] matrix the
// flow step's first GEMM consumes (input columns [off, off + c) -> an early exit is a column offset).
// All matrices row-major fp32 [c][c]; c <= 256.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float lower_at(const float* __restrict__ lower, const float* __restrict__ ldiag, int c,
                                          int k, int j) {   // L[k][j], k >= j
  return k == j ? ldiag[k] : lower[k * c + j];
}
__device__ __forceinline__ float upper_at(const float* __restrict__ upper, const float* __restrict__ udiag, int c,
                                          int j, int n) {   // U[j][n], j <= n
  return j == n ? udiag[n] : upper[j * c + n];
}

__global__ __launch_bounds__(NT) void lu_weight_fwd_kernel(const float* __restrict__ P, const float* __restrict__ lower,
                                                           const float* __restrict__ ldiag, const float* __restrict__ upper,
                                                           const float* __restrict__ udiag, int c, float* __restrict__ W,
                                                           int ldw, int off, float* __restrict__ logdet) {
  __shared__ float q[NT], w[NT];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r < c) {
    // q[j] = sum_k P[r][k] L[k][j]
    if (tid < c) {
      float s = 0.f;
      for (int k = tid; k < c; ++k) s = fmaf(P[r * c + k], lower_at(lower, ldiag, c, k, tid), s);
      q[tid] = s;
    }
    __syncthreads();
    // W[r][n] = sum_j q[j] U[j][n]
    if (tid < c) {
      float s = 0.f;
      for (int j = 0; j <= tid; ++j) s = fmaf(q[j], upper_at(upper, udiag, c, j, tid), s);
      w[tid] = s;
    }
    __syncthreads();
  }
  for (int col = tid; col < ldw; col += NT)
    W[(long long)r * ldw + col] = (r < c && col >= off && col < off + c) ? w[col - off] : 0.f;
  if (r == 0 && logdet) {
    __syncthreads();
    q[tid] = tid < c ? logf(fabsf(udiag[tid])) : 0.f;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
      if (tid < s) q[tid] += q[tid + s];
      __syncthreads();
    }
    if (tid == 0) *logdet = q[0];
  }
}

// gW: the [c][c] block of the [ldw][ldw] gradient at columns [off, off + c).
//   gU = (P L)^T gW          -> g_upper = triu(gU, 1), g_udiag = diag(gU) + g_logdet / upper_diag
//   gL = (P^T gW) U^T        -> g_lower = tril(gL, -1)
// Workgroup r produces row r of both.
__global__ __launch_bounds__(NT) void lu_weight_bwd_kernel(const float* __restrict__ P, const float* __restrict__ lower,
                                                           const float* __restrict__ ldiag, const float* __restrict__ upper,
                                                           const float* __restrict__ udiag, int c,
                                                           const float* __restrict__ gW, int ldw, int off,
                                                           const float* __restrict__ g_logdet, float* __restrict__ g_lower,
                                                           float* __restrict__ g_upper, float* __restrict__ g_udiag) {
  __shared__ float qc[NT], gm[NT];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (tid < c) {
    // qc[i] = (P L)[i][r] = sum_{k >= r} P[i][k] L[k][r]
    float s = 0.f;
    for (int k = r; k < c; ++k) s = fmaf(P[tid * c + k], lower_at(lower, ldiag, c, k, r), s);
    qc[tid] = s;
    // gm[n] = (P^T gW)[r][n] = sum_i P[i][r] gW[i][n]
    float m = 0.f;
    for (int i = 0; i < c; ++i) m = fmaf(P[i * c + r], gW[(long long)i * ldw + off + tid], m);
    gm[tid] = m;
  }
  __syncthreads();
  if (tid < c) {
    // row r of gU
    float u = 0.f;
    if (tid >= r)
      for (int i = 0; i < c; ++i) u = fmaf(qc[i], gW[(long long)i * ldw + off + tid], u);
    g_upper[r * c + tid] = tid > r ? u : 0.f;
    if (tid == r) g_udiag[r] = u + (g_logdet ? *g_logdet / udiag[r] : 0.f);
    // row r of gL: gL[r][j] = sum_{n >= j} gm[n] U[j][n]
    float l = 0.f;
    if (tid < r)
      for (int n = tid; n < c; ++n) l = fmaf(gm[n], upper_at(upper, udiag, c, tid, n), l);
    g_lower[r * c + tid] = l;
  }
}

}  // namespace

extern "C" int radmmm_lu_weight_fwd(const float* P, const float* lower, const float* lower_diag, const float* upper,
                                    const float* upper_diag, int c, float* W, int ldw, int col_offset, float* logdet,
                                    void* stream) {
  RADMMM_REQUIRE(P && lower && lower_diag && upper && upper_diag && W, "lu_weight_fwd: null pointer");
  RADMMM_REQUIRE(c > 0 && c <= NT && col_offset >= 0 && ldw >= col_offset + c, "lu_weight_fwd: bad dims");
  hipLaunchKernelGGL(lu_weight_fwd_kernel, dim3(ldw), dim3(NT), 0, static_cast<hipStream_t>(stream), P, lower, lower_diag,
                     upper, upper_diag, c, W, ldw, col_offset, logdet);
  return radmmm::check_launch("lu_weight_fwd");
}

extern "C" int radmmm_lu_weight_bwd(const float* P, const float* lower, const float* lower_diag, const float* upper,
                                    const float* upper_diag, int c, const float* gW, int ldw, int col_offset,
                                    const float* g_logdet, float* g_lower, float* g_upper, float* g_upper_diag,
                                    void* stream) {
  RADMMM_REQUIRE(P && lower && lower_diag && upper && upper_diag && gW && g_lower && g_upper && g_upper_diag,
                 "lu_weight_bwd: null pointer");
  RADMMM_REQUIRE(c > 0 && c <= NT && col_offset >= 0 && ldw >= col_offset + c, "lu_weight_bwd: bad dims");
  hipLaunchKernelGGL(lu_weight_bwd_kernel, dim3(c), dim3(NT), 0, static_cast<hipStream_t>(stream), P, lower, lower_diag,
                     upper, upper_diag, c, gW, ldw, col_offset, g_logdet, g_lower, g_upper, g_upper_diag);
  return radmmm::check_launch("lu_weight_bwd");
}
