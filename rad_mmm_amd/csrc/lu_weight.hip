// Channel-mix matrix of the LUS invertible 1x1 conv (reference common.py:507-548):
//   W = P (L U),  L = tril(lower, -1) + diag(lower_diag),  U = triu(upper, 1) + diag(upper_diag),
//   log|det W| = sum log|upper_diag|.
// The reference (and a torch restatement) spends ~16 tiny launches per flow on this in the forward and
// as many in the backward (triu/tril/diag_embed/add/mm/pad/abs/log/sum at 5-30 us each); here it is one
// launch each way (~5 us), one workgroup per row, and the forward writes the zero-padded [ldw][ldw] matrix the
// flow step's first GEMM consumes (input columns [off, off + c) -> an early exit is a column offset).
// All matrices row-major fp32 [c][c]; c <= 256.
#include "common.h"

namespace {

constexpr int NT = 1024, CP = 256, NPART = NT / CP, NWAVE = NT / 64;   // c <= CP; sums are split NPART ways over the threads

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The loops below are short (c = 160) but every iteration is a global load: what matters is how many
// are in flight.  Each sum is therefore split over NPART thread groups (or one wave per row with the
// lanes along the contiguous index), with unconditional loads (masks applied to the values) so the
// compiler can issue a whole unrolled group before the first use.
__global__ __launch_bounds__(NT) void lu_weight_fwd_kernel(const float* __restrict__ P, const float* __restrict__ lower,
                                                           const float* __restrict__ ldiag, const float* __restrict__ upper,
                                                           const float* __restrict__ udiag, int c, float* __restrict__ W,
                                                           int ldw, int off, float* __restrict__ logdet) {
  __shared__ float prow[CP], q[CP], w[CP], part[NPART][CP];
  const int r = blockIdx.x, tid = threadIdx.x, j = tid & (CP - 1), pt = tid / CP;
  const int kc = (c + NPART - 1) / NPART, k0 = pt * kc, k1 = min(c, k0 + kc);
  if (r < c) {
    if (tid < c) prow[tid] = P[r * c + tid];
    __syncthreads();
    // q[j] = sum_k P[r][k] L[k][j],  L[k][j] = lower[k][j] (k > j), ldiag[k] (k == j), 0 (k < j)
    float s = 0.f;
    if (j < c) {
#pragma unroll 8
      for (int k = k0; k < k1; ++k) {
        const float lv = lower[k * c + j];
        s = fmaf(prow[k], k > j ? lv : 0.f, s);
      }
      if (j >= k0 && j < k1) s = fmaf(prow[j], ldiag[j], s);
    }
    part[pt][j] = s;
    __syncthreads();
    if (tid < c) q[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    // W[r][n] = sum_j q[j] U[j][n],  U[j][n] = upper[j][n] (j < n), udiag[n] (j == n), 0 (j > n)
    s = 0.f;
    if (j < c) {
#pragma unroll 8
      for (int k = k0; k < k1; ++k) {
        const float uv = upper[k * c + j];
        s = fmaf(q[k], k < j ? uv : 0.f, s);
      }
      if (j >= k0 && j < k1) s = fmaf(q[j], udiag[j], s);
    }
    __syncthreads();
    part[pt][j] = s;
    __syncthreads();
    if (tid < c) w[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
  }
  for (int col = tid; col < ldw; col += NT)
    W[(long long)r * ldw + col] = (r < c && col >= off && col < off + c) ? w[col - off] : 0.f;
  if (r == 0 && logdet) {
    __syncthreads();
    if (tid < CP) q[tid] = tid < c ? logf(fabsf(udiag[tid])) : 0.f;
    __syncthreads();
    for (int st = CP / 2; st > 0; st >>= 1) {
      if (tid < st) q[tid] += q[tid + st];
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
  __shared__ float lcol[CP], pcol[CP], qc[CP], gm[CP], pu[NPART][CP], pm[NPART][CP];
  const int r = blockIdx.x, tid = threadIdx.x, n = tid & (CP - 1), pt = tid / CP;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < c) {
    const float lv = lower[tid * c + r];                       // column r of L and of P (one gather each)
    lcol[tid] = tid > r ? lv : (tid == r ? ldiag[r] : 0.f);
    pcol[tid] = P[tid * c + r];
  }
  __syncthreads();
  // qc[i] = (P L)[i][r] = sum_k P[i][k] L[k][r]: one wave per row i, lanes along k (coalesced)
  for (int i = wave; i < c; i += NWAVE) {
    float s = 0.f;
    for (int k = lane; k < c; k += 64) s = fmaf(P[i * c + k], lcol[k], s);
    s = wave_sum(s);
    if (lane == 0) qc[i] = s;
  }
  __syncthreads();
  // one pass over gW: gU[r][n] = sum_i qc[i] gW[i][n],  gm[n] = (P^T gW)[r][n] = sum_i P[i][r] gW[i][n]
  {
    const int ic = (c + NPART - 1) / NPART, i0 = pt * ic, i1 = min(c, i0 + ic);
    float su = 0.f, sm = 0.f;
    if (n < c) {
#pragma unroll 8
      for (int i = i0; i < i1; ++i) {
        const float g = gW[(long long)i * ldw + off + n];
        su = fmaf(qc[i], g, su);
        sm = fmaf(pcol[i], g, sm);
      }
    }
    pu[pt][n] = su;
    pm[pt][n] = sm;
  }
  __syncthreads();
  if (tid < c) {
    const float u = (pu[0][tid] + pu[1][tid]) + (pu[2][tid] + pu[3][tid]);
    gm[tid] = (pm[0][tid] + pm[1][tid]) + (pm[2][tid] + pm[3][tid]);
    g_upper[r * c + tid] = tid > r ? u : 0.f;
    if (tid == r) g_udiag[r] = u + (g_logdet ? *g_logdet / udiag[r] : 0.f);
  }
  __syncthreads();
  // gL[r][j] = sum_{n >= j} gm[n] U[j][n] for j < r: one wave per j, lanes along n (coalesced rows of upper)
  for (int jj = wave; jj < c; jj += NWAVE) {
    float s = 0.f;
    if (jj < r) {
      for (int nn = lane; nn < c; nn += 64) {
        const float uv = upper[jj * c + nn];
        s = fmaf(gm[nn], nn > jj ? uv : (nn == jj ? udiag[jj] : 0.f), s);
      }
      s = wave_sum(s);
    }
    if (lane == 0) g_lower[r * c + jj] = s;
  }
}

}  // namespace

extern "C" int radmmm_lu_weight_fwd(const float* P, const float* lower, const float* lower_diag, const float* upper,
                                    const float* upper_diag, int c, float* W, int ldw, int col_offset, float* logdet,
                                    void* stream) {
  RADMMM_REQUIRE(P && lower && lower_diag && upper && upper_diag && W, "lu_weight_fwd: null pointer");
  RADMMM_REQUIRE(c > 0 && c <= CP && col_offset >= 0 && ldw >= col_offset + c, "lu_weight_fwd: bad dims (c <= 256)");
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
  RADMMM_REQUIRE(c > 0 && c <= CP && col_offset >= 0 && ldw >= col_offset + c, "lu_weight_bwd: bad dims (c <= 256)");
  hipLaunchKernelGGL(lu_weight_bwd_kernel, dim3(c), dim3(NT), 0, static_cast<hipStream_t>(stream), P, lower, lower_diag,
                     upper, upper_diag, c, gW, ldw, col_offset, g_logdet, g_lower, g_upper, g_upper_diag);
  return radmmm::check_launch("lu_weight_bwd");
}
