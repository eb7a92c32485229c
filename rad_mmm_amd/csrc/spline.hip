// Piecewise-quadratic spline coupling transform (splines.py:241-339, forward branch) and its
// gradient.  One thread per (frame, channel) element; the element's 2K+1 parameters are staged
// through LDS with a coalesced block copy and then walked by the owning thread (row stride
// 2K+1 is odd -> conflict-free banks), overwriting them in place with exp values (forward) or
// with the parameter gradients (backward) that are then copied out coalesced.  HBM-bound on the
// q tensor: (2K+1)*4 bytes per element each way.
#include "common.h"

namespace {

constexpr int SP_THREADS = 128;
constexpr int SP_KMAX = 63;  // 2K+1 <= 127

struct SplineCore {
  // quantities of the forward pass needed by both directions
  float A;        // normalising area of the pdf
  int b;          // bin index
  float w_b, w_l, v_b, v_r, c_l, alpha, wbc, y0, L;
  float edge_r;   // right edge of the chosen bin as the search compared it (running sum of the widths; 1 for the last bin)
};

// In place: P[0..K) <- w_j (normalised), P[K..2K+1) <- ev_j (un-normalised exp(v-max)+1e-8)
__device__ __forceinline__ void spline_forward_core(float* P, int K, float x, SplineCore& o) {
  const float eps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
  float mw = -INFINITY, mv = -INFINITY;
  for (int j = 0; j < K; ++j) mw = fmaxf(mw, P[j]);
  for (int j = 0; j <= K; ++j) mv = fmaxf(mv, P[K + j]);
  float Z = 0.f;
  for (int j = 0; j < K; ++j) {
    const float e = expf(P[j] - mw);
    P[j] = e;
    Z += e;
  }
  for (int j = 0; j < K; ++j) P[j] = P[j] / Z;
  for (int j = 0; j <= K; ++j) P[K + j] = expf(P[K + j] - mv) + 1e-8f;
  float A = 0.f;
  for (int j = 0; j < K; ++j) A += (P[K + j] + P[K + j + 1]) / 2.f * P[j];
  // scan: first bin with cumulative width >= x (searchsorted, right=False); last edge forced to 1
  float wc = 0.f, cdf = 0.f;
  int b = K - 1;
  float w_l = 0.f, c_l = 0.f;
  bool found = false;
  for (int j = 0; j < K; ++j) {
    const float wj = P[j];
    const float wl_j = wc, cl_j = cdf;
    wc += wj;
    cdf += (P[K + j] / A + P[K + j + 1] / A) / 2.f * wj;
    const float edge = (j == K - 1) ? 1.f : wc;
    if (!found && edge >= x) {
      found = true;
      b = j;
      w_l = wl_j;
      c_l = cl_j;
      o.edge_r = edge;
    }
  }
  if (!found) {  // x > 1 cannot happen for inside elements; keep last bin
    b = K - 1;
    o.edge_r = 1.f;
  }
  o.A = A;
  o.b = b;
  o.w_b = P[b];
  o.w_l = w_l;
  o.c_l = c_l;
  o.v_b = P[K + b] / A;
  o.v_r = P[K + b + 1] / A;
  o.wbc = fmaxf(o.w_b, eps);
  o.alpha = (x - w_l) / o.wbc;
  o.y0 = o.alpha * o.alpha / 2.f * (o.v_r - o.v_b) * o.w_b + o.alpha * o.v_b * o.w_b + c_l;
  o.L = o.v_b + o.alpha * (o.v_r - o.v_b);  // torch.lerp(start, end, w) = start + w*(end-start) for w < 0.5
  if (o.alpha >= 0.5f) o.L = o.v_r - (o.v_r - o.v_b) * (1.f - o.alpha);
}

__global__ __launch_bounds__(SP_THREADS) void pq_spline_fwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ q, float* __restrict__ y, int ldy,
    float* __restrict__ logj_elem, int rows, int h, int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long e0 = (long long)blockIdx.x * SP_THREADS;
  const int ne = (int)((total - e0) < SP_THREADS ? (total - e0) : SP_THREADS);
  const float* src = q + e0 * nb;
  for (int i = threadIdx.x; i < ne * nb; i += SP_THREADS) sm[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < ne) {
    const long long e = e0 + threadIdx.x;
    const int r = (int)(e / h), c = (int)(e - (long long)r * h);
    const float xv = x[(long long)r * ldx + c];
    const float eps = 1.1920928955078125e-07f;
    float yo = xv, lj = 0.f;
    if (xv >= 0.f && xv < 1.f) {
      SplineCore o;
      spline_forward_core(sm + threadIdx.x * nb, K, xv, o);
      lj = logf(fmaxf(o.L, eps));
      yo = fminf(fmaxf(o.y0, eps), 1.f - eps);
    }
    y[(long long)r * ldy + c] = yo;
    logj_elem[e] = lj;
  }
}

// Inverse branch (splines.py:306-307, 327-339): locate the bin by the cdf, solve the quadratic for
// alpha (larger root), x = alpha * w_b + W_{b-1}; elements outside [0, 1) pass through
// (splines.py:241-265).  No log-jacobian in this direction.
__global__ __launch_bounds__(SP_THREADS) void pq_spline_inv_kernel(
    const float* __restrict__ y, int ldy, const float* __restrict__ q, float* __restrict__ x, int ldx, int rows, int h,
    int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long e0 = (long long)blockIdx.x * SP_THREADS;
  const int ne = (int)((total - e0) < SP_THREADS ? (total - e0) : SP_THREADS);
  const float* src = q + e0 * nb;
  for (int i = threadIdx.x; i < ne * nb; i += SP_THREADS) sm[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < ne) {
    const long long e = e0 + threadIdx.x;
    const int r = (int)(e / h), c = (int)(e - (long long)r * h);
    const float yv = y[(long long)r * ldy + c];
    const float eps = 1.1920928955078125e-07f;
    float xo = yv;
    if (yv >= 0.f && yv < 1.f) {
      float* P = sm + threadIdx.x * nb;
      float mw = -INFINITY, mv = -INFINITY;
      for (int j = 0; j < K; ++j) mw = fmaxf(mw, P[j]);
      for (int j = 0; j <= K; ++j) mv = fmaxf(mv, P[K + j]);
      float Z = 0.f;
      for (int j = 0; j < K; ++j) {
        const float ex = expf(P[j] - mw);
        P[j] = ex;
        Z += ex;
      }
      for (int j = 0; j < K; ++j) P[j] = P[j] / Z;
      for (int j = 0; j <= K; ++j) P[K + j] = expf(P[K + j] - mv) + 1e-8f;
      float A = 0.f;
      for (int j = 0; j < K; ++j) A += (P[K + j] + P[K + j + 1]) / 2.f * P[j];
      float wc = 0.f, cdf = 0.f, w_l = 0.f, c_l = 0.f;
      int b = K - 1;
      bool found = false;
      for (int j = 0; j < K; ++j) {
        const float wl_j = wc, cl_j = cdf;
        wc += P[j];
        cdf += (P[K + j] / A + P[K + j + 1] / A) / 2.f * P[j];
        const float edge = (j == K - 1) ? 1.f : cdf;             // cdf[..., -1] = 1 (splines.py:300)
        if (!found && edge >= yv) {
          found = true;
          b = j;
          w_l = wl_j;
          c_l = cl_j;
        }
      }
      const float w_b = P[b], v_b = P[K + b] / A, v_r = P[K + b + 1] / A;
      const float qa = (v_r - v_b) * w_b / 2.f, qb = v_b * w_b, qc = c_l - yv;
      const float alpha = (-qb + sqrtf(qb * qb - 4.f * qa * qc)) / (2.f * qa);
      xo = fminf(fmaxf(alpha * w_b + w_l, eps), 1.f - eps);
    }
    x[(long long)r * ldx + c] = xo;
  }
}

// INDEX accounting (tests): the bin the forward search picks for every element (-1: outside [0, 1), passed through) and the
// two edges of that bin as the search saw them -- its running sum of the softmax widths, not torch's cumsum
__global__ __launch_bounds__(SP_THREADS) void pq_spline_bins_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ q,
                                                                    int* __restrict__ bins, float* __restrict__ edge_l,
                                                                    float* __restrict__ edge_r, int rows, int h, int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long e0 = (long long)blockIdx.x * SP_THREADS;
  const int ne = (int)((total - e0) < SP_THREADS ? (total - e0) : SP_THREADS);
  const float* src = q + e0 * nb;
  for (int i = threadIdx.x; i < ne * nb; i += SP_THREADS) sm[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < ne) {
    const long long e = e0 + threadIdx.x;
    const int r = (int)(e / h), c = (int)(e - (long long)r * h);
    const float xv = x[(long long)r * ldx + c];
    int b = -1;
    float el = 0.f, er = 0.f;
    if (xv >= 0.f && xv < 1.f) {
      SplineCore o;
      spline_forward_core(sm + threadIdx.x * nb, K, xv, o);
      b = o.b;
      el = o.w_l;
      er = o.edge_r;
    }
    bins[e] = b;
    edge_l[e] = el;
    edge_r[e] = er;
  }
}

// logj_sum[r] = sum_c logj_elem[r*h + c]   (torch.sum(log_s, 1), common.py:1068)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ v, float* __restrict__ out,
                                                     int rows, int h) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < h; c += 64) s += v[(long long)r * h + c];
  s = radmmm::wave_sum(s);
  if (lane == 0) out[r] = s;
}

__global__ __launch_bounds__(SP_THREADS) void pq_spline_bwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ q, const float* __restrict__ gy,
    int ldgy, const float* __restrict__ glogj, float* __restrict__ gx, int ldgx,
    float* __restrict__ gq, int rows, int h, int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long e0 = (long long)blockIdx.x * SP_THREADS;
  const int ne = (int)((total - e0) < SP_THREADS ? (total - e0) : SP_THREADS);
  const float* src = q + e0 * nb;
  for (int i = threadIdx.x; i < ne * nb; i += SP_THREADS) sm[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < ne) {
    const long long e = e0 + threadIdx.x;
    const int r = (int)(e / h), c = (int)(e - (long long)r * h);
    const float xv = x[(long long)r * ldx + c];
    const float gyv = gy[(long long)r * ldgy + c];
    const float glj = glogj ? glogj[r] : 0.f;
    const float eps = 1.1920928955078125e-07f;
    float* P = sm + threadIdx.x * nb;
    float gxv = gyv;
    if (xv >= 0.f && xv < 1.f) {
      SplineCore o;
      spline_forward_core(P, K, xv, o);
      const float A = o.A, al = o.alpha;
      const int b = o.b;
      const float gy0 = (o.y0 >= eps && o.y0 <= 1.f - eps) ? gyv : 0.f;
      const float gL = (o.L >= eps) ? glj / o.L : 0.f;
      const float ga = gy0 * o.w_b * (o.v_b + al * (o.v_r - o.v_b)) + gL * (o.v_r - o.v_b);
      gxv = ga / o.wbc;
      const float gvb = gy0 * (al - al * al / 2.f) * o.w_b + gL * (1.f - al);
      const float gvr = gy0 * (al * al / 2.f) * o.w_b + gL * al;
      const float gwb = gy0 * (al * al / 2.f * (o.v_r - o.v_b) + al * o.v_b) -
                        ((o.w_b >= eps) ? ga * al / o.wbc : 0.f);
      const float gwl = -ga / o.wbc;  // d/d w_l, spread over w_j, j < b
      // direct gradients w.r.t. normalised v_j and w_j as closed forms of j
      auto Gv = [&](int j) {
        float g = 0.f;
        if (j == b) g += gvb;
        if (j == b + 1) g += gvr;
        if (j < b) g += gy0 * P[j] / 2.f;                    // c_l term, v_j
        if (j >= 1 && j <= b) g += gy0 * P[j - 1] / 2.f;     // c_l term, v_{j} as right end of bin j-1
        return g;
      };
      auto Gw_direct = [&](int j) {
        float g = 0.f;
        if (j < b) g += gy0 * (P[K + j] + P[K + j + 1]) / (2.f * A) + gwl;
        if (j == b) g += gwb;
        return g;
      };
      float Sv = 0.f;
      for (int j = 0; j <= K; ++j) Sv += Gv(j) * (P[K + j] / A);
      const float GA = -Sv / A;
      float Sw = 0.f;
      for (int j = 0; j < K; ++j) Sw += (Gw_direct(j) + GA * (P[K + j] + P[K + j + 1]) / 2.f) * P[j];
      // interleaved in-place write (slot j of either region is dead once step j has read it)
      float w_prev = 0.f;
      for (int j = 0; j <= K; ++j) {
        const float wj = j < K ? P[j] : 0.f;
        const float evj = P[K + j];
        const float evn = j < K ? P[K + j + 1] : 0.f;
        // slot j-1 of the width region is already overwritten: its w value is carried in w_prev
        float gvj = 0.f;
        if (j == b) gvj += gvb;
        if (j == b + 1) gvj += gvr;
        if (j < b) gvj += gy0 * wj / 2.f;
        if (j >= 1 && j <= b) gvj += gy0 * w_prev / 2.f;
        const float dA_dev = (wj + w_prev) / 2.f;
        const float g_ev = gvj / A + GA * dA_dev;
        if (j < K) {
          float gwj = GA * (evj + evn) / 2.f;
          if (j < b) gwj += gy0 * (evj + evn) / (2.f * A) + gwl;
          if (j == b) gwj += gwb;
          P[j] = wj * (gwj - Sw);
        }
        P[K + j] = g_ev * (evj - 1e-8f);
        w_prev = wj;
      }
    } else {
      for (int j = 0; j < nb; ++j) P[j] = 0.f;
    }
    gx[(long long)r * ldgx + c] = gxv;
  }
  __syncthreads();
  float* dst = gq + e0 * nb;
  for (int i = threadIdx.x; i < ne * nb; i += SP_THREADS) dst[i] = sm[i];
}

}  // namespace

extern "C" int radmmm_pq_spline_fwd(const float* x, int ldx, const float* q, int ldq, float* y, int ldy,
                                    float* logj_sum, int rows, int h, int K, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && q && y && logj_sum, "pq_spline_fwd: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && K >= 1 && K <= SP_KMAX, "pq_spline_fwd: bad dims");
  RADMMM_REQUIRE(ldq == h * (2 * K + 1), "pq_spline_fwd: q must be dense (ldq == h*(2K+1))");
  // logj_sum holds rows + rows*h floats: [0, rows) the per-frame sums, then the per-element
  // log-jacobians they are reduced from (see include/radmmm_hip.h)
  float* logj_elem = logj_sum + rows;
  const long long total = (long long)rows * h;
  const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
  const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
  hipLaunchKernelGGL(pq_spline_fwd_kernel, dim3(nblk), dim3(SP_THREADS), smem,
                     static_cast<hipStream_t>(stream), x, ldx, q, y, ldy, logj_elem, rows, h, K);
  hipLaunchKernelGGL(rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0,
                     static_cast<hipStream_t>(stream), logj_elem, logj_sum, rows, h);
  return radmmm::check_launch("pq_spline_fwd");
}

extern "C" int radmmm_pq_spline_bins(const float* x, int ldx, const float* q, int ldq, int32_t* bins, float* edge_l, float* edge_r,
                                     int rows, int h, int K, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && q && bins && edge_l && edge_r, "pq_spline_bins: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && K >= 1 && K <= SP_KMAX, "pq_spline_bins: bad dims");
  RADMMM_REQUIRE(ldq == h * (2 * K + 1), "pq_spline_bins: q must be dense (ldq == h*(2K+1))");
  const long long total = (long long)rows * h;
  const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
  const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
  hipLaunchKernelGGL(pq_spline_bins_kernel, dim3(nblk), dim3(SP_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx, q, bins,
                     edge_l, edge_r, rows, h, K);
  return radmmm::check_launch("pq_spline_bins");
}

extern "C" int radmmm_pq_spline_inv(const float* y, int ldy, const float* q, int ldq, float* x, int ldx, int rows, int h,
                                    int K, radmmm_stream_t stream) {
  RADMMM_REQUIRE(y && q && x, "pq_spline_inv: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && K >= 1 && K <= SP_KMAX, "pq_spline_inv: bad dims");
  RADMMM_REQUIRE(ldq == h * (2 * K + 1), "pq_spline_inv: q must be dense (ldq == h*(2K+1))");
  const long long total = (long long)rows * h;
  const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
  const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
  hipLaunchKernelGGL(pq_spline_inv_kernel, dim3(nblk), dim3(SP_THREADS), smem, static_cast<hipStream_t>(stream), y, ldy, q,
                     x, ldx, rows, h, K);
  return radmmm::check_launch("pq_spline_inv");
}

extern "C" int radmmm_pq_spline_bwd(const float* x, int ldx, const float* q, int ldq, const float* gy,
                                    int ldgy, const float* glogj, float* gx, int ldgx, float* gq,
                                    int ldgq, int rows, int h, int K, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && q && gy && gx && gq, "pq_spline_bwd: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && K >= 1 && K <= SP_KMAX, "pq_spline_bwd: bad dims");
  RADMMM_REQUIRE(ldq == h * (2 * K + 1) && ldgq == ldq, "pq_spline_bwd: q/gq must be dense");
  const long long total = (long long)rows * h;
  const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
  const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
  hipLaunchKernelGGL(pq_spline_bwd_kernel, dim3(nblk), dim3(SP_THREADS), smem,
                     static_cast<hipStream_t>(stream), x, ldx, q, gy, ldgy, glogj, gx, ldgx, gq, rows,
                     h, K);
  return radmmm::check_launch("pq_spline_bwd");
}
