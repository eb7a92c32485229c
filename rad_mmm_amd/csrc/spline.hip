// Piecewise-quadratic spline coupling transform (splines.py:241-339, forward branch) and its
// gradient.  One thread per (frame, channel) element; the element's 2K+1 parameters are staged
// through LDS with a coalesced block copy and then walked by the owning thread (row stride
// 2K+1 is odd -> conflict-free banks), overwriting them in place with exp values (forward) or
// with the parameter gradients (backward) that are then copied out coalesced.  HBM-bound on the
// q tensor: (2K+1)*4 bytes per element each way.
//
// Round 4: the shipped bin counts (K = 32: decoders.py:51-61 hard-codes n_bins = 32 for the decoder's spline flows; K = 8: the
// layer's own default, common.py:1014) run REGISTER-RESIDENT kernels (pq_spline_*_reg_kernel<K>): 16-byte coalesced copies
// between HBM and LDS, the element's 2K+1 parameters read ONCE from LDS into registers (stride 2K+1 words: conflict-free),
// every loop unrolled over the compile-time K with the chosen bin's quantities captured by selects during the scan (no
// dynamic register indexing).  The runtime-K kernels above them walked the parameters in LDS six times with a dependent
// ds_read per iteration at two waves per SIMD: latency-bound at 21-22 % of the HBM roofline (profiles/r03_h_c5_kernel_stats.txt).
// Same float operations in the same order on the width path (softmax, running sum, bin search): identical bins.
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

// ---- register-resident kernels for compile-time K ------------------------------------------------------------------------
constexpr int SPR_THREADS = 128;

typedef float v4f __attribute__((ext_vector_type(4)));

// A block walks chunks of SPR_THREADS elements (chunk = blockIdx.x, += gridDim.x).  The NEXT chunk's parameters are fetched
// into registers with coalesced 16-byte loads while the current chunk computes (the loads fly for the whole compute phase;
// LDS -- 33 KB per chunk at K = 32 -- stays single-buffered, so four blocks = eight waves per CU keep their residency), then
// parked in LDS, from where every thread reads its own element's 2K+1 values (odd stride: conflict-free).
template <int K>
struct ChunkPrefetch {
  static constexpr int nb = 2 * K + 1;
  static constexpr int N4 = SPR_THREADS * nb / 4;                           // float4s of a full chunk (128 nb is a multiple of 4)
  static constexpr int NV = (N4 + SPR_THREADS - 1) / SPR_THREADS;
  v4f v[NV];
  float tail;                                                               // a last chunk's (ne nb) % 4 floats, one per thread
  __device__ __forceinline__ void fetch(const float* __restrict__ q, long long e0, int ne, int tid) {
    const v4f* s4 = reinterpret_cast<const v4f*>(q + e0 * nb);
    const int n = ne * nb, n4 = n >> 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * SPR_THREADS;
      if (idx < n4) v[i] = __builtin_nontemporal_load(s4 + idx);
    }
    tail = (4 * n4 + tid < n) ? q[e0 * nb + 4 * n4 + tid] : 0.f;
  }
  __device__ __forceinline__ void park(float* sm, int ne, int tid) const {
    v4f* d4 = reinterpret_cast<v4f*>(sm);
    const int n = ne * nb, n4 = n >> 2;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * SPR_THREADS;
      if (idx < n4) d4[idx] = v[i];
    }
    if (4 * n4 + tid < n) sm[4 * n4 + tid] = tail;
  }
};
__device__ __forceinline__ void copy_out16(float* __restrict__ dst, const float* __restrict__ src, int n, int tid, int nthr) {
  const int n4 = n >> 2;
  const v4f* s4 = reinterpret_cast<const v4f*>(src);
  v4f* d4 = reinterpret_cast<v4f*>(dst);
  for (int i = tid; i < n4; i += nthr) __builtin_nontemporal_store(s4[i], d4 + i);
  for (int i = 4 * n4 + tid; i < n; i += nthr) dst[i] = src[i];
}

// exp(d) for d <= 0 (softmax numerators): 2^(d log2 e) on the hardware exp2 with the product's rounding error carried as
// a first-order correction -- t + lo = d log2(e) to ~2^-48, so the result is the 1-ulp v_exp_f32 plus ~0.5 ulp whatever |d|
// (a plain __expf loses |t| 2^-24 ln 2 relative: 3.5 ulp at d = -7, which would move bin edges).  6 instructions against
// ~15 of libm's expf (no overflow / denormal branches: d <= 0, and results below 2^-126 flush to 0 like a softmax wants).
__device__ __forceinline__ float exp_nonpos(float d) {
  const float L2E = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-08f, LN2 = 0.693147182464599609375f;
  const float t = d * L2E;
  float lo = __builtin_fmaf(d, L2E, -t);
  lo = __builtin_fmaf(d, L2E_LO, lo);
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * LN2, e);
}
// 1 / a to < 1 ulp: v_rcp_f32 + one Newton step
__device__ __forceinline__ float rcp_nr(float a) {
  const float r = __builtin_amdgcn_rcpf(a);
  return __builtin_fmaf(__builtin_fmaf(-a, r, 1.f), r, r);
}
// n / d from r = rcp_nr(d): one residual correction (the IEEE sequence without range scaling and special cases: operands
// here are softmax terms in (0, 1] over sums in [1, K])
__device__ __forceinline__ float div_r(float n, float d, float r) {
  const float q = n * r;
  return __builtin_fmaf(__builtin_fmaf(-q, d, n), r, q);
}

// The forward quantities of one element from its parameters in registers.  In: w[] = w~, ev[] = v~.  Out: w[] normalised
// widths (softmax), ev[] = exp(v~ - max) + 1e-8 (un-normalised).
// One pass does the area sum, the cdf and the bin search: the edges are the running sum of the widths (monotone), so the
// bin is the number of edges below x and the chosen bin's quantities are the LAST ones selected under `edge_j < x`
// (same bin as the walk's "first edge >= x"; the last edge is 1 and x < 1); the cdf at an edge is the area up to it over A.
template <int K>
__device__ __forceinline__ void spline_forward_core_reg(float (&w)[K], float (&ev)[K + 1], float x, SplineCore& o) {
  const float eps = 1.1920928955078125e-07f;
  float mw = w[0], mv = ev[0];
#pragma unroll
  for (int j = 1; j < K; ++j) mw = fmaxf(mw, w[j]);
#pragma unroll
  for (int j = 1; j <= K; ++j) mv = fmaxf(mv, ev[j]);
  float Z = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = exp_nonpos(w[j] - mw);
    Z += w[j];
  }
  const float rZ = rcp_nr(Z);
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = div_r(w[j], Z, rZ);
#pragma unroll
  for (int j = 0; j <= K; ++j) ev[j] = exp_nonpos(ev[j] - mv) + 1e-8f;
  float wc = 0.f, area = 0.f, w_l = 0.f, a_l = 0.f;
  int b = 0;
  float w_b = w[0], ev_b = ev[0], ev_r = ev[1], edge_r = (K == 1) ? 1.f : 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    area = __builtin_fmaf((ev[j] + ev[j + 1]) * 0.5f, w[j], area);
    wc += w[j];
    if (j == 0 && K > 1) edge_r = wc;
    if (j < K - 1) {
      const bool lt = wc < x;
      b = lt ? j + 1 : b;
      w_l = lt ? wc : w_l;
      a_l = lt ? area : a_l;
      w_b = lt ? w[j + 1 < K ? j + 1 : 0] : w_b;
      ev_b = lt ? ev[j + 1] : ev_b;
      ev_r = lt ? ev[j + 2 <= K ? j + 2 : 0] : ev_r;
    }
  }
  const float A = area, rA = rcp_nr(A);
  o.edge_r = edge_r;          // (tests: completed by the caller for b > 0, see bins mode)
  o.A = A;
  o.b = b;
  o.w_b = w_b;
  o.w_l = w_l;
  o.c_l = a_l * rA;
  o.v_b = ev_b * rA;
  o.v_r = ev_r * rA;
  o.wbc = fmaxf(o.w_b, eps);
  o.alpha = (x - w_l) / o.wbc;
  o.y0 = o.alpha * o.alpha / 2.f * (o.v_r - o.v_b) * o.w_b + o.alpha * o.v_b * o.w_b + o.c_l;
  o.L = o.v_b + o.alpha * (o.v_r - o.v_b);
  if (o.alpha >= 0.5f) o.L = o.v_r - (o.v_r - o.v_b) * (1.f - o.alpha);
}

template <int K>
__device__ __forceinline__ void load_params_reg(const float* P, float (&w)[K], float (&ev)[K + 1]) {
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = P[j];
#pragma unroll
  for (int j = 0; j <= K; ++j) ev[j] = P[K + j];
}

// Inverse branch on registers (splines.py:306-307, 327-339): the bin is found on the cdf edges (partial areas over the total
// area, monotone like the widths), then the quadratic is solved for alpha as the walk above does.
template <int K>
__device__ __forceinline__ float spline_inverse_core_reg(float (&w)[K], float (&ev)[K + 1], float yv) {
  const float eps = 1.1920928955078125e-07f;
  float mw = w[0], mv = ev[0];
#pragma unroll
  for (int j = 1; j < K; ++j) mw = fmaxf(mw, w[j]);
#pragma unroll
  for (int j = 1; j <= K; ++j) mv = fmaxf(mv, ev[j]);
  float Z = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = exp_nonpos(w[j] - mw);
    Z += w[j];
  }
  const float rZ = rcp_nr(Z);
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = div_r(w[j], Z, rZ);
#pragma unroll
  for (int j = 0; j <= K; ++j) ev[j] = exp_nonpos(ev[j] - mv) + 1e-8f;
  float A = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) A = __builtin_fmaf((ev[j] + ev[j + 1]) * 0.5f, w[j], A);
  const float rA = rcp_nr(A);
  float wc = 0.f, area = 0.f, w_l = 0.f, c_l = 0.f;
  float w_b = w[0], ev_b = ev[0], ev_r = ev[1];
#pragma unroll
  for (int j = 0; j < K - 1; ++j) {
    area = __builtin_fmaf((ev[j] + ev[j + 1]) * 0.5f, w[j], area);
    wc += w[j];
    const float cdf = area * rA;
    const bool lt = cdf < yv;                          // the last edge is 1 and yv < 1
    w_l = lt ? wc : w_l;
    c_l = lt ? cdf : c_l;
    w_b = lt ? w[j + 1] : w_b;
    ev_b = lt ? ev[j + 1] : ev_b;
    ev_r = lt ? ev[j + 2 <= K ? j + 2 : 0] : ev_r;
  }
  const float v_b = ev_b * rA, v_r = ev_r * rA;
  const float qa = (v_r - v_b) * w_b / 2.f, qb = v_b * w_b, qc = c_l - yv;
  const float alpha = (-qb + sqrtf(qb * qb - 4.f * qa * qc)) / (2.f * qa);
  return fminf(fmaxf(alpha * w_b + w_l, eps), 1.f - eps);
}

// MODE 0: forward (y, logj_elem); MODE 1: bins (tests); MODE 2: inverse (x = the values to invert, y = the result)
template <int K, int MODE>
__global__ __launch_bounds__(SPR_THREADS, 2) void pq_spline_fwd_reg_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ q,
                                                                         float* __restrict__ y, int ldy, float* __restrict__ logj_elem,
                                                                         int* __restrict__ bins, float* __restrict__ edge_l,
                                                                         float* __restrict__ edge_r, int rows, int h) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long nchunks = (total + SPR_THREADS - 1) / SPR_THREADS;
  const int tid = threadIdx.x;
  ChunkPrefetch<K> pre;
  long long chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  float x_nxt = 2.f;
  int r_nxt = 0, c_nxt = 0;
  auto fetch = [&](long long ch) __attribute__((always_inline)) {
    const long long e0 = ch * SPR_THREADS;
    const int ne = (int)((total - e0) < SPR_THREADS ? (total - e0) : SPR_THREADS);
    x_nxt = 2.f;
    if (tid < ne) {
      const long long e = e0 + tid;
      r_nxt = (int)(e / h);
      c_nxt = (int)(e - (long long)r_nxt * h);
      x_nxt = x[(long long)r_nxt * ldx + c_nxt];
    }
    pre.fetch(q, e0, ne, tid);
  };
  fetch(chunk);
  while (true) {
    const long long e0 = chunk * SPR_THREADS;
    const int ne = (int)((total - e0) < SPR_THREADS ? (total - e0) : SPR_THREADS);
    const long long e = e0 + tid;
    const float xv = x_nxt;
    const int r = r_nxt, c = c_nxt;
    pre.park(sm, ne, tid);
    __syncthreads();
    const long long next = chunk + gridDim.x;
    const bool inside = tid < ne && xv >= 0.f && xv < 1.f;
    float w[K], ev[K + 1];
    if (inside) load_params_reg<K>(sm + tid * nb, w, ev);
    __syncthreads();                                   // every thread holds its parameters: LDS is free for the next chunk
    if (next < nchunks) fetch(next);                   // in flight during the arithmetic below
    if (tid < ne) {
      const float eps = 1.1920928955078125e-07f;
      float yo = xv, lj = 0.f, el = 0.f, er = 0.f;
      int bo = -1;
      if (inside && MODE == 2) {
        yo = spline_inverse_core_reg<K>(w, ev, xv);
      } else if (inside) {
        SplineCore o;
        spline_forward_core_reg<K>(w, ev, xv, o);
        lj = logf(fmaxf(o.L, eps));
        yo = fminf(fmaxf(o.y0, eps), 1.f - eps);
        bo = o.b;
        el = o.w_l;
        if (MODE == 1) {            // right edge of the chosen bin as the search saw it: the running sum again (1 for the last bin)
          float wc = 0.f;
          er = 1.f;
#pragma unroll
          for (int j = 0; j < K - 1; ++j) {
            wc += w[j];
            if (j == bo) er = wc;
          }
        }
      }
      if (MODE == 2) {
        y[(long long)r * ldy + c] = yo;
      } else if (MODE == 0) {
        y[(long long)r * ldy + c] = yo;
        logj_elem[e] = lj;
      } else {
        bins[e] = bo;
        edge_l[e] = el;
        edge_r[e] = er;
      }
    }
    if (next >= nchunks) break;
    chunk = next;
  }
}

// Gradient.  With v_j = ev_j / A, t_j = (ev_j + ev_{j+1}) / 2 (splines.py:267-326 differentiated by hand, as the walk above):
//   gv_j  = [j = b] gvb + [j = b+1] gvr + gy0/2 ([j < b] w_j + [1 <= j <= b] w_{j-1})          d / d v_j
//   GA    = -sum_j gv_j v_j / A                                                                   d / d A
//   d ev_j = gv_j / A + GA (w_j + w_{j-1}) / 2;     d v~_j = d ev_j (ev_j - 1e-8)
//   gw_j  = [j < b] (gy0 t_j / A + gwl) + [j = b] gwb + GA t_j;   d w~_j = w_j (gw_j - sum_i gw_i w_i)    (softmax)
template <int K>
__global__ __launch_bounds__(SPR_THREADS, 2) void pq_spline_bwd_reg_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ q,
                                                                         const float* __restrict__ gy, int ldgy,
                                                                         const float* __restrict__ glogj, float* __restrict__ gx, int ldgx,
                                                                         float* __restrict__ gq, int rows, int h) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int nb = 2 * K + 1;
  const long long total = (long long)rows * h;
  const long long nchunks = (total + SPR_THREADS - 1) / SPR_THREADS;
  const int tid = threadIdx.x;
  ChunkPrefetch<K> pre;
  long long chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  float x_nxt = 2.f, gy_nxt = 0.f, glj_nxt = 0.f;
  int r_nxt = 0, c_nxt = 0;
  auto fetch = [&](long long ch) __attribute__((always_inline)) {
    const long long e0 = ch * SPR_THREADS;
    const int ne = (int)((total - e0) < SPR_THREADS ? (total - e0) : SPR_THREADS);
    x_nxt = 2.f;
    if (tid < ne) {
      const long long e = e0 + tid;
      r_nxt = (int)(e / h);
      c_nxt = (int)(e - (long long)r_nxt * h);
      x_nxt = x[(long long)r_nxt * ldx + c_nxt];
      gy_nxt = gy[(long long)r_nxt * ldgy + c_nxt];
      glj_nxt = glogj ? glogj[r_nxt] : 0.f;
    }
    pre.fetch(q, e0, ne, tid);
  };
  fetch(chunk);
  while (true) {
    const long long e0 = chunk * SPR_THREADS;
    const int ne = (int)((total - e0) < SPR_THREADS ? (total - e0) : SPR_THREADS);
    const float xv = x_nxt, gyv = gy_nxt, glj = glj_nxt;
    const int r = r_nxt, c = c_nxt;
    pre.park(sm, ne, tid);
    __syncthreads();
    const long long next = chunk + gridDim.x;
    if (next < nchunks) fetch(next);                   // in flight during the arithmetic below
    if (tid < ne) {
      const float eps = 1.1920928955078125e-07f;
      float* P = sm + tid * nb;                        // read and overwritten by this thread only
      float gxv = gyv;
      if (xv >= 0.f && xv < 1.f) {
        float w[K], ev[K + 1];
        load_params_reg<K>(P, w, ev);
        SplineCore o;
        spline_forward_core_reg<K>(w, ev, xv, o);
        const float A = o.A, al = o.alpha, rA = rcp_nr(A);
        const int b = o.b;
        const float gy0 = (o.y0 >= eps && o.y0 <= 1.f - eps) ? gyv : 0.f;
        const float gL = (o.L >= eps) ? glj / o.L : 0.f;
        const float ga = gy0 * o.w_b * (o.v_b + al * (o.v_r - o.v_b)) + gL * (o.v_r - o.v_b);
        gxv = ga / o.wbc;
        const float gvb = gy0 * (al - al * al / 2.f) * o.w_b + gL * (1.f - al);
        const float gvr = gy0 * (al * al / 2.f) * o.w_b + gL * al;
        const float gwb = gy0 * (al * al / 2.f * (o.v_r - o.v_b) + al * o.v_b) - ((o.w_b >= eps) ? ga * al / o.wbc : 0.f);
        const float gwl = -ga / o.wbc;
        const float hg = gy0 * 0.5f;
        // the per-bin gradients wait in the element's own LDS slots between the passes (its parameters live in registers,
        // the slots are free): gv_j in P[K + j], gw_j in P[j] -- 65 registers less than keeping them, which is what lets the
        // next chunk's 68 prefetch registers fly during this arithmetic at two waves per SIMD
        float Sv = 0.f;                       // sum_j gv_j ev_j  (= A sum_j gv_j v_j)
#pragma unroll
        for (int j = 0; j <= K; ++j) {
          float ws = 0.f;                     // [j < b] w_j + [1 <= j <= b] w_{j-1}
          if (j < K) ws = (j < b) ? w[j < K ? j : 0] : 0.f;
          if (j >= 1) ws += (j <= b) ? w[j >= 1 ? j - 1 : 0] : 0.f;
          float g = hg * ws;
          g += (j == b) ? gvb : 0.f;
          g += (j == b + 1) ? gvr : 0.f;
          P[K + j] = g;
          Sv = __builtin_fmaf(g, ev[j], Sv);
        }
        const float GA = -(Sv * rA) * rA;
        const float c1 = __builtin_fmaf(gy0, rA, GA);   // coefficient of t_j below the bin
        float Sw = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float cf = (j < b) ? c1 : GA;
          const float ad = (j < b) ? gwl : ((j == b) ? gwb : 0.f);
          const float gw = __builtin_fmaf(cf, (ev[j] + ev[j + 1]) * 0.5f, ad);
          P[j] = gw;
          Sw = __builtin_fmaf(gw, w[j], Sw);
        }
        asm volatile("" ::: "memory");        // (no store-to-load forwarding: the point is to NOT hold these in registers)
        const float hGA = 0.5f * GA;
#pragma unroll
        for (int j = 0; j < K; ++j) P[j] = w[j] * (P[j] - Sw);
#pragma unroll
        for (int j = 0; j <= K; ++j) {
          const float wj = j < K ? w[j < K ? j : 0] : 0.f;
          const float w_prev = j >= 1 ? w[j >= 1 ? j - 1 : 0] : 0.f;
          const float g_ev = __builtin_fmaf(P[K + j], rA, hGA * (wj + w_prev));
          P[K + j] = g_ev * (ev[j] - 1e-8f);
        }
      } else {
#pragma unroll
        for (int j = 0; j < nb; ++j) P[j] = 0.f;
      }
      gx[(long long)r * ldgx + c] = gxv;
    }
    __syncthreads();
    copy_out16(gq + e0 * nb, sm, ne * nb, tid, SPR_THREADS);
    if (next >= nchunks) break;
    __syncthreads();                                   // the copy has read LDS: the next chunk may be parked
    chunk = next;
  }
}

// blocks of the chunk-walking kernels: what is resident at once (LDS: 4 blocks per CU at K = 32; wave slots at K = 8)
inline int spline_reg_grid(long long total, int K) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  const long long nchunks = (total + SPR_THREADS - 1) / SPR_THREADS;
  const long long res = (long long)cus * (K == 32 ? 4 : 8);
  return (int)(nchunks < res ? nchunks : res);
}
inline bool spline_generic_forced() {
  const char* v = radmmm::debug_env("RADMMM_SPLINE");
  return v && v[0] == 'g';          // RADMMM_DEBUG=1 RADMMM_SPLINE=generic: the runtime-K kernels for every K (A/B, bit-identity tests)
}
inline bool spline_reg_ok(const float* q, int K) { return (K == 8 || K == 32) && radmmm::aligned16(q) && !spline_generic_forced(); }

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
  if (spline_reg_ok(q, K)) {
    const int nblk = spline_reg_grid(total, K);
    const size_t smem = (size_t)SPR_THREADS * (2 * K + 1) * sizeof(float);
    if (K == 32)
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<32, 0>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx,
                         q, y, ldy, logj_elem, nullptr, nullptr, nullptr, rows, h);
    else
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<8, 0>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx,
                         q, y, ldy, logj_elem, nullptr, nullptr, nullptr, rows, h);
  } else {
    const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
    const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
    hipLaunchKernelGGL(pq_spline_fwd_kernel, dim3(nblk), dim3(SP_THREADS), smem,
                       static_cast<hipStream_t>(stream), x, ldx, q, y, ldy, logj_elem, rows, h, K);
  }
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
  if (spline_reg_ok(q, K)) {        // the bins of the kernel that radmmm_pq_spline_fwd runs for this K
    const int nblk = spline_reg_grid(total, K);
    const size_t smem = (size_t)SPR_THREADS * (2 * K + 1) * sizeof(float);
    if (K == 32)
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<32, 1>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx,
                         q, nullptr, 0, nullptr, bins, edge_l, edge_r, rows, h);
    else
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<8, 1>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx,
                         q, nullptr, 0, nullptr, bins, edge_l, edge_r, rows, h);
    return radmmm::check_launch("pq_spline_bins");
  }
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
  if (spline_reg_ok(q, K)) {
    const int nblk = spline_reg_grid(total, K);
    const size_t smem = (size_t)SPR_THREADS * (2 * K + 1) * sizeof(float);
    if (K == 32)
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<32, 2>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), y, ldy,
                         q, x, ldx, nullptr, nullptr, nullptr, nullptr, rows, h);
    else
      hipLaunchKernelGGL((pq_spline_fwd_reg_kernel<8, 2>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), y, ldy,
                         q, x, ldx, nullptr, nullptr, nullptr, nullptr, rows, h);
    return radmmm::check_launch("pq_spline_inv");
  }
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
  if (spline_reg_ok(q, K) && radmmm::aligned16(gq)) {
    const int nblk = spline_reg_grid(total, K);
    const size_t smem = (size_t)SPR_THREADS * (2 * K + 1) * sizeof(float);
    if (K == 32)
      hipLaunchKernelGGL((pq_spline_bwd_reg_kernel<32>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx, q,
                         gy, ldgy, glogj, gx, ldgx, gq, rows, h);
    else
      hipLaunchKernelGGL((pq_spline_bwd_reg_kernel<8>), dim3(nblk), dim3(SPR_THREADS), smem, static_cast<hipStream_t>(stream), x, ldx, q,
                         gy, ldgy, glogj, gx, ldgx, gq, rows, h);
    return radmmm::check_launch("pq_spline_bwd");
  }
  const int nblk = (int)((total + SP_THREADS - 1) / SP_THREADS);
  const size_t smem = (size_t)SP_THREADS * (2 * K + 1) * sizeof(float);
  hipLaunchKernelGGL(pq_spline_bwd_kernel, dim3(nblk), dim3(SP_THREADS), smem,
                     static_cast<hipStream_t>(stream), x, ldx, q, gy, ldgy, glogj, gx, ldgx, gq, rows,
                     h, K);
  return radmmm::check_launch("pq_spline_bwd");
}
