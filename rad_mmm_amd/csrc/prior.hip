// On-device data path: the attention prior and the energy average the reference's CPU
// dataset workers compute per utterance (data.py:61-102, 339-342, 363-366, 396-417; padded into the batch by
// DataCollate, data.py:678-741).  At 8 GPUs x 32 utterances x 800 frames per step the scipy
// beta-binomial / ndimage.zoom calls are a CPU-side bottleneck; here the anchor priors are built once
// per rounded size on the device and a whole batch is interpolated, renormalised and zero-padded in
// one launch.  All arithmetic in float64 as the reference (scipy) does; the batch tensor is fp32 as
// DataCollate's FloatTensor.
#include "common.h"

namespace {

// out[i][k] = pmf_{BetaBinomial(n = P-1, a = s (i+1), b = s (M - i))}(k)   (data.py:90-102)
//           = C(n, k) B(k + a, n - k + b) / B(a, b), in log space
__global__ void betabinom_prior_kernel(int P, int M, double s, double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * M) return;
  const int i = (int)(idx / P), kk = (int)(idx - (long long)i * P);
  const double n = P - 1, k = kk;
  const double a = s * (i + 1), b = s * (M - i);
  const double logc = lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
  const double lb1 = lgamma(k + a) + lgamma(n - k + b) - lgamma(n + a + b);
  const double lb0 = lgamma(a) + lgamma(b) - lgamma(a + b);
  out[idx] = exp(logc + lb1 - lb0);
}

struct ZoomItem {            // one utterance: anchor prior [bh][bw] -> [m][p]
  const double* bank;
  long long bh, bw, m, p;
};

// scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=False): output index o samples the input at
// o (n_in - 1) / (n_out - 1) (coordinate 0 when n_out == 1), bilinear; then rows are renormalised
// (data.py:80-84).  One workgroup per (output row, utterance); rows >= m and columns >= p are zeros.
__global__ __launch_bounds__(256) void prior_zoom_kernel(const ZoomItem* __restrict__ items, float* __restrict__ out,
                                                         int Tmax, int Nmax) {
  __shared__ double red[256];
  const ZoomItem it = items[blockIdx.y];
  const int r = blockIdx.x, tid = threadIdx.x;
  float* o = out + ((long long)blockIdx.y * Tmax + r) * Nmax;
  if (r >= it.m) {
    for (int c = tid; c < Nmax; c += 256) o[c] = 0.f;
    return;
  }
  const int bh = (int)it.bh, bw = (int)it.bw, m = (int)it.m, p = (int)it.p;
  const double zr = m > 1 ? (double)(bh - 1) / (double)(m - 1) : 1.0;
  const double zc = p > 1 ? (double)(bw - 1) / (double)(p - 1) : 1.0;
  const double cr = r * zr;
  int r0 = (int)floor(cr);
  r0 = r0 > bh - 1 ? bh - 1 : r0;
  const double fr = cr - r0;
  const int r1 = r0 + 1 > bh - 1 ? bh - 1 : r0 + 1;
  const double* b0 = it.bank + (long long)r0 * bw;
  const double* b1 = it.bank + (long long)r1 * bw;
  double sum = 0.0;
  for (int c = tid; c < p; c += 256) {
    const double cc = c * zc;
    int c0 = (int)floor(cc);
    c0 = c0 > bw - 1 ? bw - 1 : c0;
    const double fc = cc - c0;
    const int c1 = c0 + 1 > bw - 1 ? bw - 1 : c0 + 1;
    const double top = b0[c0] * (1.0 - fc) + b0[c1] * fc;
    const double bot = b1[c0] * (1.0 - fc) + b1[c1] * fc;
    sum += top * (1.0 - fr) + bot * fr;
  }
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double inv = 1.0 / red[0];
  for (int c = tid; c < Nmax; c += 256) {
    double v = 0.0;
    if (c < p) {                                               // recomputed: cheaper than keeping a row in LDS
      const double cc = c * zc;
      int c0 = (int)floor(cc);
      c0 = c0 > bw - 1 ? bw - 1 : c0;
      const double fc = cc - c0;
      const int c1 = c0 + 1 > bw - 1 ? bw - 1 : c0 + 1;
      const double top = b0[c0] * (1.0 - fc) + b0[c1] * fc;
      const double bot = b1[c0] * (1.0 - fc) + b1[c1] * fc;
      v = (top * (1.0 - fr) + bot * fr) * inv;
    }
    o[c] = (float)v;
  }
}

// energy_avg[b][t] = mean_c mel[b][c][t], then (x + 20) / 20 (data.py:339-342, 363-366)
__global__ void energy_average_kernel(const float* __restrict__ mel, float* __restrict__ out, int n_mel, int T, int scaled) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const float* mp = mel + (long long)b * n_mel * T + t;
  float s = 0.f;
  for (int c = 0; c < n_mel; ++c) s += mp[(long long)c * T];
  float e = s / (float)n_mel;
  if (scaled) e = (e + 20.0f) / 20.0f;
  out[(long long)b * T + t] = e;
}

}  // namespace

extern "C" int radmmm_betabinom_prior(int P, int M, double scaling, double* out, void* stream) {
  RADMMM_REQUIRE(out, "betabinom_prior: null pointer");
  RADMMM_REQUIRE(P > 0 && M > 0 && scaling > 0.0, "betabinom_prior: bad dims");
  const long long n = (long long)P * M;
  hipLaunchKernelGGL(betabinom_prior_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     P, M, scaling, out);
  return radmmm::check_launch("betabinom_prior");
}

extern "C" int radmmm_prior_zoom_batch(const int64_t* items, int B, float* out, int Tmax, int Nmax, void* stream) {
  static_assert(sizeof(ZoomItem) == 5 * sizeof(int64_t), "item layout: {bank pointer, bh, bw, m, p} as 5 x int64");
  RADMMM_REQUIRE(items && out, "prior_zoom_batch: null pointer");
  RADMMM_REQUIRE(B > 0 && Tmax > 0 && Nmax > 0 && B <= 65535, "prior_zoom_batch: bad dims");
  hipLaunchKernelGGL(prior_zoom_kernel, dim3(Tmax, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const ZoomItem*>(items), out, Tmax, Nmax);
  return radmmm::check_launch("prior_zoom_batch");
}

extern "C" int radmmm_energy_average(const float* mel, float* out, int B, int n_mel, int T, int scaled, void* stream) {
  RADMMM_REQUIRE(mel && out, "energy_average: null pointer");
  RADMMM_REQUIRE(B > 0 && n_mel > 0 && T > 0 && B <= 65535, "energy_average: bad dims");
  hipLaunchKernelGGL(energy_average_kernel, dim3((T + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream), mel, out,
                     n_mel, T, scaled);
  return radmmm::check_launch("energy_average");
}
