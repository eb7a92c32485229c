// Masked InstanceNorm1d (+ optional ReLU) on channels-last rows, for the text encoder.
// The reference normalises each utterance separately inside a per-item Python loop
// (common.py:476-484: ConvNorm -> nn.InstanceNorm1d(affine=True) -> ReLU on x[b, :, :len_b]); here
// the whole padded batch is one launch: statistics over the frames t < lens[b] of item b (biased
// variance, eps inside the rsqrt, as torch), frames >= lens[b] are written as zeros.
// x, y, gy, gx: [B*T][ld] fp32, first C columns; mean, rstd: [B][C].
#include "common.h"

namespace {

constexpr int CH = 64, RL = 4;          // block = 64 channels x 4 row lanes

__device__ __forceinline__ float reduce_rl(float v, float (*sh)[CH], int rl, int cl) {
  __syncthreads();
  sh[rl][cl] = v;
  __syncthreads();
  return (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            const int* __restrict__ lens, int T, int C, float eps, int relu) {
  __shared__ float sh[RL][CH];
  const int cl = threadIdx.x & (CH - 1), rl = threadIdx.x >> 6;
  const int c = blockIdx.x * CH + cl, b = blockIdx.y;
  const bool ok = c < C;
  int len = lens ? lens[b] : T;
  len = len < 0 ? 0 : (len > T ? T : len);
  const float* xb = x + (long long)b * T * ldx + c;
  float s = 0.f;
  if (ok)
    for (int t = rl; t < len; t += RL) s += xb[(long long)t * ldx];
  const float n = (float)(len > 0 ? len : 1);
  const float mu = reduce_rl(s, sh, rl, cl) / n;
  float q = 0.f;
  if (ok)
    for (int t = rl; t < len; t += RL) {
      const float d = xb[(long long)t * ldx] - mu;
      q = fmaf(d, d, q);
    }
  const float rs = rsqrtf(reduce_rl(q, sh, rl, cl) / n + eps);
  if (!ok) return;
  if (rl == 0) {
    mean[(long long)b * C + c] = mu;
    rstd[(long long)b * C + c] = rs;
  }
  const float ww = w ? w[c] : 1.f, bb = bias ? bias[c] : 0.f;
  float* yb = y + (long long)b * T * ldy + c;
  for (int t = rl; t < T; t += RL) {
    float v = 0.f;
    if (t < len) {
      v = (xb[(long long)t * ldx] - mu) * rs * ww + bb;
      if (relu) v = v > 0.f ? v : 0.f;
    }
    yb[(long long)t * ldy] = v;
  }
}

// gx; dwp/dbp [B][C] per-item partial sums of the affine parameters' gradients
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ gy, int ldg, const float* __restrict__ x,
                                                            int ldx, const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ gx, int ldgx,
                                                            float* __restrict__ dwp, float* __restrict__ dbp,
                                                            const int* __restrict__ lens, int T, int C, int relu) {
  __shared__ float sh[RL][CH];
  const int cl = threadIdx.x & (CH - 1), rl = threadIdx.x >> 6;
  const int c = blockIdx.x * CH + cl, b = blockIdx.y;
  const bool ok = c < C;
  int len = lens ? lens[b] : T;
  len = len < 0 ? 0 : (len > T ? T : len);
  const long long r0 = (long long)b * T;
  const float mu = ok ? mean[(long long)b * C + c] : 0.f, rs = ok ? rstd[(long long)b * C + c] : 0.f;
  const float ww = (ok && w) ? w[c] : 1.f;
  float sg = 0.f, sgx = 0.f;
  if (ok)
    for (int t = rl; t < len; t += RL) {
      float g = gy[(r0 + t) * ldg + c];
      if (relu && !(y[(r0 + t) * ldy + c] > 0.f)) g = 0.f;
      const float xh = (x[(r0 + t) * ldx + c] - mu) * rs;
      sg += g;
      sgx = fmaf(g, xh, sgx);
    }
  const float n = (float)(len > 0 ? len : 1);
  const float tg = reduce_rl(sg, sh, rl, cl), tgx = reduce_rl(sgx, sh, rl, cl);
  if (!ok) return;
  if (rl == 0) {
    dbp[(long long)b * C + c] = tg;
    dwp[(long long)b * C + c] = tgx;
  }
  const float m1 = tg * ww / n, m2 = tgx * ww / n;
  for (int t = rl; t < T; t += RL) {
    float v = 0.f;
    if (t < len) {
      float g = gy[(r0 + t) * ldg + c];
      if (relu && !(y[(r0 + t) * ldy + c] > 0.f)) g = 0.f;
      const float xh = (x[(r0 + t) * ldx + c] - mu) * rs;
      v = rs * (g * ww - m1 - xh * m2);
    }
    gx[(r0 + t) * ldgx + c] = v;
  }
}

}  // namespace

extern "C" int radmmm_instnorm_fwd(const float* x, int ldx, const float* weight, const float* bias, float* y, int ldy,
                                   float* mean, float* rstd, const int32_t* lens, int B, int T, int C, float eps, int relu,
                                   radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && y && mean && rstd, "instnorm_fwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && C > 0 && ldx >= C && ldy >= C, "instnorm_fwd: bad dims");
  hipLaunchKernelGGL(instnorm_fwd_kernel, dim3((C + CH - 1) / CH, B), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                     weight, bias, y, ldy, mean, rstd, lens, T, C, eps, relu);
  return radmmm::check_launch("instnorm_fwd");
}

extern "C" int radmmm_instnorm_bwd(const float* gy, int ldg, const float* x, int ldx, const float* y, int ldy,
                                   const float* weight, const float* mean, const float* rstd, float* gx, int ldgx,
                                   float* dw_part, float* db_part, const int32_t* lens, int B, int T, int C, int relu,
                                   radmmm_stream_t stream) {
  RADMMM_REQUIRE(gy && x && y && mean && rstd && gx && dw_part && db_part, "instnorm_bwd: null pointer");
  RADMMM_REQUIRE(B > 0 && T > 0 && C > 0 && ldg >= C && ldx >= C && ldy >= C && ldgx >= C, "instnorm_bwd: bad dims");
  hipLaunchKernelGGL(instnorm_bwd_kernel, dim3((C + CH - 1) / CH, B), dim3(256), 0, static_cast<hipStream_t>(stream), gy, ldg,
                     x, ldx, y, ldy, weight, mean, rstd, gx, ldgx, dw_part, db_part, lens, T, C, relu);
  return radmmm::check_launch("instnorm_bwd");
}
