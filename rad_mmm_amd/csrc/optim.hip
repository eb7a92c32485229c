// Optimizer step on flat fp32 buffers: the reference's vendored RAdam
// (radam.py:63-142) fused with Lightning's global-norm gradient clip
// (configs/RADMMM_train_config.yaml:7-8).  Parameters, gradients and both moments of one
// bucket are contiguous arrays (rad_mmm_amd/optim.py makes the parameters views of a flat buffer,
// as ddp.py already does for the gradients), so the whole update of 26.5 M parameters is one
// streaming kernel: 4 reads + 3 writes per element, HBM bound.
#include "common.h"

namespace {

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float sh[17];
  float s = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      s = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, s))));
    } else {
      for (long long k = i; k < n; ++k) s = fmaf(x[k], x[k], s);
    }
  }
  s = radmmm::block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// clip[0] (device scalar, may be null) multiplies the gradient; use_denom: N_sma >= 5 branch
struct RAdamCoef {
  float c, beta1, beta2, eps, step_size, wd_lr;
  int use_denom;
};
__device__ __forceinline__ void radam_one(float& pi, float gi, float& mi, float& vi, const RAdamCoef& k) {
  gi *= k.c;
  vi = vi * k.beta2 + (1.f - k.beta2) * gi * gi;
  mi = mi * k.beta1 + (1.f - k.beta1) * gi;
  pi += -k.wd_lr * pi;
  pi += k.use_denom ? -k.step_size * (mi / (sqrtf(vi) + k.eps)) : -k.step_size * mi;
}
// 28 bytes of HBM traffic per parameter and nothing else: four parameters per lane and access (16-byte loads / stores,
// streaming: every byte is touched once per step and the four arrays together are far larger than the last-level cache)
__global__ __launch_bounds__(256) void radam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    const float* __restrict__ clip, float beta1, float beta2, float eps,
                                                    float step_size, float wd_lr, int use_denom) {
  const RAdamCoef k{clip ? clip[0] : 1.f, beta1, beta2, eps, step_size, wd_lr, use_denom};
  const long long stride = (long long)gridDim.x * blockDim.x, tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n4 = n >> 2;
  using f4 = __attribute__((ext_vector_type(4))) float;
  f4* p4 = reinterpret_cast<f4*>(p);
  f4* m4 = reinterpret_cast<f4*>(m);
  f4* v4 = reinterpret_cast<f4*>(v);
  const f4* g4 = reinterpret_cast<const f4*>(g);
  for (long long i = tid; i < n4; i += stride) {
    const f4 gi = __builtin_nontemporal_load(g4 + i);
    const f4 pv = __builtin_nontemporal_load(p4 + i), mv = __builtin_nontemporal_load(m4 + i), vv = __builtin_nontemporal_load(v4 + i);
    float pj[4], mj[4], vj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pj[j] = pv[j];
      mj[j] = mv[j];
      vj[j] = vv[j];
      radam_one(pj[j], gi[j], mj[j], vj[j], k);
    }
    __builtin_nontemporal_store(f4{vj[0], vj[1], vj[2], vj[3]}, v4 + i);
    __builtin_nontemporal_store(f4{mj[0], mj[1], mj[2], mj[3]}, m4 + i);
    __builtin_nontemporal_store(f4{pj[0], pj[1], pj[2], pj[3]}, p4 + i);
  }
  for (long long i = (n4 << 2) + tid; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    radam_one(pi, g[i], mi, vi, k);
    v[i] = vi;
    m[i] = mi;
    p[i] = pi;
  }
}

}  // namespace

extern "C" int64_t radmmm_sumsq_scratch_floats(void) { return SUMSQ_BLOCKS; }

// partial[0 .. radmmm_sumsq_scratch_floats()) = per-block sums of x^2 (the caller adds them; no atomics)
extern "C" int radmmm_sumsq(const float* x, int64_t n, float* partial, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && partial && n > 0, "sumsq: bad arguments");
  RADMMM_REQUIRE(radmmm::aligned16(x), "sumsq: x must be 16B aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)n, partial);
  return radmmm::check_launch("sumsq");
}

// One RAdam update of a flat buffer.  step_size and the N_sma >= 5 switch are computed by the
// host from the step count exactly as radam.py:101-123; wd_lr = weight_decay * lr;
// clip_coef: optional device scalar applied to the gradient (global-norm clip), null = 1.
extern "C" int radmmm_radam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* clip_coef,
                                 float beta1, float beta2, float eps, float step_size, float wd_lr, int use_denom,
                                 radmmm_stream_t stream) {
  RADMMM_REQUIRE(p && g && m && v && n > 0, "radam_step: bad arguments");
  RADMMM_REQUIRE(radmmm::aligned16(p) && radmmm::aligned16(g) && radmmm::aligned16(m) && radmmm::aligned16(v),
                 "radam_step: the flat buffers must be 16B aligned");
  const long long blocks = (n / 4 + 255) / 256 + 1;
  hipLaunchKernelGGL(radam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, g, m, v, (long long)n, clip_coef, beta1, beta2, eps, step_size,
                     wd_lr, use_denom);
  return radmmm::check_launch("radam_step");
}
