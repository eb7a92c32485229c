// Shared helpers for libradmmm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/radmmm_hip.h"

namespace radmmm {

void set_error(const char* fmt, ...);
// value of an experiment / test switch, or nullptr unless RADMMM_DEBUG=1 (error.cpp)
const char* debug_env(const char* name);

// rowgemm16_f32.hip: 16-row-granular tiling of radmmm_rowgemm_f32 (descriptor already validated)
int launch_rowgemm16(const radmmm_rowgemm_desc& d, hipStream_t stream);
int launch_rowgemm_mix(const radmmm_rowgemm_desc& d, hipStream_t stream);   // rowgemm_mix.hip: 0 launched, 1 not its shape
// wgrad16_f32.hip: fast path of radmmm_wgrad_f32 (0 launched, <0 error, 1 not applicable)
int launch_wgrad16(const radmmm_wgrad_desc& d, hipStream_t stream);

// rowgemm_h3w.hip: wide-tile (32*MB x 256, one workgroup per CU) split-f16 conv GEMM
int launch_rowgemm_h3w(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes);
int gemm_cu_slots();

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

#define RADMMM_REQUIRE(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      radmmm::set_error(__VA_ARGS__);  \
      return -1;                       \
    }                                  \
  } while (0)

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device helpers -------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. for the
// write acknowledgements of every global store issued so far (stores count in vmcnt on gfx9): in an
// epilogue that alternates "stage a block in LDS / store it" that costs one HBM write round trip (2-5 us)
// per block.  Use only where no global-memory hand-off between the waves depends on the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float softplus_f(float x) {
  // torch.nn.Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x)) = max(x, 0) + log1p(exp(-|x|)).
  // Hardware exp2/log2 (v_exp_f32 / v_log_f32, 1 ulp) instead of libm's expf/log1pf: ~12 VALU
  // instructions instead of ~100, which matters because the GEMM epilogues are VALU bound.
  // log1p(e) = log(u) * e / (u - 1) with u = fl(1 + e) keeps full relative accuracy for small e.
  const float e = __expf(-fabsf(x));
  const float u = 1.f + e;
  const float d = u - 1.f;
  const float l = (d == 0.f) ? e : __logf(u) * __fdividef(e, d);
  return x > 20.f ? x : fmaxf(x, 0.f) + l;
}
// 1 - exp(-y) for y >= 0 without cancellation for small y
__device__ __forceinline__ float one_minus_exp_neg(float y) {
  const float p = y * (1.f - y * (0.5f - y * (0.16666667f - y * (0.041666668f - y * (0.0083333338f - y * 0.0013888889f)))));
  return y < 0.25f ? p : 1.f - __expf(-y);
}
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case RADMMM_ACT_SOFTPLUS: return softplus_f(v);
    case RADMMM_ACT_RELU: return v > 0.f ? v : 0.f;
    case RADMMM_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
    default: return v;
  }
}
// derivative of the activation expressed from its OUTPUT y
__device__ __forceinline__ float dact_from_out(float y, int act) {
  switch (act) {
    case RADMMM_ACT_SOFTPLUS: return y > 20.f ? 1.f : one_minus_exp_neg(y);  // sigmoid(x) = 1 - exp(-softplus(x))
    case RADMMM_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case RADMMM_ACT_LEAKY: return y > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}
// partial-conv renormalisation ratio for frame t of an item with `len` valid frames
// (partialconv1d.py:75-81): taps/(cnt+1e-6) * clamp(cnt,0,1)
__device__ __forceinline__ float pconv_ratio(int t, int len, int taps, int dil) {
  int cnt = 0;
  const int c = taps / 2;
  for (int k = 0; k < taps; ++k) {
    const int ts = t + (k - c) * dil;
    cnt += (ts >= 0 && ts < len) ? 1 : 0;
  }
  return cnt > 0 ? (float)taps / ((float)cnt + 1e-6f) : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in all threads
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 17 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (w == 0) {
    float t = lane < nw ? sh[lane] : 0.f;
    t = wave_sum(t);
    if (lane == 0) sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

// weight of row r in a (bias-gradient) column sum: 0 none, 1 length mask, 2 mask x partial-conv ratio
__device__ __forceinline__ float colsum_row_weight(int r, int row_weight, int T, const int* lens, int taps,
                                                   int dil) {
  if (!row_weight) return 1.f;
  const int b = r / T, t = r - b * T;
  const int len = lens ? lens[b] : T;
  if (t >= len) return 0.f;
  if (row_weight != 2) return 1.f;
  int cnt = 0;
  for (int k = 0; k < taps; ++k) {
    const int ts = t + (k - taps / 2) * dil;
    cnt += (ts >= 0 && ts < len) ? 1 : 0;
  }
  return ((float)cnt + 1e-6f) / (float)taps;
}

}  // namespace radmmm
