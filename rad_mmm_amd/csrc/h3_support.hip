// Support kernels of the split-f16 GEMM path: weight-norm fold straight into split (hi/lo fp16)
// packed weights, and a tiled transpose of a split pair (the data-gradient GEMM wants the weights
// K-contiguous in the OUTPUT-channel index).
#include "common.h"

namespace {

using radmmm::block_sum;

__device__ __forceinline__ void split1(float t, _Float16& h, _Float16& l) {
  t = fminf(fmaxf(t, -60000.f), 60000.f);
  h = (_Float16)t;
  l = (_Float16)(t - (float)h);
}

// W{h,l}[tap][co][col(ci)] = split(scale * g[co] * v[co][ci][tap] / ||v[co]||); columns not hit by
// col() must have been zeroed by the caller.  g == NULL: plain conv weights (no normalisation).
__global__ __launch_bounds__(256) void weightnorm_fwd_h3_kernel(
    const float* __restrict__ v, const float* __restrict__ g, _Float16* __restrict__ Wh, _Float16* __restrict__ Wl,
    float* __restrict__ inv_norm, int Cout, int Cin, int taps, int ldk, int perm_split, int off_lo, int off_hi,
    float scale) {
  __shared__ float sh[17];
  const int co = blockIdx.x;
  const int n = Cin * taps;
  const float* vr = v + (long long)co * n;
  float wn = 1.f;                      // g / ||v||  (1 for plain weights)
  if (g) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss = fmaf(vr[i], vr[i], ss);
    ss = block_sum(ss, sh);
    const float nrm = sqrtf(ss);
    if (threadIdx.x == 0) inv_norm[co] = 1.f / nrm;
    wn = g[co] / nrm;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int ci = i / taps, k = i - ci * taps;
    const int col = ci < perm_split ? ci + off_lo : ci - perm_split + off_hi;
    _Float16 h, l;
    // same rounding as the fp32 path (w = v * (g / ||v||)), then the exact power-of-two scale
    split1((vr[i] * wn) * scale, h, l);
    const long long o = ((long long)k * Cout + co) * ldk + col;
    Wh[o] = h;
    Wl[o] = l;
  }
}

// dst[b][c][r] = src[b][r][c] for both members of a split pair; 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_pair_kernel(const _Float16* __restrict__ sh_, const _Float16* __restrict__ sl_,
                                                             int ld_src, long long src_batch, _Float16* __restrict__ dh,
                                                             _Float16* __restrict__ dl, int ld_dst, long long dst_batch,
                                                             int rows, int cols) {
  __shared__ _Float16 th[32][34], tl[32][34];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const bool ok = r < rows && c < cols;
    th[i][tx] = ok ? sh_[b * src_batch + (long long)r * ld_src + c] : (_Float16)0.f;
    tl[i][tx] = ok ? sl_[b * src_batch + (long long)r * ld_src + c] : (_Float16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) {
      dh[b * dst_batch + (long long)c * ld_dst + r] = th[tx][i];
      dl[b * dst_batch + (long long)c * ld_dst + r] = tl[tx][i];
    }
  }
}

}  // namespace

extern "C" int radmmm_weightnorm_fwd_h3(const float* v, const float* g, void* Wh, void* Wl, float* inv_norm, int Cout,
                                        int Cin, int taps, int ldk, int perm_split, int off_lo, int off_hi, float scale,
                                        radmmm_stream_t stream) {
  RADMMM_REQUIRE(v && Wh && Wl && (inv_norm || !g), "weightnorm_fwd_h3: null pointer");
  RADMMM_REQUIRE(Cout > 0 && Cin > 0 && taps > 0 && ldk >= Cin && ldk % 8 == 0, "weightnorm_fwd_h3: bad dims");
  hipLaunchKernelGGL(weightnorm_fwd_h3_kernel, dim3(Cout), dim3(256), 0, static_cast<hipStream_t>(stream), v, g,
                     static_cast<_Float16*>(Wh), static_cast<_Float16*>(Wl), inv_norm, Cout, Cin, taps, ldk, perm_split,
                     off_lo, off_hi, scale);
  return radmmm::check_launch("weightnorm_fwd_h3");
}

extern "C" int radmmm_transpose_f16_pair(const void* src_h, const void* src_l, int ld_src, int64_t src_batch, void* dst_h,
                                         void* dst_l, int ld_dst, int64_t dst_batch, int batches, int rows, int cols,
                                         radmmm_stream_t stream) {
  RADMMM_REQUIRE(src_h && src_l && dst_h && dst_l, "transpose_f16_pair: null pointer");
  RADMMM_REQUIRE(batches > 0 && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "transpose_f16_pair: bad dims");
  hipLaunchKernelGGL(transpose_pair_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batches), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(src_h),
                     static_cast<const _Float16*>(src_l), ld_src, (long long)src_batch, static_cast<_Float16*>(dst_h),
                     static_cast<_Float16*>(dst_l), ld_dst, (long long)dst_batch, rows, cols);
  return radmmm::check_launch("transpose_f16_pair");
}
