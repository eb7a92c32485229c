// FiLM residual block tail (FiLMResBlock.forward, common.py:728-735) with the masked batch-norm
// of maskedbatchnorm1d.py:53-118 fused in:
//   y   = use_bn ? (h2 - mean) * invstd * w + b : h2
//   t   = y * (c1[:, 0:C] + 1) + c1[:, C:2C]
//   out = 0.5 * (leaky_relu(t) + x1r)
// and its gradient.  HBM-bound elementwise work on [rows, C] fp32 matrices; the two column
// reductions the batch-norm gradient needs (sum g_y, sum g_y * xhat) are produced by a partial
// kernel (float4 columns, 64-row chunks) + a final add, deterministic.
#include "common.h"

extern "C" int radmmm_colsum_final(const float* part, float* out, int nparts, int cols, radmmm_stream_t stream);

namespace {

constexpr int FR = 64;  // rows per reduction block

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.01f * x; }

__global__ __launch_bounds__(256) void film_fwd_kernel(
    const float* __restrict__ h2, int ldh, const float* __restrict__ c1, int ldc, const float* __restrict__ x1r,
    int ldx, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ w,
    const float* __restrict__ b, float* __restrict__ out, int ldo, int rows, int C, int use_bn) {
  const long long total = (long long)rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i - (long long)r * C);
    float y = h2[(long long)r * ldh + c];
    if (use_bn) y = (y - mean[c]) * invstd[c] * w[c] + b[c];
    const float t = y * (c1[(long long)r * ldc + c] + 1.f) + c1[(long long)r * ldc + C + c];
    out[(long long)r * ldo + c] = 0.5f * (leaky(t) + x1r[(long long)r * ldx + c]);
  }
}

// g_y for one element (and xhat / y as by-products)
__device__ __forceinline__ float film_gy(float h, float ca, float cb, float go, float mean, float invstd,
                                         float w, float b, int use_bn, float& xhat, float& y, float& gt) {
  xhat = use_bn ? (h - mean) * invstd : h;
  y = use_bn ? xhat * w + b : h;
  const float t = y * (ca + 1.f) + cb;
  gt = 0.5f * go * (t > 0.f ? 1.f : 0.01f);
  return gt * (ca + 1.f);
}

__global__ __launch_bounds__(256) void film_bwd_reduce_kernel(
    const float* __restrict__ h2, int ldh, const float* __restrict__ c1, int ldc, const float* __restrict__ gout,
    int ldg, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ w,
    const float* __restrict__ b, float* __restrict__ part, int rows, int C) {
  // part[chunk][2][C]: sum g_y, sum g_y * xhat over the chunk's rows
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * FR;
  int r1 = r0 + FR;
  if (r1 > rows) r1 = rows;
  if (c >= C) return;
  const float m = mean[c], is = invstd[c], ww = w[c], bb = b[c];
  float s1 = 0.f, s2 = 0.f;
  for (int r = r0; r < r1; ++r) {
    float xhat, y, gt;
    const float gy = film_gy(h2[(long long)r * ldh + c], c1[(long long)r * ldc + c], c1[(long long)r * ldc + C + c],
                             gout[(long long)r * ldg + c], m, is, ww, bb, 1, xhat, y, gt);
    s1 += gy;
    s2 = fmaf(gy, xhat, s2);
  }
  part[((long long)blockIdx.y * 2 + 0) * C + c] = s1;
  part[((long long)blockIdx.y * 2 + 1) * C + c] = s2;
}

__global__ __launch_bounds__(256) void film_bwd_apply_kernel(
    const float* __restrict__ h2, int ldh, const float* __restrict__ c1, int ldc, const float* __restrict__ gout,
    int ldg, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ w,
    const float* __restrict__ b, const float* __restrict__ S, float inv_n, int T, const int* __restrict__ lens,
    float* __restrict__ gh2, int ldgh, float* __restrict__ gc1, int ldgc, float* __restrict__ gx1r, int ldgx,
    int rows, int C, int use_bn) {
  const long long total = (long long)rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i - (long long)r * C);
    const float go = gout[(long long)r * ldg + c];
    float xhat, y, gt;
    const float m = use_bn ? mean[c] : 0.f, is = use_bn ? invstd[c] : 1.f;
    const float ww = use_bn ? w[c] : 1.f, bb = use_bn ? b[c] : 0.f;
    const float gy = film_gy(h2[(long long)r * ldh + c], c1[(long long)r * ldc + c], c1[(long long)r * ldc + C + c],
                             go, m, is, ww, bb, use_bn, xhat, y, gt);
    gc1[(long long)r * ldgc + c] = gt * y;
    gc1[(long long)r * ldgc + C + c] = gt;
    gx1r[(long long)r * ldgx + c] = 0.5f * go;
    float g = gy;
    if (use_bn) {
      // dL/dh = invstd * w * gy + mask/n * ( -invstd * w * S1 - invstd * w * xhat * S2 )
      const int bi = r / T, t = r - bi * T;
      const float mk = t < (lens ? lens[bi] : T) ? 1.f : 0.f;
      g = is * ww * (gy - mk * inv_n * (S[c] + xhat * S[C + c]));
    }
    gh2[(long long)r * ldgh + c] = g;
  }
}

inline int grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int radmmm_film_fwd(const float* h2, int ldh, const float* c1, int ldc, const float* x1r, int ldx,
                               const float* mean, const float* invstd, const float* w, const float* b, float* out,
                               int ldo, int rows, int C, int use_bn, radmmm_stream_t stream) {
  RADMMM_REQUIRE(h2 && c1 && x1r && out, "film_fwd: null pointer");
  RADMMM_REQUIRE(!use_bn || (mean && invstd && w && b), "film_fwd: batch-norm operands missing");
  RADMMM_REQUIRE(rows > 0 && C > 0 && ldh >= C && ldc >= 2 * C && ldx >= C && ldo >= C, "film_fwd: bad dims");
  hipLaunchKernelGGL(film_fwd_kernel, dim3(grid_for((long long)rows * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h2, ldh, c1, ldc, x1r, ldx, mean, invstd, w, b, out, ldo, rows,
                     C, use_bn);
  return radmmm::check_launch("film_fwd");
}

extern "C" int64_t radmmm_film_bwd_scratch_floats(int rows, int C) {
  return (int64_t)((rows + FR - 1) / FR) * 2 * C + 2 * C;
}

// The two halves of radmmm_film_bwd, for synchronised masked batch-norm under data parallelism
// (maskedbatchnorm1d.py:88-95 all-reduces the statistics with an autograd-aware collective, i.e. forward AND backward):
// _sums leaves S = [sum g_y | sum g_y xhat] of THIS rank's frames at scratch + nparts*2*C (and copies it to gb / gw: the
// affine parameters' gradients stay local sums, DDP averages them like any other parameter gradient); the caller
// all-reduces S in place; _apply then uses the global S with n_valid = the global frame count.
extern "C" int radmmm_film_bwd_sums(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                                    const float* mean, const float* invstd, const float* w, const float* b, float* gw,
                                    float* gb, float* scratch, int rows, int C, radmmm_stream_t stream) {
  RADMMM_REQUIRE(h2 && c1 && gout && scratch && mean && invstd && w && b && gw && gb, "film_bwd_sums: null pointer");
  RADMMM_REQUIRE(rows > 0 && C > 0 && ldh >= C && ldc >= 2 * C && ldg >= C, "film_bwd_sums: bad dims");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nparts = (rows + FR - 1) / FR;
  float* S = scratch + (long long)nparts * 2 * C;
  hipLaunchKernelGGL(film_bwd_reduce_kernel, dim3((C + 255) / 256, nparts), dim3(256), 0, s, h2, ldh, c1, ldc, gout, ldg,
                     mean, invstd, w, b, scratch, rows, C);
  // S[i] = sum_p part[p][i] over the 2 C columns: the column-sum finisher (16 waves per 64 columns over the partials, fixed
    // order) -- one thread per column walking all rows / 64 partials took 116 us at 32 000 rows
    if (int rc = radmmm_colsum_final(scratch, S, nparts, 2 * C, stream)) return rc;
  if (hipMemcpyAsync(gb, S, sizeof(float) * C, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(gw, S + C, sizeof(float) * C, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    radmmm::set_error("film_bwd_sums: hipMemcpyAsync failed");
    return -2;
  }
  return radmmm::check_launch("film_bwd_sums");
}

extern "C" int radmmm_film_bwd_apply(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                                     const float* mean, const float* invstd, const float* w, const float* b,
                                     float n_valid, int T, const int32_t* lens, float* gh2, int ldgh, float* gc1, int ldgc,
                                     float* gx1r, int ldgx, const float* scratch, int rows, int C,
                                     radmmm_stream_t stream) {
  RADMMM_REQUIRE(h2 && c1 && gout && gh2 && gc1 && gx1r && scratch && mean && invstd && w && b, "film_bwd_apply: null pointer");
  RADMMM_REQUIRE(rows > 0 && C > 0 && ldh >= C && ldc >= 2 * C && ldg >= C && ldgh >= C && ldgc >= 2 * C && ldgx >= C &&
                     n_valid > 0 && T > 0 && rows % T == 0, "film_bwd_apply: bad dims");
  const int nparts = (rows + FR - 1) / FR;
  const float* S = scratch + (long long)nparts * 2 * C;
  hipLaunchKernelGGL(film_bwd_apply_kernel, dim3(grid_for((long long)rows * C)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h2, ldh, c1, ldc, gout, ldg, mean, invstd, w, b, S, 1.f / n_valid, T,
                     lens, gh2, ldgh, gc1, ldgc, gx1r, ldgx, rows, C, 1);
  return radmmm::check_launch("film_bwd_apply");
}

extern "C" int radmmm_film_bwd(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                               const float* mean, const float* invstd, const float* w, const float* b, float n_valid,
                               int T, const int32_t* lens, float* gh2, int ldgh, float* gc1, int ldgc, float* gx1r,
                               int ldgx, float* gw, float* gb, float* scratch, int rows, int C, int use_bn,
                               radmmm_stream_t stream) {
  RADMMM_REQUIRE(h2 && c1 && gout && gh2 && gc1 && gx1r && scratch, "film_bwd: null pointer");
  RADMMM_REQUIRE(!use_bn || (mean && invstd && w && b && gw && gb && n_valid > 0 && T > 0 && rows % T == 0),
                 "film_bwd: batch-norm operands missing");
  RADMMM_REQUIRE(rows > 0 && C > 0 && ldh >= C && ldc >= 2 * C && ldg >= C && ldgh >= C && ldgc >= 2 * C && ldgx >= C,
                 "film_bwd: bad dims");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nparts = (rows + FR - 1) / FR;
  float* S = scratch + (long long)nparts * 2 * C;
  if (use_bn) {
    hipLaunchKernelGGL(film_bwd_reduce_kernel, dim3((C + 255) / 256, nparts), dim3(256), 0, s, h2, ldh, c1, ldc, gout,
                       ldg, mean, invstd, w, b, scratch, rows, C);
    // S[i] = sum_p part[p][i] over the 2 C columns: the column-sum finisher (16 waves per 64 columns over the partials, fixed
    // order) -- one thread per column walking all rows / 64 partials took 116 us at 32 000 rows
    if (int rc = radmmm_colsum_final(scratch, S, nparts, 2 * C, stream)) return rc;
    // dL/db = S1, dL/dw = S2
    if (hipMemcpyAsync(gb, S, sizeof(float) * C, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(gw, S + C, sizeof(float) * C, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      radmmm::set_error("film_bwd: hipMemcpyAsync failed");
      return -2;
    }
  }
  hipLaunchKernelGGL(film_bwd_apply_kernel, dim3(grid_for((long long)rows * C)), dim3(256), 0, s, h2, ldh, c1, ldc,
                     gout, ldg, mean, invstd, w, b, S, use_bn ? 1.f / n_valid : 0.f, T > 0 ? T : 1, lens, gh2, ldgh, gc1,
                     ldgc, gx1r, ldgx, rows, C, use_bn);
  return radmmm::check_launch("film_bwd");
}
