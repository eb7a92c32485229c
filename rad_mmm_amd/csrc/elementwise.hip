// HBM-bound kernels of the flow step: weight-norm fold/unfold, WN input assembly, affine
// coupling, activation-derivative products, column sums (bias gradients) and the masked
// NLL reductions.  All fp32, channels-last rows; every kernel keeps consecutive lanes on
// consecutive addresses (coalesced 256 B per wave-instruction or float4 where the layout
// allows) and reduces through wavefront shuffles (64 lanes) before touching LDS.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "split_pack.h"

namespace {

using radmmm::block_sum;
using radmmm::wave_sum;

__device__ __forceinline__ int perm_col(int ci, int perm_split, int off_lo, int off_hi) {
  return ci < perm_split ? ci + off_lo : ci - perm_split + off_hi;
}

// ------------------------------------------------------------------ weight norm
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(
    const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ W,
    float* __restrict__ inv_norm, int Cout, int Cin, int taps, int ldw, int perm_split,
    int off_lo, int off_hi) {
  __shared__ float sh[17];
  const int co = blockIdx.x;
  const int n = Cin * taps;
  const float* vr = v + (long long)co * n;
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) ss = fmaf(vr[i], vr[i], ss);
  ss = block_sum(ss, sh);
  const float nrm = sqrtf(ss);
  const float scale = g[co] / nrm;
  if (threadIdx.x == 0) inv_norm[co] = 1.f / nrm;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int ci = i / taps, k = i - ci * taps;
    W[((long long)k * Cout + co) * ldw + perm_col(ci, perm_split, off_lo, off_hi)] = vr[i] * scale;
  }
}

__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(
    const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ inv_norm,
    const float* __restrict__ dW, int splits, long long split_stride, float* __restrict__ dv,
    float* __restrict__ dg, int Cout, int Cin, int taps, int ldw, int perm_split, int off_lo,
    int off_hi, const float* __restrict__ poison) {
  __shared__ float sh[17];
  const int co = blockIdx.x;
  const int n = Cin * taps;
  const float* vr = v + (long long)co * n;
  auto gw_at = [&](int i) {
    const int ci = i / taps, k = i - ci * taps;
    const long long off = ((long long)k * Cout + co) * ldw + perm_col(ci, perm_split, off_lo, off_hi);
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += dW[sp * split_stride + off];
    return s;
  };
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dot = fmaf(gw_at(i), vr[i], dot);
  dot = block_sum(dot, sh);
  const float inv = inv_norm[co], gg = g[co];
  if (threadIdx.x == 0) dg[co] = dot * inv + (poison ? poison[0] : 0.f);   // poison: 0, or NaN after a non-finite upstream gradient
  const float a = gg * inv, b = gg * dot * inv * inv * inv;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    dv[(long long)co * n + i] = a * gw_at(i) - b * vr[i];
}

// Same result with one coalesced pass over the gradient slabs: the summed gradient row of this
// output channel is staged in LDS in checkpoint order ([ci][tap]), reading dW plane by plane with ci
// fastest (the kernel above walks [ci][tap] directly and touches `taps` planes per 64-byte sector,
// twice).  Needs Cin * taps * 4 bytes of LDS.
template <bool VEC>
__global__ __launch_bounds__(256) void weightnorm_bwd_lds_kernel(
    const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ inv_norm,
    const float* __restrict__ dW, int splits, long long split_stride, float* __restrict__ dv,
    float* __restrict__ dg, int Cout, int Cin, int taps, int ldw, int perm_split, int off_lo,
    int off_hi, const float* __restrict__ poison) {
  extern __shared__ __attribute__((aligned(16))) float gw[];          // [Cin * taps]
  __shared__ float sh[17];
  const int co = blockIdx.x;
  const int n = Cin * taps;
  const float* vr = v + (long long)co * n;
  for (int k = 0; k < taps; ++k) {
    const float* plane = dW + ((long long)k * Cout + co) * ldw;
    if constexpr (VEC) {
      // 16-byte loads of 4 consecutive input channels (the host checked that the column permutation keeps groups
      // of 4 together and aligned); the split-K slabs are independent loads, issued back to back
      for (int ci = threadIdx.x * 4; ci < Cin; ci += blockDim.x * 4) {
        const int col = perm_col(ci, perm_split, off_lo, off_hi);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int sp = 0; sp < splits; ++sp) {
          const float4 t = *reinterpret_cast<const float4*>(plane + sp * split_stride + col);
          s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
        }
        gw[ci * taps + k] = s0.x;
        gw[(ci + 1) * taps + k] = s0.y;
        gw[(ci + 2) * taps + k] = s0.z;
        gw[(ci + 3) * taps + k] = s0.w;
      }
    } else {
      for (int ci = threadIdx.x; ci < Cin; ci += blockDim.x) {
        const int col = perm_col(ci, perm_split, off_lo, off_hi);
        float s0 = 0.f;
        for (int sp = 0; sp < splits; ++sp) s0 += plane[sp * split_stride + col];
        gw[ci * taps + k] = s0;
      }
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dot = fmaf(gw[i], vr[i], dot);
  dot = block_sum(dot, sh);
  const float inv = inv_norm[co], gg = g[co];
  if (threadIdx.x == 0) dg[co] = dot * inv + (poison ? poison[0] : 0.f);   // poison: 0, or NaN after a non-finite upstream gradient
  const float a = gg * inv, b = gg * dot * inv * inv * inv;
  if (VEC && n % 4 == 0) {
    for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
      const float4 vv = *reinterpret_cast<const float4*>(vr + i);
      const float4 gv = *reinterpret_cast<const float4*>(gw + i);
      *reinterpret_cast<float4*>(dv + (long long)co * n + i) =
          make_float4(a * gv.x - b * vv.x, a * gv.y - b * vv.y, a * gv.z - b * vv.z, a * gv.w - b * vv.w);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dv[(long long)co * n + i] = a * gw[i] - b * vr[i];
  }
}

// The same result without LDS and with every global load in flight at once (round 3): thread t owns input channels
// 4t .. 4t + 3 of this output channel for ALL taps -- in checkpoint order ([ci][tap]) that is ONE contiguous run of 4 * TAPS
// floats of v and of dv, and TAPS 16-byte pieces (x splits) of the slab planes.  The kernels above issue their loads in
// three dependent waves (slabs -> LDS, then v, then inv_norm / g after the block sum): ~4 memory round trips per workgroup,
// which is what a launch with one workgroup per output channel costs (20 us for a 1x1 conv's 12 MB, 32 us for a 5-tap
// conv's 100 MB); here the only dependency is the block sum.  Needs the 16-byte conditions of the VEC path.
template <int TAPS>
__global__ __launch_bounds__(256) void weightnorm_bwd_reg_kernel(
    const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ inv_norm,
    const float* __restrict__ dW, int splits, long long split_stride, float* __restrict__ dv,
    float* __restrict__ dg, int Cout, int Cin, int ldw, int perm_split, int off_lo, int off_hi,
    const float* __restrict__ poison) {
  __shared__ float sh[17];
  const int co = blockIdx.x;
  const long long n = (long long)Cin * TAPS;
  const float inv = inv_norm[co], gg = g[co], pz = poison ? poison[0] : 0.f;
  constexpr int MAXJ = 2;                                           // Cin <= 2048 (the host checks)
  float4 vv[MAXJ][TAPS], gs[MAXJ][TAPS];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int ci = threadIdx.x * 4 + j * 1024;
    const bool on = ci < Cin;
    const int cic = on ? ci : 0;
    const int col = perm_col(cic, perm_split, off_lo, off_hi);
    const float* vp = v + (long long)co * n + (long long)cic * TAPS;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) vv[j][k] = on ? *reinterpret_cast<const float4*>(vp + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
      const float* plane = dW + ((long long)k * Cout + co) * ldw + col;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on) {
#pragma unroll 4
        for (int sp = 0; sp < splits; ++sp) {
          const float4 t = *reinterpret_cast<const float4*>(plane + sp * split_stride);
          s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
        }
      }
      gs[j][k] = s0;                                               // gradient of channels ci .. ci + 3 at tap k
    }
  }
  // element (channel c of the four, tap k) sits at c * TAPS + k of the thread's run
  auto run_at = [](const float4 (&r)[TAPS], int idx) { const float4 q = r[idx >> 2]; return (idx & 3) == 0 ? q.x : (idx & 3) == 1 ? q.y : (idx & 3) == 2 ? q.z : q.w; };
  auto chan_of = [](const float4& q, int c) { return c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w; };
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) dot = fmaf(chan_of(gs[j][k], c), run_at(vv[j], c * TAPS + k), dot);
  dot = block_sum(dot, sh);
  if (threadIdx.x == 0) dg[co] = dot * inv + pz;                   // poison: 0, or NaN after a non-finite upstream gradient
  const float a = gg * inv, b = gg * dot * inv * inv * inv;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int ci = threadIdx.x * 4 + j * 1024;
    if (ci < Cin) {
      float o[4 * TAPS];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < TAPS; ++k) o[c * TAPS + k] = a * chan_of(gs[j][k], c) - b * run_at(vv[j], c * TAPS + k);
      float* dp = dv + (long long)co * n + (long long)ci * TAPS;
#pragma unroll
      for (int k = 0; k < TAPS; ++k) *reinterpret_cast<float4*>(dp + 4 * k) = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
  }
}

// ------------------------------------------------------------------ WN input assembly
__global__ __launch_bounds__(256) void wn_input_fwd_kernel(
    const float* __restrict__ ctx, int ldctx, const float* __restrict__ z, int ldz,
    float* __restrict__ X0, int ldx0, int rows, int D, int h, _Float16* __restrict__ Xh,
    _Float16* __restrict__ Xl, int fmt, float x8_mul, void* __restrict__ lo16) {
  const long long total = (long long)rows * ldx0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldx0), c = (int)(i - (long long)r * ldx0);
    float v = 0.f;
    if (c < D) v = ctx[(long long)r * ldctx + c];
    else if (c < D + h) v = z[(long long)r * ldz + (c - D)];
    if (X0) X0[i] = v;
    if (Xh) radmmm::store_split1_fmt(Xh, Xl, (long long)r * ldx0, c, fmt, x8_mul, 1.f, v, lo16);   // split copy (same pitch), scale 1
  }
}

// the same assembly, four columns per thread (16-byte loads / stores): D, ldctx, ldz, ldx0 multiples of 4 and 16-byte
// aligned bases -- every shape of the shipped configs; h is arbitrary (79, 78, 77 after the early exits): the last
// group of the z part goes element by element
__global__ __launch_bounds__(256) void wn_input_fwd4_kernel(
    const float* __restrict__ ctx, int ldctx, const float* __restrict__ z, int ldz, float* __restrict__ X0, int ldx0,
    int rows, int D, int h, _Float16* __restrict__ Xh, _Float16* __restrict__ Xl, int fmt, float x8_mul,
    void* __restrict__ lo16) {
  const int q = ldx0 >> 2;
  const long long total = (long long)rows * q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / q), c = (int)(i - (long long)r * q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) v = *reinterpret_cast<const float4*>(ctx + (long long)r * ldctx + c);
    else if (c + 3 < D + h) v = *reinterpret_cast<const float4*>(z + (long long)r * ldz + (c - D));
    else if (c < D + h) {
      const float* zp = z + (long long)r * ldz + (c - D);
      v.x = zp[0];
      if (c + 1 < D + h) v.y = zp[1];
      if (c + 2 < D + h) v.z = zp[2];
    }
    if (X0) *reinterpret_cast<float4*>(X0 + (long long)r * ldx0 + c) = v;
    if (Xh) radmmm::store_split4_fmt(Xh, Xl, (long long)r * ldx0, c, fmt, x8_mul, 1.f, v.x, v.y, v.z, v.w, lo16);
  }
}

__global__ __launch_bounds__(256) void wn_input_bwd4_kernel(
    const float* __restrict__ gX0, int ldx0, float* __restrict__ gctx, int ldctx, int ctx_accum,
    float* __restrict__ gz, int ldz, int rows, int D, int h) {
  const int q = (D + h + 3) >> 2;
  const long long total = (long long)rows * q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / q), c = (int)(i - (long long)r * q) * 4;
    const float4 v = *reinterpret_cast<const float4*>(gX0 + (long long)r * ldx0 + c);     // ldx0 >= round_up(D + h, 4)
    if (c >= D && c + 3 >= D + h) {                                                       // ragged end of the z part
      float* zp = gz + (long long)r * ldz + (c - D);
      zp[0] += v.x;
      if (c + 1 < D + h) zp[1] += v.y;
      if (c + 2 < D + h) zp[2] += v.z;
      continue;
    }
    float4* p = reinterpret_cast<float4*>(c < D ? gctx + (long long)r * ldctx + c : gz + (long long)r * ldz + (c - D));
    float4 o = v;
    if (c >= D || ctx_accum) {
      const float4 t = *p;
      o = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
    }
    *p = o;
  }
}

__global__ __launch_bounds__(256) void wn_input_bwd_kernel(
    const float* __restrict__ gX0, int ldx0, float* __restrict__ gctx, int ldctx, int ctx_accum,
    float* __restrict__ gz, int ldz, int rows, int D, int h) {
  const int W = D + h;
  const long long total = (long long)rows * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i - (long long)r * W);
    const float v = gX0[(long long)r * ldx0 + c];
    if (c < D) {
      float* p = gctx + (long long)r * ldctx + c;
      *p = ctx_accum ? *p + v : v;
    } else {
      gz[(long long)r * ldz + (c - D)] += v;
    }
  }
}

// ------------------------------------------------------------------ affine coupling
__device__ __forceinline__ void scale_fwd(float su, int mode, float& s, float& ls) {
  switch (mode) {
    case RADMMM_SCALE_TANH: s = tanhf(su) + 1.f + 1e-6f; ls = logf(s); break;
    case RADMMM_SCALE_EXP: s = expf(su); ls = su; break;
    case RADMMM_SCALE_SIGMOID: s = 1.f / (1.f + expf(-(su + 10.f))) + 1e-6f; ls = logf(s); break;
    default: s = 1.f; ls = 0.f; break;
  }
}

__global__ __launch_bounds__(256) void affine_coupling_fwd_kernel(
    const float* __restrict__ O, int ldo, const float* __restrict__ z, int ldz,
    float* __restrict__ zout, float* __restrict__ log_s, int rows, int h, int mode) {
  const long long total = (long long)rows * ldz;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldz), c = (int)(i - (long long)r * ldz);
    float v = z[i];
    if (c >= h && c < 2 * h) {
      const float su = O[(long long)r * ldo + (c - h)], b = O[(long long)r * ldo + c];
      float s, ls;
      scale_fwd(su, mode, s, ls);
      v = s * v + b;
      log_s[(long long)r * h + (c - h)] = ls;
    }
    zout[i] = v;
  }
}

__global__ __launch_bounds__(256) void affine_coupling_bwd_kernel(
    const float* __restrict__ O, int ldo, const float* __restrict__ z, int ldz,
    const float* __restrict__ gzout, const float* __restrict__ glog_s, float* __restrict__ gO,
    float* __restrict__ gz, int rows, int h, int mode) {
  const long long total = (long long)rows * ldz;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldz), c = (int)(i - (long long)r * ldz);
    float gv = gzout[i];
    if (c >= h && c < 2 * h) {
      const float su = O[(long long)r * ldo + (c - h)];
      const float z1 = z[i];
      const float gl = glog_s ? glog_s[(long long)r * h + (c - h)] : 0.f;
      float s, gsu;
      switch (mode) {
        case RADMMM_SCALE_TANH: {
          const float th = tanhf(su);
          s = th + 1.f + 1e-6f;
          gsu = (gv * z1 + gl / s) * (1.f - th * th);
        } break;
        case RADMMM_SCALE_EXP: s = expf(su); gsu = gv * z1 * s + gl; break;
        case RADMMM_SCALE_SIGMOID: {
          const float sg = 1.f / (1.f + expf(-(su + 10.f)));
          s = sg + 1e-6f;
          gsu = (gv * z1 + gl / s) * sg * (1.f - sg);
        } break;
        default: s = 1.f; gsu = 0.f; break;
      }
      gO[(long long)r * ldo + (c - h)] = gsu;
      gO[(long long)r * ldo + c] = gv;
      gv = gv * s;
    }
    gz[i] = gv;
  }
}

// ------------------------------------------------------------------ dact product
__global__ __launch_bounds__(256) void dact_mul_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ saved, int lds,
    float* __restrict__ y, int ldy, int rows, int cols, int dact, int rowscale, int T,
    const int* __restrict__ lens, int taps, int dil, _Float16* __restrict__ yh, _Float16* __restrict__ yl,
    int ldyh, float yscale, int fmt, float x8_mul, int* __restrict__ sat_flag) {
  float sat = 0.f;
  const int c4n = (cols + 3) / 4;
  const long long total = (long long)rows * c4n;
  const bool vec = (cols % 4 == 0) && (ldg % 4 == 0) && (lds % 4 == 0) && (ldy % 4 == 0);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4n), c = (int)(i - (long long)r * c4n) * 4;
    float rs = 1.f;
    if (rowscale) {
      const int b = r / T, t = r - b * T;
      const int len = lens ? lens[b] : T;
      rs = t < len ? 1.f : 0.f;
      if (rowscale == 2) rs *= radmmm::pconv_ratio(t, len, taps, dil);
    }
    if (vec) {
      const float4 gv = *reinterpret_cast<const float4*>(g + (long long)r * ldg + c);
      float4 o;
      if (dact) {
        const float4 sv = *reinterpret_cast<const float4*>(saved + (long long)r * lds + c);
        o.x = gv.x * radmmm::dact_from_out(sv.x, dact) * rs;
        o.y = gv.y * radmmm::dact_from_out(sv.y, dact) * rs;
        o.z = gv.z * radmmm::dact_from_out(sv.z, dact) * rs;
        o.w = gv.w * radmmm::dact_from_out(sv.w, dact) * rs;
      } else {
        o.x = gv.x * rs; o.y = gv.y * rs; o.z = gv.z * rs; o.w = gv.w * rs;
      }
      *reinterpret_cast<float4*>(y + (long long)r * ldy + c) = o;
      if (yh) sat = fmaxf(sat, radmmm::store_split4_fmt(yh, yl, (long long)r * ldyh, c, fmt, x8_mul, yscale, o.x, o.y, o.z, o.w));
    } else {
      for (int e = 0; e < 4 && c + e < cols; ++e) {
        const float d = dact ? radmmm::dact_from_out(saved[(long long)r * lds + c + e], dact) : 1.f;
        const float ov = g[(long long)r * ldg + c + e] * d * rs;
        y[(long long)r * ldy + c + e] = ov;
        if (yh) sat = fmaxf(sat, radmmm::store_split1_fmt(yh, yl, (long long)r * ldyh, c + e, fmt, x8_mul, yscale, ov));
      }
    }
  }
  radmmm::raise_sat_flag(sat_flag, sat, fmt ? x8_mul : 0.f);
}

// ------------------------------------------------------------------ column sums
// Stage 1: block (bx, by) sums rows [by*CS_ROWS, +CS_ROWS) of the 1024-column slab bx; each thread
// owns 4 adjacent columns (float4 loads: a block reads 4 KiB contiguous per row).  Stage 2 adds
// the per-chunk partials.  Deterministic (no atomics).
constexpr int CS_ROWS_PER_BLOCK = 64;
// frame-rate inputs (thousands of rows x >= 512 columns: the bias sums of the WN backward) take 16 rows per block: 64
// serial 16-byte loads per thread in only rows / 64 blocks left the pass at 1.2 TB/s
__host__ __device__ inline int colsum_rows_per_block(int rows, int cols) {
  return (rows >= 4096 && cols >= 512) ? 16 : CS_ROWS_PER_BLOCK;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(
    const float* __restrict__ X, int ldx, float* __restrict__ part, int rows, int cols,
    int row_weight, int T, const int* __restrict__ lens, int taps, int dil, int square, int rpb) {
  // narrow inputs (cols < 1024: the 160-wide flow tensors) would leave most of the block idle with 64
  // serial loads per thread: the spare threads become row lanes (thread = column group + ncg * row lane)
  // whose partial sums are combined through LDS in a fixed order
  __shared__ float4 sh[256];
  const int span = min(cols - (int)blockIdx.x * 1024, 1024);
  const int ncg = (span + 3) / 4, nrl = 256 / ncg;             // column groups of 4, row lanes
  const int cg = threadIdx.x % ncg, rl = threadIdx.x / ncg;
  const int c = blockIdx.x * 1024 + cg * 4;
  const int r0 = blockIdx.y * rpb;
  int r1 = r0 + rpb;
  if (r1 > rows) r1 = rows;
  const bool vec = (ldx % 4 == 0) && radmmm::aligned16(X) && c + 3 < cols;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (rl < nrl) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += nrl) {
      const float w = radmmm::colsum_row_weight(r, row_weight, T, lens, taps, dil);
      float4 v;
      if (vec) {
        v = *reinterpret_cast<const float4*>(X + (long long)r * ldx + c);
      } else {
        const float* q = X + (long long)r * ldx + c;
        v.x = q[0];
        v.y = c + 1 < cols ? q[1] : 0.f;
        v.z = c + 2 < cols ? q[2] : 0.f;
        v.w = c + 3 < cols ? q[3] : 0.f;
      }
      if (square) { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
      a0 = fmaf(w, v.x, a0); a1 = fmaf(w, v.y, a1); a2 = fmaf(w, v.z, a2); a3 = fmaf(w, v.w, a3);
    }
  }
  if (nrl > 1) {
    sh[threadIdx.x] = make_float4(a0, a1, a2, a3);
    __syncthreads();
    if (rl == 0)
      for (int k = 1; k < nrl; ++k) {
        const float4 o = sh[k * ncg + cg];
        a0 += o.x; a1 += o.y; a2 += o.z; a3 += o.w;
      }
  }
  if (rl == 0) {
    float* o = part + (long long)blockIdx.y * cols + c;
    o[0] = a0;
    if (c + 1 < cols) o[1] = a1;
    if (c + 2 < cols) o[2] = a2;
    if (c + 3 < cols) o[3] = a3;
  }
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part,
                                                            float* __restrict__ out, int nparts,
                                                            int cols) {
  // 16 waves per 64 columns, each a slice of the partials, then a fixed-order LDS combine
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  // four loads in flight per wave (a wave's slice of 512 partials is 32 dependent L2 round trips otherwise: 12.7 us per
  // launch at 32 000 rows); fixed order: partial q goes to accumulator (q / 16) % 4
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int q = pl;
    for (; q + 48 < nparts; q += 64) {
      const float a0 = part[(long long)q * cols + c], a1 = part[(long long)(q + 16) * cols + c];
      const float a2 = part[(long long)(q + 32) * cols + c], a3 = part[(long long)(q + 48) * cols + c];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    if (q < nparts) s0 += part[(long long)q * cols + c];
    if (q + 16 < nparts) s1 += part[(long long)(q + 16) * cols + c];
    if (q + 32 < nparts) s2 += part[(long long)(q + 32) * cols + c];
  }
  sh[pl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w][cl];
    out[c] = t;
  }
}

// ------------------------------------------------------------------ masked reductions
constexpr int MR_BLOCKS = 512;
__global__ __launch_bounds__(256) void masked_reduce_kernel(
    const float* __restrict__ x, int B, int C, int T, long long sb, long long sc, long long st,
    const int* __restrict__ lens, int mode, float* __restrict__ part) {
  __shared__ float sh[17];
  const long long total = (long long)B * C * T;
  const bool c_inner = sc < st;  // iterate the stride-1 dim fastest
  float acc = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int b, c, t;
    if (c_inner) {
      c = (int)(i % C);
      const long long j = i / C;
      t = (int)(j % T);
      b = (int)(j / T);
    } else {
      t = (int)(i % T);
      const long long j = i / T;
      c = (int)(j % C);
      b = (int)(j / C);
    }
    const int len = lens ? lens[b] : T;
    if (t < len) {
      const float v = x[b * sb + c * sc + t * st];
      acc += mode ? v * v : v;
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ part, int n,
                                                           float* __restrict__ out) {
  __shared__ float sh[17];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) out[0] = acc;
}
__global__ __launch_bounds__(256) void masked_reduce_bwd_kernel(
    const float* __restrict__ x, int B, int C, int T, long long sb, long long sc, long long st,
    const int* __restrict__ lens, int mode, const float* __restrict__ coef,
    float* __restrict__ gx) {
  // gx has the SAME strides as x (torch.empty_like of a dense, possibly permuted, view)
  const long long total = (long long)B * C * T;
  const bool c_inner = sc < st;
  const float cf = coef[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int b, c, t;
    if (c_inner) {
      c = (int)(i % C);
      const long long j = i / C;
      t = (int)(j % T);
      b = (int)(j / T);
    } else {
      t = (int)(i % T);
      const long long j = i / T;
      c = (int)(j % C);
      b = (int)(j / C);
    }
    const int len = lens ? lens[b] : T;
    const long long off = b * sb + c * sc + t * st;
    float v = 0.f;
    if (t < len) v = mode ? 2.f * cf * x[off] : cf;
    gx[off] = v;
  }
}

__global__ __launch_bounds__(256) void fatsm_kernel(const float* __restrict__ a,
                                                    const float* __restrict__ b, int ld,
                                                    float* __restrict__ y, int ldy, int rows, int n) {
  const long long total = (long long)rows * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long long)r * n);
    const float t = a[(long long)r * ld + c] + b[(long long)r * ld + c];
    const float s = a[(long long)r * ld + n + c] + b[(long long)r * ld + n + c];
    y[(long long)r * ldy + c] = tanhf(t) * (1.f / (1.f + expf(-s)));
  }
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 8) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int radmmm_weightnorm_fwd(const float* v, const float* g, float* W, float* inv_norm,
                                     int Cout, int Cin, int taps, int ldw, int perm_split,
                                     int off_lo, int off_hi, radmmm_stream_t stream) {
  RADMMM_REQUIRE(v && g && W && inv_norm, "weightnorm_fwd: null pointer");
  RADMMM_REQUIRE(Cout > 0 && Cin > 0 && taps > 0 && ldw >= Cin, "weightnorm_fwd: bad dims");
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(Cout), dim3(256), 0, ST(stream), v, g, W, inv_norm,
                     Cout, Cin, taps, ldw, perm_split, off_lo, off_hi);
  return radmmm::check_launch("weightnorm_fwd");
}

extern "C" int radmmm_weightnorm_bwd(const float* v, const float* g, const float* inv_norm,
                                     const float* dW, int splits, int64_t split_stride, float* dv,
                                     float* dg, int Cout, int Cin, int taps, int ldw,
                                     int perm_split, int off_lo, int off_hi, const float* poison,
                                     radmmm_stream_t stream) {
  RADMMM_REQUIRE(v && g && inv_norm && dW && dv && dg, "weightnorm_bwd: null pointer");
  RADMMM_REQUIRE(Cout > 0 && Cin > 0 && taps > 0 && ldw >= Cin && splits >= 1, "weightnorm_bwd: bad dims");
  const size_t lds = (size_t)Cin * taps * sizeof(float);
  if (lds <= 48 * 1024) {
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = Cin % 4 == 0 && perm_split % 4 == 0 && off_lo % 4 == 0 && off_hi % 4 == 0 && ldw % 4 == 0 &&
                     split_stride % 4 == 0 && a16(dW) && a16(v) && a16(dv);
    static const bool reg_ok = !(radmmm::debug_env("RADMMM_WNBWD_REG") && atoi(radmmm::debug_env("RADMMM_WNBWD_REG")) == 0);
    if (vec && reg_ok && Cin <= 2048 && (taps == 1 || taps == 3 || taps == 5)) {
      // (Cin % 4 == 0 and 16-byte aligned v / dv rows: every thread's run of 4 * taps floats starts on a 16-byte boundary)
      if (taps == 1)
        hipLaunchKernelGGL(weightnorm_bwd_reg_kernel<1>, dim3(Cout), dim3(256), 0, ST(stream), v, g, inv_norm, dW, splits,
                           (long long)split_stride, dv, dg, Cout, Cin, ldw, perm_split, off_lo, off_hi, poison);
      else if (taps == 3)
        hipLaunchKernelGGL(weightnorm_bwd_reg_kernel<3>, dim3(Cout), dim3(256), 0, ST(stream), v, g, inv_norm, dW, splits,
                           (long long)split_stride, dv, dg, Cout, Cin, ldw, perm_split, off_lo, off_hi, poison);
      else
        hipLaunchKernelGGL(weightnorm_bwd_reg_kernel<5>, dim3(Cout), dim3(256), 0, ST(stream), v, g, inv_norm, dW, splits,
                           (long long)split_stride, dv, dg, Cout, Cin, ldw, perm_split, off_lo, off_hi, poison);
    } else if (vec)
      hipLaunchKernelGGL(weightnorm_bwd_lds_kernel<true>, dim3(Cout), dim3(256), lds, ST(stream), v, g, inv_norm, dW,
                         splits, (long long)split_stride, dv, dg, Cout, Cin, taps, ldw, perm_split, off_lo, off_hi, poison);
    else
      hipLaunchKernelGGL(weightnorm_bwd_lds_kernel<false>, dim3(Cout), dim3(256), lds, ST(stream), v, g, inv_norm, dW,
                         splits, (long long)split_stride, dv, dg, Cout, Cin, taps, ldw, perm_split, off_lo, off_hi, poison);
  } else {
    hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(Cout), dim3(256), 0, ST(stream), v, g, inv_norm, dW,
                       splits, (long long)split_stride, dv, dg, Cout, Cin, taps, ldw, perm_split,
                       off_lo, off_hi, poison);
  }
  return radmmm::check_launch("weightnorm_bwd");
}

extern "C" int radmmm_wn_input_fwd(const float* ctx, int ldctx, const float* z, int ldz, float* X0,
                                   int ldx0, int rows, int D, int h, void* Xh, void* Xl,
                                   const radmmm_split_opts* so, radmmm_stream_t stream) {
  RADMMM_REQUIRE(ctx && z && (X0 || Xh), "wn_input_fwd: null pointer (X0 may be NULL when the split copy Xh / Xl is written)");
  RADMMM_REQUIRE(rows > 0 && D > 0 && h > 0 && ldx0 >= D + h && ldctx >= D && ldz >= h, "wn_input_fwd: bad dims");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(fmt == RADMMM_SPLIT_F16 || !Xh || (ldx0 % 32 == 0 && abs(so->x8_exp) <= 16), "wn_input_fwd: 8-bit format needs ldx0 %% 32 == 0");
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  auto a8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
  if (D % 4 == 0 && ldctx % 4 == 0 && ldz % 4 == 0 && ldx0 % 4 == 0 && a16(ctx) && a16(z) && a16(X0) &&
      a8(Xh) && a8(Xl) && a8(so ? so->lo16 : nullptr)) {
    hipLaunchKernelGGL(wn_input_fwd4_kernel, dim3(grid_for((long long)rows * (ldx0 / 4))), dim3(256), 0, ST(stream), ctx,
                       ldctx, z, ldz, X0, ldx0, rows, D, h, static_cast<_Float16*>(Xh), static_cast<_Float16*>(Xl), fmt,
                       ldexpf(1.f, so ? so->x8_exp : 0), so ? so->lo16 : nullptr);
    return radmmm::check_launch("wn_input_fwd");
  }
  hipLaunchKernelGGL(wn_input_fwd_kernel, dim3(grid_for((long long)rows * ldx0)), dim3(256), 0,
                     ST(stream), ctx, ldctx, z, ldz, X0, ldx0, rows, D, h, static_cast<_Float16*>(Xh),
                     static_cast<_Float16*>(Xl), fmt, ldexpf(1.f, so ? so->x8_exp : 0), so ? so->lo16 : nullptr);
  return radmmm::check_launch("wn_input_fwd");
}

extern "C" int radmmm_wn_input_bwd(const float* gX0, int ldx0, float* gctx, int ldctx, int ctx_accum,
                                   float* gz, int ldz, int rows, int D, int h,
                                   radmmm_stream_t stream) {
  RADMMM_REQUIRE(gX0 && gctx && gz, "wn_input_bwd: null pointer");
  RADMMM_REQUIRE(rows > 0 && D > 0 && h > 0 && ldx0 >= D + h && ldctx >= D && ldz >= h, "wn_input_bwd: bad dims");
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (D % 4 == 0 && ldctx % 4 == 0 && ldz % 4 == 0 && ldx0 % 4 == 0 && ldx0 >= (D + h + 3) / 4 * 4 && a16(gX0) && a16(gctx) &&
      a16(gz)) {
    hipLaunchKernelGGL(wn_input_bwd4_kernel, dim3(grid_for((long long)rows * ((D + h + 3) / 4))), dim3(256), 0, ST(stream), gX0,
                       ldx0, gctx, ldctx, ctx_accum, gz, ldz, rows, D, h);
    return radmmm::check_launch("wn_input_bwd");
  }
  hipLaunchKernelGGL(wn_input_bwd_kernel, dim3(grid_for((long long)rows * (D + h))), dim3(256), 0,
                     ST(stream), gX0, ldx0, gctx, ldctx, ctx_accum, gz, ldz, rows, D, h);
  return radmmm::check_launch("wn_input_bwd");
}

extern "C" int radmmm_affine_coupling_fwd(const float* O, int ldo, const float* z, int ldz,
                                          float* zout, float* log_s, int rows, int h, int scaling,
                                          radmmm_stream_t stream) {
  RADMMM_REQUIRE(O && z && zout && log_s, "affine_coupling_fwd: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && ldo >= 2 * h && ldz >= 2 * h, "affine_coupling_fwd: bad dims");
  hipLaunchKernelGGL(affine_coupling_fwd_kernel, dim3(grid_for((long long)rows * ldz)), dim3(256), 0,
                     ST(stream), O, ldo, z, ldz, zout, log_s, rows, h, scaling);
  return radmmm::check_launch("affine_coupling_fwd");
}

extern "C" int radmmm_affine_coupling_bwd(const float* O, int ldo, const float* z, int ldz,
                                          const float* gzout, const float* glog_s, float* gO,
                                          float* gz, int rows, int h, int scaling,
                                          radmmm_stream_t stream) {
  RADMMM_REQUIRE(O && z && gzout && gO && gz, "affine_coupling_bwd: null pointer");
  RADMMM_REQUIRE(rows > 0 && h > 0 && ldo >= 2 * h && ldz >= 2 * h, "affine_coupling_bwd: bad dims");
  hipLaunchKernelGGL(affine_coupling_bwd_kernel, dim3(grid_for((long long)rows * ldz)), dim3(256), 0,
                     ST(stream), O, ldo, z, ldz, gzout, glog_s, gO, gz, rows, h, scaling);
  return radmmm::check_launch("affine_coupling_bwd");
}

extern "C" int radmmm_dact_mul(const float* g, int ldg, const float* saved, int lds, float* y,
                               int ldy, int rows, int cols, int dact, int rowscale, int T,
                               const int32_t* lens, int taps, int dil, void* yh, void* yl, int ldyh,
                               float yscale, const radmmm_split_opts* so, radmmm_stream_t stream) {
  RADMMM_REQUIRE(!yh || (yl && ldyh >= cols && ldyh % 4 == 0), "dact_mul: split output needs ldyh %% 4 == 0");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(fmt == RADMMM_SPLIT_F16 || !yh || (ldyh % 32 == 0 && cols % 4 == 0 && abs(so->x8_exp) <= 16),
                 "dact_mul: 8-bit format needs ldyh %% 32 == 0");
  RADMMM_REQUIRE(g && y && (saved || !dact), "dact_mul: null pointer");
  RADMMM_REQUIRE(rows > 0 && cols > 0 && ldg >= cols && ldy >= cols && (!dact || lds >= cols), "dact_mul: bad dims");
  RADMMM_REQUIRE(!rowscale || (T > 0 && rows % T == 0), "dact_mul: rows must be a multiple of T");
  RADMMM_REQUIRE(rowscale != 2 || (taps >= 1 && dil >= 1), "dact_mul: taps/dil");
  hipLaunchKernelGGL(dact_mul_kernel, dim3(grid_for((long long)rows * ((cols + 3) / 4))), dim3(256), 0,
                     ST(stream), g, ldg, saved, lds, y, ldy, rows, cols, dact, rowscale, T > 0 ? T : 1, lens, taps,
                     dil, static_cast<_Float16*>(yh), static_cast<_Float16*>(yl), ldyh, yscale, fmt,
                     ldexpf(1.f, so ? so->x8_exp : 0), so ? so->sat_flag : nullptr);
  return radmmm::check_launch("dact_mul");
}

extern "C" int64_t radmmm_colsum_scratch_floats(int rows, int cols) {
  const int rpb = colsum_rows_per_block(rows, cols);
  const int64_t nparts = (rows + rpb - 1) / rpb;
  return nparts * cols;
}

extern "C" int radmmm_colsum(const float* X, int ldx, float* out, float* scratch, int rows, int cols,
                             int row_weight, int T, const int32_t* lens, int taps, int dil, int square,
                             radmmm_stream_t stream) {
  RADMMM_REQUIRE(X && out && scratch, "colsum: null pointer");
  RADMMM_REQUIRE(rows > 0 && cols > 0 && ldx >= cols, "colsum: bad dims");
  RADMMM_REQUIRE(row_weight == 0 || (T > 0 && rows % T == 0), "colsum: rows must be a multiple of T");
  RADMMM_REQUIRE(row_weight != 2 || (taps >= 1 && dil >= 1), "colsum: taps/dil");
  const int rpb = colsum_rows_per_block(rows, cols);
  const int nparts = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 1023) / 1024, nparts), dim3(256), 0, ST(stream),
                     X, ldx, scratch, rows, cols, row_weight, T > 0 ? T : 1, lens, taps, dil, square, rpb);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 63) / 64), dim3(1024), 0, ST(stream), scratch,
                     out, nparts, cols);
  return radmmm::check_launch("colsum");
}

// several independent finals in one launch (item = blockIdx.y; the items travel as the kernel argument): the bias sums of a
// flow step's backward are nine 5-microsecond launches otherwise.  Same summation order per item as colsum_final_kernel.
struct CsMulti {
  const float* part[16];
  float* out[16];
  int nparts[16], cols[16];
};
__global__ __launch_bounds__(1024) void colsum_final_multi_kernel(const CsMulti m) {
  __shared__ float sh[16][64];
  const int it = blockIdx.y;
  const float* __restrict__ part = m.part[it];
  const int nparts = m.nparts[it], cols = m.cols[it];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  if (blockIdx.x * 64 >= cols) return;                       // (uniform per block)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int q = pl;
    for (; q + 48 < nparts; q += 64) {
      const float a0 = part[(long long)q * cols + c], a1 = part[(long long)(q + 16) * cols + c];
      const float a2 = part[(long long)(q + 32) * cols + c], a3 = part[(long long)(q + 48) * cols + c];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    if (q < nparts) s0 += part[(long long)q * cols + c];
    if (q + 16 < nparts) s1 += part[(long long)(q + 16) * cols + c];
    if (q + 32 < nparts) s2 += part[(long long)(q + 32) * cols + c];
  }
  sh[pl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w][cl];
    m.out[it][c] = t;
  }
}

extern "C" int radmmm_colsum_final_multi(const radmmm_cs_item* items, int n, radmmm_stream_t stream) {
  RADMMM_REQUIRE(items && n > 0, "colsum_final_multi: bad arguments");
  for (int k = 0; k < n; k += 16) {
    CsMulti m;
    const int cnt = n - k < 16 ? n - k : 16;
    int maxc = 0;
    for (int i = 0; i < cnt; ++i) {
      const radmmm_cs_item& it = items[k + i];
      RADMMM_REQUIRE(it.part && it.out && it.nparts > 0 && it.cols > 0, "colsum_final_multi: bad item");
      m.part[i] = it.part; m.out[i] = it.out; m.nparts[i] = it.nparts; m.cols[i] = it.cols;
      maxc = it.cols > maxc ? it.cols : maxc;
    }
    hipLaunchKernelGGL(colsum_final_multi_kernel, dim3((maxc + 63) / 64, cnt), dim3(1024), 0, ST(stream), m);
    const int rc = radmmm::check_launch("colsum_final_multi");
    if (rc) return rc;
  }
  return 0;
}

// out[c] = sum_p part[p][c] (fixed order): second stage for partials produced by another kernel
// (radmmm_transpose_split_act_colsum)
extern "C" int radmmm_colsum_final(const float* part, float* out, int nparts, int cols, radmmm_stream_t stream) {
  RADMMM_REQUIRE(part && out && nparts > 0 && cols > 0, "colsum_final: bad arguments");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 63) / 64), dim3(1024), 0, ST(stream), part, out, nparts, cols);
  return radmmm::check_launch("colsum_final");
}

extern "C" int64_t radmmm_masked_reduce_scratch_floats(int, int, int) { return MR_BLOCKS; }

extern "C" int radmmm_masked_reduce(const float* x, int B, int C, int T, int64_t sb, int64_t sc,
                                    int64_t st, const int32_t* lens, int mode, float* out,
                                    float* scratch, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && out && scratch, "masked_reduce: null pointer");
  RADMMM_REQUIRE(B > 0 && C > 0 && T > 0, "masked_reduce: bad dims");
  const int nb = grid_for((long long)B * C * T, 256, MR_BLOCKS);
  hipLaunchKernelGGL(masked_reduce_kernel, dim3(nb), dim3(256), 0, ST(stream), x, B, C, T,
                     (long long)sb, (long long)sc, (long long)st, lens, mode, scratch);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, ST(stream), scratch, nb, out);
  return radmmm::check_launch("masked_reduce");
}

extern "C" int radmmm_masked_reduce_bwd(const float* x, int B, int C, int T, int64_t sb, int64_t sc,
                                        int64_t st, const int32_t* lens, int mode, const float* coef,
                                        float* gx, radmmm_stream_t stream) {
  RADMMM_REQUIRE(coef && gx && (x || mode == 0), "masked_reduce_bwd: null pointer");
  RADMMM_REQUIRE(B > 0 && C > 0 && T > 0, "masked_reduce_bwd: bad dims");
  hipLaunchKernelGGL(masked_reduce_bwd_kernel, dim3(grid_for((long long)B * C * T)), dim3(256), 0,
                     ST(stream), x, B, C, T, (long long)sb, (long long)sc, (long long)st, lens, mode,
                     coef, gx);
  return radmmm::check_launch("masked_reduce_bwd");
}

extern "C" int radmmm_fused_add_tanh_sigmoid_multiply(const float* a, const float* b, int ld, float* y,
                                                      int ldy, int rows, int n,
                                                      radmmm_stream_t stream) {
  RADMMM_REQUIRE(a && b && y, "fused_add_tanh_sigmoid_multiply: null pointer");
  RADMMM_REQUIRE(rows > 0 && n > 0 && ld >= 2 * n && ldy >= n, "fused_add_tanh_sigmoid_multiply: bad dims");
  hipLaunchKernelGGL(fatsm_kernel, dim3(grid_for((long long)rows * n)), dim3(256), 0, ST(stream), a, b,
                     ld, y, ldy, rows, n);
  return radmmm::check_launch("fused_add_tanh_sigmoid_multiply");
}

// ------------------------------------------------------------------ squeeze (nn.Unfold(kernel=(g,1), stride=g))
// out[(b*Tg + t') * ld + col0 + c*g + k] = in[(b*C + c)*T + t'*g + k]   (decoders.py:118-122,178; models/radmmm.py:114-120)
// straight from the reference layout [B, C, T] into channels-last rows of a wider matrix (the LSTM input, the flow
// variable), so neither the permuted copy nor the concatenation exists as a separate pass.  32 channels x 64 samples per
// workgroup through LDS: reads coalesced along t, writes along the 32*g consecutive columns of a row.
// INVERSE: the gradient scatter, gin[b, c, t] = gout[row(t / g)][col0 + c*g + t % g] for t < Tg*g, else 0.
namespace {
template <bool INVERSE>
__global__ __launch_bounds__(256) void squeeze_rows_kernel(float* __restrict__ in, float* __restrict__ out, int C, int T,
                                                           int g, int Tg, int ld, int col0) {
  __shared__ float tile[32][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const int rows_per_tile = 64 / g, cols_per_row = 32 * g;
  const long long row0 = (long long)b * Tg + t0 / g;
  if (!INVERSE) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 4 + (tid >> 6), t = t0 + (tid & 63);
      tile[i * 4 + (tid >> 6)][tid & 63] = (c < C && t < Tg * g) ? in[((long long)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = tid + 256 * j;
      const int r = e / cols_per_row, cc = e - r * cols_per_row;
      const int c = cc / g, k = cc - c * g;
      if (c0 + c < C && t0 / g + r < Tg) out[(row0 + r) * ld + col0 + (c0 + c) * g + k] = tile[c][r * g + k];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = tid + 256 * j;
      const int r = e / cols_per_row, cc = e - r * cols_per_row;
      const int c = cc / g, k = cc - c * g;
      tile[c][r * g + k] = (c0 + c < C && t0 / g + r < Tg) ? out[(row0 + r) * ld + col0 + (c0 + c) * g + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 4 + (tid >> 6), t = t0 + (tid & 63);
      if (c < C && t < T) in[((long long)b * C + c) * T + t] = (t < Tg * g) ? tile[i * 4 + (tid >> 6)][tid & 63] : 0.f;
    }
  }
  (void)rows_per_tile;
}
}  // namespace

extern "C" int radmmm_squeeze_rows(const float* in, float* out, int B, int C, int T, int g, int ld, int col0,
                                   radmmm_stream_t stream) {
  RADMMM_REQUIRE(in && out, "squeeze_rows: null pointer");
  RADMMM_REQUIRE(B > 0 && C > 0 && T > 0 && g >= 1 && 64 % g == 0 && T / g > 0 && col0 >= 0 && ld >= col0 + C * g,
                 "squeeze_rows: bad dims (group size must divide 64)");
  const int Tg = T / g;
  hipLaunchKernelGGL(squeeze_rows_kernel<false>, dim3((Tg * g + 63) / 64, (C + 31) / 32, B), dim3(256), 0, ST(stream),
                     const_cast<float*>(in), out, C, T, g, Tg, ld, col0);
  return radmmm::check_launch("squeeze_rows");
}

extern "C" int radmmm_unsqueeze_rows(const float* gout, float* gin, int B, int C, int T, int g, int ld, int col0,
                                     radmmm_stream_t stream) {
  RADMMM_REQUIRE(gin && gout, "unsqueeze_rows: null pointer");
  RADMMM_REQUIRE(B > 0 && C > 0 && T > 0 && g >= 1 && 64 % g == 0 && T / g > 0 && col0 >= 0 && ld >= col0 + C * g,
                 "unsqueeze_rows: bad dims (group size must divide 64)");
  const int Tg = T / g;
  hipLaunchKernelGGL(squeeze_rows_kernel<true>, dim3((T + 63) / 64, (C + 31) / 32, B), dim3(256), 0, ST(stream), gin,
                     const_cast<float*>(gout), C, T, g, Tg, ld, col0);
  return radmmm::check_launch("unsqueeze_rows");
}
