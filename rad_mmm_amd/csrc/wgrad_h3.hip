// wgrad_h3: weight gradient (contraction over frames) on the split-f16 path.
//
//   P[split][tap][m][n] = acc_scale * sum_k GYt[m][k] * Xt[n][k + shift(tap)]
//
// Both operands are TRANSPOSED, time-contiguous split copies produced by
// radmmm_transpose_split_act: row = channel, column k' = FRONT + b*Tp + t with Tp >= T + FRONT,
// zeros in the gaps between utterances, in masked frames (t >= len) and in the FRONT leading
// columns.  The zero gaps make the tap shift a plain column offset: a shifted read can never
// reach a neighbouring utterance, so no masking is needed in the GEMM at all.  Odd shifts
// (dilation 1) would be 2-byte misaligned for 16-byte loads; the producer therefore also emits
// a copy advanced by one column (X1[k] = X[k+1]) and odd shifts read that one at an even offset.
//
// Kernel = the NT split-f16 GEMM of h3_gemm.hip (128x128x32 tile, 3 MFMA per product block)
// plus a scalar column offset on the B operand and split-K slabs.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 40, TILE_H = BM * PITCH;
constexpr int SMEM_BYTES = 2 * 4 * TILE_H * 2;   // 80 KiB
constexpr int OOB = 0x7fffffff;

struct WgradH3Args {
  const _Float16 *GYh, *GYl;      // [Mc][ldk]
  const _Float16 *Xh, *Xl;        // [Nc][ldk]           even shifts
  const _Float16 *X1h, *X1l;      // [Nc][ldk] advanced by one column (odd shifts); may be null if no odd shift
  int ldk, k0, Kt;                // contraction over columns [k0, k0 + Kt), Kt % 32 == 0, k0 % 8 == 0
  float* P; int ldp; long long split_stride;
  int Mc, Nc, taps, dil, splits;
  float acc_scale;
  int a_bytes, b_bytes;           // exact extents of the operand arrays (buffer range check)
};

__global__ __launch_bounds__(256, 2) void wgrad_h3_kernel(const WgradH3Args a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smh[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (a.Mc + BM - 1) / BM, ntn = (a.Nc + BN - 1) / BN;
  int id = blockIdx.x;
  const int tn = id % ntn; id /= ntn;
  const int tm = id % ntm; id /= ntm;
  const int tap = id % a.taps;
  const int split = id / a.taps;
  const int m0 = tm * BM, n0 = tn * BN;
  const int shift = (tap - a.taps / 2) * a.dil;
  // odd shift -> read the advanced copy at shift-1 (even)
  const bool odd = (shift & 1) != 0;
  const int bshift = odd ? shift - 1 : shift;

  const int steps_total = a.Kt / BK;
  const int steps_per = (steps_total + a.splits - 1) / a.splits;
  const int step_lo = split * steps_per;
  int step_hi = step_lo + steps_per;
  if (step_hi > steps_total) step_hi = steps_total;
  const int nsteps = step_hi - step_lo;

  const int s_row = tid >> 2, s_chunk = tid & 3;
  int a_voff[2], b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = m0 + s_row + 64 * i, rb = n0 + s_row + 64 * i;
    a_voff[i] = ra < a.Mc ? (ra * a.ldk + s_chunk * 8) * 2 : OOB;
    b_voff[i] = rb < a.Nc ? (rb * a.ldk + s_chunk * 8) * 2 : OOB;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYh), 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYl), 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(odd ? a.X1h : a.Xh), 0, a.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(odd ? a.X1l : a.Xl), 0, a.b_bytes, 0x00020000);

  struct Regs {
    u32x4 v[4][2];
  };
  auto load_tiles = [&](int step, Regs& R) __attribute__((always_inline)) {
    const int so_a = (a.k0 + step * BK) * 2;
    // columns k + bshift >= 0 because k0 >= max |shift| (checked by the host): offsets never go
    // negative; a read past the row end lands in the next row's leading zeros or, for the last
    // row, is range-checked to zero
    const int so_b = so_a + bshift * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      R.v[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rAh, a_voff[i], so_a, 0);
      R.v[1][i] = __builtin_amdgcn_raw_buffer_load_b128(rAl, a_voff[i], so_a, 0);
      R.v[2][i] = __builtin_amdgcn_raw_buffer_load_b128(rBh, b_voff[i], so_b, 0);
      R.v[3][i] = __builtin_amdgcn_raw_buffer_load_b128(rBl, b_voff[i], so_b, 0);
    }
  };
  auto store_tiles = [&](int buf, const Regs& R) __attribute__((always_inline)) {
    _Float16* base = smh + buf * 4 * TILE_H;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(base + o * TILE_H + (s_row + 64 * i) * PITCH + s_chunk * 8) = R.v[o][i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int f_off = (lane & 31) * PITCH + (lane >> 5) * 8;
  if (nsteps > 0) {
    Regs R;
    load_tiles(step_lo, R);
    store_tiles(0, R);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      const int nxt = s + 1 < nsteps ? s + 1 : s;
      load_tiles(step_lo + nxt, R);
      const _Float16* base = smh + buf * 4 * TILE_H;
#pragma unroll
      for (int kb = 0; kb < BK / 16; ++kb) {
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ro = (wm * 64 + t * 32) * PITCH + kb * 16 + f_off;
          const int co = (wn * 64 + t * 32) * PITCH + kb * 16 + f_off;
          ah[t] = *reinterpret_cast<const f16x8*>(base + 0 * TILE_H + ro);
          al[t] = *reinterpret_cast<const f16x8*>(base + 1 * TILE_H + ro);
          bh[t] = *reinterpret_cast<const f16x8*>(base + 2 * TILE_H + co);
          bl[t] = *reinterpret_cast<const f16x8*>(base + 3 * TILE_H + co);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      store_tiles(buf ^ 1, R);
      __syncthreads();
    }
  }

  // epilogue through LDS: [128][128] fp32 (64 KiB of the 80 KiB)
  float* smf = reinterpret_cast<float*>(smh);
  {
    float* base = smf + (wm * 64 + 4 * (lane >> 5)) * BN + wn * 64 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          base[(mi * 32 + (e & 3) + 8 * (e >> 2)) * BN + ni * 32] = acc[mi][ni][e] * a.acc_scale;
  }
  __syncthreads();
  float* P = a.P + (long long)split * a.split_stride + (long long)tap * a.Mc * a.ldp;
  const bool vec_ok = (a.ldp % 4 == 0) && radmmm::aligned16(a.P) && (a.split_stride % 4 == 0);
  const int c4 = (tid & 31) * 4;
  const int col = n0 + c4;
  for (int i = 0; i < 16; ++i) {
    const int rl = i * 8 + (tid >> 5);
    const int row = m0 + rl;
    if (row < a.Mc && col < a.Nc) {
      const float4 a4 = *reinterpret_cast<const float4*>(smf + rl * BN + c4);
      if (vec_ok && col + 3 < a.Nc) {
        *reinterpret_cast<float4*>(P + (long long)row * a.ldp + col) = a4;
      } else {
        const float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < a.Nc) P[(long long)row * a.ldp + col + e] = v[e];
      }
    }
  }
}

// fp32 [B*T rows][ld] (first C columns) -> split, transposed, zero-gapped [C][ldk]:
//   out[c][FRONT + b*Tp + t] = split(scale * x[b*T + t][c]) for t < len_b (mask_mode 1) or t < T;
// everything else zero.  64 frames x 32 channels per block through LDS.
__global__ __launch_bounds__(256) void transpose_split_act_kernel(
    const float* __restrict__ x, int ld, int C, int T, int Tp, int front, const int* __restrict__ lens, int mask_mode,
    float scale, _Float16* __restrict__ oh, _Float16* __restrict__ ol, _Float16* __restrict__ o1h,
    _Float16* __restrict__ o1l, int ldk) {
  __shared__ float tile[64][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * 64, c0 = blockIdx.x * 32;
  const int len = (mask_mode && lens) ? lens[b] : T;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 64; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    tile[i][tx] = (t < len && t < T && c < C) ? x[((long long)b * T + t) * ld + c] * scale : 0.f;   // 0 in gap/masked frames
  }
  __syncthreads();
  // lanes run along time: for one channel, 64 consecutive frames = 128 contiguous bytes per array.
  // Frames t in [T, Tp) are the zero gap after the utterance and are (re)written here as zeros.
  const int fl = threadIdx.x & 63, cl0 = threadIdx.x >> 6;   // 64 x 4
  const int t = t0 + fl;
  if (t < Tp) {
    const long long k = (long long)front + (long long)b * Tp + t;
    for (int ci = cl0; ci < 32; ci += 4) {
      const int c = c0 + ci;
      if (c < C) {
        const float v = fminf(fmaxf(tile[fl][ci], -60000.f), 60000.f);
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        oh[(long long)c * ldk + k] = h;
        ol[(long long)c * ldk + k] = l;
        if (o1h) {                                   // advanced by one column: X1[k-1] = X[k]
          o1h[(long long)c * ldk + k - 1] = h;
          o1l[(long long)c * ldk + k - 1] = l;
        }
      }
    }
  }
}

}  // namespace

extern "C" int radmmm_transpose_split_act(const float* x, int ld, int C, int B, int T, int Tp, int front,
                                          const int32_t* lens, int mask_mode, float scale, void* oh, void* ol, void* o1h,
                                          void* o1l, int ldk, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && oh && ol, "transpose_split_act: null pointer");
  RADMMM_REQUIRE(C > 0 && B > 0 && T > 0 && Tp >= T && front >= 1 && ldk % 8 == 0 && ldk >= front + B * Tp,
                 "transpose_split_act: bad dims");
  // the output must be pre-zeroed by the caller (front columns, row tails); gaps are written here
  const int ty = (Tp + 63) / 64;
  hipLaunchKernelGGL(transpose_split_act_kernel, dim3((C + 31) / 32, ty, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ld, C, T, Tp, front, lens, mask_mode, scale,
                     static_cast<_Float16*>(oh), static_cast<_Float16*>(ol), static_cast<_Float16*>(o1h),
                     static_cast<_Float16*>(o1l), ldk);
  return radmmm::check_launch("transpose_split_act");
}

extern "C" int radmmm_wgrad_h3(const void* GYh, const void* GYl, const void* Xh, const void* Xl, const void* X1h,
                               const void* X1l, int ldk, int k0, int Kt, float* P, int ldp, int64_t split_stride, int Mc, int Nc,
                               int taps, int dil, int splits, float acc_scale, radmmm_stream_t stream) {
  RADMMM_REQUIRE(GYh && GYl && Xh && Xl && P, "wgrad_h3: null pointer");
  RADMMM_REQUIRE(Mc > 0 && Nc > 0 && taps >= 1 && dil >= 1 && splits >= 1 && Kt > 0 && Kt % BK == 0 && ldk % 8 == 0 && k0 % 8 == 0 &&
                     ldk >= k0 + Kt + (taps / 2) * dil && k0 >= (taps / 2) * dil,
                 "wgrad_h3: bad dims (Kt %% 32 == 0, ldk %% 8 == 0, k0 and the row tail must cover the largest tap shift)");
  bool has_odd = false;
  for (int t = 0; t < taps; ++t) has_odd = has_odd || (((t - taps / 2) * dil) & 1);
  RADMMM_REQUIRE(!has_odd || (X1h && X1l), "wgrad_h3: odd tap shifts need the advanced copy X1");
  const long long a_bytes = (long long)Mc * ldk * 2, b_bytes = (long long)Nc * ldk * 2;
  RADMMM_REQUIRE(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "wgrad_h3: operand >= 2 GiB");
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  WgradH3Args a;
  a.GYh = static_cast<const _Float16*>(GYh); a.GYl = static_cast<const _Float16*>(GYl);
  a.Xh = static_cast<const _Float16*>(Xh); a.Xl = static_cast<const _Float16*>(Xl);
  a.X1h = static_cast<const _Float16*>(X1h); a.X1l = static_cast<const _Float16*>(X1l);
  a.ldk = ldk; a.k0 = k0; a.Kt = Kt; a.P = P; a.ldp = ldp; a.split_stride = split_stride;
  a.Mc = Mc; a.Nc = Nc; a.taps = taps; a.dil = dil; a.splits = splits; a.acc_scale = acc_scale;
  a.a_bytes = (int)a_bytes; a.b_bytes = (int)b_bytes;
  const int ntm = (Mc + BM - 1) / BM, ntn = (Nc + BN - 1) / BN;
  hipLaunchKernelGGL(wgrad_h3_kernel, dim3(ntm * ntn * taps * splits), dim3(256), SMEM_BYTES,
                     static_cast<hipStream_t>(stream), a);
  return radmmm::check_launch("wgrad_h3");
}
