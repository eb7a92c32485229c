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
#include <math.h>
#include <cstdlib>

#include "common.h"
#include "split_pack.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

constexpr int BN = 256, BK = 32, ROWB = 64;
constexpr int OOB = 0x7fffffff;
constexpr int SGB_VMEM = 0x020, SGB_MFMA = 0x008, SGB_DSR = 0x100;
constexpr int LOOKAHEAD = 2;

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

// Same tile machine as rowgemm_h3w.hip's rowgemm_h3d_kernel (one workgroup per CU, (32*MB) x 256
// tile, 4 waves x (MB x 2) accumulators, swizzled 64-byte LDS rows filled by LDS-DMA, pinned
// software pipeline); see that file for the layout.  Here both operands are plain [rows][ldk]
// arrays, the tap is a constant column offset of the B operand and the K range is one split.
template <int MB>
struct Geo {
  static constexpr int BMR = MB * 32;
  static constexpr int A_BYTES = BMR * ROWB;
  static constexpr int B_BYTES = BN * ROWB;
  static constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int SMEM = 2 * STAGE + 4096;      // + dump area of the surplus A piece (PR == 1, odd MB)
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

// PR: MFMA products per fp32 product, 3 (split) or 1 (hi halves only), see rowgemm_h3w.hip
template <int MB, int PR>
struct Pieces {
  static constexpr int A = PR == 3 ? MB : (2 * MB + 3) / 4;
  static constexpr int B = PR == 3 ? 8 : 4;
  static constexpr int N = A + B;
};

template <int MB, int T, int PR>
__device__ __forceinline__ void pin_items() {
  constexpr int NT = 2 * MB, NPT = Pieces<MB, PR>::N;
  if constexpr (T < NT - LOOKAHEAD) {
    __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR == 3 ? 2 : 1, 0);
    if constexpr (T + LOOKAHEAD == MB) __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR == 3 ? 4 : 2, 0);
    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 2 * PR, 0);
    constexpr int lo = 2 * T, hi = (2 * (T + 1) < NPT) ? 2 * (T + 1) : NPT;
    if constexpr (hi > lo) __builtin_amdgcn_sched_group_barrier(SGB_VMEM, hi - lo, 0);
    pin_items<MB, T + 1, PR>();
  }
}

template <int MB, int I>
__device__ __forceinline__ void store_blocks(const f32x16 (&acc)[MB][2], float* smf, float* P, int ldp, int Mc, int Nc,
                                             float sc, int m0, int n0, int tid, int lane, int wave, bool vec_ok) {
  if constexpr (I < MB) {
    if (I > 0) radmmm::lds_barrier();      // LDS only: the previous block's global stores stay in flight
    float* wbase = smf + (4 * (lane >> 5)) * BN + wave * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) wbase[((e & 3) + 8 * (e >> 2)) * BN + j * 32] = acc[I][j][e] * sc;
    radmmm::lds_barrier();
    const int c4 = (tid & 63) * 4, col = n0 + c4;
#pragma unroll 4
    for (int k = 0; k < 8; ++k) {
      const int rl = k * 4 + (tid >> 6);
      const int row = m0 + I * 32 + rl;
      if (row < Mc && col < Nc) {
        const float4 a4 = *reinterpret_cast<const float4*>(smf + rl * BN + c4);
        if (vec_ok && col + 3 < Nc) {
          *reinterpret_cast<float4*>(P + (long long)row * ldp + col) = a4;
        } else {
          const float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < Nc) P[(long long)row * ldp + col + e] = v[e];
        }
      }
    }
    store_blocks<MB, I + 1>(acc, smf, P, ldp, Mc, Nc, sc, m0, n0, tid, lane, wave, vec_ok);
  }
}

template <int MB, int PR = 3>
__global__ __launch_bounds__(256, 1) void wgrad_h3_kernel(const WgradH3Args a) {
  using G = Geo<MB>;
  constexpr int NT = 2 * MB, D = LOOKAHEAD, NG = 2 * MB, NPA = Pieces<MB, PR>::A, NP = Pieces<MB, PR>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntm = (a.Mc + G::BMR - 1) / G::BMR, ntn = (a.Nc + BN - 1) / BN;
  const int nt = ntm * ntn * a.taps * a.splits, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  int id = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int tn = id % ntn; id /= ntn;
  const int tm = id % ntm; id /= ntm;
  const int tap = id % a.taps;
  const int split = id / a.taps;
  const int m0 = tm * G::BMR, n0 = tn * BN;
  const int shift = (tap - a.taps / 2) * a.dil;
  const bool odd = (shift & 1) != 0;                 // odd shift -> the advanced copy at shift - 1
  const int bshift = odd ? shift - 1 : shift;

  const int steps_total = a.Kt / BK;
  const int steps_per = (steps_total + a.splits - 1) / a.splits;
  const int step_lo = split * steps_per;
  int step_hi = step_lo + steps_per;
  if (step_hi > steps_total) step_hi = steps_total;
  const int nsteps = step_hi - step_lo;

  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int a_vo[NPA], a_dst[NPA], a_isl[NPA], b_vo[4], b_dst[4];
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int c = 4 * k + wave;
    a_isl[k] = (PR == 3 && c >= NG) ? 1 : 0;
    const bool real = PR == 3 || c < NG;               // PR == 1, odd MB: surplus piece -> zeros into the dump area
    const int j = a_isl[k] ? c - NG : c;
    const int r = m0 + 16 * j + d_row;
    a_vo[k] = (real && r < a.Mc) ? (r * a.ldk + d_chunk * 8 + a.k0) * 2 : OOB;
    a_dst[k] = real ? a_isl[k] * G::A_BYTES + j * 1024 : -1;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * k + wave;
    const int n = n0 + 16 * j + d_row;
    b_vo[k] = n < a.Nc ? (n * a.ldk + d_chunk * 8 + a.k0 + bshift) * 2 : OOB;
    b_dst[k] = 2 * G::A_BYTES + j * 1024;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYh), 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYl), 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(odd ? a.X1h : a.Xh), 0, a.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(odd ? a.X1l : a.Xl), 0, a.b_bytes, 0x00020000);

  // piece w of 0 .. MB+7 of K step `step` (absolute) into stage `buf`.  OOB + (step offset) stays
  // >= 2^31, i.e. out of range.
  auto dma_piece = [&](int buf, int w, int step) __attribute__((always_inline)) {
    const int sbase = buf * G::STAGE;
    const int koff = step * (BK * 2);
    if (w < NPA) {
      const int dst = a_dst[w] < 0 ? 2 * G::STAGE + wave * 1024 : sbase + a_dst[w];
      dma16((PR == 3 && a_isl[w]) ? rAl : rAh, (lds_u32_ptr)(sm + dst), a_vo[w] + koff);
    } else {
      const int k = (w - NPA) & 3, arr = (w - NPA) >> 2;
      dma16(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + sbase + b_dst[k] + arr * G::B_BYTES), b_vo[k] + koff);
    }
  };

  f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3;
  const int f_off0 = f_row + (((0 + (lane >> 5)) ^ f_swz) << 4);
  const int f_off1 = f_row + (((2 + (lane >> 5)) ^ f_swz) << 4);
  f16x8 fah[NT], fal[NT], bh[2][2], bl[2][2];
  auto read_a = [&](int bsel, int t) __attribute__((always_inline)) {
    const unsigned char* st = sm + bsel * G::STAGE;
    const int fo = (t >= MB) ? f_off1 : f_off0;
    const int i = t >= MB ? t - MB : t;
    fah[t] = *reinterpret_cast<const f16x8*>(st + i * 32 * ROWB + fo);
    if constexpr (PR == 3) fal[t] = *reinterpret_cast<const f16x8*>(st + G::A_BYTES + i * 32 * ROWB + fo);
  };
  auto read_b = [&](int bsel, int kb) __attribute__((always_inline)) {
    const unsigned char* sB = sm + bsel * G::STAGE + 2 * G::A_BYTES + wave * 64 * ROWB;
    const int fo = kb ? f_off1 : f_off0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + fo);
      if constexpr (PR == 3) bl[kb][j] = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + j * 32 * ROWB + fo);
    }
  };
  auto mfma_item = [&](int t) __attribute__((always_inline)) {
    const int kb = t >= MB ? 1 : 0, i = t >= MB ? t - MB : t;
    if constexpr (PR == 3) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[t], bh[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[t], bh[kb][1], acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bl[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bl[kb][1], acc[i][1], 0, 0, 0);
    }
    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][0], acc[i][0], 0, 0, 0);
    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][1], acc[i][1], 0, 0, 0);
  };

  if (nsteps > 0) {
#pragma unroll
    for (int w = 0; w < NP; ++w) dma_piece(0, w, step_lo);
    __syncthreads();
    read_b(0, 0);
#pragma unroll
    for (int t = 0; t < D; ++t) read_a(0, t);
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      const int nxt = step_lo + (s + 1 < nsteps ? s + 1 : s);   // clamped: the last step re-fetches itself
#pragma unroll
      for (int t = 0; t < NT - D; ++t) {
        read_a(buf, t + D);
        if (t + D == MB) read_b(buf, 1);
        mfma_item(t);
        if (2 * t < NP) dma_piece(buf ^ 1, 2 * t, nxt);
        if (2 * t + 1 < NP) dma_piece(buf ^ 1, 2 * t + 1, nxt);
      }
      pin_items<MB, 0, PR>();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      read_b(buf ^ 1, 0);
#pragma unroll
      for (int t = 0; t < D; ++t) read_a(buf ^ 1, t);
#pragma unroll
      for (int t = NT - D; t < NT; ++t) mfma_item(t);
      __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR == 3 ? 4 + 2 * D : 2 + D, 0);
      __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 2 * PR * D, 0);
    }
    __syncthreads();
  }

  float* P = a.P + (long long)split * a.split_stride + (long long)tap * a.Mc * a.ldp;
  const bool vec_ok = (a.ldp % 4 == 0) && radmmm::aligned16(a.P) && (a.split_stride % 4 == 0);
  store_blocks<MB, 0>(acc, reinterpret_cast<float*>(sm), P, a.ldp, a.Mc, a.Nc, a.acc_scale, m0, n0, tid, lane, wave, vec_ok);
}

template <int MB, int PR = 3>
int launch_wgrad(const WgradH3Args& a, hipStream_t stream) {
  using G = Geo<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h3_kernel<MB, PR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(wgrad_h3<%d>): %s", MB, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const int ntm = (a.Mc + G::BMR - 1) / G::BMR, ntn = (a.Nc + BN - 1) / BN;
  hipLaunchKernelGGL((wgrad_h3_kernel<MB, PR>), dim3(ntm * ntn * a.taps * a.splits), dim3(256), G::SMEM, stream, a);
  return radmmm::check_launch("wgrad_h3");
}

// rows per workgroup tile the kernel will use for Mc output channels: the largest MB whose last
// row tile is at least 3/4 full (1024 -> 8 x 32 = 256 rows)
int wgrad_mb(int Mc) {
  for (int mb = 8; mb > 4; --mb) {
    const int rows = 32 * mb, nt = (Mc + rows - 1) / rows;
    if (nt * rows - Mc <= rows / 4) return mb;
  }
  return 4;
}

// fp32 [B*T rows][ld] (first C columns) -> split, transposed, zero-gapped [C][ldk]:
//   out[c][FRONT + b*Tp + t] = split(scale * x[b*T + t][c]) for t < len_b (mask_mode 1) or t < T;
// everything else zero.  64 frames x 32 channels per block through LDS.
__global__ __launch_bounds__(256) void transpose_split_act_kernel(
    const float* __restrict__ x, int ld, int C, int T, int Tp, int front, const int* __restrict__ lens, int mask_mode,
    float scale, _Float16* __restrict__ oh, _Float16* __restrict__ ol, _Float16* __restrict__ o1h,
    _Float16* __restrict__ o1l, int ldk, float* __restrict__ part, int sum_weight, int sum_taps, int sum_dil, int vec4) {
  __shared__ float tile[64][33];
  __shared__ float red[8][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * 64, c0 = blockIdx.x * 32;
  const int len = (mask_mode && lens) ? lens[b] : T;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  float csum = 0.f;
  for (int i = ty; i < 64; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    const float raw = (t < T && c < C) ? x[((long long)b * T + t) * ld + c] : 0.f;
    tile[i][tx] = t < len ? raw * scale : 0.f;             // 0 in gap / masked frames
    if (part && t < T) csum = fmaf(radmmm::colsum_row_weight(b * T + t, sum_weight, T, lens, sum_taps, sum_dil), raw, csum);
  }
  if (part) red[ty][tx] = csum;
  __syncthreads();
  // bias-gradient by-product: this block's column sums (all T frames, own row weights) as one row of
  // partials; radmmm_colsum_final adds the rows in a fixed order
  if (part && ty == 0 && c0 + tx < C) {
    float t8 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t8 += red[w][tx];
    part[((long long)b * gridDim.y + blockIdx.y) * C + c0 + tx] = t8;
  }
  // lanes run along time: for one channel, 64 consecutive frames = 128 contiguous bytes per array.
  // Frames t in [T, Tp) are the zero gap after the utterance and are (re)written here as zeros.
  if (vec4) {
    // four consecutive frames per lane (8-byte stores; front, Tp multiples of 4 keep them aligned): a quarter of the
    // store instructions of the element-wise form below
    const int q = threadIdx.x & 15, cl0 = threadIdx.x >> 4;          // 16 frame quads x 16 channels per pass
    const int t = t0 + 4 * q;
    if (t < Tp) {                                                       // Tp % 4 == 0: whole quads
      const long long k = (long long)front + (long long)b * Tp + t;
      for (int ci = cl0; ci < 32; ci += 16) {
        const int c = c0 + ci;
        if (c < C) {
          radmmm::f16x4_t h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = radmmm::clamp_f16(tile[4 * q + e][ci]);
            h[e] = (_Float16)v;
            l[e] = (_Float16)(v - (float)h[e]);
          }
          *reinterpret_cast<radmmm::f16x4_t*>(oh + (long long)c * ldk + k) = h;
          *reinterpret_cast<radmmm::f16x4_t*>(ol + (long long)c * ldk + k) = l;
          if (o1h) {                                 // advanced by one column: X1[k-1] = X[k] (2-byte aligned only)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o1h[(long long)c * ldk + k - 1 + e] = h[e];
              o1l[(long long)c * ldk + k - 1 + e] = l[e];
            }
          }
        }
      }
    }
    return;
  }
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

// The same pass on 64-frame x 64-channel tiles with 16-byte loads (256 contiguous bytes per frame row instead of 128)
// and 8-byte transposed stores: the shape every flow-step call has (C % 4 == 0, front / Tp / ldk multiples of 4).
// 105 MB of traffic per call at the benchmark size; measured against the 32-channel kernel above in DESIGN.md §4.6.
__global__ __launch_bounds__(256) void transpose_split_act64_kernel(
    const float* __restrict__ x, int ld, int C, int T, int Tp, int front, const int* __restrict__ lens, int mask_mode,
    float scale, _Float16* __restrict__ oh, _Float16* __restrict__ ol, _Float16* __restrict__ o1h,
    _Float16* __restrict__ o1l, int ldk, float* __restrict__ part, int sum_weight, int sum_taps, int sum_dil) {
  __shared__ float tile[64][65];
  __shared__ float red[16][65];
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int len = (mask_mode && lens) ? lens[b] : T;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 float4 columns x 16 rows per pass
  const int c = c0 + tx * 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tl = i * 16 + ty, t = t0 + tl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T && c < C) {                                       // C % 4 == 0: whole groups of 4 columns
      v = *reinterpret_cast<const float4*>(x + ((long long)b * T + t) * ld + c);
      if (part) {
        const float w = radmmm::colsum_row_weight(b * T + t, sum_weight, T, lens, sum_taps, sum_dil);
        s0 = fmaf(w, v.x, s0); s1 = fmaf(w, v.y, s1); s2 = fmaf(w, v.z, s2); s3 = fmaf(w, v.w, s3);
      }
    }
    const float m = t < len ? scale : 0.f;                      // 0 in gap / masked frames
    tile[tl][tx * 4 + 0] = v.x * m;
    tile[tl][tx * 4 + 1] = v.y * m;
    tile[tl][tx * 4 + 2] = v.z * m;
    tile[tl][tx * 4 + 3] = v.w * m;
  }
  if (part) {
    red[ty][tx * 4 + 0] = s0; red[ty][tx * 4 + 1] = s1; red[ty][tx * 4 + 2] = s2; red[ty][tx * 4 + 3] = s3;
  }
  __syncthreads();
  if (part && threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float t16 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t16 += red[w][threadIdx.x];
    part[((long long)b * gridDim.y + blockIdx.y) * C + c0 + threadIdx.x] = t16;
  }
  const int q = threadIdx.x & 15, cl0 = threadIdx.x >> 4;      // 16 frame quads x 16 channels per pass
  const int t = t0 + 4 * q;
  if (t < Tp) {                                                 // Tp % 4 == 0: whole quads
    const long long k = (long long)front + (long long)b * Tp + t;
#pragma unroll
    for (int ci = cl0; ci < 64; ci += 16) {
      const int cc = c0 + ci;
      if (cc < C) {
        radmmm::f16x4_t h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = radmmm::clamp_f16(tile[4 * q + e][ci]);
          h[e] = (_Float16)v;
          l[e] = (_Float16)(v - (float)h[e]);
        }
        *reinterpret_cast<radmmm::f16x4_t*>(oh + (long long)cc * ldk + k) = h;
        *reinterpret_cast<radmmm::f16x4_t*>(ol + (long long)cc * ldk + k) = l;
        if (o1h) {                                   // advanced by one column: X1[k-1] = X[k] (2-byte aligned only)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o1h[(long long)cc * ldk + k - 1 + e] = h[e];
            o1l[(long long)cc * ldk + k - 1 + e] = l[e];
          }
        }
      }
    }
  }
}

// y = g * act'(saved) (the step from dL/d(act output) to dL/d(conv accumulator), radmmm_dact_mul without row weights)
// written THREE ways in one pass over g and saved, with no fp32 copy of y: (1) the row-major split pair that feeds the
// data-gradient GEMM (any split format, saturation flag), (2) the transposed zero-gapped split-f16 pair that feeds the
// weight-gradient GEMM (radmmm_transpose_split_act's layout, no mask, no advanced copy: the consumer is a 1-tap
// gradient), (3) the column sums of y (bias gradient) as one row of partials per block.  Replaces radmmm_dact_mul +
// radmmm_transpose_split_act_colsum for the res/skip branch: 104 MB less HBM traffic per layer at the benchmark size.
// 64 frames x 64 channels per workgroup.
__global__ __launch_bounds__(256) void dact_transposed_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ saved, int lds, int C, int T, int Tp, int front, int dact,
    float scale, void* __restrict__ yh, void* __restrict__ yl, int ldyh, int fmt, float x8_mul, int* __restrict__ sat_flag,
    _Float16* __restrict__ oh, _Float16* __restrict__ ol, int ldk, float* __restrict__ part, int vec4,
    void* __restrict__ ylo16, int rowscale, const int* __restrict__ lens, int taps, int dil) {
  // rowscale (radmmm_dact_mul's): 0 none; 1: y *= [t < len]; 2: y *= [t < len] * partial-conv ratio(t).  The column sums are
  // then those of g * act' * [t < len] -- what radmmm_colsum(y, row_weight = rowscale) returns (ratio x its inverse weight)
  __shared__ float tile[64][65];
  __shared__ float red[16][65];
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 float4 columns x 16 rows per pass
  const int c = c0 + tx * 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sat = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tl = i * 16 + ty, t = t0 + tl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T && c < C) {                                       // C % 4 == 0: whole groups of 4 columns
      const long long row = (long long)b * T + t;
      const float4 gv = *reinterpret_cast<const float4*>(g + row * ldg + c);
      v = gv;
      if (dact) {
        const float4 sv = *reinterpret_cast<const float4*>(saved + row * lds + c);
        v.x = gv.x * radmmm::dact_from_out(sv.x, dact);
        v.y = gv.y * radmmm::dact_from_out(sv.y, dact);
        v.z = gv.z * radmmm::dact_from_out(sv.z, dact);
        v.w = gv.w * radmmm::dact_from_out(sv.w, dact);
      }
      if (rowscale) {
        const int len = lens ? lens[b] : T;
        const float mk = t < len ? 1.f : 0.f;
        s0 = fmaf(v.x, mk, s0); s1 = fmaf(v.y, mk, s1); s2 = fmaf(v.z, mk, s2); s3 = fmaf(v.w, mk, s3);
        const float rs = rowscale == 2 ? mk * radmmm::pconv_ratio(t, len, taps, dil) : mk;
        v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
      } else {
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
      }
      if (yh) sat = fmaxf(sat, radmmm::store_split4_fmt(yh, yl, row * ldyh, c, fmt, x8_mul, scale, v.x, v.y, v.z, v.w, ylo16));
    }
    if (oh) {                                                     // (no transposed copy wanted: nothing to stage)
      tile[tl][tx * 4 + 0] = v.x * scale;
      tile[tl][tx * 4 + 1] = v.y * scale;
      tile[tl][tx * 4 + 2] = v.z * scale;
      tile[tl][tx * 4 + 3] = v.w * scale;
    }
  }
  red[ty][tx * 4 + 0] = s0; red[ty][tx * 4 + 1] = s1; red[ty][tx * 4 + 2] = s2; red[ty][tx * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
    float t16 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t16 += red[w][threadIdx.x];
    part[((long long)b * gridDim.y + blockIdx.y) * C + c0 + threadIdx.x] = t16;
  }
  // transposed copy: lanes run along time (64 consecutive frames of one channel = 128 contiguous bytes per array);
  // frames in [T, Tp) are the zero gap after the utterance and are (re)written here as zeros (tile rows hold zeros there)
  if (!oh) {
    // no transposed copy wanted (the weight gradient contracts the row-major pair: radmmm_wgrad_rm)
  } else if (vec4) {
    const int q = threadIdx.x & 15, cl0 = threadIdx.x >> 4;          // 16 frame quads x 16 channels per pass, 8-byte stores
    const int t = t0 + 4 * q;
    if (t < Tp) {
      const long long k = (long long)front + (long long)b * Tp + t;
      for (int ci = cl0; ci < 64; ci += 16) {
        const int cc = c0 + ci;
        if (cc < C) {
          radmmm::f16x4_t h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = radmmm::clamp_f16(tile[4 * q + e][ci]);
            h[e] = (_Float16)v;
            l[e] = (_Float16)(v - (float)h[e]);
          }
          *reinterpret_cast<radmmm::f16x4_t*>(oh + (long long)cc * ldk + k) = h;
          *reinterpret_cast<radmmm::f16x4_t*>(ol + (long long)cc * ldk + k) = l;
        }
      }
    }
  } else {
    const int fl = threadIdx.x & 63, cl0 = threadIdx.x >> 6;
    const int t = t0 + fl;
    if (t < Tp) {
      const long long k = (long long)front + (long long)b * Tp + t;
      for (int ci = cl0; ci < 64; ci += 4) {
        const int cc = c0 + ci;
        if (cc < C) {
          const float v = radmmm::clamp_f16(tile[fl][ci]);
          const _Float16 h = (_Float16)v;
          oh[(long long)cc * ldk + k] = h;
          ol[(long long)cc * ldk + k] = (_Float16)(v - (float)h);
        }
      }
    }
  }
  radmmm::raise_sat_flag(sat_flag, sat, fmt ? x8_mul : 0.f);
}

// radmmm_dact_mul_rows_multi: y_j = g * act'(saved_j) for up to four saved tensors in ONE pass over g (round 5: the four
// res/skip layers of a WN share the gradient of the skip sum, 52 MB that four separate passes read four times).  Per j: the
// row-major split pair (+ optional fp16 lo part) and one row of column-sum partials per (utterance, 64-frame block), as
// dact_transposed_kernel writes them with no row scale and no transposed copy -- same arithmetic per element.
struct DactMulti {
  const float* saved[4];
  void* yh[4];
  void* yl[4];
  void* lo16[4];
  float* part[4];
};

template <int NJ>
__global__ __launch_bounds__(256) void dact_rows_multi_kernel(const float* __restrict__ g, int ldg, const DactMulti a, int lds,
                                                              int C, int T, int dact, float scale, int ldyh, int fmt,
                                                              float x8_mul, int* __restrict__ sat_flag) {
  __shared__ float red[NJ][16][65];
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = c0 + tx * 4;
  float s[NJ][4], sat = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) s[j][0] = s[j][1] = s[j][2] = s[j][3] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + i * 16 + ty;
    if (t < T && c < C) {
      const long long row = (long long)b * T + t;
      const float4 gv = *reinterpret_cast<const float4*>(g + row * ldg + c);
      float4 sv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) sv[j] = *reinterpret_cast<const float4*>(a.saved[j] + row * lds + c);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float4 v;
        v.x = gv.x * radmmm::dact_from_out(sv[j].x, dact);
        v.y = gv.y * radmmm::dact_from_out(sv[j].y, dact);
        v.z = gv.z * radmmm::dact_from_out(sv[j].z, dact);
        v.w = gv.w * radmmm::dact_from_out(sv[j].w, dact);
        s[j][0] += v.x; s[j][1] += v.y; s[j][2] += v.z; s[j][3] += v.w;
        sat = fmaxf(sat, radmmm::store_split4_fmt(a.yh[j], a.yl[j], row * ldyh, c, fmt, x8_mul, scale, v.x, v.y, v.z, v.w, a.lo16[j]));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    red[j][ty][tx * 4 + 0] = s[j][0]; red[j][ty][tx * 4 + 1] = s[j][1]; red[j][ty][tx * 4 + 2] = s[j][2]; red[j][ty][tx * 4 + 3] = s[j][3];
  }
  __syncthreads();
  {
    const int j = threadIdx.x >> 6, cl = threadIdx.x & 63;       // 4 x 64 threads: one of the (up to four) sums per 64-thread group
    if (j < NJ && c0 + cl < C) {
      float t16 = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t16 += red[j][w][cl];
      a.part[j][((long long)b * gridDim.y + blockIdx.y) * C + c0 + cl] = t16;
    }
  }
  radmmm::raise_sat_flag(sat_flag, sat, fmt ? x8_mul : 0.f);
}

}  // namespace

extern "C" int radmmm_dact_mul_rows_multi(const float* g, int ldg, const radmmm_dact_item* items, int n, int lds, int C, int B,
                                          int T, int dact, float scale, int ldyh, const radmmm_split_opts* so,
                                          radmmm_stream_t stream) {
  RADMMM_REQUIRE(g && items && n >= 1 && n <= 4 && dact, "dact_mul_rows_multi: null pointer / 1 .. 4 items / an activation");
  RADMMM_REQUIRE(C > 0 && C % 4 == 0 && B > 0 && T > 0 && ldg >= C && ldg % 4 == 0 && lds >= C && lds % 4 == 0,
                 "dact_mul_rows_multi: bad dims (C, ldg, lds multiples of 4)");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(ldyh >= C && ldyh % 4 == 0 && (fmt == RADMMM_SPLIT_F16 || ldyh % 32 == 0),
                 "dact_mul_rows_multi: row-major split output (ldyh %% 4 == 0; 8-bit formats: %% 32)");
  RADMMM_REQUIRE(radmmm::aligned16(g), "dact_mul_rows_multi: 16-byte aligned g");
  DactMulti a{};
  for (int j = 0; j < n; ++j) {
    RADMMM_REQUIRE(items[j].saved && items[j].yh && items[j].yl && items[j].part && radmmm::aligned16(items[j].saved),
                   "dact_mul_rows_multi: item with a null pointer / unaligned saved tensor");
    a.saved[j] = items[j].saved; a.yh[j] = items[j].yh; a.yl[j] = items[j].yl; a.lo16[j] = items[j].ylo16; a.part[j] = items[j].part;
  }
  const dim3 grid((C + 63) / 64, (T + 63) / 64, B);
  const float x8_mul = ldexpf(1.f, so ? so->x8_exp : 0);
  int* flag = so ? so->sat_flag : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (n) {
    case 1: hipLaunchKernelGGL(dact_rows_multi_kernel<1>, grid, dim3(256), 0, st, g, ldg, a, lds, C, T, dact, scale, ldyh, fmt, x8_mul, flag); break;
    case 2: hipLaunchKernelGGL(dact_rows_multi_kernel<2>, grid, dim3(256), 0, st, g, ldg, a, lds, C, T, dact, scale, ldyh, fmt, x8_mul, flag); break;
    case 3: hipLaunchKernelGGL(dact_rows_multi_kernel<3>, grid, dim3(256), 0, st, g, ldg, a, lds, C, T, dact, scale, ldyh, fmt, x8_mul, flag); break;
    default: hipLaunchKernelGGL(dact_rows_multi_kernel<4>, grid, dim3(256), 0, st, g, ldg, a, lds, C, T, dact, scale, ldyh, fmt, x8_mul, flag); break;
  }
  return radmmm::check_launch("dact_mul_rows_multi");
}

static int launch_transpose(const float* x, int ld, int C, int B, int T, int Tp, int front, const int32_t* lens,
                            int mask_mode, float scale, void* oh, void* ol, void* o1h, void* o1l, int ldk, float* part,
                            int sum_weight, int sum_taps, int sum_dil, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && oh && ol, "transpose_split_act: null pointer");
  RADMMM_REQUIRE(C > 0 && B > 0 && T > 0 && Tp >= T && front >= 1 && ldk % 8 == 0 && ldk >= front + B * Tp,
                 "transpose_split_act: bad dims");
  // the output must be pre-zeroed by the caller (front columns, row tails); gaps are written here
  const int ty = (Tp + 63) / 64;
  const bool vec_out = front % 4 == 0 && Tp % 4 == 0 && ldk % 4 == 0 && (reinterpret_cast<uintptr_t>(oh) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(ol) & 7) == 0;
  static const bool narrow = radmmm::debug_env("RADMMM_TRANSPOSE32") != nullptr;       // A/B switch: the 32-channel kernel
  if (vec_out && !narrow && C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    hipLaunchKernelGGL(transpose_split_act64_kernel, dim3((C + 63) / 64, ty, B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ld, C, T, Tp, front, lens, mask_mode, scale,
                       static_cast<_Float16*>(oh), static_cast<_Float16*>(ol), static_cast<_Float16*>(o1h),
                       static_cast<_Float16*>(o1l), ldk, part, sum_weight, sum_taps, sum_dil);
    return radmmm::check_launch("transpose_split_act");
  }
  hipLaunchKernelGGL(transpose_split_act_kernel, dim3((C + 31) / 32, ty, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ld, C, T, Tp, front, lens, mask_mode, scale,
                     static_cast<_Float16*>(oh), static_cast<_Float16*>(ol), static_cast<_Float16*>(o1h),
                     static_cast<_Float16*>(o1l), ldk, part, sum_weight, sum_taps, sum_dil, vec_out ? 1 : 0);
  return radmmm::check_launch("transpose_split_act");
}

extern "C" int radmmm_transpose_split_act(const float* x, int ld, int C, int B, int T, int Tp, int front,
                                          const int32_t* lens, int mask_mode, float scale, void* oh, void* ol, void* o1h,
                                          void* o1l, int ldk, radmmm_stream_t stream) {
  return launch_transpose(x, ld, C, B, T, Tp, front, lens, mask_mode, scale, oh, ol, o1h, o1l, ldk, nullptr, 0, 1, 1, stream);
}

// Same, plus the weighted column sums of x (bias gradient) as a by-product of the one pass over x:
// part [B * ceil(Tp / 64)][C] partial rows (radmmm_colsum_final adds them).  sum_weight / taps / dil
// as radmmm_colsum's row_weight (0 plain, 1 length mask, 2 mask x partial-conv ratio), independent
// of mask_mode (which governs the transposed copy).
extern "C" int radmmm_transpose_split_act_colsum(const float* x, int ld, int C, int B, int T, int Tp, int front,
                                                 const int32_t* lens, int mask_mode, float scale, void* oh, void* ol,
                                                 void* o1h, void* o1l, int ldk, float* part, int sum_weight, int sum_taps,
                                                 int sum_dil, radmmm_stream_t stream) {
  RADMMM_REQUIRE(part, "transpose_split_act_colsum: null partial buffer");
  RADMMM_REQUIRE(sum_weight == 0 || sum_weight == 1 || (sum_weight == 2 && sum_taps >= 1 && sum_dil >= 1),
                 "transpose_split_act_colsum: bad row weight");
  return launch_transpose(x, ld, C, B, T, Tp, front, lens, mask_mode, scale, oh, ol, o1h, o1l, ldk, part, sum_weight,
                          sum_taps, sum_dil, stream);
}

extern "C" int radmmm_wgrad_h3_tiles(int Mc, int Nc, int taps) {
  if (Mc <= 0 || Nc <= 0 || taps <= 0) return 0;
  const int rows = 32 * wgrad_mb(Mc);
  return ((Mc + rows - 1) / rows) * ((Nc + BN - 1) / BN) * taps;
}

extern "C" int radmmm_wgrad_h3(const void* GYh, const void* GYl, const void* Xh, const void* Xl, const void* X1h,
                               const void* X1l, int ldk, int k0, int Kt, float* P, int ldp, int64_t split_stride, int Mc,
                               int Nc, int taps, int dil, int splits, float acc_scale, int nprod, radmmm_stream_t stream) {
  RADMMM_REQUIRE(GYh && GYl && Xh && Xl && P, "wgrad_h3: null pointer");
  RADMMM_REQUIRE(Mc > 0 && Nc > 0 && taps >= 1 && dil >= 1 && splits >= 1 && Kt > 0 && Kt % BK == 0 && ldk % 8 == 0 && k0 % 8 == 0 &&
                     ldk >= k0 + Kt + (taps / 2) * dil && k0 >= (taps / 2) * dil,
                 "wgrad_h3: bad dims (Kt %% 32 == 0, ldk %% 8 == 0, k0 and the row tail must cover the largest tap shift)");
  bool has_odd = false;
  for (int t = 0; t < taps; ++t) has_odd = has_odd || (((t - taps / 2) * dil) & 1);
  RADMMM_REQUIRE(!has_odd || (X1h && X1l), "wgrad_h3: odd tap shifts need the advanced copy X1");
  const long long a_bytes = (long long)Mc * ldk * 2, b_bytes = (long long)Nc * ldk * 2;
  RADMMM_REQUIRE(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "wgrad_h3: operand >= 2 GiB");
  WgradH3Args a;
  a.GYh = static_cast<const _Float16*>(GYh); a.GYl = static_cast<const _Float16*>(GYl);
  a.Xh = static_cast<const _Float16*>(Xh); a.Xl = static_cast<const _Float16*>(Xl);
  a.X1h = static_cast<const _Float16*>(X1h); a.X1l = static_cast<const _Float16*>(X1l);
  a.ldk = ldk; a.k0 = k0; a.Kt = Kt; a.P = P; a.ldp = ldp; a.split_stride = split_stride;
  a.Mc = Mc; a.Nc = Nc; a.taps = taps; a.dil = dil; a.splits = splits; a.acc_scale = acc_scale;
  a.a_bytes = (int)a_bytes; a.b_bytes = (int)b_bytes;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nprod == 1) {                                    // 16-bit throughput mode: hi halves only
    switch (wgrad_mb(Mc)) {
      case 8: return launch_wgrad<8, 1>(a, st);
      case 7: return launch_wgrad<7, 1>(a, st);
      case 6: return launch_wgrad<6, 1>(a, st);
      case 5: return launch_wgrad<5, 1>(a, st);
      default: return launch_wgrad<4, 1>(a, st);
    }
  }
  switch (wgrad_mb(Mc)) {
    case 8: return launch_wgrad<8>(a, st);
    case 7: return launch_wgrad<7>(a, st);
    case 6: return launch_wgrad<6>(a, st);
    case 5: return launch_wgrad<5>(a, st);
    default: return launch_wgrad<4>(a, st);
  }
}


extern "C" int radmmm_dact_mul_transposed(const float* g, int ldg, const float* saved, int lds, int C, int B, int T, int Tp,
                                          int front, int dact, float scale, void* yh, void* yl, int ldyh,
                                          const radmmm_split_opts* so, void* oh, void* ol, int ldk, float* part,
                                          radmmm_stream_t stream) {
  RADMMM_REQUIRE(g && (oh != nullptr) == (ol != nullptr) && part && (saved || !dact), "dact_mul_transposed: null pointer");
  RADMMM_REQUIRE(C > 0 && C % 4 == 0 && B > 0 && T > 0 && Tp >= T && front >= 1 && (!oh || (ldk % 8 == 0 && ldk >= front + B * Tp)) &&
                     ldg >= C && ldg % 4 == 0 && (!dact || (lds >= C && lds % 4 == 0)),
                 "dact_mul_transposed: bad dims (C, ldg, lds multiples of 4)");
  RADMMM_REQUIRE(radmmm::aligned16(g) && (!dact || radmmm::aligned16(saved)), "dact_mul_transposed: 16-byte aligned inputs");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(!yh || (yl && ldyh >= C && ldyh % 4 == 0 && (fmt == RADMMM_SPLIT_F16 || ldyh % 32 == 0)),
                 "dact_mul_transposed: row-major split output (ldyh %% 4 == 0; 8-bit formats: %% 32)");
  hipLaunchKernelGGL(dact_transposed_kernel, dim3((C + 63) / 64, (Tp + 63) / 64, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, ldg, saved, lds, C, T, Tp, front, dact, scale, yh, yl, ldyh, fmt,
                     ldexpf(1.f, so ? so->x8_exp : 0), so ? so->sat_flag : nullptr, static_cast<_Float16*>(oh),
                     static_cast<_Float16*>(ol), ldk, part,
                     (front % 4 == 0 && Tp % 4 == 0 && ldk % 4 == 0 && (reinterpret_cast<uintptr_t>(oh) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(ol) & 7) == 0) ? 1 : 0,
                     so ? so->lo16 : nullptr, 0, nullptr, 1, 1);
  return radmmm::check_launch("dact_mul_transposed");
}

// radmmm_dact_mul without the fp32 copy of y and with its bias-gradient sums: y = g * act'(saved) * row factor as the
// row-major split pair only, the column sums of g * act' * [t < len] as partial rows (include/radmmm_hip.h)
extern "C" int radmmm_dact_mul_rows(const float* g, int ldg, const float* saved, int lds, int C, int B, int T, int dact,
                                    int rowscale, const int32_t* lens, int taps, int dil, float scale, void* yh, void* yl,
                                    int ldyh, const radmmm_split_opts* so, float* part, radmmm_stream_t stream) {
  RADMMM_REQUIRE(g && yh && yl && part && (saved || !dact), "dact_mul_rows: null pointer");
  RADMMM_REQUIRE(C > 0 && C % 4 == 0 && B > 0 && T > 0 && ldg >= C && ldg % 4 == 0 && (!dact || (lds >= C && lds % 4 == 0)) &&
                     rowscale >= 0 && rowscale <= 2 && (rowscale != 2 || (taps >= 1 && dil >= 1)),
                 "dact_mul_rows: bad dims (C, ldg, lds multiples of 4)");
  RADMMM_REQUIRE(radmmm::aligned16(g) && (!dact || radmmm::aligned16(saved)), "dact_mul_rows: 16-byte aligned inputs");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(ldyh >= C && ldyh % 4 == 0 && (fmt == RADMMM_SPLIT_F16 || ldyh % 32 == 0),
                 "dact_mul_rows: row-major split output (ldyh %% 4 == 0; 8-bit formats: %% 32)");
  hipLaunchKernelGGL(dact_transposed_kernel, dim3((C + 63) / 64, (T + 63) / 64, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, ldg, saved, lds, C, T, T, 0, dact, scale, yh, yl, ldyh, fmt,
                     ldexpf(1.f, so ? so->x8_exp : 0), so ? so->sat_flag : nullptr, static_cast<_Float16*>(nullptr),
                     static_cast<_Float16*>(nullptr), 0, part, 0, so ? so->lo16 : nullptr, rowscale, lens, taps > 0 ? taps : 1,
                     dil > 0 ? dil : 1);
  return radmmm::check_launch("dact_mul_rows");
}
