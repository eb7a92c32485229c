// EXPERIMENTAL (round-1 probe, not on the product path yet): fp32-class GEMM on the f16 matrix
// cores by operand splitting.
//
//   x*s = hi + lo,  hi = fp16(x*s),  lo = fp16(x*s - hi)      (s a power of two: exact scaling)
//   A.B ~= Ah.Bh + Ah.Bl + Al.Bh   (fp32 accumulate in the MFMA; the dropped Al.Bl term is 2^-22)
//
// Three v_mfma_f32_32x32x16_f16 per 16-deep k block replace 8 v_mfma_f32_32x32x2_f32: 3/16 of the
// fp32-MFMA issue cycles for ~1e-6 relative accuracy, and the f16 MFMA pipe does not share the
// VALU datapath.  This file holds the split kernel and a plain NT GEMM (C = A B^T, both operands
// K-contiguous) used to measure what the approach delivers before the conv family is moved onto
// it.
//
// Tile 128x128x32, 4 waves (2x2) x (2x2) MFMA tiles.  LDS rows are 32 halves = 64 B padded to an
// 80-B pitch: 80 B = 20 banks and r -> 5r (mod 16) is a bijection, so the 16 rows a ds_read_b128
// lane group touches start in 16 distinct 4-bank slots (conflict-free).
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 40;                       // halves per LDS row (80 B)
constexpr int TILE_H = BM * PITCH;              // halves per operand tile
constexpr int SMEM_BYTES = 2 * 4 * TILE_H * 2;  // 2 buffers x {Ah, Al, Bh, Bl} = 80 KiB
constexpr int OOB = 0x7fffffff;

__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int ld, __half* __restrict__ hi,
                                                        __half* __restrict__ lo, int ldh, int rows, int cols,
                                                        float scale) {
  const int c4n = ldh / 4;
  const long long total = (long long)rows * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4n), c = (int)(i - (long long)r * c4n) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (c + e < cols) ? x[(long long)r * ld + c + e] * scale : 0.f;
    __half h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = __float2half_rn(v[e]);
      l[e] = __float2half_rn(v[e] - __half2float(h[e]));
    }
    *reinterpret_cast<uint2*>(hi + (long long)r * ldh + c) = *reinterpret_cast<uint2*>(h);
    *reinterpret_cast<uint2*>(lo + (long long)r * ldh + c) = *reinterpret_cast<uint2*>(l);
  }
}

__global__ __launch_bounds__(256, 2) void h3gemm_nt_kernel(const __half* __restrict__ Ah, const __half* __restrict__ Al,
                                                            int lda, const __half* __restrict__ Bh,
                                                            const __half* __restrict__ Bl, int ldb, float* __restrict__ C,
                                                            int ldc, int M, int N, int K, float out_scale, int a_bytes,
                                                            int b_bytes) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smh[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
  // XCD-aware remap (see gemm_f32.hip)
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, q = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nsteps = K / BK;

  // staging: 128 rows x 4 chunks of 16 B per operand tile = 512 chunks -> 2 per thread
  const int s_row = tid >> 2, s_chunk = tid & 3;       // rows s_row, s_row + 64
  int a_voff[2], b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = m0 + s_row + 64 * i, rb = n0 + s_row + 64 * i;
    a_voff[i] = ra < M ? (ra * lda + s_chunk * 8) * 2 : OOB;
    b_voff[i] = rb < N ? (rb * ldb + s_chunk * 8) * 2 : OOB;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(Bl), 0, b_bytes, 0x00020000);

  struct Regs {
    u32x4 v[4][2];   // {Ah, Al, Bh, Bl} x 2 rows
  };
  auto load_tiles = [&](int step, Regs& R) __attribute__((always_inline)) {
    const int so = step * (BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      R.v[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rAh, a_voff[i], so, 0);
      R.v[1][i] = __builtin_amdgcn_raw_buffer_load_b128(rAl, a_voff[i], so, 0);
      R.v[2][i] = __builtin_amdgcn_raw_buffer_load_b128(rBh, b_voff[i], so, 0);
      R.v[3][i] = __builtin_amdgcn_raw_buffer_load_b128(rBl, b_voff[i], so, 0);
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

  // fragment: lane l holds row (l & 31), k = 8*(l >> 5) .. +7 of the 16-deep block
  const int f_off = (lane & 31) * PITCH + (lane >> 5) * 8;
  Regs R;
  load_tiles(0, R);
  store_tiles(0, R);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    const int nxt = step + 1 < nsteps ? step + 1 : step;
    load_tiles(nxt, R);
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

  // epilogue: C/D layout of 32x32: lane l, register e -> column l&31, row (e&3) + 8*(e>>2) + 4*(l>>5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
        if (row < M && col < N) C[(long long)row * ldc + col] = acc[mi][ni][e] * out_scale;
      }
}

}  // namespace

extern "C" int radmmm_split_f16(const float* x, int ld, void* hi, void* lo, int ldh, int rows, int cols, float scale,
                                radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && hi && lo, "split_f16: null pointer");
  RADMMM_REQUIRE(rows > 0 && cols > 0 && ld >= cols && ldh >= cols && ldh % 8 == 0, "split_f16: bad dims (ldh %% 8 == 0)");
  long long total = (long long)rows * (ldh / 4);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_f16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld,
                     static_cast<__half*>(hi), static_cast<__half*>(lo), ldh, rows, cols, scale);
  return radmmm::check_launch("split_f16");
}

extern "C" int radmmm_h3gemm_nt(const void* Ah, const void* Al, int lda, const void* Bh, const void* Bl, int ldb,
                                float* C, int ldc, int M, int N, int K, float out_scale, radmmm_stream_t stream) {
  RADMMM_REQUIRE(Ah && Al && Bh && Bl && C, "h3gemm_nt: null pointer");
  RADMMM_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0, "h3gemm_nt: K must be a multiple of 32");
  RADMMM_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N, "h3gemm_nt: bad leading dims");
  const long long a_bytes = (long long)M * lda * 2, b_bytes = (long long)N * ldb * 2;
  RADMMM_REQUIRE(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "h3gemm_nt: operand >= 2 GiB");
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(h3gemm_nt_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  hipLaunchKernelGGL(h3gemm_nt_kernel, dim3(ntm * ntn), dim3(256), SMEM_BYTES, static_cast<hipStream_t>(stream),
                     static_cast<const __half*>(Ah), static_cast<const __half*>(Al), lda, static_cast<const __half*>(Bh),
                     static_cast<const __half*>(Bl), ldb, C, ldc, M, N, K, out_scale, (int)a_bytes, (int)b_bytes);
  return radmmm::check_launch("h3gemm_nt");
}
