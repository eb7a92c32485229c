// rowgemm_win8: the shared-window 5-tap GEMM (rowgemm_win.hip) with TWO WAVES PER SIMD, round 3.
//
// Why: an LDS-DMA instruction costs the wave that issues it 60-185 cycles of issue time (MI355X_MICROARCH.md, constants
// table; tools/mfma_dma_mix.hip: the K step's MFMA and DMA times ADD in a wave that is alone on its SIMD) -- and a wave that
// is alone on its SIMD has nobody to fill that time.  With two waves per SIMD the partner's MFMAs do (probe: 56 MFMA + 14
// pieces per SIMD and step take 1.62 us as one wave, 1.44 us as two).  Every wave of a kernel gets the same register
// budget, so two waves per SIMD means <= 256 registers each: the 32 MB x 256 tile is split by ROWS -- wave (cg, rh) owns
// row blocks [I0, I0 + MBW) of column group cg (MB = 7: 4 + 3 blocks, 8 or 6 accumulators of 32x32) -- which leaves the A
// fragment reads as they were (each A row is still read by four waves) and doubles only the B fragment reads (+32 KiB per
// K step and CU).  The two waves of a column group share their B rows, so the step has a barrier again (as rowgemm_h3d);
// everything else -- two-segment window with zero halos, fragment addressing, interleaved B rows, direct epilogue, extra K
// segment after the tap slices -- is rowgemm_win's, and the MFMAs of an output element run in the same order: the results
// are bit-identical to both other kernels (tests/test_hip_round3.py).
// MEASURED: no gain -- in_layer forward 296 us against 290 us with four waves.  Kept as a debug variant
// (RADMMM_WIN8=1 under RADMMM_DEBUG; epilogue kinds PLAIN / SPLIT only).
#include <type_traits>

#include "rowgemm_h3w_kernel.h"

namespace {

constexpr int WTAPS = 5, WDMAX = 8, NWV = 8;

template <int MB>
struct WGeo8 {
  static constexpr int BMR = MB * 32;
  static constexpr int WR = BMR + 8 * WDMAX;
  static constexpr int WP = WR / 16;
  static constexpr int B_BYTES = BN * ROWB;
  static constexpr int B_STAGE = 2 * B_BYTES;
  static constexpr int W_BASE = 2 * B_STAGE;
  static constexpr int W_PLANE = WR * ROWB;
  static constexpr int W_BYTES = 2 * W_PLANE;
  static constexpr int DUMP = W_BASE + 2 * W_BYTES;
  static constexpr int SMEM = DUMP + NWV * 1024;
  static constexpr int NPW = (2 * WP + NWV - 1) / NWV;     // window pieces per wave and k slice (5 at MB = 7 / 8)
  static_assert(NPW <= WTAPS && SMEM <= 160 * 1024, "one window piece per wave and K step; LDS map");
};

// this wave's part of the kernel: row blocks I0 .. I0 + MBW - 1 of column group cg
template <int MB, int EK, bool XT, int MBW, int I0>
__device__ __forceinline__ void win8_body(const radmmm_rowgemm_h3_desc& q, const int a_bytes, const int b_bytes, unsigned char* sm,
                                          const int wave, const int cg, const int rh) {
  using G = WGeo8<MB>;
  constexpr int NT = 2 * MBW, D = LOOKAHEAD, NPW = G::NPW;
  constexpr int NSLOT = NT - D;                                       // pipeline items that may carry DMA pieces
  static_assert(D <= 2 && NSLOT >= 3, "pipeline shape");
  const radmmm_rowgemm_desc& p = q.base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BMR - 1) / G::BMR;
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * G::BMR, n0 = tn * BN;
  const int kpt = p.K / BK;
  const int dil = p.dil, sg = p.sign;
  const int b0 = m0 / p.T, t0 = m0 - b0 * p.T;
  const int nb = (p.T - t0) < G::BMR ? (p.T - t0) : G::BMR;
  const int seg1 = nb + 4 * dil;

  // ---- DMA setup (rowgemm_win.hip; the window's 2 WP pieces are dealt to eight waves, a column group's eight B pieces to
  // its two waves: the hi plane to rh 0, the cross plane to rh 1)
  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int w_vo[NPW], w_dst[NPW], w_isl[NPW], b_voff[4], b_dst[4];
  const int nutt = p.M / p.T;
  const bool masked = p.a_mask_mode && p.lens;
  const int lim0 = masked ? p.lens[b0] : p.T;
  const int lim1 = b0 + 1 < nutt ? (masked ? p.lens[b0 + 1] : p.T) : 0;
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int c = NWV * k + wave;
    w_isl[k] = c >= G::WP ? 1 : 0;
    const int pj = w_isl[k] ? c - G::WP : c;
    w_dst[k] = c < 2 * G::WP ? w_isl[k] * G::W_PLANE + pj * 1024 : -1;
    const int wr = 16 * pj + d_row;
    const bool s1 = wr >= seg1;
    const int b = b0 + (s1 ? 1 : 0);
    const int f = s1 ? wr - seg1 - 2 * dil : t0 - 2 * dil + wr;
    const bool used = c < 2 * G::WP && (s1 ? (nb < G::BMR && wr < G::BMR + 8 * dil) : true);
    const int lim = used ? (s1 ? lim1 : lim0) : 0;
    w_vo[k] = (f >= 0 && f < lim) ? ((b * p.T + f) * q.lda_h + d_chunk * 8) * 2 : OOB;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * cg + k;
    const int lr = 16 * j + d_row;
    const int n = n0 + (lr & ~63) + 2 * (lr & 31) + ((lr >> 5) & 1);
    b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
    b_dst[k] = j * 1024 + rh * G::B_BYTES;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rh ? q.Bl : q.Bh), 0, b_bytes, 0x00020000);
  auto dma_win = [&](int k, int par, int kb) __attribute__((always_inline)) {
    const int dst = w_dst[k] < 0 ? G::DUMP + wave * 1024 : G::W_BASE + par * G::W_BYTES + w_dst[k];
    dma16(w_isl[k] ? rAl : rAh, (lds_u32_ptr)(sm + dst), w_vo[k] + kb * (BK * 2));
  };
  auto dma_b = [&](int k, int buf, int tap, int kb) __attribute__((always_inline)) {
    dma16(rB, (lds_u32_ptr)(sm + buf * G::B_STAGE + b_dst[k]), b_voff[k] + (int)(tap * q.b_tap_stride_h * 2) + kb * (BK * 2));
  };

  f32x16 acc[MBW][2];
#pragma unroll
  for (int i = 0; i < MBW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3, half = lane >> 5;
  const int f_off0 = f_row + (((0 + half) ^ f_swz) << 4);
  const int f_off1 = f_row + (((2 + half) ^ f_swz) << 4);
  int wrow[MBW];
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int ri = 32 * (I0 + i) + (lane & 31);
    wrow[i] = G::W_BASE / ROWB + ri + 2 * dil + (ri >= nb ? 4 * dil : 0);
  }
  auto a_off = [&](int i, int sh) __attribute__((always_inline)) {
    const int w = wrow[i] + sh;
    return (w << 6) + (((half ^ (w >> 2)) & 3) << 4);
  };
  auto shift_of = [&](int tap, int kb) __attribute__((always_inline)) {
    return sg * (tap - WTAPS / 2) * dil + (kb & 1) * (G::W_BYTES / ROWB);
  };
  f16x8 fah[NT], fal[NT], bh[2][2], bl[2][2];
  auto read_a = [&](int t, int sh) __attribute__((always_inline)) {
    const int o = a_off(t >> 1, sh) ^ ((t & 1) << 5);
    fah[t] = *reinterpret_cast<const f16x8*>(sm + o);
    fal[t] = *reinterpret_cast<const f16x8*>(sm + o + G::W_PLANE);
  };
  auto read_b = [&](int bsel, int kb) __attribute__((always_inline)) {
    const unsigned char* sB = sm + bsel * G::B_STAGE + cg * 64 * ROWB;
    const int fo = kb ? f_off1 : f_off0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + fo);
      bl[kb][j] = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + j * 32 * ROWB + fo);
    }
  };
  const int x_sa = (lane >> 5) ? 127 - 11 - q.a8_exp : 127 - q.a8_exp;
  const int x_sb = (lane >> 5) ? 127 - q.b8_exp : 127 - 11 - q.b8_exp;
  auto cross = [&](int i, int j) __attribute__((always_inline)) {
    const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, fal[2 * i]), __builtin_bit_cast(i32x4, fal[2 * i + 1]),
                                             0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, bl[0][j]), __builtin_bit_cast(i32x4, bl[1][j]),
                                             0, 1, 2, 3, 4, 5, 6, 7);
    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, x_sa, 0, x_sb);
  };
  auto mfma_item = [&](int t) __attribute__((always_inline)) {
    const int kb = t & 1, i = t >> 1;
    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][0], acc[i][0], 0, 0, 0);
    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][1], acc[i][1], 0, 0, 0);
    if (kb == 1) cross(i, 0);
    else if (i > 0) cross(i - 1, 1);
  };

  // ---- K loop: one barrier per K step (the B rows are shared by the two waves of a column group); DMA of a step = this
  // wave's 4 pieces of the B tile of step + 1 and 1 piece of the window of k slice kb + 1
#pragma unroll
  for (int k = 0; k < NPW; ++k) dma_win(k, 0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) dma_b(k, 0, 0, 0);
  __syncthreads();
  read_b(0, 0);
  read_b(0, 1);
  {
    const int sh = shift_of(0, 0);
#pragma unroll
    for (int t = 0; t < D; ++t) read_a(t, sh);
  }
  auto kstep = [&](auto tapc, int kb, int bsel) __attribute__((always_inline)) {
    constexpr int tap = decltype(tapc)::value;
    const int sh = shift_of(tap, kb);
    const int ntap = tap == WTAPS - 1 ? 0 : tap + 1, nkb = tap == WTAPS - 1 ? kb + 1 : kb;
#pragma unroll
    for (int t = 0; t < NSLOT; ++t) {
      read_a(t + D, sh);
      mfma_item(t);
      // 5 pieces over NSLOT items (6 or 4): piece u of {B 0..3, window}
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if ((u * NSLOT) / 5 == t) {
          if (u < 4) dma_b(u, bsel ^ 1, ntap, nkb);
          else if (tap < NPW) dma_win(tap, (kb + 1) & 1, kb + 1);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = NSLOT; t < NT; ++t) mfma_item(t);
    cross(MBW - 1, 1);
    read_b(bsel ^ 1, 0);
    read_b(bsel ^ 1, 1);
    {
      const int shn = shift_of(ntap, nkb);
#pragma unroll
      for (int t = 0; t < D; ++t) read_a(t, shn);
    }
  };
  int bsel = 0;
  for (int kb = 0; kb < kpt; ++kb) {
    kstep(std::integral_constant<int, 0>{}, kb, bsel); bsel ^= 1;
    kstep(std::integral_constant<int, 1>{}, kb, bsel); bsel ^= 1;
    kstep(std::integral_constant<int, 2>{}, kb, bsel); bsel ^= 1;
    kstep(std::integral_constant<int, 3>{}, kb, bsel); bsel ^= 1;
    kstep(std::integral_constant<int, 4>{}, kb, bsel); bsel ^= 1;
  }
  __syncthreads();

  // ---- optional extra K segment (rowgemm_win.hip): plain double-buffered loop in the window space
  if constexpr (XT) {
    constexpr int NPX = (4 * MB + NWV - 1) / NWV;                      // A pieces per wave and step
    int x_vo[NPX], x_dst[NPX], x_isl[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int c = NWV * k + wave;
      x_isl[k] = c >= 2 * MB ? 1 : 0;
      const int j = x_isl[k] ? c - 2 * MB : c;
      const int r = m0 + 16 * j + d_row;
      x_vo[k] = (c < 4 * MB && r < p.M) ? ((q.extra_a_rows + r) * q.lda_h + d_chunk * 8) * 2 : OOB;
      x_dst[k] = c < 4 * MB ? x_isl[k] * G::W_PLANE + j * 1024 : -1;
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) wrow[i] = G::W_BASE / ROWB + 32 * (I0 + i) + (lane & 31);
    auto dma_x = [&](int w, int par, int kb) __attribute__((always_inline)) {      // piece w of 0 .. NPX + 3
      if (w < NPX) {
        const int dst = x_dst[w] < 0 ? G::DUMP + wave * 1024 : G::W_BASE + par * G::W_BYTES + x_dst[w];
        dma16(x_isl[w] ? rAl : rAh, (lds_u32_ptr)(sm + dst), x_vo[w] + kb * (BK * 2));
      } else {
        dma_b(w - NPX, par, WTAPS, kb);
      }
    };
#pragma unroll
    for (int w = 0; w < NPX + 4; ++w) dma_x(w, 0, 0);
    __syncthreads();
    read_b(0, 0);
    read_b(0, 1);
#pragma unroll
    for (int t = 0; t < D; ++t) read_a(t, 0);
    for (int kb = 0; kb < kpt; ++kb) {
      const int par = kb & 1, sh = par * (G::W_BYTES / ROWB);
#pragma unroll
      for (int t = 0; t < NSLOT; ++t) {
        read_a(t + D, sh);
        mfma_item(t);
#pragma unroll
        for (int u = 0; u < NPX + 4; ++u)
          if ((u * NSLOT) / (NPX + 4) == t) dma_x(u, par ^ 1, kb + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = NSLOT; t < NT; ++t) mfma_item(t);
      cross(MBW - 1, 1);
      read_b(par ^ 1, 0);
      read_b(par ^ 1, 1);
#pragma unroll
      for (int t = 0; t < D; ++t) read_a(t, sh ^ (G::W_BYTES / ROWB));
    }
    __syncthreads();
  }

  // ---- epilogue: per-row factors of the whole tile, then this wave's row blocks through the direct epilogue
  const radmmm::EpilogueCtx ec(p);
  float sat = 0.f;
  float4* rowf4 = reinterpret_cast<float4*>(sm);
  if (tid < G::BMR) {
    float mk, rt;
    radmmm::epilogue_row_factors(p, ec, m0 + tid, mk, rt);
    const float pre = (p.pconv ? rt : 1.f) * (p.premask ? mk : 1.f);
    const float post = p.postmask ? mk : 1.f;
    const float rsc = p.rowscale == 1 ? mk : (p.rowscale == 2 ? mk * rt : 1.f);
    rowf4[tid] = make_float4(q.acc_scale * pre, post, rsc, 0.f);
  }
  __syncthreads();
  direct_epilogue<MBW, EK, true>(acc, rowf4 + I0 * 32, p, m0 + I0 * 32, n0, lane, cg, sat);
  radmmm::raise_sat_flag(p.sat_flag, sat, (p.Ch && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, p.ch_x8_exp) : 0.f);
}

template <int MB, int EK, bool XT>
__global__ __launch_bounds__(512, 1) void rowgemm_win8_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes, const int b_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wave & 3, rh = wave >> 2;                            // waves w and w + 4 share a SIMD and a column group
  constexpr int M0 = (MB + 1) / 2, M1 = MB / 2;
  if (rh == 0) win8_body<MB, EK, XT, M0, 0>(q, a_bytes, b_bytes, sm, wave, cg, 0);
  else win8_body<MB, EK, XT, M1, M0>(q, a_bytes, b_bytes, sm, wave, cg, 1);
}

template <int MB, int EK, bool XT>
int launch_win8(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = WGeo8<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_win8_kernel<MB, EK, XT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_win8<%d,%d>): %s", MB, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_win8_kernel<MB, EK, XT>), dim3(ntm * ntn), dim3(512), G::SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_win8");
}

template <int MB>
int launch_win8_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  // (the data-gradient kinds are not instantiated: their epilogue wants more than the 256 registers of this variant)
  return ek == EK_SPLIT ? launch_win8<MB, EK_SPLIT, false>(d, stream, a_bytes, b_bytes)
                        : launch_win8<MB, EK_PLAIN, false>(d, stream, a_bytes, b_bytes);
}

}  // namespace

namespace radmmm {
// same launches as rowgemm_win_ok (rowgemm_win.hip) accepts
int launch_rowgemm_win8(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
#ifndef RADMMM_QUICK
  if (mb == 8) return launch_win8_ek<8>(ek, d, stream, a_bytes, b_bytes);
#endif
  return launch_win8_ek<7>(ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm
