// rowgemm_win: the wide split conv GEMM (FP8-cross scheme) for 5-TAP convs with a SHARED A WINDOW, round 3.
//
// Why (profiles/r03_dma_path_probe.txt, r03_mfma_dma_mix.txt): the K step of rowgemm_h3d carries 0.94 us of
// MFMA work but takes 1.6 us, because the wave that feeds the matrix pipe also issues the step's 15 LDS-DMA instructions and
// 60 KiB per step and CU in 64-byte row segments (half cache lines) is what the L2 -> L1 path delivers to 232 CUs in 0.9 us.
// With half the DMA the same loop runs at 1.25 us.  Of those 60 KiB, 28 are the A tile -- and with the taps innermost the
// five taps of a k slice fetch THE SAME ROWS five times, shifted by the dilation: tap t wants frames r + (t - 2) d.
// Here the A rows of a k slice come to LDS ONCE, as a window of the tile's 32 MB rows plus a halo of 2 d rows on either
// side, and the tap is a row offset of the fragment reads:
//
//     per k slice (5 K steps):  A window (32 MB + 8 d_max) rows x 64 B x {hi, cross}  = 36 KiB   (was 5 x 28 = 140 KiB)
//     per K step:               B tile 256 rows x 64 B x {hi, cross}                  = 32 KiB   (as before)
//     -> 39 KiB and 10 DMA instructions per wave and K step instead of 60 KiB and 15.
//
// Utterances: a tile's rows may straddle one utterance boundary (T >= 32 MB is required, so at most one).  A tap must not
// read across it -- frames outside [0, len) are zeros (conv padding / partial-conv mask) -- while the same global row can
// be a neighbour's real data.  The window therefore holds TWO segments, one per utterance, each with its own halo:
//     LDS row wr in [0, nb + 4d):            utterance b0,     frame t0 - 2d + wr                  (nb = tile rows in b0)
//     LDS row wr in [nb + 4d, 32 MB + 8d):   utterance b0 + 1, frame -2d + (wr - nb - 4d)
// and tile row i reads LDS row i + 2d + s + (i >= nb ? 4d : 0) for the tap shift s.  Frames outside [0, lim_b) are fetched
// with an out-of-range buffer offset (the DMA writes zeros): the zero padding costs nothing in the K loop.  The fragment
// address of a lane is computed per (row block, tap) -- 5 VALU operations next to 6 MFMAs -- because the XOR swizzle of the
// 16-byte chunks follows the LDS row (rows 4 apart share banks).
//
// Everything else is rowgemm_h3d's: one workgroup per CU, (32 MB) x 256 tile, 4 waves x (MB x 2) accumulators, B rows
// interleaved for the direct epilogue (rowgemm_h3w_kernel.h), one barrier per K step, pinned instruction order.
// Scope: taps = 5, dilation <= 8, FP8-cross scheme, epilogue kinds PLAIN / SPLIT / DGRAD, MB 7 / 8, no extra K segment
// (rowgemm_h3w.hip decides; everything else keeps rowgemm_h3d).
#include "rowgemm_h3w_kernel.h"

namespace {

constexpr int WTAPS = 5, WDMAX = 8;

template <int MB>
struct WGeo {
  static constexpr int BMR = MB * 32;
  static constexpr int WR = BMR + 8 * WDMAX;        // window rows (two segments, four halos)
  static constexpr int WP = WR / 16;                // 16-row DMA pieces per plane
  static constexpr int B_BYTES = BN * ROWB;         // one of {Bh, Bl}
  static constexpr int B_STAGE = 2 * B_BYTES;
  static constexpr int W_BASE = 2 * B_STAGE;        // B stages first: their fragment reads keep 16-bit immediate offsets
  static constexpr int W_PLANE = WR * ROWB;         // one of {Ah, Al}
  static constexpr int W_BYTES = 2 * W_PLANE;
  static constexpr int DUMP = W_BASE + 2 * W_BYTES; // 1 KiB per wave for the surplus DMA slots
  static constexpr int SMEM = DUMP + 4096;
  static constexpr int NPW = (2 * WP + 3) / 4;      // window pieces per wave and k slice
  static constexpr int SPT = (NPW + WTAPS - 1) / WTAPS;   // ... per K step
  static constexpr int NP = 8 + SPT;                // DMA pieces per wave and K step
  static_assert(W_BASE % 1024 == 0 && W_BYTES % 1024 == 0 && SMEM <= 160 * 1024, "LDS map");
};

template <int MB, int T>
__device__ __forceinline__ void pin_items_win() {
  constexpr int NT = 2 * MB, NP = WGeo<MB>::NP;
  if constexpr (T < NT - LOOKAHEAD) {
    __builtin_amdgcn_sched_group_barrier(SGB_DSR, 2, 0);
    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, T == 0 ? 2 : 3, 0);
    if constexpr (T < NP) __builtin_amdgcn_sched_group_barrier(SGB_VMEM, 1, 0);
    pin_items_win<MB, T + 1>();
  }
}

template <int MB, int EK>
__global__ __launch_bounds__(256, 1) void rowgemm_win_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes, const int b_bytes) {
  using G = WGeo<MB>;
  constexpr int NT = 2 * MB, D = LOOKAHEAD, NPW = G::NPW, SPT = G::SPT;
  static_assert(D <= 2, "the look-ahead items belong to row block 0");
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const radmmm_rowgemm_desc& p = q.base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BMR - 1) / G::BMR;
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * G::BMR, n0 = tn * BN;
  const int kpt = p.K / BK;
  const int dil = p.dil, sg = p.sign;
  const int b0 = m0 / p.T, t0 = m0 - b0 * p.T;
  const int nb = (p.T - t0) < G::BMR ? (p.T - t0) : G::BMR;          // tile rows of utterance b0
  const int seg1 = nb + 4 * dil;                                      // first LDS row of the second segment

  // ---- DMA setup.  This lane's row within a 16-row piece and its (source-side swizzled) chunk:
  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int w_vo[NPW], w_dst[NPW], w_isl[NPW], b_voff[4], b_dst[4];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int c = 4 * k + wave;                                       // wave-uniform: piece c of 2 WP (hi plane, then cross plane)
    w_isl[k] = c >= G::WP ? 1 : 0;
    const int pj = w_isl[k] ? c - G::WP : c;
    w_dst[k] = c < 2 * G::WP ? w_isl[k] * G::W_PLANE + pj * 1024 : -1;
    const int wr = 16 * pj + d_row;
    const bool s1 = wr >= seg1;
    const int b = b0 + (s1 ? 1 : 0);
    const int f = s1 ? wr - seg1 - 2 * dil : t0 - 2 * dil + wr;
    const bool used = c < 2 * G::WP && (s1 ? (nb < G::BMR && wr < G::BMR + 8 * dil) : true) && (long long)b * p.T < p.M;
    int lim = 0;
    if (used) lim = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
    w_vo[k] = (f >= 0 && f < lim) ? ((b * p.T + f) * q.lda_h + d_chunk * 8) * 2 : OOB;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * k + wave;
    const int lr = 16 * j + d_row;                                    // LDS row 0 .. 255, interleaved as in rowgemm_h3d
    const int n = n0 + (lr & ~63) + 2 * (lr & 31) + ((lr >> 5) & 1);
    b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
    b_dst[k] = j * 1024;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);
  // window piece k (compile-time) of k slice kb into window `par`
  auto dma_win = [&](int k, int par, int kb) __attribute__((always_inline)) {
    const int dst = w_dst[k] < 0 ? G::DUMP + wave * 1024 : G::W_BASE + par * G::W_BYTES + w_dst[k];
    dma16(w_isl[k] ? rAl : rAh, (lds_u32_ptr)(sm + dst), w_vo[k] + kb * (BK * 2));
  };
  // B piece w (0 .. 7) of tile (tap, kb) into B stage `buf`
  auto dma_b = [&](int w, int buf, int tap, int kb) __attribute__((always_inline)) {
    const int k = w & 3, arr = w >> 2;
    const int vo = b_voff[k] + (int)(tap * q.b_tap_stride_h * 2) + kb * (BK * 2);
    dma16(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + buf * G::B_STAGE + b_dst[k] + arr * G::B_BYTES), vo);
  };

  f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- fragment addressing
  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3, half = lane >> 5;
  const int f_off0 = f_row + (((0 + half) ^ f_swz) << 4);            // B tiles: fixed rows, as in rowgemm_h3d
  const int f_off1 = f_row + (((2 + half) ^ f_swz) << 4);
  int wrow[MB];                                                       // LDS row (in 64-byte units from sm) of tile row 32 i + (lane & 31) at shift 0, window 0
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int ri = 32 * i + (lane & 31);
    wrow[i] = G::W_BASE / ROWB + ri + 2 * dil + (ri >= nb ? 4 * dil : 0);
  }
  // byte offset of (row block i, k block 0) for the uniform row shift sh (tap shift + window parity); k block 1 is ^ 32
  auto a_off = [&](int i, int sh) __attribute__((always_inline)) {
    const int w = wrow[i] + sh;
    return (w << 6) + (((half ^ (w >> 2)) & 3) << 4);
  };
  auto shift_of = [&](int tap, int kb) __attribute__((always_inline)) {
    return sg * (tap - WTAPS / 2) * dil + (kb & 1) * (G::W_BYTES / ROWB);
  };

  f16x8 fah[NT], fal[NT], bh[2][2], bl[2][2];
  auto read_a = [&](int t, int sh) __attribute__((always_inline)) {   // item t = 2 i + kblock
    const int o = a_off(t >> 1, sh) ^ ((t & 1) << 5);
    fah[t] = *reinterpret_cast<const f16x8*>(sm + o);
    fal[t] = *reinterpret_cast<const f16x8*>(sm + o + G::W_PLANE);
  };
  auto read_b = [&](int bsel, int kb) __attribute__((always_inline)) {
    const unsigned char* sB = sm + bsel * G::B_STAGE + wave * 64 * ROWB;
    const int fo = kb ? f_off1 : f_off0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + fo);
      bl[kb][j] = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + j * 32 * ROWB + fo);
    }
  };
  const int x_sa = (lane >> 5) ? 127 - 11 - q.a8_exp : 127 - q.a8_exp;       // E8M0 block scales (rowgemm_h3d, PR 2)
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

  // ---- prologue: window of k slice 0, B tile of step 0
#pragma unroll
  for (int k = 0; k < NPW; ++k) dma_win(k, 0, 0);
#pragma unroll
  for (int w = 0; w < 8; ++w) dma_b(w, 0, 0, 0);
  __syncthreads();
  read_b(0, 0);
  read_b(0, 1);
  {
    const int sh = shift_of(0, 0);
#pragma unroll
    for (int t = 0; t < D; ++t) read_a(t, sh);
  }
  int bsel = 0;                                                       // B stage of the current step
  for (int kb = 0; kb < kpt; ++kb) {
#pragma unroll
    for (int tap = 0; tap < WTAPS; ++tap) {
      const int sh = shift_of(tap, kb);
      const int ntap = tap == WTAPS - 1 ? 0 : tap + 1, nkb = tap == WTAPS - 1 ? kb + 1 : kb;      // tile of step + 1
      // items 0 .. NT-D-1: fragments of item t + D | MFMAs of item t | one DMA piece: the B tile of step + 1, then this
      // step's share of the window of k slice kb + 1
#pragma unroll
      for (int t = 0; t < NT - D; ++t) {
        read_a(t + D, sh);
        mfma_item(t);
        if (t < 8) dma_b(t, bsel ^ 1, ntap, nkb);
        else if (t < 8 + SPT) {
          const int k = SPT * tap + (t - 8);                          // (compile-time)
          if (k < NPW) dma_win(k, (kb + 1) & 1, kb + 1);
          else dma16(rAh, (lds_u32_ptr)(sm + G::DUMP + wave * 1024), OOB);   // keep the instruction count of a step fixed
        }
      }
      pin_items_win<MB, 0>();
      // every read of this step's stage / window rows has been issued: retire them and this wave's DMA, meet the other
      // waves, then the last D items' MFMAs and the first fragments of the next step
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = NT - D; t < NT; ++t) mfma_item(t);
      cross(MB - 1, 1);
      bsel ^= 1;
      read_b(bsel, 0);
      read_b(bsel, 1);
      {
        const int shn = shift_of(ntap, nkb);
#pragma unroll
        for (int t = 0; t < D; ++t) read_a(t, shn);
      }
    }
  }
  __syncthreads();                                                    // stray fragment reads / DMA past the last tile

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
  direct_epilogue<MB, EK, true>(acc, rowf4, p, m0, n0, lane, wave, sat);
  radmmm::raise_sat_flag(p.sat_flag, sat, (p.Ch && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, p.ch_x8_exp) : 0.f);
}

template <int MB, int EK>
int launch_win(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = WGeo<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_win_kernel<MB, EK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_win<%d,%d>): %s", MB, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_win_kernel<MB, EK>), dim3(ntm * ntn), dim3(256), G::SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_win");
}

template <int MB>
int launch_win_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  switch (ek) {
    case EK_SPLIT: return launch_win<MB, EK_SPLIT>(d, stream, a_bytes, b_bytes);
    case EK_DGRAD: return launch_win<MB, EK_DGRAD>(d, stream, a_bytes, b_bytes);
    default: return launch_win<MB, EK_PLAIN>(d, stream, a_bytes, b_bytes);
  }
}

}  // namespace

namespace radmmm {
// can this launch take the shared-window kernel?  (mb, ek as chosen by rowgemm_h3w.hip)
bool rowgemm_win_ok(int mb, int ek, const radmmm_rowgemm_h3_desc& d) {
  const radmmm_rowgemm_desc& p = d.base;
#ifdef RADMMM_QUICK
  if (mb != 7) return false;
#endif
  return d.nprod == 2 && (mb == 7 || mb == 8) && (ek == EK_PLAIN || ek == EK_SPLIT || ek == EK_DGRAD) && p.taps == WTAPS &&
         !d.extra_tap && p.dil >= 1 && p.dil <= WDMAX && p.T >= 32 * mb && p.M % p.T == 0 && (p.sign == 1 || p.sign == -1);
}
int launch_rowgemm_win(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
#ifndef RADMMM_QUICK
  if (mb == 8) return launch_win_ek<8>(ek, d, stream, a_bytes, b_bytes);
#endif
  return launch_win_ek<7>(ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm
