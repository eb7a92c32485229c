// rowgemm_one: the wide split conv GEMM (FP8-cross scheme) for ONE-TAP convs -- the WN's res_skip / start / end 1x1 convs and
// their data gradients -- round 4.  Tile machine, operand formats, B-row interleave and direct epilogue are rowgemm_h3d's;
// the K loop is rowgemm_onetap.h's: three A stages + wave-private B with two K steps of flight time, slot-pinned
// instruction order, no address arithmetic in the loop (in-step 1-tap launches of round 3: ~100 us for 26.8 GFLOP, the K
// loop of 32 steps at ~1.7 us each against 1.08 us of MFMA work).
// Scope: nprod 2, taps = 1, no extra K segment, K / 32 even, epilogue kinds PLAIN / SPLIT / RES / DGRAD, MB 4 .. 8;
// everything else keeps rowgemm_h3d (rowgemm_h3w.hip decides; RADMMM_ONE=0 under RADMMM_DEBUG forces it).
#include <type_traits>

#include "rowgemm_onetap.h"

namespace {

template <int MB, int EK, int PR = 2>
__global__ __launch_bounds__(256, 1) void rowgemm_one_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes, const int b_bytes) {
  using G = OneGeo<MB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const radmmm_rowgemm_desc& p = q.base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BMR - 1) / G::BMR;
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;   // XCD x runs a contiguous run of the tile sequence
  // column tiles in PAIRS, row tiles inside a pair (rowgemm_win.hip): an XCD's L2 holds half of the weights and ~ a quarter of A
  const int per_pair = 2 * ntm, pr = tile / per_pair, rr = tile - pr * per_pair;
  const int gw = (ntn - 2 * pr) < 2 ? (ntn - 2 * pr) : 2;
  const int tm = gw == 2 ? (rr >> 1) : rr, tn = 2 * pr + (gw == 2 ? (rr & 1) : 0);
  const int m0 = tm * G::BMR, n0 = tn * BN;
  const int kpt = p.K / BK;

  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int a_vo[MB], a_dst[MB], a_isl[MB], b_voff[4], b_dst[4];
  const bool masked = p.a_mask_mode && p.lens;
#pragma unroll
  for (int k = 0; k < MB; ++k) {
    const int c = 4 * k + wave;                                       // wave-uniform: 16-row group c of 4 MB (hi plane, then cross plane)
    a_isl[k] = c >= 2 * MB ? 1 : 0;
    const int j = a_isl[k] ? c - 2 * MB : c;
    const int r = m0 + 16 * j + d_row;
    bool ok = r < p.M;
    if (ok && masked) {
      const int b = r / p.T;
      ok = r - b * p.T < p.lens[b];
    }
    a_vo[k] = ok ? (r * q.lda_h + d_chunk * 8) * 2 : OOB;
    a_dst[k] = G::A_BASE + a_isl[k] * G::A_PLANE + j * 1024;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * wave + k;                                       // THIS wave's 64 B rows: nobody else reads them
    const int lr = 16 * j + d_row;                                    // LDS row 0 .. 255, interleaved as in rowgemm_h3d
    const int n = n0 + (lr & ~63) + 2 * (lr & 31) + ((lr >> 5) & 1);
    b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
    b_dst[k] = j * 1024;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);
  auto dma_a = [&](int k, int stage, int soff) __attribute__((always_inline)) {
    dma16s(a_isl[k] ? rAl : rAh, (lds_u32_ptr)(sm + stage * G::A_STAGE + a_dst[k]), a_vo[k], soff);
  };
  auto dma_b = [&](int w, int stage, int soff) __attribute__((always_inline)) {
    const int k = w & 3, arr = w >> 2;
    dma16s(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + stage * G::B_STAGE + b_dst[k] + arr * G::B_BYTES), b_voff[k], soff);
  };

  f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int x_sa = (lane >> 5) ? 127 - 11 - q.a8_exp : 127 - q.a8_exp;       // E8M0 block scales (rowgemm_h3d, PR 2)
  const int x_sb = (lane >> 5) ? 127 - q.b8_exp : 127 - 11 - q.b8_exp;
  one_tap_steps<MB, PR>(acc, sm, kpt, lane, wave, x_sa, x_sb, dma_a, dma_b);

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
  direct_epilogue<MB, EK, PR == 2>(acc, rowf4, p, m0, n0, lane, wave, sat);
  {
    const int xe = p.Ch ? (p.C2h && p.c2h_x8_exp > p.ch_x8_exp ? p.c2h_x8_exp : p.ch_x8_exp) : p.c2h_x8_exp;
    radmmm::raise_sat_flag(p.sat_flag, sat, ((p.Ch || p.C2h) && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, xe) : 0.f);
  }
}

template <int MB, int EK, int PR = 2>
int launch_one(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = OneGeo<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_one_kernel<MB, EK, PR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_one<%d,%d>): %s", MB, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_one_kernel<MB, EK, PR>), dim3(ntm * ntn), dim3(256), G::SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_one");
}

template <int MB>
int launch_one_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  switch (ek) {
    case EK_SPLIT: return launch_one<MB, EK_SPLIT>(d, stream, a_bytes, b_bytes);
    case EK_RES: return launch_one<MB, EK_RES>(d, stream, a_bytes, b_bytes);
    case EK_DGRAD: return launch_one<MB, EK_DGRAD>(d, stream, a_bytes, b_bytes);
    default: return launch_one<MB, EK_PLAIN>(d, stream, a_bytes, b_bytes);
  }
}

}  // namespace

namespace radmmm {
bool rowgemm_one_ok(int mb, int ek, const radmmm_rowgemm_h3_desc& d) {
  const radmmm_rowgemm_desc& p = d.base;
#ifdef RADMMM_QUICK
  if (mb != 7) return false;
#endif
  // (three f16 products, round 5: the C-only launches of the tall tiles -- the FiLM stacks' 1x1 convs and their data
  //  gradients at configs[4]'s 32 000 rows, the context LSTM's projection -- take the slot-pinned loop as well)
  const bool scheme_ok = d.nprod == 2 || ((d.nprod == 3 || d.nprod == 0) && (mb == 7 || mb == 8) && ek == EK_PLAIN);
  return scheme_ok && mb >= 4 && mb <= 8 && (ek == EK_PLAIN || ek == EK_SPLIT || ek == EK_RES || ek == EK_DGRAD) && p.taps == 1 &&
         !d.extra_tap && (p.K / BK) % 2 == 0 && p.K >= 2 * BK && (!(p.a_mask_mode && p.lens) || (p.T > 0 && p.M % p.T == 0));
}
int launch_rowgemm_one(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  if (d.nprod != 2) {                                 // three f16 products: EK_PLAIN, MB 7 / 8 (rowgemm_one_ok)
#ifndef RADMMM_QUICK
    if (mb == 8) return launch_one<8, EK_PLAIN, 3>(d, stream, a_bytes, b_bytes);
#endif
    return launch_one<7, EK_PLAIN, 3>(d, stream, a_bytes, b_bytes);
  }
#ifndef RADMMM_QUICK
  switch (mb) {
    case 4: return launch_one_ek<4>(ek, d, stream, a_bytes, b_bytes);
    case 5: return launch_one_ek<5>(ek, d, stream, a_bytes, b_bytes);
    case 6: return launch_one_ek<6>(ek, d, stream, a_bytes, b_bytes);
    case 8: return launch_one_ek<8>(ek, d, stream, a_bytes, b_bytes);
    default: break;
  }
#endif
  return launch_one_ek<7>(ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm
