// rowgemm_rs: ROLE-SPLIT version of the wide split conv GEMM (FP8-cross scheme), round 3.
//
// Why (tools/mfma_dma_mix.hip): a wave that issues an LDS-DMA instruction between its MFMAs stalls its own
// MFMA stream for ~78 cycles; the K step of rowgemm_h3d -- 0.94 us of MFMA work and 0.89 us of operand DMA per CU -- takes
// 1.6 us because the same four waves issue both, and 1.07 us in the probe when SEPARATE waves issue them.  Every wave of a
// kernel gets the same register allocation, so the split needs three waves per SIMD at <= 168 registers:
//
//   waves 0..7   CONSUMERS: wave c owns all MB row blocks of ONE 32-column block (column group c & 3, block c >> 2):
//                MB x 16 accumulators (112 at MB = 7); per K step 4 B fragments + 4 A fragments per row block from LDS,
//                2 f16 MFMAs + 1 scaled FP8 MFMA per row block; never a vector-memory instruction inside the K loop;
//   waves 8..11  PRODUCERS: the DMA code of rowgemm_h3d, unchanged in what it fetches (15 pieces per wave and step at
//                MB = 7: tap shifts, utterance masks, zero fill through out-of-range buffer offsets), nothing else.
//   Wave w and w + 4 share a SIMD (round-robin placement): each SIMD hosts two consumers and one producer.
//
// One barrier per K step for all twelve waves (two LDS stages as before).  The consumers' 32 columns are CONTIGUOUS (the B
// rows are not interleaved here), so the direct epilogue stores one column per lane: 128 / 64 / 32 contiguous bytes per
// row and half wave for the fp32 / fp16 / 8-bit outputs.  Epilogue kinds: EK_PLAIN and EK_SPLIT (the forward convs and the
// plain data gradients); launches with side inputs keep rowgemm_h3d.
#include "rowgemm_h3w_kernel.h"

namespace {

template <int MB, int I>
__device__ __forceinline__ void take_block1(const f32x16 (&acc)[MB], int sel, float (&v)[16]) {
  if constexpr (I < MB) {
    int s2 = sel;
    asm volatile("" : "+s"(s2));
    if (s2 == I) {
      asm volatile("; accumulators of row block %0" : : "n"(I));
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[I][e];
    }
    take_block1<MB, I + 1>(acc, sel, v);
  }
}

// one column per lane: split copy of y (hi fp16, 8-bit cross array in format fmt [, fp16 lo])
__device__ __forceinline__ float store_one_split(__amdgpu_buffer_rsrc_t rH, __amdgpu_buffer_rsrc_t rL, __amdgpu_buffer_rsrc_t rLo16,
                                                 bool has_lo16, int vH, int vXh, int vXl, int sH, float x8_mul, float s, float y) {
  const float u = y * s;
  const float t = radmmm::clamp_f16(u);
  const _Float16 h = (_Float16)t;
  const float r = t - (float)h;
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rH, vH, sH, 0);
  const int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(radmmm::clamp_e4m3(t * x8_mul), radmmm::clamp_e4m3(r * x8_mul * 2048.f), 0, false);
  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w8 & 0xff), rL, vXh, sH, 0);
  __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((w8 >> 8) & 0xff), rL, vXl, sH, 0);
  if (has_lo16) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)r), rLo16, vH, sH, 0);
  return fabsf(u);
}

template <int MB, int EK, int ACTK>
__device__ __forceinline__ void rs_epilogue(const f32x16 (&acc)[MB], const float4* rowf, const radmmm_rowgemm_desc& p, int m0, int col,
                                            int lane, float& sat) {
  constexpr bool SPLIT = EK == EK_SPLIT;
  const int h = lane >> 5;
  const bool cok = col < p.N;
  const float b0 = (p.bias && cok) ? p.bias[col] : 0.f;
  const int act = p.act;
  const long long M = p.M;
  const __amdgpu_buffer_rsrc_t rC = rsrc_of(p.C, M * p.ldc * 4);
  const int vC = cok ? (4 * h * p.ldc + col) * 4 : OOB;
  const int fmt = p.split_fmt;
  const bool has_lo16 = SPLIT && p.Clo != nullptr;
  const __amdgpu_buffer_rsrc_t rH = rsrc_of(SPLIT ? p.Ch : nullptr, M * p.ldch * 2);
  const __amdgpu_buffer_rsrc_t rL = rsrc_of(SPLIT ? p.Cl : nullptr, M * p.ldch * 2);
  const __amdgpu_buffer_rsrc_t rLo16 = rsrc_of(has_lo16 ? p.Clo : nullptr, M * p.ldch * 2);
  const int vH = cok ? (4 * h * p.ldch + col) * 2 : OOB;
  const int vXh = cok ? (int)(4 * h * p.ldch * 2 + radmmm::x8_hi_off(col, fmt)) : OOB;
  const int vXl = cok ? (int)(4 * h * p.ldch * 2 + radmmm::x8_lo_off(col, fmt)) : OOB;
  const float x8_mul = __builtin_ldexpf(1.f, p.ch_x8_exp);
  auto row_of = [](int e) { return 8 * (e >> 2) + (e & 3); };
  const int left = (p.M - m0 + 31) / 32;
  const int nblk = left < MB ? left : MB;
#pragma unroll 1
  for (int I = 0; I < nblk; ++I) {
    const int r0 = m0 + I * 32;
    float v[16];
    take_block1<MB, 0>(acc, I, v);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ru = r0 + row_of(e);
      const float4 rf = rowf[I * 32 + row_of(e) + 4 * h];
      float x = (v[e] * rf.x + b0) * rf.y * rf.z;
      if constexpr (ACTK == 1) x = softplus_nb(x);
      else if constexpr (ACTK == 2) x = radmmm::act_apply(x, act);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rC, vC, ru * p.ldc * 4, 0);
      if constexpr (SPLIT)
        sat = fmaxf(sat, store_one_split(rH, rL, rLo16, has_lo16, vH, vXh, vXl, ru * p.ldch * 2, x8_mul, p.ch_scale, x));
    }
  }
}

template <int MB, int EK>
__global__ __launch_bounds__(768, 1) void rowgemm_rs_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes, const int b_bytes) {
  using G = Geo<MB>;
  constexpr int NPA = MB, NP = MB + 8;                  // DMA pieces per producer wave and step (as rowgemm_h3d, PR 2)
  constexpr int NG = 2 * MB;
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
  const int ntaps = p.taps + (q.extra_tap ? 1 : 0);
  const int nsteps = kpt * ntaps;
  // per-row factors of the epilogue, behind the two stages (+ the producers' dump area): written before the first barrier
  float4* rowf4 = reinterpret_cast<float4*>(sm + 2 * G::STAGE + 4096);
  {
    const radmmm::EpilogueCtx ec(p);
    if (tid < G::BMR) {
      float mk, rt;
      radmmm::epilogue_row_factors(p, ec, m0 + tid, mk, rt);
      const float pre = (p.pconv ? rt : 1.f) * (p.premask ? mk : 1.f);
      const float post = p.postmask ? mk : 1.f;
      const float rsc = p.rowscale == 1 ? mk : (p.rowscale == 2 ? mk * rt : 1.f);
      rowf4[tid] = make_float4(q.acc_scale * pre, post, rsc, 0.f);
    }
  }

  if (wave >= 8) {
    // ------------------------------------------------------------------ producer
    const int pw = wave - 8;
    const int extra_bytes = q.extra_a_rows * q.lda_h * 2;
    const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    int a_t[NPA], a_lim[NPA], a_base[NPA], a_vo[NPA], a_dst[NPA], a_isl[NPA], b_voff[4], b_dst[4];
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int c = 4 * k + pw;
      a_isl[k] = c >= NG ? 1 : 0;
      const int j = a_isl[k] ? c - NG : c;
      const int r = m0 + 16 * j + d_row;
      a_t[k] = 0;
      a_lim[k] = -1;
      a_base[k] = 0;
      if (r < p.M) {
        const int b = r / p.T;
        a_t[k] = r - b * p.T;
        a_lim[k] = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
        a_base[k] = (b * p.T * q.lda_h + d_chunk * 8) * 2;
      }
      a_dst[k] = a_isl[k] * G::A_BYTES + j * 1024;
      a_vo[k] = OOB;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = 4 * k + pw;
      const int n = n0 + 16 * j + d_row;                             // B rows in natural order: a consumer's 32 columns are contiguous
      b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
      b_dst[k] = 2 * G::A_BYTES + j * 1024;
    }
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
      const bool ex = tap >= p.taps;
      const int s = ex ? 0 : p.sign * (tap - p.taps / 2) * p.dil;
      const int xb = ex ? extra_bytes : 0;
#pragma unroll
      for (int k = 0; k < NPA; ++k) {
        const int ts = a_t[k] + s;
        const int ok = -(int)((ts >= 0) & (ts < a_lim[k]));
        a_vo[k] = ((a_base[k] + ts * q.lda_h * 2 + xb) & ok) | (OOB & ~ok);
      }
    };
    auto dma_tile = [&](int buf, int tap, int kb) __attribute__((always_inline)) {
      const int sbase = buf * G::STAGE;
#pragma unroll
      for (int w = 0; w < NP; ++w) {
        if (w < NPA) {
          dma16(a_isl[w] ? rAl : rAh, (lds_u32_ptr)(sm + sbase + a_dst[w]), a_vo[w] + kb * (BK * 2));
        } else {
          const int k = (w - NPA) & 3, arr = (w - NPA) >> 2;
          const int vo = b_voff[k] + (int)(tap * q.b_tap_stride_h * 2) + kb * (BK * 2);
          dma16(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + sbase + b_dst[k] + arr * G::B_BYTES), vo);
        }
      }
    };
    int l_tap = 0, l_kb = 0;
    set_tap(0);
    dma_tile(0, 0, 0);
    __syncthreads();                                               // tile 0 (and the row factors) are in LDS
    for (int step = 0; step < nsteps; ++step) {
      if (step + 1 < nsteps) {                                      // taps innermost (rowgemm_h3d)
        const bool wrap = l_tap == ntaps - 1;
        l_tap = wrap ? 0 : l_tap + 1;
        l_kb = wrap ? l_kb + 1 : l_kb;
        set_tap(l_tap);
        dma_tile((step + 1) & 1, l_tap, l_kb);
      }
      __syncthreads();                                             // (vmcnt(0) + barrier) tile step + 1 landed; stage `step & 1` is free
    }
    return;
  }

  // -------------------------------------------------------------------- consumer
  const int cg = wave & 3, jb = wave >> 2;
  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3;
  const int f_off0 = f_row + (((0 + (lane >> 5)) ^ f_swz) << 4);
  const int f_off1 = f_row + (((2 + (lane >> 5)) ^ f_swz) << 4);
  const int x_sa = (lane >> 5) ? 127 - 11 - q.a8_exp : 127 - q.a8_exp;
  const int x_sb = (lane >> 5) ? 127 - q.b8_exp : 127 - 11 - q.b8_exp;
  const int b_rows = (cg * 64 + jb * 32) * ROWB;
  __syncthreads();                                                 // tile 0
  for (int step = 0; step < nsteps; ++step) {
    const unsigned char* st = sm + (step & 1) * G::STAGE;
    const unsigned char* sB = st + 2 * G::A_BYTES + b_rows;
    const f16x8 bh0 = *reinterpret_cast<const f16x8*>(sB + f_off0), bh1 = *reinterpret_cast<const f16x8*>(sB + f_off1);
    const f16x8 bl0 = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + f_off0), bl1 = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + f_off1);
    const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, bl0), __builtin_bit_cast(i32x4, bl1), 0, 1, 2, 3, 4, 5, 6, 7);
    f16x8 ah0[2], ah1[2], al0[2], al1[2];
    auto read_a = [&](int slot, int i) __attribute__((always_inline)) {
      const unsigned char* sA = st + i * 32 * ROWB;
      ah0[slot] = *reinterpret_cast<const f16x8*>(sA + f_off0);
      ah1[slot] = *reinterpret_cast<const f16x8*>(sA + f_off1);
      al0[slot] = *reinterpret_cast<const f16x8*>(sA + G::A_BYTES + f_off0);
      al1[slot] = *reinterpret_cast<const f16x8*>(sA + G::A_BYTES + f_off1);
    };
    read_a(0, 0);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int sl = i & 1;
      if (i + 1 < MB) read_a(sl ^ 1, i + 1);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0[sl], bh0, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[sl], bh1, acc[i], 0, 0, 0);
      const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, al0[sl]), __builtin_bit_cast(i32x4, al1[sl]), 0, 1, 2, 3, 4, 5, 6, 7);
      acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i], 0, 0, 0, x_sa, 0, x_sb);
    }
    __syncthreads();
  }
  float sat = 0.f;
  const int col = n0 + cg * 64 + jb * 32 + (lane & 31);
  if (p.act == RADMMM_ACT_SOFTPLUS) rs_epilogue<MB, EK, 1>(acc, rowf4, p, m0, col, lane, sat);
  else if (p.act == RADMMM_ACT_NONE) rs_epilogue<MB, EK, 0>(acc, rowf4, p, m0, col, lane, sat);
  else rs_epilogue<MB, EK, 2>(acc, rowf4, p, m0, col, lane, sat);
  radmmm::raise_sat_flag(p.sat_flag, sat, (EK == EK_SPLIT && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, p.ch_x8_exp) : 0.f);
}

template <int MB, int EK>
int launch_rs(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = Geo<MB>;
  constexpr int SMEM = 2 * G::STAGE + 4096 + G::BMR * 16;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_rs_kernel<MB, EK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_rs<%d,%d>): %s", MB, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_rs_kernel<MB, EK>), dim3(ntm * ntn), dim3(768), SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_rs");
}

template <int MB>
int launch_rs_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  return ek == EK_SPLIT ? launch_rs<MB, EK_SPLIT>(d, stream, a_bytes, b_bytes) : launch_rs<MB, EK_PLAIN>(d, stream, a_bytes, b_bytes);
}

}  // namespace

namespace radmmm {
// FP8-cross scheme, epilogue kinds EK_PLAIN (1) / EK_SPLIT (2) only (rowgemm_h3w.hip decides)
int launch_rowgemm_rs(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  switch (mb) {
    case 4: return launch_rs_ek<4>(ek, d, stream, a_bytes, b_bytes);
    case 5: return launch_rs_ek<5>(ek, d, stream, a_bytes, b_bytes);
    case 6: return launch_rs_ek<6>(ek, d, stream, a_bytes, b_bytes);
    case 7: return launch_rs_ek<7>(ek, d, stream, a_bytes, b_bytes);
    default: return launch_rs_ek<8>(ek, d, stream, a_bytes, b_bytes);
  }
}
}  // namespace radmmm
