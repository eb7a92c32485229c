// rowgemm_h3: the channels-last Conv1d family (radmmm_rowgemm_f32's contract: taps as row shifts,
// length masking, fused epilogue) on the f16 matrix cores with fp32-class accuracy by operand
// splitting:  x*s = hi + lo (fp16),  A.B ~= Ah.Bh + Ah.Bl + Al.Bh,  fp32 accumulate.
//
// Measured on MI355X (tests/test_hip_h3probe.py, bench.py --kernel-only): max rel. error 2.2e-6 at
// K = 5120 (the fp32 MFMA kernel: 3.8e-6) at 2.3x the fp32-MFMA kernel's rate.  Three
// v_mfma_f32_32x32x16_f16 per 16-deep k block do the work of eight v_mfma_f32_32x32x2_f32, and
// the f16 MFMA pipe -- unlike the fp32 one -- does not compete with VALU instructions.
//
// Geometry: 128x128 tile, K step 32, 256 threads = 4 waves (2x2), each wave 2x2 MFMA tiles.
// Four operand tiles {Ah, Al, Bh, Bl} of [128 rows][32 halves] per step, LDS row pitch 80 B
// (r -> 5r mod 16 is a bijection: the 16 rows of a ds_read_b128 lane group start in 16 distinct
// 4-bank slots), double buffered = 80 KiB -> 2 workgroups per CU.  Operands are staged with
// buffer loads: per-thread byte offsets change only with the tap, the K position is a scalar
// offset, masked / out-of-item frames get an out-of-range offset (zeros from the buffer unit).
// Both operands are K-contiguous; the data-gradient uses a transposed split copy of the weights.
#include <stdlib.h>

#include "common.h"
#include "rowgemm_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 40;                       // halves per LDS row (80 B)
constexpr int TILE_H = BM * PITCH;              // halves per operand tile
constexpr int SMEM_BYTES = 2 * 4 * TILE_H * 2;  // 80 KiB (>= 64 KiB needed by the epilogue stage)
constexpr int OOB = 0x7fffffff;

__global__ __launch_bounds__(256, 2) void rowgemm_h3_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes,
                                                             const int b_bytes) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smh[];
  const radmmm_rowgemm_desc& p = q.base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kpt = p.K / BK;
  const int nsteps = kpt * p.taps;

  // staging: 128 rows x 4 chunks of 16 B per operand tile -> rows s_row, s_row + 64, chunk s_chunk
  const int s_row = tid >> 2, s_chunk = tid & 3;
  int a_t[2], a_lim[2], a_base[2], a_voff[2], b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + s_row + 64 * i;
    a_t[i] = 0;
    a_lim[i] = 0;
    a_base[i] = 0;
    a_voff[i] = OOB;
    if (r < p.M) {
      const int b = r / p.T;
      a_t[i] = r - b * p.T;
      a_lim[i] = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
      a_base[i] = (b * p.T * q.lda_h + s_chunk * 8) * 2;
    }
    const int n = n0 + s_row + 64 * i;
    b_voff[i] = n < p.N ? (n * q.ldb_h + s_chunk * 8) * 2 : OOB;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);

  struct Regs {
    u32x4 v[4][2];   // {Ah, Al, Bh, Bl} x 2 rows
  };
  auto load_tiles = [&](int step, Regs& R) __attribute__((always_inline)) {
    const int tap = step / kpt, kb = step - tap * kpt;
    if (kb == 0) {                                     // uniform branch, once per tap
      const int s = p.sign * (tap - p.taps / 2) * p.dil;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ts = a_t[i] + s;
        a_voff[i] = (ts >= 0 && ts < a_lim[i]) ? a_base[i] + ts * q.lda_h * 2 : OOB;
      }
    }
    const int so_a = kb * (BK * 2);
    const int so_b = (int)(tap * q.b_tap_stride_h * 2) + kb * (BK * 2);
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

  // fragment: lane l holds row (l & 31), k = 8*(l >> 5) .. +7 of each 16-deep block
  const int f_off = (lane & 31) * PITCH + (lane >> 5) * 8;
  Regs R;
  load_tiles(0, R);
  store_tiles(0, R);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    const int nxt = step + 1 < nsteps ? step + 1 : step;     // straight-line loop
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

  // ---- epilogue through LDS (row-major [128][128] fp32 = 64 KiB of the 80 KiB) -----------------
  float* smf = reinterpret_cast<float*>(smh);
  {
    float* base = smf + (wm * 64 + 4 * (lane >> 5)) * BN + wn * 64 + (lane & 31);
    const float sc = q.acc_scale;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          base[(mi * 32 + (e & 3) + 8 * (e >> 2)) * BN + ni * 32] = acc[mi][ni][e] * sc;
  }
  __syncthreads();
  const radmmm::EpilogueCtx ec(p);
  const int c4 = (tid & 31) * 4;
  float sat = 0.f;
  for (int i = 0; i < 16; ++i) {
    const int rl = i * 8 + (tid >> 5);
    const float4 a4 = *reinterpret_cast<const float4*>(smf + rl * BN + c4);
    sat = fmaxf(sat, radmmm::epilogue_store4(p, ec, m0 + rl, n0 + c4, a4));
  }
  radmmm::raise_sat_flag(p.sat_flag, sat, (p.Ch && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, p.ch_x8_exp) : 0.f);
}

}  // namespace

// Which kernel a descriptor takes -- ONE decision for the launcher and for radmmm_rowgemm_h3_colsum_rows (the two had
// drifted apart under the RADMMM_H3_1X1 experiment switch).  Default: the wide-tile kernel (rowgemm_h3w.hip), one workgroup per
// CU.  A small batch does not give it enough tiles (M = 3200: 100 workgroups of its smallest 128 x 256 tile on 256 CUs): below
// half a round this file's 128 x 128 kernel, two workgroups per CU, is faster (B = 8, T = 800: 40.0 vs 43.9 ms per step).
// RADMMM_H3_TILE=128 / 256 forces one or the other (A/B runs; RADMMM_DEBUG=1 only, read per launch then: tests switch it).
static bool takes_narrow_kernel(const radmmm_rowgemm_h3_desc& d) {
  const radmmm_rowgemm_desc& p = d.base;
  const char* forced_env = radmmm::debug_env("RADMMM_H3_TILE");
  const int forced = forced_env ? atoi(forced_env) : 0;
  static const bool narrow_1x1 = [] {                 // experiment: short-K launches on the 2-workgroup-per-CU kernel
    const char* e = radmmm::debug_env("RADMMM_H3_1X1");
    return e && atoi(e) == 128;
  }();
  const long long wide_wgs = (long long)((p.M + 127) / 128) * ((p.N + 255) / 256);
  // (the FP8 cross-term scheme exists on the wide kernel only)
  if (d.nprod == 2 || d.extra_tap) return false;
  return forced == 128 || (forced != 256 && wide_wgs < 128) || (narrow_1x1 && p.taps == 1);
}

extern "C" int radmmm_rowgemm_h3(const radmmm_rowgemm_h3_desc* d, radmmm_stream_t stream) {
  RADMMM_REQUIRE(d != nullptr, "rowgemm_h3: null descriptor");
  const radmmm_rowgemm_desc& p = d->base;
  RADMMM_REQUIRE(d->Ah && d->Al && d->Bh && d->Bl, "rowgemm_h3: null operand");
  RADMMM_REQUIRE(p.C || p.Ch, "rowgemm_h3: C may be NULL only when the split copy Ch / Cl carries the result");
  RADMMM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.taps >= 1 && p.T > 0 && p.M % p.T == 0, "rowgemm_h3: bad dims");
  RADMMM_REQUIRE(p.K % BK == 0, "rowgemm_h3: K=%d must be a multiple of 32", p.K);
  RADMMM_REQUIRE(p.b_layout == 0, "rowgemm_h3: both operands are K-contiguous (use a transposed weight copy)");
  RADMMM_REQUIRE(p.a_item_stride == 0, "rowgemm_h3: a_item_stride is not supported");
  RADMMM_REQUIRE(d->lda_h % 8 == 0 && d->ldb_h % 8 == 0 && d->lda_h >= p.K && d->ldb_h >= p.K && d->b_tap_stride_h % 8 == 0,
                 "rowgemm_h3: split operands need ld %% 8 == 0 and ld >= K");
  RADMMM_REQUIRE(radmmm::aligned16(d->Ah) && radmmm::aligned16(d->Al) && radmmm::aligned16(d->Bh) && radmmm::aligned16(d->Bl),
                 "rowgemm_h3: split operands must be 16B aligned");
  RADMMM_REQUIRE(p.sign == 1 || p.sign == -1, "rowgemm_h3: sign must be +-1");
  RADMMM_REQUIRE(!(p.pconv || p.rowscale == 2) || (p.ratio_taps >= 1 && p.ratio_dil >= 1), "rowgemm_h3: ratio_taps/ratio_dil");
  RADMMM_REQUIRE(!p.dact || (p.dact_src != nullptr) != (p.dact_h != nullptr), "rowgemm_h3: dact needs dact_src or the pair dact_h / dact_x");
  RADMMM_REQUIRE(!p.dact_h || (p.dact_x && p.lddact_h % 32 == 0 && p.lddact_h >= p.N && abs(p.dact_x8_exp) <= 16 &&
                               (reinterpret_cast<uintptr_t>(p.dact_h) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.dact_x) & 3) == 0),
                 "rowgemm_h3: dact_h / dact_x: an 8-bit split pair with ld %% 32 == 0");
  RADMMM_REQUIRE(!p.Ch || (p.Cl && p.ldch % 4 == 0 && p.ldch >= ((p.N + 3) & ~3)), "rowgemm_h3: Ch/Cl");
  RADMMM_REQUIRE(p.n_c2_src >= 0 && p.n_c2_src <= 3, "rowgemm_h3: n_c2_src must be 0 .. 3");
  for (int k = 0; k < p.n_c2_src; ++k)
    RADMMM_REQUIRE(p.c2_src[k] && radmmm::aligned16(p.c2_src[k]), "rowgemm_h3: c2_src[%d] null / not 16-byte aligned", k);
  RADMMM_REQUIRE(p.n_c2_src == 0 || ((p.C2 || p.C2h) && p.ldc2 % 4 == 0 && p.ldc2 >= p.N),
                 "rowgemm_h3: c2_src needs C2 or C2h to receive the sum, and the sources' pitch in ldc2");
  RADMMM_REQUIRE(!p.C2h || (p.C2l && (p.C2 || p.n_c2_src > 0) && p.ldc2h % 4 == 0 && p.ldc2h >= ((p.N + 3) & ~3)), "rowgemm_h3: C2h/C2l");
  RADMMM_REQUIRE(p.split_fmt >= RADMMM_SPLIT_F16 && p.split_fmt <= RADMMM_SPLIT_X8B, "rowgemm_h3: split_fmt");
  RADMMM_REQUIRE(p.split_fmt == RADMMM_SPLIT_F16 || ((!p.Ch || p.ldch % 32 == 0) && (!p.C2h || p.ldc2h % 32 == 0) &&
                                                      abs(p.ch_x8_exp) <= 16 && abs(p.c2h_x8_exp) <= 16),
                 "rowgemm_h3: 8-bit split outputs need ld %% 32 == 0 and |x8_exp| <= 16");
  RADMMM_REQUIRE(d->nprod >= 0 && d->nprod <= 3, "rowgemm_h3: nprod");
  RADMMM_REQUIRE(!p.colsum_out || (p.colsum_scratch && !p.add && !p.C2 && !p.n_c2_src), "rowgemm_h3: colsum_out needs colsum_scratch (and no add / C2 input)");
  RADMMM_REQUIRE(d->nprod != 2 || (abs(d->a8_exp) <= 16 && abs(d->b8_exp) <= 16 && d->lda_h % 32 == 0 && d->ldb_h % 32 == 0 &&
                                   d->b_tap_stride_h % 32 == 0),
                 "rowgemm_h3: nprod 2 needs ld %% 32 == 0 and |x8_exp| <= 16");
  RADMMM_REQUIRE(!d->extra_tap || (d->extra_a_rows >= p.M && p.taps >= 1 && !p.a_mask_mode),
                 "rowgemm_h3: the extra K segment needs extra_a_rows >= M (second matrix behind the first) and a_mask_mode 0");
  const int ntaps = p.taps + (d->extra_tap ? 1 : 0);
  const long long a_bytes = ((long long)p.M + (d->extra_tap ? d->extra_a_rows : 0)) * d->lda_h * 2;
  const long long b_bytes = ((long long)(ntaps - 1) * d->b_tap_stride_h + (long long)p.N * d->ldb_h) * 2;
  RADMMM_REQUIRE(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "rowgemm_h3: operand >= 2 GiB");
  if (!takes_narrow_kernel(*d))
    return radmmm::launch_rowgemm_h3w(*d, static_cast<hipStream_t>(stream), (int)a_bytes, (int)b_bytes);
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_h3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL(rowgemm_h3_kernel, dim3(ntm * ntn), dim3(256), SMEM_BYTES, static_cast<hipStream_t>(stream), *d,
                     (int)a_bytes, (int)b_bytes);
  const int rc = radmmm::check_launch("rowgemm_h3");
  if (rc || !p.colsum_out) return rc;
  RADMMM_REQUIRE(p.C, "rowgemm_h3: colsum_out on this kernel sums C afterwards: C must not be NULL");
  return radmmm_colsum(p.C, p.ldc, p.colsum_out, p.colsum_scratch, p.M, p.N, p.rowscale == 2 ? 2 : (p.rowscale == 1 ? 1 : 0), p.T,
                       p.lens, p.ratio_taps, p.ratio_dil, 0, stream);
}

namespace radmmm { int h3w_colsum_rows(const radmmm_rowgemm_h3_desc& d); }
// rows of partial column sums a launch of *d leaves in colsum_scratch when colsum_out is NULL (include/radmmm_hip.h); 0 = this
// descriptor takes a kernel that leaves none (generic epilogue, the narrow kernel): give it colsum_out
extern "C" int radmmm_rowgemm_h3_colsum_rows(const radmmm_rowgemm_h3_desc* d) {
  if (!d) return 0;
  const radmmm_rowgemm_desc& p = d->base;
  if (p.M <= 0 || p.N <= 0 || !p.colsum_scratch) return 0;
  if (takes_narrow_kernel(*d)) return 0;
  return radmmm::h3w_colsum_rows(*d);
}

// scratch of the optional column sums (radmmm_rowgemm_desc.colsum_scratch): one partial row per row tile of the smallest
// tile height (padded to whole column tiles), or what radmmm_colsum wants when a launch falls back to it
extern "C" int64_t radmmm_rowgemm_h3_colsum_scratch_floats(int M, int N) {
  const int64_t fused = (int64_t)((M + 127) / 128) * N;
  const int64_t plain = radmmm_colsum_scratch_floats(M, N);
  return fused > plain ? fused : plain;
}
