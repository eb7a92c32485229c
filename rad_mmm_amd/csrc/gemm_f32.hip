// fp32 MFMA GEMM kernels for the WN conv stack (gfx950 / CDNA4 only).
//
//  rowgemm_f32  : channels-last Conv1d family (forward and data-gradient), taps folded into
//                 the K loop as row-shifted re-reads of the activation matrix.
//  wgrad_f32    : weight gradient (contraction over frames), split-K into slabs.
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): a 128x128 workgroup
// tile, 4 waves in a 2x2 grid, each wave a 2x2 grid of 32x32 MFMA tiles (64 accumulator
// registers).  Operand tiles live in LDS k-major ([BK][128] floats) so that the MFMA
// fragment read (lane l: element [k = l>>5][i = l&31]) is one conflict-free ds_read_b32
// whichever way the operand is laid out in global memory; the two global layouts differ
// only in how the tile is staged (transposing 4x ds_write_b32 vs straight ds_write_b128).
// Global->LDS staging goes through registers, double-buffered in LDS: the loads of tile
// i+1 are issued before the MFMA loop of tile i and written to the other buffer after it,
// one barrier per K step.  fp32 MFMA issues once per 64 cycles per SIMD, which leaves
// ample issue room for the staging traffic (see DESIGN.md for the cycle budget).
#include <stdlib.h>

#include "common.h"
#include "rowgemm_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = 128;                 // LDS row stride (floats); no padding needed, see above
constexpr int TILE = BK * LDT;           // floats per operand tile
constexpr int SMEM_BYTES = 4 * TILE * 4; // 2 operands x 2 buffers = 64 KiB

__device__ __forceinline__ int xcd_remap(int wg, int nt) {
  // Workgroup b is dispatched to XCD b % 8 (observed; speed only).  Give each XCD a
  // contiguous chunk of tile ids so tiles that share an A panel share an L2.
  const int xcd = wg & 7, loc = wg >> 3;
  const int q = nt >> 3, r = nt & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__device__ __forceinline__ float4 sel4(float4 v, bool ok, int k, int K) {
  v.x = (ok && k + 0 < K) ? v.x : 0.f;
  v.y = (ok && k + 1 < K) ? v.y : 0.f;
  v.z = (ok && k + 2 < K) ? v.z : 0.f;
  v.w = (ok && k + 3 < K) ? v.w : 0.f;
  return v;
}

__device__ __forceinline__ void mfma_step(const float* as, const float* bs, f32x16 (&acc)[2][2]) {
  // as/bs already offset by wave sub-tile + (lane&31) + (lane>>5)*LDT.  Fragments of k-pair
  // kk+1 are read from LDS before the MFMAs of k-pair kk issue, so the LDS latency sits
  // under 4 MFMAs (256 cycles) instead of in front of them.
  float a[2][2], b[2][2];
  a[0][0] = as[0]; a[0][1] = as[32];
  b[0][0] = bs[0]; b[0][1] = bs[32];
#pragma unroll
  for (int kk = 0; kk < BK / 2; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < BK / 2) {
      a[nxt][0] = as[(kk + 1) * 2 * LDT]; a[nxt][1] = as[(kk + 1) * 2 * LDT + 32];
      b[nxt][0] = bs[(kk + 1) * 2 * LDT]; b[nxt][1] = bs[(kk + 1) * 2 * LDT + 32];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this k-pair's MFMAs
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][0], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][1], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][0], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][1], acc[1][1], 0, 0, 0);
  }
}

// Write a workgroup's accumulators to LDS as a row-major [128][128] fp32 tile.  MFMA 32x32
// C/D layout: lane l, register e -> column l&31, row (e&3) + 8*(e>>2) + 4*(l>>5).
__device__ __forceinline__ void stage_acc_to_lds(float* smem, const f32x16 (&acc)[2][2], int wm,
                                                 int wn, int lane) {
  float* base = smem + (wm * 64 + 4 * (lane >> 5)) * BN + wn * 64 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        base[(mi * 32 + (e & 3) + 8 * (e >> 2)) * BN + ni * 32] = acc[mi][ni][e];
}

// ------------------------------------------------------------------------------------
// rowgemm
// ------------------------------------------------------------------------------------
template <int B_LAYOUT>
__global__ __launch_bounds__(256, 2) void rowgemm_f32_kernel(const radmmm_rowgemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;             // [2][BK][LDT]
  float* Bs = smem + 2 * TILE;  // [2][BK][LDT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntn * ntm);
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kpt = (p.K + BK - 1) / BK;  // K steps per tap
  const int nsteps = kpt * p.taps;

  // ---- per-thread staging coordinates -------------------------------------------------
  // A (always [row][k], k contiguous): thread owns row a_row, 16-byte chunks a_kc + 2*i
  const int a_row = tid & 127, a_kc = tid >> 7;
  const int r = m0 + a_row;
  const bool r_ok = r < p.M;
  int t_in = 0, lim = 0;
  const float* a_item_ptr = p.A;   // frame 0 of this thread's item (always valid memory)
  if (r_ok) {
    const int b = r / p.T;
    t_in = r - b * p.T;
    lim = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
    a_item_ptr = p.A + (p.a_item_stride ? (long long)b * p.a_item_stride : (long long)b * p.T * p.lda);
  }
  // B layout 0 ([n][k]): same shape as A.  B layout 1 ([k][n]): thread owns 4 n at b_n4*4,
  // k rows b_k0 + 8*i.
  const int b_n4 = tid & 31, b_k0 = tid >> 5;
  const int bn = n0 + a_row;  // layout 0 row
  const bool bn_ok = bn < p.N;

  float4 ra[4], rb[4];

  // Loads only ISSUE here (addresses clamped to valid memory); the validity selects are
  // applied in store_tiles, i.e. after the MFMA loop of the current tile, so that the
  // s_waitcnt for these loads lands behind the MFMAs and not in front of them.
  auto load_tiles = [&](int step) {
    const int tap = step / kpt, kb = step - tap * kpt;
    const int s = p.sign * (tap - p.taps / 2) * p.dil;
    const int ts = t_in + s;
    const bool av = r_ok && ts >= 0 && ts < lim;
    const float* arow = a_item_ptr + (long long)(av ? ts : 0) * p.lda;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb * BK + (a_kc + 2 * i) * 4;
      ra[i] = *reinterpret_cast<const float4*>(arow + ((av && k < p.K) ? k : 0));
    }
    const float* bbase = p.B + (long long)tap * p.b_tap_stride;
    if (B_LAYOUT == 0) {
      const float* brow = bbase + (long long)(bn_ok ? bn : 0) * p.ldb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kb * BK + (a_kc + 2 * i) * 4;
        rb[i] = *reinterpret_cast<const float4*>(brow + ((bn_ok && k < p.K) ? k : 0));
      }
    } else {
      const int n = n0 + b_n4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kb * BK + b_k0 + 8 * i;
        const bool kv = k < p.K && n < p.N;
        rb[i] = *reinterpret_cast<const float4*>(bbase + (long long)(kv ? k : 0) * p.ldb + (kv ? n : 0));
      }
    }
  };
  auto store_tiles = [&](int step, int buf) {
    const int tap = step / kpt, kb = step - tap * kpt;
    const int ts = t_in + p.sign * (tap - p.taps / 2) * p.dil;
    const bool av = r_ok && ts >= 0 && ts < lim;
    float* as = As + buf * TILE;
    float* bs = Bs + buf * TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kc = (a_kc + 2 * i) * 4;
      const int k = kb * BK + kc;
      const float4 v = sel4(ra[i], av && k < p.K, k, p.K);
      as[(kc + 0) * LDT + a_row] = v.x;
      as[(kc + 1) * LDT + a_row] = v.y;
      as[(kc + 2) * LDT + a_row] = v.z;
      as[(kc + 3) * LDT + a_row] = v.w;
    }
    if (B_LAYOUT == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kc = (a_kc + 2 * i) * 4;
        const int k = kb * BK + kc;
        const float4 v = sel4(rb[i], bn_ok && k < p.K, k, p.K);
        bs[(kc + 0) * LDT + a_row] = v.x;
        bs[(kc + 1) * LDT + a_row] = v.y;
        bs[(kc + 2) * LDT + a_row] = v.z;
        bs[(kc + 3) * LDT + a_row] = v.w;
      }
    } else {
      const int n = n0 + b_n4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kb * BK + b_k0 + 8 * i;
        *reinterpret_cast<float4*>(bs + (b_k0 + 8 * i) * LDT + b_n4 * 4) = sel4(rb[i], k < p.K && n < p.N, n, p.N);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  const int frag_off = (lane >> 5) * LDT + (lane & 31);
  for (int step = 0; step < nsteps; ++step) {
    // single basic block: the last iteration re-stages the final tile into the idle buffer
    // instead of branching, so the loads' wait sits behind the MFMAs in every iteration
    const int buf = step & 1;
    const int nxt = step + 1 < nsteps ? step + 1 : step;
    load_tiles(nxt);
    mfma_step(As + buf * TILE + wm * 64 + frag_off, Bs + buf * TILE + wn * 64 + frag_off, acc);
    store_tiles(nxt, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------
  // Stage the 128x128 accumulator tile through LDS (free after the K loop; the loop's last
  // barrier has passed) so that global traffic is row-contiguous float4 and the per-row
  // mask/ratio is computed once per 4 outputs.
  stage_acc_to_lds(smem, acc, wm, wn, lane);
  __syncthreads();
  const radmmm::EpilogueCtx ec(p);
  const int c4 = (tid & 31) * 4;
  for (int i = 0; i < 16; ++i) {
    const int rl = i * 8 + (tid >> 5);
    const float4 a4 = *reinterpret_cast<const float4*>(smem + rl * BN + c4);
    radmmm::epilogue_store4(p, ec, m0 + rl, n0 + c4, a4);
  }
}

// ------------------------------------------------------------------------------------
// wgrad: P[split][tap][m][n] = sum_r GY[r][m] * Xm[r + shift(tap)][n]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const radmmm_wgrad_desc p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.Mc + BM - 1) / BM, ntn = (p.Nc + BN - 1) / BN;
  // blockIdx.x -> (tile_n fastest, tile_m, tap, split)
  int id = blockIdx.x;
  const int tn = id % ntn; id /= ntn;
  const int tm = id % ntm; id /= ntm;
  const int tap = id % p.taps;
  const int split = id / p.taps;
  const int m0 = tm * BM, n0 = tn * BN;
  const int shift = (tap - p.taps / 2) * p.dil;

  // rows of this split, in units of BK
  const int steps_total = (p.R + BK - 1) / BK;
  const int steps_per = (steps_total + p.splits - 1) / p.splits;
  const int step_lo = split * steps_per;
  int step_hi = step_lo + steps_per;
  if (step_hi > steps_total) step_hi = steps_total;

  const int c4 = tid & 31, k0 = tid >> 5;  // 4 columns at c4*4, k rows k0 + 8*i
  const int am = m0 + c4 * 4, bn = n0 + c4 * 4;

  float4 ra[4], rb[4];
  auto x_row = [&](int rr, bool& ok) {   // source row of the (shifted, masked) activation operand
    ok = rr < p.R && bn < p.Nc;
    int src = 0;
    if (ok) {
      const int b = rr / p.T;
      const int t = rr - b * p.T + shift;
      const int lim = (p.x_mask_mode && p.lens) ? p.lens[b] : p.T;
      ok = t >= 0 && t < lim;
      src = rr + shift;
    }
    return ok ? src : 0;
  };
  auto load_tiles = [&](int step) {     // issue only; selects happen in store_tiles
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = step * BK + k0 + 8 * i;
      const bool aok = rr < p.R && am < p.Mc;
      ra[i] = *reinterpret_cast<const float4*>(p.GY + (long long)(aok ? rr : 0) * p.ldgy + (aok ? am : 0));
      bool bok;
      const int src = x_row(rr, bok);
      rb[i] = *reinterpret_cast<const float4*>(p.X + (long long)src * p.ldx + (bok ? bn : 0));
    }
  };
  auto store_tiles = [&](int step, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = step * BK + k0 + 8 * i;
      const bool aok = rr < p.R && am < p.Mc;
      bool bok;
      (void)x_row(rr, bok);
      *reinterpret_cast<float4*>(As + buf * TILE + (k0 + 8 * i) * LDT + c4 * 4) = sel4(ra[i], aok, am, p.Mc);
      *reinterpret_cast<float4*>(Bs + buf * TILE + (k0 + 8 * i) * LDT + c4 * 4) = sel4(rb[i], bok, bn, p.Nc);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_off = (lane >> 5) * LDT + (lane & 31);
  if (step_lo < step_hi) {
    load_tiles(step_lo);
    store_tiles(step_lo, 0);
    __syncthreads();
    for (int step = step_lo; step < step_hi; ++step) {
      const int buf = (step - step_lo) & 1;
      const int nxt = step + 1 < step_hi ? step + 1 : step;
      load_tiles(nxt);
      mfma_step(As + buf * TILE + wm * 64 + frag_off, Bs + buf * TILE + wn * 64 + frag_off, acc);
      store_tiles(nxt, buf ^ 1);
      __syncthreads();
    }
  }

  float* P = p.P + (long long)split * p.split_stride + (long long)tap * p.Mc * p.ldp;
  stage_acc_to_lds(smem, acc, wm, wn, lane);
  __syncthreads();
  const bool vec_ok = (p.ldp % 4 == 0) && radmmm::aligned16(p.P) && (p.split_stride % 4 == 0);
  const int col = n0 + c4 * 4;
  for (int i = 0; i < 16; ++i) {
    const int rl = i * 8 + (tid >> 5);
    const int row = m0 + rl;
    if (row >= p.Mc || col >= p.Nc) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(smem + rl * BN + c4 * 4);
    if (vec_ok && col + 3 < p.Nc) {
      *reinterpret_cast<float4*>(P + (long long)row * p.ldp + col) = a4;
    } else {
      const float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < p.Nc) P[(long long)row * p.ldp + col + e] = v[e];
    }
  }
}

template <typename K>
int ensure_smem(K kernel) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
  if (e != hipSuccess) {
    radmmm::set_error("hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
    return -2;
  }
  return 0;
}

}  // namespace

extern "C" int radmmm_rowgemm_f32(const radmmm_rowgemm_desc* d, radmmm_stream_t stream) {
  RADMMM_REQUIRE(d != nullptr, "rowgemm: null descriptor");
  RADMMM_REQUIRE(d->A && d->B && d->C, "rowgemm: null operand");
  RADMMM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->taps >= 1 && d->T > 0, "rowgemm: bad dims M=%d N=%d K=%d taps=%d T=%d", d->M, d->N, d->K, d->taps, d->T);
  RADMMM_REQUIRE(d->M % d->T == 0, "rowgemm: M=%d is not a multiple of T=%d", d->M, d->T);
  RADMMM_REQUIRE(d->lda % 4 == 0 && radmmm::aligned16(d->A) && d->a_item_stride % 4 == 0, "rowgemm: A must be 16B aligned with lda %% 4 == 0 (lda=%d)", d->lda);
  RADMMM_REQUIRE(d->a_item_stride != 0 || d->lda >= ((d->K + 3) & ~3), "rowgemm: lda >= roundup4(K) required unless a_item_stride is set (lda=%d K=%d)", d->lda, d->K);
  if (d->b_layout == 0) {
    RADMMM_REQUIRE(d->ldb % 4 == 0 && d->ldb >= ((d->K + 3) & ~3), "rowgemm: B[n][k] needs ldb %% 4 == 0 and ldb >= roundup4(K) (ldb=%d K=%d)", d->ldb, d->K);
  } else {
    RADMMM_REQUIRE(d->b_layout == 1, "rowgemm: b_layout must be 0 or 1");
    RADMMM_REQUIRE(d->ldb % 4 == 0 && d->ldb >= ((d->N + 3) & ~3), "rowgemm: B[k][n] needs ldb %% 4 == 0 and ldb >= roundup4(N) (ldb=%d N=%d)", d->ldb, d->N);
  }
  RADMMM_REQUIRE(radmmm::aligned16(d->B) && d->b_tap_stride % 4 == 0, "rowgemm: B must be 16B aligned, tap stride %% 4 == 0");
  RADMMM_REQUIRE(d->sign == 1 || d->sign == -1, "rowgemm: sign must be +-1");
  RADMMM_REQUIRE(!(d->pconv || d->rowscale == 2) || (d->ratio_taps >= 1 && d->ratio_dil >= 1), "rowgemm: ratio_taps/ratio_dil required with pconv/rowscale=2");
  RADMMM_REQUIRE(!d->dact || d->dact_src, "rowgemm: dact needs dact_src");
  {
    // the flow steps' 160 x 160 channel mix and its data gradient: the whole weight in LDS, no barrier in the K loop
    // (rowgemm_mix.hip)
    const int rc = radmmm::launch_rowgemm_mix(*d, static_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  static const bool use32 = [] {
    const char* e = radmmm::debug_env("RADMMM_ROWGEMM_TILE");
    return e && atoi(e) == 32;
  }();
  if (!use32 && d->K % 16 == 0) {
    // fast path: 16-row-granular tiling with a VALU-free K loop (needs K % 16 == 0, operands < 2 GiB)
    const int rc = radmmm::launch_rowgemm16(*d, static_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  const int ntm = (d->M + BM - 1) / BM, ntn = (d->N + BN - 1) / BN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->b_layout == 0) {
    static int once = ensure_smem(rowgemm_f32_kernel<0>);
    if (once) return once;
    hipLaunchKernelGGL(rowgemm_f32_kernel<0>, dim3(ntm * ntn), dim3(256), SMEM_BYTES, s, *d);
  } else {
    static int once = ensure_smem(rowgemm_f32_kernel<1>);
    if (once) return once;
    hipLaunchKernelGGL(rowgemm_f32_kernel<1>, dim3(ntm * ntn), dim3(256), SMEM_BYTES, s, *d);
  }
  return radmmm::check_launch("rowgemm_f32");
}

extern "C" int radmmm_wgrad_f32(const radmmm_wgrad_desc* d, radmmm_stream_t stream) {
  RADMMM_REQUIRE(d != nullptr, "wgrad: null descriptor");
  RADMMM_REQUIRE(d->GY && d->X && d->P, "wgrad: null operand");
  RADMMM_REQUIRE(d->R > 0 && d->Mc > 0 && d->Nc > 0 && d->taps >= 1 && d->T > 0 && d->splits >= 1, "wgrad: bad dims");
  RADMMM_REQUIRE(d->R % d->T == 0, "wgrad: R=%d is not a multiple of T=%d", d->R, d->T);
  RADMMM_REQUIRE(d->ldgy % 4 == 0 && d->ldgy >= ((d->Mc + 3) & ~3) && radmmm::aligned16(d->GY), "wgrad: GY alignment (ldgy=%d Mc=%d)", d->ldgy, d->Mc);
  RADMMM_REQUIRE(d->ldx % 4 == 0 && d->ldx >= ((d->Nc + 3) & ~3) && radmmm::aligned16(d->X), "wgrad: X alignment (ldx=%d Nc=%d)", d->ldx, d->Nc);
  RADMMM_REQUIRE(d->ldp >= d->Nc, "wgrad: ldp < Nc");
  static const bool generic_only = [] {
    const char* e = radmmm::debug_env("RADMMM_ROWGEMM_TILE");
    return e && atoi(e) == 32;
  }();
  if (!generic_only) {
    const int rc = radmmm::launch_wgrad16(*d, static_cast<hipStream_t>(stream));
    if (rc <= 0) return rc;
  }
  const int ntm = (d->Mc + BM - 1) / BM, ntn = (d->Nc + BN - 1) / BN;
  static int once = ensure_smem(wgrad_f32_kernel);
  if (once) return once;
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(ntm * ntn * d->taps * d->splits), dim3(256), SMEM_BYTES,
                     static_cast<hipStream_t>(stream), *d);
  return radmmm::check_launch("wgrad_f32");
}
