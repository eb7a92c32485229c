// rowgemm_h3w: wide-tile version of the split-f16 conv GEMM (same contract as rowgemm_h3.hip).
//
// PMC on the 128x128 kernel (profiles/r01_pmc_h3.txt) showed MFMA busy 32 %, LDS busy 32 % and
// the waves waiting 52 % of their cycles: two small workgroups per CU with one barrier per 24
// MFMAs are latency bound.  This version gives every CU ONE workgroup with a (32*MB) x 256 tile:
//
//   * 4 waves = 4 column groups of 64; each wave owns all MB row blocks -> MB x 2 accumulators of
//     32x32 (up to 256 registers; one wave per SIMD has the full 512-register file);
//   * per 16-deep k block a wave reads 2*MB A fragments + 4 B fragments for 6*MB MFMAs (0.43
//     ds_read_b128 per MFMA at MB = 7, the 128x128 kernel: 0.67) and a K step (32) carries
//     12*MB MFMAs per wave between barriers (84 vs 24);
//   * LDS rows are 64 B (32 halves) with the 16-byte chunk index XOR-swizzled by row bits 2..3:
//     no padding, conflict-free for the 16-lane groups of ds_read_b128 and for the staging
//     stores (stage = (2*32*MB + 512) * 64 B <= 64 KiB, double buffered);
//   * MB in {4..8} is chosen by the host so that ceil(M / 32MB) * ceil(N / 256) fills whole
//     rounds of the 256 CUs: 12 800 frames x 1024 channels -> MB = 7 -> 58 x 4 = 232 workgroups
//     in one round (89 % of the MFMA slots useful; 128-row tiles: 78 %).
#include <stdlib.h>

#include "common.h"
#include "rowgemm_epilogue.h"
#include "split_pack.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BN = 256, BK = 32, ROWB = 64;
constexpr int OOB = 0x7fffffff;

template <int MB>
struct Geo {
  static constexpr int BMR = MB * 32;
  static constexpr int A_BYTES = BMR * ROWB;     // one of {Ah, Al}
  static constexpr int B_BYTES = BN * ROWB;      // one of {Bh, Bl}
  static constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int SMEM = 2 * STAGE + 4096;   // + dump area for the masked half of an odd A pass
  static constexpr int AI = (MB + 1) / 2;        // A rows staged per thread (64 rows per pass)
};

// Fast path of the fused epilogue for one thread's 4 columns of one row: everything that does not
// depend on the row (column validity, bias, base pointers) is hoisted by the caller, all accesses are
// 16-byte (8-byte for the fp16 copies) and the option flags are wave-uniform branches.  Preconditions
// (checked once per workgroup): EpilogueCtx::vec_ok, N % 4 == 0.  Same arithmetic and order as
// radmmm::epilogue_store4_pre.  The side inputs (add, dact_src, C2) arrive in registers: the caller
// fetched them before the block was parked in LDS, so their latency is not exposed here.
// All indices are 32-bit element offsets from the (wave-uniform) base pointers: M * ld < 2^31 is checked
// by the host, and a scalar base + 32-bit vector offset keeps the epilogue's VGPR demand small enough
// not to push accumulators into scratch.
struct EpiConst {
  float b0, b1, b2, b3;          // bias of this thread's 4 columns
  float ch_mul, c2h_mul;         // 2^x8_exp of the two split outputs
};

__device__ __forceinline__ float epilogue_row_fast(const radmmm_rowgemm_desc& p, int row, int col, float4 a4, float maskv,
                                                   float ratio, const EpiConst& ec, float4 addv, float4 dsv, float4 c2v) {
  const float pre = (p.pconv ? ratio : 1.f) * (p.premask ? maskv : 1.f);
  const float post = (p.postmask ? maskv : 1.f);
  const float rsc = p.rowscale == 1 ? maskv : (p.rowscale == 2 ? maskv * ratio : 1.f);
  auto one = [&](float acc, float bias, float add, float ds) __attribute__((always_inline)) {
    float x = (acc * pre + bias + add) * post;
    if (p.dact) x *= radmmm::dact_from_out(ds, p.dact);
    return radmmm::act_apply(x * rsc, p.act);
  };
  const float v0 = one(a4.x, ec.b0, addv.x, dsv.x), v1 = one(a4.y, ec.b1, addv.y, dsv.y);
  const float v2 = one(a4.z, ec.b2, addv.z, dsv.z), v3 = one(a4.w, ec.b3, addv.w, dsv.w);
  c2v.x += v0; c2v.y += v1; c2v.z += v2; c2v.w += v3;
  float amax = 0.f;
  if (p.Ch)
    amax = radmmm::store_split4_fmt(p.Ch, p.Cl, (unsigned)(row * p.ldch), col, p.split_fmt, ec.ch_mul, p.ch_scale, v0, v1, v2, v3,
                                    p.Clo);
  if (p.C2h)
    amax = fmaxf(amax, radmmm::store_split4_fmt(p.C2h, p.C2l, (unsigned)(row * p.ldc2h), col, p.split_fmt, ec.c2h_mul,
                                                p.c2h_scale, c2v.x, c2v.y, c2v.z, c2v.w));
  *reinterpret_cast<float4*>(p.C + (unsigned)(row * p.ldc + col)) = make_float4(v0, v1, v2, v3);
  if (p.C2) *reinterpret_cast<float4*>(p.C2 + (unsigned)(row * p.ldc2 + col)) = c2v;
  return amax;
}

// side inputs of one output row (4 columns) of the fused epilogue, requested ahead of their use
struct Side {
  float4 a, d, c;      // add, dact_src, C2
};

__device__ __forceinline__ void fetch_side(const radmmm_rowgemm_desc& p, bool live, int row, int col, float4& a, float4& d,
                                           float4& c) {
  a = d = c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && row < p.M) {
    if (p.add) a = *reinterpret_cast<const float4*>(p.add + (unsigned)(row * p.ldadd + col));
    if (p.dact) d = *reinterpret_cast<const float4*>(p.dact_src + (unsigned)(row * p.lddact + col));
    if (p.C2 && p.c2_accum) c = *reinterpret_cast<const float4*>(p.C2 + (unsigned)(row * p.ldc2 + col));
  }
}

// Epilogue of row block I (compile-time index: a runtime-indexed accumulator array would live in
// scratch): the four waves park their 32x64 pieces in LDS, then all threads run the fused epilogue
// on coalesced float4 rows.  A thread owns rows rl = 4 k + (tid >> 6), k = 0..7, of the block and handles them in four
// pairs; the side inputs (add, dact_src, C2) of the first pair are requested before the park and those of pair n + 1
// before pair n is processed, so that no global-load latency sits between the LDS read-out and the stores (one
// workgroup per CU: nothing else would hide it).
template <int MB, int I, bool FAST>
__device__ __forceinline__ void epilogue_blocks(const f32x16 (&acc)[MB][2], float* smf, const float2* rowf,
                                                const radmmm_rowgemm_desc& p, const radmmm::EpilogueCtx& ec, float sc,
                                                int m0, int n0, int tid, int lane, int wave, const float (&biasv)[4],
                                                const EpiConst& kc, float& sat) {
  if constexpr (I < MB) {
    const int c4 = (tid & 63) * 4;
    const int rbase = m0 + I * 32 + (tid >> 6);
    const bool live = FAST && (m0 + I * 32 < p.M) && (n0 + c4 < p.N);
    // side inputs of the pair being processed (c*) and of the next pair (n*); plain structs, no arrays.  The row loop
    // stays ROLLED (one pair per trip): seven blocks x eight fully unrolled rows of this epilogue are ~130 KB of code,
    // twice the instruction cache, and cost ~50 us per launch in instruction fetch (measured: 1-tap launch 107 -> 162 us)
    Side c0, c1, n0s, n1s;
    if constexpr (FAST) {
      fetch_side(p, live, rbase, n0 + c4, c0.a, c0.d, c0.c);
      fetch_side(p, live, rbase + 4, n0 + c4, c1.a, c1.d, c1.c);
    }
    if (I > 0) radmmm::lds_barrier();      // the previous block has been read out (its global stores stay in flight)
    float* wbase = smf + (4 * (lane >> 5)) * BN + wave * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) wbase[((e & 3) + 8 * (e >> 2)) * BN + j * 32] = acc[I][j][e] * sc;
    radmmm::lds_barrier();
    if (m0 + I * 32 < p.M) {
      if constexpr (FAST) {
        if (n0 + c4 < p.N) {
#pragma unroll 1
          for (int pr = 0; pr < 4; ++pr) {
            const int rl0 = (2 * pr) * 4 + (tid >> 6), rl1 = rl0 + 4;
            // next pair's side inputs (the last trip fetches nothing: its rows lie beyond the block -> `live` off)
            fetch_side(p, live && pr < 3, rbase + (2 * pr + 2) * 4, n0 + c4, n0s.a, n0s.d, n0s.c);
            fetch_side(p, live && pr < 3, rbase + (2 * pr + 3) * 4, n0 + c4, n1s.a, n1s.d, n1s.c);
            if (m0 + I * 32 + rl0 < p.M) {
              const float4 a4 = *reinterpret_cast<const float4*>(smf + rl0 * BN + c4);
              const float2 rf = rowf[I * 32 + rl0];
              sat = fmaxf(sat, epilogue_row_fast(p, m0 + I * 32 + rl0, n0 + c4, a4, rf.x, rf.y, kc, c0.a, c0.d, c0.c));
            }
            if (m0 + I * 32 + rl1 < p.M) {
              const float4 a4 = *reinterpret_cast<const float4*>(smf + rl1 * BN + c4);
              const float2 rf = rowf[I * 32 + rl1];
              sat = fmaxf(sat, epilogue_row_fast(p, m0 + I * 32 + rl1, n0 + c4, a4, rf.x, rf.y, kc, c1.a, c1.d, c1.c));
            }
            c0 = n0s;
            c1 = n1s;
          }
        }
      } else {
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
          const int rl = k * 4 + (tid >> 6);
          const float4 a4 = *reinterpret_cast<const float4*>(smf + rl * BN + c4);
          const float2 rf = rowf[I * 32 + rl];
          sat = fmaxf(sat, radmmm::epilogue_store4_pre(p, ec, m0 + I * 32 + rl, n0 + c4, a4, rf.x, rf.y, biasv));
        }
      }
    }
    epilogue_blocks<MB, I + 1, FAST>(acc, smf, rowf, p, ec, sc, m0, n0, tid, lane, wave, biasv, kc, sat);
  }
}

// instruction-order pinning of the K step (sched_group_barrier masks); without it the scheduler keeps a single
// ds_read in flight and every MFMA group waits for LDS (measured: 39 % MFMA busy inside a workgroup)
constexpr int SGB_VMEM = 0x020, SGB_MFMA = 0x008, SGB_DSR = 0x100;
constexpr int LOOKAHEAD = 2;    // fragment look-ahead in pipeline items
constexpr int DPI = 2;          // DMA pieces issued per pipeline item

// ---------------------------------------------------------------------------------------------------
// LDS-DMA staging: the operand tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), no
// staging registers and no ds_write instructions.  A wave instruction writes 1 KiB = 16 LDS rows
// lane-linearly (lane l -> row l >> 2, 16-byte slot l & 3), so the XOR swizzle is applied on the
// SOURCE side: lane l fetches chunk (l & 3) ^ ((l >> 4) & 3) of its row.  Out-of-range lanes (rows
// beyond M / N, frames outside the utterance or masked) get an out-of-range buffer offset and
// the DMA writes zeros.  The tile for step s + 1 is issued, two pieces per item, early in step s into
// the other LDS stage and has the rest of the step to land; the barrier's vmcnt(0) retires it.
// PR = product scheme: 3 split-f16 (Ah.Bh + Ah.Bl + Al.Bh on the f16 pipe, fp32-class accuracy); 1 plain fp16
// operands (the hi halves only -- the "16-bit throughput mode", half the operand traffic and a third of the MFMAs);
// 2 "FP8 cross terms": Ah.Bh on the f16 pipe and Ah.Bl + Al.Bh as ONE block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 per
// 32-deep k step and output tile, reading the 8-bit cross arrays (split_pack.h) through the very same LDS tile, DMA
// pieces and ds_read_b128 pattern as the f16 lo arrays: chunk c of a 64-byte row is k 16c..16c+15 of hi8 (c < 2) or lo8
// (c >= 2), and the instruction wants from lane (row, half h) exactly chunk h then chunk 2 + h (measured layout,
// tools/mfma_f8_layout.hip).  MFMA time 2/3 of the split-f16 scheme, same operand bytes.
template <int MB, int PR>
struct Pieces {
  static constexpr int A = PR != 1 ? MB : (2 * MB + 3) / 4;   // DMA pieces of A per wave
  static constexpr int B = PR != 1 ? 8 : 4;
  static constexpr int N = A + B;
};

template <int MB, int T, int PR>
__device__ __forceinline__ void pin_items_dma() {
  constexpr int NT = 2 * MB, NPT = Pieces<MB, PR>::N;
  if constexpr (T < NT - LOOKAHEAD) {
    __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR != 1 ? 2 : 1, 0);
    if constexpr (PR != 2 && T + LOOKAHEAD == MB) __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR == 3 ? 4 : 2, 0);
    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, PR == 2 ? (T == 0 ? 2 : 3) : 2 * PR, 0);
    constexpr int lo = DPI * T, hi = (DPI * (T + 1) < NPT) ? DPI * (T + 1) : NPT;
    if constexpr (hi > lo) __builtin_amdgcn_sched_group_barrier(SGB_VMEM, hi - lo, 0);
    pin_items_dma<MB, T + 1, PR>();
  }
}

typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

// 16 bytes per lane, global -> LDS at (wave-uniform dst) + 16 * lane.  Kept out of the kernel template
// and behind the device-compile guard: in the host pass the builtin is unknown and silently drops the
// whole kernel template's host stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

template <int MB, int PR = 3, bool FASTEPI = false>
__global__ __launch_bounds__(256, 1) void rowgemm_h3d_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes,
                                                              const int b_bytes) {
  using G = Geo<MB>;
  constexpr int NT = 2 * MB, D = LOOKAHEAD, NG = 2 * MB;                // NG: 16-row groups of an A array
  constexpr int NPA = Pieces<MB, PR>::A, NP = Pieces<MB, PR>::N;        // DMA pieces per wave
  static_assert(MB >= 4 && MB <= 8 && D <= MB && DPI * (NT - D) >= NP && PR >= 1 && PR <= 3, "pipeline shape");
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
  const int ntaps = p.taps + (q.extra_tap ? 1 : 0);     // + the optional extra K segment (include/radmmm_hip.h)
  const int nsteps = kpt * ntaps;
  const int extra_bytes = q.extra_a_rows * q.lda_h * 2; // byte distance of the extra segment's A rows

  // DMA pieces of one wave per step: MB pieces of A (the 4*MB 16-row groups of {Ah, Al} dealt round
  // robin to the 4 waves) + 8 pieces of B (4 groups of Bh, 4 of Bl).  This lane's row and chunk:
  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int a_t[NPA], a_lim[NPA], a_base[NPA], a_vo[NPA], a_dst[NPA], a_isl[NPA], b_voff[4], b_dst[4];
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int c = 4 * k + wave;                       // wave-uniform
    // PR == 3: the 4*MB groups of {Ah, Al}; PR == 1: the 2*MB groups of Ah, the surplus (odd MB) is a
    // zero-writing piece into the dump area behind the stages
    a_isl[k] = (PR != 1 && c >= NG) ? 1 : 0;
    const bool real = PR != 1 || c < NG;
    const int j = a_isl[k] ? c - NG : c;
    const int r = m0 + 16 * j + d_row;
    a_t[k] = 0;
    a_lim[k] = -1;
    a_base[k] = 0;
    if (real && r < p.M) {
      const int b = r / p.T;
      a_t[k] = r - b * p.T;
      a_lim[k] = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
      a_base[k] = (b * p.T * q.lda_h + d_chunk * 8) * 2;
    }
    a_dst[k] = real ? a_isl[k] * G::A_BYTES + j * 1024 : -1;
    a_vo[k] = OOB;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * k + wave;
    const int n = n0 + 16 * j + d_row;
    b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
    b_dst[k] = 2 * G::A_BYTES + j * 1024;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);

  // per-lane A offsets of the tap being fetched, branch-free (control flow would split the pinned
  // schedule): out-of-item / masked frames get OOB.  OOB + (k offset) stays >= 2^31 = out of range.
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    const bool ex = tap >= p.taps;                                 // the extra segment: no shift, rows of the second matrix
    const int s = ex ? 0 : p.sign * (tap - p.taps / 2) * p.dil;
    const int xb = ex ? extra_bytes : 0;
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int ts = a_t[k] + s;
      const int ok = -(int)((ts >= 0) & (ts < a_lim[k]));        // all ones when the frame is readable
      a_vo[k] = ((a_base[k] + ts * q.lda_h * 2 + xb) & ok) | (OOB & ~ok);
    }
  };
  // piece w of 0 .. NP-1 of tile (tap, kb) into stage `buf`
  auto dma_piece = [&](int buf, int w, int tap, int kb) __attribute__((always_inline)) {
    const int sbase = buf * G::STAGE;
    if (w < NPA) {
      const int dst = a_dst[w] < 0 ? 2 * G::STAGE + wave * 1024 : sbase + a_dst[w];
      dma16((PR != 1 && a_isl[w]) ? rAl : rAh, (lds_u32_ptr)(sm + dst), a_vo[w] + kb * (BK * 2));
    } else {
      const int k = (w - NPA) & 3, arr = (w - NPA) >> 2;
      const int vo = b_voff[k] + (int)(tap * q.b_tap_stride_h * 2) + kb * (BK * 2);
      dma16(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + sbase + b_dst[k] + arr * G::B_BYTES), vo);
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

  int l_tap = 0, l_kb = 0;                             // tile being fetched; clamped to the last one
  // Taps run INNERMOST: for one 32-channel k slice the taps re-read the same operand rows shifted by <= 2*dil frames,
  // i.e. lines that the previous tap brought into this XCD's L2 a K step ago.  With the taps outermost (round 1) a tap's
  // pass over all k slices pushed 10 MB through the 4 MB L2 before the next tap came back to the same rows: the A panel
  // was fetched from the Infinity Cache five times per launch (FETCH_SIZE 464 MB against 220 MB of per-XCD unique data).
  auto advance = [&]() __attribute__((always_inline)) {
    const bool last = (l_tap == ntaps - 1) && (l_kb == kpt - 1);
    const bool wrap = l_tap == ntaps - 1;
    l_tap = last ? l_tap : (wrap ? 0 : l_tap + 1);
    l_kb = (wrap && !last) ? l_kb + 1 : l_kb;
  };
  set_tap(0);
#pragma unroll
  for (int w = 0; w < NP; ++w) dma_piece(0, w, 0, 0);
  __syncthreads();
  f16x8 fah[NT], fal[NT], bh[2][2], bl[2][2];
  // item t of a K step = (k block kb, row block i): PR 1 / 3 run kb-major (all row blocks of k block 0, then of k block
  // 1); PR 2 runs i-major (t = 2 i + kb) because its scaled FP8 MFMA needs both k blocks of a row block's cross fragment
  auto item_kb = [](int t) { return PR == 2 ? (t & 1) : (t >= MB ? 1 : 0); };
  auto item_i = [](int t) { return PR == 2 ? (t >> 1) : (t >= MB ? t - MB : t); };
  // fragment readers of LDS stage `bsel`
  auto read_a = [&](int bsel, int t) __attribute__((always_inline)) {
    const unsigned char* st = sm + bsel * G::STAGE;
    const int fo = item_kb(t) ? f_off1 : f_off0;
    const int i = item_i(t);
    fah[t] = *reinterpret_cast<const f16x8*>(st + i * 32 * ROWB + fo);
    if constexpr (PR != 1) fal[t] = *reinterpret_cast<const f16x8*>(st + G::A_BYTES + i * 32 * ROWB + fo);
  };
  auto read_b = [&](int bsel, int kb) __attribute__((always_inline)) {
    const unsigned char* sB = sm + bsel * G::STAGE + 2 * G::A_BYTES + wave * 64 * ROWB;
    const int fo = kb ? f_off1 : f_off0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * ROWB + fo);
      if constexpr (PR != 1) bl[kb][j] = *reinterpret_cast<const f16x8*>(sB + G::B_BYTES + j * 32 * ROWB + fo);
    }
  };
  // PR 2: E8M0 block scales of the cross MFMA.  The scale byte of lane (row, half 0) applies to k block 0 = the first
  // 16 bytes of both halves' fragments, that of lane (row, half 1) to the second 16 bytes: A = [hi8 | lo8 * 2^11],
  // B = [lo8 * 2^11 | hi8], each further multiplied by 2^a8_exp / 2^b8_exp when it was written.
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
    const int kb = item_kb(t), i = item_i(t);
    if constexpr (PR == 1) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][1], acc[i][1], 0, 0, 0);
    } else if constexpr (PR == 2) {
      // 2 f16 + 1 scaled FP8 MFMA per item (the second cross MFMA of a row block rides with the next row block's first
      // item, the last one follows the loop): every item carries the same MFMA time
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][1], acc[i][1], 0, 0, 0);
      if (kb == 1) cross(i, 0);
      else if (i > 0) cross(i - 1, 1);
    } else {
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[t], bh[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[t], bh[kb][1], acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bl[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bl[kb][1], acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][0], acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[t], bh[kb][1], acc[i][1], 0, 0, 0);
    }
  };
  // first fragments of step 0; every later step gets them from the tail of the previous one
  read_b(0, 0);
  if constexpr (PR == 2) read_b(0, 1);
#pragma unroll
  for (int t = 0; t < D; ++t) read_a(0, t);
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    advance();                                       // -> tile step + 1
    set_tap(l_tap);
    // items 0 .. NT-D-1: fragments of item t + D | MFMAs of item t | two DMA pieces of tile step + 1
#pragma unroll
    for (int t = 0; t < NT - D; ++t) {
      read_a(buf, t + D);
      if (PR != 2 && t + D == MB) read_b(buf, 1);
      mfma_item(t);
#pragma unroll
      for (int q = 0; q < DPI; ++q)
        if (DPI * t + q < NP) dma_piece(buf ^ 1, DPI * t + q, l_tap, l_kb);
    }
    pin_items_dma<MB, 0, PR>();
    // every read of stage `buf` has been issued: retire them and this wave's DMA, meet the other
    // waves, then fetch the first fragments of the next step while the last D items' MFMAs run.
    // PR 2: the last items still need this step's B cross fragments and their own A cross fragments, so the next
    // step's fragments go to registers only after those MFMAs have been issued (program order below).
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PR == 2) {
#pragma unroll
      for (int t = NT - D; t < NT; ++t) mfma_item(t);
      cross(MB - 1, 1);
      read_b(buf ^ 1, 0);
      read_b(buf ^ 1, 1);
#pragma unroll
      for (int t = 0; t < D; ++t) read_a(buf ^ 1, t);
    } else {
      read_b(buf ^ 1, 0);
#pragma unroll
      for (int t = 0; t < D; ++t) read_a(buf ^ 1, t);
#pragma unroll
      for (int t = NT - D; t < NT; ++t) mfma_item(t);
      __builtin_amdgcn_sched_group_barrier(SGB_DSR, PR == 3 ? 4 + 2 * D : 2 + D, 0);
      __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 2 * PR * D, 0);
    }
  }
  __syncthreads();                                   // stray fragment reads / DMA of the clamped extra tile

  const radmmm::EpilogueCtx ec(p);
  float* smf = reinterpret_cast<float*>(sm);
  float2* rowf = reinterpret_cast<float2*>(sm + 32768);
  if (tid < G::BMR) {
    float mk, rt;
    radmmm::epilogue_row_factors(p, ec, m0 + tid, mk, rt);
    rowf[tid] = make_float2(mk, rt);
  }
  float biasv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const int c = n0 + (tid & 63) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) biasv[e] = (c + e < p.N) ? p.bias[c + e] : 0.f;
  }
  const EpiConst kc = {biasv[0], biasv[1], biasv[2], biasv[3], __builtin_ldexpf(1.f, p.ch_x8_exp), __builtin_ldexpf(1.f, p.c2h_x8_exp)};
  float sat = 0.f;
  epilogue_blocks<MB, 0, FASTEPI>(acc, smf, rowf, p, ec, q.acc_scale, m0, n0, tid, lane, wave, biasv, kc, sat);
  radmmm::raise_sat_flag(p.sat_flag, sat);
}

template <int MB, int PR = 3, bool FASTEPI = false>
int launch_dma(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = Geo<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_h3d_kernel<MB, PR, FASTEPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_h3d<%d>): %s", MB, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_h3d_kernel<MB, PR, FASTEPI>), dim3(ntm * ntn), dim3(256), G::SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_h3d");
}

}  // namespace

namespace radmmm {

// Row-block count per workgroup.  Cost model fitted to measurements at M = 12 800 and 32 000
// (DESIGN.md §4.2): a workgroup costs (MB + 1.5) units (MFMA work ~ MB, the 256-row B tile and
// the fixed parts ~ 1.5); the grid runs `full` rounds of all CUs plus a tail round which, because the
// chip is power bound, runs faster when few CUs are busy: 0.65 + 0.35 * (tail workgroups / CUs).
int pick_h3w_mb(int M, int N, int slots) {
  const int ntn = (N + BN - 1) / BN;
  int best = 4;
  double best_cost = 1e300;
  // MB = 8 (256-row tiles, 20-36 bytes of scratch per lane) costs what this model says it should on the shape it was
  // measured on (12 800 x 1024: 405 us against 380 us at MB = 7, model 8.77 : 8.22) and is taken where it saves a round
  // of workgroups: N = 1152 (the start conv's data gradient) and N = 1052 (the LSTM's input gradient) fit 50 x 5 = 250
  // workgroups into ONE round instead of 290-500 in two.  RADMMM_H3W_MB8=0: candidates 4..7 only (A/B runs).
  static const int mb_max = (getenv("RADMMM_H3W_MB8") && atoi(getenv("RADMMM_H3W_MB8")) == 0) ? 7 : 8;
  for (int mb = 4; mb <= mb_max; ++mb) {
    const long long wg = (long long)((M + 32 * mb - 1) / (32 * mb)) * ntn;
    const long long full = wg / slots, tail = wg % slots;
    const double rounds = (double)full + (tail ? 0.65 + 0.35 * (double)tail / slots : 0.0);
    const double cost = rounds * (mb + 1.5);
    if (cost <= best_cost + 1e-9) {
      best_cost = cost;
      best = mb;
    }
  }
  return best;
}

// Workgroup slots a GEMM grid is sized for: the device's CUs, or RADMMM_GEMM_CUS when set (data-parallel runs
// leave a few CUs to RCCL's channel kernels so that an all-reduce landing during a GEMM launch does not push the
// grid into a second round: rad_mmm_amd/ddp.py reserve_collective_cus).  Read once per process.
int gemm_cu_slots() {
  static const int slots = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    if (const char* e = getenv("RADMMM_GEMM_CUS")) {
      const int v = atoi(e);
      if (v >= 32 && v <= n) n = v;
    }
    return n;
  }();
  return slots;
}

int launch_rowgemm_h3w(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  int mb = pick_h3w_mb(d.base.M, d.base.N, gemm_cu_slots());
  if (const char* e = getenv("RADMMM_H3W_MB")) {
    const int v = atoi(e);
    if (v >= 4 && v <= 8) mb = v;
  }
  // lean epilogue (16-byte accesses, 32-bit element offsets) when every output / side input allows it
  const radmmm_rowgemm_desc& p = d.base;
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  auto a8 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 7) == 0; };
  auto fits = [&](long long ld) { return (long long)p.M * ld < 0x7fffffffLL; };
  const bool fast = p.N % 4 == 0 && p.ldc % 4 == 0 && a16(p.C) && fits(p.ldc) &&
                    (!p.add || (p.ldadd % 4 == 0 && a16(p.add) && fits(p.ldadd))) &&
                    (!p.dact || (p.lddact % 4 == 0 && a16(p.dact_src) && fits(p.lddact))) &&
                    (!p.C2 || (p.ldc2 % 4 == 0 && a16(p.C2) && fits(p.ldc2))) &&
                    (!p.Ch || (p.ldch % 4 == 0 && fits(p.ldch) && a8(p.Ch) && a8(p.Cl) && a8(p.Clo))) &&
                    (!p.C2h || (p.ldc2h % 4 == 0 && fits(p.ldc2h) && a8(p.C2h) && a8(p.C2l)));
  // (the 8-bit split formats additionally need ld % 32 == 0 and 4-byte aligned cross arrays: checked by the caller
  //  radmmm_rowgemm_h3 for every path)
#define RADMMM_H3D_CASE(MBV, PRV)                                                 \
  case MBV:                                                                       \
    return fast ? launch_dma<MBV, PRV, true>(d, stream, a_bytes, b_bytes)         \
                : launch_dma<MBV, PRV, false>(d, stream, a_bytes, b_bytes);
  if (d.nprod == 2) {                                // FP8 cross terms
    switch (mb) {
      RADMMM_H3D_CASE(4, 2) RADMMM_H3D_CASE(5, 2) RADMMM_H3D_CASE(6, 2) RADMMM_H3D_CASE(7, 2)
      default: return fast ? launch_dma<8, 2, true>(d, stream, a_bytes, b_bytes) : launch_dma<8, 2, false>(d, stream, a_bytes, b_bytes);
    }
  }
  if (d.nprod == 1) {                                // 16-bit throughput mode: hi halves only
    switch (mb) {
      RADMMM_H3D_CASE(4, 1) RADMMM_H3D_CASE(5, 1) RADMMM_H3D_CASE(6, 1) RADMMM_H3D_CASE(7, 1)
      default: return fast ? launch_dma<8, 1, true>(d, stream, a_bytes, b_bytes) : launch_dma<8, 1, false>(d, stream, a_bytes, b_bytes);
    }
  }
  switch (mb) {
    RADMMM_H3D_CASE(4, 3) RADMMM_H3D_CASE(5, 3) RADMMM_H3D_CASE(6, 3) RADMMM_H3D_CASE(7, 3)
    default: return fast ? launch_dma<8, 3, true>(d, stream, a_bytes, b_bytes) : launch_dma<8, 3, false>(d, stream, a_bytes, b_bytes);
  }
#undef RADMMM_H3D_CASE
}

}  // namespace radmmm

extern "C" int radmmm_gemm_cu_slots(void) { return radmmm::gemm_cu_slots(); }
