// rowgemm_h3w_kernel.h: the wide-tile split conv GEMM kernel (same contract as rowgemm_h3.hip), included by one
// translation unit per product scheme (rowgemm_h3w_pr{1,2,3}.hip: they compile in parallel); host side: rowgemm_h3w.hip.
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
#pragma once
// timing-only build switches (results wrong): -DRADMMM_EPI_NOSTORE removes the epilogue's stores (its arithmetic stays),
// -DRADMMM_EPI_NONE the whole epilogue of the window kernel -- what the launch costs without them (tools/epi_cost.sh)
#ifdef RADMMM_EPI_NOSTORE
#define RADMMM_EPI_STORE(...) ((void)0)
#else
#define RADMMM_EPI_STORE(...) __VA_ARGS__
#endif
#include <stdlib.h>
#include <type_traits>

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
// operand format of the scaled cross-term MFMA: 0 = FP8 e4m3 (the product), 2 = FP6 e2m3 in a TIMING-ONLY build
// (-DRADMMM_X_FMT=2: wrong results; what would MXFP6 cross terms buy the K loop?  DESIGN 7)
#ifndef RADMMM_X_FMT
#define RADMMM_X_FMT 0
#endif
constexpr int OOB = 0x7fffffff;

// -DRADMMM_PHASE_TIMERS (measurement builds only, tools/phase_probe.py): every workgroup records the 100 MHz wall clock
// at kernel entry, after the prologue (first tile in LDS), after the K loop and after the epilogue.
#ifdef RADMMM_PHASE_TIMERS
__device__ unsigned long long g_phase[4096 * 4];
#define RADMMM_PHASE(i)                                                                      \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_phase[blockIdx.x * 4 + (i)] = wall_clock64(); \
  } while (0)
#else
#define RADMMM_PHASE(i) do {} while (0)
#endif

template <int MB>
struct Geo {
  static constexpr int BMR = MB * 32;
  static constexpr int A_BYTES = BMR * ROWB;     // one of {Ah, Al}
  static constexpr int B_BYTES = BN * ROWB;      // one of {Bh, Bl}
  static constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;
  static constexpr int SMEM = 2 * STAGE + 4096;   // + dump area for the masked half of an odd A pass
  static constexpr int AI = (MB + 1) / 2;        // A rows staged per thread (64 rows per pass)
};

// Generic epilogue of row block I (compile-time index: a runtime-indexed accumulator array would live in scratch): the
// four waves park their 32x64 pieces in LDS, then all threads run the fully general fused epilogue
// (radmmm::epilogue_store4_pre, every option of the descriptor, any alignment) on float4 rows.  Launches of the flow
// step do not come here: they take the direct epilogue below.
template <int MB, int I>
__device__ __forceinline__ void epilogue_blocks(const f32x16 (&acc)[MB][2], float* smf, const float2* rowf,
                                                const radmmm_rowgemm_desc& p, const radmmm::EpilogueCtx& ec, float sc,
                                                int m0, int n0, int tid, int lane, int wave, const float (&biasv)[4], float& sat) {
  if constexpr (I < MB) {
    const int c4 = (tid & 63) * 4;
    if (I > 0) radmmm::lds_barrier();      // the previous block has been read out (its global stores stay in flight)
    float* wbase = smf + (4 * (lane >> 5)) * BN + wave * 64 + 2 * (lane & 31);     // columns 2 jj + j (interleaved B rows)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) wbase[((e & 3) + 8 * (e >> 2)) * BN + j] = acc[I][j][e] * sc;
    radmmm::lds_barrier();
    if (m0 + I * 32 < p.M) {
#pragma unroll 1
      for (int k = 0; k < 8; ++k) {
        const int rl = k * 4 + (tid >> 6);
        const float4 a4 = *reinterpret_cast<const float4*>(smf + rl * BN + c4);
        const float2 rf = rowf[I * 32 + rl];
        sat = fmaxf(sat, radmmm::epilogue_store4_pre(p, ec, m0 + I * 32 + rl, n0 + c4, a4, rf.x, rf.y, biasv));
      }
    }
    epilogue_blocks<MB, I + 1>(acc, smf, rowf, p, ec, sc, m0, n0, tid, lane, wave, biasv, sat);
  }
}

// ---------------------------------------------------------------------------------------------------
// DIRECT epilogue (round 3).  In-kernel phase timers (tools/phase_probe.py, profiles/r03_phase_*.txt) showed the
// LDS-parking epilogue of round 2 taking 47 us (C only) to 83 us (partial conv + softplus + split copy) of a launch --
// more than the whole K loop of a 1x1 conv (53 us): its row loop carried `s_waitcnt vmcnt(0)` (on gfx9 the side-input
// loads share the in-order vmcnt counter with the stores, and around conditional loads the compiler waits for
// everything), i.e. one full store round trip per pair of rows.  This version
//   * leaves the accumulators where the MFMA put them: with the B rows interleaved (see the DMA setup) lane jj of a wave
//     owns columns 2 jj, 2 jj + 1 of 16 rows per 32-row block, so every output is an 8-byte (fp32) / 4-byte (fp16) /
//     2-byte (8-bit) store of one row pair -- 256 / 128 / 64 contiguous bytes per row and half wave; no LDS, no barrier,
//     the four waves run independently;
//   * addresses everything through buffer descriptors: per-lane offset (column, half) + scalar row offset; rows >= M and
//     columns >= N fall outside the descriptor's range and are dropped / read as zero by the hardware: no predication;
//   * is a template parameter of the kernel (EK_*: which arrays exist), so the row code has no conditional memory
//     operation and the side inputs of row block I + 1 are requested BEFORE block I's stores: the wait for them counts
//     the younger stores instead of draining them.
// A runtime loop over the row blocks copies one block (2 x 16 accumulators) into fixed registers through a chain of
// uniform branches (the asm statements keep the optimizer from turning the chain into a dynamic index -- which would
// move all accumulators to scratch -- or into 7-way selects), so the body exists once (~13 KB of code; seven unrolled
// copies do not fit the instruction cache: measured +50 us per launch in round 2).
// TIMING-ONLY builds (wrong results; tools/floor_probe.sh, profiles/r05_nprod1_floor.txt): what is a launch made of?
//   -DRADMMM_TIMING=1  the cross-term (FP8) MFMAs are not issued: half the matrix work, everything else in place
//   -DRADMMM_TIMING=2  "nprod = 1" in the instruction stream: additionally no cross-fragment LDS reads (the DMA stays: the
//                      counted vmcnt waits of the loops depend on the number of pieces)
//   -DRADMMM_TIMING=3  no MFMA at all (fragments are still read: operands pinned by empty asm): the launch's non-MFMA floor
#ifndef RADMMM_TIMING
#define RADMMM_TIMING 0
#endif
#if RADMMM_TIMING == 3
#define RADMMM_MFMA_F16(A, B, C) ([&] { asm volatile("" : : "v"(A), "v"(B)); return (C); }())
#else
#define RADMMM_MFMA_F16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

//   EK_GENERIC  LDS-parking epilogue, every descriptor option
//   EK_PLAIN    C                                   (bias, row factors, activation)
//   EK_SPLIT    C + split copy Ch / Cl (/ Clo)
//   EK_RES      C + C2 (+)= y (+ split copy C2h / C2l)     side input: C2 when accumulating
//   EK_DGRAD    C + split copy, y multiplied by act'(dact_src)   side input: dact_src
enum { EK_GENERIC = 0, EK_PLAIN, EK_SPLIT, EK_RES, EK_DGRAD, EK_COUNT };

// f(integral_constant<int, I>) for I = First, First + Step, ... < Last: loop indices that are constants INSIDE a lambda (an index
// that is a run-time parameter until inlining keeps a register array in scratch memory)
template <int I, int Last, int Step, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < Last) {
    f(std::integral_constant<int, I>{});
    static_for<I + Step, Last, Step>(f);
  }
}

template <int MB, int I>
__device__ __forceinline__ void take_block(const f32x16 (&acc)[MB][2], int sel, float (&v)[2][16]) {
  if constexpr (I < MB) {
    int s2 = sel;
    asm volatile("" : "+s"(s2));                      // opaque copy of the selector per case
    if (s2 == I) {
      asm volatile("; accumulators of row block %0" : : "n"(I));     // (a volatile statement cannot be if-converted)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[j][e] = acc[I][j][e];
    }
    take_block<MB, I + 1>(acc, sel, v);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* ptr, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, ptr ? (int)bytes : 0, 0x00020000);
}

// softplus on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each; radmmm::softplus_f's libm-style
// __logf costs ~12 instructions more per element): max(x, 0) + log1p(exp(-|x|)), log1p(e) = log(u) * e / (u - 1) with
// u = fl(1 + e) keeps full relative accuracy for small e.  u is in [1, 2]: no denormal handling needed around the log.
__device__ __forceinline__ float softplus_nb(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
  const float u = 1.f + e;
  const float d = u - 1.f;
  const float l = (0.693147180559945309f * __builtin_amdgcn_logf(u)) * (e * __builtin_amdgcn_rcpf(d));
  const float lp = (d == 0.f) ? e : l;
  // (no `x > 20 ? x : ...` shortcut: there e < 2.1e-9 < ulp(x) / 2, so u == 1, lp == e and x + e rounds to x -- the same bits,
  //  and without the select the compiler emits straight-line code instead of a divergent branch around the transcendentals,
  //  so that the chains of the 32 elements of a row block interleave: one wave per SIMD has nothing else to hide their latency)
  //  (also in the one-row-at-a-time loops: configs[4]'s 256-row tiles 96.19 / 96.19 -> 95.91 / 95.75 ms per step, no spills)
  return fmaxf(x, 0.f) + lp;
}
// split copy of one column pair (col even) of one row: hi fp16 pair, second array = fp16 lo pair (X8 false) or the 8-bit
// cross array (X8: format fmt = RADMMM_SPLIT_X8A / X8B through the offsets vXh / vXl), optional fp16 lo pair beside it.
// v* are per-lane byte offsets (out of range for columns >= N), sH the row's scalar byte offset (all arrays of a split
// copy have the same row pitch in bytes).  Returns max |scale * x|.
template <bool X8, bool has_lo16>
__device__ __forceinline__ float store_pair_split(__amdgpu_buffer_rsrc_t rH, __amdgpu_buffer_rsrc_t rL, __amdgpu_buffer_rsrc_t rLo16,
                                                  int vH, int vXh, int vXl, int sH, float x8_mul, float s, float y0, float y1) {
  const float u0 = y0 * s, u1 = y1 * s;
  const float amax = fmaxf(fabsf(u0), fabsf(u1));
  const float t0 = radmmm::clamp_f16(u0), t1 = radmmm::clamp_f16(u1);
  const _Float16 h0 = (_Float16)t0, h1 = (_Float16)t1;
  const float r0 = t0 - (float)h0, r1 = t1 - (float)h1;
  f16x2 hp;
  hp[0] = h0; hp[1] = h1;
  RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hp), rH, vH, sH, 0));
  if constexpr (!X8) {
    f16x2 lp;
    lp[0] = (_Float16)r0; lp[1] = (_Float16)r1;
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lp), rL, vH, sH, 0));
  } else {
    const float lm = x8_mul * 2048.f;
    const int w8h = __builtin_amdgcn_cvt_pk_fp8_f32(radmmm::clamp_e4m3(t0 * x8_mul), radmmm::clamp_e4m3(t1 * x8_mul), 0, false);
    const int w8l = __builtin_amdgcn_cvt_pk_fp8_f32(radmmm::clamp_e4m3(r0 * lm), radmmm::clamp_e4m3(r1 * lm), 0, false);
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b16((unsigned short)w8h, rL, vXh, sH, 0));
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b16((unsigned short)w8l, rL, vXl, sH, 0));
    if constexpr (has_lo16) {
      f16x2 lp;
      lp[0] = (_Float16)r0; lp[1] = (_Float16)r1;
      RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lp), rLo16, vH, sH, 0));
    }
  }
  return amax;
}

// the same with the optional fp16 lo part chosen at run time (the one-row-at-a-time epilogues: round 4's code, kept verbatim)
template <bool X8>
__device__ __forceinline__ float store_pair_split_rt(__amdgpu_buffer_rsrc_t rH, __amdgpu_buffer_rsrc_t rL, __amdgpu_buffer_rsrc_t rLo16,
                                                  bool has_lo16, int vH, int vXh, int vXl, int sH, float x8_mul, float s, float y0,
                                                  float y1) {
  const float u0 = y0 * s, u1 = y1 * s;
  const float amax = fmaxf(fabsf(u0), fabsf(u1));
  const float t0 = radmmm::clamp_f16(u0), t1 = radmmm::clamp_f16(u1);
  const _Float16 h0 = (_Float16)t0, h1 = (_Float16)t1;
  const float r0 = t0 - (float)h0, r1 = t1 - (float)h1;
  f16x2 hp;
  hp[0] = h0; hp[1] = h1;
  RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hp), rH, vH, sH, 0));
  if constexpr (!X8) {
    f16x2 lp;
    lp[0] = (_Float16)r0; lp[1] = (_Float16)r1;
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lp), rL, vH, sH, 0));
  } else {
    const float lm = x8_mul * 2048.f;
    const int w8h = __builtin_amdgcn_cvt_pk_fp8_f32(radmmm::clamp_e4m3(t0 * x8_mul), radmmm::clamp_e4m3(t1 * x8_mul), 0, false);
    const int w8l = __builtin_amdgcn_cvt_pk_fp8_f32(radmmm::clamp_e4m3(r0 * lm), radmmm::clamp_e4m3(r1 * lm), 0, false);
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b16((unsigned short)w8h, rL, vXh, sH, 0));
    RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b16((unsigned short)w8l, rL, vXl, sH, 0));
    if (has_lo16) {
      f16x2 lp;
      lp[0] = (_Float16)r0; lp[1] = (_Float16)r1;
      RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lp), rLo16, vH, sH, 0));
    }
  }
  return amax;
}

// rowf[r] = (acc_scale * pre, post, rowscale factor, -) of tile row r, see the caller:
//   x = (acc * pre + bias) * post;  x *= act'(dact_src) [EK_DGRAD];  x *= rowscale factor;  y = act(x)
// X8: the split copies are written in the 8-bit cross format (the FP8-cross scheme's kernels) / as fp16 pairs.
// ACTK / DACTK: 0 none, 1 softplus, 2 the descriptor's code at run time -- the caller branches ONCE per launch between
// the specialised copies of the row-block loop, so the per-element code has no activation switch.  DACTK 3 / 4 = 1 / 2 with
// the saved output read from its 8-bit split pair (radmmm_rowgemm_desc.dact_h / dact_x: a 4-byte load of the fp16 hi pair
// + a 2-byte load of the e4m3 lo pair per row and column pair instead of 8 bytes of fp32; y = hi + lo * 2^-(11 + e)).
// C == NULL (the split copy alone carries the result): its descriptor has size 0 and the hardware drops the stores.
template <int MB, int EK, bool X8, int ACTK, int DACTK_>
__device__ __forceinline__ void direct_blocks(const f32x16 (&acc)[MB][2], const float4* rowf, const radmmm_rowgemm_desc& p,
                                              int m0, int n0, int lane, int wave, float& sat) {
  constexpr bool DACT = EK == EK_DGRAD, C2M = EK == EK_RES, SPLIT = EK == EK_SPLIT || EK == EK_DGRAD;
  constexpr bool DPAIR = DACT && DACTK_ >= 3;
  constexpr int DACTK = DPAIR ? DACTK_ - 2 : DACTK_;
  constexpr bool SIDE = DACT || C2M;
  // rows per group of the element loop below: four under the FP8-cross scheme's forward kinds at tile heights <= 224 rows, else
  // one (round 4's loop, verbatim).  Measured / compiled (tools/spill_report.py): the data-gradient kinds spill 344 registers
  // with groups of four (+10 % per launch), the 256-row tiles 34-100, the f16-pair kinds 2-3.
  constexpr int GR = (DACT || MB >= 8 || !X8) ? 1 : 4;
  const int jj = lane & 31, h = lane >> 5;
  const int col = n0 + wave * 64 + 2 * jj;
  const bool cok = col + 1 < p.N;                               // N is even on this path
  float b0 = 0.f, b1 = 0.f;
  if (p.bias && cok) {
    b0 = p.bias[col];
    b1 = p.bias[col + 1];
  }
  const int act = p.act, dact = p.dact;
  const long long M = p.M;
  const __amdgpu_buffer_rsrc_t rC = rsrc_of(p.C, M * p.ldc * 4);
  const int vC = cok ? (4 * h * p.ldc + col) * 4 : OOB;
  // side input: dact_src (EK_DGRAD) or C2 when accumulating (EK_RES; a null descriptor reads as zeros)
  const void* side_ptr = DPAIR ? p.dact_h : (DACT ? (const void*)p.dact_src : ((C2M && p.c2_accum && p.n_c2_src <= 0) ? (const void*)p.C2 : nullptr));
  const int ldside = DPAIR ? p.lddact_h : (DACT ? p.lddact : p.ldc2);
  constexpr int SESZ = DPAIR ? 2 : 4;                            // bytes per element of the side array's rows
  const __amdgpu_buffer_rsrc_t rS = rsrc_of(SIDE ? side_ptr : nullptr, M * ldside * SESZ);
  const int vS = cok ? (4 * h * ldside + col) * SESZ : OOB;
  const __amdgpu_buffer_rsrc_t rSx = rsrc_of(DPAIR ? p.dact_x : nullptr, M * ldside * 2);          // the pair's cross array
  // (the lo8 pair of columns col, col + 1 is 2-byte aligned; it is fetched as the aligned 4-byte word around it and shifted
  //  down -- a 2-byte buffer load through the builtin was folded away by the compiler (ROCm 7.2: no buffer_load_ushort in the
  //  ISA, the hi pair's register was converted instead), found by tests/test_hip_round5.py)
  const unsigned xlo_off = radmmm::x8_lo_off(col, RADMMM_SPLIT_X8A);
  const int vSx = (DPAIR && cok) ? (int)(4 * h * ldside * 2 + (xlo_off & ~3u)) : OOB;
  const int xlo_sh = (int)(xlo_off & 2u) * 8;
  const float dp_lsc = __builtin_ldexpf(1.f, -(11 + p.dact_x8_exp));
  // (the pair's two words stay INTEGERS end to end: [0] the fp16 hi pair, [1] the aligned word around the e4m3 lo pair)
  auto load_side = [&](int row) __attribute__((always_inline)) {
    u32x2 r;
    if constexpr (DPAIR) {
      r[0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rS, vS, row * ldside * 2, 0);
      r[1] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rSx, vSx, row * ldside * 2, 0);
    } else {
      r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rS, vS, row * ldside * 4, 0));
    }
    return r;
  };
  // EK_RES with source arrays (radmmm_rowgemm_desc.c2_src): the side value is ((src0 + src1) + src2), summed when a block's
  // loads have landed; missing sources are null descriptors (zeros), C2 itself is not read
  const bool c2src = C2M && p.n_c2_src > 0;
  const __amdgpu_buffer_rsrc_t rQ0 = rsrc_of(c2src ? p.c2_src[0] : nullptr, M * p.ldc2 * 4);
  const __amdgpu_buffer_rsrc_t rQ1 = rsrc_of((c2src && p.n_c2_src > 1) ? p.c2_src[1] : nullptr, M * p.ldc2 * 4);
  const __amdgpu_buffer_rsrc_t rQ2 = rsrc_of((c2src && p.n_c2_src > 2) ? p.c2_src[2] : nullptr, M * p.ldc2 * 4);
  auto sum_src = [&](const u32x2& q0, const u32x2& q1, const u32x2& q2) __attribute__((always_inline)) {
    const f32x2 a = __builtin_bit_cast(f32x2, q0), b = __builtin_bit_cast(f32x2, q1), c = __builtin_bit_cast(f32x2, q2);
    f32x2 r;
    r[0] = (a[0] + b[0]) + c[0];
    r[1] = (a[1] + b[1]) + c[1];
    return __builtin_bit_cast(u32x2, r);
  };
  const __amdgpu_buffer_rsrc_t rC2 = rsrc_of(C2M ? p.C2 : nullptr, M * p.ldc2 * 4);
  const int vC2 = cok ? (4 * h * p.ldc2 + col) * 4 : OOB;
  // split outputs (of y: EK_SPLIT / EK_DGRAD; of C2: EK_RES, optional)
  const int fmt = p.split_fmt;
  const bool c2split = C2M && p.C2h != nullptr;
  const void* hp = SPLIT ? p.Ch : (c2split ? p.C2h : nullptr);
  const void* lp = SPLIT ? p.Cl : (c2split ? p.C2l : nullptr);
  const int ldh = SPLIT ? p.ldch : p.ldc2h;
  const bool has_lo16 = X8 && SPLIT && p.Clo != nullptr;
  const __amdgpu_buffer_rsrc_t rH = rsrc_of(hp, M * ldh * 2);
  const __amdgpu_buffer_rsrc_t rL = rsrc_of(lp, M * ldh * 2);
  const __amdgpu_buffer_rsrc_t rLo16 = rsrc_of(has_lo16 ? p.Clo : nullptr, M * ldh * 2);
  const int vH = cok ? (4 * h * ldh + col) * 2 : OOB;
  const int vXh = cok ? (int)(4 * h * ldh * 2 + radmmm::x8_hi_off(col, fmt)) : OOB;
  const int vXl = cok ? (int)(4 * h * ldh * 2 + radmmm::x8_lo_off(col, fmt)) : OOB;
  const float x8_mul = __builtin_ldexpf(1.f, SPLIT ? p.ch_x8_exp : p.c2h_x8_exp);
  const float sp_scale = SPLIT ? p.ch_scale : p.c2h_scale;

  auto row_of = [](int e) { return 8 * (e >> 2) + (e & 3); };     // + 4 h: tile row of accumulator element e
  auto actf = [&](float x) __attribute__((always_inline)) {
    if constexpr (ACTK == 0) return x;
    else if constexpr (ACTK == 1) return softplus_nb(x);
    else return radmmm::act_apply(x, act);
  };
  auto dactf = [&](float y) __attribute__((always_inline)) {
    if constexpr (DACTK == 1) {
      // softplus' from the output y >= 0 = radmmm::dact_from_out's value, without its branches: for y > 20 the exponential is
      // < 2^-25 and 1 - it rounds to the 1 the shortcut returns; both candidates of one_minus_exp_neg are evaluated (opaque to
      // the optimiser) and the choice is a v_cndmask -- the two elements of a column pair then run side by side instead of
      // two divergent branch sequences each (step A/B 40.57 / 40.44 / 40.48 -> 40.36 / 40.19 / 40.33 ms; same bits)
      float pl = y * (1.f - y * (0.5f - y * (0.16666667f - y * (0.041666668f - y * (0.0083333338f - y * 0.0013888889f)))));
      float q = 1.f - __expf(-y);
      asm("" : "+v"(pl), "+v"(q));
      return y < 0.25f ? pl : q;
    } else {
      return radmmm::dact_from_out(y, dact);
    }
  };
  u32x2 side[16], side_n[16];
  u32x2 q1_n[C2M ? 16 : 1], q2_n[C2M ? 16 : 1];                  // (EK_RES: the second and third source of the next block)
  if constexpr (SIDE) {
    if (c2src) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = (m0 + row_of(e)) * ldside * 4;
        side[e] = sum_src(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ0, vS, ro, 0)),
                          __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ1, vS, ro, 0)),
                          __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ2, vS, ro, 0)));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) side[e] = load_side(m0 + row_of(e));
    }
  }
  const int left = (p.M - m0 + 31) / 32;
  const int nblk = left < MB ? left : MB;
  // optional column sums of the value before its row scale, over the rows inside their utterance (include/radmmm_hip.h:
  // colsum_out): two running sums per lane, combined across the lane halves and written as this tile's partial row below
  const bool cs_on = p.colsum_scratch != nullptr;
  float cs0 = 0.f, cs1 = 0.f;
#pragma unroll 1
  for (int I = 0; I < nblk; ++I) {
    const int r0 = m0 + I * 32;
    if constexpr (SIDE) {                                          // next block's side inputs, ahead of this block's stores
      if (c2src) {
        if constexpr (C2M) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ro = (r0 + 32 + row_of(e)) * ldside * 4;
            side_n[e] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ0, vS, ro, 0));
            q1_n[e] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ1, vS, ro, 0));
            q2_n[e] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rQ2, vS, ro, 0));
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) side_n[e] = load_side(r0 + 32 + row_of(e));
      }
    }
    float4 rfs[16];                                                // the block's row factors (LDS), all requested up front
#pragma unroll
    for (int e = 0; e < 16; ++e) rfs[e] = rowf[I * 32 + row_of(e) + 4 * h];
    float v[2][16];
    take_block<MB, 0>(acc, I, v);
    // GR = four rows (eight elements) at a time.  Phase 1, the values: STRAIGHT-LINE code (no branch between the elements -- the
    // activation and its derivative are select-free, the column sums are accumulated with a 0 / 1 factor whether wanted or
    // not), so that the scheduler interleaves the eight dependent chains (v_exp -> add -> v_log -> v_rcp ...): with one wave
    // per SIMD nothing else hides a transcendental's latency.  Phase 2, the stores and split conversions of those rows, in
    // the launch-uniform variant (one scalar branch per group instead of one per element).  GR == 1: round 4's loop.
    if constexpr (GR == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ru = r0 + row_of(e);                               // uniform part of the row
        const float4 rf = rfs[e];
        float x0 = (v[0][e] * rf.x + b0) * rf.y, x1 = (v[1][e] * rf.x + b1) * rf.y;
        const f32x2 sidef = __builtin_bit_cast(f32x2, side[e]);
        if constexpr (DPAIR) {
          const f16x2 hp = __builtin_bit_cast(f16x2, side[e][0]);
          const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)(side[e][1] >> xlo_sh), false);
          x0 *= dactf(fmaf(lo[0], dp_lsc, (float)hp[0]));
          x1 *= dactf(fmaf(lo[1], dp_lsc, (float)hp[1]));
        } else if constexpr (DACT) {
          x0 *= dactf(sidef[0]);
          x1 *= dactf(sidef[1]);
        }
        if (cs_on) {                                                 // (uniform)
          const float m = (rf.z != 0.f && ru + 4 * h < p.M) ? 1.f : 0.f;
          cs0 = fmaf(m, x0, cs0);
          cs1 = fmaf(m, x1, cs1);
        }
        x0 = actf(x0 * rf.z);
        x1 = actf(x1 * rf.z);
        f32x2 y;
        y[0] = x0; y[1] = x1;
        RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y), rC, vC, ru * p.ldc * 4, 0));
        if constexpr (C2M) {
          f32x2 c2;
          c2[0] = sidef[0] + x0; c2[1] = sidef[1] + x1;              // (side reads as zero when not accumulating)
          RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c2), rC2, vC2, ru * p.ldc2 * 4, 0));
          if (c2split)
            sat = fmaxf(sat, store_pair_split_rt<X8>(rH, rL, rLo16, false, vH, vXh, vXl, ru * ldh * 2, x8_mul, sp_scale, c2[0], c2[1]));
        }
        if constexpr (SPLIT)
          sat = fmaxf(sat, store_pair_split_rt<X8>(rH, rL, rLo16, has_lo16, vH, vXh, vXl, ru * ldh * 2, x8_mul, sp_scale, x0, x1));
      }
    } else {
      auto store_rows = [&](auto E0, auto LO16, auto C2S) __attribute__((always_inline)) {
        constexpr int e0 = decltype(E0)::value;
#pragma unroll
        for (int e = e0; e < e0 + GR; ++e) {
          const int ru = r0 + row_of(e);
          const float x0 = v[0][e], x1 = v[1][e];
          f32x2 y;
          y[0] = x0; y[1] = x1;
          RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, y), rC, vC, ru * p.ldc * 4, 0));
          if constexpr (C2M) {
            const f32x2 sidef = __builtin_bit_cast(f32x2, side[e]);
            f32x2 c2;
            c2[0] = sidef[0] + x0; c2[1] = sidef[1] + x1;            // (side reads as zero when not accumulating)
            RADMMM_EPI_STORE(__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c2), rC2, vC2, ru * p.ldc2 * 4, 0));
            if constexpr (decltype(C2S)::value)
              sat = fmaxf(sat, store_pair_split<X8, false>(rH, rL, rLo16, vH, vXh, vXl, ru * ldh * 2, x8_mul, sp_scale, c2[0], c2[1]));
          }
          if constexpr (SPLIT)
            sat = fmaxf(sat, store_pair_split<X8, decltype(LO16)::value>(rH, rL, rLo16, vH, vXh, vXl, ru * ldh * 2, x8_mul, sp_scale, x0, x1));
        }
      };
      static_for<0, 16, GR>([&](auto E0) __attribute__((always_inline)) {
        constexpr int e0 = decltype(E0)::value;
#pragma unroll
        for (int e = e0; e < e0 + GR; ++e) {
          const int ru = r0 + row_of(e);                             // uniform part of the row
          const float4 rf = rfs[e];
          float x0 = (v[0][e] * rf.x + b0) * rf.y, x1 = (v[1][e] * rf.x + b1) * rf.y;
          if constexpr (DPAIR) {
            const f16x2 hp = __builtin_bit_cast(f16x2, side[e][0]);
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)(side[e][1] >> xlo_sh), false);
            x0 *= dactf(fmaf(lo[0], dp_lsc, (float)hp[0]));
            x1 *= dactf(fmaf(lo[1], dp_lsc, (float)hp[1]));
          } else if constexpr (DACT) {
            const f32x2 sidef = __builtin_bit_cast(f32x2, side[e]);
            x0 *= dactf(sidef[0]);
            x1 *= dactf(sidef[1]);
          }
          {
            const float m = (cs_on && rf.z != 0.f && ru + 4 * h < p.M) ? 1.f : 0.f;
            cs0 = fmaf(m, x0, cs0);
            cs1 = fmaf(m, x1, cs1);
          }
          v[0][e] = actf(x0 * rf.z);
          v[1][e] = actf(x1 * rf.z);
        }
        if constexpr (C2M) {
          if (c2split) store_rows(E0, std::false_type{}, std::true_type{});
          else store_rows(E0, std::false_type{}, std::false_type{});
        } else if constexpr (SPLIT && X8) {
          if (has_lo16) store_rows(E0, std::true_type{}, std::false_type{});
          else store_rows(E0, std::false_type{}, std::false_type{});
        } else {
          store_rows(E0, std::false_type{}, std::false_type{});
        }
      });
    }
    if constexpr (SIDE) {
      if (c2src) {
        if constexpr (C2M) {
#pragma unroll
          for (int e = 0; e < 16; ++e) side[e] = sum_src(side_n[e], q1_n[e], q2_n[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) side[e] = side_n[e];
      }
    }
  }
  if (cs_on) {
    cs0 += __shfl_xor(cs0, 32);
    cs1 += __shfl_xor(cs1, 32);
    if (h == 0 && cok) {
      f32x2 o;
      o[0] = cs0; o[1] = cs1;
      // partial row of this row tile, pitch N (radmmm_colsum_final's layout); col is even and N even: 8-byte aligned
      *reinterpret_cast<f32x2*>(p.colsum_scratch + (long long)(m0 / (32 * MB)) * p.N + col) = o;
    }
  }
}

template <int MB, int EK, bool X8>
__device__ __forceinline__ void direct_epilogue(const f32x16 (&acc)[MB][2], const float4* rowf, const radmmm_rowgemm_desc& p,
                                                int m0, int n0, int lane, int wave, float& sat) {
  if constexpr (EK == EK_DGRAD) {                      // (host: act none on this kind)
    if constexpr (X8) {                                // the saved output as its 8-bit split pair (dact_h / dact_x)
      if (p.dact_h) {
        if (p.dact == RADMMM_ACT_SOFTPLUS) direct_blocks<MB, EK, X8, 0, 3>(acc, rowf, p, m0, n0, lane, wave, sat);
        else direct_blocks<MB, EK, X8, 0, 4>(acc, rowf, p, m0, n0, lane, wave, sat);
        return;
      }
    }
    if (p.dact == RADMMM_ACT_SOFTPLUS) direct_blocks<MB, EK, X8, 0, 1>(acc, rowf, p, m0, n0, lane, wave, sat);
    else direct_blocks<MB, EK, X8, 0, 2>(acc, rowf, p, m0, n0, lane, wave, sat);
  } else {
    if (p.act == RADMMM_ACT_SOFTPLUS) direct_blocks<MB, EK, X8, 1, 0>(acc, rowf, p, m0, n0, lane, wave, sat);
    else if (p.act == RADMMM_ACT_NONE) direct_blocks<MB, EK, X8, 0, 0>(acc, rowf, p, m0, n0, lane, wave, sat);
    else direct_blocks<MB, EK, X8, 2, 0>(acc, rowf, p, m0, n0, lane, wave, sat);
  }
}

// instruction-order pinning of the K step (sched_group_barrier masks); without it the scheduler keeps a single
// ds_read in flight and every MFMA group waits for LDS (measured: 39 % MFMA busy inside a workgroup)
constexpr int SGB_VMEM = 0x020, SGB_MFMA = 0x008, SGB_DSR = 0x100;
constexpr int LOOKAHEAD = 2;    // fragment look-ahead in pipeline items
#ifndef RADMMM_DPI
#define RADMMM_DPI 2
#endif
constexpr int DPI = RADMMM_DPI;   // DMA pieces issued per pipeline item

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

// the same with a wave-uniform byte offset in the instruction's SCALAR offset operand (no vector add per piece; the
// scalar offset takes part in the range check on gfx950, so an out-of-range vector offset stays out of range)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);
#endif
}

template <int MB, int PR, int EK>
__global__ __launch_bounds__(256, 1) void rowgemm_h3d_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes,
                                                              const int b_bytes) {
  using G = Geo<MB>;
  RADMMM_PHASE(0);
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
      a_base[k] = ((b * p.T + a_t[k]) * q.lda_h + d_chunk * 8) * 2;      // this lane's row at shift 0
    }
    a_dst[k] = real ? a_isl[k] * G::A_BYTES + j * 1024 : -1;
    a_vo[k] = OOB;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * k + wave;
    // B rows are INTERLEAVED between a wave's two 32-column blocks: LDS row (column group cg, block jb, jj) of the tile
    // holds output column n0 + 64 cg + 2 jj + jb, so that lane jj of a wave owns the ADJACENT columns 2 jj, 2 jj + 1 in
    // its two accumulators and the direct epilogue stores 8-byte pairs (256 contiguous bytes per row and half wave)
    const int lr = 16 * j + d_row;                                // LDS row 0 .. 255
    const int n = n0 + (lr & ~63) + 2 * (lr & 31) + ((lr >> 5) & 1);
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
    const int sb = s * q.lda_h * 2 + (ex ? extra_bytes : 0);      // uniform byte offset of the tap (no per-lane multiply)
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
      const int ts = a_t[k] + s;
      const int ok = -(int)((ts >= 0) & (ts < a_lim[k]));        // all ones when the frame is readable
      a_vo[k] = ((a_base[k] + sb) & ok) | (OOB & ~ok);
    }
  };
  // piece w of 0 .. NP-1 of tile (tap, kb) into stage `buf`
  auto dma_piece = [&](int buf, int w, int tap, int kb) __attribute__((always_inline)) {
#ifdef RADMMM_ABL_NODMA                                // measurement builds: K loop without operand delivery
    if (buf >= 0) return;
#endif
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
#ifdef RADMMM_ABL_NOMFMA                               // measurement builds: operand delivery alone
    if (t >= 0) return;
#endif
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
  RADMMM_PHASE(1);
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
  RADMMM_PHASE(2);

  const radmmm::EpilogueCtx ec(p);
  float sat = 0.f;
  if constexpr (EK != EK_GENERIC) {                    // direct epilogue: per-row factors into LDS, then wave-private
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
  } else {
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
    epilogue_blocks<MB, 0>(acc, smf, rowf, p, ec, q.acc_scale, m0, n0, tid, lane, wave, biasv, sat);
  }
  {
    // (8-bit outputs: Ch with ch_x8_exp, C2h with c2h_x8_exp; one bound for both = the larger multiplier)
    const int xe = p.Ch ? (p.C2h && p.c2h_x8_exp > p.ch_x8_exp ? p.c2h_x8_exp : p.ch_x8_exp) : p.c2h_x8_exp;
    radmmm::raise_sat_flag(p.sat_flag, sat, ((p.Ch || p.C2h) && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, xe) : 0.f);
  }
  RADMMM_PHASE(3);
}

template <int MB, int PR, int EK>
int launch_dma(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = Geo<MB>;
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_h3d_kernel<MB, PR, EK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_h3d<%d,%d,%d>): %s", MB, PR, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_h3d_kernel<MB, PR, EK>), dim3(ntm * ntn), dim3(256), G::SMEM, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_h3d");
}

template <int MB, int PR>
int launch_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  switch (ek) {
    case EK_PLAIN: return launch_dma<MB, PR, EK_PLAIN>(d, stream, a_bytes, b_bytes);
    case EK_SPLIT: return launch_dma<MB, PR, EK_SPLIT>(d, stream, a_bytes, b_bytes);
    case EK_RES: return launch_dma<MB, PR, EK_RES>(d, stream, a_bytes, b_bytes);
    case EK_DGRAD: return launch_dma<MB, PR, EK_DGRAD>(d, stream, a_bytes, b_bytes);
    default: return launch_dma<MB, PR, EK_GENERIC>(d, stream, a_bytes, b_bytes);
  }
}

// every (row blocks per workgroup, epilogue kind) instantiation of one product scheme
template <int PR>
int launch_pr(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
#ifdef RADMMM_QUICK                                    // development builds: one tile height only
  return launch_ek<7, PR>(ek, d, stream, a_bytes, b_bytes);
#else
  switch (mb) {
    case 4: return launch_ek<4, PR>(ek, d, stream, a_bytes, b_bytes);
    case 5: return launch_ek<5, PR>(ek, d, stream, a_bytes, b_bytes);
    case 6: return launch_ek<6, PR>(ek, d, stream, a_bytes, b_bytes);
    case 7: return launch_ek<7, PR>(ek, d, stream, a_bytes, b_bytes);
    default: return launch_ek<8, PR>(ek, d, stream, a_bytes, b_bytes);
  }
#endif
}

}  // namespace
