// rowgemm_onetap.h: the K loop of a ONE-TAP segment of the wide split conv GEMM under the FP8-cross scheme (round 4):
//     acc[i][j] += A[rows of the tile][k] . B[columns of the wave][k]      over kpt K steps of 32 channels,
// for a workgroup of 4 waves with (32 MB) x 256 tiles, wave w = columns 64 w .. 64 w + 63, MB x 2 accumulators of 32 x 32
// (rowgemm_h3w_kernel.h's accumulator layout and B-row interleave: the direct epilogue applies unchanged).  Used by
// rowgemm_one.hip (1x1 convs) and by the extra K segment of the shared-window kernel (rowgemm_win.hip).
//
// What differs from rowgemm_h3d's loop (two stages holding A and B, one barrier per step that waited for DMA issued in the
// same step, compiler-ordered items with address arithmetic in them):
//   * A lives in a ring of THREE stages, B is wave-private in two: tile s + 2 of both is issued during step s, so every
//     piece has more than a K step of flight time before anybody waits for it (LDS-DMA latency under load ~ 1 us ~ a step);
//     the one barrier per step stands behind the wave's own counted vmcnt and publishes A(s + 1) only;
//   * the step is a sequence of SLOTS closed by sched_barrier(0) -- {f16 MFMA | one fragment read} x 2, {fp8 MFMA | DMA
//     piece(s)} -- because one wave per SIMD issues everything in order and whatever stands between two MFMAs must fit under
//     the first one's 32 / 64 cycles (rowgemm_win.hip has the measurements); no vector arithmetic besides one add per read;
//   * the fragment look-ahead (2 items) and the B fragments of step s + 1 cross the step boundary: they are read under the
//     last two items of step s, behind the barrier.
// Same MFMAs in the same order per accumulator as rowgemm_h3d<MB, 2, *>: bit-identical results.
#pragma once
#include "rowgemm_h3w_kernel.h"

namespace {

template <int MB>
struct OneGeo {
  static constexpr int BMR = MB * 32;
  static constexpr int B_BYTES = BN * ROWB;          // one of {Bh, Bl}
  static constexpr int B_STAGE = 2 * B_BYTES;
  static constexpr int A_BASE = 2 * B_STAGE;         // B stages first (64 KiB), the A ring behind them
  static constexpr int A_PLANE = BMR * ROWB;         // one of {Ah, Al}
  static constexpr int A_STAGE = 2 * A_PLANE;
  static constexpr int SMEM = A_BASE + 3 * A_STAGE;  // MB = 8: exactly 160 KiB
  static constexpr int NP = MB + 8;                  // DMA pieces per wave and step: MB of A (4 MB groups of 16 rows over 4 waves), 8 of B
  static_assert(SMEM <= 160 * 1024, "LDS");
};

// DMA pieces of tile `rel` (relative K step) -- the caller's lambdas:  dma_a(k, stage, soff)  k = 0 .. MB-1,
// dma_b(w, stage, soff)  w = 0 .. 7;  soff = rel * 64 bytes (the K position as the instruction's scalar offset).
// PR = product scheme: 2 (FP8 cross terms; Al / Bl = 8-bit cross arrays) or 3 (three f16 products; Al / Bl = fp16 lo arrays of
// the same row pitch: the DMA pieces, the swizzle and the fragment reads are identical, only the slot-C MFMAs differ -- the one
// block-scaled FP8 MFMA of a (row block, column block) becomes Al.Bh + Ah.Bl for both k blocks: four f16 MFMAs)
template <int MB, int PR = 2, class DmaA, class DmaB>
__device__ __forceinline__ void one_tap_steps(f32x16 (&acc)[MB][2], unsigned char* sm, const int kpt, const int lane, const int wave,
                                              const int x_sa, const int x_sb, DmaA dma_a, DmaB dma_b) {
  using G = OneGeo<MB>;
  constexpr int NT = 2 * MB, D = 2, NP = G::NP, TW = NT - 2;
  // piece p of a step is issued in C slot (p * NT) / NP  (C slot c closes item c + 1; slot NT - 1 follows the last item):
  // before the step's wait (in front of item TW) the slots 0 .. TW - 2 have been issued
  constexpr auto slot_of = [](int p) { return p * NT / NP; };
  constexpr int issued_before_wait = [] {
    int n = 0;
    for (int p = 0; p < NP; ++p) n += (p * NT / NP) <= TW - 2 ? 1 : 0;
    return n;
  }();
  auto dma_piece = [&](int p, int stage_a, int stage_b, int soff) __attribute__((always_inline)) {
    if (p < MB) dma_a(p, stage_a, soff);
    else dma_b(p - MB, stage_b, soff);
  };
  const int half = lane >> 5;
  int aad0[MB], aad1[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int w = 32 * i + (lane & 31);
    aad0[i] = G::A_BASE + (w << 6) + (((half ^ (w >> 2)) & 3) << 4);
    aad1[i] = aad0[i] ^ 32;
  }
  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3;
  const int bad0 = wave * 64 * ROWB + f_row + (((0 + half) ^ f_swz) << 4);
  const int bad1 = wave * 64 * ROWB + f_row + (((2 + half) ^ f_swz) << 4);
  f16x8 fah[NT], fal[NT], bh[2][2][2], bl[2][2][2];                   // B fragments: [register set][k block][column block]
  auto read_hi = [&](int t, int a_off) __attribute__((always_inline)) {
    fah[t] = *reinterpret_cast<const f16x8*>(sm + ((t & 1) ? aad1 : aad0)[t >> 1] + a_off);
  };
  auto read_lo = [&](int t, int a_off) __attribute__((always_inline)) {
#if RADMMM_TIMING != 2
    fal[t] = *reinterpret_cast<const f16x8*>(sm + ((t & 1) ? aad1 : aad0)[t >> 1] + a_off + G::A_PLANE);
#endif
  };
  auto read_b1 = [&](int set, int kb, int j) __attribute__((always_inline)) {      // B stage = register set
    const int fo = (kb ? bad1 : bad0) + set * G::B_STAGE + j * 32 * ROWB;
    bh[set][kb][j] = *reinterpret_cast<const f16x8*>(sm + fo);
#if RADMMM_TIMING != 2
    bl[set][kb][j] = *reinterpret_cast<const f16x8*>(sm + fo + G::B_BYTES);
#endif
  };
  auto cross = [&](int set, int i, int j) __attribute__((always_inline)) {
    if constexpr (PR == 3) {
#if RADMMM_TIMING == 0
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[2 * i + kb], bh[set][kb][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[2 * i + kb], bl[set][kb][j], acc[i][j], 0, 0, 0);
      }
#endif
      return;
    }
    const i32x8 a8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, fal[2 * i]), __builtin_bit_cast(i32x4, fal[2 * i + 1]),
                                             0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 b8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, bl[set][0][j]), __builtin_bit_cast(i32x4, bl[set][1][j]),
                                             0, 1, 2, 3, 4, 5, 6, 7);
#if RADMMM_TIMING == 0
    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], RADMMM_X_FMT, RADMMM_X_FMT, 0, x_sa, 0, x_sb);
#elif RADMMM_TIMING != 2
    asm volatile("" : : "v"(a8), "v"(b8));
#endif
  };

  // prologue: tiles 0 and 1 (A stages 0 / 1, B stages 0 / 1); tile 0 must have landed everywhere, tile 1 may still fly
#pragma unroll
  for (int p = 0; p < NP; ++p) dma_piece(p, 0, 0, 0);
#pragma unroll
  for (int p = 0; p < NP; ++p) dma_piece(p, 1, 1, BK * 2);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(NP) : "memory");
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int j = 0; j < 2; ++j) read_b1(0, kb, j);
#pragma unroll
  for (int t = 0; t < D; ++t) {
    read_hi(t, 0);
    read_lo(t, 0);
  }
  auto step = [&](auto setc, const int a_cur, const int a_nxt, const int a_st2, const int soff2) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;                        // step parity: B register set and B stage of this step
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int kbk = t & 1, i = t >> 1;
      if (t == TW) {
        // everything older than this step's pieces has landed (this wave's share of A(s + 1), all of its B(s + 1)); every
        // read of A(s) has been issued and is waited for: the barrier publishes A(s + 1) and frees A(s)'s stage for s + 3
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(issued_before_wait) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      // slot A
      acc[i][0] = RADMMM_MFMA_F16(fah[t], bh[set][kbk][0], acc[i][0]);
      if (t + D < NT) read_hi(t + D, a_cur);
      else read_hi(t + D - NT, a_nxt);
      if (t >= TW) read_b1(set ^ 1, t - TW, 0);
      __builtin_amdgcn_sched_barrier(0);
      // slot B
      acc[i][1] = RADMMM_MFMA_F16(fah[t], bh[set][kbk][1], acc[i][1]);
      if (t + D < NT) read_lo(t + D, a_cur);
      else read_lo(t + D - NT, a_nxt);
      if (t >= TW) read_b1(set ^ 1, t - TW, 1);
      __builtin_amdgcn_sched_barrier(0);
      // slot C (closes items 1 .. NT - 1; the last one follows the loop)
      if (t > 0) {
        if (kbk == 1) cross(set, i, 0);
        else cross(set, i - 1, 1);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          if (slot_of(p) == t - 1) dma_piece(p, a_st2, set, soff2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cross(set, MB - 1, 1);
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (slot_of(p) == NT - 1) dma_piece(p, a_st2, set, soff2);
    __builtin_amdgcn_sched_barrier(0);
  };
  // stage of A(s) = s mod 3, kept as byte offsets (cur, next, the one being filled)
  int o_cur = 0, o_nxt = G::A_STAGE, o_st2 = 2 * G::A_STAGE, st2 = 2;
  for (int s = 0; s < kpt; s += 2) {
    step(std::integral_constant<int, 0>{}, o_cur, o_nxt, st2, (s + 2) * (BK * 2));
    { const int t = o_cur; o_cur = o_nxt; o_nxt = o_st2; o_st2 = t; st2 = st2 == 2 ? 0 : st2 + 1; }
    step(std::integral_constant<int, 1>{}, o_cur, o_nxt, st2, (s + 3) * (BK * 2));
    { const int t = o_cur; o_cur = o_nxt; o_nxt = o_st2; o_st2 = t; st2 = st2 == 2 ? 0 : st2 + 1; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // stray look-ahead reads / DMA past the last tile
}

}  // namespace
