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
// Scope: taps = 5, dilation <= 8, FP8-cross scheme, epilogue kinds PLAIN / SPLIT / DGRAD (the latter also with the extra K
// segment of the fused data gradient), MB 7 / 8
// (rowgemm_h3w.hip decides; everything else keeps rowgemm_h3d).
#include <type_traits>

#include "rowgemm_onetap.h"

namespace {

constexpr int WTAPS = 5, WDMAX = 8;
#ifndef RADMMM_WIN_LOOKAHEAD
#define RADMMM_WIN_LOOKAHEAD 2
#endif
constexpr int WLOOK = RADMMM_WIN_LOOKAHEAD;          // fragment look-ahead in pipeline items (measured: 2 / 3 / 4)

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
  static constexpr int WPT = (NPW + 2) / 3;         // ... per K step, all in the first three taps of the slice before
  static constexpr int nw(int tap) { return tap >= 3 ? 0 : (NPW - WPT * tap < WPT ? (NPW - WPT * tap > 0 ? NPW - WPT * tap : 0) : WPT); }
  static constexpr int bstart(int tap) { return nw(tap) > 3 ? nw(tap) : 3; }   // first item that may overwrite the B stage
  static_assert(W_BASE % 1024 == 0 && W_BYTES % 1024 == 0 && SMEM <= 160 * 1024, "LDS map");
};

template <int MB, int TAP, int T>
__device__ __forceinline__ void pin_items_win() {
  using G = WGeo<MB>;
  constexpr int NT = 2 * MB;
  if constexpr (T < NT - WLOOK) {
    __builtin_amdgcn_sched_group_barrier(SGB_DSR, 2, 0);
    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, T == 0 ? 2 : 3, 0);
    if constexpr (T < G::nw(TAP) || (T >= G::bstart(TAP) && T < G::bstart(TAP) + 8)) __builtin_amdgcn_sched_group_barrier(SGB_VMEM, 1, 0);
    pin_items_win<MB, TAP, T + 1>();
  }
}

// one K step (compile-time tap) -- see the kernel
template <int MB, int EK, bool XT, int PR = 2>
__global__ __launch_bounds__(256, 1) void rowgemm_win_kernel(const radmmm_rowgemm_h3_desc q, const int a_bytes, const int b_bytes) {
  using G = WGeo<MB>;
  constexpr int NT = 2 * MB, D = WLOOK, NPW = G::NPW;
  static_assert(NT - D >= G::bstart(0) + 8, "DMA slots of a step");
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const radmmm_rowgemm_desc& p = q.base;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + G::BMR - 1) / G::BMR;
  const int nt = ntn * ntm, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;   // XCD x runs a contiguous run of the tile sequence
  // Tile sequence: column tiles in PAIRS, row tiles inside a pair -- an XCD's run then covers ~ntm/4 row tiles of ONE pair
  // of column tiles at N = 1024: its L2 fetches half of the weights and a quarter of A (8 x (10.5 + 13) MB instead of
  // 8 x (21 + 6.5) MB per launch with all four column tiles of a row tile in sequence).
  const int per_pair = 2 * ntm, pr = tile / per_pair, rr = tile - pr * per_pair;
  const int gw = (ntn - 2 * pr) < 2 ? (ntn - 2 * pr) : 2;            // column tiles in this pair (1 for the last one of an odd ntn)
  const int tm = gw == 2 ? (rr >> 1) : rr, tn = 2 * pr + (gw == 2 ? (rr & 1) : 0);
  const int m0 = tm * G::BMR, n0 = tn * BN;
  const int kpt = p.K / BK;
  const int dil = p.dil, sg = p.sign;
  const int b0 = m0 / p.T, t0 = m0 - b0 * p.T;
  const int nb = (p.T - t0) < G::BMR ? (p.T - t0) : G::BMR;          // tile rows of utterance b0
  const int seg1 = nb + 4 * dil;                                      // first LDS row of the second segment

  // ---- DMA setup.  This lane's row within a 16-row piece and its (source-side swizzled) chunk:
  const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int w_vo[NPW], w_dst[NPW], w_isl[NPW], b_voff[4], b_dst[4];
  // readable frames of the (at most) two utterances of this tile: two loads, issued together, before anything depends on them
  const int nutt = p.M / p.T;
  const bool masked = p.a_mask_mode && p.lens;
  const int lim0 = masked ? p.lens[b0] : p.T;
  const int lim1 = b0 + 1 < nutt ? (masked ? p.lens[b0 + 1] : p.T) : 0;
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
    const bool used = c < 2 * G::WP && (s1 ? (nb < G::BMR && wr < G::BMR + 8 * dil) : true);
    const int lim = used ? (s1 ? lim1 : lim0) : 0;
    w_vo[k] = (f >= 0 && f < lim) ? ((b * p.T + f) * q.lda_h + d_chunk * 8) * 2 : OOB;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = 4 * wave + k;                                       // THIS wave's 64 B rows: nobody else reads them
    const int lr = 16 * j + d_row;                                    // LDS row 0 .. 255, interleaved as in rowgemm_h3d
    const int n = n0 + (lr & ~63) + 2 * (lr & 31) + ((lr >> 5) & 1);
    b_voff[k] = n < p.N ? (n * q.ldb_h + d_chunk * 8) * 2 : OOB;
#ifdef RADMMM_TIMING_BOOB      // timing only (wrong results): 1 = every second B piece, 2 = every B piece is an out-of-range
    if (RADMMM_TIMING_BOOB == 2 || (k & 1)) b_voff[k] = OOB;   // DMA -- zeros written to LDS, nothing fetched from L2
#endif
    b_dst[k] = j * 1024;
  }
  const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Ah), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Al), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bh), 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.Bl), 0, b_bytes, 0x00020000);
  f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- fragment addressing: NO arithmetic in the K loop.  The byte address of (row block i, tap) -- window 0, hi plane,
  // k block 0 / 1 -- is computed once per tile (2 x 35 registers); the window parity and the cross plane are immediate
  // offsets of the ds_read (the parity moves a row by 2 WR = 576 rows: the XOR swizzle, which follows (row >> 2) & 3, does
  // not change), the k block flips bit 5.
  const int half = lane >> 5;
  static_assert((2 * G::WR) % 16 == 0 && G::W_BYTES + G::W_PLANE < 65536, "window parity / plane as ds_read immediates");
  int aad0[MB][WTAPS], aad1[MB][WTAPS];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int ri = 32 * i + (lane & 31);
    const int w0 = G::W_BASE / ROWB + ri + 2 * dil + (ri >= nb ? 4 * dil : 0);
#pragma unroll
    for (int tp = 0; tp < WTAPS; ++tp) {
      const int w = w0 + sg * (tp - WTAPS / 2) * dil;
      aad0[i][tp] = (w << 6) + (((half ^ (w >> 2)) & 3) << 4);
      aad1[i][tp] = aad0[i][tp] ^ 32;
    }
  }
  const int f_row = (lane & 31) * ROWB, f_swz = (lane >> 2) & 3;
  const int bad0 = wave * 64 * ROWB + f_row + (((0 + half) ^ f_swz) << 4);   // B tiles: fixed rows, as in rowgemm_h3d
  const int bad1 = wave * 64 * ROWB + f_row + (((2 + half) ^ f_swz) << 4);

  f16x8 fah[NT], fal[NT], bh[2][2][2], bl[2][2][2];                   // B fragments: [register set][k block][column block]
  auto read_hi = [&](int t, auto tapc, auto parc) __attribute__((always_inline)) {        // item t = 2 i + kblock
    constexpr int tap = decltype(tapc)::value, par = decltype(parc)::value;
    fah[t] = *reinterpret_cast<const f16x8*>(sm + ((t & 1) ? aad1 : aad0)[t >> 1][tap] + par * G::W_BYTES);
  };
  auto read_lo = [&](int t, auto tapc, auto parc) __attribute__((always_inline)) {
    constexpr int tap = decltype(tapc)::value, par = decltype(parc)::value;
#if RADMMM_TIMING == 2
    (void)t;
#elif defined(RADMMM_SKIP_LO_READS)         // TIMING-ONLY build (wrong results): the cross-term MFMAs run on the hi fragments -- what do
    fal[t] = fah[t];                        // the 14 lo-fragment LDS reads per wave and K step cost a power-bound launch?
#else
#ifdef RADMMM_WIN_SKIP6       // TIMING-ONLY build (wrong results): 6 of the 36 fragment reads per wave and K step are not
    if ((t & 1) == 0 && t < 12) {   // issued -- the read count of a 2 x 2 wave layout on 16 x 16 MFMAs (30 per wave), same MFMAs, same DMA
      fal[t] = fah[t];              // (profiles/r06_tile_probes.txt)
      return;
    }
#endif
    fal[t] = *reinterpret_cast<const f16x8*>(sm + ((t & 1) ? aad1 : aad0)[t >> 1][tap] + par * G::W_BYTES + G::W_PLANE);
#endif
  };
  auto read_b1 = [&](int set, int stage, int kb, int j) __attribute__((always_inline)) {
    const int fo = kb ? bad1 : bad0;
    bh[set][kb][j] = *reinterpret_cast<const f16x8*>(sm + stage * G::B_STAGE + j * 32 * ROWB + fo);
#if RADMMM_TIMING != 2
    bl[set][kb][j] = *reinterpret_cast<const f16x8*>(sm + stage * G::B_STAGE + G::B_BYTES + j * 32 * ROWB + fo);
#endif
  };
#ifdef RADMMM_B_GLOBAL
  // EXPERIMENT (round 5, -DRADMMM_B_GLOBAL; MEASURED AND REJECTED, profiles/r05_nprod1_floor.txt): the wave-private B fragments
  // straight from global memory into registers -- no LDS-DMA pieces and no ds_reads for B: 32 KB of LDS writes and 32 KB of LDS
  // reads per CU and K step less on a port that is 81 % busy.  Bit-identical (38 shared-window cases, the step's loss), but the
  // launch gets SLOWER: 5-tap forward 255 -> 282 us, fused data gradient 316 -> 353 us, step 41.9 -> 43.5 ms -- a fragment load
  // touches 32 weight rows with 32 bytes each (half cache lines through the texture path), which costs more than the LDS port
  // it relieves; the LDS-DMA path moves the same bytes as 16 rows x 64 bytes per instruction.  Lane
  // (n = lane & 31, half) of column block j wants the 16 bytes of weight row n0 + 64 wave + 2 n + j (the interleaved order of
  // the direct epilogue) at k block kb: chunk 2 kb + half of the 64-byte K step.
  int gb_vo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wave * 64 + 2 * (lane & 31) + j;
    gb_vo[j] = n < p.N ? (n * q.ldb_h + half * 8) * 2 : OOB;
  }
  auto load_b_piece = [&](int w, int set, int soff) __attribute__((always_inline)) {      // w = 4 arr + 2 kb + j
    const int arr = w >> 2, kb = (w >> 1) & 1, j = w & 1;
    const f16x8 v = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(arr ? rBl : rBh, gb_vo[j] + kb * 32, soff, 0));
    if (arr) bl[set][kb][j] = v;
    else bh[set][kb][j] = v;
  };
#endif
  const int x_sa = (lane >> 5) ? 127 - 11 - q.a8_exp : 127 - q.a8_exp;       // E8M0 block scales (rowgemm_h3d, PR 2)
  const int x_sb = (lane >> 5) ? 127 - q.b8_exp : 127 - 11 - q.b8_exp;
  auto cross = [&](int set, int i, int j) __attribute__((always_inline)) {
    if constexpr (PR == 3) {                // three f16 products (rowgemm_onetap.h): Al.Bh + Ah.Bl of both k blocks
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
  // DMA with the step's position as the instruction's SCALAR offset (no vector add per piece; an out-of-range vector offset
  // stays out of range: the scalar offset takes part in the range check on gfx950)
  auto dma_win2 = [&](int k, int par, int kb) __attribute__((always_inline)) {
    const int dst = w_dst[k] < 0 ? G::DUMP + wave * 1024 : G::W_BASE + par * G::W_BYTES + w_dst[k];
    dma16s(w_isl[k] ? rAl : rAh, (lds_u32_ptr)(sm + dst), w_vo[k], kb * (BK * 2));
  };
  auto dma_b2 = [&](int w, int buf, int soff) __attribute__((always_inline)) {
    const int k = w & 3, arr = w >> 2;
    dma16s(arr == 0 ? rBh : rBl, (lds_u32_ptr)(sm + buf * G::B_STAGE + b_dst[k] + arr * G::B_BYTES), b_voff[k], soff);
  };
  const int b_tap_bytes = (int)(q.b_tap_stride_h * 2);

  // ---- K loop (round 4).  One wave per SIMD issues EVERYTHING in order, so whatever stands between two MFMAs must fit
  // under the first one's 32 (f16) / 64 (fp8) cycles or the matrix pipe idles (tools/issue_cost_probe.hip: <= 4 light
  // instructions or one LDS / DMA instruction per f16 MFMA).  Round 3's sched_group_barrier pins left bursts of 28 reads +
  // 11 DMA pieces without an MFMA in three of the five steps of a slice; here the step is a sequence of SLOTS, each closed
  // by a sched_barrier(0) the scheduler cannot move anything across:
  //     slot A: MFMA acc[i][0] (f16) | hi fragment of item t + 2              (+ 2 B reads in the step's last two items)
  //     slot B: MFMA acc[i][1] (f16) | cross fragment of item t + 2           (+ 2 B reads ...)
  //     slot C: cross MFMA (fp8, 64 cycles) | one DMA piece
  // Who waits for what:
  //   B tile: wave-private rows.  B(s + 1) is read into the OTHER register set during the last two items of step s, behind
  //     this wave's own vmcnt (in-order counter: everything but the step's own pieces has landed), so the first MFMA of
  //     step s + 1 finds its operands in registers; the stage is free for B(s + 3)... i.e. B(s + 2) goes into stage s & 1
  //     from the first slot of step s on: two K steps of flight time.
  //   A window of slice kb + 1: issued in taps 0..2 of slice kb; the ONE barrier per slice sits behind the last read of
  //     the old window (item 11 of tap 4, whose look-ahead reads item 13), and publishes the new one for the look-ahead
  //     reads of items 12 / 13 (items 0 / 1 of the next slice).
  //   The look-ahead crosses the step boundary: items 12 / 13 read items 0 / 1 of the NEXT step (next tap's row shift).
#pragma unroll
  for (int k = 0; k < NPW; ++k) dma_win2(k, 0, 0);
#ifdef RADMMM_B_GLOBAL
#pragma unroll
  for (int w = 0; w < 8; ++w) load_b_piece(w, 0, 0);
  __syncthreads();
#else
#pragma unroll
  for (int w = 0; w < 8; ++w) dma_b2(w, 0, 0);
#pragma unroll
  for (int w = 0; w < 8; ++w) dma_b2(w, 1, b_tap_bytes);
  __syncthreads();
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int j = 0; j < 2; ++j) read_b1(0, 0, kb, j);
#endif
#pragma unroll
  for (int t = 0; t < D; ++t) {
    read_hi(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_lo(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  }
  constexpr int TW = NT - 2;                                          // the step's wait (and the slice's barrier) stand in front of item TW
  auto kstep = [&](auto tapc, auto parc, int kb) __attribute__((always_inline)) {
    constexpr int tap = decltype(tapc)::value, par = decltype(parc)::value;      // par = kb & 1
    constexpr int set = (par + tap) & 1;                              // step index 5 kb + tap is even / odd: B register set AND B stage
    constexpr int NW = G::nw(tap);
    constexpr int ntap = tap == WTAPS - 1 ? 0 : tap + 1, npar = tap == WTAPS - 1 ? (par ^ 1) : par;   // tile of step + 1
    constexpr int tap2 = (tap + 2) % WTAPS;                           // tile of step + 2
    const int soff2 = tap2 * b_tap_bytes + (kb + (tap + 2) / WTAPS) * (BK * 2);
    const int soff1 = ntap * b_tap_bytes + (kb + (tap + 1) / WTAPS) * (BK * 2);        // tile of step + 1 (RADMMM_B_GLOBAL)
    (void)soff1;
    using TapC = std::integral_constant<int, tap>;
    using ParC = std::integral_constant<int, par>;
    using NTapC = std::integral_constant<int, ntap>;
    using NParC = std::integral_constant<int, npar>;
    static_assert(NW + 8 <= NT - 2, "DMA slots of a step");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int kbk = t & 1, i = t >> 1;
      if (t == TW) {
        // everything but this step's own pieces has landed: B(s + 1) in particular; in the slice's last step every read of
        // the old window has been issued (and is waited for), the barrier publishes the new window
        if constexpr (tap == WTAPS - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" : : "n"(NW + 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW + 8) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      // slot A
      acc[i][0] = RADMMM_MFMA_F16(fah[t], bh[set][kbk][0], acc[i][0]);
      if (t + D < NT) read_hi(t + D, TapC{}, ParC{});
      else read_hi(t + D - NT, NTapC{}, NParC{});
#ifndef RADMMM_B_GLOBAL
      if (t >= TW) read_b1(set ^ 1, set ^ 1, t - TW, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
      // slot B
      acc[i][1] = RADMMM_MFMA_F16(fah[t], bh[set][kbk][1], acc[i][1]);
      if (t + D < NT) read_lo(t + D, TapC{}, ParC{});
      else read_lo(t + D - NT, NTapC{}, NParC{});
#ifndef RADMMM_B_GLOBAL
      if (t >= TW) read_b1(set ^ 1, set ^ 1, t - TW, 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      // slot C (items 1 .. 13; the 14th follows the loop)
      if (t > 0) {
        if (kbk == 1) cross(set, i, 0);
        else cross(set, i - 1, 1);
        const int c = t - 1;
#ifdef RADMMM_B_GLOBAL
        // B(s + 1) first (into the register set this step does not use: a whole step of flight time), the window pieces behind
        if (c < 8) load_b_piece(c, set ^ 1, soff1);
        else if (c - 8 < NW) dma_win2(G::WPT * tap + (c - 8), par ^ 1, kb + 1);
#else
        if (c < NW) dma_win2(G::WPT * tap + c, par ^ 1, kb + 1);
        else if (c - NW < 8) dma_b2(c - NW, set, soff2);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cross(set, MB - 1, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kb = 0; kb < kpt; kb += 2) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    kstep(std::integral_constant<int, 0>{}, P0{}, kb);
    kstep(std::integral_constant<int, 1>{}, P0{}, kb);
    kstep(std::integral_constant<int, 2>{}, P0{}, kb);
    kstep(std::integral_constant<int, 3>{}, P0{}, kb);
    kstep(std::integral_constant<int, 4>{}, P0{}, kb);
    kstep(std::integral_constant<int, 0>{}, P1{}, kb + 1);
    kstep(std::integral_constant<int, 1>{}, P1{}, kb + 1);
    kstep(std::integral_constant<int, 2>{}, P1{}, kb + 1);
    kstep(std::integral_constant<int, 3>{}, P1{}, kb + 1);
    kstep(std::integral_constant<int, 4>{}, P1{}, kb + 1);
  }
  __syncthreads();                                                    // stray fragment reads / DMA past the last tile

  // ---- optional EXTRA K segment (include/radmmm_hip.h: extra_tap): acc += A2 . B[taps], A2 = the rows extra_a_rows below,
  // no shift, no mask.  It runs AFTER all tap slices (rowgemm_h3d interleaves it per k slice: same products, another fp32
  // summation order), as a plain double-buffered loop -- one A tile and one B tile per K step, one barrier per step -- in
  // the LDS the windows no longer need -- rowgemm_onetap.h's loop (three A stages, wave-private B, slot-pinned order).
  if constexpr (XT) {
    using OG = OneGeo<MB>;                                            // B stages where they were, a ring of three A stages behind them
    int x_vo[MB], x_dst[MB], x_isl[MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
      const int c = 4 * k + wave;                                     // piece c of 4 MB: 16-row groups of the hi plane, then of the cross plane
      x_isl[k] = c >= 2 * MB ? 1 : 0;
      const int j = x_isl[k] ? c - 2 * MB : c;
      const int r = m0 + 16 * j + d_row;
      x_vo[k] = r < p.M ? ((q.extra_a_rows + r) * q.lda_h + d_chunk * 8) * 2 : OOB;
      x_dst[k] = OG::A_BASE + x_isl[k] * OG::A_PLANE + j * 1024;
    }
    auto dma_xa = [&](int k, int stage, int soff) __attribute__((always_inline)) {
      dma16s(x_isl[k] ? rAl : rAh, (lds_u32_ptr)(sm + stage * OG::A_STAGE + x_dst[k]), x_vo[k], soff);
    };
    auto dma_xb = [&](int w, int stage, int soff) __attribute__((always_inline)) { dma_b2(w, stage, WTAPS * b_tap_bytes + soff); };
    one_tap_steps<MB, PR>(acc, sm, kpt, lane, wave, x_sa, x_sb, dma_xa, dma_xb);
  }

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
#ifdef RADMMM_EPI_NONE
  {                                                                    // (timing only: keep the accumulators alive)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][0][e] + acc[i][1][e];
    if (s == 1.2345e-30f) p.C[0] = s;
  }
#else
  direct_epilogue<MB, EK, PR == 2>(acc, rowf4, p, m0, n0, lane, wave, sat);
  radmmm::raise_sat_flag(p.sat_flag, sat, (p.Ch && p.split_fmt != RADMMM_SPLIT_F16) ? __builtin_ldexpf(1.f, p.ch_x8_exp) : 0.f);
#endif
}

template <int MB, int EK, bool XT, int PR = 2>
int launch_win(const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  using G = WGeo<MB>;
  constexpr int smem_bytes = XT && OneGeo<MB>::SMEM > G::SMEM ? OneGeo<MB>::SMEM : G::SMEM;     // (the extra segment's A ring)
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_win_kernel<MB, EK, XT, PR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_win<%d,%d>): %s", MB, EK, hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  const radmmm_rowgemm_desc& p = d.base;
  const int ntm = (p.M + G::BMR - 1) / G::BMR, ntn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((rowgemm_win_kernel<MB, EK, XT, PR>), dim3(ntm * ntn), dim3(256), smem_bytes, stream, d, a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm_win");
}

template <int MB>
int launch_win_ek(int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  if (d.nprod != 2) return launch_win<MB, EK_PLAIN, false, 3>(d, stream, a_bytes, b_bytes);   // (rowgemm_win_ok: this kind only)
  if (d.extra_tap) return launch_win<MB, EK_DGRAD, true>(d, stream, a_bytes, b_bytes);      // (rowgemm_win_ok: this kind only)
  switch (ek) {
    case EK_SPLIT: return launch_win<MB, EK_SPLIT, false>(d, stream, a_bytes, b_bytes);
    case EK_DGRAD: return launch_win<MB, EK_DGRAD, false>(d, stream, a_bytes, b_bytes);
    default: return launch_win<MB, EK_PLAIN, false>(d, stream, a_bytes, b_bytes);
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
  // (three f16 products, round 5: C-only launches without the extra segment -- the FiLM stacks' 5-tap hidden convs)
  const bool scheme_ok = d.nprod == 2 || ((d.nprod == 3 || d.nprod == 0) && ek == EK_PLAIN && !d.extra_tap);
  return scheme_ok && (mb == 7 || mb == 8) && (ek == EK_PLAIN || ek == EK_SPLIT || ek == EK_DGRAD) && p.taps == WTAPS &&
         (!d.extra_tap || (ek == EK_DGRAD && !p.a_mask_mode)) && (p.K / BK) % 2 == 0 && p.dil >= 1 && p.dil <= WDMAX && p.T >= 32 * mb && p.M % p.T == 0 && (p.sign == 1 || p.sign == -1);
}
int launch_rowgemm_win(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
#ifndef RADMMM_QUICK
  if (mb == 8) return launch_win_ek<8>(ek, d, stream, a_bytes, b_bytes);
#endif
  return launch_win_ek<7>(ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm
