// Weight gradient of a (dilated, k-tap) conv on ROW-MAJOR split operands under the FP8-cross scheme (round 3):
//   P[split][tap][m][n] = acc_scale * sum_f GY[f][m] * X[f + s][n],   s = (tap - taps/2) * dil
// as wgrad_rm.hip (same tile machine, masking rules, split-K slabs, XCD-aware tile order), but the split product is
//   GYh.Xh                on the f16 pipe (two v_mfma_f32_32x32x16_f16 per 32-frame K step and 32x32 tile), and
//   GYh.Xl + GYl.Xh       as ONE block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (twice the f16 rate),
// i.e. two thirds of wgrad_rm's MFMA time -- that kernel is MFMA-bound (63 % busy, profiles/r02_pmc_wgrad_rm.txt).
// Operands per tensor: the fp16 hi plane [frames][ld] and an 8-bit lo plane (e4m3 of lo * 2^(11 + e)): the lo8 half of the
// tensor's 8-bit cross array (RADMMM_SPLIT_X8A: per 32 channels 64 bytes [hi8 x 32 | lo8 x 32], the array the GEMM
// epilogues write anyway) -- described by (row pitch, block stride, block offset) so that a plain lo8 plane fits too.
// No fp16 lo array is needed any more.  The hi8 halves of the FP8 operands are NOT read: they are a function of the fp16 hi
// fragments already in registers (v_cvt_scalef32_pk_fp8_f16 under MODE.FP16_OVFL = saturating, round to nearest even;
// measured semantics: tools/cvt_tr_probe.hip, profiles/r03_cvt_tr_probe.txt), 8 conversions per fragment pair under
// the MFMAs.
// LDS stage (48 KiB): GYh, Xh 32 frames x 512 B each (layout and ds_read_b64_tr_b16 fragments of wgrad_rm.hip), GYl8, Xl8
// 32 frames x 256 B; an 8-bit fragment (8 consecutive frames of one channel) is one ds_read_b64_tr_b8 -- a 16-lane group
// reads an [8 frames][16 channels] byte block and lane p receives column p (measured) -- with the 32-byte channel units of
// frame row r stored XOR-ed by (r & 7): the eight rows of a read fall into eight different bank groups.
// THREE stages: the 12 DMA pieces a wave issues during step s belong to step s + 2, and the barrier at the end of step s
// waits with vmcnt(12) -- for the pieces of step s + 1, issued a whole step earlier (vmcnt retires in order) -- so no
// barrier ever waits for a transfer that has just been started (with two stages the 32-frame step took 2.4 us against
// 1.08 us of MFMA work).  Every LDS read of the loop is inline asm: the compiler would otherwise order its own reads
// behind all outstanding LDS-DMA with vmcnt(0).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

constexpr int BK = 32, TM = 256, TN = 256;          // frames per K step, output tile
constexpr int HARR = BK * TM * 2;                    // bytes of one fp16 hi array in a stage: 32 rows x 512 B
constexpr int LARR = BK * TM;                        // bytes of one lo8 array in a stage: 32 rows x 256 B
constexpr int STAGE = 2 * HARR + 2 * LARR;           // GYh, Xh, GYl8, Xl8 = 48 KiB
constexpr int NSTAGE = 3;                            // LDS ring: the tile of step s + 2 is fetched during step s
constexpr int SMEM = NSTAGE * STAGE;                 // 144 KiB
constexpr int OOB = 0x7fffffff;

struct Rm8Args {
  const _Float16 *GYh, *Xh;               // [R][ld] row-major fp16 hi planes
  const unsigned char *GYl, *Xl;         // 8-bit arrays holding the lo8 parts
  int gl_pitch, gl_bstride, gl_boff;     // lo8 of GY channel c of frame f: GYl[f * pitch + (c >> 5) * bstride + boff + (c & 31)]
  int xl_pitch, xl_bstride, xl_boff;
  int g8_exp, x8_exp;                    // the lo8 parts hold lo * 2^(11 + e)
  const int* lens;                       // [R / T] or null; used when x_mask
  int x_mask;
  int R, T, ldg, ldx, Mc, Nc, taps, dil, splits;
  float* P; int ldp; long long split_stride;
  float acc_scale;
  int g_bytes, x_bytes, gl_bytes, xl_bytes;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);
#endif
}

#ifdef RADMMM_WG8_NOMASK                              // TIMING-ONLY build (wrong at utterance boundaries): no row masks on the X pieces
using WG8Plain = std::true_type;
#else
using WG8Plain = std::false_type;
#endif

// TIMING-ONLY builds (wrong results; rowgemm_h3w_kernel.h has the list): 1 / 2 no cross-term MFMAs, 3 no MFMA at all
#ifndef RADMMM_TIMING
#define RADMMM_TIMING 0
#endif
#if RADMMM_TIMING == 3
#define WG8_MFMA_F16(A, B, C) ([&] { asm volatile("" : : "v"(A), "v"(B)); return (C); }())
#else
#define WG8_MFMA_F16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#endif
#if RADMMM_TIMING == 0
#define WG8_MFMA_X(A, B, C, SA, SB) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((A), (B), (C), 0, 0, 0, (SA), 0, (SB))
#else
#define WG8_MFMA_X(A, B, C, SA, SB) ([&] { asm volatile("" : : "v"(A), "v"(B)); return (C); }())
#endif

struct Frag { i32x2 lo, hi; };
// Fragment addressing.  All lane-dependent parts of a fragment's LDS address are computed ONCE (round 3, second pass: the K
// step carried ~270 VALU instructions beside its 48 MFMAs -- more than the MFMA gaps hide -- of which 30 were these address
// sums and ~110 the per-piece DMA masks); per step only the stage base is added, everything else is an immediate offset.
//   fp16 hi planes (wgrad_rm.hip): this lane's 8 bytes of (32-channel unit u, k block kb) -- rows k and k + 4 of the lane's
//   half k block: k = 16 kb + 8 (gq >> 1) + (p >> 2), byte k * 512 + ((u ^ (p >> 2)) << 6) + (gq & 1) * 32 + (p & 3) * 8.  The XOR
//   touches the two low bits of u only: base[u & 3] + (u >> 2) * 256 (+ 8192 for k block 1, + 2048 for rows k + 4).
__device__ __forceinline__ int hi_lane_base(int q, int lane) {
  const int p = lane & 15, gq = lane >> 4;
  return (8 * (gq >> 1) + (p >> 2)) * 512 + ((q ^ (p >> 2)) << 6) + (gq & 1) * 32 + (p & 3) * 8;
}
template <int OFF>
__device__ __forceinline__ void frag_issue(Frag& f, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "n"(OFF), "n"(OFF + 2048) : "memory");
}
__device__ __forceinline__ f16x8 frag_val(const Frag& f) {
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3));
}
// lo8 planes: this lane's (channel (gq & 1) * 16 + p of unit u, frames 8 h .. 8 h + 7 of k block 0) -- row r = 8 h + (p >> 1)
// supplies the 8-byte segment (p & 1); k block 1 lies 16 rows = 4096 bytes further (same r & 7).  The XOR with r & 7 = p >> 1
// touches all three bits of u: one lane base per unit.
__device__ __forceinline__ int lo8_lane_base(int u, int lane) {
  const int p = lane & 15, gq = lane >> 4;
  const int r = 8 * (gq >> 1) + (p >> 1);
  return r * 256 + ((u ^ (r & 7)) << 5) + (gq & 1) * 16 + (p & 1) * 8;
}
__device__ __forceinline__ void frag8_issue(Frag& f, unsigned addr) {      // .lo: k block 0, .hi: k block 1
  asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:4096" : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr) : "memory");
}
// wait until at most N LDS operations issued AFTER these fragments are outstanding (LDS returns in order)
template <int N>
__device__ __forceinline__ void frag_wait2(Frag& a, Frag& b) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_wait3(Frag& a, Frag& b, Frag& c) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi) : "n"(N) : "memory");
}
// e4m3(hi * 2^e) of the 16 fp16 values of a fragment pair (k block 0, k block 1) -> 16 bytes in fragment order.
// inv = 2^-e: the instruction divides by its scale operand; saturating under MODE.FP16_OVFL.
__device__ __forceinline__ int cvt4_fp8(int lo2, int hi2, float inv) {
  // two packed conversions into the two halves of ONE register.  As inline asm with a write-only destination: the builtin's
  // destination is read-modify-write, and the compiler materialises its (dead) initial value with a v_mov -- 40 per K step.
#ifdef RADMMM_TIMING_NOCVT                 // TIMING-ONLY build (wrong results): the 80 hi8 conversions per K step are not issued --
  return lo2 ^ hi2;                         // upper bound of "read the hi8 halves from the cross arrays instead" (profiles/r05_wgrad_hi8.txt)
#else
  int o;
  asm("v_cvt_scalef32_pk_fp8_f16 %0, %1, %3\n\tv_cvt_scalef32_pk_fp8_f16 %0, %2, %3 op_sel:[0,0,1]" : "=&v"(o) : "v"(lo2), "v"(hi2), "v"(inv));
  return o;
#endif
}
__device__ __forceinline__ i32x4 hi8_of(const Frag& f0, const Frag& f1, float inv) {
  i32x4 r;
  r[0] = cvt4_fp8(f0.lo[0], f0.lo[1], inv);
  r[1] = cvt4_fp8(f0.hi[0], f0.hi[1], inv);
  r[2] = cvt4_fp8(f1.lo[0], f1.lo[1], inv);
  r[3] = cvt4_fp8(f1.hi[0], f1.hi[1], inv);
  return r;
}

__global__ __launch_bounds__(256, 1) void wgrad_rm8_kernel(const Rm8Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");      // MODE.FP16_OVFL: the fp8 conversions saturate
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntm = (a.Mc + TM - 1) / TM, ntn = (a.Nc + TN - 1) / TN;
  const int nt = ntm * ntn * a.taps * a.splits, wg = blockIdx.x;
  const int xcd = wg & 7, loc = wg >> 3, qq = nt >> 3, r8 = nt & 7;
  int id = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + loc;
  const int tn = id % ntn; id /= ntn;
  const int tm = id % ntm; id /= ntm;
  const int tap = id % a.taps;
  const int split = id / a.taps;
  const int m0 = tm * TM, n0 = tn * TN;
  const int shift = (tap - a.taps / 2) * a.dil;
  const int steps_total = (a.R + BK - 1) / BK;
  const int steps_per = (steps_total + a.splits - 1) / a.splits;
  const int step_lo = split * steps_per;
  int step_hi = step_lo + steps_per;
  if (step_hi > steps_total) step_hi = steps_total;
  const int nsteps = step_hi - step_lo;

  // DMA pieces of one wave per stage.
  //   hi planes: 16 pieces per array (two 512-byte frame rows each) -> piece w = 0..7: array w >> 2 (GYh, Xh), row pair
  //              4 (w & 3) + wave; lane: row of the pair lane >> 5, LDS unit (lane & 31) >> 2, 16-byte part lane & 3
  //   lo8 planes: 8 pieces per array (four 256-byte frame rows each) -> piece w = 8..11: array (w - 8) >> 1 (GYl8, Xl8), row
  //              quad 4 ((w - 8) & 1) + wave; lane: row of the quad lane >> 4, LDS unit (lane >> 1) & 7, 16-byte half lane & 1
  // Per piece: this lane's frame offset within a K step, its byte offset at step 0 of the split, and (X arrays) the
  // frame-in-utterance counter of its row at the step being fetched.
  const __amdgpu_buffer_rsrc_t rGh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYh), 0, a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Xh), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.GYl), 0, a.gl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.Xl), 0, a.xl_bytes, 0x00020000);
  // Masks.  Array ends need none: a frame beyond R (or, with a tap shift, before 0) is an offset outside the buffer and the
  // DMA writes zeros.  What needs one is the X frame of a row leaving its utterance's readable range [0, lim): per piece
  // this lane's row k of the 32-frame window and c = k + shift are constants, the window's position is UNIFORM --
  // (w_b, w_t) = utterance and frame-in-utterance of its first frame, lim0 / lim1 the readable frames of utterances w_b and
  // w_b + 1 (T >= 32: a window meets at most one boundary) -- and lives in scalar registers: a row is readable iff
  //   k + w_t < T ?  0 <= c + w_t < lim0  :  0 <= c + w_t - T < lim1
  // Round 4: the test is made ONCE per step for the 32 rows of the window (row_mask: lane l stands for row l & 31, the
  // ballot is a 32-bit scalar with bit k set when row k is NOT readable); an X piece moves its row's bit to the sign,
  // spreads it and ORs 0x7fffffff into its lane offset (3 vector instructions instead of 10 per piece: the K loop is bound by
  // the ONE wave's instruction issue).  The step's position itself is the DMA instruction's scalar offset.  Pieces of steps
  // beyond the split's end go to stages nobody reads, whatever they fetch.
  int p_off[12], p_sh[12];
  const int nb = a.R / a.T;
  auto lim_of = [&](int b) __attribute__((always_inline)) {
    if (b >= nb) return 0;
    if (!(a.x_mask && a.lens)) return a.T;
    const int l = a.lens[b];
    return l < a.T ? l : a.T;
  };
  int w_b = (step_lo * BK) / a.T, w_t = step_lo * BK - w_b * a.T;
  int lim0 = lim_of(w_b), lim1 = lim_of(w_b + 1);
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int isx = w >> 2, pr = 4 * (w & 3) + wave;
    const int k = 2 * pr + (lane >> 5);
    const int u = ((lane & 31) >> 2) ^ (k & 3);                  // source 64-byte unit that lands at this lane's LDS unit
    const int c0 = isx ? n0 : m0, ld = isx ? a.ldx : a.ldg;
    const int ch = c0 + u * 32 + (lane & 3) * 8;
    p_sh[w] = 31 - k;
    const int f0 = step_lo * BK + k;                             // GY frame of this row at the split's first step
    p_off[w] = ch < ld ? ((f0 + (isx ? shift : 0)) * ld + ch) * 2 : OOB;
  }
#pragma unroll
  for (int w = 8; w < 12; ++w) {
    const int isx = (w - 8) >> 1, q = 4 * ((w - 8) & 1) + wave;
    const int k = 4 * q + (lane >> 4);
    const int u = ((lane >> 1) & 7) ^ (k & 7);                   // source 32-channel unit that lands at this lane's LDS unit
    const int c0 = isx ? n0 : m0, ld = isx ? a.ldx : a.ldg;
    const int ch = c0 + u * 32 + (lane & 1) * 16;
    p_sh[w] = 31 - k;
    const int f0 = step_lo * BK + k;
    const int pitch = isx ? a.xl_pitch : a.gl_pitch, bs = isx ? a.xl_bstride : a.gl_bstride, bo = isx ? a.xl_boff : a.gl_boff;
    p_off[w] = ch < ld ? (f0 + (isx ? shift : 0)) * pitch + (ch >> 5) * bs + bo + (ch & 31) : OOB;
  }
  const int g_step = BK * a.ldg * 2, x_step = BK * a.ldx * 2, gl_step = BK * a.gl_pitch, xl_step = BK * a.xl_pitch;
  // bit k set: row k of the window being fetched (state w_t, lim0, lim1) is not readable through this tap's shift
  auto row_mask = [&]() __attribute__((always_inline)) {
    int k = lane & 31;
    asm volatile("" : "+v"(k));                                  // (pins the computation where it is called: see the K loop's slots)
    const int x = k + shift + w_t;
    const bool ok = (k + w_t < a.T) ? ((unsigned)x < (unsigned)lim0) : ((unsigned)(x - a.T) < (unsigned)lim1);
    return ~(unsigned)__ballot(ok);
  };
  // piece w of relative step `rel` into stage `buf`; nmask = row_mask() of that step's window (X pieces only)
  auto dma_piece = [&](int buf, int w, int rel, unsigned nmask) __attribute__((always_inline)) {
    const bool lo8 = w >= 8;
    const int isx = lo8 ? (w - 8) >> 1 : w >> 2;
    const int stepb = lo8 ? (isx ? xl_step : gl_step) : (isx ? x_step : g_step);
    // GY pieces: the step's position is the instruction's scalar offset (their lane offsets are never negative).  X pieces:
    // with a negative tap shift the lane offset of the split's first rows IS negative until the window has moved on, and the
    // range check takes the unsigned sum of lane and scalar offset: one vector add, then the mask (OOB | anything and
    // OOB + the position stay out of range)
    int vo = p_off[w];
    if (isx) {
      vo += rel * stepb;
      if (!WG8Plain::value) vo |= ((int)(nmask << p_sh[w]) >> 31) & OOB;
    }
    const int dst = lo8 ? 2 * HARR + isx * LARR + (4 * ((w - 8) & 1) + wave) * 1024 : isx * HARR + (4 * (w & 3) + wave) * 1024;
    dma16s(lo8 ? (isx ? rXl : rGl) : (isx ? rXh : rGh), (lds_u32_ptr)(sm + buf * STAGE + dst), vo, isx ? 0 : rel * stepb);
  };
  auto advance_window = [&]() __attribute__((always_inline)) {   // the window moves on by one K step (scalar work)
    w_t += BK;
    if (w_t >= a.T) {
      w_t -= a.T;
      ++w_b;
      lim0 = lim1;
      lim1 = lim_of(w_b + 1);
    }
  };

  f32x16 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // E8M0 block scales of the cross MFMA (rowgemm_h3w_kernel.h): A = [GYhi8 | GYlo8 * 2^11], B = [Xlo8 * 2^11 | Xhi8], each
  // further multiplied by 2^g8_exp / 2^x8_exp; the scale byte of lane (row, half 0) applies to the first 16 bytes
  const int x_sa = (lane >> 5) ? 127 - 11 - a.g8_exp : 127 - a.g8_exp;
  const int x_sb = (lane >> 5) ? 127 - a.x8_exp : 127 - 11 - a.x8_exp;
  const float g_inv = __builtin_ldexpf(1.f, -a.g8_exp), x_inv = __builtin_ldexpf(1.f, -a.x8_exp);

  if (nsteps > 0) {
    // prologue: tiles 0 and 1 (the pieces of a step beyond the split's end are issued all the same, with out-of-range
    // offsets -- zeros into a stage nobody reads -- so that every step issues exactly 12 pieces per wave)
    {
      const unsigned nm = row_mask();
#pragma unroll
      for (int w = 0; w < 12; ++w) dma_piece(0, w, 0, nm);
    }
    advance_window();
    {
      const unsigned nm = row_mask();
#pragma unroll
      for (int w = 0; w < 12; ++w) dma_piece(1, w, 1, nm);
    }
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    // lane parts of the fragment addresses (stage-relative; arrays: GYh at 0, Xh at HARR, GYl8 at 2 HARR, Xl8 behind it)
    const unsigned sm_base = (unsigned)reinterpret_cast<size_t>((lds_u32_ptr)sm);
    unsigned ah_base[4], a8_base[8], bh_base[2], b8_base[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) ah_base[q] = sm_base + hi_lane_base(q, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) a8_base[i] = sm_base + 2 * HARR + lo8_lane_base(i, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u = 2 * wave + j;
      bh_base[j] = sm_base + HARR + hi_lane_base(u & 3, lane) + (u >> 2) * 256;
      b8_base[j] = sm_base + 2 * HARR + LARR + lo8_lane_base(u, lane);
    }
    // ---- K loop (round 4).  One wave per SIMD issues everything in order: whatever stands between two MFMAs must fit under
    // the first one's 32 (f16) / 64 (fp8) cycles, or the matrix pipe idles.  Round 3's loop left the order to the compiler
    // (runs of 14 VALU instructions and bursts of 6 LDS reads between MFMAs, every row block waiting for its own fragments
    // and converting them right in front of its MFMAs).  Now a software pipeline three row blocks deep, in SLOTS closed by
    // sched_barrier(0): while row block i multiplies, the fragments of block i + 2 are read and the FP8 hi halves of block
    // i + 1 are converted:
    //     slot 0: MFMA (i, 0, k block 0) | hi fragments of block i + 2, k block 0
    //     slot 1: MFMA (i, 1, k block 0) | hi fragments ..., k block 1
    //     slot 2: MFMA (i, 0, k block 1) | lo8 fragments of block i + 2
    //     slot 3: MFMA (i, 1, k block 1) |
    //     slot 4: cross MFMA (i, 0)      | one DMA piece of tile s + 2 (blocks 0 .. 5) | 2 conversion pairs of block i + 1
    //     slot 5: cross MFMA (i, 1)      | one DMA piece                               | 2 conversion pairs
    // -- a single wave's ISSUE bandwidth is the limit here (48 MFMAs carry ~60 LDS reads, 12 DMA pieces and, in round 3,
    // ~170 vector instructions per step: more issue cycles than the MFMAs' 2048), so the loop also sheds vector work: DMA
    // pieces take their row mask from one 32-bit ballot per step and their position from the instruction's scalar offset
    // (per accumulator still k block 0, k block 1, cross: the results are bit-identical to round 3's kernel).  The pipeline
    // runs across the step boundary: blocks 6 / 7 read blocks 0 / 1 of the NEXT tile, so the step's wait (own vmcnt: tile
    // s + 1 has landed; lgkmcnt(0): every read of tile s is complete) and barrier stand in front of block 6.  The X-side
    // fragments (this wave's two units) are refreshed IN PLACE under block 7, each right behind the last MFMA that reads the
    // old one, and their hi8 halves converted under block 7's two cross MFMAs: the first MFMA of a step finds its operands
    // in registers.
    Frag ga0[4], ga1[4], ga8[4];                                    // A side (GY): ring of four row blocks (8 blocks per step: slot = i & 3)
    i32x4 ah8[2];                                                   // converted hi8 halves: slot = i & 1
    Frag xb0[2], xb1[2], xb8[2];                                    // X side, unit j: fp16 hi fragments of k block 0 / 1, lo8 fragments
    i32x4 xh8[2];                                                   // ... and their converted hi8 halves
    i32x8 b8[2];                                                    // FP8 operand [lo8 | hi8] of the step in progress
    auto read_a = [&](auto ic, unsigned sb) __attribute__((always_inline)) {      // all six fragment reads of row block ic (prologue)
      constexpr int i = decltype(ic)::value;
      if constexpr (i < 4) {
        frag_issue<0>(ga0[i & 3], ah_base[i & 3] + sb);
        frag_issue<8192>(ga1[i & 3], ah_base[i & 3] + sb);
      } else {
        frag_issue<256>(ga0[i & 3], ah_base[i & 3] + sb);
        frag_issue<8192 + 256>(ga1[i & 3], ah_base[i & 3] + sb);
      }
      frag8_issue(ga8[i & 3], a8_base[i] + sb);
    };
    // pipeline prologue: X fragments of tile 0, row blocks 0 and 1 of tile 0, hi8 of block 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      frag_issue<0>(xb0[j], bh_base[j]);
      frag_issue<8192>(xb1[j], bh_base[j]);
      frag8_issue(xb8[j], b8_base[j]);
    }
    read_a(std::integral_constant<int, 0>{}, 0u);
    read_a(std::integral_constant<int, 1>{}, 0u);
    frag_wait3<12>(xb0[0], xb1[0], xb8[0]);
    frag_wait3<12>(xb0[1], xb1[1], xb8[1]);
    xh8[0] = hi8_of(xb0[0], xb1[0], x_inv);
    xh8[1] = hi8_of(xb0[1], xb1[1], x_inv);
    frag_wait3<6>(ga0[0], ga1[0], ga8[0]);
    ah8[0] = hi8_of(ga0[0], ga1[0], g_inv);
    __builtin_amdgcn_sched_barrier(0);

    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
      advance_window();                                            // the window follows the tile being fetched: s + 2
      const int l_rel = s + 2;
      unsigned nmask = 0;
      const int nbuf = buf >= 1 ? buf - 1 : NSTAGE - 1;            // (buf + 2) % 3
      const int xbuf = buf + 1 < NSTAGE ? buf + 1 : 0;             // stage of tile s + 1
      const unsigned sb = (unsigned)(buf * STAGE), sbn = (unsigned)(xbuf * STAGE);
      // this step's X operands were read / converted under the previous step's last block (or in the prologue): the fp16
      // fragments are older than the four lo8 reads behind them; the lo8 fragments are waited for where the FP8 operand is
      // put together (slot 3 of block 0: six more reads have been issued by then)
      frag_wait2<4>(xb0[0], xb1[0]);
      frag_wait2<4>(xb0[1], xb1[1]);
      const f16x8 bh0[2] = {frag_val(xb0[0]), frag_val(xb0[1])}, bh1[2] = {frag_val(xb1[0]), frag_val(xb1[1])};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cur = i & 3, nx1 = (i + 1) & 3, nx2 = (i + 2) & 3;
        const int i2 = (i + 2) & 7;                                // row block being read (of this tile, or of the next one)
        const unsigned sb2 = i + 2 < 8 ? sb : sbn;
        if (i == 6) {
          // tile s + 1 has landed as far as this wave fetched it, every read of tile s is complete: publish, and free
          // tile s's stage for tile s + 3
#ifdef RADMMM_WG8_GY_EVERY
          if ((l_rel % RADMMM_WG8_GY_EVERY) != 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else
#endif
          asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        const f16x8 ah0 = frag_val(ga0[cur]), ah1 = frag_val(ga1[cur]);
        const i32x4 l8c = __builtin_shufflevector(ga8[cur].lo, ga8[cur].hi, 0, 1, 2, 3);
        const i32x8 a8 = __builtin_shufflevector(ah8[i & 1], l8c, 0, 1, 2, 3, 4, 5, 6, 7);
        i32x4 h8n;                                                 // hi8 of block i + 1, converted under this block's f16 MFMAs
        // slot 0
        acc[i][0] = WG8_MFMA_F16(ah0, bh0[0], acc[i][0]);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        // -DRADMMM_WG8_SKIP_READS, TIMING-ONLY (wrong results): the A-side fragment reads of every other row block are not
        // issued -- 24 of the 60 LDS reads of a K step; a 2 x 2 wave tiling of the same 256 x 256 tile would save 12 (48 reads
        // instead of 60): an upper bound of that redesign before anything is built (profiles/r05_wgrad_hi8.txt)
#ifdef RADMMM_WG8_SKIP_READS
        const bool do_rd = (i2 & 1) == 0;
#else
        const bool do_rd = true;
#endif
        if (do_rd) {
          if (i2 < 4) frag_issue<0>(ga0[nx2], ah_base[i2 & 3] + sb2);
          else frag_issue<256>(ga0[nx2], ah_base[i2 & 3] + sb2);
        }
        frag_wait3<2>(ga0[nx1], ga1[nx1], ga8[nx1]);               // block i + 1's fragments (read a whole block ago): two younger reads in flight
        if (i == 7) frag_issue<0>(xb0[0], bh_base[0] + sbn);        // (bh0[0] holds the old value: last read by the MFMA above)
        __builtin_amdgcn_sched_barrier(0);
        // slot 1
        acc[i][1] = WG8_MFMA_F16(ah0, bh0[1], acc[i][1]);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        if (do_rd) {
          if (i2 < 4) frag_issue<8192>(ga1[nx2], ah_base[i2 & 3] + sb2);
          else frag_issue<8192 + 256>(ga1[nx2], ah_base[i2 & 3] + sb2);
        }
        if (i == 7) frag_issue<0>(xb0[1], bh_base[1] + sbn);
        __builtin_amdgcn_sched_barrier(0);
        // slot 2
        acc[i][0] = WG8_MFMA_F16(ah1, bh1[0], acc[i][0]);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        if (do_rd) frag8_issue(ga8[nx2], a8_base[i2] + sb2);
        if (i == 7) frag_issue<8192>(xb1[0], bh_base[0] + sbn);
        __builtin_amdgcn_sched_barrier(0);
        // slot 3
        acc[i][1] = WG8_MFMA_F16(ah1, bh1[1], acc[i][1]);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        if (i == 0) {
          frag_wait2<6>(xb8[0], xb8[1]);                           // (younger: this block's six A reads)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            b8[j] = __builtin_shufflevector(__builtin_shufflevector(xb8[j].lo, xb8[j].hi, 0, 1, 2, 3), xh8[j], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if (i == 7) frag_issue<8192>(xb1[1], bh_base[1] + sbn);
        __builtin_amdgcn_sched_barrier(0);
        // slot 4
        acc[i][0] = WG8_MFMA_X(a8, b8[0], acc[i][0], x_sa, x_sb);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        // -DRADMMM_WG8_GY_EVERY=n, TIMING-ONLY (wrong results; VERDICT r5 item 2b measured before built): the GY pieces of a
        // tile (0..3 hi, 8 / 9 lo8: half of the 12) are issued in one step of n only -- the stage keeps older GY frames, real
        // data -- i.e. the instruction and L2 -> LDS byte mix of a tile that keeps GY resident for all five taps (n = 5: 40 % of
        // the DMA pieces gone); the step's wait counts what the step issued.  profiles/r06_tile_probes.txt
#ifdef RADMMM_WG8_GY_EVERY
#define WG8_PIECE_ON(w) (((w) >= 4 && (w) != 8 && (w) != 9) || gy_step)
        const bool gy_step = (l_rel % RADMMM_WG8_GY_EVERY) == 0;
#else
#define WG8_PIECE_ON(w) true
#endif
        if (i < 6 && WG8_PIECE_ON(2 * i)) dma_piece(nbuf, 2 * i, l_rel, nmask);
        if (i == 1) nmask = row_mask();                            // (first needed by block 2's pieces)
        h8n[0] = cvt4_fp8(ga0[nx1].lo[0], ga0[nx1].lo[1], g_inv);
        h8n[1] = cvt4_fp8(ga0[nx1].hi[0], ga0[nx1].hi[1], g_inv);
        if (i == 7) {
          frag8_issue(xb8[0], b8_base[0] + sbn);
          frag_wait2<4>(xb0[0], xb1[0]);                           // (xb0[0], xb1[0] of the next tile: four younger reads)
          xh8[0] = hi8_of(xb0[0], xb1[0], x_inv);
        }
        __builtin_amdgcn_sched_barrier(0);
        // slot 5
        acc[i][1] = WG8_MFMA_X(a8, b8[1], acc[i][1], x_sa, x_sb);
        __builtin_amdgcn_sched_barrier(0);                         // (the MFMA opens its slot)
        if (i < 6 && WG8_PIECE_ON(2 * i + 1)) dma_piece(nbuf, 2 * i + 1, l_rel, nmask);
        h8n[2] = cvt4_fp8(ga1[nx1].lo[0], ga1[nx1].lo[1], g_inv);
        h8n[3] = cvt4_fp8(ga1[nx1].hi[0], ga1[nx1].hi[1], g_inv);
        ah8[(i + 1) & 1] = h8n;
        if (i == 7) {
          frag8_issue(xb8[1], b8_base[1] + sbn);
          frag_wait2<4>(xb0[1], xb1[1]);                           // (xb0[1], xb1[1]: the two lo8 reads are younger)
          xh8[1] = hi8_of(xb0[1], xb1[1], x_inv);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      buf = xbuf;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                               // the trailing (out-of-range) pieces and look-ahead reads, before LDS is reused
  }
  float* P = a.P + (long long)split * a.split_stride + (long long)tap * a.Mc * a.ldp;
  // Accumulators straight to P: a half wave writes 128 contiguous bytes of one row (32 lanes = 32 columns of an accumulator
  // block).  Round 4 staged every row block through LDS for 16-byte stores (8 x {32 ds_write, barrier, 8 ds_read + store,
  // barrier}); the 4-byte stores need no barrier and no LDS: same values, step A/B 39.42 / 39.47 / 39.40 -> 39.22 / 39.26 /
  // 39.27 ms.
  {
    const int colb = n0 + wave * 64 + (lane & 31);
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + I * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = colb + j * 32;
          if (row < a.Mc && col < a.Nc) P[(long long)row * a.ldp + col] = acc[I][j][e] * a.acc_scale;
        }
  }
}

}  // namespace

// GYh / Xh [R][ldg / ldx] row-major fp16 hi planes of scale_g * gy and of x; GYx / Xx the tensors' 8-bit cross arrays
// (RADMMM_SPLIT_X8A, row pitch 2 * ld bytes) written with exponents g8_exp / x8_exp: only their lo8 halves are read.
// Everything else as radmmm_wgrad_rm.  ldg, ldx multiples of 32, 16-byte aligned operands, T >= 32, at most 1024 utterances.
extern "C" int radmmm_wgrad_rm8(const void* GYh, const void* GYx, int ldg, int g8_exp, const void* Xh, const void* Xx, int ldx,
                                int x8_exp, int R, int T, const int32_t* lens, int x_mask, float* P, int ldp, int64_t split_stride,
                                int Mc, int Nc, int taps, int dil, int splits, float acc_scale, radmmm_stream_t stream) {
  RADMMM_REQUIRE(GYh && GYx && Xh && Xx && P, "wgrad_rm8: null pointer");
  RADMMM_REQUIRE(Mc > 0 && Nc > 0 && taps >= 1 && dil >= 1 && splits >= 1 && R > 0 && T > 0 && R % T == 0 && ldg >= Mc &&
                     ldx >= Nc && ldg % 32 == 0 && ldx % 32 == 0 && ldp >= Nc && T >= BK && R / T <= 1024 &&
                     abs(g8_exp) <= 16 && abs(x8_exp) <= 16,
                 "wgrad_rm8: bad dims (ldg, ldx %% 32 == 0, R = B * T, T >= 32, B <= 1024, |x8_exp| <= 16)");
  RADMMM_REQUIRE(radmmm::aligned16(GYh) && radmmm::aligned16(GYx) && radmmm::aligned16(Xh) && radmmm::aligned16(Xx),
                 "wgrad_rm8: 16-byte aligned operands");
  const long long g_bytes = (long long)R * ldg * 2, x_bytes = (long long)R * ldx * 2;
  RADMMM_REQUIRE(g_bytes < 0x7fffffffLL && x_bytes < 0x7fffffffLL, "wgrad_rm8: operand >= 2 GiB");
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_rm8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SMEM + 4096);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(wgrad_rm8): %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  Rm8Args a;
  a.GYh = static_cast<const _Float16*>(GYh); a.Xh = static_cast<const _Float16*>(Xh);
  a.GYl = static_cast<const unsigned char*>(GYx); a.Xl = static_cast<const unsigned char*>(Xx);
  a.gl_pitch = 2 * ldg; a.gl_bstride = 64; a.gl_boff = 32;       // lo8 half of the A-role cross array
  a.xl_pitch = 2 * ldx; a.xl_bstride = 64; a.xl_boff = 32;
  a.g8_exp = g8_exp; a.x8_exp = x8_exp;
  a.lens = lens; a.x_mask = x_mask;
  a.R = R; a.T = T; a.ldg = ldg; a.ldx = ldx; a.Mc = Mc; a.Nc = Nc; a.taps = taps; a.dil = dil; a.splits = splits;
  a.P = P; a.ldp = ldp; a.split_stride = split_stride; a.acc_scale = acc_scale;
  a.g_bytes = (int)g_bytes; a.x_bytes = (int)x_bytes; a.gl_bytes = (int)g_bytes; a.xl_bytes = (int)x_bytes;
  const int tiles = ((Mc + TM - 1) / TM) * ((Nc + TN - 1) / TN) * taps;
  hipLaunchKernelGGL(wgrad_rm8_kernel, dim3(tiles * splits), dim3(256), SMEM + 4096, static_cast<hipStream_t>(stream), a);
  return radmmm::check_launch("wgrad_rm8");
}
