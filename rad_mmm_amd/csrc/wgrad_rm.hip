// Weight gradient of a (dilated, k-tap) conv on ROW-MAJOR split operands: the [frames][channels] hi/lo fp16 pairs the
// GEMM epilogues already write, contracted over the frames, transposing in the LDS read (ds_read_b64_tr_b16).  No
// transposed zero-gapped copies (wgrad_h3.hip's operands) are needed.
//   P[split][tap][m][n] = acc_scale * sum_f GY[f][m] * X[f + s][n],   s = (tap - taps/2) * dil,
//   over the frames f of one split whose partner f + s lies in the SAME utterance: rows are utterance-major,
//   T frames each (f = b*T + t, 0 <= t + s < T), and -- x_mask (partial padding: the conv's input is x * mask) -- below
//   the utterance's length (t + s < lens[b]).
// Tile machine as wgrad_h3.hip (one workgroup per CU, 256 x 256 output tile, 4 waves x (8 x 2) accumulators, LDS-DMA
// double buffering, three f16 MFMA products hi.lo + lo.hi + hi.hi, split-K, one tap per workgroup, XCD-aware tile
// order).  What differs (measured in tools/wgrad_rm_probe.hip: 361 us for the 5-tap 1024 x 1024 gradient against 382 us +
// 44 us of transposing passes):
//   * a K step is 32 FRAMES; each DMA instruction brings two frame rows of 256 channels (2 x 512 B) of one array, its
//     16-byte pieces XOR-ed by (k & 3) at 64-byte granularity so that the four rows a 16-lane group of the transposing
//     read touches fall into four different bank groups (SQ_LDS_BANK_CONFLICT = 0);
//   * an MFMA operand fragment (8 consecutive frames of one channel) is two ds_read_b64_tr_b16.  They are issued
//     through inline asm: the compiler models the builtin form as a read of all of LDS and parks an s_waitcnt vmcnt(0)
//     behind every LDS-DMA instruction in front of it (measured: 517 us instead of 361); the waits are ours
//     (frag_wait ties them to the registers they release);
//   * the tap shift is a row offset of the X operand with a per-row utterance-boundary predicate (the frame-in-utterance
//     counter of each DMA piece advances by 32 per step: three integer ops per piece and step).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

constexpr int BK = 32, TM = 256, TN = 256;          // frames per K step, output tile
constexpr int ARR = BK * TM * 2;                     // bytes of one operand array in a stage: 32 rows x 512 B
constexpr int STAGE = 4 * ARR;                       // GYh, GYl, Xh, Xl
constexpr int SMEM = 2 * STAGE;                      // 128 KiB; the epilogue reuses it
constexpr int OOB = 0x7fffffff;

struct RmArgs {
  const _Float16 *GYh, *GYl, *Xh, *Xl;   // [R][ld] row-major
  const int* lens;                       // [R / T] or null; used when x_mask
  int x_mask;
  int R, T, ldg, ldx, Mc, Nc, taps, dil, splits;
  float* P; int ldp; long long split_stride;
  float acc_scale;
  int g_bytes, x_bytes;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

struct Frag { i32x2 lo, hi; };
// LDS address of this lane's 8 bytes of (32-channel unit u, k block kb): rows k and k + 4 of the lane's half k block
__device__ __forceinline__ unsigned frag_addr(const unsigned char* arr, int u, int kb, int lane) {
  const int p = lane & 15, gq = lane >> 4;
  const int k = 16 * kb + 8 * (gq >> 1) + (p >> 2);            // (k & 3) == (p >> 2) for this row and the one 4 below
  const int off = k * 512 + ((u ^ (p >> 2)) << 6) + (gq & 1) * 32 + (p & 3) * 8;
  return (unsigned)reinterpret_cast<size_t>((lds_u32_ptr)(arr + off));
}
__device__ __forceinline__ void frag_issue(Frag& f, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr) : "memory");
}
__device__ __forceinline__ f16x8 frag_val(const Frag& f) {
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3));
}
// wait until at most N LDS operations issued AFTER these fragments are outstanding (LDS returns in order)
template <int N>
__device__ __forceinline__ void frag_wait2(Frag& a, Frag& b) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void frag_wait4(Frag& a, Frag& b, Frag& c, Frag& d) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi)
               : "n"(N) : "memory");
}


__global__ __launch_bounds__(256, 1) void wgrad_rm_kernel(const RmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
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

  // DMA pieces: 16 per array (two frame rows each), 64 per stage, 16 per wave: piece w of this wave -> array w >> 2,
  // row pair 4 * (w & 3) + wave.  Per piece: this lane's frame offset within a K step, its byte offset at step 0 of the
  // split, and (X arrays) the frame-in-utterance counter of its row at the step being fetched.
  const int d_half = lane >> 5, d_unit = (lane & 31) >> 2, d_p16 = lane & 3;
  const __amdgpu_buffer_rsrc_t rGh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYh), 0, a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYl), 0, a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Xh), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Xl), 0, a.x_bytes, 0x00020000);
  // (the Xh and Xl pieces of a row pair share their row: four (frame-in-utterance, utterance, limit) triples per wave)
  int p_k[16], p_off[16], p_t[4], p_b[4], p_lim[4];
  int* lim_tab = reinterpret_cast<int*>(sm + SMEM);              // readable frames per utterance: min(len, T) or T
  const int nb = a.R / a.T;
  for (int i = tid; i < nb; i += 256) {
    const int l = (a.x_mask && a.lens) ? a.lens[i] : a.T;
    lim_tab[i] = l < a.T ? l : a.T;
  }
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int arr = w >> 2, pr = 4 * (w & 3) + wave;
    const int k = 2 * pr + d_half;
    const int u = d_unit ^ (k & 3);                              // source 64-byte unit that lands at d_unit
    const bool isx = arr >= 2;
    const int c0 = isx ? n0 : m0, C = isx ? a.Nc : a.Mc, ld = isx ? a.ldx : a.ldg;
    const int ch = c0 + u * 32 + d_p16 * 8;
    p_k[w] = k;
    const int f0 = step_lo * BK + k;                             // GY frame of this row at the split's first step
    p_off[w] = ch < C ? ((f0 + (isx ? shift : 0)) * ld + ch) * 2 : OOB;
    if (arr == 2) {
      const int b = f0 / a.T;
      const int bc = b < nb ? b : nb - 1;
      const int l = (a.x_mask && a.lens) ? a.lens[bc] : a.T;
      p_b[w & 3] = b;
      p_t[w & 3] = f0 - b * a.T;                                   // frame within its utterance
      p_lim[w & 3] = l < a.T ? l : a.T;
    }
  }
  const int g_step = BK * a.ldg * 2, x_step = BK * a.ldx * 2;
  int l_rel = 0;                                                 // steps fetched so far (relative to step_lo)
  // all 16 pieces of relative step `rel` into stage `buf`; call with consecutive rel (the counters advance)
  auto dma_piece = [&](int buf, int w, int rel) __attribute__((always_inline)) {
    const int arr = w >> 2, pr = 4 * (w & 3) + wave;
    const int f = (step_lo + rel) * BK + p_k[w];
    int ok = -(int)(f < a.R);                                     // the GY frame exists
    if (arr >= 2) {
      const int ts = p_t[w & 3] + shift;                          // partner frame, counted within the utterance
      ok &= -(int)((unsigned)ts < (unsigned)p_lim[w & 3]);
    }
    const int vo = ((p_off[w] + rel * (arr >= 2 ? x_step : g_step)) & ok) | (OOB & ~ok);
    dma16(arr == 0 ? rGh : arr == 1 ? rGl : arr == 2 ? rXh : rXl, (lds_u32_ptr)(sm + buf * STAGE + arr * ARR + pr * 1024), vo);
  };
  auto advance_t = [&]() __attribute__((always_inline)) {        // the X pieces' rows move on by one K step
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int t = p_t[w] + BK;
      const int wrap = t >= a.T ? 1 : 0;                         // T >= 32: at most one utterance boundary per step
      t -= wrap ? a.T : 0;
      const int b = p_b[w] + wrap;
      p_t[w] = t;
      p_b[w] = b;
      p_lim[w] = lim_tab[b < nb ? b : nb - 1];
    }
  };

  f32x16 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nsteps > 0) {
#pragma unroll
    for (int w = 0; w < 16; ++w) dma_piece(0, w, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      const bool more = s + 1 < nsteps;
      if (more) {                                                  // (uniform) the counters follow the stage being fetched
        advance_t();
        l_rel = s + 1;
      }
      const unsigned char* st = sm + buf * STAGE;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        Frag fbh[2], fbl[2], fa[2][2];                             // B fragments of the k block; A fragments, two slots
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          frag_issue(fbh[j], frag_addr(st + 2 * ARR, 2 * wave + j, kb, lane));
          frag_issue(fbl[j], frag_addr(st + 3 * ARR, 2 * wave + j, kb, lane));
        }
        frag_issue(fa[0][0], frag_addr(st, 0, kb, lane));
        frag_issue(fa[0][1], frag_addr(st + ARR, 0, kb, lane));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int sl = i & 1;
          if (i + 1 < 8) {                                          // next row block's fragments ahead of this one's MFMAs
            frag_issue(fa[sl ^ 1][0], frag_addr(st, i + 1, kb, lane));
            frag_issue(fa[sl ^ 1][1], frag_addr(st + ARR, i + 1, kb, lane));
          }
          if (i == 0) {
            frag_wait4<4>(fbh[0], fbh[1], fbl[0], fbl[1]);
            frag_wait2<4>(fa[0][0], fa[0][1]);
          } else if (i + 1 < 8) {
            frag_wait2<4>(fa[sl][0], fa[sl][1]);
          } else {
            frag_wait2<0>(fa[sl][0], fa[sl][1]);
          }
          const f16x8 ah = frag_val(fa[sl][0]), al = frag_val(fa[sl][1]);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8 bh = frag_val(fbh[j]), bl = frag_val(fbl[j]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
          }
          if (kb == 0 && more) {                                    // two DMA pieces of the next stage per row block
            dma_piece(buf ^ 1, 2 * i, l_rel);
            dma_piece(buf ^ 1, 2 * i + 1, l_rel);
          }
        }
      }
      __syncthreads();
    }
  }
  float* P = a.P + (long long)split * a.split_stride + (long long)tap * a.Mc * a.ldp;
  // accumulators straight to P, 128 contiguous bytes per half wave and row (wgrad_rm8.hip has the measurement)
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

// workgroup tiles per split (the caller picks `splits` so that tiles * splits fills the CUs: one workgroup per CU)
extern "C" int radmmm_wgrad_rm_tiles(int Mc, int Nc, int taps) {
  if (Mc <= 0 || Nc <= 0 || taps <= 0) return 0;
  return ((Mc + TM - 1) / TM) * ((Nc + TN - 1) / TN) * taps;
}

// GYh/GYl [R][ldg], Xh/Xl [R][ldx]: row-major fp16 split pairs (hi, lo) of scale_g * gy and of x; R = B * T rows,
// utterance-major.  x_mask: X rows at frames >= lens[b] read as zeros.  P [splits][taps][Mc][ldp] fp32 partial slabs
// (split_stride floats apart), to be summed by the caller (radmmm_weightnorm_bwd does).  ldg, ldx multiples of 8,
// 16-byte aligned operands, T >= 32, at most 1024 utterances.
extern "C" int radmmm_wgrad_rm(const void* GYh, const void* GYl, int ldg, const void* Xh, const void* Xl, int ldx, int R, int T,
                               const int32_t* lens, int x_mask, float* P, int ldp, int64_t split_stride, int Mc, int Nc,
                               int taps, int dil, int splits, float acc_scale, radmmm_stream_t stream) {
  RADMMM_REQUIRE(GYh && GYl && Xh && Xl && P, "wgrad_rm: null pointer");
  RADMMM_REQUIRE(Mc > 0 && Nc > 0 && taps >= 1 && dil >= 1 && splits >= 1 && R > 0 && T > 0 && R % T == 0 && ldg >= Mc &&
                     ldx >= Nc && ldg % 8 == 0 && ldx % 8 == 0 && ldp >= Nc && T >= BK && R / T <= 1024,
                 "wgrad_rm: bad dims (ldg, ldx %% 8 == 0, R = B * T, T >= 32, B <= 1024)");
  RADMMM_REQUIRE(radmmm::aligned16(GYh) && radmmm::aligned16(GYl) && radmmm::aligned16(Xh) && radmmm::aligned16(Xl),
                 "wgrad_rm: 16-byte aligned operands");
  const long long g_bytes = (long long)R * ldg * 2, x_bytes = (long long)R * ldx * 2;
  RADMMM_REQUIRE(g_bytes < 0x7fffffffLL && x_bytes < 0x7fffffffLL, "wgrad_rm: operand >= 2 GiB");
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_rm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SMEM + 4096);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(wgrad_rm): %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  RmArgs a;
  a.GYh = static_cast<const _Float16*>(GYh); a.GYl = static_cast<const _Float16*>(GYl);
  a.Xh = static_cast<const _Float16*>(Xh); a.Xl = static_cast<const _Float16*>(Xl);
  a.lens = lens; a.x_mask = x_mask;
  a.R = R; a.T = T; a.ldg = ldg; a.ldx = ldx; a.Mc = Mc; a.Nc = Nc; a.taps = taps; a.dil = dil; a.splits = splits;
  a.P = P; a.ldp = ldp; a.split_stride = split_stride; a.acc_scale = acc_scale;
  a.g_bytes = (int)g_bytes; a.x_bytes = (int)x_bytes;
  const int grid = radmmm_wgrad_rm_tiles(Mc, Nc, taps) * splits;
  hipLaunchKernelGGL(wgrad_rm_kernel, dim3(grid), dim3(256), SMEM + 4096, static_cast<hipStream_t>(stream), a);
  return radmmm::check_launch("wgrad_rm");
}
