// Prototype of the weight-gradient GEMM on ROW-MAJOR split operands (the [frames][channels] hi/lo fp16 pairs the GEMM
// epilogues already write), transposing in the LDS read with ds_read_b64_tr_b16 -- the kernel that would make the
// transposed zero-gapped copies (transpose_split_act: 88 launches, 2.4 ms per step) unnecessary.
// Same tile machine as csrc/wgrad_h3.hip: one workgroup per CU, 256 x 256 output tile, 4 waves x (8 x 2) accumulators,
// LDS-DMA double buffering, three f16 MFMA products (hi.lo + lo.hi + hi.hi), split-K over the frames, one tap per
// workgroup.  What differs:
//   * a K step is 32 FRAMES: each DMA instruction brings two frame rows of 256 channels (2 x 512 B) of one array; the
//     16-byte pieces of frame row k land XOR-ed by (k & 3) at 64-byte granularity, so that the four rows a 16-lane
//     group of the transposing read touches fall into four different 64-byte bank groups;
//   * an MFMA operand fragment (8 consecutive k of one channel) is two ds_read_b64_tr_b16 (4 k each) instead of one
//     ds_read_b128: twice the LDS read instructions for the same bytes;
//   * a tap shift is a ROW offset of the X operand (this prototype applies it with a bound check on the frame index
//     only: no utterance-boundary / length predicates, which cost a few integer ops per DMA piece and step).
//   P[tap][m][n] = sum_f GY[f][m] * X[f + (tap - taps/2) * dil][n]      (frames outside [0, R) contribute nothing)
// Checks itself against a plain fp32 kernel at a small size, then times the benchmark shapes (R = 12 800, 1024 x 1024,
// 1 tap and 5 taps) and prints TFLOP/s next to the figures of the production path to compare with
// (wgrad_h3 on transposed copies: profiles/r02_g_kernel_stats.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/wgrad_rm_probe.hip -o /tmp/wgrad_rm_probe && /tmp/wgrad_rm_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
      return 1;                                                 \
    }                                                           \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h4 __attribute__((vector_size(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;
typedef __attribute__((address_space(3))) h4* lds_h4_ptr;

constexpr int BK = 32, TM = 256, TN = 256;          // frames per K step, output tile
constexpr int ARR = BK * TM * 2;                     // bytes of one operand array in a stage: 32 rows x 512 B
constexpr int STAGE = 4 * ARR;                       // GYh, GYl, Xh, Xl
constexpr int SMEM = 2 * STAGE;
constexpr int OOB = 0x7fffffff;
constexpr int SGB_VMEM = 0x020, SGB_MFMA = 0x008, SGB_DSR = 0x100;

struct Args {
  const _Float16 *GYh, *GYl, *Xh, *Xl;   // [R][ld] row-major
  int R, ldg, ldx, Mc, Nc, taps, dil, splits;
  float* P;                              // [splits][taps][Mc][Nc]
  int g_bytes, x_bytes;
  int no_store;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

// 8 consecutive k (frames) of channel (lane & 31) of a 32-channel unit: two ds_read_b64_tr_b16 (k 0..3 | 4..7 of the
// lane's half of the k block).  Issued through inline asm: the compiler models the builtin form as a read of ALL of LDS
// and parks an s_waitcnt vmcnt(0) behind every LDS-DMA instruction in front of it (the whole DMA latency, eight times per
// K step: 517 us instead of the figure below); the asm form leaves the waits to us (frag_wait ties them to the registers).
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct Frag { i32x2 lo, hi; };
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

__global__ __launch_bounds__(256, 1) void wgrad_rm_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntm = (a.Mc + TM - 1) / TM, ntn = (a.Nc + TN - 1) / TN;
  int id = blockIdx.x;
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
  // row pair 4 * (w & 3) + wave ... dealt so that every wave serves all four arrays
  const int d_half = lane >> 5, d_unit = (lane & 31) >> 2, d_p16 = lane & 3;
  const __amdgpu_buffer_rsrc_t rGh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYh), 0, a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rGl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.GYl), 0, a.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Xh), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rXl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.Xl), 0, a.x_bytes, 0x00020000);
  // per-piece constants: frame offset within a K step (tap shift included for X), byte offset of this lane's 16 bytes
  int p_k[16], p_base[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int arr = w >> 2, pr = 4 * (w & 3) + wave;             // row pair 0..15 of array arr
    const int k = 2 * pr + d_half;
    const int u = d_unit ^ (k & 3);                              // source 64-byte unit that lands at d_unit
    const bool isx = arr >= 2;
    const int c0 = isx ? n0 : m0, C = isx ? a.Nc : a.Mc, ld = isx ? a.ldx : a.ldg;
    const int ch = c0 + u * 32 + d_p16 * 8;
    p_k[w] = k + (isx ? shift : 0);
    p_base[w] = ch < C ? (p_k[w] * ld + ch) * 2 : OOB;
  }
  const int g_step = BK * a.ldg * 2, x_step = BK * a.ldx * 2;
  auto dma_piece = [&](int buf, int w, int step) __attribute__((always_inline)) {
    const int arr = w >> 2, pr = 4 * (w & 3) + wave;
    const int f = step * BK + p_k[w];
    const int ok = -(int)((unsigned)f < (unsigned)a.R);           // all ones when the frame exists
    const int vo = ((p_base[w] + step * (arr >= 2 ? x_step : g_step)) & ok) | (OOB & ~ok);
    dma16(arr == 0 ? rGh : arr == 1 ? rGl : arr == 2 ? rXh : rXl, (lds_u32_ptr)(sm + buf * STAGE + arr * ARR + pr * 1024), vo);
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
    for (int w = 0; w < 16; ++w) dma_piece(0, w, step_lo);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      const int nxt = step_lo + (s + 1 < nsteps ? s + 1 : s);
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
          if (kb == 0) {                                            // two DMA pieces of the next stage per row block
            dma_piece(buf ^ 1, 2 * i, nxt);
            dma_piece(buf ^ 1, 2 * i + 1, nxt);
          }
        }
      }
      __syncthreads();
    }
  }
  // plain epilogue: D[m][n], m = row index from A (e, lane >> 5), n = lane & 31
  float* P = a.P + ((long long)split * a.taps + tap) * a.Mc * a.Nc;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int n = n0 + 64 * wave + 32 * j + (lane & 31);
        if (m < a.Mc && n < a.Nc && (!a.no_store || acc[i][j][e] == 12345.f)) P[(long long)m * a.Nc + n] = acc[i][j][e];
      }
}

// reference: one thread per output element, fp32 from the split pairs
__global__ void ref_kernel(const Args a, float* out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.taps * a.Mc * a.Nc;
  if (idx >= total) return;
  const int n = (int)(idx % a.Nc), m = (int)((idx / a.Nc) % a.Mc), tap = (int)(idx / ((long long)a.Nc * a.Mc));
  const int shift = (tap - a.taps / 2) * a.dil;
  float s = 0.f;
  for (int f = 0; f < a.R; ++f) {
    const int fx = f + shift;
    if (fx < 0 || fx >= a.R) continue;
    const float g = (float)a.GYh[(long long)f * a.ldg + m] + (float)a.GYl[(long long)f * a.ldg + m];
    const float x = (float)a.Xh[(long long)fx * a.ldx + n] + (float)a.Xl[(long long)fx * a.ldx + n];
    s += g * x;
  }
  out[idx] = s;
}

static void fill_split(std::vector<_Float16>& h, std::vector<_Float16>& l, size_t n, unsigned seed, float scale) {
  h.resize(n);
  l.resize(n);
  unsigned s = seed;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const float v = (((s >> 8) & 0xffff) / 65536.f - 0.5f) * scale;
    h[i] = (_Float16)v;
    l[i] = (_Float16)(v - (float)h[i]);
  }
}

int run(int R, int Mc, int Nc, int taps, int dil, int splits, bool check, int reps) {
  Args a{};
  a.R = R; a.Mc = Mc; a.Nc = Nc; a.taps = taps; a.dil = dil; a.splits = splits; a.ldg = Mc; a.ldx = Nc;
  std::vector<_Float16> gh, gl, xh, xl;
  fill_split(gh, gl, (size_t)R * Mc, 1, 2.f);
  fill_split(xh, xl, (size_t)R * Nc, 2, 2.f);
  a.g_bytes = R * Mc * 2; a.x_bytes = R * Nc * 2;
  _Float16 *dgh, *dgl, *dxh, *dxl;
  CK(hipMalloc(&dgh, a.g_bytes)); CK(hipMalloc(&dgl, a.g_bytes)); CK(hipMalloc(&dxh, a.x_bytes)); CK(hipMalloc(&dxl, a.x_bytes));
  CK(hipMemcpy(dgh, gh.data(), a.g_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dgl, gl.data(), a.g_bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(dxh, xh.data(), a.x_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dxl, xl.data(), a.x_bytes, hipMemcpyHostToDevice));
  a.GYh = dgh; a.GYl = dgl; a.Xh = dxh; a.Xl = dxl;
  const size_t pn = (size_t)splits * taps * Mc * Nc;
  CK(hipMalloc(&a.P, pn * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_rm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
  const int grid = ((Mc + TM - 1) / TM) * ((Nc + TN - 1) / TN) * taps * splits;
  hipLaunchKernelGGL(wgrad_rm_kernel, dim3(grid), dim3(256), SMEM, 0, a);
  CK(hipDeviceSynchronize());
  if (check) {
    float* ref;
    const size_t on = (size_t)taps * Mc * Nc;
    CK(hipMalloc(&ref, on * 4));
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((on + 255) / 256)), dim3(256), 0, 0, a, ref);
    std::vector<float> hp(pn), hr(on);
    CK(hipMemcpy(hp.data(), a.P, pn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, on * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < on; ++i) {
      double s = 0;
      for (int sp = 0; sp < splits; ++sp) s += hp[(size_t)sp * on + i];
      num += (s - hr[i]) * (s - hr[i]);
      den += (double)hr[i] * hr[i];
    }
    printf("{\"check\": \"R=%d %dx%d taps=%d dil=%d splits=%d\", \"rel_err\": %.3e}\n", R, Mc, Nc, taps, dil, splits, std::sqrt(num / den));
    CK(hipFree(ref));
  }
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wgrad_rm_kernel, dim3(grid), dim3(256), SMEM, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    a.no_store = 1;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wgrad_rm_kernel, dim3(grid), dim3(256), SMEM, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms2 = 0;
    CK(hipEventElapsedTime(&ms2, e0, e1));
    printf("{\"without_epilogue_stores_us\": %.1f}\n", ms2 * 1e3 / reps);
    const double us = ms * 1e3 / reps, flop = 2.0 * R * Mc * Nc * taps;
    printf("{\"shape\": \"R=%d %dx%d taps=%d dil=%d splits=%d grid=%d\", \"us\": %.1f, \"fp32_equiv_tflops\": %.1f, \"f16_mfma_tflops\": %.1f}\n",
           R, Mc, Nc, taps, dil, splits, grid, us, flop / us / 1e6, 3 * flop / us / 1e6);
  }
  CK(hipFree(dgh)); CK(hipFree(dgl)); CK(hipFree(dxh)); CK(hipFree(dxl)); CK(hipFree(a.P));
  return 0;
}

int main() {
  if (run(700, 256, 256, 3, 2, 2, true, 0)) return 1;          // ragged K range, shifts, split-K
  if (run(1000, 320, 288, 5, 1, 3, true, 0)) return 1;         // partial tiles in both dimensions, odd shifts
  if (run(12800, 1024, 1024, 1, 1, 16, false, 20)) return 1;   // res_skip weight gradient: 16 tiles x 16 splits
  if (run(12800, 1024, 1024, 5, 2, 3, false, 20)) return 1;    // in_layer weight gradient: 80 tiles x 3 splits
  return 0;
}
