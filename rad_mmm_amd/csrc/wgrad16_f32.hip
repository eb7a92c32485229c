// wgrad16: fast path of radmmm_wgrad_f32 (weight gradient, contraction over frames) with a
// VALU-free K loop -- see rowgemm16_f32.hip for why fp32 MFMA needs that on gfx950.
//
//   P[split][tap][m][n] = sum_{r in split} GY[r, m] * Xm[r + shift(tap), n]
//
// 128x128 output tile per workgroup, 4 waves (2x2) x (2x2) v_mfma_f32_32x32x2_f32, K step = 16
// frames.  Requires T % 16 == 0 so that every K step lies inside ONE utterance: the item index,
// its valid length and the window of admissible frames are then wave-uniform scalars and the
// per-thread work per step is two compares and a select per load (the masked/shifted rows get
// an out-of-range buffer offset -> zeros from the buffer unit).  The shift of the tap is folded
// into the base address of the X descriptor, the K-step position into the scalar offset.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define U2F(u) __builtin_bit_cast(float, (unsigned int)(u))

constexpr int BK = 16, LDT = 128, TILE = BK * LDT;   // floats
constexpr int SMEM_BYTES = 4 * TILE * 4;              // 32 KiB (also the epilogue's [64][128] stage)
constexpr int OOB = 0x7fffffff;

__global__ __launch_bounds__(256, 3) void wgrad16_kernel(const radmmm_wgrad_desc p, const int gy_bytes,
                                                          const int x_bytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;              // [2][BK][LDT]
  float* Bs = smem + 2 * TILE;   // [2][BK][LDT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.Mc + 127) / 128, ntn = (p.Nc + 127) / 128;
  int id = blockIdx.x;
  const int tn = id % ntn; id /= ntn;
  const int tm = id % ntm; id /= ntm;
  const int tap = id % p.taps;
  const int split = id / p.taps;
  const int m0 = tm * 128, n0 = tn * 128;
  const int shift = (tap - p.taps / 2) * p.dil;

  const int steps_total = p.R / BK;
  const int steps_per = (steps_total + p.splits - 1) / p.splits;
  const int step_lo = split * steps_per;
  int step_hi = step_lo + steps_per;
  if (step_hi > steps_total) step_hi = steps_total;
  const int nsteps = step_hi - step_lo;

  const int c4 = tid & 31, k0 = tid >> 5;            // 4 columns at c4*4; k rows k0, k0+8
  const int am = m0 + c4 * 4, bn = n0 + c4 * 4;
  int a_voff[2], x_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_voff[i] = am < p.Mc ? ((k0 + 8 * i) * p.ldgy + am) * 4 : OOB;
    x_off[i] = bn < p.Nc ? ((k0 + 8 * i) * p.ldx + bn) * 4 : OOB;
  }
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.GY), 0, gy_bytes, 0x00020000);
  // base shifted by the tap: rows that would fall before/after the operand are flagged invalid
  // through the vector offset and never dereferenced
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.X) + (long long)shift * p.ldx, 0, x_bytes, 0x00020000);

  struct Regs {
    u32x4 a[2], b[2];
  };
  auto load_tiles = [&](int step, Regs& R) __attribute__((always_inline)) {
    const int r0 = step * BK;
    const int b = r0 / p.T;                       // scalar: one item per K step (T % 16 == 0)
    const int t0 = r0 - b * p.T;
    const int lim = (p.x_mask_mode && p.lens) ? p.lens[b] : p.T;
    const int lo = -(t0 + shift), hi = lim - t0 - shift;
    const int so_a = r0 * p.ldgy * 4, so_x = r0 * p.ldx * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = k0 + 8 * i;
      R.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], so_a, 0);
      R.b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (kr >= lo && kr < hi) ? x_off[i] : OOB, so_x, 0);
    }
  };
  auto store_part = [&](int buf, const Regs& R, const int part) __attribute__((always_inline)) {
    // parts 0,1: A rows k0, k0+8 ; parts 2,3: B rows
    if (part < 2) *reinterpret_cast<u32x4*>(As + buf * TILE + (k0 + 8 * part) * LDT + c4 * 4) = R.a[part];
    else *reinterpret_cast<u32x4*>(Bs + buf * TILE + (k0 + 8 * (part - 2)) * LDT + c4 * 4) = R.b[part - 2];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_off = (lane >> 5) * LDT + (lane & 31);
  auto kstep = [&](int s, Regs& Rload, const Regs& Rstore) __attribute__((always_inline)) {
    const int buf = s & 1;
    const int t2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
    load_tiles(step_lo + t2, Rload);
    const float* as = As + buf * TILE + wm * 64 + frag_off;
    const float* bs = Bs + buf * TILE + wn * 64 + frag_off;
    float a[2][2], b[2][2];
    a[0][0] = as[0]; a[0][1] = as[32];
    b[0][0] = bs[0]; b[0][1] = bs[32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cur = kk & 1, nx = cur ^ 1;
      if (kk + 1 < BK / 2) {
        a[nx][0] = as[(kk + 1) * 2 * LDT]; a[nx][1] = as[(kk + 1) * 2 * LDT + 32];
        b[nx][0] = bs[(kk + 1) * 2 * LDT]; b[nx][1] = bs[(kk + 1) * 2 * LDT + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][1], acc[1][1], 0, 0, 0);
      if (kk & 1) store_part(buf ^ 1, Rstore, kk >> 1);   // 4 parts after k-pairs 1,3,5,7
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };

  if (nsteps > 0) {
    Regs RA, RB;
    load_tiles(step_lo, RA);
#pragma unroll
    for (int part = 0; part < 4; ++part) store_part(0, RA, part);
    load_tiles(step_lo + (nsteps > 1 ? 1 : 0), RB);
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
      kstep(s, RA, RB);
      if (s + 1 < nsteps) kstep(s + 1, RB, RA);
    }
  }

  // ---- epilogue: two passes of 64 rows through the (now idle) 32 KiB of LDS ------------------
  float* P = p.P + (long long)split * p.split_stride + (long long)tap * p.Mc * p.ldp;
  const bool vec_ok = (p.ldp % 4 == 0) && radmmm::aligned16(p.P) && (p.split_stride % 4 == 0);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
      // MFMA 32x32 C/D layout: lane l, register e -> column l&31, row (e&3) + 8*(e>>2) + 4*(l>>5)
      float* base = smem + (4 * (lane >> 5)) * LDT + wn * 64 + (lane & 31);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            base[(mi * 32 + (e & 3) + 8 * (e >> 2)) * LDT + ni * 32] = acc[mi][ni][e];
    }
    __syncthreads();
    const int col = n0 + c4 * 4;
    for (int i = 0; i < 8; ++i) {
      const int rl = i * 8 + (tid >> 5);
      const int row = m0 + half * 64 + rl;
      if (row < p.Mc && col < p.Nc) {
        const float4 a4 = *reinterpret_cast<const float4*>(smem + rl * LDT + c4 * 4);
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
    __syncthreads();
  }
}

}  // namespace

// returns 0 launched, <0 error, 1 = not applicable (caller uses the generic kernel)
int radmmm::launch_wgrad16(const radmmm_wgrad_desc& d, hipStream_t stream) {
  if (d.T % BK != 0 || d.R % BK != 0) return 1;
  const long long gy_bytes = (long long)d.R * d.ldgy * 4, x_bytes = (long long)d.R * d.ldx * 4;
  if (gy_bytes >= 0x7fffffffLL || x_bytes >= 0x7fffffffLL) return 1;
  const int ntm = (d.Mc + 127) / 128, ntn = (d.Nc + 127) / 128;
  hipLaunchKernelGGL(wgrad16_kernel, dim3(ntm * ntn * d.taps * d.splits), dim3(256), SMEM_BYTES, stream, d,
                     (int)gy_bytes, (int)x_bytes);
  return radmmm::check_launch("wgrad16_f32");
}
