// rowgemm_mix: the flow steps' 1x1 channel mix  C[r][:] = A[r][:] . W (+ bias)  with a SMALL SQUARE weight (N = K = 160: the
// padded channel count of the squeezed mel, common.py:507-548 / 551-617) in exact fp32 on v_mfma_f32_16x16x4_f32 -- round 5.
//
// Why a kernel of its own: radmmm_rowgemm_f32's tilings are built for the 1024-wide conv GEMMs (ten double-buffered K steps
// with barriers, accumulators staged out through LDS): 26 us per launch for 0.66 GFLOP and 16 MB, sixteen launches per step
// (forward mix + its data gradient in every flow step).  Here the WHOLE weight (100 KB) sits in LDS for the life of the
// workgroup, a workgroup owns 64 rows, and the K loop is 40 uninterrupted k quads with no barrier in it:
//   8 waves = 4 row subtiles of 16 x 2 column halves of 80 (five 16-wide tiles: 20 accumulator registers);
//   per k quad and wave: one A fragment (ds_read_b32) + five B fragments + five MFMAs.
// LDS: W in ITS OWN layout -- B[k][n] rows with pitch 176 floats (== 16 mod 32: the four k rows of a fragment read fall into
// two disjoint bank halves) or B[n][k] rows with pitch 164 (== 4 mod 32: rows 8 apart share a bank) -- and the 64 A rows with
// pitch 164: every fragment read hits every bank exactly twice with 64 lanes, the minimum; 154.6 KB.
// Same k order as the generic kernels (0 .. 159 ascending into one accumulator per output): results agree to the last bits.
// Scope (mix_ok): N == K == 160, one tap, no mask / ratio / activation / add / second output / split copy; bias optional;
// b_layout 0 (B[n][k]: forward, W_eff rows = output channels) or 1 (B[k][n]: the data gradient).  Everything else keeps the
// generic kernels.  RADMMM_MIX=0 (RADMMM_DEBUG) disables it (A/B runs).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NK = 160, ROWS = 64, LDW = 176, LDA = 164;
constexpr int SMEM = (NK * LDW + ROWS * LDA) * 4;

template <int BL>
__global__ __launch_bounds__(512) void rowgemm_mix_kernel(const radmmm_rowgemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sW = sm;                   // the weight in its own layout (below); the A rows start behind the larger of the two
  float* sA = sm + NK * LDW;        // [row][k], pitch LDA
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * ROWS;
  // ---- stage W (as [k][n]) and this workgroup's 64 rows of A
  // (the weight keeps its own layout: B[k][n] rows with pitch 176, B[n][k] rows with pitch 164 -- either way a fragment read
  //  touches every bank exactly twice, and the copy is plain 16-byte rows)
  constexpr int PW = BL == 1 ? LDW : LDA;
  for (int i = tid; i < NK * (NK / 4); i += 512) {
    const int r = i / (NK / 4), c4 = (i - r * (NK / 4)) * 4;
    *reinterpret_cast<float4*>(sW + r * PW + c4) = *reinterpret_cast<const float4*>(p.B + (long long)r * p.ldb + c4);
  }
  for (int i = tid; i < ROWS * (NK / 4); i += 512) {
    const int r = i / (NK / 4), c4 = (i - r * (NK / 4)) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < p.M) a = *reinterpret_cast<const float4*>(p.A + (long long)(m0 + r) * p.lda + c4);
    *reinterpret_cast<float4*>(sA + r * LDA + c4) = a;
  }
  __syncthreads();
  // ---- 40 k quads: A fragment lane l = A[16 rs + (l & 15)][4 q + (l >> 4)], B fragment = W[4 q + (l >> 4)][n0 + 16 t + (l & 15)]
  const int rs = wave & 3, ch = wave >> 2;
  const float* ap = sA + (16 * rs + (lane & 15)) * LDA + (lane >> 4);
  // B fragment of k quad q, column tile t: W[k = 4 q + (l >> 4)][n = 80 ch + 16 t + (l & 15)]
  const float* bp = BL == 1 ? sW + (lane >> 4) * LDW + 80 * ch + (lane & 15) : sW + (80 * ch + (lane & 15)) * LDA + (lane >> 4);
  constexpr int BQ = BL == 1 ? 4 * LDW : 4, BT = BL == 1 ? 16 : 16 * LDA;          // fragment strides per k quad / column tile
  f32x4 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int q = 0; q < NK / 4; ++q) {
    const float a = ap[4 * q];
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[q * BQ + t * BT], acc[t], 0, 0, 0);
  }
  __syncthreads();                  // every wave is done with sA: it becomes the staging tile of the outputs
  // C/D layout of 16x16: lane l, register e -> column l & 15, row 4 (l >> 4) + e
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) sA[(16 * rs + 4 * (lane >> 4) + e) * LDA + 80 * ch + 16 * t + (lane & 15)] = acc[t][e];
  __syncthreads();
  for (int i = tid; i < ROWS * (NK / 4); i += 512) {
    const int r = i / (NK / 4), c4 = (i - r * (NK / 4)) * 4;
    if (m0 + r < p.M) {
      float4 o = *reinterpret_cast<const float4*>(sA + r * LDA + c4);
      if (p.bias) {
        o.x += p.bias[c4]; o.y += p.bias[c4 + 1]; o.z += p.bias[c4 + 2]; o.w += p.bias[c4 + 3];
      }
      *reinterpret_cast<float4*>(p.C + (long long)(m0 + r) * p.ldc + c4) = o;
    }
  }
}

template <int BL>
int launch(const radmmm_rowgemm_desc& d, hipStream_t s) {
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm_mix_kernel<BL>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(rowgemm_mix): %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  hipLaunchKernelGGL(rowgemm_mix_kernel<BL>, dim3((d.M + ROWS - 1) / ROWS), dim3(512), SMEM, s, d);
  return radmmm::check_launch("rowgemm_mix");
}

}  // namespace

namespace radmmm {
// 0: launched; 1: not this kernel's shape (the caller goes on to the generic tilings); < 0: error
int launch_rowgemm_mix(const radmmm_rowgemm_desc& d, hipStream_t s) {
  static const bool off = [] {
    const char* e = debug_env("RADMMM_MIX");
    return e && e[0] == '0';
  }();
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool ok = !off && d.N == NK && d.K == NK && d.taps == 1 && d.a_item_stride == 0 && !d.a_mask_mode && !d.pconv && !d.premask &&
                  !d.postmask && !d.add && !d.dact && !d.rowscale && d.act == RADMMM_ACT_NONE && !d.C2 && !d.Ch && !d.C2h &&
                  !d.colsum_out && !d.n_c2_src && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.ldc % 4 == 0 && d.lda >= NK && d.ldb >= NK &&
                  d.ldc >= NK && a16(d.A) && a16(d.B) && a16(d.C) && (d.b_layout == 0 || d.b_layout == 1);
  if (!ok) return 1;
  return d.b_layout == 0 ? launch<0>(d, s) : launch<1>(d, s);
}
}  // namespace radmmm
