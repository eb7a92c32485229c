// rowgemm16: radmmm_rowgemm_f32 tiled at 16-row granularity on v_mfma_f32_16x16x4_f32.
//
// Why a second tiling.  The conv GEMMs of the flow step have M = B*T' rows (12 800 at B=32,
// T=800) and N = 1024 columns.  With fixed 128x128 tiles that is 800 equal workgroups on
// 256 CUs x 2 resident workgroups: 1.56 rounds, i.e. the busiest CU does 4 tiles while the
// average is 3.1 (78 % of the machine).  fp32 MFMA is slow enough (64 FLOP/clk/SIMD) that
// operand re-use is not the constraint, so the tile may be tall and skinny: here a workgroup
// owns ROWS x 128 outputs with ROWS = 16*MT chosen by the host so that (#m-tiles x #n-tiles)
// is a whole number of rounds of the 512 workgroup slots (12 800 rows -> 64 m-tiles of 200 rows
// -> MT = 13 -> 512 workgroups, 96 % of the MFMA slots useful instead of 78 %).
//
// Geometry: 256 threads = 4 waves; every wave owns ALL MT row-subtiles of 16 and a 32-column
// slice (two 16-wide column-subtiles): MT*2 accumulators of 4 registers.  Per K step of 16:
// 4 k-quads x MT x 2 MFMAs (32 cycles each); fragments are single ds_read_b32 from k-major LDS
// tiles whose row pitch is == 16 (mod 32) floats, so the two k-rows a half-wave touches fall in
// disjoint bank halves.  Staging, masking, double buffering and the fused epilogue are those of
// the 128x128 kernel (gemm_f32.hip); the accumulator tile is staged out through LDS 64 rows at
// a time.
#include "common.h"
#include "rowgemm_epilogue.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define U2F(u) __builtin_bit_cast(float, (unsigned int)(u))

constexpr int BN = 128, BK = 16;
constexpr int LDB = 144;                       // 144 % 32 == 16
constexpr int SMEM_BYTES = 64 * 1024;          // fixed: caps residency at 2 workgroups per CU
constexpr int EP_LD = 132;                     // epilogue staging pitch (rows 4 apart -> 16 banks apart)

template <int MT> struct Geo {
  static constexpr int ROWS = MT * 16;
  static constexpr int LDA = ROWS + ((MT % 2 == 0) ? 16 : 0);   // == 16 (mod 32)
  static constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB;
  static constexpr int NI = (ROWS * 4 + 255) / 256;             // float4 chunks of A per thread
  static_assert(2 * (A_TILE + B_TILE) * 4 <= SMEM_BYTES, "LDS budget");
  static_assert(64 * EP_LD * 4 <= SMEM_BYTES, "epilogue staging");
};

__device__ __forceinline__ float4 sel4(float4 v, bool ok, int k, int K) {
  v.x = (ok && k + 0 < K) ? v.x : 0.f;
  v.y = (ok && k + 1 < K) ? v.y : 0.f;
  v.z = (ok && k + 2 < K) ? v.z : 0.f;
  v.w = (ok && k + 3 < K) ? v.w : 0.f;
  return v;
}

__device__ __forceinline__ int xcd_remap(int wg, int nt) {
  const int xcd = wg & 7, loc = wg >> 3;
  const int q = nt >> 3, r = nt & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Write row-subtiles 4*PASS .. 4*PASS+3 of a wave's accumulators to the [64][EP_LD] staging tile.
// C/D layout of 16x16: lane l, register e -> column l&15, row 4*(l>>4) + e.
template <int MT, int PASS>
__device__ __forceinline__ void stage_pass(float* smem, const f32x4 (&acc)[MT][2], int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    constexpr int base = PASS * 4;
    if (base + q < MT) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          smem[(q * 16 + 4 * (lane >> 4) + e) * EP_LD + wave * 32 + nt * 16 + (lane & 15)] =
              acc[(base + q < MT) ? base + q : 0][nt][e];
    }
  }
}

template <int MT, int B_LAYOUT>
__global__ __launch_bounds__(256, 2) void rowgemm16_kernel(const radmmm_rowgemm_desc p, const int rows_per_tile,
                                                            const int nmt, const int a_bytes, const int b_bytes) {
  using G = Geo<MT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BK][LDA]
  float* Bs = smem + 2 * G::A_TILE;        // [2][BK][LDB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, ntn * nmt);
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm * rows_per_tile, n0 = tn * BN;
  int m_end = m0 + rows_per_tile;
  if (m_end > p.M) m_end = p.M;

  const int kpt = (p.K + BK - 1) / BK;
  const int nsteps = kpt * p.taps;

  // ---- staging through BUFFER loads -----------------------------------------------------------
  // fp32 MFMA shares the SIMD's FMA datapath with ordinary VALU instructions (measured:
  // profiles/r01_mfma_valu_mix.txt), so the K loop must carry (almost) no vector ALU work.
  // Every global address is therefore split into a per-thread byte offset that only changes
  // when the TAP changes (a handful of VALU ops every K/16 steps) and a wave-uniform scalar
  // offset per K step (SALU); frames that fall outside their item / valid length get an
  // out-of-range offset, for which the buffer unit returns zeros in hardware -- no selects.
  // Requires K % 16 == 0 (checked by the host dispatcher).
  constexpr int OOB = 0x7fffffff;
  int a_row[G::NI], a_chunk[G::NI], a_t[G::NI], a_lim[G::NI], a_base[G::NI], a_voff[G::NI];
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    const int idx = tid + i * 256;
    a_chunk[i] = idx / G::ROWS;
    a_row[i] = idx - a_chunk[i] * G::ROWS;
    const int r = m0 + a_row[i];
    a_t[i] = 0;
    a_lim[i] = 0;                           // lim 0 -> never valid -> zeros
    a_base[i] = 0;
    a_voff[i] = OOB;
    if (a_chunk[i] < 4 && r < m_end) {
      const int b = r / p.T;
      a_t[i] = r - b * p.T;
      a_lim[i] = (p.a_mask_mode && p.lens) ? p.lens[b] : p.T;
      const long long item = p.a_item_stride ? (long long)b * p.a_item_stride : (long long)b * p.T * p.lda;
      a_base[i] = (int)((item + a_chunk[i] * 4) * 4);      // bytes; buffers are < 2 GiB (host check)
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.A), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.B), 0, b_bytes, 0x00020000);
  const int b_row = tid & 127, b_kc = tid >> 7;      // layout 0: row n, 16-B chunks b_kc + 2*i
  const int b_n4 = tid & 31, b_k0 = tid >> 5;        // layout 1: 4 n at b_n4*4, k rows b_k0 + 8*i
  int b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (B_LAYOUT == 0) {
      const int n = n0 + b_row;
      b_voff[i] = n < p.N ? (n * p.ldb + (b_kc + 2 * i) * 4) * 4 : OOB;
    } else {
      const int n = n0 + b_n4 * 4;
      b_voff[i] = n < p.N ? ((b_k0 + 8 * i) * p.ldb + n) * 4 : OOB;
    }
  }
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    const int s = p.sign * (tap - p.taps / 2) * p.dil;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
      const int ts = a_t[i] + s;
      a_voff[i] = (ts >= 0 && ts < a_lim[i]) ? a_base[i] + ts * p.lda * 4 : OOB;
    }
  };

  struct Regs {
    u32x4 a[G::NI];
    u32x4 b[2];
  };
  auto load_tiles = [&](int step, Regs& R) __attribute__((always_inline)) {
    const int tap = step / kpt, kb = step - tap * kpt;
    if (kb == 0) set_tap(tap);               // uniform branch, once per tap
    const int so_a = kb * (BK * 4);
    const int so_b = B_LAYOUT == 0 ? (int)(tap * p.b_tap_stride * 4) + kb * (BK * 4)
                                   : (int)(tap * p.b_tap_stride * 4) + kb * (BK * 4) * p.ldb;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) R.a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], so_a, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) R.b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[i], so_b, 0);
  };
  // Write part `part` (of 4) of a register set into LDS buffer `buf`: pure ds_write traffic.
  auto store_part = [&](int buf, const Regs& R, const int part) __attribute__((always_inline)) {
    float* as = As + buf * G::A_TILE;
    float* bs = Bs + buf * G::B_TILE;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
      if (i == part && a_chunk[i] < 4) {
        float* d = as + (a_chunk[i] * 4) * G::LDA + a_row[i];
        d[0] = U2F(R.a[i].x);
        d[G::LDA] = U2F(R.a[i].y);
        d[2 * G::LDA] = U2F(R.a[i].z);
        d[3 * G::LDA] = U2F(R.a[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == part) {
        if (B_LAYOUT == 0) {
          float* d = bs + (b_kc + 2 * i) * 4 * LDB + b_row;
          d[0] = U2F(R.b[i].x);
          d[LDB] = U2F(R.b[i].y);
          d[2 * LDB] = U2F(R.b[i].z);
          d[3 * LDB] = U2F(R.b[i].w);
        } else {
          *reinterpret_cast<u32x4*>(bs + (b_k0 + 8 * i) * LDB + b_n4 * 4) = R.b[i];
        }
      }
    }
  };

  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane l reads element [k = l>>4][i = l&15]
  const int fa = (lane >> 4) * G::LDA + (lane & 15);
  const int fb = (lane >> 4) * LDB + wave * 32 + (lane & 15);

  // One K step.  Software pipeline, prefetch distance 2: while the MFMAs of tile `step` run from
  // LDS buffer step&1, the loads of tile step+2 are in flight into Rload and tile step+1 (loaded
  // during the previous step, register set Rstore) is written to the other LDS buffer in four
  // parts slotted between the four k-quads of MFMAs.  The only work left outside the MFMA
  // stream is the address arithmetic, one barrier and the first fragment read of the next step.
  auto kstep = [&](int step, Regs& Rload, const Regs& Rstore) __attribute__((always_inline)) {
    const int buf = step & 1;
    const int t2 = step + 2 < nsteps ? step + 2 : nsteps - 1;
    load_tiles(t2, Rload);
    const float* as = As + buf * G::A_TILE + fa;
    const float* bs = Bs + buf * G::B_TILE + fb;
    float a[MT], bq[2][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = as[mt * 16];
    bq[0][0] = bs[0];
    bq[0][1] = bs[16];
    // issue the whole first k-quad's fragment reads as one batch right after the barrier (one
    // exposed LDS latency instead of one per MFMA group: the compiler otherwise sinks them)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < BK / 4; ++kq) {
      const int cur = kq & 1, nx = cur ^ 1;
      if (kq + 1 < BK / 4) {
        bq[nx][0] = bs[(kq + 1) * 4 * LDB];
        bq[nx][1] = bs[(kq + 1) * 4 * LDB + 16];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bq[cur][0], acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bq[cur][1], acc[mt][1], 0, 0, 0);
        // rolling refill: a[mt] is dead once its two MFMAs have issued
        if (kq + 1 < BK / 4) a[mt] = as[(kq + 1) * 4 * G::LDA + mt * 16];
      }
      store_part(buf ^ 1, Rstore, kq);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };

  Regs RA, RB;
  load_tiles(0, RA);
#pragma unroll
  for (int part = 0; part < 4; ++part) store_part(0, RA, part);
  load_tiles(nsteps > 1 ? 1 : 0, RB);
  __syncthreads();
  for (int step = 0; step < nsteps; step += 2) {
    kstep(step, RA, RB);
    if (step + 1 < nsteps) kstep(step + 1, RB, RA);
  }

  // ---- epilogue: 64 rows (4 row-subtiles) at a time through LDS ------------------------------
  // (passes are expanded at compile time: a runtime-indexed accumulator array would be demoted
  //  to scratch memory for the whole kernel)
  const radmmm::EpilogueCtx ec(p);
  auto drain = [&](int pass) {
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    for (int i = 0; i < 8; ++i) {
      const int rl = i * 8 + (tid >> 5);                 // 0..63
      const int row = m0 + pass * 64 + rl;
      if (pass * 64 + rl < G::ROWS && row < m_end) {
        const float4 a4 = *reinterpret_cast<const float4*>(smem + rl * EP_LD + c4);
        radmmm::epilogue_store4(p, ec, row, n0 + c4, a4);
      }
    }
    __syncthreads();
  };
  stage_pass<MT, 0>(smem, acc, wave, lane);
  drain(0);
  if constexpr (MT > 4) {
    stage_pass<MT, 1>(smem, acc, wave, lane);
    drain(1);
  }
  if constexpr (MT > 8) {
    stage_pass<MT, 2>(smem, acc, wave, lane);
    drain(2);
  }
  if constexpr (MT > 12) {
    stage_pass<MT, 3>(smem, acc, wave, lane);
    drain(3);
  }
}

template <int MT, int BL>
int launch_one(const radmmm_rowgemm_desc& d, int rows_per_tile, int nmt, int ntn, int a_bytes, int b_bytes,
               hipStream_t s) {
  static int once = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm16_kernel<MT, BL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) {
      radmmm::set_error("hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
      return -2;
    }
    return 0;
  }();
  if (once) return once;
  hipLaunchKernelGGL((rowgemm16_kernel<MT, BL>), dim3(nmt * ntn), dim3(256), SMEM_BYTES, s, d, rows_per_tile, nmt,
                     a_bytes, b_bytes);
  return radmmm::check_launch("rowgemm16_f32");
}

template <int BL>
int launch_mt(int MT, const radmmm_rowgemm_desc& d, int rows_per_tile, int nmt, int ntn, int a_bytes, int b_bytes,
              hipStream_t s) {
  switch (MT) {
    case 4: return launch_one<4, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 6: return launch_one<6, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 8: return launch_one<8, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 10: return launch_one<10, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 12: return launch_one<12, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 13: return launch_one<13, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    case 14: return launch_one<14, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);
    default: return launch_one<15, BL>(d, rows_per_tile, nmt, ntn, a_bytes, b_bytes, s);   // MT=16 would spill (256 VGPRs)
  }
}

const int kMT[] = {4, 6, 8, 10, 12, 13, 14, 15};

}  // namespace

// Host-side tiling choice: minimise (rounds of 512 workgroup slots) x (MFMA rows per workgroup
// + fixed per-step overhead), i.e. the modelled time of the busiest CU.
int radmmm::launch_rowgemm16(const radmmm_rowgemm_desc& d, hipStream_t stream) {
  const int ntn = (d.N + BN - 1) / BN;
  const int slots = 512;
  double best_cost = 1e300;
  int best_mt = 15, best_rows = 240, best_nmt = (d.M + 239) / 240;
  for (int mt : kMT) {
    const int cap = mt * 16;
    const int nmt_min = (d.M + cap - 1) / cap;
    // try the minimal tile count for this MT and the next few counts that complete a round
    for (int extra = 0; extra < 3; ++extra) {
      int nmt = nmt_min;
      if (extra) {
        const long long wg = (long long)nmt_min * ntn;
        const long long rounds = (wg + slots - 1) / slots + (extra - 1);
        nmt = (int)((rounds * slots) / ntn);
        if (nmt < nmt_min) continue;
      }
      if (nmt > d.M) nmt = d.M;
      const int rows = (d.M + nmt - 1) / nmt;
      if (rows > cap) continue;
      nmt = (d.M + rows - 1) / rows;
      const long long wg = (long long)nmt * ntn;
      const double rounds = (double)((wg + slots - 1) / slots);
      // a partially filled single round costs the same as a full one for the busiest CU, but
      // with < 256 workgroups only one workgroup runs per CU (no sharing of the MFMA pipe)
      const double share = wg <= 256 ? 0.6 : 1.0;
      const double cost = rounds * share * (mt + 1.5);
      if (cost < best_cost - 1e-9) {
        best_cost = cost;
        best_mt = mt;
        best_rows = rows;
        best_nmt = nmt;
      }
    }
  }
  // byte extents of the operands for the buffer descriptors
  const long long items = d.M / d.T;
  const long long a_last = d.a_item_stride ? (items - 1) * d.a_item_stride + (long long)(d.T - 1) * d.lda
                                           : (long long)(d.M - 1) * d.lda;
  const long long a_bytes = (a_last + d.K) * 4;
  // (measured on gfx950: the scalar offset IS part of the hardware range check, so the descriptor
  //  spans the whole operand and invalid elements are flagged through the vector offset alone)
  const long long b_bytes = ((long long)(d.taps - 1) * d.b_tap_stride +
                             (d.b_layout == 0 ? (long long)d.N * d.ldb : (long long)d.K * d.ldb)) * 4;
  if (a_bytes >= 0x7fffffffLL || b_bytes >= 0x7fffffffLL) return 1;   // caller falls back to the generic kernel
  if (d.b_layout == 0) return launch_mt<0>(best_mt, d, best_rows, best_nmt, ntn, (int)a_bytes, (int)b_bytes, stream);
  return launch_mt<1>(best_mt, d, best_rows, best_nmt, ntn, (int)a_bytes, (int)b_bytes, stream);
}
