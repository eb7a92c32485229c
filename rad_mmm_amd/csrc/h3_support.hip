// Support kernels of the split-f16 GEMM path: weight-norm fold straight into split (hi/lo fp16)
// packed weights, and a tiled transpose of a split pair (the data-gradient GEMM wants the weights
// K-contiguous in the OUTPUT-channel index).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "split_pack.h"

namespace {

using radmmm::block_sum;

// W{h,l}[tap][co][col(ci)] = split(scale * g[co] * v[co][ci][tap] / ||v[co]||); columns not hit by
// col() must have been zeroed by the caller.  g == NULL: plain conv weights (no normalisation).
// One workgroup per output channel: the checkpoint row v[co][ci][tap] (tap fastest) is read once, coalesced, into LDS;
// each tap plane is then written as whole 4-column groups (8-byte fp16 stores, 4-byte stores of the 8-bit parts) with
// consecutive lanes on consecutive columns.  (The first version wrote element by element in [ci][tap] order: adjacent
// lanes hit different tap planes, 2-byte scattered stores -- 1.7 ms per step for 849 MB of weights.)
__device__ __forceinline__ void weightnorm_fwd_h3_row(
    const float* __restrict__ v, const float* __restrict__ g, _Float16* __restrict__ Wh, _Float16* __restrict__ Wl,
    float* __restrict__ inv_norm, int Cout, int Cin, int taps, int ldk, int perm_split, int off_lo, int off_hi,
    float scale, int fmt, float x8_mul, int vec_ok, int co, float* row, float* sh) {
  const int n = Cin * taps;
  const float* vr = v + (long long)co * n;
  float ss = 0.f;
  if ((n & 3) == 0 && radmmm::aligned16(vr)) {          // 16 bytes per lane: a 20 KB row in 5 instructions per thread, not 20
    for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
      const float4 x = *reinterpret_cast<const float4*>(vr + i);
      *reinterpret_cast<float4*>(row + i) = x;
      ss = fmaf(x.x, x.x, ss);
      ss = fmaf(x.y, x.y, ss);
      ss = fmaf(x.z, x.z, ss);
      ss = fmaf(x.w, x.w, ss);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float x = vr[i];
      row[i] = x;
      ss = fmaf(x, x, ss);
    }
  }
  float wn = 1.f;                      // g / ||v||  (1 for plain weights)
  if (g) {
    ss = block_sum(ss, sh);
    const float nrm = sqrtf(ss);
    if (threadIdx.x == 0) inv_norm[co] = 1.f / nrm;
    wn = g[co] / nrm;
  } else {
    __syncthreads();
  }
  // same rounding as the fp32 path (w = v * (g / ||v||)), then the exact power-of-two scale
  for (int k = 0; k < taps; ++k) {
    const long long base = ((long long)k * Cout + co) * ldk;
    if (vec_ok) {
      for (int ci = threadIdx.x * 4; ci < Cin; ci += blockDim.x * 4) {
        const int col = ci < perm_split ? ci + off_lo : ci - perm_split + off_hi;
        radmmm::store_split4_fmt(Wh, Wl, base, col, fmt, x8_mul, scale, row[ci * taps + k] * wn, row[(ci + 1) * taps + k] * wn,
                                 row[(ci + 2) * taps + k] * wn, row[(ci + 3) * taps + k] * wn);
      }
    } else {
      for (int ci = threadIdx.x; ci < Cin; ci += blockDim.x) {
        const int col = ci < perm_split ? ci + off_lo : ci - perm_split + off_hi;
        radmmm::store_split1_fmt(Wh, Wl, base, col, fmt, x8_mul, scale, row[ci * taps + k] * wn);
      }
    }
  }
}

__global__ __launch_bounds__(256) void weightnorm_fwd_h3_kernel(
    const float* __restrict__ v, const float* __restrict__ g, _Float16* __restrict__ Wh, _Float16* __restrict__ Wl,
    float* __restrict__ inv_norm, int Cout, int Cin, int taps, int ldk, int perm_split, int off_lo, int off_hi,
    float scale, int fmt, float x8_mul, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float row[];        // [Cin * taps]
  __shared__ float sh[17];
  weightnorm_fwd_h3_row(v, g, Wh, Wl, inv_norm, Cout, Cin, taps, ldk, perm_split, off_lo, off_hi, scale, fmt, x8_mul, vec_ok,
                        blockIdx.x, row, sh);
}

// Several weight tensors in ONE launch (round 4): a flow step prepares ten conv weights, most of them 4 - 9 MB -- at that
// size a launch is two dependent memory round trips long, not bandwidth-bound (8.5 us for 8 MB).  The items travel as a
// kernel argument (no device-side table to upload); workgroup b belongs to the item whose [start, start + Cout) holds b.
constexpr int WN_MULTI_MAX = 16;
struct WnMulti {
  radmmm_wn_item it[WN_MULTI_MAX];
  int start[WN_MULTI_MAX + 1];
  int vec_ok[WN_MULTI_MAX];
  int n, fmt;
  float scale, x8_mul;
};
__global__ __launch_bounds__(256) void weightnorm_fwd_h3_multi_kernel(const WnMulti m) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  __shared__ float sh[17];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.start[k + 1]) ++k;
  const radmmm_wn_item& t = m.it[k];
  weightnorm_fwd_h3_row(t.v, t.g, static_cast<_Float16*>(t.Wh), static_cast<_Float16*>(t.Wl), t.inv_norm, t.Cout, t.Cin, t.taps,
                        t.ldk, t.perm_split, t.off_lo, t.off_hi, m.scale, m.fmt, m.x8_mul, m.vec_ok[k], blockIdx.x - m.start[k], row, sh);
}

// hi/lo [rows][ldh] split of scale * x (zero padded to ldh), any split format
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, int ld, void* __restrict__ hi,
                                                        void* __restrict__ lo, int ldh, int rows, int cols, float scale,
                                                        int fmt, float x8_mul, int* __restrict__ sat_flag,
                                                        void* __restrict__ lo16) {
  const int c4n = ldh / 4;
  const long long total = (long long)rows * c4n;
  float sat = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4n), c = (int)(i - (long long)r * c4n) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (c + e < cols) ? x[(long long)r * ld + c + e] : 0.f;
    // (lo16: with an 8-bit format the fp16 lo part as well, radmmm_split_opts.lo16 -- round 6: one pass gives a conv's input
    //  both as the three-product GEMM's (hi, lo) pair and as the FP8-cross weight gradient's (hi, cross array) pair)
    sat = fmaxf(sat, radmmm::store_split4_fmt(hi, lo, (long long)r * ldh, c, fmt, x8_mul, scale, v[0], v[1], v[2], v[3], lo16));
  }
  radmmm::raise_sat_flag(sat_flag, sat, fmt ? x8_mul : 0.f);
}

// dst[b][c][r] = src[b][r][c] for both members of a split pair; 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_pair_kernel(const _Float16* __restrict__ sh_, const _Float16* __restrict__ sl_,
                                                             int ld_src, long long src_batch, _Float16* __restrict__ dh,
                                                             _Float16* __restrict__ dl, int ld_dst, long long dst_batch,
                                                             int rows, int cols) {
  __shared__ _Float16 th[32][34], tl[32][34];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const bool ok = r < rows && c < cols;
    th[i][tx] = ok ? sh_[b * src_batch + (long long)r * ld_src + c] : (_Float16)0.f;
    tl[i][tx] = ok ? sl_[b * src_batch + (long long)r * ld_src + c] : (_Float16)0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) {
      dh[b * dst_batch + (long long)c * ld_dst + r] = th[tx][i];
      dl[b * dst_batch + (long long)c * ld_dst + r] = tl[tx][i];
    }
  }
}

// the same for a B-role 8-bit pair (hi f16 + cross array [lo8 | hi8] per 32 columns): the 8-bit lo parts move with the
// transposition, the 8-bit hi parts are re-derived from the fp16 hi (what the producer did: e4m3(hi * 2^e) of the value
// already rounded to fp16 differs from e4m3(t * 2^e) only in double-rounding ties; both are valid 8-bit images of hi)
__device__ __forceinline__ void transpose_pair_x8_tile(const _Float16* __restrict__ sh_, const unsigned char* __restrict__ sx,
                                                       int ld_src, long long src_batch, _Float16* __restrict__ dh,
                                                       unsigned char* __restrict__ dx, int ld_dst, long long dst_batch,
                                                       int rows, int cols, int fmt, float x8_mul, int bx, int by, int b,
                                                       _Float16 (*th)[34], unsigned char (*tl)[36]) {
  const int r0 = by * 32, c0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const bool ok = r < rows && c < cols;
    th[i][tx] = ok ? sh_[b * src_batch + (long long)r * ld_src + c] : (_Float16)0.f;
    tl[i][tx] = ok ? sx[2 * (b * src_batch + (long long)r * ld_src) + radmmm::x8_lo_off(c, fmt)] : (unsigned char)0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) {
      const _Float16 h = th[tx][i];
      dh[b * dst_batch + (long long)c * ld_dst + r] = h;
      unsigned char* row = dx + 2 * (b * dst_batch + (long long)c * ld_dst);
      row[radmmm::x8_hi_off(r, fmt)] = (unsigned char)(radmmm::pack_e4m3x4((float)h * x8_mul, 0.f, 0.f, 0.f) & 0xffu);
      row[radmmm::x8_lo_off(r, fmt)] = tl[tx][i];
    }
  }
}

__global__ __launch_bounds__(256) void transpose_pair_x8_kernel(const _Float16* __restrict__ sh_, const unsigned char* __restrict__ sx,
                                                                int ld_src, long long src_batch, _Float16* __restrict__ dh,
                                                                unsigned char* __restrict__ dx, int ld_dst, long long dst_batch,
                                                                int rows, int cols, int fmt, float x8_mul) {
  __shared__ _Float16 th[32][34];
  __shared__ unsigned char tl[32][36];
  transpose_pair_x8_tile(sh_, sx, ld_src, src_batch, dh, dx, ld_dst, dst_batch, rows, cols, fmt, x8_mul, blockIdx.x, blockIdx.y,
                         blockIdx.z, th, tl);
}

// 64 x 64 tile of the same transposition with wide accesses (rows, cols multiples of 4; hi arrays 8-byte, cross arrays
// 4-byte aligned rows): a source row of the tile is 128 B of fp16 hi + 2 x 32 B of lo8, a destination row 128 B of hi and
// 128 B of cross array ([lo8 | hi8] of two 32-column groups are adjacent) -- whole cache lines both ways, where the 32 x 32
// tile above moves 64-byte rows of halves and single bytes (2.9 TB/s on a flow step's weights).  Same bytes out.
__device__ __forceinline__ void transpose_pair_x8_tile64(const _Float16* __restrict__ sh_, const unsigned char* __restrict__ sx,
                                                         int ld_src, long long src_batch, _Float16* __restrict__ dh,
                                                         unsigned char* __restrict__ dx, int ld_dst, long long dst_batch,
                                                         int rows, int cols, int fmt, float x8_mul, int bx, int by, int b,
                                                         _Float16 (*th)[66], unsigned char (*tl)[68]) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int r0 = by * 64, c0 = bx * 64;
  const int q = threadIdx.x & 15, p = threadIdx.x >> 4;       // 16 groups of 4 columns x 16 rows per pass
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int rl = pass * 16 + p, r = r0 + rl, c = c0 + 4 * q;
    h4 hv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    unsigned lv = 0;
    if (r < rows && c < cols) {
      const long long base = b * src_batch + (long long)r * ld_src;
      hv = *reinterpret_cast<const h4*>(sh_ + base + c);
      lv = *reinterpret_cast<const unsigned*>(sx + 2 * base + radmmm::x8_lo_off(c, fmt));
    }
    // LDS rows of 33 / 17 words: the transposed reads below (word stride 4 x 33 / 4 x 17 across the column groups) hit 16
    // different banks; a row start is only 4-byte aligned, so the four halves go in as two words
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<h2*>(&th[rl][4 * q]) = h2{hv[0], hv[1]};
    *reinterpret_cast<h2*>(&th[rl][4 * q + 2]) = h2{hv[2], hv[3]};
    *reinterpret_cast<unsigned*>(&tl[rl][4 * q]) = lv;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int cl = pass * 16 + p, c = c0 + cl, r = r0 + 4 * q;  // destination row c, destination columns r .. r + 3
    if (c < cols && r < rows) {
      h4 hv;
      unsigned lv = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hv[e] = th[4 * q + e][cl];
        lv |= (unsigned)tl[4 * q + e][cl] << (8 * e);
      }
      const long long base = b * dst_batch + (long long)c * ld_dst;
      *reinterpret_cast<h4*>(dh + base + r) = hv;
      unsigned char* row = dx + 2 * base;
      *reinterpret_cast<unsigned*>(row + radmmm::x8_hi_off(r, fmt)) =
          radmmm::pack_e4m3x4((float)hv[0] * x8_mul, (float)hv[1] * x8_mul, (float)hv[2] * x8_mul, (float)hv[3] * x8_mul);
      *reinterpret_cast<unsigned*>(row + radmmm::x8_lo_off(r, fmt)) = lv;
    }
  }
}

// several pairs in one launch (see weightnorm_fwd_h3_multi_kernel): workgroup b -> item, then (tile column, tile row, batch)
constexpr int TP_MULTI_MAX = 16;
struct TpMulti {
  radmmm_tp_item it[TP_MULTI_MAX];
  int start[TP_MULTI_MAX + 1];
  int nbx[TP_MULTI_MAX], nby[TP_MULTI_MAX], wide[TP_MULTI_MAX];
  int n, fmt;
  float x8_mul;
};
__global__ __launch_bounds__(256) void transpose_pair_x8_multi_kernel(const TpMulti m) {
  __shared__ __attribute__((aligned(16))) _Float16 th[64][66];
  __shared__ __attribute__((aligned(16))) unsigned char tl[64][68];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.start[k + 1]) ++k;
  const radmmm_tp_item& t = m.it[k];
  int r = blockIdx.x - m.start[k];
  const int bx = r % m.nbx[k];
  r /= m.nbx[k];
  const int by = r % m.nby[k], b = r / m.nby[k];
  if (m.wide[k])
    transpose_pair_x8_tile64(static_cast<const _Float16*>(t.src_h), static_cast<const unsigned char*>(t.src_l), t.ld_src,
                             (long long)t.src_batch, static_cast<_Float16*>(t.dst_h), static_cast<unsigned char*>(t.dst_l), t.ld_dst,
                             (long long)t.dst_batch, t.rows, t.cols, m.fmt, m.x8_mul, bx, by, b, th, tl);
  else
    transpose_pair_x8_tile(static_cast<const _Float16*>(t.src_h), static_cast<const unsigned char*>(t.src_l), t.ld_src,
                           (long long)t.src_batch, static_cast<_Float16*>(t.dst_h), static_cast<unsigned char*>(t.dst_l), t.ld_dst,
                           (long long)t.dst_batch, t.rows, t.cols, m.fmt, m.x8_mul, bx, by, b,
                           reinterpret_cast<_Float16(*)[34]>(&th[0][0]), reinterpret_cast<unsigned char(*)[36]>(&tl[0][0]));
}

}  // namespace

extern "C" int radmmm_weightnorm_fwd_h3_multi(const radmmm_wn_item* items, int n, float scale, const radmmm_split_opts* so,
                                              radmmm_stream_t stream) {
  RADMMM_REQUIRE(items && n >= 1 && n <= WN_MULTI_MAX, "weightnorm_fwd_h3_multi: 1 .. %d items", WN_MULTI_MAX);
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  WnMulti m;
  m.n = n;
  m.fmt = fmt;
  m.scale = scale;
  m.x8_mul = ldexpf(1.f, so ? so->x8_exp : 0);
  size_t smem = 0;
  int blocks = 0;
  for (int k = 0; k < n; ++k) {
    const radmmm_wn_item& t = items[k];
    RADMMM_REQUIRE(t.v && t.Wh && t.Wl && (t.inv_norm || !t.g), "weightnorm_fwd_h3_multi: null pointer in item %d", k);
    RADMMM_REQUIRE(t.Cout > 0 && t.Cin > 0 && t.taps > 0 && t.ldk >= t.Cin && t.ldk % 8 == 0, "weightnorm_fwd_h3_multi: bad dims in item %d", k);
    RADMMM_REQUIRE(fmt == RADMMM_SPLIT_F16 || (t.ldk % 32 == 0 && abs(so->x8_exp) <= 16),
                   "weightnorm_fwd_h3_multi: 8-bit format needs ldk %% 32 == 0");
    const size_t sm = (size_t)t.Cin * t.taps * sizeof(float);
    RADMMM_REQUIRE(sm <= 60 * 1024, "weightnorm_fwd_h3_multi: Cin*taps=%d too large for the row buffer", t.Cin * t.taps);
    smem = sm > smem ? sm : smem;
    m.it[k] = t;
    m.start[k] = blocks;
    blocks += t.Cout;
    m.vec_ok[k] = (t.Cin % 4 == 0 && t.perm_split % 4 == 0 && t.off_lo % 4 == 0 && t.off_hi % 4 == 0 && t.ldk % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(t.Wh) & 7) == 0 && (reinterpret_cast<uintptr_t>(t.Wl) & 7) == 0) ? 1 : 0;
  }
  m.start[n] = blocks;
  hipLaunchKernelGGL(weightnorm_fwd_h3_multi_kernel, dim3(blocks), dim3(256), smem, static_cast<hipStream_t>(stream), m);
  return radmmm::check_launch("weightnorm_fwd_h3_multi");
}

extern "C" int radmmm_transpose_f16_pair_multi(const radmmm_tp_item* items, int n, int fmt, int x8_exp, radmmm_stream_t stream) {
  RADMMM_REQUIRE(items && n >= 1 && n <= TP_MULTI_MAX, "transpose_f16_pair_multi: 1 .. %d items", TP_MULTI_MAX);
  RADMMM_REQUIRE(fmt != RADMMM_SPLIT_F16 && abs(x8_exp) <= 16, "transpose_f16_pair_multi: 8-bit B-role pairs only (|x8_exp| <= 16)");
  TpMulti m;
  m.n = n;
  m.fmt = fmt;
  m.x8_mul = ldexpf(1.f, x8_exp);
  long long blocks = 0;
  for (int k = 0; k < n; ++k) {
    const radmmm_tp_item& t = items[k];
    RADMMM_REQUIRE(t.src_h && t.src_l && t.dst_h && t.dst_l, "transpose_f16_pair_multi: null pointer in item %d", k);
    RADMMM_REQUIRE(t.batches > 0 && t.rows > 0 && t.cols > 0 && t.ld_src >= t.cols && t.ld_dst >= t.rows && t.ld_src % 32 == 0 &&
                       t.ld_dst % 32 == 0, "transpose_f16_pair_multi: bad dims in item %d (ld %% 32 == 0)", k);
    m.it[k] = t;
    // the 64 x 64 tile's 4-element accesses: whole groups inside the tensor, aligned rows and batches
    const bool wide = t.rows % 4 == 0 && t.cols % 4 == 0 && t.src_batch % 4 == 0 && t.dst_batch % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(t.src_h) & 7) == 0 && (reinterpret_cast<uintptr_t>(t.dst_h) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(t.src_l) & 3) == 0 && (reinterpret_cast<uintptr_t>(t.dst_l) & 3) == 0 &&
                      !radmmm::debug_env("RADMMM_TP_NARROW");
    const int tile = wide ? 64 : 32;
    m.wide[k] = wide ? 1 : 0;
    m.nbx[k] = (t.cols + tile - 1) / tile;
    m.nby[k] = (t.rows + tile - 1) / tile;
    m.start[k] = (int)blocks;
    blocks += (long long)m.nbx[k] * m.nby[k] * t.batches;
    RADMMM_REQUIRE(blocks < 0x7fffffffLL, "transpose_f16_pair_multi: too many tiles");
  }
  m.start[n] = (int)blocks;
  hipLaunchKernelGGL(transpose_pair_x8_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), m);
  return radmmm::check_launch("transpose_f16_pair_multi");
}

extern "C" int radmmm_weightnorm_fwd_h3(const float* v, const float* g, void* Wh, void* Wl, float* inv_norm, int Cout,
                                        int Cin, int taps, int ldk, int perm_split, int off_lo, int off_hi, float scale,
                                        const radmmm_split_opts* so, radmmm_stream_t stream) {
  RADMMM_REQUIRE(v && Wh && Wl && (inv_norm || !g), "weightnorm_fwd_h3: null pointer");
  RADMMM_REQUIRE(Cout > 0 && Cin > 0 && taps > 0 && ldk >= Cin && ldk % 8 == 0, "weightnorm_fwd_h3: bad dims");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(fmt == RADMMM_SPLIT_F16 || (ldk % 32 == 0 && abs(so->x8_exp) <= 16), "weightnorm_fwd_h3: 8-bit format needs ldk %% 32 == 0");
  const size_t smem = (size_t)Cin * taps * sizeof(float);
  RADMMM_REQUIRE(smem <= 60 * 1024, "weightnorm_fwd_h3: Cin*taps=%d too large for the row buffer", Cin * taps);
  // 4-column groups need every piece of the column permutation and the row pitch to keep 4-element alignment
  const int vec_ok = (Cin % 4 == 0 && perm_split % 4 == 0 && off_lo % 4 == 0 && off_hi % 4 == 0 && ldk % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(Wh) & 7) == 0 && (reinterpret_cast<uintptr_t>(Wl) & 7) == 0) ? 1 : 0;
  hipLaunchKernelGGL(weightnorm_fwd_h3_kernel, dim3(Cout), dim3(256), smem, static_cast<hipStream_t>(stream), v, g,
                     static_cast<_Float16*>(Wh), static_cast<_Float16*>(Wl), inv_norm, Cout, Cin, taps, ldk, perm_split,
                     off_lo, off_hi, scale, fmt, ldexpf(1.f, so ? so->x8_exp : 0), vec_ok);
  return radmmm::check_launch("weightnorm_fwd_h3");
}

extern "C" int radmmm_split_f16(const float* x, int ld, void* hi, void* lo, int ldh, int rows, int cols, float scale,
                                const radmmm_split_opts* so, radmmm_stream_t stream) {
  RADMMM_REQUIRE(x && hi && lo, "split_f16: null pointer");
  RADMMM_REQUIRE(rows > 0 && cols > 0 && ld >= cols && ldh >= cols && ldh % 8 == 0, "split_f16: bad dims (ldh %% 8 == 0)");
  const int fmt = so ? so->fmt : RADMMM_SPLIT_F16;
  RADMMM_REQUIRE(fmt == RADMMM_SPLIT_F16 || (ldh % 32 == 0 && abs(so->x8_exp) <= 16), "split_f16: 8-bit format needs ldh %% 32 == 0");
  const long long total = (long long)rows * (ldh / 4);
  const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(split_f16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, hi, lo, ldh, rows,
                     cols, scale, fmt, ldexpf(1.f, so ? so->x8_exp : 0), so ? so->sat_flag : nullptr,
                     (so && fmt != RADMMM_SPLIT_F16) ? so->lo16 : nullptr);
  return radmmm::check_launch("split_f16");
}

extern "C" int radmmm_transpose_f16_pair(const void* src_h, const void* src_l, int ld_src, int64_t src_batch, void* dst_h,
                                         void* dst_l, int ld_dst, int64_t dst_batch, int batches, int rows, int cols,
                                         int fmt, int x8_exp, radmmm_stream_t stream) {
  RADMMM_REQUIRE(src_h && src_l && dst_h && dst_l, "transpose_f16_pair: null pointer");
  RADMMM_REQUIRE(batches > 0 && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "transpose_f16_pair: bad dims");
  if (fmt != RADMMM_SPLIT_F16) {
    RADMMM_REQUIRE(ld_src % 32 == 0 && ld_dst % 32 == 0 && abs(x8_exp) <= 16, "transpose_f16_pair: 8-bit format needs ld %% 32 == 0");
    hipLaunchKernelGGL(transpose_pair_x8_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batches), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const _Float16*>(src_h),
                       static_cast<const unsigned char*>(src_l), ld_src, (long long)src_batch, static_cast<_Float16*>(dst_h),
                       static_cast<unsigned char*>(dst_l), ld_dst, (long long)dst_batch, rows, cols, fmt, ldexpf(1.f, x8_exp));
    return radmmm::check_launch("transpose_f16_pair(x8)");
  }
  hipLaunchKernelGGL(transpose_pair_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batches), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(src_h),
                     static_cast<const _Float16*>(src_l), ld_src, (long long)src_batch, static_cast<_Float16*>(dst_h),
                     static_cast<_Float16*>(dst_l), ld_dst, (long long)dst_batch, rows, cols);
  return radmmm::check_launch("transpose_f16_pair");
}
