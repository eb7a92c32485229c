// Writers of the split operand copies consumed by radmmm_rowgemm_h3 (see include/radmmm_hip.h, "split formats").
//
//   fmt 0 (f16 pair)  hi = fp16(t), lo = fp16(t - hi) in two [rows][ld] half arrays, t = scale * x clamped to +-60000
//   fmt 1 (x8, A role) hi as above; the second array is the 8-bit CROSS array with the same row pitch (2 * ld bytes):
//                      per 32 columns 64 bytes = [ e4m3(t * 2^e) x 32 | e4m3((t - hi) * 2^(11 + e)) x 32 ]
//   fmt 2 (x8, B role) the two 32-byte halves swapped: [ lo8 x 32 | hi8 x 32 ]
// In the GEMM one block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 multiplies A's [hi8 | lo8] with B's [lo8 | hi8]: both
// cross terms Ah.Bl + Al.Bh of the split product in one instruction at twice the f16 rate.
#pragma once
#include "common.h"

namespace radmmm {

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float clamp_f16(float t) { return __builtin_amdgcn_fmed3f(t, -60000.f, 60000.f); }
__device__ __forceinline__ float clamp_e4m3(float t) { return __builtin_amdgcn_fmed3f(t, -448.f, 448.f); }

// four e4m3 bytes (round to nearest even; the hardware conversion does not saturate -> clamp first)
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), w, true);
  return (unsigned)w;
}

// byte offset of column `col`'s hi8 / lo8 inside a cross-array row
__device__ __forceinline__ unsigned x8_hi_off(int col, int fmt) { return ((unsigned)col >> 5) * 64u + ((unsigned)col & 31u) + (fmt == 2 ? 32u : 0u); }
__device__ __forceinline__ unsigned x8_lo_off(int col, int fmt) { return ((unsigned)col >> 5) * 64u + ((unsigned)col & 31u) + (fmt == 2 ? 0u : 32u); }

// 4 consecutive columns (col % 4 == 0) of row `row`: x0..x3 are the UNSCALED values.  Returns max |scale * x| (for the
// saturation flag: > 60000 means the fp16 clamp changed a value).  ld in halves, ld % 4 == 0.
// lo16_ (optional): with an 8-bit format, store the fp16 lo part there as well (pitch of the hi array)
template <typename OffT>
__device__ __forceinline__ float store_split4_fmt(void* hi_, void* lo_, OffT elem_off_row, int col, int fmt, float x8_mul,
                                                  float s, float x0, float x1, float x2, float x3, void* lo16_ = nullptr) {
  const float u0 = x0 * s, u1 = x1 * s, u2 = x2 * s, u3 = x3 * s;
  const float amax = fmaxf(fmaxf(fabsf(u0), fabsf(u1)), fmaxf(fabsf(u2), fabsf(u3)));
  const float t0 = clamp_f16(u0), t1 = clamp_f16(u1), t2 = clamp_f16(u2), t3 = clamp_f16(u3);
  f16x4_t hi;
  hi[0] = (_Float16)t0; hi[1] = (_Float16)t1; hi[2] = (_Float16)t2; hi[3] = (_Float16)t3;
  const float r0 = t0 - (float)hi[0], r1 = t1 - (float)hi[1], r2 = t2 - (float)hi[2], r3 = t3 - (float)hi[3];
  *reinterpret_cast<f16x4_t*>(static_cast<_Float16*>(hi_) + elem_off_row + col) = hi;
  if (fmt == 0) {
    f16x4_t lo;
    lo[0] = (_Float16)r0; lo[1] = (_Float16)r1; lo[2] = (_Float16)r2; lo[3] = (_Float16)r3;
    *reinterpret_cast<f16x4_t*>(static_cast<_Float16*>(lo_) + elem_off_row + col) = lo;
  } else {
    unsigned char* row = static_cast<unsigned char*>(lo_) + 2 * elem_off_row;
    const float lm = x8_mul * 2048.f;
    *reinterpret_cast<unsigned*>(row + x8_hi_off(col, fmt)) = pack_e4m3x4(t0 * x8_mul, t1 * x8_mul, t2 * x8_mul, t3 * x8_mul);
    *reinterpret_cast<unsigned*>(row + x8_lo_off(col, fmt)) = pack_e4m3x4(r0 * lm, r1 * lm, r2 * lm, r3 * lm);
    if (lo16_) {
      f16x4_t lo;
      lo[0] = (_Float16)r0; lo[1] = (_Float16)r1; lo[2] = (_Float16)r2; lo[3] = (_Float16)r3;
      *reinterpret_cast<f16x4_t*>(static_cast<_Float16*>(lo16_) + elem_off_row + col) = lo;
    }
  }
  return amax;
}

// one element (generic / tail paths)
__device__ __forceinline__ float store_split1_fmt(void* hi_, void* lo_, long long elem_off_row, int col, int fmt, float x8_mul,
                                                  float s, float x, void* lo16_ = nullptr) {
  const float u = x * s;
  const float t = clamp_f16(u);
  const _Float16 h = (_Float16)t;
  const float r = t - (float)h;
  static_cast<_Float16*>(hi_)[elem_off_row + col] = h;
  if (fmt == 0) {
    static_cast<_Float16*>(lo_)[elem_off_row + col] = (_Float16)r;
  } else {
    unsigned char* row = static_cast<unsigned char*>(lo_) + 2 * elem_off_row;
    row[x8_hi_off(col, fmt)] = (unsigned char)(pack_e4m3x4(t * x8_mul, 0.f, 0.f, 0.f) & 0xffu);
    row[x8_lo_off(col, fmt)] = (unsigned char)(pack_e4m3x4(r * x8_mul * 2048.f, 0.f, 0.f, 0.f) & 0xffu);
    if (lo16_) static_cast<_Float16*>(lo16_)[elem_off_row + col] = (_Float16)r;
  }
  return fabsf(u);
}

// Raise the saturation flag word from this lane's max |scale * x|: bit 0 when the fp16 clamp changed a value
// (> 60000); bit 1 when the tensor was written in an 8-bit cross format (x8_mul = 2^e > 0) and an element's 8-bit parts
// left e4m3's range (|scale * x| * 2^e > 448: both its hi8 and, bounded by the same product, its lo8 part clamp) -- that
// element then carries no cross-term correction, i.e. single-fp16-product accuracy -- and, one-hot, bit 1 + L for
// L = ceil(log2(|scale * x| * 2^e / 448)) in 1 .. 6: by how many powers of two the exponent was too large (the host lowers
// the tensor class's exponent by the highest L it sees, rad_mmm_amd/ops.py GradScale).
__device__ __forceinline__ void raise_sat_flag(int* flag, float amax, float x8_mul = 0.f) {
  if (!flag) return;
  // one atomic per wave at most, and none when the bits are already set (when a tensor saturates, most lanes see it: a
  // per-lane atomicOr on one address serialises -- measured 13 -> 66 us for a 50 MB split pass)
  const float r8 = x8_mul > 0.f ? amax * x8_mul * (1.f / 448.f) : 0.f;
  const bool b0 = amax > 60000.f, b1 = r8 > 1.f;
  const unsigned long long m0 = __ballot(b0), m1 = __ballot(b1);
  if (!(m0 | m1)) return;
  int bits = (m0 ? 1 : 0) | (m1 ? 2 : 0);
  if (m1) {
    int lvl = 1;                                                   // highest level among the wave's lanes
#pragma unroll
    for (int l = 1; l < 6; ++l)
      if (__ballot(r8 > (float)(1 << l))) lvl = l + 1;
    bits |= 2 << lvl;
  }
  if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)(m0 | m1)) - 1 && (__atomic_load_n(flag, __ATOMIC_RELAXED) & bits) != bits)
    atomicOr(flag, bits);
}

}  // namespace radmmm
