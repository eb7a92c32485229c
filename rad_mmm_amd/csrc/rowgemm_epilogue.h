// Fused epilogue of radmmm_rowgemm_f32 (see include/radmmm_hip.h for the order of operations),
// applied to 4 consecutive output columns of one row.  Shared by both tilings of the kernel.
#pragma once
#include "common.h"
#include "split_pack.h"

namespace radmmm {

struct EpilogueCtx {
  bool need_row, vec_ok;
  __device__ __forceinline__ explicit EpilogueCtx(const radmmm_rowgemm_desc& p) {
    need_row = p.pconv || p.premask || p.postmask || p.rowscale;
    vec_ok = (p.ldc % 4 == 0) && aligned16(p.C) && (!p.add || (p.ldadd % 4 == 0 && aligned16(p.add))) &&
             (!p.dact || p.dact_h || (p.lddact % 4 == 0 && aligned16(p.dact_src))) &&
             (!p.C2 || (p.ldc2 % 4 == 0 && aligned16(p.C2))) &&
             (p.n_c2_src <= 0 || (p.ldc2 % 4 == 0 && aligned16(p.c2_src[0]) && aligned16(p.c2_src[p.n_c2_src > 1 ? 1 : 0]) &&
                                  aligned16(p.c2_src[p.n_c2_src > 2 ? 2 : 0])));
  }
};

// split copy of 4 output columns (ld % 4 == 0, col % 4 == 0) in the descriptor's split format; columns >= N are
// written as zeros in the f16 format and skipped in the 8-bit formats.  Returns max |scale * x| (saturation tracking).
__device__ __forceinline__ float store_split4(void* hi_, void* lo_, int ld, float s, int fmt, int x8_exp, int row, int col,
                                              int N, const float (&x)[4], void* lo16_ = nullptr) {
  const float mul = __builtin_ldexpf(1.f, x8_exp);
  if (col + 3 < N || fmt == 0) {
    return store_split4_fmt(hi_, lo_, (long long)row * ld, col, fmt, mul, s, x[0], col + 1 < N ? x[1] : 0.f,
                            col + 2 < N ? x[2] : 0.f, col + 3 < N ? x[3] : 0.f, lo16_);
  }
  float amax = 0.f;
  for (int e = 0; e < 4 && col + e < N; ++e)
    amax = fmaxf(amax, store_split1_fmt(hi_, lo_, (long long)row * ld, col + e, fmt, mul, s, x[e], lo16_));
  return amax;
}

// saved output y of the dact step from its 8-bit split copy (include/radmmm_hip.h: dact_h / dact_x):
// y = hi + e4m3_lo * 2^-(11 + e)
__device__ __forceinline__ float dact_src_from_pair(const radmmm_rowgemm_desc& p, int row, int col) {
  const float hi = (float)static_cast<const _Float16*>(p.dact_h)[(long long)row * p.lddact_h + col];
  const unsigned char b = static_cast<const unsigned char*>(p.dact_x)[(long long)row * p.lddact_h * 2 + x8_lo_off(col, RADMMM_SPLIT_X8A)];
  const float lo = __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
  return hi + lo * __builtin_ldexpf(1.f, -(11 + p.dact_x8_exp));
}

// per-row factors of the epilogue: length mask and partial-conv renormalisation ratio
__device__ __forceinline__ void epilogue_row_factors(const radmmm_rowgemm_desc& p, const EpilogueCtx& ec, int row,
                                                     float& maskv, float& ratio) {
  maskv = 1.f;
  ratio = 1.f;
  if (ec.need_row && row < p.M) {
    const int b = row / p.T;
    const int t = row - b * p.T;
    const int len = p.lens ? p.lens[b] : p.T;
    maskv = t < len ? 1.f : 0.f;
    if (p.pconv || p.rowscale == 2) ratio = pconv_ratio(t, len, p.ratio_taps, p.ratio_dil);
  }
}

__device__ __forceinline__ float epilogue_store4_pre(const radmmm_rowgemm_desc& p, const EpilogueCtx& ec, int row, int col,
                                                     float4 a4, float maskv, float ratio, const float (&biasv)[4]);

__device__ __forceinline__ float epilogue_store4(const radmmm_rowgemm_desc& p, const EpilogueCtx& ec, int row,
                                                 int col, float4 a4) {
  if (row >= p.M || col >= p.N) return 0.f;
  float maskv, ratio;
  epilogue_row_factors(p, ec, row, maskv, ratio);
  float biasv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) biasv[e] = (col + e < p.N) ? p.bias[col + e] : 0.f;
  }
  return epilogue_store4_pre(p, ec, row, col, a4, maskv, ratio, biasv);
}

// same with the row factors and the bias of the 4 columns supplied by the caller (kernels that hoist
// them out of the per-row loop: no dependent global loads remain in the store path).  Returns max |scale * x| over the
// split outputs written (0 without split outputs).
__device__ __forceinline__ float epilogue_store4_pre(const radmmm_rowgemm_desc& p, const EpilogueCtx& ec, int row, int col,
                                                     float4 a4, float maskv, float ratio, const float (&biasv)[4]) {
  if (row >= p.M || col >= p.N) return 0.f;
  float v[4] = {a4.x, a4.y, a4.z, a4.w};
  const bool full = ec.vec_ok && col + 3 < p.N;
  float addv[4] = {0.f, 0.f, 0.f, 0.f}, dsv[4] = {0.f, 0.f, 0.f, 0.f}, c2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (full) {
    if (p.add) {
      const float4 t4 = *reinterpret_cast<const float4*>(p.add + (long long)row * p.ldadd + col);
      addv[0] = t4.x; addv[1] = t4.y; addv[2] = t4.z; addv[3] = t4.w;
    }
    if (p.dact && p.dact_h) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dsv[e] = dact_src_from_pair(p, row, col + e);
    } else if (p.dact) {
      const float4 t4 = *reinterpret_cast<const float4*>(p.dact_src + (long long)row * p.lddact + col);
      dsv[0] = t4.x; dsv[1] = t4.y; dsv[2] = t4.z; dsv[3] = t4.w;
    }
    if (p.n_c2_src > 0) {
      for (int k = 0; k < p.n_c2_src; ++k) {
        const float4 t4 = *reinterpret_cast<const float4*>(p.c2_src[k] + (long long)row * p.ldc2 + col);
        c2v[0] += t4.x; c2v[1] += t4.y; c2v[2] += t4.z; c2v[3] += t4.w;
      }
    } else if (p.C2 && p.c2_accum) {
      const float4 t4 = *reinterpret_cast<const float4*>(p.C2 + (long long)row * p.ldc2 + col);
      c2v[0] = t4.x; c2v[1] = t4.y; c2v[2] = t4.z; c2v[3] = t4.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e < p.N) {
        if (p.add) addv[e] = p.add[(long long)row * p.ldadd + col + e];
        if (p.dact) dsv[e] = p.dact_h ? dact_src_from_pair(p, row, col + e) : p.dact_src[(long long)row * p.lddact + col + e];
        if (p.n_c2_src > 0) {
          for (int k = 0; k < p.n_c2_src; ++k) c2v[e] += p.c2_src[k][(long long)row * p.ldc2 + col + e];
        } else if (p.C2 && p.c2_accum) {
          c2v[e] = p.C2[(long long)row * p.ldc2 + col + e];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e];
    if (p.pconv) x *= ratio;
    if (p.premask) x *= maskv;
    x += biasv[e];
    x += addv[e];
    if (p.postmask) x *= maskv;
    if (p.dact) x *= dact_from_out(dsv[e], p.dact);
    if (p.rowscale == 1) x *= maskv;
    if (p.rowscale == 2) x *= maskv * ratio;
    x = act_apply(x, p.act);
    v[e] = x;
    c2v[e] += x;
  }
  // optional split-fp16 copies (hi/lo of scale*x) feeding the next split-f16 GEMM
  float amax = 0.f;
  if (p.Ch) amax = store_split4(p.Ch, p.Cl, p.ldch, p.ch_scale, p.split_fmt, p.ch_x8_exp, row, col, p.N, v, p.Clo);
  if (p.C2h) amax = fmaxf(amax, store_split4(p.C2h, p.C2l, p.ldc2h, p.c2h_scale, p.split_fmt, p.c2h_x8_exp, row, col, p.N, c2v));
  if (full) {
    if (p.C) *reinterpret_cast<float4*>(p.C + (long long)row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    if (p.C2)
      *reinterpret_cast<float4*>(p.C2 + (long long)row * p.ldc2 + col) = make_float4(c2v[0], c2v[1], c2v[2], c2v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e < p.N) {
        if (p.C) p.C[(long long)row * p.ldc + col + e] = v[e];
        if (p.C2) p.C2[(long long)row * p.ldc2 + col + e] = c2v[e];
      }
    }
  }
  return amax;
}

}  // namespace radmmm
