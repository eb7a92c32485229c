// STFT magnitude -> mel filterbank -> log-clamp (audio_processing.py:137-154, 227-255) on
// gfx950.  The framed DFT is a GEMM whose A operand is the reflect-padded signal read with a row
// stride of `hop` samples (overlapping rows, a_item_stride form of radmmm_rowgemm_f32), so no
// im2col copy of the frames is ever made; the mel projection is a second row GEMM.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ audio,
                                                          float* __restrict__ xpad, int B, int S,
                                                          int pad, int pitch) {
  const long long total = (long long)B * pitch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / pitch), p = (int)(i - (long long)b * pitch);
    float v = 0.f;
    if (p < S + 2 * pad) {
      int j = p - pad;
      if (j < 0) j = -j;                    // reflect (no edge repeat)
      if (j >= S) j = 2 * (S - 1) - j;
      v = audio[(long long)b * S + j];
    }
    xpad[i] = v;
  }
}

__global__ __launch_bounds__(256) void pad_cols_kernel(const float* __restrict__ src, int ld_src,
                                                       float* __restrict__ dst, int ld_dst, int rows,
                                                       int cols) {
  const long long total = (long long)rows * ld_dst;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i - (long long)r * ld_dst);
    dst[i] = c < cols ? src[(long long)r * ld_src + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void magnitude_kernel(const float* __restrict__ spec, int lds,
                                                        float* __restrict__ mag, int ldm,
                                                        long long rows, int cutoff) {
  const long long total = rows * ldm;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / ldm;
    const int c = (int)(i - r * ldm);
    float v = 0.f;
    if (c < cutoff) {
      const float re = spec[r * lds + c], im = spec[r * lds + cutoff + c];
      v = sqrtf(re * re + im * im);
    }
    mag[i] = v;
  }
}

// mel [B, n_mel, F] = log(clamp(melT[b*F + f, m], clip))
__global__ __launch_bounds__(256) void logclamp_transpose_kernel(const float* __restrict__ melT,
                                                                 int ldt, float* __restrict__ mel,
                                                                 int B, int F, int n_mel, float clip) {
  const long long total = (long long)B * n_mel * F;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const long long j = i / F;
    const int m = (int)(j % n_mel), b = (int)(j / n_mel);
    mel[i] = logf(fmaxf(melT[((long long)b * F + f) * ldt + m], clip));
  }
}

inline int grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
inline long long r4(long long x) { return (x + 3) & ~3LL; }

struct StftLayout {
  int F, cutoff, pitch, lds, ldm, ldt;
  long long off_xpad, off_spec, off_mag, off_melb, off_melT, total;
};
inline StftLayout stft_layout(int B, int S, int n_fft, int hop, int n_mel) {
  StftLayout L;
  L.F = 1 + S / hop;
  L.cutoff = n_fft / 2 + 1;
  L.pitch = (int)r4(S + n_fft);
  L.lds = (int)r4(2 * L.cutoff);
  L.ldm = (int)r4(L.cutoff);
  L.ldt = (int)r4(n_mel);
  long long o = 0;
  L.off_xpad = o; o += r4((long long)B * L.pitch);
  L.off_spec = o; o += r4((long long)B * L.F * L.lds);
  L.off_mag = o;  o += r4((long long)B * L.F * L.ldm);
  L.off_melb = o; o += r4((long long)n_mel * L.ldm);
  L.off_melT = o; o += r4((long long)B * L.F * L.ldt);
  L.total = o;
  return L;
}

}  // namespace

extern "C" int64_t radmmm_stft_mel_scratch_floats(int B, int S, int n_fft, int hop, int n_mel) {
  return stft_layout(B, S, n_fft, hop, n_mel).total;
}

extern "C" int radmmm_stft_mel(const float* audio, const float* basis, const float* mel_basis, float* mel,
                               float* scratch, int B, int S, int n_fft, int hop, int n_mel, float clip,
                               radmmm_stream_t stream) {
  RADMMM_REQUIRE(audio && basis && mel_basis && mel && scratch, "stft_mel: null pointer");
  RADMMM_REQUIRE(B > 0 && S > n_fft / 2 && n_fft > 0 && n_fft % 4 == 0 && hop > 0 && hop % 4 == 0 && n_mel > 0,
                 "stft_mel: bad dims (need n_fft %% 4 == 0, hop %% 4 == 0, S > n_fft/2)");
  RADMMM_REQUIRE(radmmm::aligned16(basis) && radmmm::aligned16(scratch), "stft_mel: basis/scratch must be 16B aligned");
  const StftLayout L = stft_layout(B, S, n_fft, hop, n_mel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* xpad = scratch + L.off_xpad;
  float* spec = scratch + L.off_spec;
  float* mag = scratch + L.off_mag;
  float* melb = scratch + L.off_melb;
  float* melT = scratch + L.off_melT;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((long long)B * L.pitch)), dim3(256), 0, s, audio, xpad,
                     B, S, n_fft / 2, L.pitch);
  hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for((long long)n_mel * L.ldm)), dim3(256), 0, s, mel_basis,
                     L.cutoff, melb, L.ldm, n_mel, L.cutoff);
  int rc = radmmm::check_launch("stft_mel: pad");
  if (rc) return rc;
  radmmm_rowgemm_desc d = {};
  d.A = xpad; d.lda = hop; d.a_item_stride = L.pitch;
  d.B = basis; d.ldb = n_fft; d.b_tap_stride = 0; d.b_layout = 0;
  d.C = spec; d.ldc = L.lds;
  d.M = B * L.F; d.N = 2 * L.cutoff; d.K = n_fft;
  d.taps = 1; d.dil = 1; d.sign = 1; d.T = L.F; d.ratio_taps = 1; d.ratio_dil = 1;
  rc = radmmm_rowgemm_f32(&d, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for((long long)B * L.F * L.ldm)), dim3(256), 0, s, spec,
                     L.lds, mag, L.ldm, (long long)B * L.F, L.cutoff);
  radmmm_rowgemm_desc m = {};
  m.A = mag; m.lda = L.ldm;
  m.B = melb; m.ldb = L.ldm; m.b_layout = 0;
  m.C = melT; m.ldc = L.ldt;
  m.M = B * L.F; m.N = n_mel; m.K = L.cutoff;
  m.taps = 1; m.dil = 1; m.sign = 1; m.T = L.F; m.ratio_taps = 1; m.ratio_dil = 1;
  rc = radmmm_rowgemm_f32(&m, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(logclamp_transpose_kernel, dim3(grid_for((long long)B * n_mel * L.F)), dim3(256), 0, s,
                     melT, L.ldt, mel, B, L.F, n_mel, clip);
  return radmmm::check_launch("stft_mel");
}
