// Probe of the "FP8 cross terms" scheme: one f16 32x32x16 MFMA pair for Ah.Bh plus ONE block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64 carries both cross terms: A = [Ah8 | Al8], B = [Bl8 | Bh8], the lo
// blocks scaled by 2^-11 through the instruction's E8M0 block scales.
//   (1) semantics: which lane holds which k block, which lane's scale byte applies to which block -- checked against a
//       host reference built from decoded e4m3 bytes;
//   (2) rate: MFMA-only loops per 32-deep k step and output tile pair: 6 x f16 (shipped) vs 2 x f16 + 1 x scaled f8
//       (per 32x32 tile: 3 vs 1 + 0.5 instructions of equal 8-pass length; the f8 one is 16 passes).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f8x_probe.hip -o gpurun_out/mfma_f8x_probe && gpurun_out/mfma_f8x_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// OCP e4m3 (fn): 1-4-3, bias 7, no inf, 0x7f/0xff = NaN
static float e4m3_to_float(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -v : v;
}

// ---- (1) semantics: D[32][32] = sum over 64 k of A[i][k] * B[j][k] * 2^(sa[i][k/32]-127) * 2^(sb[j][k/32]-127)
__global__ void sem_kernel(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb,
                           float* D, int cvt_test, const float* fsrc, unsigned char* fdst) {
  const int lane = threadIdx.x;
  const int i = lane & 31, h = lane >> 5;
  i32x8 a, b;
  const int* pa = reinterpret_cast<const int*>(A + (i * 64 + h * 32));
  const int* pb = reinterpret_cast<const int*>(B + (i * 64 + h * 32));
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = pa[q]; b[q] = pb[q]; }
  const int scale_a = sa[i * 2 + h], scale_b = sb[i * 2 + h];      // byte 0 of the scale VGPR (opsel 0)
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0 /*A fp8 e4m3*/, 0 /*B fp8 e4m3*/, 0, scale_a, 0, scale_b);
  // C layout of 32x32 MFMA: lane l holds column (l & 31), rows 8*(e>>2) + 4*(l>>5) + (e&3)
#pragma unroll
  for (int e = 0; e < 16; ++e) D[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + i] = acc[e];
  if (cvt_test) {
    // fp32 -> e4m3 conversion of the hardware (round to nearest even, saturating?)
    const float x0 = fsrc[2 * lane], x1 = fsrc[2 * lane + 1];
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
    fdst[2 * lane] = pk & 0xff;
    fdst[2 * lane + 1] = (pk >> 8) & 0xff;
  }
}

// ---- (2) rate
template <int MODE>
__global__ __launch_bounds__(256, 1) void rate_kernel(const int* __restrict__ src, float* out, int iters) {
  constexpr int MB = 7;
  f32x16 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // per 32-deep k step: A fragments per row block: 2 x f16x8 (k blocks 0/1) hi, 2 x f16x8 lo or one 8-dword f8 fragment
  f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
  i32x8 a8[2], b8[2];
  const int t = threadIdx.x;
  auto ld = [&](int k) { return src[(t + 256 * k + blockIdx.x * 64) & 32767]; };
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      int w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ld(x * 8 + y * 4 + q);
      __builtin_memcpy(&ah[x][y], w, 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ld(16 + x * 8 + y * 4 + q);
      __builtin_memcpy(&al[x][y], w, 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ld(32 + x * 8 + y * 4 + q);
      __builtin_memcpy(&bh[x][y], w, 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = ld(48 + x * 8 + y * 4 + q);
      __builtin_memcpy(&bl[x][y], w, 16);
    }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int q = 0; q < 8; ++q) { a8[x][q] = ld(64 + x * 8 + q) & 0x3f3f3f3f; b8[x][q] = ld(80 + x * 8 + q) & 0x3f3f3f3f; }
  const int sc_lo = (t & 32) ? 127 - 11 : 127, sc_hi = (t & 32) ? 127 : 127 - 11;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int x = i & 1;
      if constexpr (MODE == 0) {            // shipped: 6 f16 MFMAs per k block pair... per 16-deep block: 3 per tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[x][kb], bh[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[x][kb], bh[1][kb], acc[i][1], 0, 0, 0);
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bl[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bl[1][kb], acc[i][1], 0, 0, 0);
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[1][kb], acc[i][1], 0, 0, 0);
        }
      } else if constexpr (MODE == 1) {     // FP8 cross terms: 2 f16 + 1 scaled f8 per tile and 32-deep step
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[1][kb], acc[i][1], 0, 0, 0);
        }
        acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[x], b8[0], acc[i][0], 0, 0, 0, sc_lo, 0, sc_hi);
        acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[x], b8[1], acc[i][1], 0, 0, 0, sc_lo, 0, sc_hi);
      } else if constexpr (MODE == 3) {     // the same with MXFP6 (e2m3) operands: format codes 2 / 2, 24 of the 32 bytes used
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[1][kb], acc[i][1], 0, 0, 0);
        }
        acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[x], b8[0], acc[i][0], 2, 2, 0, sc_lo, 0, sc_hi);
        acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[x], b8[1], acc[i][1], 2, 2, 0, sc_lo, 0, sc_hi);
      } else {                               // single product (throughput mode)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[0][kb], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[x][kb], bh[1][kb], acc[i][1], 0, 0, 0);
        }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void rate(const int* src, const char* name, const char* data) {
  float* d; hipMalloc(&d, 4);
  const int iters = 3000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // fp32-equivalent flops: 7 row blocks x 2 col blocks x 32x32 x 32 deep x 2, 4 waves, 256 workgroups
  const double eq = 256.0 * 4 * iters * 7 * 2 * 2.0 * 32 * 32 * 32;
  printf("%-34s data=%-8s %.3f ms  fp32-equivalent %.1f TFLOP/s\n", name, data, ms, eq / ms / 1e9);
  hipFree(d);
}

int main() {
  // (1)
  std::vector<unsigned char> A(32 * 64), B(32 * 64), sa(64), sb(64);
  srand(7);
  for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
  for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
  for (auto& v : sa) v = 127 - (rand() % 12);
  for (auto& v : sb) v = 127 - (rand() % 12);
  std::vector<float> fs(128);
  const float special[] = {0.f, 1.f, 1.0625f, 1.1875f, 447.f, 448.f, 449.f, 480.f, 1000.f, -1000.f, 0.001953125f, 0.0009765625f,
                           0.00146484375f, 1e-4f, 17.f, 18.f, 19.f, 0.3f, -0.3f, 240.f, 250.f, 260.f};
  for (int i = 0; i < 128; ++i) fs[i] = i < (int)(sizeof(special) / 4) ? special[i] : (rand() / (float)RAND_MAX - 0.5f) * 64.f;
  unsigned char *dA, *dB, *dsa, *dsb, *dfd; float *dD, *dfs;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 4096); hipMalloc(&dfs, 512);
  hipMalloc(&dfd, 128);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
  hipMemcpy(dfs, fs.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, 1, dfs, dfd);
  std::vector<float> D(1024); std::vector<unsigned char> fd(128);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(fd.data(), dfd, 128, hipMemcpyDeviceToHost);
  double worst = 0, worst_noscale = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, ref_ns = 0;
      for (int k = 0; k < 64; ++k) {
        const double p = (double)e4m3_to_float(A[i * 64 + k]) * e4m3_to_float(B[j * 64 + k]);
        ref += p * ldexp(1.0, (int)sa[i * 2 + k / 32] - 127) * ldexp(1.0, (int)sb[j * 2 + k / 32] - 127);
        ref_ns += p;
      }
      worst = fmax(worst, fabs(D[i * 32 + j] - ref) / (fabs(ref) + 1e-3));
      worst_noscale = fmax(worst_noscale, fabs(D[i * 32 + j] - ref_ns) / (fabs(ref_ns) + 1e-3));
    }
  printf("scaled f8 MFMA semantics: worst rel diff vs host model (lane half = k block, per-lane block scale) %.3e (ignoring scales: %.3e)\n",
         worst, worst_noscale);
  printf("cvt_pk_fp8_f32:");
  for (int i = 0; i < 22; ++i) printf(" %g->%g(0x%02x)", fs[i], e4m3_to_float(fd[i]), fd[i]);
  printf("\n");
  // (2)
  std::vector<int> h(32768);
  int* src; hipMalloc(&src, h.size() * 4);
  for (int pass = 0; pass < 2; ++pass) {
    srand(1);
    for (auto& v : h) {
      if (pass == 0) { v = 0; continue; }
      // two random halves in [-1, 1): valid f16 pairs; the f8 operands mask these words to small finite e4m3 bytes
      _Float16 a = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f), b = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
      unsigned short ua, ub; __builtin_memcpy(&ua, &a, 2); __builtin_memcpy(&ub, &b, 2);
      v = (int)(ua | ((unsigned)ub << 16));
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* name = pass == 0 ? "zeros" : "uniform";
    rate<0>(src, "6 x f16 (split-f16 x3)", name);
    rate<1>(src, "2 x f16 + 1 x scaled f8 (K=64)", name);
    rate<2>(src, "2 x f16 (single product)", name);
    rate<0>(src, "6 x f16 (split-f16 x3)", name);
    rate<1>(src, "2 x f16 + 1 x scaled f8 (K=64)", name);
    rate<3>(src, "2 x f16 + 1 x scaled f6 (K=64)", name);
    rate<3>(src, "2 x f16 + 1 x scaled f6 (K=64)", name);
  }
  return 0;
}
