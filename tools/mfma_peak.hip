// Micro-benchmark: sustained MFMA rates on this MI355X (ground truth for roofline peaks).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  f16x8 h0, h1;
  s16x8 b0, b1;
  for (int i = 0; i < 8; ++i) { h0[i] = (_Float16)(x + i); h1[i] = (_Float16)(y - i); b0[i] = (short)(threadIdx.x + i); b1[i] = (short)(blockIdx.x + i); }
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    } else if (MODE == 1) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, c3, 0, 0, 0);
    } else if (MODE == 2) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h0, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h1, a3, 0, 0, 0);
    } else if (MODE == 3) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, b1, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, b0, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, b0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, b1, a3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, h0, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h0, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, h1, c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int blocks_per_cu) {
  float* d; hipMalloc(&d, 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * flop_per_mfma;
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0>("f32 32x32x2", 2.0 * 32 * 32 * 2, b);
    run<1>("f32 16x16x4", 2.0 * 16 * 16 * 4, b);
    run<2>("f16 32x32x16", 2.0 * 32 * 32 * 16, b);
    run<3>("bf16 32x32x16", 2.0 * 32 * 32 * 16, b);
    run<4>("f16 16x16x32", 2.0 * 16 * 16 * 32, b);
  }
  return 0;
}
