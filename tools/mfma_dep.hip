// Micro-benchmark: v_mfma_f32_32x32x16_f16 rate vs accumulator dependency distance and operand data.
//   NACC independent accumulators used round-robin -> a dependent MFMA every NACC issues.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o gpurun_out/mfma_dep && gpurun_out/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const f16x8* __restrict__ src, float* out, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(threadIdx.x + 256 * i) & 4095];
    b[i] = src[(threadIdx.x + 256 * i + 1024 + blockIdx.x) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 12 / NACC + (12 % NACC ? 1 : 0); ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i + 2 * r) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run(const f16x8* src, const char* data, int blocks_per_cu) {
  float* d; hipMalloc(&d, 4);
  const int iters = 4000;
  const int per_iter = (12 / NACC + (12 % NACC ? 1 : 0)) * NACC;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, src, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, src, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * per_iter * 2.0 * 32 * 32 * 16;
  printf("nacc=%2d data=%-8s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", NACC, data, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(d);
}

int main() {
  std::vector<_Float16> h(4096 * 8);
  f16x8* src; hipMalloc(&src, h.size() * 2);
  for (int pass = 0; pass < 2; ++pass) {
    srand(1);
    for (auto& v : h) v = pass == 0 ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* name = pass == 0 ? "zeros" : "uniform";
    for (int b = 1; b <= 2; ++b) {
      run<1>(src, name, b); run<2>(src, name, b); run<3>(src, name, b); run<4>(src, name, b); run<6>(src, name, b); run<12>(src, name, b);
    }
  }
  return 0;
}
