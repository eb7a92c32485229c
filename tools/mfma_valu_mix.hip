// Does ordinary VALU work issued between fp32 MFMAs slow the MFMA stream down?
// Each wave runs 16x16x4 f32 MFMAs on 8 independent accumulators with NV independent v_fma per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c[j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float& r = v[(j + q) & 7];
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + v[i];
  if (s == 123.456f) out[0] = s;
}

template <int NV>
void run(int blocks_per_cu) {
  float* d;
  (void)hipMalloc(&d, 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, d, 50);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 8.0 * 2048.0;
  printf("NV=%d valu/mfma  blocks/CU=%d  %.3f ms  %.1f TFLOP/s (mfma)\n", NV, blocks_per_cu, ms, flops / ms / 1e9);
  (void)hipFree(d);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0>(b); run<1>(b); run<2>(b); run<3>(b); run<4>(b); run<6>(b); run<8>(b);
  }
  return 0;
}
