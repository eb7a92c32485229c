// Operand / block-scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3), found by experiment:
// each lane l (row/col i = l & 31, half h = l >> 5) supplies 32 bytes (8 VGPRs) and one E8M0 scale byte.
//   pairing : A one-hot at (i, h, q) against B row 0 holding 64 distinct values -> which B position it multiplies
//   scale_a : A one-hot at (i, h, q), all-ones B, scale bytes of lanes (i,0) / (i,1) = 2^-3 / 2^-6 -> whose scale applies
//   scale_b : same with the roles swapped
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f8_layout.hip -o /tmp/f8_layout && /tmp/f8_layout
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

static float e4m3_to_float(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -v : v;
}

// A, B: [32 rows][2 halves][32 bytes]; sa, sb: [32][2] scale bytes; D [32][32]
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  i32x8 a, b;
  const int* pa = reinterpret_cast<const int*>(A + (i * 64 + h * 32));
  const int* pb = reinterpret_cast<const int*>(B + (i * 64 + h * 32));
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = pa[q]; b[q] = pb[q]; }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, (int)sa[i * 2 + h], 0, (int)sb[i * 2 + h]);
#pragma unroll
  for (int e = 0; e < 16; ++e) D[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + i] = acc[e];
}

int main() {
  unsigned char *dA, *dB, *dsa, *dsb; float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 4096);
  std::vector<unsigned char> A(2048), B(2048), sa(64, 127), sb(64, 127);
  std::vector<float> D(1024);
  auto run = [&]() {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  };
  const unsigned char ONE = 0x38;
  // 64 distinct positive e4m3 values: bytes 0x08 .. 0x47
  unsigned char val[64];
  for (int p = 0; p < 64; ++p) val[p] = 0x08 + p;
  // sanity
  std::fill(A.begin(), A.end(), ONE); std::fill(B.begin(), B.end(), ONE);
  run();
  printf("all ones, unit scales: D[0][0] = %g (expect 64), D[5][7] = %g\n", D[0], D[5 * 32 + 7]);
  // pairing
  for (int h = 0; h < 2; ++h) {
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
    for (int i = 0; i < 32; ++i) A[i * 64 + h * 32 + i] = ONE;          // row i: one-hot at (h, q = i)
    for (int p = 0; p < 64; ++p) B[0 * 64 + p] = val[p];                 // col 0: position p = h'*32 + q'
    run();
    printf("pairing: A(h=%d, q) multiplies B position (h'*32+q'):", h);
    for (int i = 0; i < 32; ++i) {
      int found = -1;
      for (int p = 0; p < 64; ++p) if (fabsf(D[i * 32 + 0] - e4m3_to_float(val[p])) < 1e-6f) found = p;
      printf(" %d", found);
    }
    printf("\n");
  }
  // scale_a: lanes (i,0) -> 2^-3, (i,1) -> 2^-6
  for (int h = 0; h < 2; ++h) {
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), ONE);
    for (int i = 0; i < 32; ++i) { A[i * 64 + h * 32 + i] = ONE; sa[i * 2] = 124; sa[i * 2 + 1] = 121; }
    std::fill(sb.begin(), sb.end(), 127);
    run();
    printf("scale_a applied to A(h=%d, q) [log2]:", h);
    for (int i = 0; i < 32; ++i) printf(" %g", log2f(D[i * 32 + 0]));
    printf("\n");
  }
  std::fill(sa.begin(), sa.end(), 127);
  for (int h = 0; h < 2; ++h) {
    std::fill(B.begin(), B.end(), 0); std::fill(A.begin(), A.end(), ONE);
    for (int j = 0; j < 32; ++j) { B[j * 64 + h * 32 + j] = ONE; sb[j * 2] = 124; sb[j * 2 + 1] = 121; }
    run();
    printf("scale_b applied to B(h=%d, q) [log2]:", h);
    for (int j = 0; j < 32; ++j) printf(" %g", log2f(D[0 * 32 + j]));
    printf("\n");
  }
  // does the scale of lane (i, h) depend on other rows' lanes?  row 3 only scaled
  std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
  std::fill(A.begin(), A.end(), ONE); std::fill(B.begin(), B.end(), ONE);
  sa[3 * 2 + 0] = 124;
  run();
  printf("scale_a lane (3,0) = 2^-3, all ones: D[3][0] = %g, D[4][0] = %g, D[2][0] = %g (36 = 32 + 32/8: one 32-block of row 3)\n",
         D[3 * 32], D[4 * 32], D[2 * 32]);
  sa[3 * 2 + 0] = 127; sa[3 * 2 + 1] = 124;
  run();
  printf("scale_a lane (3,1) = 2^-3, all ones: D[3][0] = %g, D[4][0] = %g\n", D[3 * 32], D[4 * 32]);
  return 0;
}
