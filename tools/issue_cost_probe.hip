// What does ONE wave per SIMD pay for the instructions it issues between its MFMAs?  (round 4)
// The wide split GEMMs run one wave per SIMD (the accumulators need 224 of its 512 registers), so everything the K loop
// needs besides the MFMAs -- fragment reads, LDS-DMA, address arithmetic, waits -- is issued by the wave that feeds the
// matrix pipe, in order.  An MFMA occupies the pipe for 32 (32x32x16 f16) or 64 cycles (scaled 32x32x64 fp8); whatever is
// issued behind it is free as long as it fits under those cycles.  This probe measures how much fits: a loop of
//     MFMA ; N x <instruction of one kind>          (everything asm volatile: the order is the source order)
// on 4 waves per workgroup, one workgroup per CU, for N = 0 .. 8, and prints ns per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/issue_cost_probe.hip -o tools/issue_cost_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

enum { K_NONE, K_VALU, K_VALU_DEP, K_SALU, K_DSR128, K_DSR64, K_DMA, K_WAIT, K_NOP, K_DSR128_DMA, K_VXAD_DSR };

template <int KIND>
__device__ __forceinline__ void filler(int& v0, int& v1, int& s0, i32x4& frag, int addr, __amdgpu_buffer_rsrc_t r, int vo, int lds_dst, int j) {
  if constexpr (KIND == K_VALU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(j & 1 ? v0 : v1) : "v"(addr));
  if constexpr (KIND == K_VALU_DEP) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(addr));
  if constexpr (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
  if constexpr (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"(addr));
  if constexpr (KIND == K_DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<long long*>(&frag)) : "v"(addr));
  if constexpr (KIND == K_DMA) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(vo), "s"(r), "s"(lds_dst) : "memory");
  if constexpr (KIND == K_WAIT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
  if constexpr (KIND == K_VXAD_DSR) {                                   // address arithmetic feeding a read (dependent pair)
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(v1) : "v"(addr), "v"(v0));
    asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"(v1));
  }
}

// MF = 0: 32x32x16 f16 (8 passes), 1: scaled 32x32x64 fp8 (16 passes); N fillers of KIND behind every MFMA
template <int MF, int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * ((lane * 7 + e * 3) % 13 - 6));
    b[e] = (_Float16)(0.002f * ((lane * 5 + e) % 11 - 5));
  }
  i32x8 a8, b8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a8[e] = 0x38383838 + lane * 0x01010101 * (e & 1); b8[e] = 0x3c3c3c3c; }
  const int sc = 127;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src + (size_t)blockIdx.x * (1 << 20)), 0, 1 << 20, 0x00020000);
  int v0 = lane, v1 = 0, s0 = 0;
  i32x4 frag = {0, 0, 0, 0};
  const int addr = wave * 16384 + (lane & 31) * 64 + ((lane >> 5) << 4);      // conflict-free ds_read_b128 pattern of the GEMM (swizzle aside)
  const int vo0 = (lane >> 2) * 2048 + (lane & 3) * 16;
  const int lds_dst = 65536 + wave * 16384;
  for (int w = threadIdx.x; w < 16384; w += 256) reinterpret_cast<int*>(sm)[w] = w;
  __syncthreads();
  int vo = vo0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (MF == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc[j]) : "v"(a8), "v"(b8), "v"(sc));
#pragma unroll
      for (int q = 0; q < N; ++q) filler<KIND>(v0, v1, s0, frag, addr + ((q & 3) << 11), r, vo + q * 32768, lds_dst + ((j * N + q) & 15) * 1024, q);
    }
    if constexpr (KIND == K_DSR128 || KIND == K_DSR64 || KIND == K_VXAD_DSR) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(frag));
    if constexpr (KIND == K_DMA) { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N >= 4 ? 32 : 8 * N) : "memory"); vo = (vo + 64) & ((1 << 19) - 1); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = v0 + v1 + s0 + frag[0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 1234.5f) sink[0] = s;
}

template <int MF, int KIND, int N>
double run(const unsigned char* d, float* sink, int grid) {
  const int smem = 160 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MF, KIND, N>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int iters = 3000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, KIND, N>), dim3(grid), dim3(256), smem, 0, d, 100, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MF, KIND, N>), dim3(grid), dim3(256), smem, 0, d, iters, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (iters * 8.0);      // ns per MFMA
}

template <int MF, int KIND>
void sweep(const char* name, const unsigned char* d, float* sink, int grid) {
  printf("%-6s %-34s grid %3d  ns per MFMA at N = 0,1,2,3,4,6,8: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", MF ? "fp8x64" : "f16x16", name, grid,
         run<MF, KIND, 0>(d, sink, grid), run<MF, KIND, 1>(d, sink, grid), run<MF, KIND, 2>(d, sink, grid), run<MF, KIND, 3>(d, sink, grid),
         run<MF, KIND, 4>(d, sink, grid), run<MF, KIND, 6>(d, sink, grid), run<MF, KIND, 8>(d, sink, grid));
  fflush(stdout);
}

int main() {
  unsigned char* d;
  float* sink;
  if (hipMalloc(&d, 256u << 20) != hipSuccess) return 1;
  (void)hipMemset(d, 1, 256u << 20);
  (void)hipMalloc(&sink, 4);
  for (int grid : {8, 232}) {
    sweep<0, K_VALU>("independent v_add_u32", d, sink, grid);
    sweep<0, K_VALU_DEP>("dependent v_add_u32 chain", d, sink, grid);
    sweep<0, K_SALU>("s_add_u32", d, sink, grid);
    sweep<0, K_NOP>("s_nop 0", d, sink, grid);
    sweep<0, K_WAIT>("s_waitcnt (nothing outstanding)", d, sink, grid);
    sweep<0, K_DSR128>("ds_read_b128", d, sink, grid);
    sweep<0, K_DSR64>("ds_read_b64", d, sink, grid);
    sweep<0, K_VXAD_DSR>("v_add + dependent ds_read_b128", d, sink, grid);
    sweep<0, K_DMA>("s_mov m0 + buffer_load x4 lds", d, sink, grid);
    sweep<1, K_VALU>("independent v_add_u32", d, sink, grid);
    sweep<1, K_DSR128>("ds_read_b128", d, sink, grid);
    sweep<1, K_VXAD_DSR>("v_add + dependent ds_read_b128", d, sink, grid);
    sweep<1, K_DMA>("s_mov m0 + buffer_load x4 lds", d, sink, grid);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
