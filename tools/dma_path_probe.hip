// Per-CU operand-delivery rates (round 3): how fast can ONE workgroup of 4 waves (the GEMM's shape: one workgroup per
// CU) pull L2-resident data, per path and access pattern?
//   path  D: global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 1 KiB per wave instruction
//         V: global -> VGPR (buffer_load_dwordx4), 1 KiB per wave instruction
//         M: half the instructions of each kind
//   pattern  rows64: a wave instruction covers 16 rows x 64 B (row pitch 2 KiB) -- the GEMM's K-step slice of an operand
//            rows128: 8 rows x 128 B;  contig: 1 KiB contiguous;  xor-permuted: the 16-byte chunks of a row fetched in the
//            GEMM's swizzled lane order (source-side XOR for conflict-free ds_read_b128)
// Every workgroup streams through its own 2 MiB window (L2 resident after the first pass); `npieces` instructions per
// wave and iteration, then s_waitcnt vmcnt(0) + barrier (the GEMM's step structure).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_path_probe.hip -o tools/dma_path_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

template <int PATH, int PAT, int NP>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* src, long long window, int iters, unsigned long long* out, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (long long)blockIdx.x * window;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)window, 0x00020000);
  // per-lane offset within a piece
  int lane_off;
  if (PAT == 0) lane_off = (lane >> 2) * 2048 + (lane & 3) * 16;        // 16 rows x 64 B
  else if (PAT == 1) lane_off = (lane >> 3) * 2048 + (lane & 7) * 16;   // 8 rows x 128 B
  else if (PAT == 2) lane_off = lane * 16;                              // contiguous
  else if (PAT == 3) lane_off = (lane >> 2) * 2048 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);   // rows64, the GEMM's XOR-permuted quads
  else lane_off = (lane >> 3) * 2048 + (((lane & 7) ^ ((lane >> 4) & 3) ^ (((lane >> 3) & 1) << 2)) * 16);   // rows128, permuted octets
  const int piece_stride = PAT == 2 ? 1024 : ((PAT == 0 || PAT == 3) ? 16 * 2048 : 8 * 2048);
  i32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  int off = wave * NP * piece_stride;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const int vo = (off + w * piece_stride + lane_off) & (int)(window - 1);
      const bool use_dma = PATH == 0 || (PATH == 2 && (w & 1) == 0);
      if (use_dma) {
        dma16(r, (lds_u32_ptr)(sm + ((it & 1) * 4 * NP + wave * NP + w) * 1024), vo);
      } else {
        const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0));
        acc += v;
      }
    }
    off += 4 * NP * piece_stride;
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x7fffffff) sink[0] = 1;
}

template <int PATH, int PAT, int NP>
void run(const char* name, const unsigned char* d, unsigned long long* dout, int* sink, int grid, int iters) {
  const long long window = 2 << 20;
  const int smem = 2 * 4 * NP * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PATH, PAT, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((probe<PATH, PAT, NP>), dim3(grid), dim3(256), smem, 0, d, window, iters, dout, sink);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpy(h, dout, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double worst = 0, sum = 0;
  for (int i = 0; i < grid; ++i) {
    sum += (double)h[i];
    if ((double)h[i] > worst) worst = (double)h[i];
  }
  const double us = sum / grid / 100.0, bytes = (double)iters * 4 * NP * 1024;
  printf("%-28s grid %3d  %2d pieces/wave/iter: %7.1f us (slowest %7.1f)  %6.1f GB/s per CU  %6.3f us per iteration\n", name, grid, NP, us,
         worst / 100.0, bytes / us * 1e-3, us / iters);
}

int main() {
  unsigned char* d;
  unsigned long long* dout;
  int* sink;
  const long long total = 256LL * (2 << 20);
  if (hipMalloc(&d, total) != hipSuccess) return 1;
  hipMemset(d, 1, total);
  hipMalloc(&dout, 512 * 8);
  hipMalloc(&sink, 4);
  const int iters = 400;
  for (int grid : {8, 256}) {
    run<0, 0, 15>("DMA   rows64", d, dout, sink, grid, iters);
    run<0, 3, 15>("DMA   rows64 xor-permuted", d, dout, sink, grid, iters);
    run<0, 1, 15>("DMA   rows128", d, dout, sink, grid, iters);
    run<0, 4, 15>("DMA   rows128 xor-permuted", d, dout, sink, grid, iters);
    run<0, 2, 15>("DMA   contig", d, dout, sink, grid, iters);
    run<1, 0, 15>("VGPR  rows64", d, dout, sink, grid, iters);
    run<1, 1, 15>("VGPR  rows128", d, dout, sink, grid, iters);
    run<1, 2, 15>("VGPR  contig", d, dout, sink, grid, iters);
    run<2, 0, 16>("MIX   rows64 (8 DMA + 8 VGPR)", d, dout, sink, grid, iters);
    run<0, 0, 8>("DMA   rows64", d, dout, sink, grid, iters);
    run<0, 0, 11>("DMA   rows64", d, dout, sink, grid, iters);
    run<1, 0, 8>("VGPR  rows64", d, dout, sink, grid, iters);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
