// Does a second wave per SIMD hide the cost of issuing LDS-DMA between MFMAs?  (round 3)
// The split GEMM runs one wave per SIMD (512 registers each).  Its K step is 0.94 us of MFMA work, yet takes 1.6 us: the
// wave that feeds the matrix pipe also issues the step's 15 DMA instructions, and whenever the memory pipeline pushes
// back, its MFMA stream stops (tools/phase_probe.py: MFMA-only 1.0 us, DMA-only 1.2 us, together 1.6 us per step).
// This probe runs the same instruction mix per CU and K step -- 168 x 32x32x16 f16 MFMAs-equivalents (as 28 f16 + 14 scaled
// FP8 per 4 waves... here: f16 only, same pipe time) and 60 KiB of LDS-DMA from an L2-resident window, one barrier per
// step -- split over 4 waves (one per SIMD) or over 8 waves (two per SIMD, half the work each).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_dma_mix.hip -o tools/mfma_dma_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned int* lds_u32_ptr;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_u32_ptr dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, 0, 0, 0);
#endif
}

// NW waves per workgroup; per wave and step: NM MFMAs, NP DMA pieces (one after every NM / NP MFMAs); MODE 0 both, 1 MFMA only,
// 5 both, the DMA pieces addressed WITHOUT a VGPR (buffer resource with ADD_TID_ENABLE, stride 16: lane l reads base + soffset
// + 16 l -- a contiguous 1 KiB piece; does the per-instruction issue cost come from the 64-lane address operand?),
// 2 DMA only, 3 both with the pieces staged through VGPRs (buffer_load_dwordx4 of step s + 1 between the MFMAs of the first
// half of step s, ds_write_b128 between those of the second half), 4 role split: the first NW / 2 waves issue 2 NM MFMAs
// and no DMA, the others 2 NP pieces and no MFMA
template <int NW, int NM, int NP, int MODE>
__global__ __launch_bounds__(NW * 64, 1) void mix(const unsigned char* src, long long window, int iters, unsigned long long* out,
                                                   float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (long long)blockIdx.x * window;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)window, 0x00020000);
  const int lane_off = (lane >> 2) * 2048 + (lane & 3) * 16;           // 16 rows x 64 B (the GEMM's piece)
  // word1: stride 16 in bits 16..29; word3: ADD_TID_ENABLE (bit 23) on top of the raw-buffer flags
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 16, (int)(window / 16), 1 << 23);   // (with ADD_TID_ENABLE the DATA_FORMAT bits are stride[17:14]: keep them 0)
  constexpr int NACC = NM >= 8 ? 8 : NM;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * ((lane * 7 + e * 3) % 13 - 6));
    b[e] = (_Float16)(0.002f * ((lane * 5 + e) % 11 - 5));
  }
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  int off = wave * NP * 16 * 2048;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 stage[MODE == 3 ? (NP > 0 ? NP : 1) : 1];
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 3) {
      constexpr int HALF = NM / 2, EV = HALF / NP > 0 ? HALF / NP : 1;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
        if (m < HALF && m % EV == 0 && m / EV < NP) {
          const int w = m / EV;
          stage[w] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (off + w * 16 * 2048 + lane_off) & (int)(window - 1), 0, 0));
        }
        if (m >= HALF && (m - HALF) % EV == 0 && (m - HALF) / EV < NP) {
          const int w = (m - HALF) / EV;
          *reinterpret_cast<i32x4*>(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024 + lane * 16) = stage[w];
        }
      }
      off += NW * NP * 16 * 2048;
      __syncthreads();
      continue;
    }
    if constexpr (MODE == 5) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
        if (NP > 0 && (m % (NM / (NP > 0 ? NP : 1))) == 0 && m / (NM / (NP > 0 ? NP : 1)) < NP) {
          const int w = m / (NM / NP);
          const int so = __builtin_amdgcn_readfirstlane((off + w * 1024) & (int)(window - 1));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024), 16, 0, so, 0, 0);
        }
      }
      off += NW * NP * 1024;
      __syncthreads();
      continue;
    }
    if constexpr (MODE == 6) {                      // as MODE 0, but wave w issues its pieces w MFMAs later: no two waves issue a piece at the same time
      constexpr int EV = NM / (NP > 0 ? NP : 1);
      auto body = [&](auto woff) __attribute__((always_inline)) {
        constexpr int WOFF = decltype(woff)::value;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
          if (m >= WOFF && (m - WOFF) % EV == 0 && (m - WOFF) / EV < NP) {
            const int w = (m - WOFF) / EV;
            const int vo = (off + w * 16 * 2048 + lane_off) & (int)(window - 1);
            dma16(r, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024), vo);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      switch (wave & 3) {
        case 0: body(std::integral_constant<int, 0>{}); break;
        case 1: body(std::integral_constant<int, 1>{}); break;
        case 2: body(std::integral_constant<int, 2>{}); break;
        default: body(std::integral_constant<int, 3>{}); break;
      }
      off += NW * NP * 16 * 2048;
      __syncthreads();
      continue;
    }
    if constexpr (MODE == 8 || MODE == 9) {         // MODE 7 with OUT-OF-RANGE pieces (8: the DMA writes zeros, no memory traffic) / one lane only (9)
      constexpr int EV = NM / (NP > 0 ? NP : 1);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
        if (m % EV == 0 && m / EV < NP) {
          const int w = m / EV;
          const int vo = MODE == 8 ? 0x7fffffff : ((lane == 0 ? (off + w * 16 * 2048) & (int)(window - 1) : 0x7fffffff));
          dma16(r, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024), vo);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      off += NW * NP * 16 * 2048;
      __syncthreads();
      continue;
    }
    if constexpr (MODE == 7) {                      // MODE 0 with the same hard pins as MODE 6 (all waves aligned)
      constexpr int EV = NM / (NP > 0 ? NP : 1);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
        if (m % EV == 0 && m / EV < NP) {
          const int w = m / EV;
          const int vo = (off + w * 16 * 2048 + lane_off) & (int)(window - 1);
          dma16(r, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024), vo);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      off += NW * NP * 16 * 2048;
      __syncthreads();
      continue;
    }
    if constexpr (MODE == 4) {
      if (wave < NW / 2) {
#pragma unroll
        for (int m = 0; m < 2 * NM; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
      } else {
#pragma unroll
        for (int w = 0; w < 2 * NP; ++w)
          dma16(r, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + (wave - NW / 2) * 2 * NP + w) * 1024), (off + w * 16 * 2048 + lane_off) & (int)(window - 1));
      }
      off += NW * NP * 16 * 2048;
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (MODE != 2) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
      if (MODE != 1 && NP > 0 && (m % (NM / (NP > 0 ? NP : 1))) == 0 && m / (NM / (NP > 0 ? NP : 1)) < NP) {
        const int w = m / (NM / NP);
        const int vo = (off + w * 16 * 2048 + lane_off) & (int)(window - 1);
        dma16(r, (lds_u32_ptr)(sm + ((it & 1) * NW * NP + wave * NP + w) * 1024), vo);
      }
    }
    off += NW * NP * 16 * 2048;
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 1234.5f) sink[0] = s;
}

template <int NW, int NM, int NP, int MODE>
void run(const char* name, const unsigned char* d, unsigned long long* dout, float* sink, int grid, int iters) {
  const long long window = 2 << 20;
  const int smem = 2 * NW * (NP > 0 ? NP : 1) * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix<NW, NM, NP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((mix<NW, NM, NP, MODE>), dim3(grid), dim3(NW * 64), smem, 0, d, window, iters, dout, sink);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpy(h, dout, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < grid; ++i) sum += (double)h[i];
  const double us = sum / grid / 100.0;
  printf("%-44s grid %3d: %6.3f us per step  (%d waves x %d MFMA + %d DMA pieces)\n", name, grid, us / iters, NW, NM, NP);
}

// does the TID-addressed resource deliver lane l the 16 bytes at base + soffset + 16 l?
__global__ void verify_tid(const unsigned* src, int bytes, int soff, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned buf[256];
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 16, bytes / 16, 1 << 23);
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_u32_ptr)buf, 16, 0, soff, 0, 0);
#endif
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}

int main() {
  {
    unsigned *vs, *vo, h[1024], o[256];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&vs, 4096);
    hipMalloc(&vo, 1024);
    hipMemcpy(vs, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(verify_tid, dim3(1), dim3(64), 0, 0, vs, 4096, 2048, vo);
    hipMemcpy(o, vo, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += o[i] != (unsigned)(512 + i);
    printf("TID-addressed LDS-DMA (no VGPR operand): %s (word 0 = %u, word 255 = %u; expected 512 .. 767)\n", bad ? "WRONG DATA" : "data correct", o[0], o[255]);
  }
  unsigned char* d;
  unsigned long long* dout;
  float* sink;
  const long long total = 256LL * (2 << 20);
  if (hipMalloc(&d, total) != hipSuccess) return 1;
  hipMemset(d, 1, total);
  hipMalloc(&dout, 512 * 8);
  hipMalloc(&sink, 4);
  const int iters = 400;
  for (int grid : {8, 232}) {
    // 56 f16 MFMAs per wave = the K step's MFMA pipe time of 28 f16 + 14 FP8 (2x) instructions
    run<4, 56, 0, 1>("4 waves, MFMA only", d, dout, sink, grid, iters);
    run<4, 56, 14, 2>("4 waves, DMA only (14 pieces/wave)", d, dout, sink, grid, iters);
    run<4, 56, 14, 0>("4 waves, MFMA + DMA interleaved", d, dout, sink, grid, iters);
    run<8, 28, 0, 1>("8 waves, MFMA only", d, dout, sink, grid, iters);
    run<8, 28, 7, 2>("8 waves, DMA only (7 pieces/wave)", d, dout, sink, grid, iters);
    run<8, 28, 7, 0>("8 waves, MFMA + DMA interleaved", d, dout, sink, grid, iters);
    run<4, 56, 7, 0>("4 waves, MFMA + half the DMA", d, dout, sink, grid, iters);
    run<4, 56, 14, 3>("4 waves, MFMA + pieces staged through VGPRs", d, dout, sink, grid, iters);
    run<8, 28, 7, 3>("8 waves, MFMA + pieces staged through VGPRs", d, dout, sink, grid, iters);
    run<8, 28, 7, 4>("8 waves, role split (4 MFMA waves, 4 DMA waves)", d, dout, sink, grid, iters);
    run<4, 56, 14, 5>("4 waves, MFMA + DMA addressed by TID (no VGPR)", d, dout, sink, grid, iters);
    run<4, 56, 14, 7>("4 waves, MFMA + DMA, pinned, waves aligned", d, dout, sink, grid, iters);
    run<4, 56, 14, 6>("4 waves, MFMA + DMA, pinned, waves STAGGERED by one MFMA", d, dout, sink, grid, iters);
    run<4, 56, 14, 8>("4 waves, MFMA + 14 OUT-OF-RANGE DMA pieces (no traffic)", d, dout, sink, grid, iters);
    run<4, 56, 14, 9>("4 waves, MFMA + 14 DMA pieces with ONE lane in range", d, dout, sink, grid, iters);
    run<4, 56, 7, 7>("4 waves, MFMA + half the DMA, pinned, aligned", d, dout, sink, grid, iters);
    run<4, 56, 7, 6>("4 waves, MFMA + half the DMA, pinned, STAGGERED", d, dout, sink, grid, iters);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
