// Which physical CU does bit b of a hipExtStreamCreateWithCUMask mask enable?  (round 4: a side stream confined to a few CUs
// per XCD for the memory-bound weight passes, beside GEMM grids that leave 24 of the 256 CUs idle)
// For every bit: a stream with that bit only, one workgroup, which reports HW_REG_XCC_ID and HW_REG_HW_ID (SE / SH / CU ids).
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void who(unsigned* out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[0] = xcc; out[1] = hw; }
}

int main() {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned* d;
  (void)hipMalloc(&d, 8);
  printf("CUs %d\nbit: xcc  hw_id(cu,sh,se)\n", cus);
  for (int b = 0; b < cus; ++b) {
    uint32_t mask[16] = {0};
    mask[b >> 5] = 1u << (b & 31);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)((cus + 31) / 32), mask) != hipSuccess) { printf("%d: create failed\n", b); continue; }
    (void)hipMemsetAsync(d, 0xff, 8, s);
    hipLaunchKernelGGL(who, dim3(1), dim3(64), 0, s, d);
    unsigned h[2] = {0, 0};
    if (hipStreamSynchronize(s) != hipSuccess) { printf("%d: sync failed\n", b); (void)hipStreamDestroy(s); continue; }
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%3d: xcc %u  cu %u sh %u se %u\n", b, h[0] & 0xf, (h[1] >> 8) & 0xf, (h[1] >> 12) & 0x1, (h[1] >> 13) & 0x7);
    (void)hipStreamDestroy(s);
  }
  return 0;
}
