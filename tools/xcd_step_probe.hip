// What would one recurrent LSTM step cost if a direction's 32 workgroups lived on ONE XCD (weights resident in LDS,
// h exchanged through that XCD's L2, an XCD-local barrier per step) instead of one launch per step?
// Synthetic kernel with the forward step's communication pattern only (no arithmetic to speak of):
//   256 workgroups x 256 threads, 140 KB of LDS each (one per CU); a workgroup reads HW_REG_XCC_ID, takes a ticket on
//   its XCD; XCDs 0 and 1 keep their first 32 arrivals, everybody else exits.  Then `steps` times:
//     write own slice of h (17 units x 32 batch rows x hi/lo fp16 = 2.2 KB, plain stores) into the ping-pong buffer,
//     barrier among the 32 workgroups of the XCD (counter in global memory, agent-scope atomic add, sc1 poll),
//     read the whole h (2 x 33 KB) with L1-bypassing loads and check it.
// Prints the time per step and whether every read saw the step's data.
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_step_probe.hip -o /tmp/xcd_step_probe && /tmp/xcd_step_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

constexpr int NWG = 32;            // workgroups per direction (= CUs of one XCD)
constexpr int H = 544;             // 17 units x 32 workgroups (>= 524)
constexpr int UPW = 17;
constexpr int B = 32;
constexpr unsigned SPIN_LIMIT = 200000;     // ~0.1 s of polling, then every workgroup gives up

struct Args {
  unsigned* ticket;      // [8]
  unsigned* counter;     // [8] monotonically increasing arrival counters
  unsigned short* hbuf;  // [2 dirs][2 ping-pong][2 hi/lo][B][H]
  unsigned* bad;         // mismatches seen
  unsigned* timeout;     // spin limit hit
  int steps;
  int mode;              // 0: sc1 loads, 1: plain loads (expected stale)
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ __launch_bounds__(256) void probe_kernel(Args a) {
  extern __shared__ unsigned char lds[];
  __shared__ int s_role;
  const int tid = threadIdx.x;
  const unsigned xcc = xcc_id();
  if (tid == 0) {
    int role = -1;
    if (xcc < 2) {
      const unsigned t = atomicAdd(&a.ticket[xcc], 1u);
      if (t < NWG) role = (int)t;
    }
    s_role = role;
  }
  __syncthreads();
  const int role = s_role;
  if (role < 0) return;
  lds[tid] = (unsigned char)tid;                       // touch the big allocation
  const int d = (int)xcc;
  unsigned* ctr = a.counter + d;
  unsigned bad = 0;
  __shared__ int s_abort;
  if (tid == 0) s_abort = 0;
  for (int s = 0; s < a.steps; ++s) {
    if (s_abort) break;                                  // uniform: written before the previous trip's last barrier
    unsigned short* dst = a.hbuf + ((size_t)(d * 2 + (s & 1)) * 2) * B * H;
    // own slice: 17 units x 32 rows, hi and lo; value encodes (step, unit)
    for (int i = tid; i < UPW * B * 2; i += 256) {
      const int hl = i / (UPW * B), r = (i / UPW) % B, u = role * UPW + i % UPW;
      dst[((size_t)hl * B + r) * H + u] = (unsigned short)((s * 7 + u + hl) & 0xffff);
    }
    __syncthreads();
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)NWG * (unsigned)(s + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          atomicAdd(a.timeout, 1u);
          s_abort = 1;
          break;
        }
      }
    }
    __syncthreads();
    // read everything back: 2 x B x H halves = 69.6 KB as 16-byte pieces, all of a thread's loads in flight at once
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 2 * B * H * 2, 0x00020000);
    constexpr int n16 = 2 * B * H * 2 / 16, PER = (n16 + 255) / 256;
    u32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + 256 * k;
      v[k] = a.mode == 0 ? __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 16)      // aux 16 = sc1: bypass this CU's L1
                         : __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 0);       // (out of range reads as zeros)
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + 256 * k;
      if (i < n16) {
        const int e0 = 8 * i;                          // element 8 i of the flat [hl][r][u] array
        const int hl = e0 / (B * H), u0 = e0 % H;
        if ((unsigned short)(v[k][0] & 0xffff) != (unsigned short)((s * 7 + u0 + hl) & 0xffff)) ++bad;
      }
    }
  }
  if (bad) atomicAdd(a.bad, bad);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 400;
  Args a{};
  CK(hipMalloc(&a.ticket, 64));
  CK(hipMalloc(&a.counter, 64));
  CK(hipMalloc(&a.bad, 4));
  CK(hipMalloc(&a.timeout, 4));
  CK(hipMalloc(&a.hbuf, (size_t)2 * 2 * 2 * B * H * 2));
  a.steps = steps;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      a.mode = mode;
      CK(hipMemset(a.ticket, 0, 64));
      CK(hipMemset(a.counter, 0, 64));
      CK(hipMemset(a.bad, 0, 4));
      CK(hipMemset(a.timeout, 0, 4));
      CK(hipMemset(a.hbuf, 0xff, (size_t)2 * 2 * 2 * B * H * 2));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe_kernel, dim3(256), dim3(256), 140 * 1024, 0, a);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned bad = 0, to = 0, tk[16];
      CK(hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&to, a.timeout, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(tk, a.ticket, 64, hipMemcpyDeviceToHost));
      printf("{\"mode\": \"%s\", \"steps\": %d, \"total_ms\": %.3f, \"us_per_step\": %.3f, \"mismatches\": %u, \"timeouts\": %u, "
             "\"arrivals_xcc0\": %u, \"arrivals_xcc1\": %u}\n",
             mode == 0 ? "sc1 loads" : "plain loads", steps, ms, ms * 1e3 / steps, bad, to, tk[0], tk[1]);
    }
  }
  return 0;
}
