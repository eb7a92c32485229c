// product scheme 2 of the wide-tile split conv GEMM (rowgemm_h3w_kernel.h): f16 hi.hi + both cross terms in one block-scaled FP8 MFMA (default)
#include "rowgemm_h3w_kernel.h"

namespace radmmm {
int launch_h3d_pr2(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  return launch_pr<2>(mb, ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm

#ifdef RADMMM_PHASE_TIMERS
// measurement builds only: copy the first n workgroups' four time stamps (100 MHz ticks) to host memory
extern "C" int radmmm_debug_phase_read(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase), (size_t)n * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
