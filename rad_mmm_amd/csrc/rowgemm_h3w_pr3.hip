// product scheme 3 of the wide-tile split conv GEMM (rowgemm_h3w_kernel.h): three f16 products hi.hi + hi.lo + lo.hi
#include "rowgemm_h3w_kernel.h"

namespace radmmm {
int launch_h3d_pr3(int mb, int ek, const radmmm_rowgemm_h3_desc& d, hipStream_t stream, int a_bytes, int b_bytes) {
  return launch_pr<3>(mb, ek, d, stream, a_bytes, b_bytes);
}
}  // namespace radmmm
