// A HIP stream whose kernels may only use a subset of the CUs (hipExtStreamCreateWithCUMask): lets a long GEMM backlog
// run beside a chain of small dependent launches (the LSTM recurrence) without taking every CU away from it.
#include <stdint.h>

#include "common.h"

// Create a stream restricted to `enabled_cus` of the device's CUs (bits 0 .. enabled_cus-1 of the mask; how mask bits
// map to XCDs is the driver's business).  *out receives the hipStream_t.  The stream lives until radmmm_stream_destroy.
extern "C" int radmmm_stream_create_masked(int enabled_cus, void** out) {
  RADMMM_REQUIRE(out && enabled_cus >= 8, "stream_create_masked: bad arguments");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    radmmm::set_error("stream_create_masked: device query failed");
    return -2;
  }
  if (enabled_cus > cus) enabled_cus = cus;
  uint32_t mask[16] = {0};
  for (int i = 0; i < enabled_cus && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((cus + 31) / 32), mask);
  if (e != hipSuccess) {
    radmmm::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return -2;
  }
  *out = s;
  return 0;
}

extern "C" int radmmm_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy(static_cast<hipStream_t>(stream)) != hipSuccess) {
    radmmm::set_error("hipStreamDestroy failed");
    return -2;
  }
  return 0;
}
