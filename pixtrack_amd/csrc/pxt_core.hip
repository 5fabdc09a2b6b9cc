// Library-level entry points: version, error text, device query.
#include "pxt_common.h"

#include <cstdio>
#include <cstring>
#include <mutex>

namespace pxt {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

// The library's side streams: THREE per device, shared by every context (the second NeRF pipeline and the
// second UNet pass use #0 - they never overlap in a frame -, the two passes' coarse heads #1 and #2).  HIP maps
// streams onto 4 hardware queues round-robin in creation order; with a private set per context (6 streams per
// tracker) two streams that must run side by side often landed on ONE queue and silently serialised: the same
// tracker ran at 454 / 367 / 478 frames/s depending on how many other trackers had been created before it.
// Caller's stream + these three = four queues.  Never destroyed (process lifetime).
hipStream_t shared_side_stream(int i) {
  static std::mutex mu;
  static hipStream_t pool[16][3] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || i < 0 || i > 2) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!pool[dev][i] && hipStreamCreateWithFlags(&pool[dev][i], hipStreamNonBlocking) != hipSuccess) pool[dev][i] = nullptr;
  return pool[dev][i];
}
}  // namespace pxt

extern "C" int pxt_version(void) { return 11; }
extern "C" const char* pxt_last_error(void) { return pxt::g_last_error; }
extern "C" int pxt_device_cus(int* n_cus) {
  if (!n_cus) return PXT_E_ARG;
  int dev = 0;
  PXT_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PXT_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  *n_cus = prop.multiProcessorCount;
  return PXT_OK;
}
