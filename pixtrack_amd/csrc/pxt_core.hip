// Library-level entry points: version, error text, device query.
#include "pxt_common.h"

#include <cstdio>
#include <cstring>

namespace pxt {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}
}  // namespace pxt

extern "C" int pxt_version(void) { return 5; }
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
