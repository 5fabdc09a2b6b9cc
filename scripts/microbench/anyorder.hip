// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two spin kernels of 64 workgroups x ~200 us: back to back = 400 us, overlapped = 200 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const long long cyc = 400000;  // s_memtime ticks at the shader clock (~2 GHz): ~200 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 4; ++i) {
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, cyc, (int*)nullptr);
        else hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, (mode == 2 && (i & 1)) ? hipExtAnyOrderLaunch : 0, cyc, (int*)nullptr);
      }
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s) rep %d: 4 x 200-us kernels took %.3f ms\n", mode,
             mode == 0 ? "hipLaunchKernelGGL" : mode == 1 ? "hipExtLaunchKernelGGL, flags 0" : "every second launch hipExtAnyOrderLaunch", rep, ms);
    }
  }
  return 0;
}
