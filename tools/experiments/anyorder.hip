// anyorder.hip -- does hipExtAnyOrderLaunch let two independent launches on ONE stream overlap on gfx950?  (hip_ext.h says the flag
// "is not supported on AMD GFX9xx boards".)  Two launches of a spin kernel that each fill a quarter of the chip's wave slots:
// back to back they take 2 x t if the second waits for the first, ~t if it does not.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/anyorder tools/experiments/anyorder.hip && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  unsigned long long* out;
  CK(hipMalloc(&out, 64));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long ticks = 20000;      // 200 us at 100 MHz
  for (int flags = 0; flags <= 1; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, ticks, out + i);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("flags = %d: four 200-us launches of 256 workgroups on one stream: %.3f ms\n", flags, ms);
    }
  }
  return 0;
}
