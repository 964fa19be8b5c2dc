// Does hipExtAnyOrderLaunch clear the barrier bit on gfx950?  Kernel A holds its block for ~30 us and stamps its end;
// kernel B stamps its start.  B launched plainly starts after A's end; B launched "any order" starts before it when
// the flag is honoured (hip_ext.h says it is not on GFX9xx boards: this probe is the measurement).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_hold(unsigned long long *t, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  t[0] = t0;
  t[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void k_stamp(unsigned long long *t) { t[2] = __builtin_amdgcn_s_memrealtime(); }
int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 3 * sizeof(*d));
  hipStream_t s;
  hipStreamCreate(&s);
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, s, d, 3000);
      if (mode == 0) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, d);
      else hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
      hipStreamSynchronize(s);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%s: A %.2f us long, B starts %.2f us after A's end\n", mode ? "any-order" : "plain    ", (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0);
    }
  return 0;
}
