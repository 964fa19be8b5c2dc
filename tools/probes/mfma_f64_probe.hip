// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (1024 FMAs per instruction) beside v_fma_f64 (64 per instruction):
// cycles per instruction per SIMD with 1, 2 and 4 waves on it, 8 independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double *out, long long *cyc, double a0, double b0, int n) {
  d4 acc[8];
  double s[8];
  for (int i = 0; i < 8; i++) { acc[i] = (d4){0, 0, 0, 0}; s[i] = 0.0; }
  double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = __builtin_fma(a, b, s[i]);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double r = 0;
  for (int i = 0; i < 8; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double *out; long long *cyc, h;
  hipMalloc(&out, 8 << 20); hipMalloc(&cyc, 8);
  const int n = 2000;
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  for (int mode = 0; mode < 2; mode++)
    for (int waves = 1; waves <= 4; waves *= 2) {   // waves per SIMD: block of 64*4*waves threads, one block per CU
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, 1.0, 2.0, 10);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, 1.0, 2.0, 10);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, 1.0, 2.0, n);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, 1.0, 2.0, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double insts_per_simd = (double)n * (mode == 0 ? 8 : 32) * waves;
      const double fma = insts_per_simd * (mode == 0 ? 1024 : 64) * 1024.0;
      printf("%s waves/SIMD %d: %.3f ms, %.1f TFLOP/s, s_memtime ticks per instruction per SIMD %.2f\n", mode == 0 ? "mfma_f64_16x16x4" : "v_fma_f64       ",
             waves, ms, 2.0 * fma / (ms * 1e-3) / 1e12, (double)h / insts_per_simd);
    }
  return 0;
}
