// Developer microbenchmark: issue cost / dependent latency of the f64 instructions the hot
// kernels are made of, one wave per SIMD (gfx950).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

template <int OP, bool DEP>
__global__ void k(double *out, long long *cyc, double a0, double b0, int outer = 1) {
  double a[8], b = b0;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = a0 + i + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int o = 0; o < outer; o++)
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double &x = DEP ? a[0] : a[i];
      if (OP == 0) asm volatile("v_min_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
      if (OP == 1) asm volatile("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
      if (OP == 2) asm volatile("v_add_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
      if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(b), "v"(b));
      if (OP == 4) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b));
      if (OP == 5) asm volatile("v_rcp_f64 %0, %1" : "=v"(x) : "v"(x));
      if (OP == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
      if (OP == 7) { unsigned lo = (unsigned)__double_as_longlong(x); asm volatile("v_and_b32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(63)); x = __longlong_as_double(lo); }
      if (OP == 8) asm volatile("v_sqrt_f64 %0, %1" : "=v"(x) : "v"(x));
      if (OP == 9) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, bool DEP>
double run(int waves_per_block) {
  double *out; long long *cyc;
  hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
  int blocks = 256;
  hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.5, 0.999);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += v; m /= blocks;
  hipFree(out); hipFree(cyc);
  return m / (REP * 8.0);  // s_memtime ticks (100 MHz? shader clock?) per instruction
}

// wall-clock calibration: ns per instruction per wave, from HIP events around a long launch
template <int OP>
double run_ns(int waves_per_block, int blocks) {
  double *out; long long *cyc;
  hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
  const int outer = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<OP, false>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.5, 0.999, 10);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((k<OP, false>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.5, 0.999, outer);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipFree(out); hipFree(cyc);
  return ms * 1e6 / ((double)outer * REP * 8.0);
}

int main() {
  printf("wall clock, v_fma_f64 independent, ns per instr per wave (256 blocks): 4w %.3f  8w %.3f  12w %.3f  16w %.3f  32w(2 blocks/CU of 16) %.3f\n",
         run_ns<3>(4, 256), run_ns<3>(8, 256), run_ns<3>(12, 256), run_ns<3>(16, 256), run_ns<3>(16, 512));
  printf("wall clock, v_min_f64: 4w %.3f 8w %.3f 16w %.3f;  v_and_b32: 4w %.3f 8w %.3f 16w %.3f\n",
         run_ns<0>(4, 256), run_ns<0>(8, 256), run_ns<0>(16, 256), run_ns<7>(4, 256), run_ns<7>(8, 256), run_ns<7>(16, 256));
  const char *names[] = {"v_min_f64", "v_max_f64", "v_add_f64", "v_fma_f64", "v_mul_f64", "v_rcp_f64", "v_cmp_lt_f64", "v_and_b32", "v_sqrt_f64", "v_mov_b64"};
  printf("%-12s %10s %10s %10s %10s %10s %10s  (s_memtime ticks per instr per wave; N waves/block: 4 = 1 per SIMD ... 16 = 4 per SIMD)\n", "op", "indep/4w", "dep/4w", "indep/8w", "dep/8w", "indep/12w", "indep/16w");
#define ROW(OP) printf("%-12s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", names[OP], run<OP, false>(4), run<OP, true>(4), run<OP, false>(8), run<OP, true>(8), run<OP, false>(12), run<OP, false>(16));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9)
  return 0;
}
