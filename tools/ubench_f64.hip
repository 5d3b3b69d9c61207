// Micro-benchmark: fp64 issue rates on gfx950 -- v_fma_f64 (VALU) vs
// v_mfma_f64_16x16x4_f64 (matrix pipe), one to eight waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_fma(double *out, int iters) {
  double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  const double x = 1.0000001, y = 0.5;
  for (int i = 0; i < iters; ++i) {
    a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
    a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_mfma(double *out, int iters) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  double *d;
  hipMalloc(&d, 256 * 8 * 1024 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps;  // 256 threads = 4 waves = 1 per SIMD per block
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d, iters);
        else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double waves = blocks * 4.0;
      const double flops = which == 0 ? waves * 64 * 8.0 * 2 * iters : waves * 4.0 * 2048 * iters;
      printf("%s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", which == 0 ? "v_fma_f64 " : "mfma_f64_16x16x4", wps, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
