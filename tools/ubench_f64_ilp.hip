// Micro-benchmark: how far apart must two DEPENDENT v_fma_f64 be on gfx950?  ILP independent chains per wave, issued round-robin
// (inline asm keeps the order), at 1 / 2 / 4 waves per SIMD.  The headline kernel runs two waves per SIMD and its atom pass has
// 12 chains of 3 + 4 chains of 3 per atom: what order costs nothing?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f64_ilp.hip -o tools/ubench_f64_ilp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ void k_chain(double *out, int iters) {
  double a[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x + i;
  const double x = 1.0000001, y = 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 48 / ILP; ++rep) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
static void run(double *d, int wps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  const int blocks = 256 * wps;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 4.0 * iters * (48 / ILP) * ILP;  // wave instructions
  printf("ILP=%2d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  %.2f cycles@2.4GHz per wave instruction and SIMD\n", ILP, wps, ms,
         n * 128 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024.0));
}

int main() {
  double *d;
  hipMalloc(&d, 256 * 8 * 1024 * sizeof(double));
  for (int wps = 1; wps <= 4; wps *= 2) {
    run<1>(d, wps);
    run<2>(d, wps);
    run<3>(d, wps);
    run<4>(d, wps);
    run<6>(d, wps);
    run<8>(d, wps);
    run<12>(d, wps);
    run<16>(d, wps);
  }
  return 0;
}
