// Issue rate of VALU flavours on gfx950: independent chains per lane, 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_rates.hip -o tools/ubench_valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float seed) {
  // 8 independent accumulators so that latency never limits issue
  f2 a[8];
  double d[8];
  float f[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = f2{seed + i, seed - i};
    d[i] = seed + i;
    f[i] = seed + i;
  }
  const f2 m = f2{seed * 0.5f, seed * 0.25f};
  const double md = seed * 0.5;
  const float mf = seed * 0.5f;
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], m, m);  // v_pk_fma_f32
      if (MODE == 1) f[i] = __builtin_fmaf(f[i], mf, mf);            // v_fma_f32
      if (MODE == 2) d[i] = __builtin_fma(d[i], md, md);             // v_fma_f64
      if (MODE == 3) a[i] = a[i] * m;                                // v_pk_mul_f32
      if (MODE == 4) a[i] = a[i] + m;                                // v_pk_add_f32
      if (MODE == 5) d[i] = d[i] + md;                               // v_add_f64
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + (float)d[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE>
static void run(const char *name, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  k<MODE><<<blocks, 256>>>(out, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // wave instructions per SIMD = 8 waves x N_IT x 8; cycles at an assumed 2.4 GHz
  const double instr_per_simd = 8.0 * N_IT * 8;
  printf("%-14s %8.3f ms  -> %.2f cycles per wave instruction per SIMD (at 2.4 GHz)\n", name, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<1>("v_fma_f32", out);
  run<0>("v_pk_fma_f32", out);
  run<3>("v_pk_mul_f32", out);
  run<4>("v_pk_add_f32", out);
  run<2>("v_fma_f64", out);
  run<5>("v_add_f64", out);
  return 0;
}
