// Micro-benchmark: what a second wave on the SIMD costs an fp64-MFMA wave on gfx950.
// Workgroup = 8 waves (2 per SIMD).  Waves 0-3 issue v_mfma_f64_16x16x4_f64 back to back;
// waves 4-7 (the second wave of each SIMD) run one of: nothing, v_fma_f64, v_fma_f32, v_pk_fma_f32, integer VALU.
// Reported: time of the MFMA stream and the companion's throughput.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_coexec.hip -o tools/ubench_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
// -DFC_UB_F32: the MFMA stream is v_mfma_f32_16x16x4_f32 instead (binary tools/ubench_coexec_f32)
#ifdef FC_UB_F32
typedef float double4_t __attribute__((ext_vector_type(4)));
#define FC_UB_MFMA __builtin_amdgcn_mfma_f32_16x16x4f32
typedef float ub_t;
#else
typedef double double4_t __attribute__((ext_vector_type(4)));
#define FC_UB_MFMA __builtin_amdgcn_mfma_f64_16x16x4f64
typedef double ub_t;
#endif
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512) k_mix(double *out, int iters, int companion_iters) {
  const int wv = threadIdx.x >> 6;
  if (wv < 4) {  // waves 0-3: one per SIMD; waves 4-7 are their companions on the same SIMDs
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
    ub_t a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
      c0 = FC_UB_MFMA(a, b, c0, 0, 0, 0);
      c1 = FC_UB_MFMA(a, b, c1, 0, 0, 0);
      c2 = FC_UB_MFMA(a, b, c2, 0, 0, 0);
      c3 = FC_UB_MFMA(a, b, c3, 0, 0, 0);
      c4 = FC_UB_MFMA(a, b, c4, 0, 0, 0);
      c5 = FC_UB_MFMA(a, b, c5, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1];
  } else {
    if (MODE == 0) return;
    if (MODE == 1) {
      double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
      const double x = 1.0000001, y = 0.5;
      for (int i = 0; i < companion_iters; ++i) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
        a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
      }
      out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (MODE == 2) {
      float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
      const float x = 1.0000001f, y = 0.5f;
      for (int i = 0; i < companion_iters; ++i) {
        a0 = fmaf(a0, x, y); a1 = fmaf(a1, x, y); a2 = fmaf(a2, x, y); a3 = fmaf(a3, x, y);
        a4 = fmaf(a4, x, y); a5 = fmaf(a5, x, y); a6 = fmaf(a6, x, y); a7 = fmaf(a7, x, y);
      }
      out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (MODE == 3) {
      float2_t a0 = {(float)threadIdx.x, 1}, a1 = {1, 2}, a2 = {2, 3}, a3 = {3, 4}, a4 = {4, 5}, a5 = {5, 6}, a6 = {6, 7}, a7 = {7, 8};
      const float2_t x = {1.0000001f, 0.999999f}, y = {0.5f, 0.25f};
      for (int i = 0; i < companion_iters; ++i) {
        a0 = __builtin_elementwise_fma(a0, x, y); a1 = __builtin_elementwise_fma(a1, x, y);
        a2 = __builtin_elementwise_fma(a2, x, y); a3 = __builtin_elementwise_fma(a3, x, y);
        a4 = __builtin_elementwise_fma(a4, x, y); a5 = __builtin_elementwise_fma(a5, x, y);
        a6 = __builtin_elementwise_fma(a6, x, y); a7 = __builtin_elementwise_fma(a7, x, y);
      }
      float2_t s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
      out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1];
    } else {
      unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
      for (int i = 0; i < companion_iters; ++i) {
        a0 = a0 * 3u + 1u; a1 = a1 * 3u + 1u; a2 = a2 * 3u + 1u; a3 = a3 * 3u + 1u;
        a4 = a4 * 3u + 1u; a5 = a5 * 3u + 1u; a6 = a6 * 3u + 1u; a7 = a7 * 3u + 1u;
      }
      out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
  }
}

template <int MODE>
static float run(double *d, int iters, int comp) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(512), 0, 0, d, iters, comp);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  double *d;
  hipMalloc(&d, 256 * 512 * sizeof(double));
  const int iters = 20000;  // 120000 MFMAs per wave = 7.68 M cycles alone
  const char *names[] = {"alone", "v_fma_f64", "v_fma_f32", "v_pk_fma_f32", "v_mad_u32"};
  const float base = run<0>(d, iters, 0);
  printf("MFMA stream alone: %.3f ms (%.1f TFLOP/s on 1 wave/SIMD)\n", base, 256.0 * 4 * 6 * iters * 2048 / base / 1e9);
  // companion sized to roughly 1/4 of the MFMA stream's cycles when run alone
  const int comp = 60000;  // 8 instr x 60000 = 480k VALU instr ~ 1.9 M cycles at 4 cyc each
  float t;
  t = run<1>(d, iters, comp); printf("%-14s companion (%d x 8 instr): %.3f ms  (+%.1f %%)\n", names[1], comp, t, 100 * (t / base - 1));
  t = run<2>(d, iters, comp); printf("%-14s companion (%d x 8 instr): %.3f ms  (+%.1f %%)\n", names[2], comp, t, 100 * (t / base - 1));
  t = run<3>(d, iters, comp); printf("%-14s companion (%d x 8 instr): %.3f ms  (+%.1f %%)\n", names[3], comp, t, 100 * (t / base - 1));
  t = run<4>(d, iters, comp); printf("%-14s companion (%d x 8 instr): %.3f ms  (+%.1f %%)\n", names[4], comp, t, 100 * (t / base - 1));
  // companions alone (no MFMA): iters = 0
  t = run<1>(d, 0, comp); printf("v_fma_f64 alone: %.3f ms\n", t);
  t = run<2>(d, 0, comp); printf("v_fma_f32 alone: %.3f ms\n", t);
  t = run<3>(d, 0, comp); printf("v_pk_fma_f32 alone: %.3f ms\n", t);
  t = run<4>(d, 0, comp); printf("v_mad_u32 alone: %.3f ms\n", t);
  return 0;
}
