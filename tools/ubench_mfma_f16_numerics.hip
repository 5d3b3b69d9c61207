// What v_mfma_f32_16x16x32_f16 does to its 32 products and the accumulator on gfx950: the facts the error
// bound of the split-half screen (k_simbits_screen_mfma_h2, fc_kabsch_math.h: kabsch_h2_bounds) rests on.
//   1. are fp16 subnormal INPUTS honoured or flushed?
//   2. is the sum C + sum_k a_k b_k rounded once, or after every addition (how many extra bits inside)?
//   3. how is the result rounded (nearest / truncation)?
//   4. worst relative error on random data against an exact fp64 sum, in units of u = 2^-24 of |C| + sum |a_k b_k|
//   5. issue rates: the MFMA stream alone, and with fp32 VALU work in a sibling wave of the same SIMD
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_f16_numerics.hip -o tools/ubench_mfma_f16_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// one 16x16x32 product: A [16][32], B [32][16] (row-major halfs), C/D [16][16] floats
__global__ void __launch_bounds__(64) k_one(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
  const int lane = threadIdx.x, kq = lane >> 4, l15 = lane & 15;
  h8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[l15 * 32 + kq * 8 + j];
    b[j] = B[(kq * 8 + j) * 16 + l15];
  }
  f4_t c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * kq + r) * 16 + l15];
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + l15] = c[r];
}

static _Float16 *dA, *dB;
static float *dC, *dD;
static void run_one(const std::vector<_Float16> &A, const std::vector<_Float16> &B, const std::vector<float> &C,
                    std::vector<float> &D) {
  hipMemcpy(dA, A.data(), 512 * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 512 * 2, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 256 * 4, hipMemcpyHostToDevice);
  k_one<<<1, 64>>>(dA, dB, dC, dD);
  D.resize(256);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
}
// element (0, 0) of the product for a chosen k-vector of products and accumulator
static float probe(const std::vector<double> &a, const std::vector<double> &b, double c) {
  std::vector<_Float16> A(512, (_Float16)0.f), B(512, (_Float16)0.f);
  std::vector<float> C(256, 0.f), D;
  for (int k = 0; k < 32 && k < (int)a.size(); ++k) {
    A[0 * 32 + k] = (_Float16)a[k];
    B[k * 16 + 0] = (_Float16)b[k];
  }
  C[0] = (float)c;
  run_one(A, B, C, D);
  return D[0];
}

// ---- rates: waves 0..3 of a workgroup stream MFMAs, waves 4..7 (same SIMDs) run v_fma_f32
template <int MODE>  // 0: MFMA alone, 1: VALU alone, 2: both
__global__ void __launch_bounds__(512) k_rate(float *out, float seed, int n_it) {
  const int wv = threadIdx.x >> 6;
  if (wv < 4) {
    if (MODE == 1) return;
    h8_t a, b;
    for (int j = 0; j < 8; ++j) a[j] = (_Float16)(seed + j), b[j] = (_Float16)(seed - j);
    f4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < n_it; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = r;
  } else {
    if (MODE == 0) return;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + i;
    const float mf = seed * 0.5f;
    for (int it = 0; it < n_it; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], mf, mf);
    float r = 0;
    for (int i = 0; i < 8; ++i) r += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
  }
}
template <int MODE>
static double rate(float *out, int n_it) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_rate<MODE><<<256, 512>>>(out, 1.0001f, n_it);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_rate<MODE><<<256, 512>>>(out, 1.0001f, n_it);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipMalloc(&dA, 512 * 2);
  hipMalloc(&dB, 512 * 2);
  hipMalloc(&dC, 256 * 4);
  hipMalloc(&dD, 256 * 4);
  const double two24 = 16777216.0;
  // 1. subnormal inputs: a = 2^-20 (fp16 subnormal), b = 2^10
  printf("subnormal input   : 2^-20 * 2^10 = %.10g (exact 0.0009765625; 0 = flushed)\n",
         probe({std::ldexp(1.0, -20)}, {1024.0}, 0.0));
  printf("subnormal input 2 : 2^-24 * 2^14 = %.10g (exact 0.0009765625)\n", probe({std::ldexp(1.0, -24)}, {16384.0}, 0.0));
  // 2. accumulation: C = 2^24, 32 products of 1.0: exact 2^24 + 32; rounding after every addition gives 2^24
  {
    std::vector<double> a(32, 1.0), b(32, 1.0);
    printf("C=2^24 + 32 x 1.0 : %.1f  (single rounding: %.1f, per-addition rounding: %.1f)\n", probe(a, b, two24), two24 + 32,
           two24);
    for (int e = 1; e <= 12; ++e) {
      std::vector<double> ae(32, std::ldexp(1.0, -e));
      printf("C=2^24 + 32 x 2^-%-2d: got +%.1f (exact +%.4f)\n", e, probe(ae, b, two24) - two24, 32.0 * std::ldexp(1.0, -e));
    }
    // products only (C = 0): one big, 31 small
    std::vector<double> a2(32, 1.0), b2(32, 1.0);
    a2[0] = 4096.0;
    b2[0] = 4096.0;
    printf("C=0, 2^24 + 31 x 1.0 : got +%.1f (exact +31)\n", probe(a2, b2, 0.0) - two24);
    for (int e = 1; e <= 6; ++e) {
      std::vector<double> a3(32, std::ldexp(1.0, -e));
      a3[0] = 4096.0;
      printf("C=0, 2^24 + 31 x 2^-%d: got +%.1f (exact +%.4f)\n", e, probe(a3, b2, 0.0) - two24, 31.0 * std::ldexp(1.0, -e));
    }
  }
  // 3. rounding of the result
  printf("C=2^24, + 1 : %+.1f   + 3 : %+.1f   + 5 : %+.1f  (nearest-even: 0 +4 +4; truncation: 0 +2 +4)\n",
         probe({1.0}, {1.0}, two24) - two24, probe({3.0}, {1.0}, two24) - two24, probe({5.0}, {1.0}, two24) - two24);
  printf("C=-2^24, - 3 : %+.1f  (nearest: -4, toward zero: -2)\n", probe({-3.0}, {1.0}, -two24) + two24);
  // 4. random data
  {
    srand(12345);
    double worst = 0.0;
    for (int trial = 0; trial < 400; ++trial) {
      std::vector<_Float16> A(512), B(512);
      std::vector<float> C(256), D;
      const int mode = trial % 4;  // 0: all the same sign and size, 1: mixed signs, 2: wide range of sizes, 3: large C
      for (int i = 0; i < 512; ++i) {
        double x = (double)rand() / RAND_MAX, y = (double)rand() / RAND_MAX;
        if (mode == 1 || mode == 3) x = 2 * x - 1, y = 2 * y - 1;
        if (mode == 2) x = std::ldexp(x, -(rand() % 12)), y = std::ldexp(y, -(rand() % 12));
        A[i] = (_Float16)(x * 100.0);
        B[i] = (_Float16)(y * 100.0);
      }
      for (int i = 0; i < 256; ++i) C[i] = mode == 3 ? (float)(((double)rand() / RAND_MAX - 0.5) * 1e6) : (float)(((double)rand() / RAND_MAX) * 1e3);
      run_one(A, B, C, D);
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double exact = C[i * 16 + j], mag = std::fabs((double)C[i * 16 + j]);
          for (int k = 0; k < 32; ++k) {
            const double p = (double)A[i * 32 + k] * (double)B[k * 16 + j];
            exact += p;
            mag += std::fabs(p);
          }
          const double err = std::fabs((double)D[i * 16 + j] - exact) / (mag * 5.9604644775390625e-08);
          if (err > worst) worst = err;
        }
    }
    printf("random data: worst |D - exact| = %.3f u * (|C| + sum |a b|)   (u = 2^-24; one rounding to nearest: <= 1)\n", worst);
  }
  // 5. rates
  {
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int n_it = 4096;
    const double m0 = rate<0>(out, n_it), m1 = rate<1>(out, n_it), m2 = rate<2>(out, n_it);
    const double n_instr = (double)n_it * 8;
    printf("MFMA alone %.3f ms (%.1f cycles per MFMA per SIMD at 2.4 GHz, %.0f TFLOP/s), v_fma_f32 sibling alone %.3f ms (%.1f cycles each), both %.3f ms\n",
           m0, m0 * 1e-3 * 2.4e9 / n_instr, 256.0 * 4 * n_instr * 16384.0 / (m0 * 1e-3) * 1e-12, m1, m1 * 1e-3 * 2.4e9 / n_instr, m2);
  }
  return 0;
}
