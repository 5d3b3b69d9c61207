// What one SIMD of gfx950 issues per cycle when f16 MFMAs (v_mfma_f32_16x16x32_f16) and fp32 vector instructions
// (v_fma_f32) come from one wave or from sibling waves -- in CORE CYCLES (s_memtime), so the clock the chip holds
// under load does not enter.  Every wave runs its role for a fixed window and counts the units it finished.
//   roles: M = MFMA stream (8 independent accumulators), V = v_fma_f32 stream (8 independent chains),
//          Ik = one wave interleaving k v_fma_f32 behind every MFMA, P = phased (54 MFMAs, then 268 v_fma_f32:
//          the instruction mix of one sub-tile of k_simbits_screen_mfma_h2<2>)
// One workgroup per CU; wave w sits on SIMD w & 3, so "slot" w >> 2 names the waves that share a SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue_model.hip -o tools/ubench_issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
#ifdef UB_PK
typedef float vf_t __attribute__((ext_vector_type(2)));
#else
typedef float vf_t;
#endif

#define MFMA(i) "v_mfma_f32_16x16x32_f16 %" #i ", %16, %17, %" #i "\n"
#define MFMA32(i) "v_mfma_f32_32x32x16_f16 %" #i ", %4, %5, %" #i "\n"
#if defined(UB_PK)  // two fp32 FMAs per lane and instruction
#define VF(i) "v_pk_fma_f32 %" #i ", %" #i ", %18, %18\n"
#elif defined(UB_VOP2)  // the 4-byte encoding most of the screen's polynomial uses
#define VF(i) "v_fmac_f32_e32 %" #i ", %18, %18\n"
#else
#define VF(i) "v_fma_f32 %" #i ", %" #i ", %18, %18\n"
#endif
#define OPERANDS                                                                                                   \
  : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), \
    "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])                  \
  : "v"(a), "v"(b), "v"(mf)

enum Role { R_NONE = 0, R_M, R_V, R_I1, R_I2, R_I3, R_I4, R_I5, R_P, R_P32, R_M32 };

struct State {
  f4_t acc[8];
  float f[8];
  h8_t a, b;
  float mf;
};

// one unit of each role; returns (MFMAs, VALUs) through the table below
__device__ __forceinline__ void unit_M(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) MFMA(6) MFMA(7) OPERANDS);
}
__device__ __forceinline__ void unit_V(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(VF(8) VF(9) VF(10) VF(11) VF(12) VF(13) VF(14) VF(15) OPERANDS);
}
__device__ __forceinline__ void unit_I1(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) VF(8) MFMA(1) VF(9) MFMA(2) VF(10) MFMA(3) VF(11) MFMA(4) VF(12) MFMA(5) VF(13) MFMA(6) VF(14)
                   MFMA(7) VF(15) OPERANDS);
}
__device__ __forceinline__ void unit_I2(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) VF(8) VF(9) MFMA(1) VF(10) VF(11) MFMA(2) VF(12) VF(13) MFMA(3) VF(14) VF(15) MFMA(4) VF(8) VF(9)
                   MFMA(5) VF(10) VF(11) MFMA(6) VF(12) VF(13) MFMA(7) VF(14) VF(15) OPERANDS);
}
__device__ __forceinline__ void unit_I3(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) VF(8) VF(9) VF(10) MFMA(1) VF(11) VF(12) VF(13) MFMA(2) VF(14) VF(15) VF(8) MFMA(3) VF(9) VF(10)
                   VF(11) MFMA(4) VF(12) VF(13) VF(14) MFMA(5) VF(15) VF(8) VF(9) MFMA(6) VF(10) VF(11) VF(12) MFMA(7)
                       VF(13) VF(14) VF(15) OPERANDS);
}
__device__ __forceinline__ void unit_I4(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) VF(8) VF(9) VF(10) VF(11) MFMA(1) VF(12) VF(13) VF(14) VF(15) MFMA(2) VF(8) VF(9) VF(10) VF(11)
                   MFMA(3) VF(12) VF(13) VF(14) VF(15) MFMA(4) VF(8) VF(9) VF(10) VF(11) MFMA(5) VF(12) VF(13) VF(14)
                       VF(15) MFMA(6) VF(8) VF(9) VF(10) VF(11) MFMA(7) VF(12) VF(13) VF(14) VF(15) OPERANDS);
}
__device__ __forceinline__ void unit_I5(f4_t (&acc)[8], vf_t (&f)[8], h8_t a, h8_t b, vf_t mf) {
  asm volatile(MFMA(0) VF(8) VF(9) VF(10) VF(11) VF(12) MFMA(1) VF(13) VF(14) VF(15) VF(8) VF(9) MFMA(2) VF(10) VF(11)
                   VF(12) VF(13) VF(14) MFMA(3) VF(15) VF(8) VF(9) VF(10) VF(11) MFMA(4) VF(12) VF(13) VF(14) VF(15) VF(8)
                       MFMA(5) VF(9) VF(10) VF(11) VF(12) VF(13) MFMA(6) VF(14) VF(15) VF(8) VF(9) VF(10) MFMA(7) VF(11)
                           VF(12) VF(13) VF(14) VF(15) OPERANDS);
}

typedef float f16_t __attribute__((ext_vector_type(16)));
// four independent 32 x 32 accumulators (64 registers), the 32x32x16 form of the same product
__device__ __forceinline__ void unit_M32(f16_t (&acc32)[4], h8_t a, h8_t b) {
  asm volatile(MFMA32(0) MFMA32(1) MFMA32(2) MFMA32(3) : "+v"(acc32[0]), "+v"(acc32[1]), "+v"(acc32[2]), "+v"(acc32[3]) : "v"(a), "v"(b));
}

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

struct Config {
  int role[8];  // per slot (waves sharing a SIMD)
  int n_slots;
};

__global__ void __launch_bounds__(1024) k_issue(Config cfg, unsigned long long window, unsigned long long *units_out, float *sink,
                                               float seed) {
  const int wv = threadIdx.x >> 6;
  const int slot = wv >> 2;
  const int role = cfg.role[slot];
  f4_t acc[8];
  vf_t f[8];
  f16_t acc32[4];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 16; ++k) acc32[i][k] = 0.f;
  h8_t a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(seed + j), b[j] = (_Float16)(seed - j);
  for (int i = 0; i < 8; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f}, f[i] = vf_t(seed + i);
  const vf_t mf = vf_t(seed * 0.5f);
  __syncthreads();
  const unsigned long long t0 = now();
  unsigned long long units = 0;
  unsigned long long t = t0;
  while (t - t0 < window) {
    // 8 units between two looks at the clock
    if (role == R_M) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_M(acc, f, a, b, mf);
    } else if (role == R_V) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_V(acc, f, a, b, mf);
    } else if (role == R_I1) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_I1(acc, f, a, b, mf);
    } else if (role == R_I2) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_I2(acc, f, a, b, mf);
    } else if (role == R_I3) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_I3(acc, f, a, b, mf);
    } else if (role == R_I4) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_I4(acc, f, a, b, mf);
    } else if (role == R_I5) {
for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_I5(acc, f, a, b, mf);
    } else if (role == R_M32) {
      for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int u = 0; u < 16; ++u) unit_M32(acc32, a, b);  // 64 MFMAs per rep; "unit" bookkeeping: 8 MFMAs
    } else if (role == R_P32) {  // one 32 x 32 tile = 1024 pairs: 108 MFMAs of 32x32x16, then 4 x 268 v_fma_f32
      for (int rep = 0; rep < 128; ++rep) {
#pragma unroll
        for (int u = 0; u < 27; ++u) unit_M32(acc32, a, b);
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int u = 0; u < 33; ++u) unit_V(acc, f, a, b, mf);
          asm volatile(VF(8) VF(9) VF(10) VF(11) OPERANDS);
        }
      }
    } else if (role == R_P) {  // one "sub-tile": 54 MFMAs, then 268 v_fma_f32 (counted as ONE unit of 8 below: x8 keeps the bookkeeping uniform)
      for (int rep = 0; rep < 128; ++rep) {
#pragma unroll
        for (int u = 0; u < 6; ++u) unit_M(acc, f, a, b, mf);
        asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) MFMA(4) MFMA(5) OPERANDS);
#pragma unroll
        for (int u = 0; u < 33; ++u) unit_V(acc, f, a, b, mf);
        asm volatile(VF(8) VF(9) VF(10) VF(11) OPERANDS);
      }
    }
    units += 128;
    t = now();
  }
  if ((threadIdx.x & 63) == 0) {
    units_out[(size_t)blockIdx.x * 64 + wv * 2] = units;
    units_out[(size_t)blockIdx.x * 64 + wv * 2 + 1] = t - t0;
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + ((const float *)&f[i])[0];
  for (int i = 0; i < 4; ++i) r += acc32[i][0] + acc32[i][15];
  if (r == 12345.678f) sink[0] = r;
}

static const char *role_name(int r) {
  static const char *n[] = {"-", "M", "V", "I1", "I2", "I3", "I4", "I5", "P", "P32", "M32"};
  return n[r];
}
static void per_unit(int r, int &mfma, int &valu) {
  switch (r) {
    case R_M: mfma = 8, valu = 0; break;
    case R_V: mfma = 0, valu = 8; break;
    case R_I1: mfma = 8, valu = 8; break;
    case R_I2: mfma = 8, valu = 16; break;
    case R_I3: mfma = 8, valu = 24; break;
    case R_I4: mfma = 8, valu = 32; break;
    case R_I5: mfma = 8, valu = 40; break;
    case R_P: mfma = 54, valu = 268; break;
    case R_P32: mfma = 108, valu = 4 * 268; break;  // per 1024 pairs (R_P: per 256)
    case R_M32: mfma = 8, valu = 0; break;
    default: mfma = valu = 0;
  }
}

int main() {
  const int n_cu = 256;
  unsigned long long *d_units;
  float *d_sink;
  hipMalloc(&d_units, (size_t)n_cu * 64 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 64);
  std::vector<unsigned long long> h((size_t)n_cu * 64);
  const unsigned long long window = 6000000ull;
  const std::vector<std::vector<int>> configs = {
      {R_M},           {R_V},           {R_V, R_V},       {R_V, R_V, R_V},       {R_V, R_V, R_V, R_V},
      {R_M, R_M},      {R_M, R_V},      {R_M, R_V, R_V},  {R_M, R_V, R_V, R_V},  {R_M, R_M, R_V, R_V},
      {R_I1},          {R_I2},          {R_I3},           {R_I4},                {R_I5},
      {R_I2, R_I2},    {R_I4, R_I4},    {R_I5, R_I5},     {R_I5, R_I5, R_I5},    {R_I2, R_V},
      {R_I2, R_V, R_V}, {R_P},          {R_P, R_P},       {R_P, R_P, R_P},       {R_P, R_P, R_P, R_P},
      {R_I5, R_I5, R_I5, R_I5}, {R_M32}, {R_M32, R_M32}, {R_M32, R_V}, {R_P32}, {R_P32, R_P32}, {R_P32, R_P32, R_P32}};
  printf("config | per SIMD: cycles per MFMA (all waves), cycles per v_fma_f32 (all waves) | matrix pipe busy (16/MFMA), issue sum (8/MFMA + 4/VALU; 2/VALU)\n");
  for (const auto &c : configs) {
    Config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_slots = (int)c.size();
    std::string name;
    for (size_t s = 0; s < c.size(); ++s) cfg.role[s] = c[s], name += std::string(s ? "+" : "") + role_name(c[s]);
    hipMemset(d_units, 0, (size_t)n_cu * 64 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_issue, dim3(n_cu), dim3(256 * cfg.n_slots), 0, 0, cfg, window, d_units, d_sink, 1.0001f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_units, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // per SIMD (wave & 3) sums over its slots, averaged over CUs and SIMDs
    double mfma_rate = 0, valu_rate = 0;  // instructions per cycle per SIMD
    for (int cu = 0; cu < n_cu; ++cu)
      for (int simd = 0; simd < 4; ++simd)
        for (int s = 0; s < cfg.n_slots; ++s) {
          const int wv = s * 4 + simd;
          const double units = (double)h[(size_t)cu * 64 + wv * 2], cyc = (double)h[(size_t)cu * 64 + wv * 2 + 1];
          int m, v;
          per_unit(cfg.role[s], m, v);
          if (cfg.role[s] == R_P) {  // 8 sub-tiles per bookkeeping step of "8 units"
            mfma_rate += units * m / cyc;
            valu_rate += units * v / cyc;
          } else {
            mfma_rate += units * m / cyc;
            valu_rate += units * v / cyc;
          }
        }
    mfma_rate /= n_cu * 4.0;
    valu_rate /= n_cu * 4.0;
    printf("%-16s | %7.2f cyc/MFMA  %7.2f cyc/VALU | pipe busy %.2f, issue sum@4 %.2f, @2 %.2f\n", name.c_str(),
           mfma_rate > 0 ? 1.0 / mfma_rate : 0.0, valu_rate > 0 ? 1.0 / valu_rate : 0.0, 16.0 * mfma_rate,
           8.0 * mfma_rate + 4.0 * valu_rate, 8.0 * mfma_rate + 2.0 * valu_rate);
  }
  return 0;
}
