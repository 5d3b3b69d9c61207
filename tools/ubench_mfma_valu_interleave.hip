// Does a wave's vector work hide under ITS OWN f16 MFMAs when the two are interleaved (1 MFMA : NV vector
// instructions, sched_group_barrier), as against one phase after the other?  The instruction mix of one 16 x 16
// sub-tile of k_simbits_screen_mfma_h2 (54 MFMAs, four two-test polynomials), per SIMD at 2 / 3 / 4 waves.
// Measured on MI355X: phased 741 / 663 / 629 ns per sub-tile and SIMD, interleaved 1:4 801 / 711 / 667, 1:2 754 / 670 / 629
// -> no: the orders cost the same or the interleaved one more; resident waves are what helps.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize [-DPATTERN -DNV=4] tools/ubench_mfma_valu_interleave.hip -o tools/ubench_mfma_valu_interleave
#include <hip/hip_runtime.h>
#include "/root/repo/firecode_amd/csrc/fc_kabsch_math.h"
using namespace fc;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool poly(const float (&B)[9], float s, float h, float p0, float p2) {
#pragma clang fp contract(fast)
  const float L = s - h;
  const float n2 = B[0]*B[0]+B[1]*B[1]+B[2]*B[2]+B[3]*B[3]+B[4]*B[4]+B[5]*B[5]+B[6]*B[6]+B[7]*B[7]+B[8]*B[8];
  const float uu = L*L - n2;
  const float c00 = B[4]*B[8]-B[5]*B[7], c01 = B[5]*B[6]-B[3]*B[8], c02 = B[3]*B[7]-B[4]*B[6];
  const float c10 = B[2]*B[7]-B[1]*B[8], c11 = B[0]*B[8]-B[2]*B[6], c12 = B[1]*B[6]-B[0]*B[7];
  const float c20 = B[1]*B[5]-B[2]*B[4], c21 = B[2]*B[3]-B[0]*B[5], c22 = B[0]*B[4]-B[1]*B[3];
  const float det = B[0]*c00+B[1]*c01+B[2]*c02;
  const float e2 = c00*c00+c01*c01+c02*c02+c10*c10+c11*c11+c12*c12+c20*c20+c21*c21+c22*c22;
  const float P0 = uu*uu - 4.0f*(e2 + 2.0f*L*det);
  const float s2 = s*s;
  return !(uu > p2*s2) | !(P0 > p0*(s2*s2));
}
__global__ void __launch_bounds__(256, 2) k(const h8_t* X, const float* G, float h, float p0, float p2, unsigned long long* out, int n) {
  __shared__ h8_t lds[12*64+64];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = X[threadIdx.x]; lds[threadIdx.x+256] = X[threadIdx.x+256]; lds[threadIdx.x+512] = X[threadIdx.x+512];
  __syncthreads();
  h8_t ra[2][2][3];
  for (int s=0;s<2;++s) for (int p=0;p<2;++p) for (int c=0;c<3;++c) ra[s][p][c] = X[1024 + ((s*2+p)*3+c)*64 + lane];
  f4_t accA[9], accB[9];
  for (int e=0;e<9;++e) accA[e] = f4_t{0,0,0,0};
  unsigned long long res = 0;
  for (int it = 0; it < n; ++it) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e=0;e<9;++e) accB[e] = f4_t{0,0,0,0};
#pragma unroll
    for (int s=0;s<2;++s)
#pragma unroll
      for (int y=0;y<3;++y) {
        const h8_t bh = lds[((s*2+0)*3+y)*64 + lane], bl = lds[((s*2+1)*3+y)*64+lane];
#pragma unroll
        for (int x=0;x<3;++x) {
          accB[x*3+y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][0][x], bl, accB[x*3+y],0,0,0);
          accB[x*3+y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][1][x], bh, accB[x*3+y],0,0,0);
        }
      }
#pragma unroll
    for (int s=0;s<2;++s)
#pragma unroll
      for (int y=0;y<3;++y) {
        const h8_t bh = lds[((s*2+0)*3+y)*64 + lane];
#pragma unroll
        for (int x=0;x<3;++x) accB[x*3+y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][0][x], bh, accB[x*3+y],0,0,0);
      }
    bool may[4], redo[4]; const KabschF32Bounds bd{p0, p2, p2}; const float tf = h*4;
    const float* ldsG = (const float*)(lds + 768); const float Gq = ldsG[lane & 15];
#pragma unroll
    for (int r=0;r<4;++r) {
      float B9[9];
#pragma unroll
      for (int e=0;e<9;++e) B9[e] = accA[e][r];
      may[r] = kabsch_may_be_below_f32_2t(B9, ldsG[64 + 4*(lane>>4) + r + it*16] + Gq, h, bd, tf, redo[r]);
    }
#ifdef PATTERN
#pragma unroll
    for (int m=0;m<54;++m) { __builtin_amdgcn_sched_group_barrier(0x008,1,0); __builtin_amdgcn_sched_group_barrier(0x002,NV,0); }
#endif
    __builtin_amdgcn_sched_barrier(0);
    for (int r=0;r<4;++r) res += __builtin_amdgcn_ballot_w64(may[r]) + __builtin_amdgcn_ballot_w64(redo[r]);
#pragma unroll
    for (int e=0;e<9;++e) accA[e] = accB[e];
  }
  if (lane==0) out[blockIdx.x] = res;
}
#include <cstdio>
int main() {
  h8_t* X; float* G; unsigned long long* out;
  hipMalloc(&X, 4096*16); hipMalloc(&G, 1<<20); hipMalloc(&out, 8*4096);
  hipMemset(X, 0x3c, 4096*16); hipMemset(G, 0x40, 1<<20);
  const int n = 2000;
  for (int blocks : {512, 768, 1024}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(X, G, 1.0f, 1e-4f, 1e-5f, out, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(X, G, 1.0f, 1e-4f, 1e-5f, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks/256 waves, each n iterations
    printf("blocks %d: %.3f ms -> %.1f ns per sub-tile per wave, %.1f ns per sub-tile per SIMD\n", blocks, ms, ms*1e6/n, ms*1e6/n/(blocks/256.0));
  }
  return 0;
}
