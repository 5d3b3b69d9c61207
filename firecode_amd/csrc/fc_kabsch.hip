// fc_kabsch.hip -- batched Kabsch-RMSD kernels for gfx950 (wave64, float64).
//
// Replaces the per-pair NumPy calls of prism_pruner.rmsd.rmsd_and_max /
// get_alignment_matrix reached from firecode/utils.py:499, embedder.py:1472,
// ensemble.py:230.
//
// Data layout in HBM (built once per ensemble by k_prep):
//   Xs[(a*3+c)*Npad + n]   coordinate c of selected atom a of conformer n
//   G[n]                   sum_a |x_a|^2
// so the 64 lanes of a wavefront that own 64 consecutive conformers read one
// coordinate with one coalesced 512-byte access, and the 8 consecutive
// conformers a wavefront treats as "rows" are 64 contiguous bytes that the
// scalar unit fetches with s_load (wave-uniform address).
#include <type_traits>

#include "fc_common.h"
#include <vector>
#include "fc_kabsch_math.h"

#include <algorithm>
#include <cstdlib>

namespace fc {

// LDS budget: a column tile is A*3*64*8 bytes; 160 KiB per CU on gfx950
static constexpr size_t kLdsLimit = 160 * 1024;

// ---------------------------------------------------------------------------
// k_prep: AoS (N, A_all, 3) -> conformer-minor SoA of the selected atoms,
// optional centring on the centroid of the selection, G[n].
// One lane per conformer: reads walk the conformer's own contiguous block
// (absorbed by L2), writes are coalesced across lanes.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_prep(const double *__restrict__ coords, int64_t N, int64_t A_all, const int32_t *__restrict__ sel,
       int64_t A, int center, int64_t Npad, double *__restrict__ Xs, double *__restrict__ G,
       double *__restrict__ Xa, unsigned long long *__restrict__ gmax_bits,
       const int32_t *__restrict__ conf_idx) {  // conf_idx != nullptr: conformer n is coords[conf_idx[n]]
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npad) return;
  // atom rows A .. A4-1 (A4 = A rounded up to 4) are zero: the MFMA K loop
  // runs over whole groups of 4 atoms without a bounds test
  for (int64_t a = A; a < ((A + 3) & ~(int64_t)3); ++a) {
    Xs[(a * 3 + 0) * Npad + n] = 0.0;
    Xs[(a * 3 + 1) * Npad + n] = 0.0;
    Xs[(a * 3 + 2) * Npad + n] = 0.0;
  }
  if (n >= N) {  // zero padding keeps every later load in bounds and finite
    for (int64_t a = 0; a < A; ++a) {
      Xs[(a * 3 + 0) * Npad + n] = 0.0;
      Xs[(a * 3 + 1) * Npad + n] = 0.0;
      Xs[(a * 3 + 2) * Npad + n] = 0.0;
    }
    G[n] = 0.0;
    return;
  }
  const double *src = coords + (conf_idx ? (int64_t)conf_idx[n] : n) * A_all * 3;
  double cx = 0.0, cy = 0.0, cz = 0.0;
  if (center) {
    // same order as numpy's mean(axis=0): running sum over atoms, then / A
    for (int64_t a = 0; a < A; ++a) {
      const double *r = src + (int64_t)sel[a] * 3;
      cx += r[0];
      cy += r[1];
      cz += r[2];
    }
    cx /= (double)A;
    cy /= (double)A;
    cz /= (double)A;
  }
  double g = 0.0;
  for (int64_t a = 0; a < A; ++a) {
    const double *r = src + (int64_t)sel[a] * 3;
    const double x = r[0] - cx, y = r[1] - cy, z = r[2] - cz;
    Xs[(a * 3 + 0) * Npad + n] = x;
    Xs[(a * 3 + 1) * Npad + n] = y;
    Xs[(a * 3 + 2) * Npad + n] = z;
    Xa[(n * A + a) * 3 + 0] = x;
    Xa[(n * A + a) * 3 + 1] = y;
    Xa[(n * A + a) * 3 + 2] = z;
    g += x * x + y * y + z * z;
  }
  G[n] = g;
  // largest G (sizes the single-precision screen's band): non-negative doubles order like their bits
  if (g == g) atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(g));
}

// k_prep_tile: the same preparation for TILE conformers per workgroup through LDS -- the input
// block of TILE conformers (TILE x A_all x 3 doubles, contiguous) is read with coalesced loads and
// every output is written coalesced too: Xs across the TILE conformers of a column, Xa as the
// contiguous image of the tile.  (k_prep reads with a 1200-byte stride between lanes and keeps
// 40 CUs busy at 10^4 conformers: 75 us against ~15 us.)  Same arithmetic, same order.
// Round 5 (the trace of one prune_by_rmsd(host arrays) call showed this kernel at 40 us for 10 000 x 50: 157 workgroups of
// 64 conformers on 256 CUs, each a chain of ~110 loop turns with 64-bit divisions and one serial pass of 64 lanes): 32
// conformers per workgroup (twice the workgroups, half the turns), 32-bit index arithmetic, 16-byte loads where the source
// allows, the Xs phase without a division (a wavefront per coordinate row), the centroid as three lanes per conformer -- one
// per coordinate, each the SAME running sum over the atoms in the same order.
constexpr int kPrepTile = 32;
template <int TILE>
__global__ void __launch_bounds__(256)
k_prep_tile(const double *__restrict__ coords, int64_t N, int64_t A_all, const int32_t *__restrict__ sel,
            int64_t A, int center, int64_t Npad, double *__restrict__ Xs, double *__restrict__ G,
            double *__restrict__ Xa, unsigned long long *__restrict__ gmax_bits,
            const int32_t *__restrict__ conf_idx,  // conf_idx != nullptr: gather (the survivors of an earlier stage)
            int64_t tile0) {                       // first tile of this launch (an upload arrives in pieces)
  static_assert(TILE == 16 || TILE == 32 || TILE == 64, "a tile is a wavefront of conformers, a half or a quarter");
  extern __shared__ double tile[];  // [TILE][A_all*3 + 1] (the +1 spreads the conformers over the banks), then the selection
  __shared__ double cen[TILE][3];
  const int tid = threadIdx.x;
  const int64_t n0 = ((int64_t)blockIdx.x + tile0) * TILE;
  const uint32_t row = (uint32_t)(A_all * 3), ld = row + 1u;
  // the atom selection in LDS for the serial loops below.  Phases of one workgroup at 10^4 x 50 by wall-clock stamps: load
  // 4.6 us, centroid 1.0, G 2.6, Xs 3.2, Xa 2.2 = 13.5 us of the kernel's 23 (17 us for the ten workgroups of a 300-conformer
  // ensemble: the chain of one workgroup is what the kernel costs at any size)
  int *__restrict__ s_sel = reinterpret_cast<int *>(tile + (size_t)TILE * ld);
  for (uint32_t a = (uint32_t)tid; a < (uint32_t)A; a += 256u) s_sel[a] = sel[a];
  const int64_t left = N - n0;
  const uint32_t n_here = left < TILE ? (left > 0 ? (uint32_t)left : 0u) : (uint32_t)TILE;
  const uint32_t cnt = n_here * row;
  if (conf_idx == nullptr) {
    const double *__restrict__ src = coords + n0 * (int64_t)row;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      typedef double dbl2_t __attribute__((ext_vector_type(2)));
      const dbl2_t *__restrict__ src2 = reinterpret_cast<const dbl2_t *>(src);
      const uint32_t n2 = cnt >> 1;
      for (uint32_t k = (uint32_t)tid; k < n2; k += 256u) {
        const dbl2_t v = src2[k];
        const uint32_t k0 = 2u * k, l0 = k0 / row, l1 = (k0 + 1u) / row;
        tile[l0 * ld + (k0 - l0 * row)] = v[0];
        tile[l1 * ld + (k0 + 1u - l1 * row)] = v[1];
      }
      if ((cnt & 1u) && tid == 0) tile[((cnt - 1u) / row) * ld + ((cnt - 1u) % row)] = src[cnt - 1u];
    } else {
      for (uint32_t k = (uint32_t)tid; k < cnt; k += 256u) tile[(k / row) * ld + (k % row)] = src[k];
    }
  } else {
    for (uint32_t k = (uint32_t)tid; k < cnt; k += 256u) {
      const uint32_t l = k / row, r = k - l * row;
      tile[l * ld + r] = coords[(int64_t)conf_idx[n0 + l] * (int64_t)row + r];
    }
  }
  __syncthreads();
  // centroid: lane (conformer l, coordinate c) -- same order as numpy's mean(axis=0): running sum over atoms, then / A
  if (tid < TILE * 3) {
    const uint32_t l = (uint32_t)tid / 3u, c = (uint32_t)tid - 3u * l;
    double sum = 0.0;
    if (center && l < n_here) {
      const double *__restrict__ t = tile + l * ld + c;
      for (int64_t a = 0; a < A; ++a) sum += t[(uint32_t)s_sel[a] * 3u];
      sum /= (double)A;
    }
    cen[l][c] = sum;
  }
  __syncthreads();
  if (tid < 64) {  // (a whole wavefront: the reduction below shuffles across 64 lanes)
    double g = 0.0;
    if (tid < TILE && (uint32_t)tid < n_here) {
      const double *__restrict__ t = tile + (uint32_t)tid * ld;
      const double cx = cen[tid][0], cy = cen[tid][1], cz = cen[tid][2];
      for (int64_t a = 0; a < A; ++a) {
        const double *__restrict__ r = t + (uint32_t)s_sel[a] * 3u;
        const double x = r[0] - cx, y = r[1] - cy, z = r[2] - cz;
        g += x * x + y * y + z * z;
      }
    }
    if (tid < TILE && n0 + tid < Npad) G[n0 + tid] = g;
    // largest G (sizes the single-precision screen's band): one atomic per workgroup
    double gm = (g == g) ? g : 0.0;
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_xor(gm, off);
      if (o > gm) gm = o;
    }
    if (tid == 0) atomicMax(gmax_bits, (unsigned long long)__double_as_longlong(gm));
  }
  // Xs: element (a, c) of TILE consecutive conformers is one coalesced store; rows A..A4-1 are zero.  256 / TILE
  // coordinate rows per turn, no division: (a, c) advance with the row
  {
    constexpr uint32_t kRowsPerTurn = 256u / TILE;
    const uint32_t l = (uint32_t)tid % TILE, r0 = (uint32_t)tid / TILE;
    const uint32_t A4x3 = (uint32_t)(((A + 3) & ~(int64_t)3) * 3), Ax3 = (uint32_t)(A * 3);
    uint32_t a = r0 / 3u, c = r0 - 3u * a;
    const bool lane_in = l < n_here, col_in = n0 + l < Npad;
    const double *__restrict__ tl = tile + l * ld;
    const double c0 = cen[l][0], c1 = cen[l][1], c2 = cen[l][2];
    for (uint32_t ac = r0; ac < A4x3; ac += kRowsPerTurn) {
      double v = 0.0;
      if (ac < Ax3 && lane_in) v = tl[(uint32_t)s_sel[a] * 3u + c] - (c == 0u ? c0 : (c == 1u ? c1 : c2));
      if (col_in) Xs[(int64_t)ac * Npad + n0 + l] = v;
      c += kRowsPerTurn % 3u;
      a += kRowsPerTurn / 3u;
      if (c >= 3u) c -= 3u, ++a;
    }
  }
  // Xa: [n][a][c], contiguous for the tile
  {
    const uint32_t Ax3 = (uint32_t)(A * 3), tot = n_here * Ax3;
    double *__restrict__ xa = Xa + n0 * (int64_t)Ax3;
    for (uint32_t k = (uint32_t)tid; k < tot; k += 256u) {
      const uint32_t l = k / Ax3, rest = k - l * Ax3;
      const uint32_t a = rest / 3u, c = rest - a * 3u;
      xa[k] = tile[l * ld + (uint32_t)s_sel[a] * 3u + c] - cen[l][c];
    }
  }
}

// ---------------------------------------------------------------------------
// Exact pair evaluation on the SoA layout: covariance, optimal rotation,
// explicit rotated difference -> (rmsd, maxdev).  One lane per pair.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pair_exact(const double *__restrict__ Xs, int64_t Npad, int A,
                                           int64_t i, int64_t j, double &rmsd, double &maxdev,
                                           double *R_out = nullptr) {
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a < A; ++a) {
    const double *pa = Xs + (int64_t)(a * 3) * Npad;
    const double px = pa[i], py = pa[Npad + i], pz = pa[2 * Npad + i];
    const double qx = pa[j], qy = pa[Npad + j], qz = pa[2 * Npad + j];
    B[0] = fma(px, qx, B[0]);
    B[1] = fma(px, qy, B[1]);
    B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]);
    B[4] = fma(py, qy, B[4]);
    B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]);
    B[7] = fma(pz, qy, B[7]);
    B[8] = fma(pz, qz, B[8]);
  }
  double R[9];
  (void)kabsch_rotation(B, R);
  double ssq = 0.0, mx = 0.0;
  for (int a = 0; a < A; ++a) {
    const double *pa = Xs + (int64_t)(a * 3) * Npad;
    const double px = pa[i], py = pa[Npad + i], pz = pa[2 * Npad + i];
    const double qx = pa[j], qy = pa[Npad + j], qz = pa[2 * Npad + j];
    const double dx = px - (R[0] * qx + R[1] * qy + R[2] * qz);
    const double dy = py - (R[3] * qx + R[4] * qy + R[5] * qz);
    const double dz = pz - (R[6] * qx + R[7] * qy + R[8] * qz);
    const double s = dx * dx + dy * dy + dz * dz;
    ssq += s;
    mx = fmax(mx, s);
  }
  rmsd = sqrt(ssq / (double)A);
  maxdev = sqrt(mx);
  if (R_out) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R_out[k] = R[k];
  }
}

// Same evaluation from the conformer-major copy Xa: a lane that owns one pair
// streams two contiguous A*24-byte blocks instead of gathering 6*A strided
// doubles -- used wherever lanes of a wave hold unrelated pairs.
__device__ __forceinline__ void pair_exact_aos(const double *__restrict__ Xa, int A, int64_t i,
                                               int64_t j, double &rmsd, double &maxdev,
                                               double *R_out = nullptr) {
  const double *__restrict__ p = Xa + i * (int64_t)A * 3;
  const double *__restrict__ q = Xa + j * (int64_t)A * 3;
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a < A; ++a) {
    const double px = p[a * 3], py = p[a * 3 + 1], pz = p[a * 3 + 2];
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
  }
  double R[9];
  (void)kabsch_rotation(B, R);
  double ssq = 0.0, mx = 0.0;
  for (int a = 0; a < A; ++a) {
    const double px = p[a * 3], py = p[a * 3 + 1], pz = p[a * 3 + 2];
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    const double dx = px - (R[0] * qx + R[1] * qy + R[2] * qz);
    const double dy = py - (R[3] * qx + R[4] * qy + R[5] * qz);
    const double dz = pz - (R[6] * qx + R[7] * qy + R[8] * qz);
    const double s = dx * dx + dy * dy + dz * dz;
    ssq += s;
    mx = fmax(mx, s);
  }
  rmsd = sqrt(ssq / (double)A);
  maxdev = sqrt(mx);
  if (R_out) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R_out[k] = R[k];
  }
}

// Same evaluation with EIGHT lanes per pair (sub-lane s = lane & 7 takes atoms
// s, s+8, ...; 8 consecutive atoms are 192 contiguous bytes of Xa): the atom
// loops shrink eightfold and the partial sums meet through three xor-shuffles
// that never leave the 8-lane group.  Cuts the latency of the sparse refine,
// whose few hundred pairs cannot fill the chip anyway.
__device__ __forceinline__ double group8_sum(double v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}
__device__ __forceinline__ double group8_max(double v) {
  v = fmax(v, __shfl_xor(v, 1));
  v = fmax(v, __shfl_xor(v, 2));
  v = fmax(v, __shfl_xor(v, 4));
  return v;
}

__device__ __forceinline__ void pair_exact_group8(const double *__restrict__ Xa, int A, int64_t i,
                                                  int64_t j, int sub, double &rmsd, double &maxdev) {
  const double *__restrict__ p = Xa + i * (int64_t)A * 3;
  const double *__restrict__ q = Xa + j * (int64_t)A * 3;
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = sub; a < A; a += 8) {
    const double px = p[a * 3], py = p[a * 3 + 1], pz = p[a * 3 + 2];
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) B[e] = group8_sum(B[e]);
  double R[9];
  (void)kabsch_rotation(B, R);  // identical in the 8 lanes of the group
  double ssq = 0.0, mx = 0.0;
  for (int a = sub; a < A; a += 8) {
    const double px = p[a * 3], py = p[a * 3 + 1], pz = p[a * 3 + 2];
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    const double dx = px - (R[0] * qx + R[1] * qy + R[2] * qz);
    const double dy = py - (R[3] * qx + R[4] * qy + R[5] * qz);
    const double dz = pz - (R[6] * qx + R[7] * qy + R[8] * qz);
    const double s = dx * dx + dy * dy + dz * dz;
    ssq += s;
    mx = fmax(mx, s);
  }
  rmsd = sqrt(group8_sum(ssq) / (double)A);
  maxdev = sqrt(group8_max(mx));
}

// Same evaluation with the whole wavefront on ONE pair: lane = atom (strided),
// the nine covariance sums and the deviation sum / max are reduced across the
// wave with xor shuffles; the 4x4 eigen-solve runs redundantly in every lane.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ void pair_exact_wave(const double *__restrict__ Xs, int64_t Npad, int A,
                                                int64_t i, int64_t j, int lane, double &rmsd,
                                                double &maxdev) {
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = lane; a < A; a += 64) {
    const double *pa = Xs + (int64_t)(a * 3) * Npad;
    const double px = pa[i], py = pa[Npad + i], pz = pa[2 * Npad + i];
    const double qx = pa[j], qy = pa[Npad + j], qz = pa[2 * Npad + j];
    B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) B[e] = wave_sum_f64(B[e]);
  double R[9];
  (void)kabsch_rotation(B, R);
  double ssq = 0.0, mx = 0.0;
  for (int a = lane; a < A; a += 64) {
    const double *pa = Xs + (int64_t)(a * 3) * Npad;
    const double px = pa[i], py = pa[Npad + i], pz = pa[2 * Npad + i];
    const double qx = pa[j], qy = pa[Npad + j], qz = pa[2 * Npad + j];
    const double dx = px - (R[0] * qx + R[1] * qy + R[2] * qz);
    const double dy = py - (R[3] * qx + R[4] * qy + R[5] * qz);
    const double dz = pz - (R[6] * qx + R[7] * qy + R[8] * qz);
    const double s = dx * dx + dy * dy + dz * dz;
    ssq += s;
    mx = fmax(mx, s);
  }
  rmsd = sqrt(wave_sum_f64(ssq) / (double)A);
  maxdev = sqrt(wave_max_f64(mx));
}

__global__ void __launch_bounds__(256)
k_pairs_exact(const double *__restrict__ Xa, int A, const int64_t *__restrict__ pi,
              const int64_t *__restrict__ pj, int64_t P, double *__restrict__ rmsd,
              double *__restrict__ maxdev, double *__restrict__ Rout) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double r, m;
  pair_exact_aos(Xa, A, pi[p], pj[p], r, m, Rout ? Rout + p * 9 : nullptr);
  rmsd[p] = r;
  maxdev[p] = m;
}

// all pairs with both outputs: one wavefront per (row i, 64-column tile);
// row-major coalesced stores, lower triangle mirrored by the host.
__global__ void __launch_bounds__(256)
k_matrix_exact(const double *__restrict__ Xs, int64_t N, int64_t Npad, int A,
               double *__restrict__ rmsd, double *__restrict__ maxdev) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t NT = Npad >> 6;
  const int64_t i = wave / NT;
  const int64_t jt = wave % NT;
  if (i >= N) return;
  if (jt * 64 + 63 <= i) return;  // tile entirely at or below the diagonal
  const int64_t j = jt * 64 + lane;
  if (j >= N || j <= i) return;
  double r, m;
  pair_exact(Xs, Npad, A, i, j, r, m);
  rmsd[i * N + j] = r;
  maxdev[i * N + j] = m;
}

// Candidate pairs found by a screen kernel are also appended to a pair queue
// ((i << 32) | j, global indices) so that the exact re-evaluation can give
// every lane its own pair.  counters[6] counts ALL candidates, also those
// beyond the queue capacity Q -- the refine kernel then falls back to the
// (always complete) word queue.
__device__ __forceinline__ void push_pairs(uint64_t m, bool may, unsigned i, unsigned j,
                                           uint64_t *__restrict__ pairq, unsigned long long Q,
                                           unsigned long long *__restrict__ counters, int lane) {
  if (m == 0) return;  // wave-uniform
#ifdef FC_ABLATE_PUSH  // timing experiment only: results are wrong
  return;
#endif
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(&counters[6], (unsigned long long)__popcll(m));
  base = __shfl(base, 0);
  if (may) {
    const unsigned long long slot = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    if (slot < Q) pairq[slot] = ((uint64_t)i << 32) | (uint64_t)j;
  }
}

// Candidate staging of the MFMA screen: a workgroup collects its candidate pairs and
// non-empty words in LDS and publishes them with ONE global atomic each when it is
// done, instead of one contended atomic-with-return per non-empty ballot (measured:
// 4 % of the kernel at 10^4 conformers, 17 % at 5*10^3).  A ballot that does not fit
// goes straight to the global queue; the slots it reserved stay at the "empty" marker.
#ifndef FC_REFINE_ROUNDS
#define FC_REFINE_ROUNDS 4  // measured: 8 -> 66 us, 4 -> 53 us, 2 -> 59 us, 1 (all redundant) -> 82 us; with the rounds' loads side by side (round 5): 2 -> 44, 4 -> 34, 8 -> 70 us (scratch)
#endif
#ifndef FC_V2_ALIGN
#define FC_V2_ALIGN 1
#endif
constexpr int kStagePairs = 48;  // uint64 entries per workgroup
constexpr int kStageWords = 24;  // uint32 entries per workgroup
constexpr size_t kStageBytes = kStagePairs * 8 + kStageWords * 4 + 8;

// the single-precision screen has LDS to spare (three workgroups of 42 KB per CU): its staging
// area takes the candidates of an ensemble with ~4 % of similar pairs before it has to fall back
// to one global atomic per ballot (measured on a 10^4 x 50 ensemble with a continuous RMSD
// distribution, 1.7 % candidates: screen 4.3 ms with 48 slots)
#ifndef FC_STAGE_PAIRS_F32
#define FC_STAGE_PAIRS_F32 384
#endif
constexpr int kStagePairsF32 = FC_STAGE_PAIRS_F32;
constexpr int kStageWordsF32 = 128;
constexpr size_t kStageBytesF32 = kStagePairsF32 * 8 + kStageWordsF32 * 4 + 16 + 32 * 4;  // + counters + unit list

template <int CAP = kStagePairs>
__device__ __forceinline__ void stage_pairs(uint64_t m, bool may, unsigned i, unsigned j,
                                            uint64_t *__restrict__ sq, unsigned *__restrict__ scnt,
                                            uint64_t *__restrict__ pairq, unsigned long long Q,
                                            unsigned long long *__restrict__ counters, int lane) {
  if (m == 0) return;  // wave-uniform
#ifdef FC_ABLATE_PUSH
  return;
#endif
#ifdef FC_ABLATE_STAGE  // timing experiment only (candidates are lost; the decision stays alive through one LDS store)
  if (lane == 0) scnt[2] = (unsigned)__popcll(m);
  return;
#endif
  const unsigned n = (unsigned)__popcll(m);
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(scnt, n);
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  if (base + n <= (unsigned)CAP) {
    if (may) sq[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = ((uint64_t)i << 32) | (uint64_t)j;
  } else {
#ifndef FC_ABLATE_OVERFLOW  // timing experiment only (candidates are lost)
    push_pairs(m, may, i, j, pairq, Q, counters, lane);
#endif
  }
}

// The same with a staging region and a count PER WAVEFRONT: no LDS atomic, no wait for its result in front of the
// store (k_simbits_screen_mfma_h2; the workgroup publishes all four regions with ONE global atomic at its end).
template <int WCAP>
__device__ __forceinline__ void stage_pairs_wave(uint64_t m, bool may, unsigned i, unsigned j, uint64_t *__restrict__ sq_wave,
                                                 int &n_staged, uint64_t *__restrict__ pairq, unsigned long long Q,
                                                 unsigned long long *__restrict__ counters, int lane) {
  if (m == 0) return;  // wave-uniform
  const int n = (int)__popcll(m);
  if (n_staged + n <= WCAP) {
    if (may) sq_wave[n_staged + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((uint64_t)i << 32) | (uint64_t)j;
    n_staged += n;
  } else {
    push_pairs(m, may, i, j, pairq, Q, counters, lane);
  }
}

// The fp64 screen has no LDS to spare (two workgroups of 80 KB per CU at 50 atoms leave 48 bytes): its 384 bytes of pair
// staging hold 96 four-byte entries (row within the item << 6 | column within the tile), a share of them per wavefront,
// counted in a scalar register; a wavefront whose share is full publishes it with ONE global atomic and starts over.
// (Until round 4: 48 eight-byte entries shared through an LDS atomic, and one global atomic per 16-row ballot once
// they were full -- 4.3 of the 4.96 ms of a prune of the continuous-RMSD ensemble on this screen.)
__device__ __forceinline__ void flush_pairs_wave32(const uint32_t *__restrict__ sq_wave, int n_staged, unsigned i0, unsigned j0,
                                                   uint64_t *__restrict__ pairq, unsigned long long Q,
                                                   unsigned long long *__restrict__ counters, int lane) {
  if (n_staged == 0) return;  // wave-uniform
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(&counters[6], (unsigned long long)n_staged);
  base = __shfl(base, 0);
  for (int idx = lane; idx < n_staged; idx += 64) {
    const uint32_t e = sq_wave[idx];
    const unsigned long long slot = base + (unsigned long long)idx;
    if (slot < Q) pairq[slot] = ((uint64_t)(i0 + (e >> 6)) << 32) | (uint64_t)(j0 + (e & 63u));
  }
}
template <int WCAP>
__device__ __forceinline__ void stage_pairs_wave32(uint64_t m, bool may, unsigned i, unsigned j, unsigned i0, unsigned j0,
                                                   uint32_t *__restrict__ sq_wave, int &n_staged,
                                                   uint64_t *__restrict__ pairq, unsigned long long Q,
                                                   unsigned long long *__restrict__ counters, int lane) {
  if (m == 0) return;  // wave-uniform
  const int n = (int)__popcll(m);
  if (n > WCAP) {  // (more pairs in one ballot than a share holds)
    push_pairs(m, may, i, j, pairq, Q, counters, lane);
    return;
  }
  if (n_staged + n > WCAP) {
    flush_pairs_wave32(sq_wave, n_staged, i0, j0, pairq, Q, counters, lane);
    n_staged = 0;
  }
  if (may) sq_wave[n_staged + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((i - i0) << 6) | (j - j0);
  n_staged += n;
}

// ---------------------------------------------------------------------------
// k_simbits_screen -- the dominant kernel of the pruning stage.
//
// Workgroup = 4 wavefronts = (one 64-column tile jt) x (one block of IB rows).
// The column tile (A*3*64 doubles) is staged ONCE in LDS and reused by all
// IB rows; each wavefront takes TI=8 rows at a time: the 8 row conformers are
// wave-uniform (scalar loads, SGPR operands of the FMAs), lane l owns column
// jt*64+l, and the 8x9 covariance accumulators stay in VGPRs.  Per atom a
// lane issues 3 conflict-free ds_read_b64 and 72 v_fma_f64.
// The decision msd(i,j) < thr2 (+margin) is the division-free polynomial
// screen of fc_kabsch_math.h; the 64 lane decisions of a row are one ballot
// = one uint64 word of the bit matrix, written by lane 0 (no atomics).
// Pairs that pass the screen are re-evaluated exactly by k_simbits_refine.
// ---------------------------------------------------------------------------
template <bool USE_LDS, int TI, int NW>
__global__ void __launch_bounds__(NW * 64)
k_simbits_screen(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N,
                 int64_t Npad, int A, double A_thr2, int IB, int64_t rank, int64_t world,
                 uint64_t *__restrict__ bits, int64_t W, uint32_t *__restrict__ cand,
                 unsigned long long *__restrict__ counters, uint64_t *__restrict__ pairq,
                 unsigned long long Q) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t jt = blockIdx.x;
  const int64_t lb = blockIdx.y;                 // local row block
  const int64_t gb = global_block(lb, rank, world);  // global row block
  const int64_t i0 = gb * IB;
  if (i0 >= N) return;
  if (jt * 64 + 63 <= i0) return;                // nothing above the diagonal here
  const int64_t j = jt * 64 + lane;

  if (USE_LDS) {
    const int total = A * 3 * 64;
    for (int idx = tid; idx < total; idx += NW * 64) {
      const int ac = idx >> 6, l = idx & 63;
      lds[idx] = Xs[(int64_t)ac * Npad + jt * 64 + l];
    }
    __syncthreads();
  }
  const double Gq = G[j];
  const double *__restrict__ xq = Xs + jt * 64 + lane;

  for (int it = wv; it * TI < IB; it += NW) {
    const int64_t ib = i0 + (int64_t)it * TI;    // first row of this 8-row tile
    if (ib >= N) break;
    if (jt * 64 + 63 <= ib) break;               // later tiles are further below
    double acc[TI][9];
#pragma unroll
    for (int k = 0; k < TI; ++k)
#pragma unroll
      for (int e = 0; e < 9; ++e) acc[k][e] = 0.0;

    const double *__restrict__ xp = Xs + ib;     // wave-uniform
#pragma unroll 2
    for (int a = 0; a < A; ++a) {
      double qx, qy, qz;
      if (USE_LDS) {
        qx = lds[(a * 3 + 0) * 64 + lane];
        qy = lds[(a * 3 + 1) * 64 + lane];
        qz = lds[(a * 3 + 2) * 64 + lane];
      } else {
        qx = xq[(int64_t)(a * 3 + 0) * Npad];
        qy = xq[(int64_t)(a * 3 + 1) * Npad];
        qz = xq[(int64_t)(a * 3 + 2) * Npad];
      }
      const double *__restrict__ px = xp + (int64_t)(a * 3 + 0) * Npad;
      const double *__restrict__ py = xp + (int64_t)(a * 3 + 1) * Npad;
      const double *__restrict__ pz = xp + (int64_t)(a * 3 + 2) * Npad;
#pragma unroll
      for (int k = 0; k < TI; ++k) {
        const double x = px[k], y = py[k], z = pz[k];
        acc[k][0] = fma(x, qx, acc[k][0]);
        acc[k][1] = fma(x, qy, acc[k][1]);
        acc[k][2] = fma(x, qz, acc[k][2]);
        acc[k][3] = fma(y, qx, acc[k][3]);
        acc[k][4] = fma(y, qy, acc[k][4]);
        acc[k][5] = fma(y, qz, acc[k][5]);
        acc[k][6] = fma(z, qx, acc[k][6]);
        acc[k][7] = fma(z, qy, acc[k][7]);
        acc[k][8] = fma(z, qz, acc[k][8]);
      }
    }
#pragma unroll
    for (int k = 0; k < TI; ++k) {
      const int64_t i = ib + k;
      const double Gp = G[i];  // Npad-padded, wave-uniform
      bool may = kabsch_may_be_below(acc[k], Gp + Gq, A_thr2);
      may = may && (j > i) && (j < N) && (i < N);
      const uint64_t word = __ballot(may);
      push_pairs(word, may, (unsigned)i, (unsigned)j, pairq, Q, counters, lane);
      if (lane == 0 && i < N) {
        const int64_t lrow = (int64_t)(lb * IB) + (int64_t)it * TI + k;
        bits[lrow * W + jt] = word;
        if (word) {  // queue the word for exact re-evaluation (k_simbits_refine)
          const unsigned long long slot = atomicAdd(&counters[4], 1ull);
          cand[slot] = (uint32_t)(lrow * W + jt);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_simbits_screen_mfma -- same screen, covariance on the fp64 matrix pipe.
//
// The 9 covariance entries of ALL pairs of a (16-row x 16-column) tile are 9
// small GEMMs  C_xy[r][c] = sum_a Px[r][a] * Qy[c][a]  with K = atoms: exactly
// v_mfma_f64_16x16x4_f64 (A: lane l holds row l&15, k = l>>4; B: column l&15,
// k = l>>4; D: column l&15, row (l>>4)+4*reg).  After the K loop a lane owns
// the full 3x3 covariance of 4 pairs per tile, so the polynomial screen runs
// per lane exactly as in the VALU kernel.  Measured on MI355X
// (tools/ubench_f64.hip): v_fma_f64 reaches 38/52/65 TFLOP/s at 1/2/4 waves
// per SIMD, the f64 MFMA 71/76/78 -- and the VALU kernel additionally starves
// on scalar-cache misses for its wave-uniform operand (profiles/, round 1),
// which is why the matrix pipe is the default for this contraction.
//
// Workgroup = NW waves = (one 64-column tile) x (IB rows).  Column tile in LDS,
// k-grouped and half-swizzled so that the four k-slices a ds_read_b64 touches
// fall on disjoint banks:  [(s*3+c)][k>>1][cs][k&1][16].  Row operands stream
// from L2 one k-step ahead (3 doubles per lane).  A wave takes 16 rows at a
// time against two 16-column sub-tiles (18 accumulators of 4 doubles).
// ---------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));

// VALUES = true: same tiling, but the epilogue solves for the largest quaternion
// eigenvalue (Newton) and stores rmsd(i, j) into a dense (N, N) matrix instead of
// screening; pairs with rmsd below `A_thr2` (reused as the small-rmsd^2 * A limit)
// are queued for the exact explicit-difference evaluation.
// TC: conformers of the column tile held in LDS.  64 everywhere except the complete alignments of structures whose
// 64-column tile no longer fits the CU's LDS: 105 ... 208 atoms take 32 columns, 209 ... 416 atoms 16 (MODE 2 only)
// instead of leaving the tiled kernel for k_matrix_exact, 5.4 x slower per pair.
// EIG (MODE 2): the rmsd of a pair from its largest eigenvalue -- sum |p - R q|^2 = (Gp + Gq) - 2 lambda for the optimal
// rotation, lambda as the Newton iteration left it -- instead of a running sum in the atom pass (one fp64 instruction per
// atom and pair less, 50 of ~1 100 per pair at 50 atoms; the maximum deviation still comes from the explicit rotated
// difference).  The difference carries ~u (Gp + Gq) of rounding, so it holds 5e-11 in the rmsd only for pairs further
// apart than ~1e-3 A (msd A > 2e-10 (Gp + Gq)^2 / A): closer pairs go to the fix-up kernel like the pairs whose rotation
// was declined.  EIG = false (the launcher's retry when the fix-up queue overflowed: an ensemble of near-duplicates): the sum.
template <int NW, int MODE = 0, int TC = 64, bool EIG = false>
__global__ void __launch_bounds__(NW * 64, 2)
k_simbits_screen_mfma(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N,
                      int64_t Npad, int A, double A_thr2, int IB, int64_t rank, int64_t world,
                      uint64_t *__restrict__ bits, int64_t W, uint32_t *__restrict__ cand,
                      unsigned long long *__restrict__ counters, uint64_t *__restrict__ pairq,
                      unsigned long long Q, const uint64_t *__restrict__ item_table, unsigned long long n_items,
                      double *__restrict__ rmsd_out = nullptr, const unsigned long long *__restrict__ gate = nullptr,
                      double *__restrict__ maxdev_out = nullptr) {
  static_assert(!EIG || MODE == 2, "EIG: the complete alignments");
  constexpr bool VALUES = MODE != 0;  // 1: rmsd values (Newton), 2: (rmsd, maxdev) by explicit difference
  // sub-tiles (16 columns) per unit: two share the row operands in the screens; the complete-alignment
  // mode takes one at a time -- its epilogue (four rotations, an atom pass) needs the registers the second
  // set of 36 accumulator doubles would hold (256 VGPRs + 220 B of scratch with two: 5.17 ms per 10^4 x 10^4)
  constexpr int NT = MODE == 2 ? 1 : 2;
  static_assert(TC == 64 || ((TC == 32 || TC == 16) && MODE == 2), "column tiles: 64, or 32 / 16 for the complete alignments");
  constexpr int NU = (TC / 16) / NT;
  constexpr int KPR = TC == 64 ? 2 : 4;      // atoms of a k-group in one 1-KiB run of the LDS image (TC = 64: the other two in the next run)
  constexpr int SST = 2048 / TC;             // doubles from one 16-column sub-tile to the next inside a run
  constexpr int CST = TC == 64 ? 256 : 128;  // ... from one coordinate of a k-group to the next
  // doubles in front of k-step sl: 12 TC per k-step -- TC = 16: a run holds TWO k-steps of one coordinate ([k-step][atom][16
  // columns]), so k-steps come in pairs of three runs
  auto koff = [](int sl) { return TC == 16 ? (sl >> 1) * 384 + (sl & 1) * 64 : sl * (12 * TC); };
  // behind a speculative fp32 screen: run only when k_screen_verdict asked for it
  if (gate != nullptr && *gate == 0ull) return;
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = (A + 3) >> 2;
  double *__restrict__ ldsG = lds + (TC == 16 ? ((KS + 1) >> 1) * 384 : KS * 12 * TC);  // [TC column sums | IB row sums]
  uint64_t *__restrict__ stageQ = reinterpret_cast<uint64_t *>(ldsG + TC + IB);
  uint32_t *__restrict__ stageW = reinterpret_cast<uint32_t *>(stageQ + kStagePairs);
  unsigned *__restrict__ stageN = reinterpret_cast<unsigned *>(stageW + kStageWords);  // [pairs, words]
#ifdef FC_TIMELINE  // tuning build: per-item start / fill-done / end timestamps (100 MHz)
  unsigned long long *tl = reinterpret_cast<unsigned long long *>(rmsd_out);
#endif
  // One (row block, column tile) item per workgroup.  A persistent variant (grid = what the
  // chip holds, items drawn from a device counter) was built and measured with
  // tools/attic/timeline_probe.py: it removes the ~6 us dispatch gap between two 74 us workgroups
  // on a slot, but the 512 workgroups then run in lock-step -- all column tiles are fetched
  // in the same 12 us bursts (3.3 TB/s) instead of spread out -- and the loop-carried state
  // costs registers: 1.03 ms against 0.96 ms.  Not kept.
  const unsigned long long b = blockIdx.x;
  if (b >= n_items) return;
#ifdef FC_TIMELINE
  if (!VALUES && tl && tid == 0) {
    tl[(size_t)b * 4] = wall_clock64();
    tl[(size_t)b * 4 + 3] = b;
  }
#endif
  // (row block, column tile) of this workgroup: from the host-built list of the items that
  // touch the upper triangle (lb << 32 | jt), or the plain 2-D enumeration when there is none
  // bit 31 / bit 63 of an entry: the item covers only the first / second half of the row
  // block's 16-row tiles (the last items of a launch are halves, which shortens the tail)
  int64_t jt, lb;
  int it_first = 0, it_last = IB >> 4;
  if (item_table != nullptr) {
    const uint64_t it = item_table[b];
    lb = (int64_t)((it >> 32) & 0x7fffffffull);
    jt = (int64_t)(it & 0x7fffffffull);
    if (it & (1ull << 31)) it_last = IB >> 5;
    if (it & (1ull << 63)) it_first = IB >> 5;
  } else {
    const int64_t NT = Npad / TC;
    jt = (int64_t)(b % (unsigned long long)NT);
    lb = (int64_t)(b / (unsigned long long)NT);
  }
  const int64_t j0 = jt * TC;
  const int64_t i0 = global_block(lb, rank, world) * IB;
  if (i0 >= N) return;               // block-uniform
  if (j0 + TC - 1 <= i0) return;     // nothing above the diagonal in this item
  // MODE 0 (the screen): the pair staging area as four-byte entries, a share per wavefront (stage_pairs_wave32)
  uint32_t *__restrict__ sq32_wave = reinterpret_cast<uint32_t *>(stageQ) + wv * (kStagePairs * 2 / NW);
  int n_staged32 = 0;
  (void)sq32_wave;
  (void)n_staged32;

  {  // stage the column tile by LDS-DMA (global_load_lds_dwordx4): no VGPR staging, every
    // piece of a wave in flight at once.  The hardware writes lane l's 16 bytes at
    // (wave-uniform base) + 16 l, so each instruction fills one 1-KiB run of the swizzled
    // image -- run q = (sgrp*3 + c)*2 + (k>>1) holds [cs][k&1][16 columns] -- and the swizzle
    // is applied to the per-lane SOURCE address.  (Register staging took 4 dependent L2 round
    // trips, 9.7 us of a 74 us workgroup: tools/attic/timeline_probe.py.)
    // (TC = 32: a run holds [cs (2)][k (4)][16 columns] -- all four atoms of a k-group for one coordinate, three runs per
    // k-step; TC = 16: [k-step of a pair (2)][k (4)][16 columns], three runs per PAIR of k-steps)
    const int n_runs = TC == 64 ? KS * 6 : (TC == 32 ? KS * 3 : ((KS + 1) >> 1) * 3);
    const int cs_l = TC == 64 ? lane >> 4 : lane >> 5, k1_l = TC == 64 ? (lane >> 3) & 1 : (lane >> 3) & 3, c15_l = (lane & 7) * 2;
    for (int q = wv; q < n_runs; q += NW) {
      const int sc = TC == 64 ? q >> 1 : q, kh = TC == 64 ? q & 1 : 0;  // sc = sgrp*3 + c
      const int sg = sc / 3, c = sc - sg * 3;
      int a = sg * 4 + kh * 2 + k1_l;
      if (TC == 16) {  // (sg counts pairs of k-steps, cs_l picks the k-step; past the last atom row: any row, never read)
        a = (sg * 2 + cs_l) * 4 + k1_l;
        a = a < KS * 4 ? a : KS * 4 - 1;
      }
      const double *src = Xs + (int64_t)(a * 3 + c) * Npad + j0 + (TC == 16 ? 0 : cs_l * 16) + c15_l;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(lds + q * 128), 16, 0, 0);
    }
    // sums of squares of the tile's columns and of the block's rows: the epilogue reads them
    // from LDS (lgkmcnt) so that it never waits on vmcnt behind the row-operand prefetch
    for (int idx = tid; idx < TC + IB; idx += NW * 64) {
      const int64_t g = idx < TC ? j0 + idx : i0 + (idx - TC);
      ldsG[idx] = g < Npad ? G[g] : 0.0;
    }
    if (tid < kStagePairs) stageQ[tid] = ~0ull;
    if (tid < kStageWords) stageW[tid] = ~0u;
    if (tid < 2) stageN[tid] = 0u;
    __syncthreads();
#ifdef FC_TIMELINE
    if (!VALUES && tl && tid == 0) tl[(size_t)b * 4 + 1] = wall_clock64();
#endif
  }
  const int kq = lane >> 4, l15 = lane & 15;
  // MODE 2 computes the tiles TRANSPOSED (column conformers as the instruction's first operand, D rows =
  // columns of the pair matrix) and hands row m of the instruction the column 4 (m & 3) + (m >> 2) of the
  // sub-tile: lane (kq, l15) then owns the pairs (row ib + l15, columns 4 kq + 0..3) -- ONE row conformer
  // per lane (16 consecutive addresses per load instruction instead of four pairs of 4) and its four
  // column conformers in 32 consecutive bytes of the LDS tile (see the epilogue)
  const int lcol = MODE == 2 ? 4 * (l15 & 3) + (l15 >> 2) : l15;
  const int boff = (kq / KPR) * 128 + (kq % KPR) * 16 + lcol;
  uint16_t *bits16 = reinterpret_cast<uint16_t *>(bits);

  // Row operands of the first three k-steps of the NEXT unit (same rows for the second
  // half, the wave's next 16-row tile after it) are requested right after the last MFMA
  // of a unit has issued, so their L2 latency passes under the polynomial epilogue
  // instead of in front of the next K loop.  pre_it = row tile those registers hold.
  double a0[3], a1[3], a2[3];
  int pre_it = -1;
  auto row_offsets = [&](int it_, unsigned (&vo)[3]) {
    const int64_t ib_ = i0 + (int64_t)it_ * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) vo[c] = (unsigned)(((int64_t)(kq * 3 + c) * Npad + ib_ + l15) * 8);  // BYTES inside a k-step
  };
  auto fetch_a_at = [&](double (&a)[3], unsigned (&vo)[3], int sx) {
    const int sl = sx < KS ? sx : KS - 1;  // fewer than 3 k-steps: harmless re-read
    // wave-uniform base + the lane's 32-bit BYTE offset: one addressing mode of the load (with element offsets the compiler
    // formed a 64-bit address per load: three vector instructions per k-step beside nine matrix ones)
    const char *__restrict__ xs_s = reinterpret_cast<const char *>(Xs + (int64_t)sl * 12 * Npad);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // (the offset passes through an empty asm in place: that keeps its zero-extension next to the load -- hoisted out of the
      // K loop it is a 64-bit add per load again -- and costs no copy)
      asm volatile("" : "+v"(vo[c]));
      a[c] = *reinterpret_cast<const double *>(xs_s + vo[c]);
    }
  };

  for (int it = it_first + wv; it < it_last; it += NW) {
    const int64_t ib = i0 + (int64_t)it * 16;
    if (ib >= N) break;
    if (j0 + TC - 1 <= ib) break;
    const int64_t lrow0 = lb * IB + (int64_t)it * 16;
    unsigned nz = 0;  // lanes 0..15: OR of the 16-bit pieces written for row ib + lane
    // per-lane element offsets of the lane's row operand inside one k-step
    // (fits 32 bits: checked by the launcher)
    unsigned voff[3];
    row_offsets(it, voff);

#pragma unroll 1
    for (int half = 0; half < NU; ++half) {
      const int cs0 = half * NT;
      if (j0 + (cs0 + NT) * 16 - 1 <= ib) {  // the unit's sub-tiles at or below the diagonal
        if (!VALUES && lane < 16 && ib + lane < N) {
          bits16[((lrow0 + lane) * W + jt) * 4 + cs0] = 0;
          bits16[((lrow0 + lane) * W + jt) * 4 + cs0 + 1] = 0;
        }
        continue;
      }
      d4_t acc[NT][9];  // (written by the first k-step, whose third operand is the constant zero: no 36 moves per sub-tile)

      // Operand sets rotate through the K loop so that no register copies are
      // needed: row operands (L2 / Infinity Cache, ~1 us under load) in three
      // sets -- the set consumed by k-step s is refilled with k-step s+3 right
      // after its 18 MFMAs have issued, two full k-steps ahead of its use --
      // column operands (LDS) in two sets, one k-step ahead.
      const double *__restrict__ lb0 = lds + cs0 * SST + boff;
      double b0[NT][3], b1[NT][3];
      const int KSe = KS;
      auto fetch_a = [&](double (&a)[3], int sx) {
        fetch_a_at(a, voff, sx);  // past the end: harmless re-read, replaced by the prefetch below
      };
      auto fetch_b = [&](double (&b)[NT][3], int sx) {
        const int sl = sx < KS ? sx : KS - 1;
        const double *__restrict__ lb_s = lb0 + koff(sl);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < 3; ++c) b[t][c] = lb_s[c * CST + t * SST];
      };
      auto mma = [&](const double (&a)[3], const double (&b)[NT][3], auto first_) {
        constexpr bool first = decltype(first_)::value;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int y = 0; y < 3; ++y)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const d4_t c_in = first ? d4_t{0.0, 0.0, 0.0, 0.0} : acc[t][x * 3 + y];
              acc[t][x * 3 + y] = MODE == 2 ? __builtin_amdgcn_mfma_f64_16x16x4f64(b[t][y], a[x], c_in, 0, 0, 0)
                                            : __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[t][y], c_in, 0, 0, 0);
            }
      };
      if (pre_it != it) {  // first unit of the wave, or the prediction below missed
        fetch_a_at(a0, voff, 0);
        fetch_a_at(a1, voff, 1);
        fetch_a_at(a2, voff, 2);
      }
      fetch_b(b0, 0);
      fetch_b(b1, 1);
#define FC_KSTEP(AX, BX, U)             \
  if (sgrp + (U) < KSe) {               \
    mma(AX, BX, std::false_type{});     \
    fetch_a(AX, sgrp + (U) + 3);        \
    fetch_b(BX, sgrp + (U) + 2);        \
  }
      mma(a0, b0, std::true_type{});  // k-step 0 (there is always one)
      fetch_a(a0, 3);
      fetch_b(b0, 2);
      for (int sgrp = 1; sgrp < KSe; sgrp += 6) {
        FC_KSTEP(a1, b1, 0)
        FC_KSTEP(a2, b0, 1)
        FC_KSTEP(a0, b1, 2)
        FC_KSTEP(a1, b0, 3)
        FC_KSTEP(a2, b1, 4)
        FC_KSTEP(a0, b0, 5)
      }
#undef FC_KSTEP
      {  // request the next unit's first three k-steps now; they land during the epilogue
        const int nit = half < NU - 1 ? it : it + NW;
        const bool more = half < NU - 1 || ((nit < it_last) && (i0 + (int64_t)nit * 16 < N) &&
                                        !(j0 + TC - 1 <= i0 + (int64_t)nit * 16));
        if (more) {
          unsigned vn[3];
          row_offsets(nit, vn);
          fetch_a_at(a0, vn, 0);
          fetch_a_at(a1, vn, 1);
          fetch_a_at(a2, vn, 2);
          pre_it = nit;
        } else {
          pre_it = -1;
        }
      }
      // epilogue: lane owns pairs (ib + kq + 4r, j0 + cs*16 + l15), r = 0..3
      const int n32 = (int)N, ib32 = (int)ib;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int cs = cs0 + t;
        const int j = (int)j0 + cs * 16 + l15;
        const double Gq = ldsG[cs * 16 + l15];
        uint64_t mr[4] = {0, 0, 0, 0};
        if (MODE == 2) {
          // complete alignment (a4: rmsd_and_max); lane = (row ib + l15, columns 4 kq + r, r = 0..3).
          // Step 1: the rotation of each of the lane's four pairs from its covariance, as a unit
          // quaternion (Newton eigenvalue + adjugate column: kabsch_quaternion_qcp), then as -R.
          // Step 2: ONE pass over the atoms accumulating the explicit rotated difference of the four
          // pairs, which is what the reference computes: the row conformer's atom from L1/L2 (3 loads of 16
          // consecutive doubles per atom and wave), the four column conformers' from the LDS tile (two
          // 16-byte reads per coordinate), both requested one atom ahead of their use (ping-pong register
          // sets), d = p - R q as three fused chains on -R.  (First version: lane = 4 rows x 1 column, the
          // loop waiting on its own loads: 4.7 ms per 10^4 x 10^4; with the requests ahead 4.3 ms -- it was
          // bound by its 138 M vector-memory instructions, one per 7.6 vector instructions; this layout
          // 3.1 ms.  The fp64 MFMA and the fp64 vector instructions share the DP units on gfx950
          // (SQ_VALU_MFMA_COEXEC_CYCLES = 0): a variant that issued the NEXT unit's covariance inside this
          // atom pass -- one MFMA in front of every 28 vector instructions -- took 3.5 ms and was not kept.)
          // Pairs whose largest eigenvalue is not clearly simple are queued for k_rmsd_fix_small (Jacobi).
          const int i = ib32 + l15;
          const int jb = (int)j0 + cs * 16 + 4 * kq;
          const double Gp = ldsG[TC + it * 16 + l15];
          double nR[4][9];
          bool ok4[4];
          double msdA4[4] = {0.0, 0.0, 0.0, 0.0};  // EIG: (Gp + Gq) - 2 lambda
          {  // the four Newton iterations in one loop (independent chains side by side: 2.71 -> 2.66 ms per 10^4 x 10^4
             // against one pair after the other), -R straight from the adjugate column (no normalisation of q)
            double B4[4][9], GG[4], Q44[4][4], nq4[4], lam4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              GG[r] = Gp + ldsG[cs * 16 + 4 * kq + r];
#pragma unroll
              for (int e = 0; e < 9; ++e) B4[r][e] = acc[t][e][r];
            }
            kabsch_quaternion_qcp_lean4(B4, GG, Q44, nq4, ok4, EIG ? lam4 : nullptr);
            if constexpr (EIG) {
              const double k_small = 2e-10 / (double)A;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                msdA4[r] = GG[r] - 2.0 * lam4[r];
                // too close for the difference (also NaN): a declined pair, the fix-up kernel's explicit sum decides
                if (!(msdA4[r] > k_small * GG[r] * GG[r])) ok4[r] = false, msdA4[r] = 0.0;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // (NaN must not reach the stores: the fix-up overwrites what a declined pair leaves)
              if (!ok4[r]) Q44[r][0] = 1.0, Q44[r][1] = 0.0, Q44[r][2] = 0.0, Q44[r][3] = 0.0, nq4[r] = 1.0;
              neg_rotation_from_raw_quaternion(Q44[r], nq4[r], nR[r]);
            }
          }
          typedef double d2_t __attribute__((ext_vector_type(2)));
          const double *__restrict__ qcol = lds + cs * SST + 4 * kq;
          const double *__restrict__ prow_u = Xs + ib;
          const unsigned l15u = (unsigned)l15;
          // Atoms in rounds of two (ping-pong register sets), two rounds per k-group of four atoms; the zero rows that
          // pad the atoms to a multiple of 4 for the MFMA k-steps add nothing and are not visited (50 atoms = 25
          // rounds, not 26).  Addresses: three running scalar pointers for the row conformer's coordinate rows (advanced
          // by three rows per atom; the lane adds its fixed 32-bit offset) and one LDS pointer per k-group with
          // compile-time offsets inside it -- the first version recomputed both from the atom index: 26 scalar
          // instructions per atom beside 56 vector ones, and a wave issues ONE instruction per turn whatever its kind.
          const int n_rounds = (A + 1) >> 1, n_groups = n_rounds >> 1;
          const bool odd_round = (n_rounds & 1) != 0;
          double ssq[4] = {0.0, 0.0, 0.0, 0.0}, mx[4] = {0.0, 0.0, 0.0, 0.0};
          const int64_t Npad3 = 3 * Npad;
          // wave-uniform: the next atom's three coordinate rows at row ib (scalar base + the lane's 32-bit offset is one
          // addressing mode of the load; the compiler takes it only when it can see that the offset is small)
          const double *__restrict__ pa0 = prow_u, *__restrict__ pa1 = prow_u + Npad, *__restrict__ pa2 = prow_u + 2 * Npad;
          const double *__restrict__ qg = qcol;    // the current k-group of the LDS tile
          auto load_pq = [&](auto u_, double (&P)[3], d2_t (&Qv)[3][2]) {
            constexpr int u = decltype(u_)::value;  // position of the atom in its k-group
            const double *__restrict__ ql = qg + (u / KPR) * 128 + (u % KPR) * 16;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              Qv[c][0] = *reinterpret_cast<const d2_t *>(ql + c * CST);
              Qv[c][1] = *reinterpret_cast<const d2_t *>(ql + c * CST + 2);
            }
            P[0] = pa0[l15u];
            P[1] = pa1[l15u];
            P[2] = pa2[l15u];
            pa0 += Npad3;
            pa1 += Npad3;
            pa2 += Npad3;
          };
          auto accumulate = [&](const double (&P)[3], const d2_t (&Qv)[3][2]) {
#pragma clang fp contract(fast)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double qx = Qv[0][r >> 1][r & 1], qy = Qv[1][r >> 1][r & 1], qz = Qv[2][r >> 1][r & 1];
              const double dx = fma(nR[r][0], qx, fma(nR[r][1], qy, fma(nR[r][2], qz, P[0])));
              const double dy = fma(nR[r][3], qx, fma(nR[r][4], qy, fma(nR[r][5], qz, P[1])));
              const double dz = fma(nR[r][6], qx, fma(nR[r][7], qy, fma(nR[r][8], qz, P[2])));
              const double s2 = fma(dz, dz, fma(dy, dy, dx * dx));
              if constexpr (!EIG) ssq[r] += s2;
              // (fmax() first re-quiets its loop-carried operand: one more instruction per pair and atom)
              asm("v_max_f64 %0, %1, %2" : "=v"(mx[r]) : "v"(mx[r]), "v"(s2));
            }
          };
          using U0 = std::integral_constant<int, 0>;
          using U1 = std::integral_constant<int, 1>;
          using U2 = std::integral_constant<int, 2>;
          using U3 = std::integral_constant<int, 3>;
          double PA[3], PB[3];
          d2_t QA[3][2], QB[3][2];
          load_pq(U0{}, PA, QA);
          // (the scheduling barriers keep the requests where they are written: the machine scheduler
          // otherwise gathers the two sets' loads into one burst right in front of their first use)
          for (int g = 0; g < n_groups; ++g) {
            load_pq(U1{}, PB, QB);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PA, QA);
            __builtin_amdgcn_sched_barrier(0);
            load_pq(U2{}, PA, QA);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PB, QB);
            __builtin_amdgcn_sched_barrier(0);
            load_pq(U3{}, PB, QB);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PA, QA);
            __builtin_amdgcn_sched_barrier(0);
            qg = TC == 16 ? qcol + koff(g + 1) : qg + 12 * TC;
            if (g + 1 < n_groups || odd_round) load_pq(U0{}, PA, QA);  // wave-uniform: nothing is read past the last atom row
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PB, QB);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (odd_round) {  // the last two atoms (the second one may be a zero row)
            load_pq(U1{}, PB, QB);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PA, QA);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(PB, QB);
          }
          if constexpr (EIG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ssq[r] = msdA4[r];
          }
          const bool row_in = i < n32;
          double *__restrict__ ro = rmsd_out + (int64_t)i * N + jb;
          double *__restrict__ mo = maxdev_out + (int64_t)i * N + jb;
          const double invA = 1.0 / (double)A;
          if (row_in && jb > i && jb + 3 < n32 && ((((int64_t)i * N) & 1) == 0)) {
            // all four above the diagonal and inside, 16-byte aligned: two wide stores per matrix
            *reinterpret_cast<d2_t *>(ro) = d2_t{fc_sqrt_nonneg(ssq[0] * invA), fc_sqrt_nonneg(ssq[1] * invA)};
            *reinterpret_cast<d2_t *>(ro + 2) = d2_t{fc_sqrt_nonneg(ssq[2] * invA), fc_sqrt_nonneg(ssq[3] * invA)};
            *reinterpret_cast<d2_t *>(mo) = d2_t{fc_sqrt_nonneg(mx[0]), fc_sqrt_nonneg(mx[1])};
            *reinterpret_cast<d2_t *>(mo + 2) = d2_t{fc_sqrt_nonneg(mx[2]), fc_sqrt_nonneg(mx[3])};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jb + r;
              if (row_in && j >= i && j < n32) {  // the diagonal too (exact zeros there: the outputs need no memset)
                ro[r] = j == i ? 0.0 : fc_sqrt_nonneg(ssq[r] * invA);
                mo[r] = j == i ? 0.0 : fc_sqrt_nonneg(mx[r]);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = jb + r;
            const bool redo = row_in && (j > i) && (j < n32) && !ok4[r];
            stage_pairs(__ballot(redo), redo, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ib32 + kq + 4 * r;
          const double Gp = ldsG[TC + it * 16 + kq + 4 * r];
          double B9[9];
#pragma unroll
          for (int e = 0; e < 9; ++e) B9[e] = acc[t][e][r];
          if (VALUES) {
            const double lam = kabsch_lambda_max(B9, Gp + Gq);
            const double msdA = (Gp + Gq) - 2.0 * lam;
            const bool in = (j > i) && (j < n32) && (i < n32);
            if (in) rmsd_out[(int64_t)i * N + j] = sqrt(fmax(msdA, 0.0) / (double)A);
            const bool redo = in && (msdA < A_thr2);
            stage_pairs(__ballot(redo), redo, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
            continue;
          }
          bool may = kabsch_may_be_below(B9, Gp + Gq, A_thr2);
          may = may && (j > i) && (j < n32) && (i < n32);
          mr[r] = __ballot(may);
          if constexpr (MODE == 0)
            stage_pairs_wave32<kStagePairs * 2 / NW>(mr[r], may, (unsigned)i, (unsigned)j, (unsigned)i0, (unsigned)j0, sq32_wave,
                                                     n_staged32, pairq, Q, counters, lane);
          else
            stage_pairs(mr[r], may, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
        }
        // bit (16 kq' + c) of mr[r] belongs to row kq' + 4 r: lane l < 16 writes the piece of
        // row l = (l & 3) + 4 (l >> 2), one store per sub-tile instead of one per register
        if (!VALUES && lane < 16 && ib32 + lane < n32) {
          const int rr = lane >> 2;
          const uint64_t mine = rr == 0 ? mr[0] : rr == 1 ? mr[1] : rr == 2 ? mr[2] : mr[3];
          const unsigned piece = (unsigned)((mine >> (16 * (lane & 3))) & 0xffffull);
          bits16[((lrow0 + lane) * W + jt) * 4 + cs] = (uint16_t)piece;
          nz |= piece;
        }
      }
    }
    if (!VALUES) {  // queue the non-empty words of this row tile for the exact refine
#ifndef FC_ABLATE_PUSH
      const bool has = lane < 16 && nz != 0;
      const uint64_t mw = __ballot(has);
      if (mw != 0) {  // wave-uniform
        const unsigned n = (unsigned)__popcll(mw);
        const uint32_t word = (uint32_t)((lrow0 + lane) * W + jt);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(stageN + 1, n);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned rank_in = (unsigned)__popcll(mw & ((1ull << lane) - 1ull));
        if (base + n <= (unsigned)kStageWords) {
          if (has) stageW[base + rank_in] = word;
        } else {  // no room: straight to the global queue
          unsigned long long gbase = 0;
          if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)n);
          gbase = __shfl(gbase, 0);
          if (has) cand[gbase + rank_in] = word;
        }
      }
#endif
    }
  }
  // publish what the workgroup staged: one global atomic per queue (the screen's pairs: per wavefront)
  if constexpr (MODE == 0) flush_pairs_wave32(sq32_wave, n_staged32, (unsigned)i0, (unsigned)j0, pairq, Q, counters, lane);
  __syncthreads();
#ifdef FC_TIMELINE
  if (!VALUES && tl && tid == 0) tl[(size_t)b * 4 + 2] = wall_clock64();
#endif
  if (wv == 0) {
    if constexpr (MODE != 0) {
      const uint64_t e = lane < kStagePairs ? stageQ[lane] : ~0ull;
      const bool valid = e != ~0ull;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[6], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) {
          const unsigned long long slot = gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull));
          if (slot < Q) pairq[slot] = e;
        }
      }
    }
    if (!VALUES) {
      const uint32_t wq = lane < kStageWords ? stageW[lane] : ~0u;
      const bool valid = wq != ~0u;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) cand[gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull))] = wq;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_simbits_screen_mfma_f32 -- the same screen with the covariance on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32: twice the rate of the f64 instruction, half the LDS and L2 bytes)
// and the polynomial in fp32 against proven bounds (kabsch_may_be_below_f32).  The screen only
// has to be CONSERVATIVE -- every pair it lets through is decided by the exact fp64 refine --
// so single precision costs candidates, never results: a pair is dropped only when P, P', P''
// clear a bound on the fp32 error; pairs within ~0.03 A^2 of the threshold become candidates.
// Same tiling, item table, staging queues and outputs as the f64 kernel above.  Differences:
//   * operands come from Xsf, an fp32 copy of Xs (same layout);
//   * one LDS-DMA instruction fills a whole [(s*3+c)] run of 256 floats: lane l carries 4
//     consecutive columns of (k>>1 = l>>5, cs = (l>>3)&3, k&1 = (l>>2)&1);
//   * D layout of the f32 instruction: lane (kq, l15), register r holds row 4*kq + r
//     (the f64 one: row kq + 4*r).
// ---------------------------------------------------------------------------
typedef float f4_t __attribute__((ext_vector_type(4)));

// three workgroups per CU (42 KB of LDS each; 168 VGPRs, 10 of them spilled): 0.550 -> 0.511 ms
// against two (measured in one process sequence on one box)
#ifndef FC_F32_WGS
#define FC_F32_WGS 3
#endif
// BITS = false ("lean"): the caller consumes only the candidate-pair queue (the one-launch pair
// ladder, the multi-GPU exchange) -- the dense bit matrix, 12.6 MB of mostly zeros per launch at
// 10^4 conformers, is neither written nor queued by words; when the pair queue overflows the host
// repeats the prune with BITS = true.
//
// The lean kernel also runs the K loop in TWO STAGES (`sub` != nullptr).  The superposition residual
// of a SUBSET S of the atoms bounds the whole from below:
//     A * msd(p, q) = min_R sum_all |p_a - R q_a|^2  >=  min_{R,t} sum_S |p_a - R q_a - t|^2
//                   = G_S^c(p) + G_S^c(q) - 2 lambda_max(B_S^c)
// (G_S^c, B_S^c: norms and covariance of the subset centred on ITS OWN centroid), so a pair whose
// subset already fails  lambda_max(B_S^c) > L_S = (G_S^c(p) + G_S^c(q) - A thr2) / 2  is dissimilar
// whatever the other atoms do.  S = the atoms of the even k-steps (interleaved groups of four, so
// that the subset's centroid stays near the structure's): stage 1 accumulates those k-steps only,
// subtracts the rank-one centring term  C_S(p) C_S(q)^T / A_S  (C_S = sum of the subset's
// coordinates, kept per conformer with G_S^c and the uncentred G_S^u, which scales the error
// bounds: the accumulators hold UNcentred sums), and puts the same bounded polynomial to it.  A
// 16 x 32 unit none of whose 512 pairs may be similar ends there -- half the matrix work; otherwise
// the odd k-steps are accumulated on top and the full test decides as before.  With ~0.04 % of
// similar pairs (the bench) four units in five end after stage 1.
// STAGED: the subset stage only (flagged units go to the unit queue); otherwise the full test only --
// two kernels, so that neither carries the other's registers
// TC: conformers of the column tile: 64; 32 for structures whose 64-column tile does not fit the LDS (214 ... ~370 atoms: lean,
// single-stage launches only) -- two runs of 32 columns per LDS-DMA instruction, one 16 x 32 unit per row tile
template <int NW, bool BITS = true, bool STAGED = false, int TC = 64>
__global__ void __launch_bounds__(NW * 64, FC_F32_WGS)
k_simbits_screen_mfma_f32(const float *__restrict__ Xsf, const double *__restrict__ G, int64_t N,
                          int64_t Npad, int A, double A_thr2, KabschF32Bounds bd, int IB, int64_t rank,
                          int64_t world, uint64_t *__restrict__ bits, int64_t W, uint32_t *__restrict__ cand,
                          unsigned long long *__restrict__ counters, uint64_t *__restrict__ pairq,
                          unsigned long long Q, const uint64_t *__restrict__ item_table,
                          unsigned long long n_items, const float *__restrict__ sub, KabschF32Bounds bd1,
                          float inv_AS, uint64_t *__restrict__ unitq, int part, int gate) {
  // The subset stage is launched in two parts -- every 16th item first (part 1), the other items behind
  // a verdict on that sample (part 2) -- so that an ensemble with dense similarity (most units flagged:
  // the subset stage would be pure overhead) costs only the sample: gate 1 = run iff the verdict said
  // "sparse" (counters[15] == 0), gate 2 = run iff it said "dense" (the single-stage launch over all
  // items, sub == nullptr), gate 0 = always.
  if ((gate == 1 && counters[15] != 0ull) || (gate == 2 && counters[15] == 0ull)) return;
  extern __shared__ double lds_raw[];
  float *__restrict__ lds = reinterpret_cast<float *>(lds_raw);
  static_assert(TC == 64 || (TC == 32 && !BITS && !STAGED), "32-column tiles: lean single-stage launches only");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = (A + 3) >> 2;
  constexpr bool two = STAGED;
  static_assert(!(STAGED && BITS), "the subset stage belongs to the lean kernel");
  const float half_A_thr2 = (float)(0.5 * A_thr2);
  float *__restrict__ ldsG = lds + KS * 12 * TC;  // G/2 as fp32: [TC columns | IB rows]
  float *__restrict__ ldsS = ldsG + TC + IB;      // two: per conformer C_S (3), G_S^c / 2, G_S^u / 2
  // (every block above is an even number of floats: IB is a multiple of 32)
  uint64_t *__restrict__ stageQ = reinterpret_cast<uint64_t *>(ldsS + (two ? (TC + IB) * 5 : 0));
  uint32_t *__restrict__ stageW = reinterpret_cast<uint32_t *>(stageQ + kStagePairsF32);
  unsigned *__restrict__ stageN = reinterpret_cast<unsigned *>(stageW + kStageWordsF32);  // [pairs, words, listed units, next unit | unit list x 32]
  // part 0: item = workgroup; part 1: every 16th item; part 2: the items part 1 left out
  const unsigned long long bx = blockIdx.x;
  const unsigned long long b = part == 0 ? bx : part == 1 ? bx * 16ull : bx + bx / 15ull + 1ull;
  if (b >= n_items) return;
  int64_t jt, lb;
  int it_first = 0, it_last = IB >> 4;
  if (item_table != nullptr) {
    const uint64_t it = item_table[b];
    lb = (int64_t)((it >> 32) & 0x7fffffffull);
    jt = (int64_t)(it & 0x7fffffffull);
    if (it & (1ull << 31)) it_last = IB >> 5;
    if (it & (1ull << 63)) it_first = IB >> 5;
  } else {
    const int64_t NT = Npad / TC;
    jt = (int64_t)(b % (unsigned long long)NT);
    lb = (int64_t)(b / (unsigned long long)NT);
  }
  const int64_t j0 = jt * TC;
  const int64_t i0 = global_block(lb, rank, world) * IB;
  if (i0 >= N) return;               // block-uniform
  if (j0 + TC - 1 <= i0) return;     // nothing above the diagonal in this item

  {  // column tile by LDS-DMA: one 1-KiB instruction per (s, c) run
    const int n_runs = KS * 3;
    if (TC == 64) {
      const int kh_l = lane >> 5, cs_l = (lane >> 3) & 3, k1_l = (lane >> 2) & 1, c4_l = (lane & 3) * 4;
      for (int q = wv; q < n_runs; q += NW) {
        const int sg = q / 3, c = q - sg * 3;
        const int a = sg * 4 + kh_l * 2 + k1_l;
        const float *src = Xsf + (int64_t)(a * 3 + c) * Npad + j0 + cs_l * 16 + c4_l;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + q * 256), 16, 0, 0);
      }
    } else {  // a run = [kh][two sub-tiles][k1][16 columns] = 128 floats: lanes 0..31 the even run, 32..63 the odd one
      const int l32 = lane & 31;
      const int kh_l = l32 >> 4, cs_l = (l32 >> 3) & 1, k1_l = (l32 >> 2) & 1, c4_l = (l32 & 3) * 4;
      for (int q2 = wv; 2 * q2 < n_runs; q2 += NW) {
        const int q = 2 * q2 + (lane >> 5);
        if (q < n_runs) {  // (an odd number of runs: the last instruction's upper half stays off)
          const int sg = q / 3, c = q - sg * 3;
          const int a = sg * 4 + kh_l * 2 + k1_l;
          const float *src = Xsf + (int64_t)(a * 3 + c) * Npad + j0 + cs_l * 16 + c4_l;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)(lds + q2 * 256), 16, 0, 0);
        }
      }
    }
    for (int idx = tid; idx < TC + IB; idx += NW * 64) {
      const int64_t g = idx < TC ? j0 + idx : i0 + (idx - TC);
      ldsG[idx] = g < Npad ? (float)(0.5 * G[g]) : 0.f;
    }
    if (two)
      for (int idx = tid; idx < (TC + IB) * 5; idx += NW * 64) {
        const int c = idx / 5, f = idx - c * 5;
        const int64_t g = c < TC ? j0 + c : i0 + (c - TC);
        ldsS[idx] = g < Npad ? sub[g * 8 + f] : 0.f;
      }
    for (int idx = tid; idx < kStagePairsF32; idx += NW * 64) stageQ[idx] = ~0ull;
    if (tid < kStageWordsF32) stageW[tid] = ~0u;
    if (tid < 4) stageN[tid] = 0u;
    __syncthreads();
  }
  const int kq = lane >> 4, l15 = lane & 15;
  const int boff = (kq >> 1) * (2 * TC) + (kq & 1) * 16 + l15;
  uint16_t *bits16 = reinterpret_cast<uint16_t *>(bits);
  const int n32 = (int)N;

  float a0[3], a1[3], a2[3];
  int pre_key = -1;  // (it << 1 | half) whose first three k-steps a0 / a1 / a2 hold
  auto row_offsets = [&](int it_, unsigned (&vo)[3]) {
    const int64_t ib_ = i0 + (int64_t)it_ * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) vo[c] = (unsigned)((int64_t)(kq * 3 + c) * Npad + ib_ + l15);
  };
  // k-step `sl` of the layout for the lane's row
  auto fetch_a_phys = [&](float (&a)[3], const unsigned (&vo)[3], int sl) {
    const float *__restrict__ xs_s = Xsf + (int64_t)sl * 12 * Npad;  // wave-uniform base
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = xs_s[vo[c]];
  };
  auto unit_exists = [&](int it_, int half_) {
    const int64_t ib_ = i0 + (int64_t)it_ * 16;
    return it_ < it_last && ib_ < N && !(j0 + TC - 1 <= ib_) && !(j0 + (half_ * 2 + 2) * 16 - 1 <= ib_);
  };

  // ---- one 16 x 32 unit.  STAGE1: the even k-steps and the subset verdict only (returns whether any of
  // its 512 pairs may be similar); otherwise all k-steps, the full test, candidates staged / bits stored.
  // next_key: the unit whose first operands are requested behind this one's last MFMA (-1: none).
#define FC_KSTEP(AX, BX, U)                                   \
  if (kk0 + (U) < ke) {                                       \
    mma(AX, BX);                                              \
    fetch_a_phys(AX, voff, kstep(kk0 + (U) + 3 < ke ? kk0 + (U) + 3 : ke - 1)); \
    fetch_b(BX, kstep(kk0 + (U) + 2 < ke ? kk0 + (U) + 2 : ke - 1));            \
  }
  auto run_unit = [&](auto stage1_tag, int it, int half, int next_key, unsigned &nz) -> bool {
    constexpr bool STAGE1 = decltype(stage1_tag)::value;
    const int64_t ib = i0 + (int64_t)it * 16;
    const int64_t lrow0 = lb * IB + (int64_t)it * 16;
    const int cs0 = half * 2;
    const int ke = STAGE1 ? (KS + 1) >> 1 : KS;
    auto kstep = [&](int kk) { return STAGE1 ? 2 * kk : kk; };  // stage 1: the even k-steps of the layout
    unsigned voff[3];
    row_offsets(it, voff);
    f4_t acc[2][9];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 9; ++e) acc[t][e] = f4_t{0.f, 0.f, 0.f, 0.f};
    const float *__restrict__ lb0 = lds + cs0 * 32 + boff;
    float b0[2][3], b1[2][3];
    auto fetch_b = [&](float (&bb)[2][3], int sl) {
      const float *__restrict__ lb_s = lb0 + sl * (12 * TC);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) bb[t][c] = lb_s[c * (4 * TC) + t * 32];
    };
    auto mma = [&](const float (&a)[3], const float (&bb)[2][3]) {
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[t][x * 3 + y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], bb[t][y], acc[t][x * 3 + y], 0, 0, 0);
    };
    if (pre_key != ((it << 1) | half)) {  // first unit of the wave, or nobody predicted this one
      fetch_a_phys(a0, voff, kstep(0));
      fetch_a_phys(a1, voff, kstep(1 < ke ? 1 : ke - 1));
      fetch_a_phys(a2, voff, kstep(2 < ke ? 2 : ke - 1));
    }
    fetch_b(b0, kstep(0));
    fetch_b(b1, kstep(1 < ke ? 1 : ke - 1));
#ifdef FC_F32_ABLATE_K  // timing experiment only (results are wrong): one round of k-steps instead of all
    for (int kk0 = 0; kk0 < 1; kk0 += 6) {
#else
    for (int kk0 = 0; kk0 < ke; kk0 += 6) {
#endif
      FC_KSTEP(a0, b0, 0)
      FC_KSTEP(a1, b1, 1)
      FC_KSTEP(a2, b0, 2)
      FC_KSTEP(a0, b1, 3)
      FC_KSTEP(a1, b0, 4)
      FC_KSTEP(a2, b1, 5)
    }
    // request the next unit's first three k-steps now; they land during the epilogue
    if (next_key >= 0) {
      unsigned vn[3];
      row_offsets(next_key >> 1, vn);
      fetch_a_phys(a0, vn, kstep(0));
      fetch_a_phys(a1, vn, kstep(1 < ke ? 1 : ke - 1));
      fetch_a_phys(a2, vn, kstep(2 < ke ? 2 : ke - 1));
    }
    pre_key = next_key;
    const int ib32 = (int)ib;
    if (STAGE1) {
      // the subset's centred covariance = accumulated sums - C_S(p) C_S(q)^T / A_S
      bool any1 = false;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int cs = cs0 + t;
        const int j = (int)j0 + cs * 16 + l15;
        const float *__restrict__ sq = ldsS + (cs * 16 + l15) * 5;
        const float qx = sq[0], qy = sq[1], qz = sq[2], gqc = sq[3], gqu = sq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ib32 + 4 * kq + r;
          const float *__restrict__ sp = ldsS + (TC + it * 16 + 4 * kq + r) * 5;
          const float px = -inv_AS * sp[0], py = -inv_AS * sp[1], pz = -inv_AS * sp[2];
          float B9[9];
          B9[0] = fmaf(px, qx, acc[t][0][r]); B9[1] = fmaf(px, qy, acc[t][1][r]); B9[2] = fmaf(px, qz, acc[t][2][r]);
          B9[3] = fmaf(py, qx, acc[t][3][r]); B9[4] = fmaf(py, qy, acc[t][4][r]); B9[5] = fmaf(py, qz, acc[t][5][r]);
          B9[6] = fmaf(pz, qx, acc[t][6][r]); B9[7] = fmaf(pz, qy, acc[t][7][r]); B9[8] = fmaf(pz, qz, acc[t][8][r]);
          bool may = kabsch_may_be_below_f32(B9, sp[3] + gqc, half_A_thr2, bd1, sp[4] + gqu);
          may = may && (j > i) && (j < n32) && (i < n32);
          any1 = any1 || may;
        }
      }
      return __any(any1);
    }
    // full epilogue: lane owns pairs (ib + 4 kq + r, j0 + cs*16 + l15), r = 0..3
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int cs = cs0 + t;
      const int j = (int)j0 + cs * 16 + l15;
      const float Gq = ldsG[cs * 16 + l15];
      uint64_t mr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib32 + 4 * kq + r;
        const float Gp = ldsG[TC + it * 16 + 4 * kq + r];
        float B9[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) B9[e] = acc[t][e][r];
#ifdef FC_F32_ABLATE_POLY  // timing experiment only (results are wrong): the K loop without the polynomial
        bool may = (B9[0] + B9[4] + B9[8]) * B9[1] * B9[2] * B9[3] * B9[5] * B9[6] * B9[7] == 12345.678f;
#else
        bool may = kabsch_may_be_below_f32(B9, Gp + Gq, half_A_thr2, bd, Gp + Gq);
#endif
        may = may && (j > i) && (j < n32) && (i < n32);
        mr[r] = __ballot(may);
        stage_pairs<kStagePairsF32>(mr[r], may, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
      }
      // bit (16 kq' + c) of mr[r] belongs to row 4 kq' + r: lane l < 16 writes the piece of
      // row l = 4 (l >> 2) + (l & 3), one store per sub-tile instead of one per register
      if (BITS && lane < 16 && ib32 + lane < n32) {
        const int rr = lane & 3;
        const uint64_t mine = rr == 0 ? mr[0] : rr == 1 ? mr[1] : rr == 2 ? mr[2] : mr[3];
        const unsigned piece = (unsigned)((mine >> (16 * (lane >> 2))) & 0xffffull);
        bits16[((lrow0 + lane) * W + jt) * 4 + cs] = (uint16_t)piece;
        nz |= piece;
      }
    }
    return true;
  };
#undef FC_KSTEP

  // ---- the wave's units in order: row tiles it = it_first + wv, + NW, ...; two 32-column halves each.
  // Lean with the subset stage: phase 1 runs stage 1 on all of them and lists the units that may
  // hold a similar pair in LDS; phase 2 hands those to the waves of the workgroup as they become free.
  unsigned *__restrict__ unit_list = stageN + 4;  // [32]
  for (int it = it_first + wv; it < it_last; it += NW) {
    const int64_t ib = i0 + (int64_t)it * 16;
    if (ib >= N) break;
    if (j0 + TC - 1 <= ib) break;
    const int64_t lrow0 = lb * IB + (int64_t)it * 16;
    unsigned nz = 0;  // lanes 0..15: OR of the 16-bit pieces written for row ib + lane
#pragma unroll 1
    for (int half = 0; half < TC / 32; ++half) {
      const int cs0 = half * 2;
      if (j0 + (cs0 + 2) * 16 - 1 <= ib) {  // both sub-tiles at or below the diagonal
        if (BITS && lane < 16 && ib + lane < N) {
          bits16[((lrow0 + lane) * W + jt) * 4 + cs0] = 0;
          bits16[((lrow0 + lane) * W + jt) * 4 + cs0 + 1] = 0;
        }
        continue;
      }
      const int nk = (TC == 64 && half == 0) ? ((it << 1) | 1)
                                               : (unit_exists(it + NW, 0) ? ((it + NW) << 1) : ((TC == 64 && unit_exists(it + NW, 1)) ? (((it + NW) << 1) | 1) : -1));
      if constexpr (STAGED) {
        const bool any = run_unit(std::true_type{}, it, half, nk, nz);
        if (any && lane == 0) unit_list[atomicAdd(stageN + 2, 1u)] = (unsigned)((it << 1) | half);
      } else {
        (void)run_unit(std::false_type{}, it, half, nk, nz);
      }
    }
    if (BITS) {  // queue the non-empty words of this row tile for the exact refine
      const bool has = lane < 16 && nz != 0;
      const uint64_t mw = __ballot(has);
      if (mw != 0) {  // wave-uniform
        const unsigned n = (unsigned)__popcll(mw);
        const uint32_t word = (uint32_t)((lrow0 + lane) * W + jt);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(stageN + 1, n);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned rank_in = (unsigned)__popcll(mw & ((1ull << lane) - 1ull));
        if (base + n <= (unsigned)kStageWordsF32) {
          if (has) stageW[base + rank_in] = word;
        } else {  // no room: straight to the global queue
          unsigned long long gbase = 0;
          if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)n);
          gbase = __shfl(gbase, 0);
          if (has) cand[gbase + rank_in] = word;
        }
      }
    }
  }
  // publish what the workgroup staged: one global atomic per queue
  __syncthreads();
  if (wv == 0) {
    if (two && stageN[2]) {
      // the units that may hold a similar pair go to the global unit queue (k_screen_units_f32 puts
      // them through the full test): (first row << 32) | first column, one reservation per workgroup
      const unsigned n_list = stageN[2];
      unsigned long long ubase = 0;
      if (lane == 0) ubase = atomicAdd(&counters[13], (unsigned long long)n_list);
      ubase = __shfl(ubase, 0);
      if ((unsigned)lane < n_list) {
        const unsigned key = unit_list[lane];
        unitq[ubase + lane] = ((uint64_t)(i0 + (int64_t)(key >> 1) * 16) << 32) | (uint64_t)(j0 + (int64_t)(key & 1u) * 32);
      }
    }
    const int used_q = min((int)stageN[0], kStagePairsF32), used_w = min((int)stageN[1], kStageWordsF32);
    for (int c0 = 0; c0 < used_q; c0 += 64) {
      const uint64_t e = c0 + lane < kStagePairsF32 ? stageQ[c0 + lane] : ~0ull;
      const bool valid = e != ~0ull;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[6], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) {
          const unsigned long long slot = gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull));
          if (slot < Q) pairq[slot] = e;
        }
      }
    }
    for (int c0 = 0; c0 < used_w; c0 += 64) {
      const uint32_t wq = c0 + lane < kStageWordsF32 ? stageW[c0 + lane] : ~0u;
      const bool valid = wq != ~0u;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) cand[gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull))] = wq;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_screen_units_f32 -- the full test for the units the subset stage could not rule out (~one in five
// on the bench ensemble): one wavefront per queued 16 x 32 unit, drawn from a device counter, both
// operands streamed from L2 three k-steps ahead (no column tile in LDS: the units are scattered over
// the whole matrix), all k-steps, the bounded polynomial, candidates appended to the pair queue.
// Steps aside (counters[15]) when the sample of the subset stage found similarity dense: the
// single-stage tiled kernel launched behind it then does the whole launch.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 3)
k_screen_units_f32(const float *__restrict__ Xsf, const double *__restrict__ G, int64_t N, int64_t Npad, int A,
                   double A_thr2, KabschF32Bounds bd, const uint64_t *__restrict__ unitq,
                   unsigned long long *__restrict__ counters, uint64_t *__restrict__ pairq, unsigned long long Q) {
  if (counters[15] != 0ull) return;  // dense similarity: the single-stage launch behind this one does everything
  const unsigned long long n_units = counters[13];
  // candidates are staged per workgroup and published with ONE global atomic at the end: nearly every
  // queued unit holds a candidate, and one atomic-with-return per unit on the same word (~88 per
  // microsecond chip-wide) would cost more than the arithmetic (measured: 0.23 ms for 2*10^4 units)
  __shared__ uint64_t stageQ[kStagePairsF32];
  __shared__ unsigned stageN[2];
  for (int idx = threadIdx.x; idx < kStagePairsF32; idx += 256) stageQ[idx] = ~0ull;
  if (threadIdx.x < 2) stageN[threadIdx.x] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int kq = lane >> 4, l15 = lane & 15;
  const int KS = (A + 3) >> 2;
  const float half_A_thr2 = (float)(0.5 * A_thr2);
  const int n32 = (int)N;
  // units dealt to the wavefronts round-robin: they cost the same, and one shared dequeue counter
  // saturates at ~88 atomics per microsecond (measured here: 0.24 ms for 2*10^4 units)
  const unsigned long long wave0 = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned long long n_waves = (unsigned long long)gridDim.x * 4;
  for (unsigned long long u = wave0; u < n_units; u += n_waves) {
    const uint64_t key = unitq[u];
    const int64_t ib = (int64_t)(key >> 32), jc = (int64_t)(key & 0xffffffffull);
    unsigned va[3], vb[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      va[c] = (unsigned)((int64_t)(kq * 3 + c) * Npad + ib + l15);
      vb[0][c] = (unsigned)((int64_t)(kq * 3 + c) * Npad + jc + l15);
      vb[1][c] = vb[0][c] + 16u;
    }
    f4_t acc[2][9];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 9; ++e) acc[t][e] = f4_t{0.f, 0.f, 0.f, 0.f};
    float a0[3], a1[3], a2[3], b0[2][3], b1[2][3], b2[2][3];
    auto fetch = [&](float (&a)[3], float (&bb)[2][3], int sx) {
      const float *__restrict__ xs_s = Xsf + (int64_t)(sx < KS ? sx : KS - 1) * 12 * Npad;  // wave-uniform base
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        a[c] = xs_s[va[c]];
        bb[0][c] = xs_s[vb[0][c]];
        bb[1][c] = xs_s[vb[1][c]];
      }
    };
    auto mma = [&](const float (&a)[3], const float (&bb)[2][3]) {
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[t][x * 3 + y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], bb[t][y], acc[t][x * 3 + y], 0, 0, 0);
    };
    // (scheduling barriers: the loads must be ISSUED in the order they are consumed -- vmcnt counts in issue
    // order, and the scheduler had put the first k-step's loads last, i.e. a full drain in front of every MFMA group)
    fetch(a0, b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(a1, b1, 1);
    __builtin_amdgcn_sched_barrier(0);
    fetch(a2, b2, 2);
    __builtin_amdgcn_sched_barrier(0);
    // whole rounds of three k-steps without a branch inside (with a guard per k-step the compiler merges
    // the "skipped" paths into its wait counts and drains the load counter in front of every MFMA group:
    // 111 us for this kernel where the matrix work is 58 us); loads past the last k-step re-read it
    int s0 = 0;
    for (; s0 + 3 <= KS; s0 += 3) {
      mma(a0, b0);
      fetch(a0, b0, s0 + 3);
      mma(a1, b1);
      fetch(a1, b1, s0 + 4);
      mma(a2, b2);
      fetch(a2, b2, s0 + 5);
    }
    if (s0 < KS) mma(a0, b0);
    if (s0 + 1 < KS) mma(a1, b1);
    const int ib32 = (int)ib;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = (int)jc + t * 16 + l15;
      const float Gq = (float)(0.5 * G[j]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib32 + 4 * kq + r;
        const float Gp = (float)(0.5 * G[i]);
        float B9[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) B9[e] = acc[t][e][r];
        bool may = kabsch_may_be_below_f32(B9, Gp + Gq, half_A_thr2, bd, Gp + Gq);
        may = may && (j > i) && (j < n32) && (i < n32);
        stage_pairs<kStagePairsF32>(__ballot(may), may, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
      }
    }
  }
  __syncthreads();
  if ((threadIdx.x >> 6) == 0) {
    const int used_q = min((int)stageN[0], kStagePairsF32);
    for (int c0 = 0; c0 < used_q; c0 += 64) {
      const uint64_t e = c0 + lane < kStagePairsF32 ? stageQ[c0 + lane] : ~0ull;
      const bool valid = e != ~0ull;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[6], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) {
          const unsigned long long slot = gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull));
          if (slot < Q) pairq[slot] = e;
        }
      }
    }
  }
}

// verdict on the sample of the subset stage (every 16th item): more than max_sample_units of its units
// flagged = dense similarity -> counters[15] = 1, the sample's queue is discarded
__global__ void k_screen_density_verdict(unsigned long long *__restrict__ counters, unsigned long long max_sample_units) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool dense = counters[13] > max_sample_units;
    counters[15] = dense ? 1ull : 0ull;
    if (dense) counters[13] = 0ull;
  }
}

// ---------------------------------------------------------------------------
// k_simbits_screen_mfma_h2 -- the single-precision screen with the covariance on the HALF-precision
// matrix pipe (v_mfma_f32_16x16x32_f16: 16x the rate of the f32 instruction) at single-precision
// accuracy: every coordinate is split into two halfs, x * 2^e = hi + lo (+ at most 2^-22 |x|), and
//     B = sum hi hi^T + sum hi lo^T + sum lo hi^T        (lo lo^T <= 2^-22 s is left to the bound)
// costs three f16 products instead of one f32 product at a sixteenth of the rate each: 54 instructions
// of 16 cycles per 16 x 16 pairs and 64 atoms against 117 of 32 cycles for the f32 kernel (50 atoms).
// What is left is the polynomial epilogue, which is the same (kabsch_may_be_below_f32 on the fp32
// accumulators, bounds: kabsch_h2_bounds), as are the tiling, the item table, the staging
// queues and the outputs.  The screen stays a FILTER: what it lets through is decided by the fp64 refine.
//   * operands: Xh, halfs in the instruction's own operand layout -- run q = ((s*2 + part)*3 + c)*4 + kq
//     holds for every conformer n the 8 halfs of atoms s*32 + kq*8 + 0..7 (part 0 = hi, 1 = lo) of
//     coordinate c: [q][n][8]; lane (kq, l15) of a row or column sub-tile reads ONE 16-byte piece per
//     (s, part, c), 16 lanes 256 contiguous bytes;
//   * column tile: 24 KS2 runs of 64 x 16 B by LDS-DMA (one 1-KiB instruction each), read back with
//     conflict-free ds_read_b128;
//   * row operands: all 6 KS2 pieces of a 16-row tile in registers, loaded once per row tile (its four
//     16-column sub-tiles use them, one at a time: 9 accumulators), the next row tile's requested behind the
//     last MFMA of this one;
//   * order of accumulation: the cross terms of ALL k-steps first (their sum is <= 2^-10 s, what the
//     matrix pipe does to it is far below the bound), then the hi hi^T terms -- KS2 instructions whose
//     accumulator is of the order of the result.  Charged per instruction: 66 u (|C| + sum |a b|), a bound that holds
//     for ANY order of the 33 additions (kabsch_h2_entry_bound, fc_kabsch_math.h); what the instruction is observed to do
//     (tools/ubench_mfma_f16_numerics.hip: products and C aligned to the largest, 1 to 3 bits kept below that one's
//     last place, the rest truncated, one rounding: <= 5.4 u seen) is only a tripwire at 18 u (fc_h2_check.hip);
//   * the accumulators hold 2^2e B: G and the threshold are scaled by 2^2e (exact), the polynomial test
//     is homogeneous.
// KS2 = k-steps of 32 atoms (template: everything unrolled, operand arrays in registers).
// ---------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
// k-steps of 32 atoms the split-half kernel is built for: 6 = 192 atoms, a 144 KB column tile, one workgroup per CU
// (round 4; 4 = 128 atoms until then, larger structures on the fp32 matrix pipe: 1.7-2.3 ms per 7.2e7 pairs at 129-200
// atoms against 0.74 at 128)
constexpr int64_t kH2MaxKS2 = 6;
#ifndef FC_H2_WGS
#define FC_H2_WGS 3
#endif
// (Not kept: running the polynomials of sub-tile cs inside the matrix work of sub-tile cs + 1 -- two accumulator sets,
// four vector instructions behind every MFMA by sched_group_barrier.  In a reduced kernel with exactly this
// instruction mix the interleaved order takes 801 ns per sub-tile and SIMD at two waves per SIMD against 741 ns for
// the phased order, and 667 against 629 ns at four: on gfx950 the vector work of a wave does not hide under its own
// f16 MFMAs any better than under a sibling wave's; more resident waves is what helps.)
// TC: conformers of the column tile.  64; 32 for structures of 193 ... 416 atoms (7 ... 13 k-steps; lean launches only:
// the bit-matrix layout is one 64-column word per tile) -- 12 KB of tile per k-step instead of 24, the row operands of
// all k-steps still in registers (24 per k-step, one wavefront per SIMD has 512)
template <int KS2, bool BITS, int TC = 64>
__global__ void __launch_bounds__(256, (KS2 <= 2 ? FC_H2_WGS : KS2 <= 3 ? 2 : 1))  // (four k-steps and more: the column tile leaves one workgroup per CU anyway)
k_simbits_screen_mfma_h2(const h8_t *__restrict__ Xh, const double *__restrict__ G, int64_t N, int64_t Npad,
                         float half_A_thr2, float tiny_floor, float scale2, KabschF32Bounds bd, int IB, int64_t rank,
                         int64_t world, uint64_t *__restrict__ bits, int64_t W, uint32_t *__restrict__ cand,
                         unsigned long long *__restrict__ counters, uint64_t *__restrict__ pairq,
                         unsigned long long Q, const uint64_t *__restrict__ item_table, unsigned long long n_items) {
  extern __shared__ double lds_raw[];
  static_assert(TC == 64 || (TC == 32 && !BITS), "32-column tiles: lean launches only");
  constexpr int NW = 4;
  constexpr int n_runs = KS2 * 24;
  h8_t *__restrict__ lds8 = reinterpret_cast<h8_t *>(lds_raw);  // [run][64 columns]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *__restrict__ ldsG = reinterpret_cast<float *>(lds8 + n_runs * TC);  // 2^2e G/2 as fp32: [TC columns | IB rows]
  uint64_t *__restrict__ stageQ = reinterpret_cast<uint64_t *>(ldsG + TC + IB);  // (IB is a multiple of 32)
  uint32_t *__restrict__ stageW = reinterpret_cast<uint32_t *>(stageQ + kStagePairsF32);
  unsigned *__restrict__ stageN = reinterpret_cast<unsigned *>(stageW + kStageWordsF32);  // [-, words, -, - | pairs staged by wavefront 0..3 | queue base (64 bit)]
  const unsigned long long b = blockIdx.x;
  if (b >= n_items) return;
#ifdef FC_H2_TIMELINE  // tuning build (lean launches only: `bits` carries the stamp buffer): start / filled / end / where
  unsigned long long *tl = BITS ? nullptr : reinterpret_cast<unsigned long long *>(bits);
  if (tl && tid == 0) {
    tl[(size_t)b * 4] = wall_clock64();
    tl[(size_t)b * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                            ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_ID, XCC_ID
  }
#endif
  int64_t jt, lb;
  int it_first = 0, it_last = IB >> 4;
  if (item_table != nullptr) {
    const uint64_t it = item_table[b];
    lb = (int64_t)((it >> 32) & 0x7fffffffull);
    jt = (int64_t)(it & 0x7fffffffull);
    if (it & (1ull << 31)) it_last = IB >> 5;
    if (it & (1ull << 63)) it_first = IB >> 5;
  } else {
    const int64_t NT = Npad / TC;
    jt = (int64_t)(b % (unsigned long long)NT);
    lb = (int64_t)(b / (unsigned long long)NT);
  }
  const int64_t j0 = jt * TC;
  const int64_t i0 = global_block(lb, rank, world) * IB;
  if (i0 >= N) return;            // block-uniform
  if (j0 + TC - 1 <= i0) return;  // nothing above the diagonal in this item

  const int kq = lane >> 4, l15 = lane & 15;
  uint16_t *bits16 = reinterpret_cast<uint16_t *>(bits);
  const int n32 = (int)N;

  // row operands of one 16-row tile: [k-step][hi | lo][coordinate]
  h8_t ra[KS2][2][3];
  auto fetch_rows = [&](int it_) {
    const unsigned vo = (unsigned)((int64_t)kq * Npad + i0 + (int64_t)it_ * 16 + l15);
#pragma unroll
    for (int s = 0; s < KS2; ++s)
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const h8_t *__restrict__ base = Xh + (int64_t)(((s * 2 + part) * 3 + c) * 4) * Npad;  // wave-uniform
          ra[s][part][c] = base[vo];
        }
  };
  auto tile_exists = [&](int it_) {
    const int64_t ib_ = i0 + (int64_t)it_ * 16;
    return it_ < it_last && ib_ < N && !(j0 + TC - 1 <= ib_);
  };
  // the first row tile's operands travel while the column tile is filled
  int it = it_first + wv;
  if (tile_exists(it)) fetch_rows(it);
  {  // column tile by LDS-DMA: one 1-KiB instruction per run
    if (TC == 64) {
      for (int q = wv; q < n_runs; q += NW) {
        const h8_t *src = Xh + ((int64_t)q * Npad + j0 + lane);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds8 + q * TC), 16, 0, 0);
      }
    } else {  // two runs of 32 columns per instruction: lanes 0..31 the even run, 32..63 the odd one
      for (int q2 = wv; q2 < n_runs / 2; q2 += NW) {
        const h8_t *src = Xh + ((int64_t)(2 * q2 + (lane >> 5)) * Npad + j0 + (lane & 31));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds8 + q2 * 64), 16, 0, 0);
      }
    }
    for (int idx = tid; idx < TC + IB; idx += NW * 64) {
      const int64_t g = idx < TC ? j0 + idx : i0 + (idx - TC);
      ldsG[idx] = g < Npad ? (float)(0.5 * G[g]) * scale2 : 0.f;
    }
    if (tid < kStageWordsF32) stageW[tid] = ~0u;
    if (tid < 2) stageN[tid] = 0u;
    __syncthreads();
#ifdef FC_H2_TIMELINE
    if (tl && tid == 0) tl[(size_t)b * 4 + 1] = wall_clock64();
#endif
  }
  const h8_t *__restrict__ lcol = lds8 + kq * TC + l15;  // + run * (4 TC) + sub-tile * 16
  // candidates: a quarter of the staging area per wavefront, counted in a scalar register
  constexpr int WCAP = kStagePairsF32 / NW;
  uint64_t *__restrict__ sq_wave = stageQ + wv * WCAP;
  int n_staged = 0;
  for (; tile_exists(it); it += NW) {
    const int64_t ib = i0 + (int64_t)it * 16;
    const int64_t lrow0 = lb * IB + (int64_t)it * 16;
    const int ib32 = (int)ib;
    unsigned nz = 0;  // lanes 0..15: OR of the 16-bit pieces written for row ib + lane
    // one 16 x 16 sub-tile at a time: the row operands stay in registers, so a wider unit would share nothing
#pragma unroll 1
    for (int cs = 0; cs < TC / 16; ++cs) {
      if (j0 + (cs + 1) * 16 - 1 <= ib) {  // at or below the diagonal (never the last sub-tile)
        if (BITS && lane < 16 && ib + lane < N) bits16[((lrow0 + lane) * W + jt) * 4 + cs] = 0;
        continue;
      }
      f4_t acc[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) acc[e] = f4_t{0.f, 0.f, 0.f, 0.f};
      const h8_t *__restrict__ lc = lcol + cs * 16;
      // cross terms of every k-step first
#ifdef FC_H2_ABLATE_K  // timing experiment only (results are wrong): one k-step of cross terms instead of everything
      constexpr int KSX = 1;
#else
      constexpr int KSX = KS2;
#endif
#pragma unroll
      for (int s = 0; s < KSX; ++s)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
          const h8_t bh = lc[(((s * 2 + 0) * 3 + y) * 4) * TC];
          const h8_t bl = lc[(((s * 2 + 1) * 3 + y) * 4) * TC];
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            acc[x * 3 + y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][0][x], bl, acc[x * 3 + y], 0, 0, 0);
            acc[x * 3 + y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][1][x], bh, acc[x * 3 + y], 0, 0, 0);
          }
        }
      // then hi x hi
#ifndef FC_H2_ABLATE_K
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
          const h8_t bh = lc[(((s * 2 + 0) * 3 + y) * 4) * TC];
#pragma unroll
          for (int x = 0; x < 3; ++x)
            acc[x * 3 + y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s][0][x], bh, acc[x * 3 + y], 0, 0, 0);
        }
#endif
      // the next row tile's operands are requested now: they land during the epilogue
#ifndef FC_H2_ABLATE_ROWS  // timing experiment only (results are wrong): every row tile computed from the first one's operands
      if (cs == TC / 16 - 1 && tile_exists(it + NW)) fetch_rows(it + NW);
#endif
      // epilogue: lane owns pairs (ib + 4 kq + r, j0 + cs*16 + l15), r = 0..3.  The four polynomials in one
      // straight line; what is rare (a pair for the three-test form, a sub-tile on the diagonal or at the end
      // of the ensemble, a candidate to stage) sits behind ONE wave-uniform branch per sub-tile each
      const int j = (int)j0 + cs * 16 + l15;
      const float Gq = ldsG[cs * 16 + l15];
      uint64_t mr[4], rdm[4];  // per r: lanes whose pair may be similar / lanes that need the three-test form
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float Gp = ldsG[TC + it * 16 + 4 * kq + r];
        float B9[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) B9[e] = acc[e][r];
#ifdef FC_H2_ABLATE_POLY  // timing experiment only (results are wrong): the K loop without the polynomial
        mr[r] = __builtin_amdgcn_ballot_w64((B9[0] + B9[4] + B9[8]) * B9[1] * B9[2] * B9[3] * B9[5] * B9[6] * B9[7] ==
                                            12345.678f * (Gp + Gq));
        rdm[r] = 0;
#else
        mr[r] = kabsch_may_be_below_f32_2t_wave(B9, Gp + Gq, half_A_thr2, bd, tiny_floor, rdm[r]);
#endif
      }
#ifndef FC_ABLATE_REDO  // timing experiment only
      if ((rdm[0] | rdm[1] | rdm[2] | rdm[3]) != 0ull) {  // nearly collinear structures only
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float Gp = ldsG[TC + it * 16 + 4 * kq + r];
          float B9[9];
#pragma unroll
          for (int e = 0; e < 9; ++e) B9[e] = acc[e][r];
          const uint64_t m3 =
              __builtin_amdgcn_ballot_w64(kabsch_may_be_below_f32(B9, Gp + Gq, half_A_thr2, bd, Gp + Gq, tiny_floor));
          mr[r] = (mr[r] & ~rdm[r]) | (m3 & rdm[r]);
        }
      }
#endif
      // (strictly above the diagonal and inside the ensemble -- nearly every sub-tile -- no index test)
      const bool interior = (int)j0 + cs * 16 > ib32 + 15 && (int)j0 + cs * 16 + 15 < n32 && ib32 + 15 < n32;
      if (!interior) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ib32 + 4 * kq + r;
          mr[r] &= __builtin_amdgcn_ballot_w64((j > i) && (j < n32) && (i < n32));
        }
      }
      if ((mr[0] | mr[1] | mr[2] | mr[3]) != 0ull) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage_pairs_wave<WCAP>(mr[r], ((mr[r] >> lane) & 1ull) != 0ull, (unsigned)(ib32 + 4 * kq + r), (unsigned)j, sq_wave,
                                 n_staged, pairq, Q, counters, lane);
      }
      if (BITS && lane < 16 && ib32 + lane < n32) {
        const int rr = lane & 3;
        const uint64_t mine = rr == 0 ? mr[0] : rr == 1 ? mr[1] : rr == 2 ? mr[2] : mr[3];
        const unsigned piece = (unsigned)((mine >> (16 * (lane >> 2))) & 0xffffull);
        bits16[((lrow0 + lane) * W + jt) * 4 + cs] = (uint16_t)piece;
        nz |= piece;
      }
    }
    if (BITS) {  // queue the non-empty words of this row tile for the exact refine
      const bool has = lane < 16 && nz != 0;
      const uint64_t mw = __ballot(has);
      if (mw != 0) {  // wave-uniform
        const unsigned n = (unsigned)__popcll(mw);
        const uint32_t word = (uint32_t)((lrow0 + lane) * W + jt);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(stageN + 1, n);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned rank_in = (unsigned)__popcll(mw & ((1ull << lane) - 1ull));
        if (base + n <= (unsigned)kStageWordsF32) {
          if (has) stageW[base + rank_in] = word;
        } else {  // no room: straight to the global queue
          unsigned long long gbase = 0;
          if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)n);
          gbase = __shfl(gbase, 0);
          if (has) cand[gbase + rank_in] = word;
        }
      }
    }
  }
  // publish what the workgroup staged.  Pairs: ONE global atomic for the four wavefronts' regions, every wavefront copies
  // its own (round 4: 945 025 candidates of the continuous-RMSD ensemble cost the screen 142 of 302 us while wave 0
  // flushed one shared region 64 entries and one returning atomic at a time behind an LDS atomic per ballot;
  // ~12 ns per atomic on one address at the L2, the same bound the refine kernels met)
  if (lane == 0) stageN[4 + wv] = (unsigned)n_staged;
  __syncthreads();
  {
    const unsigned c0 = stageN[4], c1 = stageN[5], c2 = stageN[6], c3 = stageN[7];
    const unsigned total = c0 + c1 + c2 + c3;
    if (total != 0u) {  // block-uniform
      unsigned long long *gslot = reinterpret_cast<unsigned long long *>(stageN + 8);
      if (tid == 0) *gslot = atomicAdd(&counters[6], (unsigned long long)total);
      __syncthreads();
      const unsigned long long gb = *gslot + (wv == 0 ? 0u : wv == 1 ? c0 : wv == 2 ? c0 + c1 : c0 + c1 + c2);
      for (int idx = lane; idx < n_staged; idx += 64) {
        const unsigned long long slot = gb + (unsigned long long)idx;
        if (slot < Q) pairq[slot] = sq_wave[idx];
      }
    }
  }
#ifdef FC_H2_TIMELINE
  if (tl && tid == 0) tl[(size_t)b * 4 + 2] = wall_clock64();
#endif
  if (wv == 0) {
    const int used_w = min((int)stageN[1], kStageWordsF32);
    for (int c0 = 0; c0 < used_w; c0 += 64) {
      const uint32_t wq = c0 + lane < kStageWordsF32 ? stageW[c0 + lane] : ~0u;
      const bool valid = wq != ~0u;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) cand[gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull))] = wq;
      }
    }
  }
}

// (Not kept: the same screen as RESIDENT workgroups -- grid = 3 per CU walking the item table with a stride, every
// item taken as two 32-column halves so that the next half arrives by LDS-DMA in a second 24 KB buffer while this
// one is computed: no wait for the column tile (16 % of a workgroup's 17 us, tools/attic/h2_timeline_probe.py), no 1.4 us
// between two workgroups on a CU slot.  Same results; 0.25 ms per launch against 0.177 ms (0.22 ms with 2 or 4
// workgroups per CU): statically dealt items leave the slower CUs behind, the four waves meet at a barrier per
// half, and the loop-carried state costs 68 spilled SGPRs and 40 B of scratch.  The hardware's own dispatch of
// 6 753 short workgroups balances better than that.)
// Xs (fp64, [(a*3+c)*Npad + n], A4 atoms) -> Xh: the split-half operand layout of the kernel above.
// One thread per (run-without-part, conformer): 8 atoms of one coordinate.  x * scale (a power of two:
// exact) = hi + lo + d with hi = half(x scale), lo = half(x scale - hi) (the difference is exact in fp64):
// |d| <= 2^-22 (1 + 2^-12) |x scale| or, where lo is subnormal, <= 2^-25 (conversions through fp32
// round twice: covered by the 2^-12).
__global__ void __launch_bounds__(256)
k_f64_to_h2(const double *__restrict__ Xs, int64_t Npad, int A4, int KS2, double scale, h8_t *__restrict__ Xh) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;  // (s*3 + c)*4 + kq
  if (n >= Npad) return;
  const int kq = r & 3, c = (r >> 2) % 3, s = (r >> 2) / 3;
  h8_t hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int a = s * 32 + kq * 8 + j;
    const double x = a < A4 ? Xs[(int64_t)(a * 3 + c) * Npad + n] * scale : 0.0;
    const _Float16 h = (_Float16)(float)x;
    hi[j] = h;
    lo[j] = (_Float16)(float)(x - (double)h);
  }
  Xh[(int64_t)(((s * 2 + 0) * 3 + c) * 4 + kq) * Npad + n] = hi;
  Xh[(int64_t)(((s * 2 + 1) * 3 + c) * 4 + kq) * Npad + n] = lo;
}

// per-conformer statistics of the stage-1 atom subset (the atoms of the even k-steps of the prepared
// layout): sub[n*8 + 0..2] = sum of the subset's coordinates, [3] = G_S^c / 2 (centred on the subset's
// centroid), [4] = G_S^u / 2 (as stored); fp64 arithmetic, fp32 results
__global__ void __launch_bounds__(256)
k_subset_stats(const double *__restrict__ Xs, int64_t Npad, int A, float *__restrict__ sub) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= Npad) return;
  double cx = 0.0, cy = 0.0, cz = 0.0, g = 0.0;
  int cnt = 0;
  for (int a = 0; a < A; ++a) {
    if ((a >> 2) & 1) continue;
    const double x = Xs[(int64_t)(a * 3) * Npad + n], y = Xs[(int64_t)(a * 3 + 1) * Npad + n],
                 z = Xs[(int64_t)(a * 3 + 2) * Npad + n];
    cx += x; cy += y; cz += z;
    g += x * x + y * y + z * z;
    ++cnt;
  }
  const double gc = g - (cx * cx + cy * cy + cz * cz) / (double)(cnt > 0 ? cnt : 1);
  sub[n * 8 + 0] = (float)cx;
  sub[n * 8 + 1] = (float)cy;
  sub[n * 8 + 2] = (float)cz;
  sub[n * 8 + 3] = (float)(0.5 * (gc > 0.0 ? gc : 0.0));
  sub[n * 8 + 4] = (float)(0.5 * g);
  sub[n * 8 + 5] = sub[n * 8 + 6] = sub[n * 8 + 7] = 0.f;
}

// ---------------------------------------------------------------------------
// k_screen_verdict -- was the single-precision screen worth it?  Its undecidable band turns
// dissimilar pairs into candidates (queue traffic in the screen, one atom pass in the refine); a
// second, fp64 screen costs about what 1.5 % of the owned pairs cost as candidates (50 atoms).  One workgroup: when the fp32 screen queued
// more than `max_false` candidates, 256 of them (spread over the queue) are put through the fp64
// polynomial on their exact covariance; if the candidates that fail it, scaled to the whole
// queue, exceed `max_false`, counters[11] := 1 and the queues are reset -- the gated fp64 screen
// behind this kernel then redoes the launch (optimistic: counters[12] := 1 instead and no such launch follows; the
// pair ladder declines and the caller redoes the prune).  Otherwise counters[11] := 0 and the fp64 screen's
// workgroups return at once.  Results do not depend on the verdict (both screens only ever add
// candidates); time is bounded by fp32 screen + fp64 screen whatever the data look like.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_screen_verdict(const double *__restrict__ Xa, const double *__restrict__ G, int A, double A_thr2,
                 const uint64_t *__restrict__ pairq, unsigned long long Q, unsigned long long max_false,
                 unsigned long long *__restrict__ counters, int optimistic) {
  const unsigned long long n = counters[6];
  if (n <= max_false) {  // uniform over the launch
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[11] = 0ull;
    return;
  }
  const unsigned long long avail = n < Q ? n : Q;
  const unsigned long long sampled = avail < 256ull ? avail : 256ull;
  // FOUR lanes per sampled pair (atoms a = sub, sub + 4, ...; the partial covariances meet by two shuffle steps) in FOUR
  // workgroups of 256 threads: this kernel sits in front of the refine on its lane, and one lane walking 2 x 50 atoms of
  // dependent loads took 50-90 us of every prune with a long queue.  (One workgroup of 1 024 threads does not find room
  // on a CU beside the next prune's screen, whose workgroups hold two register sets per SIMD: it waited for a CU to
  // drain, 0.03 ms per step of the pipelined prune.)  counters[34] / [35]: samples that passed / workgroups done.
  const unsigned sample = blockIdx.x * 64u + (threadIdx.x >> 2), sub = threadIdx.x & 3u;
  int pass = 0;
  {
    const bool on = sample < sampled;
    const uint64_t pr = pairq[on ? (unsigned long long)sample * avail / sampled : 0ull];
    const int64_t i = (int64_t)(pr >> 32), j = (int64_t)(pr & 0xffffffffull);
    const double *__restrict__ p = Xa + i * (int64_t)A * 3, *__restrict__ q = Xa + j * (int64_t)A * 3;
    double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = (int)sub; a < A; a += 4) {
#pragma unroll
      for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) B[x * 3 + y] = fma(p[a * 3 + x], q[a * 3 + y], B[x * 3 + y]);
    }
#pragma unroll
    for (int e9 = 0; e9 < 9; ++e9) {
      B[e9] += __shfl_xor(B[e9], 1);
      B[e9] += __shfl_xor(B[e9], 2);
    }
    pass = (on && sub == 0 && kabsch_may_be_below(B, G[i] + G[j], A_thr2)) ? 1 : 0;
  }
  const int passed_here = __syncthreads_count(pass);
  if (threadIdx.x == 0) {
    atomicAdd(&counters[34], (unsigned long long)passed_here);
    __threadfence();
    if (atomicAdd(&counters[35], 1ull) == (unsigned long long)gridDim.x - 1ull) {  // the last workgroup decides
      const unsigned long long passed = atomicAdd(&counters[34], 0ull);
      const double est_false = (double)n * (1.0 - (double)passed / (double)sampled);
      const bool redo = est_false > (double)max_false;
      counters[11] = (redo && !optimistic) ? 1ull : 0ull;
      if (redo) counters[4] = 0ull, counters[6] = 0ull;
      if (redo && optimistic) counters[12] = 1ull;  // no fp64 launch behind this one: the pair ladder declines (Context::optimistic_screen)
      counters[34] = 0ull;
      counters[35] = 0ull;
    }
  }
}

// largest element of a non-negative array (bit patterns of non-negative doubles order like integers)
__global__ void k_max_nonneg(const double *__restrict__ x, int64_t n, unsigned long long *__restrict__ out) {
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    if (v > m) m = v;  // false for NaN
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(m, off);
    if (o > m) m = o;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

__global__ void k_f64_to_f32(const double *__restrict__ x, int64_t n, float *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}

// ---------------------------------------------------------------------------
// k_screen_rowsweep -- the same screen, other schedule.  MEASURED SLOWER, kept as evidence
// (FC_SCREEN_V2=1): 1.05 ms with the phase barriers, 1.00 ms without them (-DFC_V2_ALIGN=0), against
// 0.96 ms for the kernel above at 10^4 conformers (3.80 / 3.64 / 3.43 ms at 2*10^4).
//
// What the one-item-per-workgroup kernel above loses is known: the
// polynomial epilogue of one wavefront is ADDED to the MFMA stream of the other wavefront of
// its SIMD and issues at half rate while it runs alone; a freed slot waits ~6 us for its
// next workgroup; the column tile is fetched in front of every item.  Here a workgroup is
// 8 wavefronts (2 per SIMD, one workgroup per CU) that stays resident and sweeps a contiguous
// run of items:
//   * the next item's column tile arrives by LDS-DMA in the OTHER half of LDS while this one
//     is computed (two 78 KB buffers);
//   * barriers keep the wavefronts in phase -- K loop, epilogue, K loop, epilogue -- so the
//     two wavefronts of a SIMD do their fp64 VALU work together, at full issue rate, and
//     their MFMA work together;
//   * no dispatch between items, equal item counts per workgroup (tail <= one item).
// Wavefront w owns the 16-row tile w of the 128-row block.  A <= 52 atoms, row blocks of 128.
// Why it loses: two fp64-VALU wavefronts on a SIMD only reach 1.4x the rate of one
// (tools/ubench_f64: 52 vs 38 TFLOP/s), so aligned epilogues gain little, while waves in phase
// stall together on the first operand loads of every K loop and wait for the slowest of eight
// at every barrier; without the phase barriers the per-item barrier and the spills (124 B)
// still cost more than the dispatch gap and the tile fill they remove.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 1)
k_screen_rowsweep(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N, int64_t Npad,
                  int A, double A_thr2, int64_t rank, int64_t world, uint64_t *__restrict__ bits, int64_t W,
                  uint32_t *__restrict__ cand, unsigned long long *__restrict__ counters,
                  uint64_t *__restrict__ pairq, unsigned long long Q, const uint64_t *__restrict__ item_table,
                  unsigned long long n_items) {
  extern __shared__ double lds[];
  constexpr int TC = 64, IB = 128, NW = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = (A + 3) >> 2;
  const int tile_doubles = KS * 12 * TC;
  double *__restrict__ ldsGc = lds + 2 * tile_doubles;  // [2][TC] column sums of squares
  double *__restrict__ ldsGr = ldsGc + 2 * TC;          // [IB] row sums of squares
  uint64_t *__restrict__ stageQ = reinterpret_cast<uint64_t *>(ldsGr + IB);
  uint32_t *__restrict__ stageW = reinterpret_cast<uint32_t *>(stageQ + kStagePairs);
  unsigned *__restrict__ stageN = reinterpret_cast<unsigned *>(stageW + kStageWords);
  const unsigned long long t_begin = n_items * blockIdx.x / gridDim.x;
  const unsigned long long t_end = n_items * (blockIdx.x + 1ull) / gridDim.x;
  if (t_begin >= t_end) return;
  const int kq = lane >> 4, l15 = lane & 15;
  const int boff = (kq >> 1) * (2 * TC) + (kq & 1) * 16 + l15;
  const int cs_l = lane >> 4, k1_l = (lane >> 3) & 1, c15_l = (lane & 7) * 2;
  uint16_t *bits16 = reinterpret_cast<uint16_t *>(bits);
  const int n32 = (int)N;

  auto dma_tile = [&](int64_t jt_, int which) {  // column tile jt_ -> buffer `which` (see the kernel above)
    double *dst = lds + which * tile_doubles;
    const int n_runs = KS * 6;
    for (int q = wv; q < n_runs; q += NW) {
      const int sc = q >> 1, kh = q & 1;
      const int sg = sc / 3, c = sc - sg * 3;
      const int a = sg * 4 + kh * 2 + k1_l;
      const double *src = Xs + (int64_t)(a * 3 + c) * Npad + jt_ * TC + cs_l * 16 + c15_l;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + q * 128), 16, 0, 0);
    }
    if (tid < TC) {
      const int64_t g = jt_ * TC + tid;
      ldsGc[which * TC + tid] = g < Npad ? G[g] : 0.0;
    }
  };
  auto flush_stage = [&]() {  // wavefront 0: publish and reset the staged candidates
    {
      const uint64_t e = lane < kStagePairs ? stageQ[lane] : ~0ull;
      const bool valid = e != ~0ull;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[6], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) {
          const unsigned long long slot = gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull));
          if (slot < Q) pairq[slot] = e;
        }
      }
      if (lane < kStagePairs) stageQ[lane] = ~0ull;
    }
    {
      const uint32_t wq = lane < kStageWords ? stageW[lane] : ~0u;
      const bool valid = wq != ~0u;
      const uint64_t mv = __ballot(valid);
      if (mv != 0) {
        unsigned long long gbase = 0;
        if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)__popcll(mv));
        gbase = __shfl(gbase, 0);
        if (valid) cand[gbase + (unsigned long long)__popcll(mv & ((1ull << lane) - 1ull))] = wq;
      }
      if (lane < kStageWords) stageW[lane] = ~0u;
    }
    if (lane < 2) stageN[lane] = 0u;
  };

  if (tid < kStagePairs) stageQ[tid] = ~0ull;
  if (tid < kStageWords) stageW[tid] = ~0u;
  if (tid < 2) stageN[tid] = 0u;
  {
    const uint64_t it0 = item_table[t_begin];
    dma_tile((int64_t)(it0 & 0x7fffffffull), 0);
  }
  int64_t cur_lb = -1;
  double a0[3], a1[3], a2[3];
  unsigned voff[3] = {0, 0, 0};
  bool pre_ok = false;  // a0..a2 hold the first three k-steps of this wavefront's rows

  for (unsigned long long t = t_begin; t < t_end; ++t) {
    const uint64_t item = item_table[t];
    const int64_t lb = (int64_t)((item >> 32) & 0x7fffffffull);
    const int64_t jt = (int64_t)(item & 0x7fffffffull);
    const int p = (int)((t - t_begin) & 1ull);
    const int64_t j0 = jt * TC;
    const int64_t i0 = global_block(lb, rank, world) * IB;
    const int64_t ib = i0 + (int64_t)wv * 16;
    const int64_t lrow0 = lb * IB + (int64_t)wv * 16;
    const double *__restrict__ cur = lds + p * tile_doubles;
    __syncthreads();  // (A) this item's tile has landed; everybody is done with the other buffer
    if (t + 1 < t_end) dma_tile((int64_t)(item_table[t + 1] & 0x7fffffffull), p ^ 1);
    if (wv == 0 && t != t_begin) flush_stage();
    if (lb != cur_lb) {
      if (tid < IB) {
        const int64_t g = i0 + tid;
        ldsGr[tid] = g < Npad ? G[g] : 0.0;  // read after barrier (B)
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) voff[c] = (unsigned)((int64_t)(kq * 3 + c) * Npad + ib + l15);
      cur_lb = lb;
      pre_ok = false;
#if !FC_V2_ALIGN
      __syncthreads();  // block-uniform branch: the row sums are read in the first epilogue
#endif
    }
    const bool row_on = ib < N && !(j0 + TC - 1 <= ib);
    unsigned nz0 = 0, nz1 = 0, nz2 = 0, nz3 = 0;

#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      const int cs0 = half * 2;
      const bool below = j0 + (cs0 + 2) * 16 - 1 <= ib;  // both sub-tiles at or below the diagonal
      const bool active = row_on && !below;
      d4_t acc[2][9];
      if (active) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int e = 0; e < 9; ++e) acc[tt][e] = d4_t{0.0, 0.0, 0.0, 0.0};
        const double *__restrict__ lb0 = cur + cs0 * 32 + boff;
        double b0[2][3], b1[2][3];
        auto fetch_a = [&](double (&a)[3], int sx) {
          const int sl = sx < KS ? sx : KS - 1;
          const double *__restrict__ xs_s = Xs + (int64_t)sl * 12 * Npad;
#pragma unroll
          for (int c = 0; c < 3; ++c) a[c] = xs_s[voff[c]];
        };
        auto fetch_b = [&](double (&b)[2][3], int sx) {
          const int sl = sx < KS ? sx : KS - 1;
          const double *__restrict__ lb_s = lb0 + sl * (12 * TC);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int c = 0; c < 3; ++c) b[tt][c] = lb_s[c * (4 * TC) + tt * 32];
        };
        auto mma = [&](const double (&a)[3], const double (&b)[2][3]) {
#pragma unroll
          for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y)
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
                acc[tt][x * 3 + y] =
                    __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[tt][y], acc[tt][x * 3 + y], 0, 0, 0);
        };
        if (!pre_ok) {
          fetch_a(a0, 0);
          fetch_a(a1, 1);
          fetch_a(a2, 2);
        }
        fetch_b(b0, 0);
        fetch_b(b1, 1);
#define FC_KSTEP2(AX, BX, U)          \
  if (sgrp + (U) < KS) {              \
    mma(AX, BX);                      \
    fetch_a(AX, sgrp + (U) + 3);      \
    fetch_b(BX, sgrp + (U) + 2);      \
  }
        for (int sgrp = 0; sgrp < KS; sgrp += 6) {
          FC_KSTEP2(a0, b0, 0)
          FC_KSTEP2(a1, b1, 1)
          FC_KSTEP2(a2, b0, 2)
          FC_KSTEP2(a0, b1, 3)
          FC_KSTEP2(a1, b0, 4)
          FC_KSTEP2(a2, b1, 5)
        }
#undef FC_KSTEP2
        // the same rows serve the next unit (second half, or the next item of this row block)
        fetch_a(a0, 0);
        fetch_a(a1, 1);
        fetch_a(a2, 2);
        pre_ok = true;
      }
#if FC_V2_ALIGN
      __syncthreads();  // (B)/(D): both wavefronts of a SIMD leave their K loops together
#endif
      if (active) {
        const int ib32 = (int)ib;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int cs = cs0 + tt;
          const int j = (int)j0 + cs * 16 + l15;
          const double Gq = ldsGc[p * TC + cs * 16 + l15];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ib32 + kq + 4 * r;
            const double Gp = ldsGr[wv * 16 + kq + 4 * r];
            double B9[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) B9[e] = acc[tt][e][r];
            bool may = kabsch_may_be_below(B9, Gp + Gq, A_thr2);
            may = may && (j > i) && (j < n32) && (i < n32);
            const uint64_t m = __ballot(may);
            stage_pairs(m, may, (unsigned)i, (unsigned)j, stageQ, stageN, pairq, Q, counters, lane);
            if (lane < 4) {
              const unsigned piece = (unsigned)((m >> (16 * lane)) & 0xffffull);
              if (ib32 + lane + 4 * r < n32) {
                bits16[((lrow0 + lane + 4 * r) * W + jt) * 4 + cs] = (uint16_t)piece;
                if (r == 0) nz0 |= piece;
                if (r == 1) nz1 |= piece;
                if (r == 2) nz2 |= piece;
                if (r == 3) nz3 |= piece;
              }
            }
          }
        }
      } else if (row_on && below && lane < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = ib + lane + 4 * r;
          if (row < N) {
            bits16[((lrow0 + lane + 4 * r) * W + jt) * 4 + cs0] = 0;
            bits16[((lrow0 + lane + 4 * r) * W + jt) * 4 + cs0 + 1] = 0;
          }
        }
      }
#if FC_V2_ALIGN
      if (half == 0) __syncthreads();  // (C): ... and their epilogues together
#endif
    }
    if (row_on) {  // queue the non-empty words of this row tile for the exact refine
      const unsigned nz[4] = {nz0, nz1, nz2, nz3};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool has = lane < 4 && nz[r] != 0;
        const uint64_t mw = __ballot(has);
        if (mw == 0) continue;  // wave-uniform
        const unsigned n = (unsigned)__popcll(mw);
        const uint32_t word = (uint32_t)((lrow0 + lane + 4 * r) * W + jt);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(stageN + 1, n);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned rank_in = (unsigned)__popcll(mw & ((1ull << lane) - 1ull));
        if (base + n <= (unsigned)kStageWords) {
          if (has) stageW[base + rank_in] = word;
        } else {
          unsigned long long gbase = 0;
          if (lane == 0) gbase = atomicAdd(&counters[4], (unsigned long long)n);
          gbase = __shfl(gbase, 0);
          if (has) cand[gbase + rank_in] = word;
        }
      }
    }
  }
  __syncthreads();
  if (wv == 0) flush_stage();
}

// ---------------------------------------------------------------------------
// k_refine_pairs: the exact fp64 decision of every queued candidate pair, ONE LANE PER PAIR on the
// conformer-minor layout Xs (round 3; the pair mode of k_simbits_refine below is what it replaces).
// The queue is written sub-tile by sub-tile of the screen (a ballot = up to 4 rows x 16 consecutive columns,
// a workgroup's stage = one 128-row x 64-column item), so the 64 pairs of a wavefront name a handful of row
// conformers and runs of consecutive column conformers: in Xs[(a*3 + c) * Npad + n] one load instruction then
// touches 2-6 cache lines, where the 8-lanes-per-pair walk over the conformer-major copy touched 16-24
// (192-byte pieces at a 24-byte lane stride) and spent 82 % of its wave cycles waiting (SQ_WAIT_ANY,
// profiles/r03_pmc_secondary_before.txt: 0.73 ms for 8.9e5 candidates).  No shuffles, no group reductions:
// covariance (9 fused multiply-adds per atom, operands one atom ahead), the fp64 screen polynomial, the rotation
// as a quaternion (kabsch_quaternion_qcp; Jacobi sweeps where that declines), the explicit rotated difference on
// -R, the decision -- all per lane.  Same outputs as the pair mode of k_simbits_refine: counters[1..3], simq, bits.
// ---------------------------------------------------------------------------
// counters words of the XCD-partitioned candidate queue (k_pairq_partition -> k_refine_pairs)

__global__ void __launch_bounds__(256)
k_refine_pairs(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N, int64_t Npad, int A,
               double max_rmsd, double max_dev, const double *__restrict__ energies, double max_dE, int IB,
               int64_t world, uint64_t *__restrict__ bits, int64_t W, unsigned long long *__restrict__ counters,
               const uint64_t *__restrict__ pairq, unsigned long long Q, uint64_t *__restrict__ simq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const unsigned long long n_pairs = counters[6];
  if (n_pairs > Q) return;  // the queue overflowed: the word queue (k_simbits_refine) holds the work
  // a short queue is a latency problem, not a bandwidth one: 2*10^4 pairs are 312 wavefronts here, each walking
  // 2 x 50 atoms of dependent loads (106 us), where the 8-lanes-per-pair walk of k_simbits_refine spreads them
  // over four times as many (56 us) -- it keeps the queues below kRefineLanesMin pairs
  if (n_pairs <= kRefineLanesMin) return;
  const int A2 = (A + 1) & ~1;  // Xs rows are padded to a multiple of 4 atoms with zeros
  // (Splitting a long queue by column range into one part per XCD, each XCD drawing batches from its own part
  // so that its 4 MB L2 sees an eighth of the column conformers, was built and measured on the 8.9e5-candidate
  // ensemble: 0.62 ms + 0.08 ms for the split against 0.44 ms -- the parts interleave the row blocks that the
  // screen's emission order keeps together.  Not kept.)
  // counters[1] / [3] (statistics) are added once per wavefront, at the end: with counters[2] they share one line, and a
  // wavefront's two or three atomics per 64 pairs on it -- 3e4 per launch at 9.45e5 candidates -- serialise at the L2
  unsigned long long acc_refined = 0, acc_grey = 0;
  for (int64_t base = wave0 * 64; base < (int64_t)n_pairs; base += nwaves * 64) {
    const bool on = base + lane < (int64_t)n_pairs;
    const uint64_t e = pairq[on ? base + lane : base];
    const unsigned i = (unsigned)(e >> 32), j = (unsigned)(e & 0xffffffffull);
    double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto load_atom = [&](int a, double (&P)[3], double (&Qv)[3]) {
      const int al = a < A2 ? a : A2 - 1;
      const double *__restrict__ row = Xs + (int64_t)(al * 3) * Npad;  // wave-uniform
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        P[c] = (row + (int64_t)c * Npad)[i];
        Qv[c] = (row + (int64_t)c * Npad)[j];
      }
    };
    {
      double PA[3], QA[3], PB[3], QB[3];
      auto cov = [&](const double (&P)[3], const double (&Qv)[3]) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
          for (int y = 0; y < 3; ++y) B[x * 3 + y] = fma(P[x], Qv[y], B[x * 3 + y]);
      };
      load_atom(0, PA, QA);
      for (int a = 0; a < A2; a += 2) {
        load_atom(a + 1, PB, QB);
        __builtin_amdgcn_sched_barrier(0);
        cov(PA, QA);
        __builtin_amdgcn_sched_barrier(0);
        load_atom(a + 2, PA, QA);
        __builtin_amdgcn_sched_barrier(0);
        cov(PB, QB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const double Gs = G[i] + G[j];
    // the fp64 screen polynomial on this exact covariance first: the single-precision screens pass dissimilar
    // pairs inside their band, and what the fp64 screen would have dropped needs neither a rotation nor a
    // second pass over the atoms
    const bool may = on && kabsch_may_be_below(B, Gs, (double)A * (max_rmsd * max_rmsd + kScreenMargin));
    double ssq = 0.0, mx = 0.0;
    if (__any(may)) {
      double nR[9];
      {
        double Q4[4];
        const bool fast = kabsch_quaternion_qcp(B, Gs, Q4);
        if (__any(may && !fast)) {  // not clearly simple (symmetric or degenerate structures): the Jacobi sweeps
          double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
          if (may && !fast) (void)kabsch_rotation(B, R);
          if (fast) neg_rotation_from_quaternion(Q4, nR);
          else
#pragma unroll
            for (int k = 0; k < 9; ++k) nR[k] = -R[k];
        } else {
          neg_rotation_from_quaternion(Q4, nR);
        }
      }
      double PA[3], QA[3], PB[3], QB[3];
      auto dev = [&](const double (&P)[3], const double (&Qv)[3]) {
#pragma clang fp contract(fast)
        const double dx = fma(nR[0], Qv[0], fma(nR[1], Qv[1], fma(nR[2], Qv[2], P[0])));
        const double dy = fma(nR[3], Qv[0], fma(nR[4], Qv[1], fma(nR[5], Qv[2], P[1])));
        const double dz = fma(nR[6], Qv[0], fma(nR[7], Qv[1], fma(nR[8], Qv[2], P[2])));
        const double s2 = fma(dz, dz, fma(dy, dy, dx * dx));
        ssq += s2;
        asm("v_max_f64 %0, %0, %1" : "+v"(mx) : "v"(s2));
      };
      load_atom(0, PA, QA);
      for (int a = 0; a < A2; a += 2) {
        load_atom(a + 1, PB, QB);
        __builtin_amdgcn_sched_barrier(0);
        dev(PA, QA);
        __builtin_amdgcn_sched_barrier(0);
        load_atom(a + 2, PA, QA);
        __builtin_amdgcn_sched_barrier(0);
        dev(PB, QB);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    bool sim = false, grey = false;
    if (on) {
      const double r = sqrt(ssq / (double)A), m = sqrt(mx);
      sim = may && (r < max_rmsd) && (m < max_dev);
      grey = may && ((fabs(r - max_rmsd) < 1e-9) || (r < max_rmsd && fabs(m - max_dev) < 1e-9));
      if (energies != nullptr) sim = sim && (fabs(energies[i] - energies[j]) < max_dE);
      if (!sim && bits != nullptr) {
        const int64_t lrow = (((int64_t)i / IB) / world) * IB + ((int64_t)i % IB);
        atomicAnd(reinterpret_cast<unsigned long long *>(&bits[lrow * W + (j >> 6)]), ~(1ull << (j & 63)));
      }
    }
    const uint64_t mo = __ballot(on), ms = __ballot(on && sim), mg = __ballot(on && grey);
    unsigned long long sbase = 0;
    acc_refined += (unsigned long long)__popcll(mo);
    acc_grey += (unsigned long long)__popcll(mg);
    if (lane == 0 && ms) sbase = atomicAdd(&counters[2], (unsigned long long)__popcll(ms));
    sbase = __shfl(sbase, 0);
    if (on && sim) simq[sbase + (unsigned long long)__popcll(ms & ((1ull << lane) - 1ull))] = e;
  }
  if (lane == 0) {
    if (acc_refined) atomicAdd(&counters[1], acc_refined);
    if (acc_grey) atomicAdd(&counters[3], acc_grey);
  }
}

// ---------------------------------------------------------------------------
// The long candidate queues, BUCKET BY BUCKET (round 4; replaces the straight walk of k_refine_pairs above, which stays
// as the FC_REFINE_BUCKETS=0 form).  k_refine_pairs fetched both conformers of every pair through the XCD's L2 twice
// (covariance pass, deviation pass): 2.07 GB of fabric-side traffic per launch for a 12.5 MB ensemble at 9.45e5
// candidates, L2 hit rate 0.7, 68 % of the wave cycles waiting on loads (profiles/r03_pmc_refine.json).  Here the queue
// is first ordered by the (128-row block, 64-column tile) its pairs fall in -- a counting sort in three small launches:
// k_bucket_count, k_bucket_scan, k_bucket_scatter -- and k_refine_buckets gives each non-empty bucket to one workgroup:
// the bucket's COLUMN tile goes to LDS once (LDS-DMA, 512-byte rows [atom * 3 + c][64 columns]) and serves both passes
// of all the bucket's pairs; the row conformers are 128 consecutive ones (1 KiB per coordinate row), read by every lane
// of the workgroup and by the buckets of the same row block that follow on this XCD: L2 hits.  One lane per pair, the
// arithmetic of k_refine_pairs unchanged (same functions, same order of operations: same bits).
// Control block `bk` (int32): kBkCtrl control words (below), then the counts per bucket (n_buckets + 1), then the scatter
// cursors (n_buckets); zeroed by the launcher.
// ---------------------------------------------------------------------------
// Order of the buckets (round 4, second step): SUPERTILE-major -- 8 row blocks x 8 column tiles (1 024 x 512 conformers,
// 1.9 MB of coordinates at 50 atoms: half an XCD's L2) -- and k_refine_buckets hands the work items of supertile s to
// the workgroups of XCD s mod 8 (HW_REG_XCC_ID picks the cursor a workgroup draws from first; any workgroup may draw
// from any cursor, so nothing depends on where the hardware puts workgroups).  With buckets in plain row-block-major
// order every staged column tile was new to its XCD's L2 (79 tiles of 80 KB per row block against 4 MB): 3.25 M L2
// misses per launch, and the covariance pass -- the first touch of the row conformers, competing with those misses --
// took 0.23 ms of the 0.51 while the identical loads of the deviation pass, L2 hits, were free.
constexpr int kBkCtrl = 512;    // int32 words in front of the counts: [0] work items, [64 + 32 x] cursor of XCD x (a line each)
// A bucket = 1 024 rows x 64 columns.  Only the COLUMN tile costs LDS (the rows are read from L2), so the rows of a
// bucket can be many -- and have to be: what a CU can keep busy is (pairs of its resident tiles) / 64 wavefronts, and
// with 128-row buckets (152 pairs at 1.9 % candidates) the two resident workgroups had 2 x 2.4 active wavefronts on
// four SIMDs: every instruction latency exposed (~480 cycles per atom step whatever the loads did), 0.51 ms.
constexpr int kSuper = 8;       // column tiles per supertile (one row bucket x 8 tiles = 1 024 x 512 conformers)

struct BucketGeom {
  unsigned NT, n_sc;            // column tiles, supertile columns = ceil(NT / 8)
};
__device__ __forceinline__ unsigned bucket_of(uint64_t e, BucketGeom g) {
  const unsigned rb = (unsigned)(e >> 32) / (unsigned)kBucketRows, jt = (unsigned)(e & 0xffffffffull) >> kBucketColShift;
  return ((rb * g.n_sc + (jt >> 3)) << 3) | (jt & 7u);
}

// one atomic per distinct bucket of a wavefront's 64 queue entries (the queue comes in runs of one screen workgroup's
// pairs: a wavefront usually names one or two buckets); want_pos: every lane also learns its place inside the run
template <bool WANT_POS>
__device__ __forceinline__ unsigned bucket_add_wave(bool on, unsigned b, int *__restrict__ counter_base, int lane) {
  unsigned pos = 0;
  uint64_t todo = __ballot(on);
  while (todo != 0ull) {  // wave-uniform
    const int leader = __builtin_ctzll(todo);
    const unsigned bl = (unsigned)__builtin_amdgcn_readlane((int)b, leader);
    const bool mine = on && b == bl;
    const uint64_t m = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(counter_base + bl, (int)__popcll(m));
    if (WANT_POS) {
      base = __builtin_amdgcn_readlane(base, leader);
      if (mine) pos = (unsigned)base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    }
    todo &= ~m;
  }
  return pos;
}

__global__ void __launch_bounds__(256)
k_bucket_count(const uint64_t *__restrict__ pairq, const unsigned long long *__restrict__ counters, unsigned long long Q,
               BucketGeom geom, int *__restrict__ bk) {
  const unsigned long long n_pairs = counters[6];
  if (n_pairs > Q || n_pairs <= kRefineLanesMin) return;
  const int lane = threadIdx.x & 63;
  for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ll; p0 < (int64_t)n_pairs;
       p0 += (int64_t)gridDim.x * blockDim.x) {
    const bool on = p0 + lane < (int64_t)n_pairs;
    const uint64_t e = pairq[on ? p0 + lane : p0];
    (void)bucket_add_wave<false>(on, bucket_of(e, geom), bk + kBkCtrl, lane);
  }
}

// ONE workgroup: exclusive scan of the bucket counts -> off[0 .. n_buckets]; the work items of k_refine_buckets --
// a bucket in pieces of at most kBucketChunk pairs (one round of a workgroup: a bucket may hold 8 192 pairs, and one
// workgroup walking those alone was the tail of the whole launch), encoded bucket | piece << 24, buckets ascending
// (row block major: consecutive workgroups share their row block) -> list, their number -> bk[0]
#ifndef FC_RB_CHUNK
#define FC_RB_CHUNK 512
#endif
#ifndef FC_RB_WPS
#define FC_RB_WPS 3
#endif
constexpr int kBucketChunk = FC_RB_CHUNK;
__global__ void __launch_bounds__(1024)
k_bucket_scan(const unsigned long long *__restrict__ counters, unsigned long long Q, int64_t n_buckets, int *__restrict__ bk,
              int *__restrict__ off, int *__restrict__ list, int *__restrict__ st_off, int *__restrict__ xoff, int n_st, int x_stride) {
  const unsigned long long n_pairs = counters[6];
  if (n_pairs > Q || n_pairs <= kRefineLanesMin) return;
  __shared__ int s_sum[1024], s_cnt[1024];
  const int t = threadIdx.x;
  const int *__restrict__ cnt = bk + kBkCtrl;
  // whole supertiles (8 buckets) per thread, so that a supertile's first work item is known to the thread that owns it
  const int64_t per = ((n_buckets / kSuper + 1023) / 1024) * kSuper, b0 = (int64_t)t * per, b1 = b0 + per < n_buckets ? b0 + per : n_buckets;
  int sum = 0, ne = 0;
  for (int64_t b = b0; b < b1; ++b) {
    const int c = cnt[b];
    sum += c;
    ne += (c + kBucketChunk - 1) / kBucketChunk;
  }
  s_sum[t] = sum;
  s_cnt[t] = ne;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scans of both
    const int a = t >= d ? s_sum[t - d] : 0, c = t >= d ? s_cnt[t - d] : 0;
    __syncthreads();
    s_sum[t] += a;
    s_cnt[t] += c;
    __syncthreads();
  }
  int run = s_sum[t] - sum, place = s_cnt[t] - ne;
  for (int64_t b = b0; b < b1; ++b) {
    const int c = cnt[b];
    if ((b & (kSuper - 1)) == 0) st_off[b / kSuper] = place;
    off[b] = run;
    for (int piece = 0; piece * kBucketChunk < c; ++piece) list[place++] = (int)b | (piece << 24);
    run += c;
  }
  if (t == 1023) {
    off[n_buckets] = s_sum[1023];
    st_off[n_st] = s_cnt[1023];
    bk[0] = s_cnt[1023];
  }
  __syncthreads();  // (one workgroup: its global writes above are visible to it behind the barrier)
  // work items in front of each supertile WITHIN the sequence of its XCD (supertiles x, x + 8, x + 16, ...): xoff[x][r]
  if (t < 8) {
    int acc = 0, r = 0;
    for (int sidx = t; sidx < n_st; sidx += 8, ++r) {
      xoff[t * x_stride + r] = acc;
      acc += st_off[sidx + 1] - st_off[sidx];
    }
    xoff[t * x_stride + r] = acc;  // total of XCD t (r = its number of supertiles)
  }
}

__global__ void __launch_bounds__(256)
k_bucket_scatter(const uint64_t *__restrict__ pairq, const unsigned long long *__restrict__ counters, unsigned long long Q,
                 BucketGeom geom, int64_t n_buckets, int *__restrict__ bk, const int *__restrict__ off,
                 uint64_t *__restrict__ sorted) {
  const unsigned long long n_pairs = counters[6];
  if (n_pairs > Q || n_pairs <= kRefineLanesMin) return;
  const int lane = threadIdx.x & 63;
  int *__restrict__ fill = bk + kBkCtrl + n_buckets + 1;
  for (int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ll; p0 < (int64_t)n_pairs;
       p0 += (int64_t)gridDim.x * blockDim.x) {
    const bool on = p0 + lane < (int64_t)n_pairs;
    const uint64_t e = pairq[on ? p0 + lane : p0];
    const unsigned b = bucket_of(e, geom);
    const unsigned pos = bucket_add_wave<true>(on, b, fill, lane);
    if (on) sorted[(int64_t)off[b] + pos] = e;
  }
}

__global__ void __launch_bounds__(kBucketChunk, FC_RB_WPS)  // (second figure: wavefronts per SIMD; 3 = 168 registers, no spills: 0.320 ms against 0.332 with 4 and 37 spilled)
k_refine_buckets(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N, int64_t Npad, int A,
                 double max_rmsd, double max_dev, const double *__restrict__ energies, double max_dE, int IB,
                 int64_t world, uint64_t *__restrict__ bits, int64_t W, unsigned long long *__restrict__ counters,
                 unsigned long long Q, BucketGeom geom, int *__restrict__ bk, const int *__restrict__ off,
                 const int *__restrict__ list, const int *__restrict__ st_off, const int *__restrict__ xoff, int n_st,
                 int x_stride, const uint64_t *__restrict__ sorted, uint64_t *__restrict__ simq) {
  extern __shared__ double lds[];  // [A4 * 3][TCB]: the bucket's column tile
  constexpr int TCB = 1 << kBucketColShift;  // conformers per column tile
  __shared__ int s_next;
  __shared__ int s_bin[kBucketRows];          // pairs per row of the block, then their exclusive scan
  __shared__ uint64_t s_pairs[kBucketChunk];  // the item's pairs in row order
  __shared__ unsigned s_wsim[kBucketChunk / 64], s_won[kBucketChunk / 64], s_wgrey[kBucketChunk / 64];  // per wavefront of the item ...
  __shared__ unsigned long long s_wbase[kBucketChunk / 64];  // ... and where each wavefront's go in simq
  unsigned long long acc_refined = 0, acc_grey = 0;          // (thread 0) counters[1], [3]: added once, at the end
  const unsigned long long n_pairs = counters[6];
  if (n_pairs > Q || n_pairs <= kRefineLanesMin) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int A4 = (A + 3) & ~3;  // Xs rows are padded to a multiple of 4 atoms with zeros
  const int n_rows = A4 * 3;    // even
  const double A_thr2 = (double)A * (max_rmsd * max_rmsd + kScreenMargin);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(63508) & 7u);  // HW_REG_XCC_ID: which XCD (and L2) this workgroup runs on
  int turn = 0;  // XCD sequences tried so far: the own one first, then the others' leftovers
  while (true) {
    __syncthreads();  // everyone is done with the previous tile (and with s_next)
    if (tid == 0) {
      int item_at = -1;
      while (turn < 8) {
        const int x = (xcc + turn) & 7;
        const int n_x = (n_st - x + 7) / 8;          // supertiles of XCD x
        const int *__restrict__ xo = xoff + x * x_stride;
        const int kx = atomicAdd(bk + 64 + 32 * x, 1);
        if (n_x > 0 && kx < xo[n_x]) {
          int lo = 0, hi = n_x;  // largest r with xo[r] <= kx
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xo[mid] <= kx) lo = mid;
            else hi = mid;
          }
          item_at = st_off[x + 8 * lo] + (kx - xo[lo]);
          break;
        }
        ++turn;
      }
      s_next = item_at;
    }
    __syncthreads();
    const int k = s_next;
    if (k < 0) break;  // block-uniform: every sequence is exhausted
#ifdef FC_RB_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    const unsigned item = (unsigned)list[k];
    const unsigned b = item & 0xffffffu, piece = item >> 24;
    const unsigned jt = (((b >> 3) % geom.n_sc) << 3) | (b & 7u);
    const int64_t j0 = (int64_t)jt * TCB;
    // column tile by LDS-DMA: one instruction moves 1 KiB = two 512-byte rows (lanes 0-31 the first, 32-63 the second; four
    // rows of a 32-conformer tile)
    constexpr int kRowsPer = 128 / TCB, kLanesPerRow = TCB / 2;
#ifdef FC_RB_NOSTAGE
    for (int q = wv; q < 0; q += kBucketChunk / 64) {
#else
    for (int q = wv; q < n_rows / kRowsPer; q += kBucketChunk / 64) {
#endif
      const double *src = Xs + (int64_t)(kRowsPer * q + lane / kLanesPerRow) * Npad + j0 + (lane % kLanesPerRow) * 2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(lds + q * 128), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the tile is in LDS
    __syncthreads();
#ifdef FC_RB_TIMELINE
    const unsigned long long tl1 = wall_clock64();
#endif
    const int p_begin = off[b] + (int)piece * kBucketChunk, p_end = min(off[b + 1], p_begin + kBucketChunk);
    // The item's pairs in ROW order (a counting sort over the 128 rows of the block, in LDS).  A lane reads its row
    // conformer with per-lane addresses, and the address unit works through a wavefront's load 16 lanes at a time, one
    // cycle per distinct 128-byte line of the group: with the pairs in arrival order a group's 16 rows lie anywhere in
    // the block's 1 KiB (up to 8 lines: ~30 cycles per load instruction, 156 of them per pass -- that, not latency or
    // L2 misses, was what the passes cost: prefetch depth 1 .. 12, supertile order, rows from LDS all left the 0.51 ms
    // where they were); in row order 16 consecutive lanes name ~13 consecutive rows: one or two lines.
    {
      for (int r = tid; r < kBucketRows; r += kBucketChunk) s_bin[r] = 0;
      __syncthreads();
      const int p = p_begin + tid;
      const bool have = p < p_end;
      const uint64_t e_in = have ? sorted[p] : 0ull;
      const int row = (int)((unsigned)(e_in >> 32) % (unsigned)kBucketRows);
      int rank = 0;
      if (have) rank = atomicAdd(&s_bin[row], 1);
      __syncthreads();
      if (wv == 0) {  // exclusive scan of the bins (1 024: sixteen per lane)
        constexpr int kPerLane = kBucketRows / 64;
        static_assert(kBucketRows % 64 == 0 && kBucketRows >= 64, "bins are scanned by one wavefront");
        int c[kPerLane], tot = 0;
#pragma unroll
        for (int k2 = 0; k2 < kPerLane; ++k2) {
          c[k2] = s_bin[kPerLane * lane + k2];
          tot += c[k2];
        }
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int up = __shfl_up(incl, d);
          if (lane >= d) incl += up;
        }
        int run = incl - tot;
#pragma unroll
        for (int k2 = 0; k2 < kPerLane; ++k2) {
          s_bin[kPerLane * lane + k2] = run;
          run += c[k2];
        }
      }
      __syncthreads();
      if (have) s_pairs[s_bin[row] + rank] = e_in;
      __syncthreads();
    }
    const int n_item = p_end - p_begin;
    // what this wavefront found in the item: set inside the round below, published behind it
    uint64_t e_mine = 0ull, ms_mine = 0ull;
    unsigned n_on = 0, n_grey = 0;
    bool sim_mine = false;
#ifdef FC_RB_NOCOMPUTE
    for (int p0 = wv * 64; p0 < 0; p0 += kBucketChunk) {
#else
    for (int p0 = wv * 64; p0 < n_item; p0 += kBucketChunk) {  // (one round: a piece is at most kBucketChunk pairs)
#endif
      const bool on = p0 + lane < n_item;
      const uint64_t e = s_pairs[on ? p0 + lane : p0];
      e_mine = e;
      const unsigned i = (unsigned)(e >> 32), j = (unsigned)(e & 0xffffffffull);
      const double *__restrict__ qcol = lds + (j - (unsigned)j0);
      double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      // Row operands come from L2 (~0.5-1 us under load), column operands from LDS: the row loads run kAhead atoms
      // ahead of their use in a ring of register sets (with one atom ahead, as in k_refine_pairs, a pass of 50 atoms is
      // a chain of 50 L2 round trips: 0.51 ms for 9.45e5 candidates, no better than the straight walk); the Xs rows
      // beyond A are zeros up to a multiple of 4 atoms, and the passes run over all of them
#ifndef FC_RB_AHEAD
#define FC_RB_AHEAD 4
#endif
      constexpr int kAhead = FC_RB_AHEAD;
      auto load_row = [&](int a, double (&P)[3]) {
        const int al = a < A4 ? a : A4 - 1;  // past the end: harmless re-read, never used
#ifdef FC_RB_ROWS_FROM_LDS  // tuning build (WRONG results): what the kernel costs without its global row loads
#pragma unroll
        for (int c = 0; c < 3; ++c) P[c] = lds[(al * 3 + c) * TCB + (i & (TCB - 1))];
#else
        const double *__restrict__ row = Xs + (int64_t)(al * 3) * Npad;  // wave-uniform
#ifdef FC_RB_SAMEROW  // tuning build (WRONG results): every lane reads the first lane's row -- one line per load
        const unsigned iu = (unsigned)__builtin_amdgcn_readfirstlane((int)i) + (unsigned)(lane & 15);
#pragma unroll
        for (int c = 0; c < 3; ++c) P[c] = (row + (int64_t)c * Npad)[iu];
#else
#pragma unroll
        for (int c = 0; c < 3; ++c) P[c] = (row + (int64_t)c * Npad)[i];
#endif
#endif
      };
      auto load_col = [&](int a, double (&Qv)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Qv[c] = qcol[(a * 3 + c) * TCB];
      };
      {
        double P[kAhead][3], Qv[3];
        auto cov = [&](const double (&Pp)[3], const double (&Qq)[3]) {
#pragma unroll
          for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) B[x * 3 + y] = fma(Pp[x], Qq[y], B[x * 3 + y]);
        };
#pragma unroll
        for (int u = 0; u < kAhead; ++u) load_row(u, P[u]);
#ifdef FC_RB_NOPASS1
        for (int a = 0; a < 4; a += kAhead) {
#else
        for (int a = 0; a < A4; a += kAhead) {
#endif
#pragma unroll
          for (int u = 0; u < kAhead; ++u) {
            if (kAhead > 4 && a + u >= A4) break;  // (A4 is a multiple of 4; wave-uniform)
            load_col(a + u, Qv);
            __builtin_amdgcn_sched_barrier(0);
            cov(P[u], Qv);
            __builtin_amdgcn_sched_barrier(0);
            load_row(a + u + kAhead, P[u]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#ifdef FC_RB_TIMELINE
      const unsigned long long tp1 = wall_clock64();
#endif
      const double Gs = G[i] + G[j];
      const bool may = on && kabsch_may_be_below(B, Gs, A_thr2);
      double ssq = 0.0, mx = 0.0;
      if (__any(may)) {
        double nR[9];
        {
          double Q4[4];
#ifdef FC_RB_NOJACOBI
          (void)kabsch_quaternion_qcp(B, Gs, Q4);
          const bool fast = true;
#else
          const bool fast = kabsch_quaternion_qcp(B, Gs, Q4);
#endif
          if (__any(may && !fast)) {  // not clearly simple (symmetric or degenerate structures): the Jacobi sweeps
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (may && !fast) (void)kabsch_rotation(B, R);
            if (fast) neg_rotation_from_quaternion(Q4, nR);
            else
#pragma unroll
              for (int kk = 0; kk < 9; ++kk) nR[kk] = -R[kk];
          } else {
            neg_rotation_from_quaternion(Q4, nR);
          }
        }
#ifdef FC_RB_TIMELINE
        if (wv == 0 && lane == 0) atomicAdd(&counters[24], wall_clock64() - tp1);  // polynomial + rotation
#endif
        double P[kAhead][3], Qv[3];
        auto dev = [&](const double (&Pp)[3], const double (&Qq)[3]) {
#pragma clang fp contract(fast)
          const double dx = fma(nR[0], Qq[0], fma(nR[1], Qq[1], fma(nR[2], Qq[2], Pp[0])));
          const double dy = fma(nR[3], Qq[0], fma(nR[4], Qq[1], fma(nR[5], Qq[2], Pp[1])));
          const double dz = fma(nR[6], Qq[0], fma(nR[7], Qq[1], fma(nR[8], Qq[2], Pp[2])));
          const double s2 = fma(dz, dz, fma(dy, dy, dx * dx));
          ssq += s2;
          asm("v_max_f64 %0, %0, %1" : "+v"(mx) : "v"(s2));
        };
#pragma unroll
        for (int u = 0; u < kAhead; ++u) load_row(u, P[u]);
#ifdef FC_RB_NOPASS2
        for (int a = 0; a < 4; a += kAhead) {
#else
        for (int a = 0; a < A4; a += kAhead) {
#endif
#pragma unroll
          for (int u = 0; u < kAhead; ++u) {
            if (kAhead > 4 && a + u >= A4) break;  // (A4 is a multiple of 4; wave-uniform)
            load_col(a + u, Qv);
            __builtin_amdgcn_sched_barrier(0);
            dev(P[u], Qv);
            __builtin_amdgcn_sched_barrier(0);
            load_row(a + u + kAhead, P[u]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
#ifdef FC_RB_TIMELINE
      if (wv == 0 && lane == 0) {
        atomicAdd(&counters[25], wall_clock64() - tp1);  // polynomial + rotation + deviation pass
        atomicAdd(&counters[26], tp1 - tl1);             // pair load + covariance pass (first round of the item)
        atomicAdd(&counters[27], 1ull);
      }
#endif
      bool sim = false, grey = false;
      if (on) {
        const double r = sqrt(ssq / (double)A), m = sqrt(mx);
        sim = may && (r < max_rmsd) && (m < max_dev);
        grey = may && ((fabs(r - max_rmsd) < 1e-9) || (r < max_rmsd && fabs(m - max_dev) < 1e-9));
        if (energies != nullptr) sim = sim && (fabs(energies[i] - energies[j]) < max_dE);
        if (!sim && bits != nullptr) {
          const int64_t lrow = (((int64_t)i / IB) / world) * IB + ((int64_t)i % IB);
          atomicAnd(reinterpret_cast<unsigned long long *>(&bits[lrow * W + (j >> 6)]), ~(1ull << (j & 63)));
        }
      }
      ms_mine = __ballot(on && sim);
      n_on = (unsigned)__popcll(__ballot(on));
      n_grey = (unsigned)__popcll(__ballot(on && grey));
      sim_mine = on && sim;
    }
    // ONE atomic per item for the similar-pair slots, the two statistics once per workgroup at the end.  (A wavefront's
    // own atomics on counters[1..3] -- one line, two or three per 64 pairs, 3e4 per launch at 9.45e5 candidates -- were
    // what both long-queue kernels waited for: they serialise at the L2, ~12 ns each, 0.35 of the 0.48 ms, whatever the
    // loads did and however many workgroups ran.)
    if (lane == 0) {
      s_wsim[wv] = (unsigned)__popcll(ms_mine);
      s_won[wv] = n_on;
      s_wgrey[wv] = n_grey;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned tot = 0;
      for (int w = 0; w < kBucketChunk / 64; ++w) {
        tot += s_wsim[w];
        acc_refined += s_won[w];
        acc_grey += s_wgrey[w];
      }
      unsigned long long at = tot ? atomicAdd(&counters[2], (unsigned long long)tot) : 0ull;
      for (int w = 0; w < kBucketChunk / 64; ++w) {
        s_wbase[w] = at;
        at += s_wsim[w];
      }
    }
    __syncthreads();
    if (sim_mine) simq[s_wbase[wv] + (unsigned long long)__popcll(ms_mine & ((1ull << lane) - 1ull))] = e_mine;
#ifdef FC_RB_TIMELINE  // tuning build: 100 MHz ticks spent staging / computing (per wavefront), items, wave-rounds
    if (lane == 0 && wv == 0) {
      const unsigned long long tl2 = wall_clock64();
      atomicAdd(&counters[20], tl1 - tl0);
      atomicAdd(&counters[21], tl2 - tl1);
      atomicAdd(&counters[22], 1ull);
      if (wv * 64 < n_item) atomicAdd(&counters[23], 1ull);
    }
#endif
  }
  if (tid == 0) {
    if (acc_refined) atomicAdd(&counters[1], acc_refined);
    if (acc_grey) atomicAdd(&counters[3], acc_grey);
  }
}

// ---------------------------------------------------------------------------
// k_simbits_refine: one wavefront per bit word; lane b re-evaluates pair
// (row, jt*64+b) exactly if its screen bit is set and the word is rewritten
// with the final decision  rmsd < max_rmsd && maxdev < max_dev
// [&& |E_i - E_j| < max_dE].  counters[1] += candidates, [2] += similar,
// [3] += pairs within 1e-9 of a threshold ("grey").
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_simbits_refine(const double *__restrict__ Xs, const double *__restrict__ Xa, int64_t N,
                 int64_t Npad, int A, double max_rmsd,
                 double max_dev, const double *__restrict__ energies, double max_dE, int IB,
                 int64_t rank, int64_t world, int64_t rows_local, uint64_t *__restrict__ bits,
                 int64_t W, const uint32_t *__restrict__ cand,
                 unsigned long long *__restrict__ counters, const uint64_t *__restrict__ pairq,
                 unsigned long long Q, uint64_t *__restrict__ simq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const unsigned long long n_pairs = counters[6];
  if (n_pairs <= Q && n_pairs > kRefineLanesMin && IB < 0) return;  // IB < 0: k_refine_pairs takes the long pair queues
  if (IB < 0) IB = -IB;
  if (n_pairs <= Q) {
    // pair mode: the queue is complete.  A wavefront takes 64 candidate pairs.  The two atom
    // passes (covariance, rotated difference) run with EIGHT lanes per pair, 8 pairs per
    // round, 8 rounds -- coalesced 192-byte pieces of Xa -- while the 4x4 eigen-solve between
    // them, 85 % of the arithmetic, runs ONCE with one lane per pair instead of redundantly
    // in the 8 lanes of a group (that redundancy made the kernel VALU bound: 84 us for 3*10^4
    // pairs).  Pair (round r, group s) is owned by lane 8 r + s; sums travel by shuffles.
    const int sub = lane & 7, slot = lane >> 3;
    constexpr int kRounds = FC_REFINE_ROUNDS;  // pairs per wavefront = 8 * kRounds
    for (int64_t base = wave0 * (8 * kRounds); base < (int64_t)n_pairs; base += nwaves * (8 * kRounds)) {
      const int64_t p_own = base + lane;
      const bool on = lane < 8 * kRounds && p_own < (int64_t)n_pairs;
      const uint64_t e = on ? pairq[p_own] : 0ull;
      double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double Gs = 0.0;  // sum of squares of both structures: start value of the eigenvalue iteration
      // The rounds side by side (round 5): every step of the atom loop requests the atoms of ALL rounds' pairs before it
      // uses any -- kRounds x 6 loads in flight instead of 6, the loop's chain of exposed latencies kRounds times shorter
      // (a short queue leaves most of the chip empty: the kernel's time is that chain).  Per pair and lane the same atoms in
      // the same order into the same accumulators: same bits.  A group without a pair reads conformer 0 and its sums are
      // never taken.
      const double *__restrict__ ppr[kRounds], *__restrict__ qqr[kRounds];
#pragma unroll
      for (int round = 0; round < kRounds; ++round) {
        const uint64_t eg = __shfl(e, round * 8 + slot);  // the pair of this 8-lane group
        const bool og = base + round * 8 + slot < (int64_t)n_pairs;
        ppr[round] = Xa + (og ? (int64_t)(eg >> 32) * (int64_t)A * 3 : 0);
        qqr[round] = Xa + (og ? (int64_t)(eg & 0xffffffffull) * (int64_t)A * 3 : 0);
      }
      {
        double Bg[kRounds][9], gg[kRounds];
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
          gg[round] = 0.0;
#pragma unroll
          for (int k = 0; k < 9; ++k) Bg[round][k] = 0.0;
        }
        for (int a = sub; a < A; a += 8) {
          double P[kRounds][3], Qv[kRounds][3];
#pragma unroll
          for (int round = 0; round < kRounds; ++round)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              P[round][c] = ppr[round][a * 3 + c];
              Qv[round][c] = qqr[round][a * 3 + c];
            }
#pragma unroll
          for (int round = 0; round < kRounds; ++round) {
            const double px = P[round][0], py = P[round][1], pz = P[round][2];
            const double qx = Qv[round][0], qy = Qv[round][1], qz = Qv[round][2];
            double(&Bq)[9] = Bg[round];
            Bq[0] = fma(px, qx, Bq[0]); Bq[1] = fma(px, qy, Bq[1]); Bq[2] = fma(px, qz, Bq[2]);
            Bq[3] = fma(py, qx, Bq[3]); Bq[4] = fma(py, qy, Bq[4]); Bq[5] = fma(py, qz, Bq[5]);
            Bq[6] = fma(pz, qx, Bq[6]); Bq[7] = fma(pz, qy, Bq[7]); Bq[8] = fma(pz, qz, Bq[8]);
            gg[round] += (px * px + py * py + pz * pz) + (qx * qx + qy * qy + qz * qz);
          }
        }
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const double tot = group8_sum(Bg[round][k]);
            const double mine = __shfl(tot, sub * 8);  // group `sub` holds the pair lane 8*round+sub owns
            if (slot == round) B[k] = mine;
          }
          const double gmine = __shfl(group8_sum(gg[round]), sub * 8);
          if (slot == round) Gs = gmine;
        }
      }
      // The fp64 screen polynomial on this exact covariance first: the single-precision screen
      // passes dissimilar pairs inside its band, and what the fp64 screen would have dropped
      // needs neither a rotation nor a second pass over the atoms (same decision as with the
      // fp64 screen; a wave without any other pair skips both).
      const bool may = on && kabsch_may_be_below(B, Gs, (double)A * (max_rmsd * max_rmsd + kScreenMargin));
      const bool any_may = __any(may);
      // rotation: Newton eigenvalue + adjugate eigenvector where the eigenvalue is clearly simple
      // (every candidate of a sane ensemble), the Jacobi sweeps otherwise -- same R to ~1e-14
      double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (any_may) {
        const bool fast = may && kabsch_rotation_qcp(B, Gs, R);
        if (may && !fast) (void)kabsch_rotation(B, R);
      }
      double ssq_own = 0.0, mx_own = 0.0;
      if (any_may) {  // (the deviation pass the same way: all rounds' atoms requested before any is used)
        double Rg[kRounds][9], ssq[kRounds], mx[kRounds];
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
          ssq[round] = 0.0;
          mx[round] = 0.0;
#pragma unroll
          for (int k = 0; k < 9; ++k) Rg[round][k] = __shfl(R[k], round * 8 + slot);
        }
        for (int a = sub; a < A; a += 8) {
          double P[kRounds][3], Qv[kRounds][3];
#pragma unroll
          for (int round = 0; round < kRounds; ++round)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              P[round][c] = ppr[round][a * 3 + c];
              Qv[round][c] = qqr[round][a * 3 + c];
            }
#pragma unroll
          for (int round = 0; round < kRounds; ++round) {
            const double px = P[round][0], py = P[round][1], pz = P[round][2];
            const double qx = Qv[round][0], qy = Qv[round][1], qz = Qv[round][2];
            const double(&Rq)[9] = Rg[round];
            const double dx = px - (Rq[0] * qx + Rq[1] * qy + Rq[2] * qz);
            const double dy = py - (Rq[3] * qx + Rq[4] * qy + Rq[5] * qz);
            const double dz = pz - (Rq[6] * qx + Rq[7] * qy + Rq[8] * qz);
            const double sq = dx * dx + dy * dy + dz * dz;
            ssq[round] += sq;
            mx[round] = fmax(mx[round], sq);
          }
        }
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
          const double ts = __shfl(group8_sum(ssq[round]), sub * 8), tm = __shfl(group8_max(mx[round]), sub * 8);
          if (slot == round) {
            ssq_own = ts;
            mx_own = tm;
          }
        }
      }
      bool sim = false, grey = false;
      if (on) {
        const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
        const double r = sqrt(ssq_own / (double)A), m = sqrt(mx_own);
        sim = may && (r < max_rmsd) && (m < max_dev);
        grey = may && ((fabs(r - max_rmsd) < 1e-9) || (r < max_rmsd && fabs(m - max_dev) < 1e-9));
        if (energies != nullptr) sim = sim && (fabs(energies[i] - energies[j]) < max_dE);
        if (!sim && bits != nullptr) {
          const int64_t lrow = ((i / IB) / world) * IB + (i % IB);
          atomicAnd(reinterpret_cast<unsigned long long *>(&bits[lrow * W + (j >> 6)]),
                    ~(1ull << (j & 63)));
        }
      }
      const uint64_t mo = __ballot(on), ms = __ballot(on && sim), mg = __ballot(on && grey);
      unsigned long long sbase = 0;
      if (lane == 0) {
        atomicAdd(&counters[1], (unsigned long long)__popcll(mo));
        if (ms) sbase = atomicAdd(&counters[2], (unsigned long long)__popcll(ms));
        if (mg) atomicAdd(&counters[3], (unsigned long long)__popcll(mg));
      }
      // list of exactly-similar pairs (same capacity as the candidate queue):
      // what the one-launch ladder and the multi-GPU exchange consume
      sbase = __shfl(sbase, 0);
      if (on && sim) simq[sbase + (unsigned long long)__popcll(ms & ((1ull << lane) - 1ull))] = e;
    }
    return;
  }
  if (bits == nullptr) return;  // lean screen: the host repeats this prune with the bit matrix
  const int64_t n_cand = (int64_t)counters[4];
  for (int64_t c = wave0; c < n_cand; c += nwaves) {
  const int64_t widx = cand[c];
  const int64_t lrow = widx / W;
  const int64_t jt = widx % W;
  const int64_t lb = lrow / IB;
  const int64_t i = global_block(lb, rank, world) * IB + (lrow % IB);
  const uint64_t word = bits[lrow * W + jt];
  uint64_t out = 0, g = 0;
  if (__popcll(word) < 8) {
    // sparse word (the usual case): the whole wave works on one pair at a time
    uint64_t rest = word;
    while (rest) {
      const int b = __ffsll((unsigned long long)rest) - 1;
      rest &= rest - 1;
      const int64_t j = jt * 64 + b;
      double r, m;
      pair_exact_wave(Xs, Npad, A, i, j, lane, r, m);
      bool sim = (r < max_rmsd) && (m < max_dev);
      const bool grey = (fabs(r - max_rmsd) < 1e-9) || (r < max_rmsd && fabs(m - max_dev) < 1e-9);
      if (energies != nullptr) sim = sim && (fabs(energies[i] - energies[j]) < max_dE);
      if (sim) out |= 1ull << b;
      if (grey) g |= 1ull << b;
    }
  } else {
    // dense word: one lane per pair
    const int64_t j = jt * 64 + lane;
    const bool cand = (word >> lane) & 1ull;
    bool sim = false, grey = false;
    if (cand) {
      double r, m;
      pair_exact(Xs, Npad, A, i, j, r, m);
      sim = (r < max_rmsd) && (m < max_dev);
      grey = (fabs(r - max_rmsd) < 1e-9) || (r < max_rmsd && fabs(m - max_dev) < 1e-9);
      if (energies != nullptr) sim = sim && (fabs(energies[i] - energies[j]) < max_dE);
    }
    out = __ballot(sim);
    g = __ballot(grey);
  }
  if (lane == 0) {
    bits[lrow * W + jt] = out;
    atomicAdd(&counters[1], (unsigned long long)__popcll(word));
    atomicAdd(&counters[2], (unsigned long long)__popcll(out));
    if (g) atomicAdd(&counters[3], (unsigned long long)__popcll(g));
  }
  }
}

// ---------------------------------------------------------------------------
// k_align_to_first: a8 align_structures.  One lane per conformer t: fit
// subset idx (n_idx atoms) centred, Kabsch onto conformer 0, rotate the whole
// structure about the subset centroid.  AoS in / AoS out.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_align_to_first(const double *__restrict__ coords, int64_t N, int64_t A,
                 const int64_t *__restrict__ idx, int64_t n_idx, double *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const double *ref = coords;
  const double *tgt = coords + t * A * 3;
  double rc[3] = {0, 0, 0}, tc[3] = {0, 0, 0};
  for (int64_t k = 0; k < n_idx; ++k) {
    const int64_t a = idx ? idx[k] : k;
    for (int c = 0; c < 3; ++c) {
      rc[c] += ref[a * 3 + c];
      tc[c] += tgt[a * 3 + c];
    }
  }
  for (int c = 0; c < 3; ++c) {
    rc[c] /= (double)n_idx;
    tc[c] /= (double)n_idx;
  }
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (t != 0) {
    double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = 0; k < n_idx; ++k) {
      const int64_t a = idx ? idx[k] : k;
      const double px = ref[a * 3 + 0] - rc[0], py = ref[a * 3 + 1] - rc[1], pz = ref[a * 3 + 2] - rc[2];
      const double qx = tgt[a * 3 + 0] - tc[0], qy = tgt[a * 3 + 1] - tc[1], qz = tgt[a * 3 + 2] - tc[2];
      B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
      B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
      B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
    }
    (void)kabsch_rotation(B, R);
  }
  double *o = out + t * A * 3;
  for (int64_t a = 0; a < A; ++a) {
    const double x = tgt[a * 3 + 0] - tc[0], y = tgt[a * 3 + 1] - tc[1], z = tgt[a * 3 + 2] - tc[2];
    o[a * 3 + 0] = R[0] * x + R[1] * y + R[2] * z;
    o[a * 3 + 1] = R[3] * x + R[4] * y + R[5] * z;
    o[a * 3 + 2] = R[6] * x + R[7] * y + R[8] * z;
  }
}

// a9: batched get_alignment_matrix on AoS pairs p[k], q[k] of A atoms
__global__ void __launch_bounds__(64)
k_alignment_matrices(const double *__restrict__ p, const double *__restrict__ q, int64_t n,
                     int64_t A, double *__restrict__ M) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double *pp = p + k * A * 3, *qq = q + k * A * 3;
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t a = 0; a < A; ++a) {
    const double px = pp[a * 3], py = pp[a * 3 + 1], pz = pp[a * 3 + 2];
    const double qx = qq[a * 3], qy = qq[a * 3 + 1], qz = qq[a * 3 + 2];
    B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
  }
  double R[9];
  (void)kabsch_rotation(B, R);
  for (int e = 0; e < 9; ++e) M[k * 9 + e] = R[e];
}

// exact (explicit rotated difference) rmsd for the queued small-rmsd pairs of the
// VALUES kernel: eight lanes per pair, overwrites the Newton value
__global__ void __launch_bounds__(256)
k_rmsd_fix_small(const double *__restrict__ Xa, int A, int64_t N, const uint64_t *__restrict__ pairq,
                 const unsigned long long *__restrict__ counters, unsigned long long Q,
                 double *__restrict__ rmsd_out, double *__restrict__ maxdev_out) {
  const int lane = threadIdx.x & 63, sub = lane & 7, slot = lane >> 3;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  unsigned long long n = counters[6];
  if (n > Q) n = Q;  // overflow is reported by the host wrapper
  for (int64_t base = wave0 * 8; base < (int64_t)n; base += nwaves * 8) {
    const int64_t p = base + slot;
    if (p < (int64_t)n) {
      const uint64_t e = pairq[p];
      const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
      double r, m;
      pair_exact_group8(Xa, A, i, j, sub, r, m);
      if (sub == 0) {
        rmsd_out[i * N + j] = r;
        if (maxdev_out) maxdev_out[i * N + j] = m;
      }
    }
  }
}

// full bit matrix from a list of similar pairs ((i << 32) | j, i < j): the
// receiving side of the multi-GPU exchange
__global__ void __launch_bounds__(256)
k_scatter_pairs(const uint64_t *__restrict__ pairs, int64_t n_pairs, int64_t N, int64_t W,
                uint64_t *__restrict__ bits) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const uint64_t e = pairs[p];
  const int64_t i = (int64_t)(e >> 32), j = (int64_t)(e & 0xffffffffull);
  if (i < 0 || j <= i || j >= N) return;  // padding / malformed entries are ignored
  atomicOr(reinterpret_cast<unsigned long long *>(&bits[i * W + (j >> 6)]), 1ull << (j & 63));
}

// Items of the MFMA screen = (local row block, column tile) pairs that touch the upper
// triangle, in dispatch order (row blocks ascending, tiles left to right); built on the host
// and kept on the device until N or the sharding changes.  Without it (row blocks that are not
// a multiple of 64) the kernel enumerates all NT x n_lblocks pairs and skips the empty ones.
static int screen_item_table(fc_ensemble *e, int64_t NT, int64_t n_lblocks, bool halves = true, int64_t tc = 64) {
  const int64_t rb_key = (e->row_block * 2 + (halves ? 1 : 0)) * 128 + tc;
  if (e->item_key[0] == e->N && e->item_key[1] == e->rank && e->item_key[2] == e->world &&
      e->item_key[3] == rb_key)
    return FC_OK;
  e->item_total = 0;
  e->item_key[0] = e->N; e->item_key[1] = e->rank; e->item_key[2] = e->world; e->item_key[3] = rb_key;
  if (e->row_block % tc != 0) return FC_OK;
  const int64_t r = e->row_block / tc;
  std::vector<uint64_t> items;
  for (int64_t l = 0; l < n_lblocks; ++l) {
    const int64_t first = r * global_block(l, e->rank, e->world);
    for (int64_t jt = first; jt < NT; ++jt) items.push_back(((uint64_t)l << 32) | (uint64_t)jt);
  }
  // the last items of the launch as two half-row-block items each: workgroups finish within
  // half an item of each other instead of a whole one (FC_SCREEN_TAIL_SLOTS items, default
  // one round of resident workgroups; needs 8 row tiles per block and 4-wave workgroups)
  {
    int64_t tail = 2 * (int64_t)ctx().n_cu;
    if (const char *v = getenv("FC_SCREEN_TAIL_SLOTS")) tail = std::strtoll(v, nullptr, 10);
    if (halves && e->row_block == 128 && tail > 0 && (int64_t)items.size() > 4 * tail && e->A <= 52) {
      std::vector<uint64_t> halves;
      for (int64_t k = (int64_t)items.size() - tail; k < (int64_t)items.size(); ++k) {
        halves.push_back(items[(size_t)k] | (1ull << 31));
        halves.push_back(items[(size_t)k] | (1ull << 63));
      }
      items.resize(items.size() - (size_t)tail);
      items.insert(items.end(), halves.begin(), halves.end());
    }
  }
  if (items.empty() || items.size() >= (1ull << 31)) return FC_OK;
  FC_TRY(e->item_table.reserve(items.size() * sizeof(uint64_t)));
  e->item_host.swap(items);  // the copy is asynchronous: its source stays with the ensemble (no host wait here)
  FC_TRY(h2d(e->item_table.p, e->item_host.data(), e->item_host.size() * sizeof(uint64_t)));
  e->item_total = (int64_t)e->item_host.size();
  return FC_OK;
}

// the item table the default screens of a prune will ask for (64-column tiles, half items at the tail), built and sent
// ahead of the launch; a screen that wants another table rebuilds it
int prebuild_screen_items(fc_ensemble *e) {
  const int64_t n_lblocks = local_block_count(ceil_div(e->N, e->row_block), e->rank, e->world);
  if (n_lblocks <= 0 || (e->Npad >> 6) == 0) return FC_OK;
  return screen_item_table(e, e->Npad >> 6, n_lblocks, /*halves=*/true, 64);
}

// all-pairs RMSD values on the matrix pipe (world == 1 layout); rmsd_dev: (N, N), pre-zeroed.
// maxdev_dev != nullptr: the complete alignment of every pair -- (rmsd, maxdev) from the explicit
// rotated difference (MODE 2 of the kernel); pairs it could not rotate are redone by the fix-up.
// one instantiation of the complete-alignment kernel: LDS attribute + launch
template <int NW, int TC, bool EIG>
static int launch_complete_variant(fc_ensemble *e, dim3 grid, size_t lds_m, double A_small, int64_t rb, int64_t rank, int64_t world,
                                   unsigned long long *cnt, const uint64_t *item_table_dev, unsigned long long n_items,
                                   double *rmsd_dev, double *maxdev_dev) {
  if (lds_m > 64 * 1024) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(k_simbits_screen_mfma<NW, 2, TC, EIG>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
    if (err != hipSuccess) return set_error(FC_E_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(err));
  }
  hipLaunchKernelGGL((k_simbits_screen_mfma<NW, 2, TC, EIG>), grid, dim3(NW * 64), lds_m, ctx().stream, e->Xs.as<double>(),
                     e->G.as<double>(), e->N, e->Npad, (int)e->A, A_small, (int)rb, rank, world, nullptr, e->W, nullptr, cnt,
                     e->pairq.as<uint64_t>(), (unsigned long long)e->pairq_cap, item_table_dev, n_items, rmsd_dev, nullptr,
                     maxdev_dev);
  return FC_OK;
}

// explicit_sum (complete alignments only): the rmsd from the running sum of the atom pass instead of the eigenvalue -- the
// caller's retry when the eigenvalue form sent more near-duplicate pairs to the fix-up than its queue holds
int launch_rmsd_values(fc_ensemble *e, double small_rmsd, double *rmsd_dev, double *maxdev_dev, int64_t rank,
                       int64_t world, bool explicit_sum) {
  // rank / world: this launch covers the row blocks (of 128) dealt to `rank` in snake order -- the rows a
  // rank of the multi-GPU bench owns; (0, 1) = the whole upper triangle
  const int64_t rb = 128;
  const int64_t n_lblocks = local_block_count(ceil_div(e->N, rb), rank, world);
  if (n_lblocks <= 0) return FC_OK;
  const size_t lds_64 = ((size_t)((e->A + 3) / 4) * 4 * 3 * 64 + 64 + (size_t)rb) * sizeof(double) + kStageBytes;
  const size_t lds_32 = ((size_t)((e->A + 3) / 4) * 4 * 3 * 32 + 32 + (size_t)rb) * sizeof(double) + kStageBytes;
  const size_t lds_16 = ((size_t)(((e->A + 3) / 4 + 1) / 2) * 384 + 16 + (size_t)rb) * sizeof(double) + kStageBytes;
  // the 64-column tile no longer fits (105 atoms and more): the complete alignments go on with 32 columns, from 209 atoms with 16
  const bool narrow = lds_64 > kLdsLimit && maxdev_dev != nullptr;
  const int64_t tc = !narrow ? 64 : (lds_32 <= kLdsLimit ? 32 : 16);
  const size_t lds_m = tc == 64 ? lds_64 : (tc == 32 ? lds_32 : lds_16);
  const int64_t NT = e->Npad / tc;
  if (lds_m > kLdsLimit || NT == 0)
    return set_error(FC_E_LIMIT, "A=%lld atoms exceed the LDS column tile of the value kernel", (long long)e->A);
  const bool fits32 = (uint64_t)((e->A + 3) / 4 * 4) * 3 * (uint64_t)e->Npad < (1ull << 32) &&
                      96 * (uint64_t)e->Npad < (1ull << 32);  // (byte offsets inside one k-step; the atom pass of MODE 2: of three rows)
  if (!fits32) return set_error(FC_E_LIMIT, "ensemble too large for 32-bit operand offsets");
  auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
  const bool two_blocks = 2 * lds_m <= kLdsLimit;
  const bool complete = maxdev_dev != nullptr;
  const void *fn = two_blocks ? reinterpret_cast<const void *>(k_simbits_screen_mfma<4, 1>)
                              : reinterpret_cast<const void *>(k_simbits_screen_mfma<8, 1>);
  if (!complete && lds_m > 64 * 1024) {
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
    if (err != hipSuccess) return set_error(FC_E_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(err));
  }
  const double A_small = (double)e->A * small_rmsd * small_rmsd;
  // one workgroup per upper-triangle item (see launch_simbits_screen); the value kernel uses
  // the world == 1 layout with its own row block, so it keeps its own table
  const int64_t saved_rank = e->rank, saved_world = e->world, saved_rb = e->row_block;
  e->rank = rank; e->world = world; e->row_block = rb;
  const int rc_tbl = screen_item_table(e, NT, n_lblocks, true, tc);
  e->rank = saved_rank; e->world = saved_world; e->row_block = saved_rb;
  e->item_key[3] = -1;  // the table was built for rb, not for the ensemble's own sharding
  if (rc_tbl != FC_OK) return rc_tbl;
  const int compact = e->item_total > 0 ? 1 : 0;
  const unsigned long long n_items =
      compact ? (unsigned long long)e->item_total : (unsigned long long)NT * (unsigned long long)n_lblocks;
  const uint64_t *item_table_dev = compact ? e->item_table.as<uint64_t>() : nullptr;
  if (n_items >= (1ull << 31)) return set_error(FC_E_LIMIT, "too many value-kernel items for one launch");
  const dim3 grid((unsigned)n_items);
#define FC_LAUNCH_VALUES(NW_, MODE_)                                                                      \
  hipLaunchKernelGGL((k_simbits_screen_mfma<NW_, MODE_>), grid, dim3(NW_ * 64), lds_m, ctx().stream,      \
                     e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_small, (int)rb,   \
                     rank, world, nullptr, e->W, nullptr, cnt, e->pairq.as<uint64_t>(),                   \
                     (unsigned long long)e->pairq_cap, item_table_dev, n_items, rmsd_dev, nullptr, maxdev_dev)
  const char *eig_env = getenv("FC_COMPLETE_EIG");  // 0: always the running sum (read per call: the tests compare both forms)
  const bool eig = complete && !(eig_env && atoi(eig_env) == 0) && !explicit_sum;
#define FC_LAUNCH_COMPLETE(NW_, TC_)                                                                                        \
  FC_TRY((eig ? launch_complete_variant<NW_, TC_, true>(e, grid, lds_m, A_small, rb, rank, world, cnt, item_table_dev, n_items, \
                                                       rmsd_dev, maxdev_dev)                                               \
              : launch_complete_variant<NW_, TC_, false>(e, grid, lds_m, A_small, rb, rank, world, cnt, item_table_dev,     \
                                                        n_items, rmsd_dev, maxdev_dev)))
  if (narrow && tc == 32) {
    FC_LAUNCH_COMPLETE(8, 32);
  } else if (narrow) {
    FC_LAUNCH_COMPLETE(8, 16);
  } else if (complete) {
    if (two_blocks) FC_LAUNCH_COMPLETE(4, 64);
    else FC_LAUNCH_COMPLETE(8, 64);
  } else {
    if (two_blocks) FC_LAUNCH_VALUES(4, 1);
    else FC_LAUNCH_VALUES(8, 1);
  }
#undef FC_LAUNCH_COMPLETE
#undef FC_LAUNCH_VALUES
  FC_TRY(check_launch("k_simbits_screen_mfma<values>"));
  if (ctx().mark_after_screen) (void)hipEventRecord(ctx().mark_after_screen, ctx().stream);  // bench hook: end of the tiled kernel
  hipLaunchKernelGGL(k_rmsd_fix_small, dim3((unsigned)(ctx().n_cu * 4)), dim3(256), 0, ctx().stream,
                     e->Xa.as<double>(), (int)e->A, e->N, e->pairq.as<uint64_t>(), cnt,
                     (unsigned long long)e->pairq_cap, rmsd_dev, maxdev_dev);
  return check_launch("k_rmsd_fix_small");
}

// elements (i, j) of the two dense (N, N) outputs of the complete-alignment pass, for the checkers
// (bench value_check, full-size parity tests): the kernel writes the upper triangle + diagonal only,
// so (i, j) with i > j reads (j, i)
__global__ void __launch_bounds__(256)
k_gather_matrix_pairs(const double *__restrict__ rmsd_m, const double *__restrict__ maxdev_m, int64_t N,
                      const int64_t *__restrict__ pi, const int64_t *__restrict__ pj, int64_t P,
                      double *__restrict__ rmsd_out, double *__restrict__ maxdev_out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int64_t i = pi[p], j = pj[p];
  if (i > j) { const int64_t t = i; i = j; j = t; }
  rmsd_out[p] = rmsd_m[i * N + j];
  maxdev_out[p] = maxdev_m[i * N + j];
}

int launch_gather_matrix_pairs(const double *rmsd_m, const double *maxdev_m, int64_t N, const int64_t *pi_dev,
                               const int64_t *pj_dev, int64_t P, double *rmsd_out, double *maxdev_out) {
  if (P == 0) return FC_OK;
  hipLaunchKernelGGL(k_gather_matrix_pairs, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, ctx().stream, rmsd_m,
                     maxdev_m, N, pi_dev, pj_dev, P, rmsd_out, maxdev_out);
  return check_launch("k_gather_matrix_pairs");
}

// lower triangle := upper triangle of an (N, N) matrix, in place: 32 x 32 tiles through LDS (the host's element loop over
// two 800 MB matrices took ~0.1 s of every fc_ensemble_rmsd_and_max_all at 10^4 conformers)
__global__ void __launch_bounds__(256)
k_mirror_upper(double *__restrict__ m, int64_t N) {
  __shared__ double tile[32][33];
  const int64_t bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;  // block-uniform: tiles on and above the diagonal are the sources
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = bi * 32 + r, j = bj * 32 + tx;
    tile[r][tx] = (i < N && j < N) ? m[i * N + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t j = bj * 32 + r, i = bi * 32 + tx;  // element (j, i) := (i, j)
    if (i < N && j < N && j > i) m[j * N + i] = tile[tx][r];
  }
}

int launch_mirror_upper(double *m_dev, int64_t N) {
  if (N < 2) return FC_OK;
  const unsigned nb = (unsigned)ceil_div(N, 32);
  hipLaunchKernelGGL(k_mirror_upper, dim3(nb, nb), dim3(256), 0, ctx().stream, m_dev, N);
  return check_launch("k_mirror_upper");
}

int launch_scatter_pairs(const uint64_t *pairs_dev, int64_t n_pairs, int64_t N, int64_t W,
                         uint64_t *bits_dev) {
  if (n_pairs == 0) return FC_OK;
  hipLaunchKernelGGL(k_scatter_pairs, dim3((unsigned)ceil_div(n_pairs, 256)), dim3(256), 0,
                     ctx().stream, pairs_dev, n_pairs, N, W, bits_dev);
  return check_launch("k_scatter_pairs");
}

// ---------------------------------------------------------------------------
// host-side launchers (called from fc_api.cpp)
// ---------------------------------------------------------------------------
// the largest G lands in the last counter word (zeroed here); ensemble_build reads it behind its own wait
int launch_prep_begin(fc_ensemble *e) {
  auto *gmax_bits = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
  FC_HIP_TRY(hipMemsetAsync(gmax_bits, 0, sizeof(unsigned long long), ctx().stream));
  e->xsf_valid = false;
  e->xh_valid = false;
  e->g_max = -1.0;
  return FC_OK;
}
// whether the tile kernel (64 conformers per workgroup through LDS) applies: only it can take a range of tiles
bool prep_by_tiles(int64_t A_all) {
  const size_t lds_tile = (size_t)kPrepTile * (size_t)(A_all * 3 + 1) * sizeof(double) + (size_t)A_all * sizeof(int) + 8;
  return lds_tile + 2048 <= kLdsLimit && (uint64_t)kPrepTile * (uint64_t)(A_all * 3 + 1) < (1ull << 31) &&
         !getenv("FC_PREP_LANES");  // FC_PREP_LANES=1: the one-lane-per-conformer kernel
}
// tiles [tile0, tile0 + n_tiles) of 64 conformers (behind launch_prep_begin)
int launch_prep_tiles(const double *coords_dev, int64_t N, int64_t A_all, const int32_t *sel_dev, int64_t A, int center,
                      fc_ensemble *e, const int32_t *conf_idx_dev, int64_t tile0, int64_t n_tiles) {
  if (n_tiles <= 0) return FC_OK;
  auto *gmax_bits = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
  // (tile0, n_tiles count 64-conformer tiles; the kernel takes kPrepTile conformers per workgroup)
  constexpr int64_t kPer64 = 64 / kPrepTile;
  const size_t lds_tile = (size_t)kPrepTile * (size_t)(A_all * 3 + 1) * sizeof(double) + (size_t)A_all * sizeof(int) + 8;  // tile + selection
  if (lds_tile > 64 * 1024)
    FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep_tile<kPrepTile>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
  hipLaunchKernelGGL(k_prep_tile<kPrepTile>, dim3((unsigned)(n_tiles * kPer64)), dim3(256), lds_tile, ctx().stream, coords_dev, N,
                     A_all, sel_dev, A, center, e->Npad, e->Xs.as<double>(), e->G.as<double>(), e->Xa.as<double>(), gmax_bits,
                     conf_idx_dev, tile0 * kPer64);
  return check_launch("k_prep_tile");
}
int launch_prep_body(const double *coords_dev, int64_t N, int64_t A_all, const int32_t *sel_dev, int64_t A, int center,
                     fc_ensemble *e, const int32_t *conf_idx_dev);
int launch_prep(const double *coords_dev, int64_t N, int64_t A_all, const int32_t *sel_dev,
                int64_t A, int center, fc_ensemble *e, const int32_t *conf_idx_dev) {
  FC_TRY(launch_prep_begin(e));
  return launch_prep_body(coords_dev, N, A_all, sel_dev, A, center, e, conf_idx_dev);
}
// (behind launch_prep_begin)
int launch_prep_body(const double *coords_dev, int64_t N, int64_t A_all, const int32_t *sel_dev, int64_t A, int center,
                     fc_ensemble *e, const int32_t *conf_idx_dev) {
  if (prep_by_tiles(A_all))
    return launch_prep_tiles(coords_dev, N, A_all, sel_dev, A, center, e, conf_idx_dev, 0, e->Npad / 64);
  auto *gmax_bits = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
  const int64_t blocks = ceil_div(e->Npad, 256);
  hipLaunchKernelGGL(k_prep, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, coords_dev, N,
                     A_all, sel_dev, A, center, e->Npad, e->Xs.as<double>(), e->G.as<double>(),
                     e->Xa.as<double>(), gmax_bits, conf_idx_dev);
  return check_launch("k_prep");
}

int launch_pairs_exact(const fc_ensemble *e, const int64_t *pi_dev, const int64_t *pj_dev,
                       int64_t P, double *rmsd_dev, double *maxdev_dev, double *R_dev) {
  if (P == 0) return FC_OK;
  hipLaunchKernelGGL(k_pairs_exact, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, ctx().stream,
                     e->Xa.as<double>(), (int)e->A, pi_dev, pj_dev, P, rmsd_dev, maxdev_dev, R_dev);
  return check_launch("k_pairs_exact");
}

int launch_matrix_exact(const fc_ensemble *e, double *rmsd_dev, double *maxdev_dev) {
  const int64_t waves = e->N * (e->Npad >> 6);
  if (waves == 0) return FC_OK;
  hipLaunchKernelGGL(k_matrix_exact, dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0,
                     ctx().stream, e->Xs.as<double>(), e->N, e->Npad, (int)e->A, rmsd_dev,
                     maxdev_dev);
  return check_launch("k_matrix_exact");
}


// which screen the last launch_simbits_screen used: 16 (split-half f16 MFMA), 32 (fp32 MFMA), 64 (fp64 MFMA), 1 (VALU), 0 (none yet)
static int g_last_screen = 0;
int last_screen_kind() { return g_last_screen; }
// fc_screen_select: 0 = automatic (FC_SCREEN_F32 / FC_SCREEN_H2 / band estimate), 16 / 32 / 64 = that screen whatever the band
static int g_screen_forced = 0;
void screen_select(int kind) { g_screen_forced = kind; }

int h2_model_ok(bool *ok);  // fc_h2_check.hip

// makes e->Xh for the scale the split-half screen would use; scale_out = 0: the screen does not apply
static int ensure_h2_operands_upto(fc_ensemble *e, double *scale_out, int64_t max_ks2) {
  *scale_out = 0.0;
  const int64_t KS2 = (e->A + 31) / 32, A4 = (e->A + 3) / 4 * 4;
  if (KS2 > max_ks2 || (uint64_t)(24 * KS2) * (uint64_t)e->Npad >= (1ull << 32)) return FC_OK;
  if (!(e->g_max > 0.0) || !std::isfinite(e->g_max)) return FC_OK;
  // largest |coordinate| <= sqrt(g_max): scaled into [2^12, 2^13] (halfs reach 65504; s^4 stays in fp32)
  int ex = 0;
  (void)std::frexp(std::sqrt(e->g_max), &ex);  // sqrt(g_max) = m 2^ex, m in [0.5, 1)
  const double scale = std::ldexp(1.0, 13 - ex);
  if (!e->xh_valid || e->xh_scale != scale) {
    FC_TRY(e->Xh.reserve((size_t)KS2 * 24 * (size_t)e->Npad * 16));
    hipLaunchKernelGGL(k_f64_to_h2, dim3((unsigned)ceil_div(e->Npad, 256), (unsigned)(KS2 * 12)), dim3(256), 0, ctx().stream,
                       e->Xs.as<double>(), e->Npad, (int)A4, (int)KS2, scale, e->Xh.as<h8_t>());
    FC_TRY(check_launch("k_f64_to_h2"));
    e->xh_valid = true;
    e->xh_scale = scale;
  }
  *scale_out = scale;
  return FC_OK;
}
int ensure_h2_operands(fc_ensemble *e, double *scale_out) { return ensure_h2_operands_upto(e, scale_out, kH2MaxKS2); }

int launch_simbits_screen(fc_ensemble *e, double thr2_margin) {
  // Context::mark_after_screen: recorded once, right behind the main screen kernel of this launch
  bool marked = false;
  auto mark_main = [&]() {
    if (!marked && ctx().mark_after_screen) (void)hipEventRecord(ctx().mark_after_screen, ctx().stream);
    if (!marked && ctx().after_main_stream && ctx().after_main_event) {  // the rest of this launch: on the caller's other stream
      (void)hipEventRecord(ctx().after_main_event, ctx().stream);
      (void)hipStreamWaitEvent(ctx().after_main_stream, ctx().after_main_event, 0);
      ctx().stream = ctx().after_main_stream;
    }
    marked = true;
  };
  const int64_t NT = e->Npad >> 6;
  const int64_t n_gblocks = ceil_div(e->N, e->row_block);
  const int64_t n_lblocks = local_block_count(n_gblocks, e->rank, e->world);
  if (n_lblocks <= 0 || NT == 0) {
    mark_main();
    return FC_OK;
  }
  const double A_thr2 = (double)e->A * thr2_margin;
  const size_t lds = (size_t)e->A * 3 * 64 * sizeof(double);
  dim3 grid((unsigned)NT, (unsigned)n_lblocks);
  // tuning / evidence knob: "mfma" (default), "valu8x4", "valu4x8"
  const char *cfg = getenv("FC_SCREEN_CFG");
  const bool want_valu = cfg && std::strncmp(cfg, "valu", 4) == 0;
  const bool alt = cfg && std::strcmp(cfg, "valu4x8") == 0;
  // Structures of 193 ... 416 atoms (the poses of two or three docked molecules, firecode/embedder.py:1472-1474): the
  // split-half screen with a 32-column tile, lean launches only.  Until round 5 they went to the fp32 matrix pipe up to
  // 213 atoms (2.4 x the time per pair) and to the fp64 vector screen beyond (8 x).  No speculative mode here (no fp64
  // matrix-pipe screen stands behind it at these sizes): the band rule decides alone, as for 105 ... 192 atoms.
  {
    const int64_t KS2n = (e->A + 31) / 32;
    static const int h2_env_n = [] {
      const char *v = getenv("FC_SCREEN_H2");
      return v ? atoi(v) : 1;
    }();
    const char *f32_env_n = getenv("FC_SCREEN_F32");
    const size_t lds_n = (size_t)KS2n * 24 * 512 + (32 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32;
    if (!want_valu && e->lean && KS2n > kH2MaxKS2 && KS2n <= 13 && h2_env_n != 0 && (g_screen_forced == 0 || g_screen_forced == 16) &&
        !(f32_env_n && f32_env_n[0] == '0') && e->row_block % 32 == 0 && lds_n <= kLdsLimit &&
        (uint64_t)(24 * KS2n) * (uint64_t)e->Npad < (1ull << 32)) {
      if (e->g_max < 0.0) {
        auto *cnt_max = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        hipLaunchKernelGGL(k_max_nonneg, dim3((unsigned)std::min<int64_t>(ceil_div(e->Npad, 256), 256)), dim3(256), 0, ctx().stream,
                           e->G.as<double>(), e->Npad, cnt_max);
        FC_TRY(check_launch("k_max_nonneg"));
        unsigned long long bits_max = 0;
        FC_TRY(d2h(&bits_max, cnt_max, sizeof bits_max));
        FC_TRY(sync());
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        std::memcpy(&e->g_max, &bits_max, sizeof(double));
      }
      bool model_ok = false;
      FC_TRY(h2_model_ok(&model_ok));
      double scale_n = 0.0;
      if (model_ok) FC_TRY(ensure_h2_operands_upto(e, &scale_n, 13));
      const KabschF32Bounds bdn = kabsch_h2_bounds(KS2n);
      const double s2 = scale_n * scale_n;
      const float hthr = (float)(0.5 * A_thr2 * s2);
      static const double kBandMaxN = [] {
        const char *v = getenv("FC_SCREEN_BAND_MAX");
        const double x = v ? atof(v) : 4.0;
        return x > 0.0 ? x : 4.0;
      }();
      const double band = (double)bdn.p0 * 2.0 * e->g_max / (double)e->A;
      const bool band_ok = g_screen_forced == 16 || (f32_env_n && (f32_env_n[0] == '2' || f32_env_n[0] == '3')) || band <= kBandMaxN * thr2_margin;
      if (scale_n > 0.0 && std::isfinite(hthr) && hthr > 0.f && band_ok) {
        const int64_t NTn = e->Npad / 32;
        FC_TRY(screen_item_table(e, NTn, n_lblocks, /*halves=*/false, 32));
        unsigned long long n_items = (unsigned long long)NTn * (unsigned long long)n_lblocks;
        const uint64_t *item_table_dev = nullptr;
        if (e->item_total > 0) n_items = (unsigned long long)e->item_total, item_table_dev = e->item_table.as<uint64_t>();
        if (n_items >= (1ull << 31)) return set_error(FC_E_LIMIT, "too many screen items for one launch");
        auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
        const float tiny_floor = std::max(4.0f * hthr, (float)e->A);
#define FC_LAUNCH_H2N(KS2_)                                                                                                 \
  do {                                                                                                                      \
    const void *fn_ = reinterpret_cast<const void *>(k_simbits_screen_mfma_h2<KS2_, false, 32>);                            \
    FC_HIP_TRY(hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));                           \
    hipLaunchKernelGGL((k_simbits_screen_mfma_h2<KS2_, false, 32>), dim3((unsigned)n_items), dim3(256), lds_n, ctx().stream, \
                       e->Xh.as<h8_t>(), e->G.as<double>(), e->N, e->Npad, hthr, tiny_floor, (float)s2, bdn, (int)e->row_block, \
                       e->rank, e->world, nullptr, e->W, e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),              \
                       (unsigned long long)e->pairq_cap, item_table_dev, n_items);                                          \
  } while (0)
        switch (KS2n) {
          case 7: FC_LAUNCH_H2N(7); break;
          case 8: FC_LAUNCH_H2N(8); break;
          case 9: FC_LAUNCH_H2N(9); break;
          case 10: FC_LAUNCH_H2N(10); break;
          case 11: FC_LAUNCH_H2N(11); break;
          case 12: FC_LAUNCH_H2N(12); break;
          default: FC_LAUNCH_H2N(13); break;
        }
#undef FC_LAUNCH_H2N
        FC_TRY(check_launch("k_simbits_screen_mfma_h2<narrow>"));
        e->item_key[3] = -1;  // (the table holds 32-column items: the next launch of another screen rebuilds it)
        g_last_screen = 16;
        mark_main();
        return FC_OK;
      }
    }
  }
  // Structures whose band is too wide for the split-half bound (extended ones: radius of gyration beyond ~11-14 A at these
  // sizes) and whose 64-column fp32 tile does not fit the LDS (214 atoms and more): the fp32 matrix-pipe kernel with a
  // 32-column tile, lean single-stage launches, up to the size at which its own bound stops being small (p0 < 2e-3:
  // ~370 atoms).  They went to the fp64 vector screen (8-12 ms per 7.2e7 pairs at 224 ... 384 atoms).
  {
    const int64_t A4n = (e->A + 3) / 4 * 4;
    const char *f32_env_n = getenv("FC_SCREEN_F32");
    const size_t lds_wide = (size_t)A4n * 3 * 64 * sizeof(float) + (64 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32;
    const size_t lds_n = (size_t)A4n * 3 * 32 * sizeof(float) + (32 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32;
    const KabschF32Bounds bfn = kabsch_f32_bounds(A4n);
    if (!want_valu && e->lean && lds_wide > kLdsLimit && lds_n <= kLdsLimit && (g_screen_forced == 0 || g_screen_forced == 32) &&
        !(f32_env_n && f32_env_n[0] == '0') && e->row_block % 32 == 0 && bfn.p0 < 2.0e-3f &&
        (uint64_t)A4n * 3 * (uint64_t)e->Npad < (1ull << 32)) {
      if (e->g_max < 0.0) {
        auto *cnt_max = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        hipLaunchKernelGGL(k_max_nonneg, dim3((unsigned)std::min<int64_t>(ceil_div(e->Npad, 256), 256)), dim3(256), 0, ctx().stream,
                           e->G.as<double>(), e->Npad, cnt_max);
        FC_TRY(check_launch("k_max_nonneg"));
        unsigned long long bits_max = 0;
        FC_TRY(d2h(&bits_max, cnt_max, sizeof bits_max));
        FC_TRY(sync());
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        std::memcpy(&e->g_max, &bits_max, sizeof(double));
      }
      static const double kBandMaxF = [] {
        const char *v = getenv("FC_SCREEN_BAND_MAX");
        const double x = v ? atof(v) : 4.0;
        return x > 0.0 ? x : 4.0;
      }();
      const double band_f = (double)bfn.p0 * 2.0 * e->g_max / (double)e->A;
      const bool band_ok = g_screen_forced == 32 || (f32_env_n && (f32_env_n[0] == '2' || f32_env_n[0] == '3')) || band_f <= kBandMaxF * thr2_margin;
      if (band_ok && std::isfinite(e->g_max)) {
        if (!e->xsf_valid) {
          const int64_t n = A4n * 3 * e->Npad;
          FC_TRY(e->Xsf.reserve((size_t)n * sizeof(float)));
          hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream, e->Xs.as<double>(), n,
                             e->Xsf.as<float>());
          FC_TRY(check_launch("k_f64_to_f32"));
          FC_TRY(e->sub.reserve((size_t)e->Npad * 8 * sizeof(float)));
          hipLaunchKernelGGL(k_subset_stats, dim3((unsigned)ceil_div(e->Npad, 256)), dim3(256), 0, ctx().stream, e->Xs.as<double>(),
                             e->Npad, (int)e->A, e->sub.as<float>());
          FC_TRY(check_launch("k_subset_stats"));
          e->xsf_valid = true;
        }
        const int64_t NTn = e->Npad / 32;
        FC_TRY(screen_item_table(e, NTn, n_lblocks, /*halves=*/false, 32));
        unsigned long long n_items = (unsigned long long)NTn * (unsigned long long)n_lblocks;
        const uint64_t *item_table_dev = nullptr;
        if (e->item_total > 0) n_items = (unsigned long long)e->item_total, item_table_dev = e->item_table.as<uint64_t>();
        if (n_items >= (1ull << 31)) return set_error(FC_E_LIMIT, "too many screen items for one launch");
        auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
        const void *fn_ = reinterpret_cast<const void *>(k_simbits_screen_mfma_f32<4, false, false, 32>);
        FC_HIP_TRY(hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));
        hipLaunchKernelGGL((k_simbits_screen_mfma_f32<4, false, false, 32>), dim3((unsigned)n_items), dim3(256), lds_n, ctx().stream,
                           e->Xsf.as<float>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, bfn, (int)e->row_block, e->rank,
                           e->world, nullptr, e->W, e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                           (unsigned long long)e->pairq_cap, item_table_dev, n_items, nullptr, bfn, 0.f, nullptr, 0, 0);
        FC_TRY(check_launch("k_simbits_screen_mfma_f32<narrow>"));
        e->item_key[3] = -1;  // (the table holds 32-column items: the next launch of another screen rebuilds it)
        g_last_screen = 32;
        mark_main();
        return FC_OK;
      }
    }
  }
  {
    const size_t lds_m = ((size_t)((e->A + 3) / 4) * 4 * 3 * 64 + 64 + (size_t)e->row_block) * sizeof(double) + kStageBytes;
    const bool fits32 = (uint64_t)((e->A + 3) / 4 * 4) * 3 * (uint64_t)e->Npad < (1ull << 32);
    const bool two_blocks_fit = 2 * lds_m <= kLdsLimit;
    // the fp64 matrix-pipe screen needs its column tile in LDS (up to ~100 atoms); the single-precision
    // screens reach further (fp32 tile: half the bytes; split-half: 128 atoms) and then run without the
    // fp64 screen behind them -- no speculative mode, and a band too wide for them means the VALU screen
    const bool mfma64_ok = fits32 && 96 * (uint64_t)e->Npad < (1ull << 32) /* row operands by 32-bit byte offsets */ &&
                           lds_m <= kLdsLimit && e->row_block % (two_blocks_fit ? 64 : 128) == 0;
    const int64_t A4s = (e->A + 3) / 4 * 4, KS2s = (e->A + 31) / 32;
    const bool single_ok = fits32 && e->row_block % 64 == 0 &&
                           ((size_t)A4s * 3 * 64 * sizeof(float) + (64 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32 <= kLdsLimit ||
                            (KS2s <= kH2MaxKS2 && (size_t)KS2s * 24 * 1024 + (64 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32 <= kLdsLimit));
    bool done = true;  // false: the matrix-pipe path declined, the VALU screen below takes the launch
    auto mfma_path = [&]() -> int {
      auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
      const bool two_blocks = 2 * lds_m <= kLdsLimit;
      const void *fn = two_blocks ? reinterpret_cast<const void *>(k_simbits_screen_mfma<4>)
                                  : reinterpret_cast<const void *>(k_simbits_screen_mfma<8>);
      if (mfma64_ok && lds_m > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
        if (err != hipSuccess)
          return set_error(FC_E_HIP, "hipFuncSetAttribute(LDS=%zu) failed: %s", lds_m,
                           hipGetErrorString(err));
      }
      if (mfma64_ok && getenv("FC_DEBUG")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, two_blocks ? 256 : 512, lds_m);
        fprintf(stderr, "[fc] screen_mfma<%d>: LDS %zu B, occupancy API says %d blocks/CU\n",
                two_blocks ? 4 : 8, lds_m, nb);
      }
      // one workgroup per item; world == 1: only the items that touch the upper triangle
      unsigned long long n_items = (unsigned long long)NT * (unsigned long long)n_lblocks;
      const uint64_t *item_table_dev = nullptr;
      static int use_v2 = -1;
      if (use_v2 < 0) {
        const char *v = getenv("FC_SCREEN_V2");
        use_v2 = (v && v[0] == '1') ? 1 : 0;
      }
      const bool want_v2 = use_v2 && mfma64_ok && two_blocks && e->row_block == 128;
      FC_TRY(screen_item_table(e, NT, n_lblocks, /*halves=*/!want_v2));
      if (e->item_total > 0) {
        n_items = (unsigned long long)e->item_total;
        item_table_dev = e->item_table.as<uint64_t>();
      }
      if (n_items >= (1ull << 31)) return set_error(FC_E_LIMIT, "too many screen items for one launch");
      const dim3 mgrid((unsigned)n_items);
      double *dbg = nullptr;
      const unsigned long long *gate = nullptr;  // set behind a speculative fp32 screen
#ifdef FC_TIMELINE
      static DevBuf tlbuf;
      const bool timeline = two_blocks && getenv("FC_TIMELINE_OUT") != nullptr;
      if (timeline) {
        FC_TRY(tlbuf.reserve(n_items * 4 * sizeof(unsigned long long)));
        FC_HIP_TRY(hipMemsetAsync(tlbuf.p, 0, n_items * 4 * sizeof(unsigned long long), ctx().stream));
        dbg = tlbuf.as<double>();
      }
#endif
      if (want_v2 && e->item_total > 0 && dbg == nullptr) {
        // row-sweep schedule: plain item list (no half items), one resident workgroup per CU
        const size_t tile = (size_t)((e->A + 3) / 4) * 4 * 3 * 64 * sizeof(double);
        const size_t lds2 = 2 * tile + (size_t)(2 * 64 + 128) * sizeof(double) + kStageBytes;
        if (lds2 <= kLdsLimit) {
          FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen_rowsweep),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
          const unsigned wgs = (unsigned)std::min<unsigned long long>((unsigned long long)ctx().n_cu,
                                                                      (unsigned long long)e->item_total);
          hipLaunchKernelGGL(k_screen_rowsweep, dim3(wgs), dim3(512), lds2, ctx().stream, e->Xs.as<double>(),
                             e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, e->rank, e->world,
                             e->bits.as<uint64_t>(), e->W, e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                             (unsigned long long)e->pairq_cap, e->item_table.as<uint64_t>(),
                             (unsigned long long)e->item_total);
          g_last_screen = 64;
          mark_main();
          return check_launch("k_screen_rowsweep");
        }
      }
      // Default: the single-precision screen (fp32 MFMA + bounded fp32 polynomial; candidates are
      // decided by the exact fp64 refine either way).  FC_SCREEN_F32=0, a timeline build's probe,
      // or so many atoms that the proven bounds stop being small select the fp64 screen below.
      const char *f32_env = g_screen_forced == 64 ? "0" : (g_screen_forced == 32 || g_screen_forced == 16) ? "2" : getenv("FC_SCREEN_F32");
      const int64_t A4 = (e->A + 3) / 4 * 4;
      auto ensure_gmax = [&]() -> int {
        if (e->g_max >= 0.0) return FC_OK;
        auto *cnt_max = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        hipLaunchKernelGGL(k_max_nonneg, dim3((unsigned)std::min<int64_t>(ceil_div(e->Npad, 256), 256)), dim3(256), 0,
                           ctx().stream, e->G.as<double>(), e->Npad, cnt_max);
        FC_TRY(check_launch("k_max_nonneg"));
        unsigned long long bits_max = 0;
        FC_TRY(d2h(&bits_max, cnt_max, sizeof bits_max));
        FC_TRY(sync());
        FC_HIP_TRY(hipMemsetAsync(cnt_max, 0, sizeof(unsigned long long), ctx().stream));
        std::memcpy(&e->g_max, &bits_max, sizeof(double));
        return FC_OK;
      };
      // The split-half kernel (f16 matrix pipe at single-precision accuracy) takes the place of the fp32-MFMA
      // kernel where it applies: at most 128 atoms, a finite non-zero largest norm to take the scale from.
      // FC_SCREEN_H2=0 / fc_screen_select(32): the fp32-MFMA kernel; fc_screen_select(16): this one whatever the band.
      static const int h2_env = [] {
        const char *v = getenv("FC_SCREEN_H2");
        return v ? atoi(v) : 1;
      }();
      const int64_t KS2 = (e->A + 31) / 32;
      bool use_h2 = false;
      double h2_scale = 1.0;
      const bool f32_allowed = !(f32_env && f32_env[0] == '0') && dbg == nullptr && e->row_block % 64 == 0;
      if (!mfma64_ok && !f32_allowed) {
        done = false;
        return FC_OK;
      }
      if (f32_allowed && h2_env != 0 && g_screen_forced != 32 && KS2 <= kH2MaxKS2 &&
          (uint64_t)(24 * KS2) * (uint64_t)e->Npad < (1ull << 32)) {
        FC_TRY(ensure_gmax());
        bool model_ok = false;
        FC_TRY(h2_model_ok(&model_ok));  // this device's f16 matrix pipe behaves as kabsch_h2_bounds assumes (checked once)
        if (model_ok) {
          FC_TRY(ensure_h2_operands(e, &h2_scale));
          const double s2 = h2_scale * h2_scale;
          use_h2 = h2_scale > 0.0 && std::isfinite((float)(0.5 * A_thr2 * s2)) && (float)(0.5 * A_thr2 * s2) > 0.f;
        }
      }
      KabschF32Bounds bd = use_h2 ? kabsch_h2_bounds(KS2) : kabsch_f32_bounds(A4);
      const size_t lds_f32_tile = (size_t)A4 * 3 * 64 * sizeof(float) + (64 + (size_t)e->row_block) * sizeof(float) * 6 + kStageBytesF32;
      // (p0: 1.9e-3 at five, 2.2e-3 at six k-steps of the split-half kernel; the fp32 kernel reaches 2e-3 at ~370 atoms)
      bool use_f32 = f32_allowed && bd.p0 < (use_h2 ? 2.5e-3f : 2.0e-3f) && (use_h2 || lds_f32_tile <= kLdsLimit);
      if (g_screen_forced == 16 && !use_h2) return set_error(FC_E_INVALID, "fc_screen_select(16): the split-half screen does not apply to this ensemble");
      bool speculative = f32_env && f32_env[0] == '3';  // FC_SCREEN_F32=3: always with the verdict
      if (use_f32 && !(f32_env && (f32_env[0] == '2' || f32_env[0] == '3'))) {  // FC_SCREEN_F32=2 / 3: no matter how wide the band
        // Band of mean square deviations above the threshold that the bounded fp32 test cannot
        // rule out: the factor (L - lambda_max)/s = A (msd - thr2) / (2 s) of P has to clear
        // p0 / (product of the other three factors: 8 (f2+f3)(f1+f3)(f1+f2) for singular values
        // f_i s of B, 1.3 for a chain, 2.4 for a ball; 1 assumed).  Large structures with a
        // tight threshold make it wide -- many candidates for the exact refine -- so the fp64
        // screen takes those at once; in between, k_screen_verdict decides on the device.
        FC_TRY(ensure_gmax());
        const double band = (double)bd.p0 * 2.0 * e->g_max / (double)e->A;
        // Beyond kBandMax thresholds^2 the fp64 screen takes the launch at once.  (1.0 until round 4: 30 000 x 80 ensembles
        // whose skeleton happened to be stretched -- radius of gyration 10.7 instead of 7.4 A: band 0.265 against 0.127 A^2
        // at a threshold of 0.25 -- fell off that edge and took 11.7 instead of 2.3 ms per prune although not one of
        // their pairs lies in the band.  Whether the band is POPULATED is what k_screen_verdict measures on the device;
        // the bound only says how wide it is, and up to 4 thresholds^2 -- pairs below 2.2 x the rmsd threshold -- the
        // speculative launch is the better bet: a verdict against it costs the split-half screen once, ~ 1/5 of the
        // fp64 screen it then runs.)
        static const double kBandMax = [] {
          const char *v = getenv("FC_SCREEN_BAND_MAX");
          const double x = v ? atof(v) : 4.0;
          return x > 0.0 ? x : 4.0;
        }();
        use_f32 = band <= kBandMax * thr2_margin;
        // narrow band: not worth the verdict's ~10 us.  Nor is a small ensemble (all its pairs fit the short candidate queue,
        // N <= 512): whatever the band holds costs the exact refine less than the verdict and the gated launch behind it cost
        // every call (9 of the 150-180 us of a drop-in prune at FIRECODE's own sizes)
        const bool worth_verdict = (double)e->N * (double)e->N > 2.0 * (double)kRefineLanesMin;
        speculative = band > 0.1 * thr2_margin && worth_verdict;
        if (!use_f32 && use_h2 && g_screen_forced != 16 && lds_f32_tile <= kLdsLimit) {
          // the split-half bound is about twice the fp32 kernel's (66 u per 32 atoms against one u per atom): where its band
          // is too wide and the fp32 kernel's is not -- 160 stretched atoms -- the fp32 matrix pipe is still 2.5 x faster
          // than the vector screen that would take the launch otherwise
          const KabschF32Bounds bf = kabsch_f32_bounds(A4);
          const double band_f = (double)bf.p0 * 2.0 * e->g_max / (double)e->A;
          if (bf.p0 < 2.0e-3f && band_f <= kBandMax * thr2_margin) {
            use_h2 = false;
            bd = bf;
            use_f32 = true;
            speculative = band_f > 0.1 * thr2_margin && worth_verdict;
          }
        }
      }
      if (!mfma64_ok) {
        if (!use_f32) {  // no fp64 matrix-pipe screen to fall back on at this size
          done = false;
          return FC_OK;
        }
        speculative = false;
      }
      if (use_f32 && use_h2) {
        const size_t lds_h = (size_t)KS2 * 24 * 1024 + (64 + (size_t)e->row_block) * sizeof(float) + kStageBytesF32;
        if (lds_h > kLdsLimit) return set_error(FC_E_LIMIT, "split-half screen: row block of %lld rows does not fit LDS", (long long)e->row_block);
        const double s2 = h2_scale * h2_scale;
        const float hthr = (float)(0.5 * A_thr2 * s2);
        const float tiny_floor = std::max(4.0f * hthr, (float)e->A);
#define FC_LAUNCH_H2(KS2_, BITS_)                                                                                       \
  do {                                                                                                                  \
    const void *fn_ = reinterpret_cast<const void *>(k_simbits_screen_mfma_h2<KS2_, BITS_>);                            \
    if (lds_h > 64 * 1024) FC_HIP_TRY(hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h)); \
    hipLaunchKernelGGL((k_simbits_screen_mfma_h2<KS2_, BITS_>), mgrid, dim3(256), lds_h, ctx().stream, e->Xh.as<h8_t>(), \
                       e->G.as<double>(), e->N, e->Npad, hthr, tiny_floor, (float)s2, bd, (int)e->row_block, e->rank,   \
                       e->world, e->bits.as<uint64_t>(), e->W, e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),    \
                       (unsigned long long)e->pairq_cap, item_table_dev, n_items);                                      \
  } while (0)
#define FC_LAUNCH_H2_K(BITS_)                  \
  switch (KS2) {                               \
    case 1: FC_LAUNCH_H2(1, BITS_); break;     \
    case 2: FC_LAUNCH_H2(2, BITS_); break;     \
    case 3: FC_LAUNCH_H2(3, BITS_); break;     \
    case 4: FC_LAUNCH_H2(4, BITS_); break;     \
    case 5: FC_LAUNCH_H2(5, BITS_); break;     \
    default: FC_LAUNCH_H2(6, BITS_); break;    \
  }
#ifdef FC_H2_TIMELINE
        static DevBuf tlbuf;
        const bool h2_timeline = e->lean && getenv("FC_H2_TIMELINE_OUT") != nullptr;
        uint64_t *const bits_saved = e->bits.as<uint64_t>();
        if (h2_timeline) {
          FC_TRY(tlbuf.reserve(n_items * 4 * sizeof(unsigned long long)));
          FC_HIP_TRY(hipMemsetAsync(tlbuf.p, 0, n_items * 4 * sizeof(unsigned long long), ctx().stream));
          e->bits.p = tlbuf.p;
        } else if (e->lean) {
          e->bits.p = nullptr;
        }
#endif
        if (e->lean) {
          FC_LAUNCH_H2_K(false)
        } else {
          FC_LAUNCH_H2_K(true)
        }
#undef FC_LAUNCH_H2_K
#undef FC_LAUNCH_H2
#ifdef FC_H2_TIMELINE
        e->bits.p = bits_saved;
        if (h2_timeline) {
          std::vector<unsigned long long> h(n_items * 4);
          FC_HIP_TRY(hipMemcpyAsync(h.data(), tlbuf.p, n_items * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx().stream));
          FC_HIP_TRY(hipStreamSynchronize(ctx().stream));
          if (FILE *f = fopen(getenv("FC_H2_TIMELINE_OUT"), "wb")) {
            fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
            fclose(f);
          }
        }
#endif
        FC_TRY(check_launch("k_simbits_screen_mfma_h2"));
        mark_main();
        g_last_screen = 16;
      } else if (use_f32) {
        // Subset stage (lean prunes only).  Measured on one box, screen kernels alone on the chip:
        // 10^4 x 50: 0.432 ms (subset stage 0.31 + per-unit kernel 0.10 + verdict and gated launches)
        // against 0.462 ms single-stage; 3*10^4 x 80: 4.77 ms against 6.81 ms (two workgroups per CU there);
        // continuous RMSD distribution (dense: the sample's verdict sends everything to the single-stage
        // launch): 1.70 ms against 1.72 ms.  FC_SCREEN_STAGES=1: never; structures below four k-steps and
        // row blocks above 256 rows (no room for their unit list in the staging area) take the single stage.
        static const int stages_env = [] {
          const char *v = getenv("FC_SCREEN_STAGES");
          return v ? atoi(v) : 0;
        }();
        const bool two = e->lean && stages_env != 1 && A4 >= 16 && e->row_block <= 256;
        const size_t lds_f = (size_t)A4 * 3 * 64 * sizeof(float) + (64 + (size_t)e->row_block) * sizeof(float) * (two ? 6 : 1) +
                             kStageBytesF32;
        if (!e->xsf_valid) {
          const int64_t n = A4 * 3 * e->Npad;
          FC_TRY(e->Xsf.reserve((size_t)n * sizeof(float)));
          hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx().stream,
                             e->Xs.as<double>(), n, e->Xsf.as<float>());
          FC_TRY(check_launch("k_f64_to_f32"));
          FC_TRY(e->sub.reserve((size_t)e->Npad * 8 * sizeof(float)));
          hipLaunchKernelGGL(k_subset_stats, dim3((unsigned)ceil_div(e->Npad, 256)), dim3(256), 0, ctx().stream,
                             e->Xs.as<double>(), e->Npad, (int)e->A, e->sub.as<float>());
          FC_TRY(check_launch("k_subset_stats"));
          e->xsf_valid = true;
        }
        if (lds_f > 64 * 1024) {
          FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_simbits_screen_mfma_f32<4, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
          FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_simbits_screen_mfma_f32<4, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
          FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_simbits_screen_mfma_f32<4, false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
        }
        // stage 1 accumulates the even k-steps: A4_S atoms (padding included), A_S real ones; its bounds
        // carry 8 u more per entry for the rank-one centring term (two rounded factors, 1/A_S, the fma)
        const int64_t KS_all = A4 / 4, KS1 = (KS_all + 1) / 2;
        int64_t A_S = 0;
        for (int64_t a = 0; a < e->A; ++a) A_S += ((a >> 2) & 1) ? 0 : 1;
        const KabschF32Bounds bd1 = kabsch_f32_bounds(4 * KS1 + 8);
        const float inv_AS = (float)(1.0 / (double)std::max<int64_t>(A_S, 1));
        const float *sub_dev = two ? e->sub.as<float>() : nullptr;
        if (two) {
          // subset stage over all units -> queue of the units it could not rule out -> the full test for
          // those, one wavefront per unit; when more than ~a third of the units are queued (dense
          // similarity) the per-unit kernel steps aside and the single-stage tiled kernel redoes the launch
          const unsigned long long units_per_item = (unsigned long long)(e->row_block / 16) * 2ull;
          FC_TRY(e->unitq.reserve((size_t)(n_items * units_per_item) * sizeof(uint64_t)));
          const unsigned long long n_sample = (n_items + 15ull) / 16ull, n_rest = n_items - n_sample;
          // "dense": more than a third of the sampled units could not be ruled out by the subset stage
          const unsigned long long max_sample_units = n_sample * units_per_item / 3ull;
#define FC_LAUNCH_F32_LEAN(GRID, SUB, PART, GATE, STAGED_)                                                            \
  hipLaunchKernelGGL((k_simbits_screen_mfma_f32<4, false, STAGED_>), dim3((unsigned)(GRID)), dim3(256), lds_f, ctx().stream, \
                     e->Xsf.as<float>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, bd, (int)e->row_block,  \
                     e->rank, e->world, e->bits.as<uint64_t>(), e->W, e->cand.as<uint32_t>(), cnt,                    \
                     e->pairq.as<uint64_t>(), (unsigned long long)e->pairq_cap, item_table_dev, n_items, SUB, bd1,    \
                     inv_AS, e->unitq.as<uint64_t>(), PART, GATE)
          FC_LAUNCH_F32_LEAN(n_sample, sub_dev, 1, 0, true);
          hipLaunchKernelGGL(k_screen_density_verdict, dim3(1), dim3(64), 0, ctx().stream, cnt, max_sample_units);
          if (n_rest > 0) FC_LAUNCH_F32_LEAN(n_rest, sub_dev, 2, 1, true);
          FC_TRY(check_launch("k_simbits_screen_mfma_f32<subset stage>"));
          hipLaunchKernelGGL(k_screen_units_f32, dim3((unsigned)(ctx().n_cu * 3)), dim3(256), 0, ctx().stream,
                             e->Xsf.as<float>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, bd,
                             e->unitq.as<uint64_t>(), cnt, e->pairq.as<uint64_t>(), (unsigned long long)e->pairq_cap);
          FC_TRY(check_launch("k_screen_units_f32"));
          FC_LAUNCH_F32_LEAN(n_items, nullptr, 0, 2, false);
#undef FC_LAUNCH_F32_LEAN
        } else if (e->lean)
          hipLaunchKernelGGL((k_simbits_screen_mfma_f32<4, false>), mgrid, dim3(256), lds_f, ctx().stream,
                             e->Xsf.as<float>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, bd,
                             (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W,
                             e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                             (unsigned long long)e->pairq_cap, item_table_dev, n_items, nullptr, bd1, inv_AS, nullptr, 0, 0);
        else
          hipLaunchKernelGGL((k_simbits_screen_mfma_f32<4, true>), mgrid, dim3(256), lds_f, ctx().stream,
                             e->Xsf.as<float>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2, bd,
                             (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W,
                             e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                             (unsigned long long)e->pairq_cap, item_table_dev, n_items, nullptr, bd1, inv_AS, nullptr, 0, 0);
        FC_TRY(check_launch("k_simbits_screen_mfma_f32"));
        mark_main();
        g_last_screen = 32;
      }
      if (use_f32) {
        if (!speculative) return FC_OK;
        // speculative: the verdict kernel decides on the device whether the fp64 screen below
        // has to redo the launch (k_screen_verdict); its workgroups return at once otherwise
        const double owned_pairs = 0.5 * (double)e->N * (double)e->N / (double)e->world;
        // a false candidate costs ~1 ns (staging, one atom pass of the refine: tools/broad_probe.py), a second
        // screen in fp64 ~0.017 ns per owned pair at 52 padded atoms and in proportion to them beyond
        const auto max_false = (unsigned long long)std::max(1024.0, 0.015 * ((double)A4 / 52.0) * owned_pairs);
        hipLaunchKernelGGL(k_screen_verdict, dim3(4), dim3(256), 0, ctx().stream, e->Xa.as<double>(),
                           e->G.as<double>(), (int)e->A, A_thr2, e->pairq.as<uint64_t>(),
                           (unsigned long long)e->pairq_cap, max_false, cnt, ctx().optimistic_screen ? 1 : 0);
        FC_TRY(check_launch("k_screen_verdict"));
        if (ctx().optimistic_screen) return FC_OK;
        gate = cnt + 11;
      }
      if (two_blocks)
        hipLaunchKernelGGL(k_simbits_screen_mfma<4>, mgrid, dim3(256), lds_m, ctx().stream,
                           e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2,
                           (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W,
                           e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                           (unsigned long long)e->pairq_cap, item_table_dev, n_items, dbg, gate);
      else
        hipLaunchKernelGGL(k_simbits_screen_mfma<8>, mgrid, dim3(512), lds_m, ctx().stream,
                           e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2,
                           (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W,
                           e->cand.as<uint32_t>(), cnt, e->pairq.as<uint64_t>(),
                           (unsigned long long)e->pairq_cap, item_table_dev, n_items, dbg, gate);
      if (gate == nullptr) g_last_screen = 64;
      FC_TRY(check_launch("k_simbits_screen_mfma"));
      mark_main();
#ifdef FC_TIMELINE
      if (timeline) {
        std::vector<unsigned long long> h(n_items * 4);
        FC_HIP_TRY(hipMemcpyAsync(h.data(), tlbuf.p, n_items * 4 * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, ctx().stream));
        FC_HIP_TRY(hipStreamSynchronize(ctx().stream));
        if (FILE *f = fopen(getenv("FC_TIMELINE_OUT"), "wb")) {
          fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
          fclose(f);
        }
      }
#endif
      return FC_OK;
    };
    if (!want_valu && (mfma64_ok || single_ok)) {
      FC_TRY(mfma_path());
      if (done) return FC_OK;
    }
  }
#define FC_LAUNCH_SCREEN(LDSFLAG, TI_, NW_, SMEM)                                                   \
  hipLaunchKernelGGL((k_simbits_screen<LDSFLAG, TI_, NW_>), grid, dim3(NW_ * 64), SMEM,            \
                     ctx().stream, e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad,           \
                     (int)e->A, A_thr2, (int)e->row_block, e->rank, e->world,                      \
                     e->bits.as<uint64_t>(), e->W, e->cand.as<uint32_t>(),                         \
                     reinterpret_cast<unsigned long long *>(e->counters.p),                        \
                     e->pairq.as<uint64_t>(), (unsigned long long)e->pairq_cap)
  if (lds <= kLdsLimit) {
    const void *fn = alt ? reinterpret_cast<const void *>(k_simbits_screen<true, 4, 8>)
                         : reinterpret_cast<const void *>(k_simbits_screen<true, 8, 4>);
    if (lds > 64 * 1024) {
      hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (err != hipSuccess)
        return set_error(FC_E_HIP, "hipFuncSetAttribute(LDS=%zu) failed: %s", lds,
                         hipGetErrorString(err));
    }
    if (alt) FC_LAUNCH_SCREEN(true, 4, 8, lds);
    else FC_LAUNCH_SCREEN(true, 8, 4, lds);
  } else {
    if (alt) FC_LAUNCH_SCREEN(false, 4, 8, 0);
    else FC_LAUNCH_SCREEN(false, 8, 4, 0);
  }
#undef FC_LAUNCH_SCREEN
  g_last_screen = 1;
  mark_main();
  return check_launch("k_simbits_screen");
}

int launch_simbits_refine(fc_ensemble *e, double max_rmsd, double max_dev, const double *energies_dev,
                          double max_dE) {
  if (e->rows_local * e->W == 0) return FC_OK;
  // persistent-style grids: wavefronts stride over the candidate queues whose lengths the screen kernel
  // left in the counters (same stream, no host sync).  The pair queue, when it did not overflow, is taken by
  // k_refine_pairs (FC_REFINE_LANES=0: by the pair mode of k_simbits_refine, the kernel of rounds 1-2)
  static const bool lanes = [] {
    const char *v = getenv("FC_REFINE_LANES");
    return !(v && atoi(v) == 0);
  }();
  static const bool buckets = [] {
    const char *v = getenv("FC_REFINE_BUCKETS");  // 0: the straight queue walk of k_refine_pairs (round 3)
    return !(v && atoi(v) == 0);
  }();
  const size_t lds_bk = (size_t)(((e->A + 3) & ~(int64_t)3) * 3 * (1 << kBucketColShift)) * sizeof(double);
  // the column tile shares the CU's LDS with the kernel's own arrays (bins, sorted pairs, per-wavefront sums): 104 atoms
  // are 156 KB of tile and do not fit beside them (tools/refine_stress.py found that launch refused)
  static const size_t lds_bk_static = [] {
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_refine_buckets)) == hipSuccess ? (size_t)fa.sharedSizeBytes
                                                                                                        : (size_t)(16 * 1024);
  }();
  static const bool debug_form = getenv("FC_DEBUG") != nullptr;
  const bool bucket_form = lanes && buckets && e->bk_buckets > 0 && lds_bk + lds_bk_static <= kLdsLimit &&
                           e->last_candidates > (int64_t)kRefineLanesMin;
  if (debug_form)  // (which kernels a measurement ran: the form is chosen from the last prune seen, not from this one)
    fprintf(stderr, "[fc] refine form: %s (last prune seen: %lld candidates, %lld similar)\n",
            bucket_form ? "bucket sort + k_refine_buckets" : (lanes ? "k_refine_pairs / k_simbits_refine" : "k_simbits_refine"),
            (long long)e->last_candidates, (long long)e->last_similar);
  if (bucket_form) {
    auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
    const unsigned NT = (unsigned)(e->Npad >> kBucketColShift);
    const BucketGeom geom{NT, (NT + 7) / 8};
    const int64_t nb = e->bk_buckets;
    const int n_st = (int)(nb / 8), x_stride = (n_st + 7) / 8 + 1;
    int *bk = e->bk.as<int>();
    int *off = e->bk_off.as<int>(), *st_off = off + nb + 1, *xoff = st_off + n_st + 1;
    FC_HIP_TRY(hipMemsetAsync(bk, 0, (size_t)(kBkCtrl + 2 * nb + 1) * sizeof(int), ctx().stream));
    const unsigned g = (unsigned)(ctx().n_cu * 2);
    hipLaunchKernelGGL(k_bucket_count, dim3(g), dim3(256), 0, ctx().stream, e->pairq.as<uint64_t>(), cnt,
                       (unsigned long long)e->pairq_cap, geom, bk);
    FC_TRY(check_launch("k_bucket_count"));
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, ctx().stream, cnt, (unsigned long long)e->pairq_cap, nb, bk,
                       off, e->bk_list.as<int>(), st_off, xoff, n_st, x_stride);
    FC_TRY(check_launch("k_bucket_scan"));
    hipLaunchKernelGGL(k_bucket_scatter, dim3(g), dim3(256), 0, ctx().stream, e->pairq.as<uint64_t>(), cnt,
                       (unsigned long long)e->pairq_cap, geom, nb, bk, off, e->sortq.as<uint64_t>());
    FC_TRY(check_launch("k_bucket_scatter"));
    if (lds_bk > 64 * 1024)
      FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_refine_buckets),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bk));
    static const int per_cu_b = [] {
      const char *v = getenv("FC_REFINE_GRID");
      const int k = v ? atoi(v) : 2;
      return k >= 1 && k <= 8 ? k : 2;
    }();
    hipLaunchKernelGGL(k_refine_buckets, dim3((unsigned)(ctx().n_cu * per_cu_b)), dim3(kBucketChunk), lds_bk, ctx().stream,
                       e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, max_rmsd, max_dev, energies_dev,
                       max_dE, (int)e->row_block, e->world, e->lean ? nullptr : e->bits.as<uint64_t>(), e->W, cnt,
                       (unsigned long long)e->pairq_cap, geom, bk, off, e->bk_list.as<int>(), st_off, xoff, n_st, x_stride,
                       e->sortq.as<uint64_t>(), e->simq.as<uint64_t>());
    FC_TRY(check_launch("k_refine_buckets"));
  } else if (lanes && (double)e->N * (double)(e->N - 1) * 0.5 > (double)kRefineLanesMin) {
    // (an ensemble whose WHOLE pair matrix fits the short queue -- N <= 512 -- never has a long one: no launch that finds
    // nothing to do; k_simbits_refine below takes every queue of up to kRefineLanesMin pairs either way)
    static const int per_cu = [] {
      // workgroups per CU (tuning knob).  The kernel waits for the memory system, not for lanes: alone it takes 0.45 / 0.49 /
      // 0.47 ms for 9.45e5 pairs with 16 / 2 / 1 workgroups per CU; beside a screen, two per CU leave the screen the
      // registers of the other SIMD slots: the overlapped step 0.80 -> 0.77 ms (tools/refine_alone_probe.py)
      const char *v = getenv("FC_REFINE_GRID");
      const int k = v ? atoi(v) : 2;
      return k >= 1 && k <= 64 ? k : 2;
    }();
    hipLaunchKernelGGL(k_refine_pairs, dim3((unsigned)(ctx().n_cu * per_cu)), dim3(256), 0, ctx().stream,
                       e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, max_rmsd, max_dev, energies_dev,
                       max_dE, (int)e->row_block, e->world, e->lean ? nullptr : e->bits.as<uint64_t>(), e->W,
                       reinterpret_cast<unsigned long long *>(e->counters.p), e->pairq.as<uint64_t>(),
                       (unsigned long long)e->pairq_cap, e->simq.as<uint64_t>());
    FC_TRY(check_launch("k_refine_pairs"));
  }
  static const int per_cu8 = [] {
    const char *v = getenv("FC_REFINE_GRID8");  // workgroups per CU of the 8-lanes-per-pair kernel (tuning knob)
    const int k = v ? atoi(v) : 16;
    return k >= 1 && k <= 64 ? k : 16;
  }();
  hipLaunchKernelGGL(k_simbits_refine, dim3((unsigned)(ctx().n_cu * per_cu8)), dim3(256), 0,
                     ctx().stream, e->Xs.as<double>(), e->Xa.as<double>(), e->N, e->Npad, (int)e->A,
                     max_rmsd, max_dev, energies_dev, max_dE, lanes ? -(int)e->row_block : (int)e->row_block, e->rank, e->world,
                     e->rows_local,
                     e->lean ? nullptr : e->bits.as<uint64_t>(), e->W, e->cand.as<uint32_t>(),
                     reinterpret_cast<unsigned long long *>(e->counters.p), e->pairq.as<uint64_t>(),
                     (unsigned long long)e->pairq_cap, e->simq.as<uint64_t>());
  return check_launch("k_simbits_refine");
}

int launch_align_to_first(const double *coords_dev, int64_t N, int64_t A, const int64_t *idx_dev,
                          int64_t n_idx, double *out_dev) {
  hipLaunchKernelGGL(k_align_to_first, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, ctx().stream,
                     coords_dev, N, A, idx_dev, n_idx, out_dev);
  return check_launch("k_align_to_first");
}

int launch_alignment_matrices(const double *p_dev, const double *q_dev, int64_t n, int64_t A,
                              double *M_dev) {
  if (n == 0) return FC_OK;
  hipLaunchKernelGGL(k_alignment_matrices, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0,
                     ctx().stream, p_dev, q_dev, n, A, M_dev);
  return check_launch("k_alignment_matrices");
}

// fc_warmup(): the first launch from a translation unit makes the runtime load that unit's code object (milliseconds);
// a no-op launch moves that cost out of the first real call
__global__ void k_warm_kabsch() {}
int warm_kabsch() {
  hipLaunchKernelGGL(k_warm_kabsch, dim3(1), dim3(64), 0, ctx().stream);
  return check_launch("k_warm_kabsch");
}

}  // namespace fc
