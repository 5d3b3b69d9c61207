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
#include "fc_common.h"
#include "fc_kabsch_math.h"

namespace fc {

// ---------------------------------------------------------------------------
// k_prep: AoS (N, A_all, 3) -> conformer-minor SoA of the selected atoms,
// optional centring on the centroid of the selection, G[n].
// One lane per conformer: reads walk the conformer's own contiguous block
// (absorbed by L2), writes are coalesced across lanes.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_prep(const double *__restrict__ coords, int64_t N, int64_t A_all, const int32_t *__restrict__ sel,
       int64_t A, int center, int64_t Npad, double *__restrict__ Xs, double *__restrict__ G) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Npad) return;
  if (n >= N) {  // zero padding keeps every later load in bounds and finite
    for (int64_t a = 0; a < A; ++a) {
      Xs[(a * 3 + 0) * Npad + n] = 0.0;
      Xs[(a * 3 + 1) * Npad + n] = 0.0;
      Xs[(a * 3 + 2) * Npad + n] = 0.0;
    }
    G[n] = 0.0;
    return;
  }
  const double *src = coords + n * A_all * 3;
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
    g += x * x + y * y + z * z;
  }
  G[n] = g;
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

__global__ void __launch_bounds__(256)
k_pairs_exact(const double *__restrict__ Xs, int64_t Npad, int A, const int64_t *__restrict__ pi,
              const int64_t *__restrict__ pj, int64_t P, double *__restrict__ rmsd,
              double *__restrict__ maxdev, double *__restrict__ Rout) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double r, m;
  pair_exact(Xs, Npad, A, pi[p], pj[p], r, m, Rout ? Rout + p * 9 : nullptr);
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
constexpr int TI = 8;

template <bool USE_LDS>
__global__ void __launch_bounds__(256)
k_simbits_screen(const double *__restrict__ Xs, const double *__restrict__ G, int64_t N,
                 int64_t Npad, int A, double A_thr2, int IB, int64_t rank, int64_t world,
                 uint64_t *__restrict__ bits, int64_t W) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t jt = blockIdx.x;
  const int64_t lb = blockIdx.y;                 // local row block
  const int64_t gb = lb * world + rank;          // global row block
  const int64_t i0 = gb * IB;
  if (i0 >= N) return;
  if (jt * 64 + 63 <= i0) return;                // nothing above the diagonal here
  const int64_t j = jt * 64 + lane;

  if (USE_LDS) {
    const int total = A * 3 * 64;
    for (int idx = tid; idx < total; idx += 256) {
      const int ac = idx >> 6, l = idx & 63;
      lds[idx] = Xs[(int64_t)ac * Npad + jt * 64 + l];
    }
    __syncthreads();
  }
  const double Gq = G[j];
  const double *__restrict__ xq = Xs + jt * 64 + lane;

  for (int it = wv; it * TI < IB; it += 4) {
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
      if (lane == 0 && i < N) {
        const int64_t lrow = (int64_t)(lb * IB) + (int64_t)it * TI + k;
        bits[lrow * W + jt] = word;
      }
    }
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
k_simbits_refine(const double *__restrict__ Xs, int64_t N, int64_t Npad, int A, double max_rmsd,
                 double max_dev, const double *__restrict__ energies, double max_dE, int IB,
                 int64_t rank, int64_t world, int64_t rows_local, uint64_t *__restrict__ bits,
                 int64_t W, unsigned long long *__restrict__ counters) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t lrow = wave / W;
  const int64_t jt = wave % W;
  if (lrow >= rows_local) return;
  const int64_t lb = lrow / IB;
  const int64_t i = (lb * world + rank) * IB + (lrow % IB);
  if (i >= N) return;
  if (jt * 64 + 63 <= i) return;
  const uint64_t word = bits[lrow * W + jt];
  if (word == 0) return;
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
  const uint64_t out = __ballot(sim);
  const uint64_t g = __ballot(grey);
  if (lane == 0) {
    bits[lrow * W + jt] = out;
    atomicAdd(&counters[1], (unsigned long long)__popcll(word));
    atomicAdd(&counters[2], (unsigned long long)__popcll(out));
    if (g) atomicAdd(&counters[3], (unsigned long long)__popcll(g));
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

// ---------------------------------------------------------------------------
// host-side launchers (called from fc_api.cpp)
// ---------------------------------------------------------------------------
int launch_prep(const double *coords_dev, int64_t N, int64_t A_all, const int32_t *sel_dev,
                int64_t A, int center, fc_ensemble *e) {
  const int64_t blocks = ceil_div(e->Npad, 256);
  hipLaunchKernelGGL(k_prep, dim3((unsigned)blocks), dim3(256), 0, ctx().stream, coords_dev, N,
                     A_all, sel_dev, A, center, e->Npad, e->Xs.as<double>(), e->G.as<double>());
  return check_launch("k_prep");
}

int launch_pairs_exact(const fc_ensemble *e, const int64_t *pi_dev, const int64_t *pj_dev,
                       int64_t P, double *rmsd_dev, double *maxdev_dev, double *R_dev) {
  if (P == 0) return FC_OK;
  hipLaunchKernelGGL(k_pairs_exact, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, ctx().stream,
                     e->Xs.as<double>(), e->Npad, (int)e->A, pi_dev, pj_dev, P, rmsd_dev,
                     maxdev_dev, R_dev);
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

// LDS budget: a column tile is A*3*64*8 bytes; 160 KiB per CU on gfx950
static constexpr size_t kLdsLimit = 160 * 1024;

int launch_simbits_screen(fc_ensemble *e, double thr2_margin) {
  const int64_t NT = e->Npad >> 6;
  const int64_t n_gblocks = ceil_div(e->N, e->row_block);
  const int64_t n_lblocks = (n_gblocks - e->rank + e->world - 1) / e->world;
  if (n_lblocks <= 0 || NT == 0) return FC_OK;
  const double A_thr2 = (double)e->A * thr2_margin;
  const size_t lds = (size_t)e->A * 3 * 64 * sizeof(double);
  dim3 grid((unsigned)NT, (unsigned)n_lblocks);
  if (lds <= kLdsLimit) {
    if (lds > 64 * 1024) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(k_simbits_screen<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (err != hipSuccess)
        return set_error(FC_E_HIP, "hipFuncSetAttribute(LDS=%zu) failed: %s", lds,
                         hipGetErrorString(err));
    }
    hipLaunchKernelGGL(k_simbits_screen<true>, grid, dim3(256), lds, ctx().stream,
                       e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2,
                       (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W);
  } else {
    hipLaunchKernelGGL(k_simbits_screen<false>, grid, dim3(256), 0, ctx().stream,
                       e->Xs.as<double>(), e->G.as<double>(), e->N, e->Npad, (int)e->A, A_thr2,
                       (int)e->row_block, e->rank, e->world, e->bits.as<uint64_t>(), e->W);
  }
  return check_launch("k_simbits_screen");
}

int launch_simbits_refine(fc_ensemble *e, double max_rmsd, double max_dev, const double *energies_dev,
                          double max_dE) {
  const int64_t waves = e->rows_local * e->W;
  if (waves == 0) return FC_OK;
  hipLaunchKernelGGL(k_simbits_refine, dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0,
                     ctx().stream, e->Xs.as<double>(), e->N, e->Npad, (int)e->A, max_rmsd, max_dev,
                     energies_dev, max_dE, (int)e->row_block, e->rank, e->world, e->rows_local,
                     e->bits.as<uint64_t>(), e->W,
                     reinterpret_cast<unsigned long long *>(e->counters.p));
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

}  // namespace fc
