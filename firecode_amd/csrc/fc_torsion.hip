// fc_torsion.hip -- torsion-scan conformer generation and torsion
// fingerprints for gfx950 (wave64, float64).
//
// Replaces the inner loops of clustered_csearch / random_csearch
// (firecode/torsion_module.py:812-856, 514-556): per angle-set apply the
// non-zero dihedral rotations in order -- rotate_dihedral (prism_pruner.utils:
// axis = x[i2]-x[i3], centre x[i3], matrix rot_mat_from_pointer(axis, angle))
// -- test torsion_comp_check (torsion_module.py:894-918: any pair between the
// rotating side `mask` and the rest minus {i2,i3} closer than thresh) and
// back off in -5 degree steps, at most angle//5 times.
//
// One wavefront per angle-set: the conformer lives in the wave's LDS slice
// (A*3 doubles); a rotation is lane-per-atom; a clash test flattens the
// (rest x moving) rectangle over the 64 lanes and reduces with a ballot.
#include "fc_common.h"
#include "fc_kabsch_math.h"

namespace fc {

double sq_threshold_lt(double t);  // fc_clash.hip

// rot_mat_from_pointer: scalar-last quaternion [sin(a/2) n, cos(a/2)] -> matrix
__device__ __forceinline__ void rot_from_axis_angle(double ax, double ay, double az,
                                                    double angle_deg, double (&M)[9]) {
  double a2 = angle_deg / 2.0;
  a2 *= 3.141592653589793 / 180.0;
  double sn, cs;
  sincos(a2, &sn, &cs);
  const double nrm = sqrt((ax * ax + ay * ay) + az * az);
  const double q1 = sn * (ax / nrm), q2 = sn * (ay / nrm), q3 = sn * (az / nrm), q0 = cs;
  M[0] = 2.0 * (q0 * q0 + q1 * q1) - 1.0;
  M[1] = 2.0 * (q1 * q2 - q0 * q3);
  M[2] = 2.0 * (q1 * q3 + q0 * q2);
  M[3] = 2.0 * (q1 * q2 + q0 * q3);
  M[4] = 2.0 * (q0 * q0 + q2 * q2) - 1.0;
  M[5] = 2.0 * (q2 * q3 - q0 * q1);
  M[6] = 2.0 * (q1 * q3 - q0 * q2);
  M[7] = 2.0 * (q2 * q3 + q0 * q1);
  M[8] = 2.0 * (q0 * q0 + q3 * q3) - 1.0;
}

// rotate the atoms flagged in `mv` (bit per atom, up to 4 words = 256 atoms)
__device__ __forceinline__ void rotate_masked(double *x, int A, const uint8_t *mask, int i2, int i3,
                                              double angle, int lane) {
  const double cx = x[i3 * 3], cy = x[i3 * 3 + 1], cz = x[i3 * 3 + 2];
  double M[9];
  rot_from_axis_angle(x[i2 * 3] - cx, x[i2 * 3 + 1] - cy, x[i2 * 3 + 2] - cz, angle, M);
  __builtin_amdgcn_wave_barrier();
  for (int a = lane; a < A; a += 64) {
    if (mask[a]) {
      const double px = x[a * 3] - cx, py = x[a * 3 + 1] - cy, pz = x[a * 3 + 2] - cz;
      x[a * 3] = ((M[0] * px + M[1] * py) + M[2] * pz) + cx;
      x[a * 3 + 1] = ((M[3] * px + M[4] * py) + M[5] * pz) + cy;
      x[a * 3 + 2] = ((M[6] * px + M[7] * py) + M[8] * pz) + cz;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// true when no (rest, moving) pair is closer than thresh (max_clashes = 0 is
// the only value the reference ever passes: torsion_module.py:827,838)
__device__ __forceinline__ bool comp_check(const double *x, const int16_t *mv, int nmv,
                                           const int16_t *rs, int nrs, double thr2, int lane) {
  bool hit = false;
  const int total = nmv * nrs;
  for (int p0 = 0; p0 < total; p0 += 64) {
    const int p = p0 + lane;
    if (p < total) {
      const int r = rs[p / nmv], m = mv[p % nmv];
      const double dx = x[r * 3] - x[m * 3], dy = x[r * 3 + 1] - x[m * 3 + 1],
                   dz = x[r * 3 + 2] - x[m * 3 + 2];
      const double d2 = ((dx * dx) + dy * dy) + dz * dz;
      hit = hit || (d2 < thr2);
    }
    if (__any(hit)) return false;
  }
  return true;
}


// dihedral (prism_pruner.algebra.dihedral, "praxeolitic" form) per lane
__device__ __forceinline__ double dihedral_deg(const double *p0, const double *p1, const double *p2,
                                               const double *p3) {
  const double b0x = -1.0 * (p1[0] - p0[0]), b0y = -1.0 * (p1[1] - p0[1]), b0z = -1.0 * (p1[2] - p0[2]);
  double b1x = p2[0] - p1[0], b1y = p2[1] - p1[1], b1z = p2[2] - p1[2];
  const double b2x = p3[0] - p2[0], b2y = p3[1] - p2[1], b2z = p3[2] - p2[2];
  const double n1 = sqrt((b1x * b1x + b1y * b1y) + b1z * b1z);
  b1x /= n1; b1y /= n1; b1z /= n1;
  const double d0 = (b0x * b1x + b0y * b1y) + b0z * b1z;
  const double d2 = (b2x * b1x + b2y * b1y) + b2z * b1z;
  const double vx = b0x - d0 * b1x, vy = b0y - d0 * b1y, vz = b0z - d0 * b1z;
  const double wx = b2x - d2 * b1x, wy = b2y - d2 * b1y, wz = b2z - d2 * b1z;
  const double xx = (vx * wx + vy * wy) + vz * wz;
  const double cx = b1y * vz - b1z * vy, cy = b1z * vx - b1x * vz, cz = b1x * vy - b1y * vx;
  const double yy = (cx * wx + cy * wy) + cz * wz;
  return atan2(yy, xx) * (180.0 / 3.141592653589793);
}

__global__ void __launch_bounds__(256)
k_torsion_scan(const double *__restrict__ base, int A, const int64_t *__restrict__ torsions, int T,
               const uint8_t *__restrict__ rotmasks, const int16_t *__restrict__ mv_idx,
               const int16_t *__restrict__ rs_idx, const int32_t *__restrict__ n_mv,
               const int32_t *__restrict__ n_rs, const int64_t *__restrict__ angles, int64_t S,
               double thr2, int backoff, double *__restrict__ out, int64_t *__restrict__ rotated,
               const int64_t *__restrict__ quads, int Q, double *__restrict__ tf) {
  extern __shared__ double s[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  double *x = s + (size_t)wv * A * 3;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t sidx = wave0; sidx < S; sidx += nwaves) {
    for (int k = lane; k < A * 3; k += 64) x[k] = base[k];
    __builtin_amdgcn_wave_barrier();
    int rot = 0;
    for (int t = 0; t < T; ++t) {
      const int angle = (int)angles[sidx * T + t];
      if (angle == 0) continue;
      const int i2 = (int)torsions[t * 4 + 1], i3 = (int)torsions[t * 4 + 2];
      const uint8_t *mask = rotmasks + (size_t)t * A;
      const int16_t *mv = mv_idx + (size_t)t * A;
      const int16_t *rs = rs_idx + (size_t)t * A;
      const int nm = n_mv[t], nr = n_rs[t];
      rotate_masked(x, A, mask, i2, i3, (double)angle, lane);
      if (!comp_check(x, mv, nm, rs, nr, thr2, lane)) {
        // Python floor division: range(angle // backoff)
        int steps = angle / backoff;
        if ((angle % backoff != 0) && ((angle < 0) != (backoff < 0))) --steps;
        for (int b = 0; b < steps; ++b) {
          rotate_masked(x, A, mask, i2, i3, (double)(-backoff), lane);
          if (comp_check(x, mv, nm, rs, nr, thr2, lane)) {
            ++rot;
            break;
          }
        }
      } else {
        ++rot;
      }
    }
    if (out != nullptr) {
      double *o = out + sidx * (int64_t)A * 3;
      for (int k = lane; k < A * 3; k += 64) o[k] = x[k];
    }
    // torsion fingerprint of the conformer while it is still in LDS (get_torsion_fingerprint,
    // torsion_module.py:1070-1077): the same doubles a separate pass over `out` would read
    if (tf != nullptr)
      for (int q = lane; q < Q; q += 64)
        tf[sidx * (int64_t)Q + q] = dihedral_deg(x + quads[q * 4] * 3, x + quads[q * 4 + 1] * 3,
                                                 x + quads[q * 4 + 2] * 3, x + quads[q * 4 + 3] * 3);
    if (lane == 0) rotated[sidx] = rot;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// k_rotcorr_simbits: similarity bits of prune_by_rmsd_rot_corr (a7,
// prism_pruner.pruner; call sites firecode/ensemble.py:253-260,
// embedder.py:1489-1496).  The third-party source is not in the tree; restated
// after the predecessor's published routine (TSCoDe,
// rotationally_corrected_rmsd_and_max) -- PARITY UNPINNED:
//   for every locally symmetric torsion t (in order), for every angle of its
//   n-fold set (0 first): rotate ONLY atom i4 of the second structure about the
//   i2-i3 bond and take the Kabsch RMSD of the four torsion atoms against the
//   first structure; the first angle with the smallest local RMSD is then
//   applied to the whole rotating side (rotation mask); finally
//   rmsd_and_max over the heavy atoms decides: rmsd < max_rmsd && maxdev < max_dev.
// One wavefront per pair (i, j > i): the second structure lives in the wave's
// LDS slice, lanes = trial angles for the local fit, lanes = atoms for the
// rotations and the final superposition.  X is centred (N, A, 3).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rotcorr_simbits(const double *__restrict__ X, int64_t N, int A, const uint8_t *__restrict__ heavy,
                  const int64_t *__restrict__ torsions, int T, const uint8_t *__restrict__ rotmasks,
                  const double *__restrict__ angles, const int32_t *__restrict__ n_angles, int max_angles,
                  double max_rmsd, double max_dev, const double *__restrict__ energies, double max_dE,
                  unsigned long long *__restrict__ bits, int64_t W) {
  extern __shared__ double s[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 4 + wv;
  if (j <= i || j >= N) return;
  if (energies != nullptr && !(fabs(energies[i] - energies[j]) < max_dE)) return;
  double *x = s + (size_t)wv * A * 3;
  const double *ref = X + i * (int64_t)A * 3;
  const double *src = X + j * (int64_t)A * 3;
  for (int k = lane; k < A * 3; k += 64) x[k] = src[k];
  __builtin_amdgcn_wave_barrier();
  for (int t = 0; t < T; ++t) {
    const int i1 = (int)torsions[t * 4], i2 = (int)torsions[t * 4 + 1], i3 = (int)torsions[t * 4 + 2],
              i4 = (int)torsions[t * 4 + 3];
    const int na = n_angles[t];
    double local = 1.0e300;
    double my_angle = 0.0;
    if (lane < na) {
      my_angle = angles[(size_t)t * max_angles + lane];
      const int id[4] = {i1, i2, i3, i4};
      double p[4][3], q[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          p[k][c] = ref[id[k] * 3 + c];
          q[k][c] = x[id[k] * 3 + c];
        }
      double M[9];
      rot_from_axis_angle(q[1][0] - q[2][0], q[1][1] - q[2][1], q[1][2] - q[2][2], my_angle, M);
      const double rx = q[3][0] - q[2][0], ry = q[3][1] - q[2][1], rz = q[3][2] - q[2][2];
      q[3][0] = ((M[0] * rx + M[1] * ry) + M[2] * rz) + q[2][0];
      q[3][1] = ((M[3] * rx + M[4] * ry) + M[5] * rz) + q[2][1];
      q[3][2] = ((M[6] * rx + M[7] * ry) + M[8] * rz) + q[2][2];
      double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) B[a * 3 + b] = fma(p[k][a], q[k][b], B[a * 3 + b]);
      double R[9];
      (void)kabsch_rotation(B, R);
      double ssq = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double dx = p[k][0] - (R[0] * q[k][0] + R[1] * q[k][1] + R[2] * q[k][2]);
        const double dy = p[k][1] - (R[3] * q[k][0] + R[4] * q[k][1] + R[5] * q[k][2]);
        const double dz = p[k][2] - (R[6] * q[k][0] + R[7] * q[k][1] + R[8] * q[k][2]);
        ssq += dx * dx + dy * dy + dz * dz;
      }
      local = sqrt(ssq / 4.0);
    }
    // first angle with the smallest local RMSD (the reference compares with a strict <)
    int best_lane = lane;
    double best = local;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int ol = __shfl_xor(best_lane, off);
      if (ob < best || (ob == best && ol < best_lane)) {
        best = ob;
        best_lane = ol;
      }
    }
    const double corr = __shfl(my_angle, best_lane);
    if (corr != 0.0) rotate_masked(x, A, rotmasks + (size_t)t * A, i2, i3, corr, lane);
  }
  // rmsd_and_max(ref[heavy], x[heavy]) -- Kabsch without centring
  double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int nh = 0;
  for (int a = lane; a < A; a += 64) {
    if (heavy[a]) {
      ++nh;
      const double px = ref[a * 3], py = ref[a * 3 + 1], pz = ref[a * 3 + 2];
      const double qx = x[a * 3], qy = x[a * 3 + 1], qz = x[a * 3 + 2];
      B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
      B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
      B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 9; ++e) B[e] += __shfl_xor(B[e], off);
    nh += __shfl_xor(nh, off);
  }
  double R[9];
  (void)kabsch_rotation(B, R);
  double ssq = 0.0, mx = 0.0;
  for (int a = lane; a < A; a += 64) {
    if (heavy[a]) {
      const double qx = x[a * 3], qy = x[a * 3 + 1], qz = x[a * 3 + 2];
      const double dx = ref[a * 3] - (R[0] * qx + R[1] * qy + R[2] * qz);
      const double dy = ref[a * 3 + 1] - (R[3] * qx + R[4] * qy + R[5] * qz);
      const double dz = ref[a * 3 + 2] - (R[6] * qx + R[7] * qy + R[8] * qz);
      const double d = dx * dx + dy * dy + dz * dz;
      ssq += d;
      mx = fmax(mx, d);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ssq += __shfl_xor(ssq, off);
    mx = fmax(mx, __shfl_xor(mx, off));
  }
  const double rmsd = sqrt(ssq / (double)nh), maxdev = sqrt(mx);
  if (lane == 0 && rmsd < max_rmsd && maxdev < max_dev) atomicOr(&bits[i * W + (j >> 6)], 1ull << (j & 63));
}

__global__ void __launch_bounds__(256)
k_torsion_fingerprint(const double *__restrict__ coords, int64_t N, int64_t A,
                      const int64_t *__restrict__ quads, int Q, double *__restrict__ tf) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= N * Q) return;
  const int64_t n = g / Q;
  const int q = (int)(g % Q);
  const double *x = coords + n * A * 3;
  tf[g] = dihedral_deg(x + quads[q * 4] * 3, x + quads[q * 4 + 1] * 3, x + quads[q * 4 + 2] * 3,
                       x + quads[q * 4 + 3] * 3);
}

// ---------------------------------------------------------------------------
int launch_torsion_scan(const double *base_dev, int64_t A, const int64_t *torsions_dev, int64_t T,
                        const uint8_t *rotmasks_dev, const int16_t *mv_dev, const int16_t *rs_dev,
                        const int32_t *nmv_dev, const int32_t *nrs_dev, const int64_t *angles_dev,
                        int64_t S, double thresh, int64_t backoff, double *out_dev,
                        int64_t *rot_dev, const int64_t *quads_dev, int64_t Q, double *tf_dev) {
  if (S == 0) return FC_OK;
  const double thr2 = sq_threshold_lt(thresh);
  int64_t blocks = ceil_div(S, 4);
  const int64_t cap = (int64_t)ctx().n_cu * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_torsion_scan, dim3((unsigned)blocks), dim3(256),
                     (size_t)4 * A * 3 * sizeof(double), ctx().stream, base_dev, (int)A,
                     torsions_dev, (int)T, rotmasks_dev, mv_dev, rs_dev, nmv_dev, nrs_dev,
                     angles_dev, S, thr2, (int)backoff, out_dev, rot_dev, quads_dev, (int)Q, tf_dev);
  return check_launch("k_torsion_scan");
}

int launch_rotcorr_simbits(const double *X_dev, int64_t N, int64_t A, const uint8_t *heavy_dev,
                           const int64_t *tors_dev, int64_t T, const uint8_t *rotmasks_dev,
                           const double *angles_dev, const int32_t *n_angles_dev, int max_angles,
                           double max_rmsd, double max_dev, const double *energies_dev, double max_dE,
                           uint64_t *bits_dev, int64_t W) {
  if (N < 2) return FC_OK;
  hipLaunchKernelGGL(k_rotcorr_simbits, dim3((unsigned)ceil_div(N, 4), (unsigned)N), dim3(256),
                     (size_t)4 * A * 3 * sizeof(double), ctx().stream, X_dev, N, (int)A, heavy_dev, tors_dev,
                     (int)T, rotmasks_dev, angles_dev, n_angles_dev, max_angles, max_rmsd, max_dev, energies_dev,
                     max_dE, reinterpret_cast<unsigned long long *>(bits_dev), W);
  return check_launch("k_rotcorr_simbits");
}

int launch_torsion_fingerprint(const double *coords_dev, int64_t N, int64_t A,
                               const int64_t *quads_dev, int64_t Q, double *tf_dev) {
  if (N * Q == 0) return FC_OK;
  hipLaunchKernelGGL(k_torsion_fingerprint, dim3((unsigned)ceil_div(N * Q, 256)), dim3(256), 0,
                     ctx().stream, coords_dev, N, A, quads_dev, (int)Q, tf_dev);
  return check_launch("k_torsion_fingerprint");
}

}  // namespace fc
