// fc_embed3.hip -- trimolecular rigid ("cyclical") embed on gfx950.
//
// Replaces, for three molecules, the body of cyclical_embed (firecode/embeds.py:409-585):
// for every (conformer triple, pivot triple) "job" and each of the 8 polygonize()
// orientations, _adjust_directions (:262-407) and the systematic-angle pose loop
// (:484-569: R, t per molecule, get_embed, the trimolecular compenetration_check of
// utils.py:556-575, and the rmsd_similarity(rmsd_thr=1) filter against the poses already
// accepted in the same group).  Job enumeration, polygonize(), _get_directions() and the
// pairings filter are scalar host work (firecode_amd/embeds.py).
//
// Shape of the work
//  * _adjust_directions feeds its result to the NEXT orientation of the same job
//    (`directions` is reassigned inside the v loop, :478-482), so k_tri_adjust walks the
//    8 orientations of a job sequentially in one wavefront; its 343 candidates are
//    lane-parallel and the stable sort's "first minimum" is a wave arg-min on (cost, index).
//  * Inside a group the transform of molecule i depends only on its own step angle, so the
//    S poses (216 by default) are built from 3 x U pre-transformed structures (U distinct
//    angles per molecule, 6 by default) kept in LDS, and the clash count of a pose is the sum
//    of three entries of U x U pair tables -- 3*U*U distance rectangles instead of 3*S.
//  * The accept filter is sequential in pose order; one wavefront walks it with the poses
//    kept so far spread over its lanes; the covariance of two poses is the sum over the
//    three molecules, read from LDS -- no pose is materialised.
#include "fc_common.h"
#include "fc_kabsch_math.h"

#include <algorithm>

namespace fc {

double sq_threshold_le(double t);  // fc_clash.hip

namespace {

__device__ __forceinline__ void rot_axis_angle3(double ax, double ay, double az, double angle_deg,
                                                double (&M)[9]) {  // rot_mat_from_pointer
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

__device__ __forceinline__ void mv(const double (&M)[9], const double (&x)[3], double (&o)[3]) {
  o[0] = (M[0] * x[0] + M[1] * x[1]) + M[2] * x[2];
  o[1] = (M[3] * x[0] + M[4] * x[1]) + M[5] * x[2];
  o[2] = (M[6] * x[0] + M[7] * x[1]) + M[8] * x[2];
}

// vec_angle (prism_pruner.algebra): degrees between two vectors
__device__ __forceinline__ double vec_angle_deg(const double (&a)[3], const double (&b)[3]) {
  const double na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
  const double nb = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
  double d = ((a[0] / na) * (b[0] / nb) + (a[1] / na) * (b[1] / nb)) + (a[2] / na) * (b[2] / nb);
  d = fmin(1.0, fmax(-1.0, d));
  return acos(d) * (180.0 / 3.141592653589793);
}

struct MolView {
  const double *coords;     // (n, A, 3)
  const int64_t *reactive;  // (nr,)
  int A, nr;
};

// alignment_rotation of one molecule (embeds.py:513-518 == :331-333) and the quantities
// around it: pivot vector, pivot mean point, mean of the reactive atoms
struct Align {
  double Al[9], pv[3], mean_pt[3], react_mean[3];
};

__device__ __forceinline__ void alignment(const MolView &m, int64_t conf, const double *piv_start,
                                          const double *piv_end, const double *vec_start,
                                          const double *vec_end, const double *direction, Align &o) {
  const double *x = m.coords + conf * m.A * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.pv[k] = piv_start[k] - piv_end[k];             // Pivot.pivot = start - end
    o.mean_pt[k] = (piv_start[k] + piv_end[k]) / 2.0;  // Pivot.meanpoint
    o.react_mean[k] = 0.0;
  }
  for (int r = 0; r < m.nr; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) o.react_mean[k] += x[m.reactive[r] * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) o.react_mean[k] /= (double)m.nr;
  double md[3] = {o.mean_pt[0] - o.react_mean[0], o.mean_pt[1] - o.react_mean[1],
                  o.mean_pt[2] - o.react_mean[2]};
  if (md[0] == 0.0 && md[1] == 0.0 && md[2] == 0.0) {
    md[0] = o.mean_pt[0];
    md[1] = o.mean_pt[1];
    md[2] = o.mean_pt[2];
  }
  // align_vec_pair(ref = [end - start, direction], tgt = [pivot, mol_direction])
  double B[9];
#pragma unroll
  for (int xx = 0; xx < 3; ++xx)
#pragma unroll
    for (int yy = 0; yy < 3; ++yy)
      B[xx * 3 + yy] = (vec_end[xx] - vec_start[xx]) * o.pv[yy] + direction[xx] * md[yy];
  (void)kabsch_rotation(B, o.Al);
}

}  // namespace

// ---------------------------------------------------------------------------
// k_tri_adjust: one wavefront per job, orientations in sequence.
//   conf (J,3) | piv_start, piv_end (J,3,3) | vecs (J,8,3,2,3) | dirs0 (J,3,3) | run (J,8)
//   rtab (J,8,3,3): reactive atom index of molecule m facing molecule k | norms (J,3)
//   dirs_out (J,8,3,3): directions used by the pose loop of orientation v
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tri_adjust(MolView m0, MolView m1, MolView m2, int64_t J, const int64_t *__restrict__ conf,
             const double *__restrict__ piv_start, const double *__restrict__ piv_end,
             const double *__restrict__ vecs, const double *__restrict__ dirs0,
             const uint8_t *__restrict__ run, const int64_t *__restrict__ rtab,
             const double *__restrict__ norms, double *__restrict__ dirs_out) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= J) return;
  const MolView mols[3] = {m0, m1, m2};
  double dir[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) dir[i][k] = dirs0[(j * 3 + i) * 3 + k];
  // triangle vertices in the plane (embeds.py:283-309)
  const double n0 = norms[j * 3], n1 = norms[j * 3 + 1], n2 = norms[j * 3 + 2];
  const double sa = n0 * n0, sb = n1 * n1, sc = n2 * n2;
  const double vx = (sa - sb + sc) / (2.0 * sqrt(sa));
  const double vy = sqrt(sc - vx * vx);
  const double V[3][3] = {{0.0, 0.0, 0.0}, {n0, 0.0, 0.0}, {vx, vy, 0.0}};

  for (int v = 0; v < 8; ++v) {
    if (run[j * 8 + v]) {  // wave-uniform
      // a[m][k]: reactive atom of molecule m facing molecule k, placed by the first estimate
      double P[3][3], Pm[3][3], a[3][3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double *vs = vecs + (((j * 8 + v) * 3 + i) * 2 + 0) * 3;
        const double *ve = vs + 3;
        Align al;
        alignment(mols[i], conf[j * 3 + i], piv_start + (j * 3 + i) * 3, piv_end + (j * 3 + i) * 3, vs,
                  ve, dir[i], al);
        double rm[3], pos[3];
        mv(al.Al, al.mean_pt, rm);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          P[i][k] = ve[k] - vs[k];
          Pm[i][k] = (ve[k] + vs[k]) / 2.0;
          pos[k] = Pm[i][k] - rm[k];
        }
        const double *x0 = mols[i].coords;  // conformer 0 (embeds.py:366-373)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k == i) continue;
          const int64_t r = rtab[((j * 8 + v) * 3 + i) * 3 + k];
          const double xr[3] = {x0[r * 3], x0[r * 3 + 1], x0[r * 3 + 2]};
          double t[3];
          mv(al.Al, xr, t);
          a[i][k][0] = t[0] + pos[0];
          a[i][k][1] = t[1] + pos[1];
          a[i][k][2] = t[2] + pos[2];
        }
      }
      // 7^3 candidates in cartesian_product order: index c -> (i1, i0, i2) with the second
      // argument slowest (firecode/utils.py:219-221), angle = i*10 - 30
      double best = 1.0e300;
      int best_c = 1 << 30;
      for (int c = lane; c < 343; c += 64) {
        const int i2 = c % 7, i0 = (c / 7) % 7, i1 = c / 49;
        const int ia[3] = {i0, i1, i2};
        double na_[3][3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double Rm[9];
          rot_axis_angle3(P[i][0], P[i][1], P[i][2], (double)ia[i] * 10.0 - 30.0, Rm);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k != i) mv(Rm, a[i][k], na_[i][k]);
        }
        double u[3], w[3], cost = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = V[0][k] - na_[0][2][k]; w[k] = na_[2][0][k] - V[0][k]; }
        cost += vec_angle_deg(u, w);
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = V[1][k] - na_[0][1][k]; w[k] = na_[1][0][k] - V[1][k]; }
        cost += vec_angle_deg(u, w);
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = V[2][k] - na_[2][1][k]; w[k] = na_[1][2][k] - V[2][k]; }
        cost += vec_angle_deg(u, w);
        if (cost < best) {  // candidates of a lane come in increasing index
          best = cost;
          best_c = c;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oc = __shfl_xor(best_c, off);
        if (ob < best || (ob == best && oc < best_c)) {
          best = ob;
          best_c = oc;
        }
      }
      {  // directions of the winner: d_i = mean(side i) - mean of its two rotated reactive atoms
        const int i2 = best_c % 7, i0 = (best_c / 7) % 7, i1 = best_c / 49;
        const int ia[3] = {i0, i1, i2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double Rm[9];
          rot_axis_angle3(P[i][0], P[i][1], P[i][2], (double)ia[i] * 10.0 - 30.0, Rm);
          const int ka = (i == 0) ? 1 : 0, kb = (i == 2) ? 1 : 2;  // (01,02) (10,12) (20,21)
          double ra[3], rb[3];
          mv(Rm, a[i][ka], ra);
          mv(Rm, a[i][kb], rb);
#pragma unroll
          for (int k = 0; k < 3; ++k) dir[i][k] = Pm[i][k] - (ra[k] + rb[k]) / 2.0;
        }
      }
    }
    if (lane < 9) dirs_out[(j * 8 + v) * 9 + lane] = dir[lane / 3][lane % 3];
  }
}

// ---------------------------------------------------------------------------
// k_tri_group: one workgroup per (job, orientation).
//   ua (3, U): distinct step angles per molecule | aidx (S, 3): pose -> index into ua
//   Rt_out (J,8,3,U,12): R (9) | t (3) of every pre-transformed structure
//   pass / accept (J,8,S)
// LDS: X[3][U][A_i][3] doubles | Rt[3][U][12] | tables 3*U*U int | pass S bytes | kept S int
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tri_group(MolView m0, MolView m1, MolView m2, int64_t J, const int64_t *__restrict__ conf,
            const double *__restrict__ piv_start, const double *__restrict__ piv_end,
            const double *__restrict__ vecs, const double *__restrict__ dirs,
            const uint8_t *__restrict__ run, const double *__restrict__ ua, int U,
            const int32_t *__restrict__ aidx, int S, double thr2_le, int max_clashes, double rmsd_thr,
            double *__restrict__ Rt_out, uint8_t *__restrict__ pass_out, uint8_t *__restrict__ accept_out) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t g = blockIdx.x;  // job * 8 + v
  if (!run[g]) {
    for (int s = tid; s < S; s += 256) {
      pass_out[g * S + s] = 0;
      accept_out[g * S + s] = 0;
    }
    return;
  }
  const int64_t j = g >> 3;
  const MolView mols[3] = {m0, m1, m2};
  const int A[3] = {m0.A, m1.A, m2.A};
  const int xoff[3] = {0, U * A[0] * 3, U * (A[0] + A[1]) * 3};
  const int Atot = A[0] + A[1] + A[2];
  double *X = lds;
  double *Rt = X + (size_t)U * Atot * 3;
  int *tab = reinterpret_cast<int *>(Rt + 3 * U * 12);
  int *kept = tab + 3 * U * U;
  uint8_t *pass = reinterpret_cast<uint8_t *>(kept + S);

  if (tid < 3 * U) {  // R, t of (molecule i, angle u): embeds.py:488-546
    const int i = tid / U, u = tid - i * U;
    const double *vs = vecs + ((g * 3 + i) * 2 + 0) * 3;
    const double *ve = vs + 3;
    Align al;
    alignment(mols[i], conf[j * 3 + i], piv_start + (j * 3 + i) * 3, piv_end + (j * 3 + i) * 3, vs, ve,
              dirs + (g * 3 + i) * 3, al);
    const double *x = mols[i].coords + conf[j * 3 + i] * A[i] * 3;
    double axis_in[3], axis[3];
    if (mols[i].nr == 2) {
      const int64_t r0 = mols[i].reactive[0], r1 = mols[i].reactive[1];
#pragma unroll
      for (int k = 0; k < 3; ++k) axis_in[k] = x[r0 * 3 + k] - x[r1 * 3 + k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) axis_in[k] = al.pv[k];
    }
    mv(al.Al, axis_in, axis);
    double St[9];
    rot_axis_angle3(axis[0], axis[1], axis[2], ua[i * U + u], St);
    double cen[3], scen[3], rmean[3];
    mv(al.Al, al.react_mean, cen);  // center_of_rotation
    mv(St, cen, scen);
    mv(al.Al, al.mean_pt, rmean);
    double *o = Rt + (i * U + u) * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        o[r * 3 + q] = (St[r * 3] * al.Al[q] + St[r * 3 + 1] * al.Al[3 + q]) + St[r * 3 + 2] * al.Al[6 + q];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double pos = (vs[k] + ve[k]) / 2.0 - rmean[k];
      o[9 + k] = (cen[k] - scen[k]) + pos;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) Rt_out[((g * 3 + i) * U + u) * 12 + k] = o[k];
  }
  __syncthreads();
  // pre-transformed structures: X[i][u][a] = R x + t  (get_embed, embeds.py:808-817)
  for (int idx = tid; idx < U * Atot; idx += 256) {
    int i = 0, rem = idx;
    if (rem >= U * A[0]) { rem -= U * A[0]; i = 1; }
    if (i == 1 && rem >= U * A[1]) { rem -= U * A[1]; i = 2; }
    const int u = rem / A[i], a = rem - u * A[i];
    const double *x = mols[i].coords + (conf[j * 3 + i] * A[i] + a) * 3;
    const double *r = Rt + (i * U + u) * 12;
    double *o = X + xoff[i] + (u * A[i] + a) * 3;
    o[0] = ((r[0] * x[0] + r[1] * x[1]) + r[2] * x[2]) + r[9];
    o[1] = ((r[3] * x[0] + r[4] * x[1]) + r[5] * x[2]) + r[10];
    o[2] = ((r[6] * x[0] + r[7] * x[1]) + r[8] * x[2]) + r[11];
  }
  __syncthreads();
  // pair tables (utils.py:565-573): t = 0: cdist(m2, m1) -> [u1][u0]; 1: cdist(m3, m2) -> [u2][u1];
  // 2: cdist(m1, m3) -> [u0][u2]; entry = number of atom pairs with d <= thresh
  for (int e = wv; e < 3 * U * U; e += 4) {
    const int t = e / (U * U), ub = (e / U) % U, uc = e % U;
    const int mb = (t == 0) ? 1 : (t == 1) ? 2 : 0;  // rows of cdist
    const int mc = (t == 0) ? 0 : (t == 1) ? 1 : 2;  // columns
    const double *xb = X + xoff[mb] + ub * A[mb] * 3;
    const double *xc = X + xoff[mc] + uc * A[mc] * 3;
    int cnt = 0;
    const int np = A[mb] * A[mc];
    for (int p = lane; p < np; p += 64) {
      const int ib = p / A[mc], ic = p - ib * A[mc];
      const double dx = xb[ib * 3] - xc[ic * 3], dy = xb[ib * 3 + 1] - xc[ic * 3 + 1],
                   dz = xb[ib * 3 + 2] - xc[ic * 3 + 2];
      const double d2 = ((dx * dx) + dy * dy) + dz * dz;
      cnt += (d2 <= thr2_le) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) tab[e] = cnt;
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    const int u0 = aidx[s * 3], u1 = aidx[s * 3 + 1], u2 = aidx[s * 3 + 2];
    const int total = tab[u1 * U + u0] + tab[U * U + u2 * U + u1] + tab[2 * U * U + u0 * U + u2];
    const uint8_t ok = total <= max_clashes ? 1 : 0;
    pass[s] = ok;
    pass_out[g * S + s] = ok;
  }
  __syncthreads();
  if (wv != 0) return;
  // sequential accept filter (embeds.py:553-566), wave 0: lanes = poses kept so far
  int n_kept = 0;
  for (int s = 0; s < S; ++s) {
    if (!pass[s]) {
      if (lane == 0) accept_out[g * S + s] = 0;
      continue;
    }
    const int pu[3] = {aidx[s * 3], aidx[s * 3 + 1], aidx[s * 3 + 2]};
    bool similar = false;
    for (int k0 = 0; k0 < n_kept && !similar; k0 += 64) {
      const int kk = k0 + lane;
      bool hit = false;
      if (kk < n_kept) {
        const int q = kept[kk];
        const int qu[3] = {aidx[q * 3], aidx[q * 3 + 1], aidx[q * 3 + 2]};
        double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) {
          const double *pp = X + xoff[i] + pu[i] * A[i] * 3;
          const double *qq = X + xoff[i] + qu[i] * A[i] * 3;
          for (int a = 0; a < A[i]; ++a) {
            const double px = pp[a * 3], py = pp[a * 3 + 1], pz = pp[a * 3 + 2];
            const double qx = qq[a * 3], qy = qq[a * 3 + 1], qz = qq[a * 3 + 2];
            B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
            B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
            B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
          }
        }
        double R[9];
        (void)kabsch_rotation(B, R);
        double ssq = 0.0, mx = 0.0;
        for (int i = 0; i < 3; ++i) {
          const double *pp = X + xoff[i] + pu[i] * A[i] * 3;
          const double *qq = X + xoff[i] + qu[i] * A[i] * 3;
          for (int a = 0; a < A[i]; ++a) {
            const double qx = qq[a * 3], qy = qq[a * 3 + 1], qz = qq[a * 3 + 2];
            const double dx = pp[a * 3] - (R[0] * qx + R[1] * qy + R[2] * qz);
            const double dy = pp[a * 3 + 1] - (R[3] * qx + R[4] * qy + R[5] * qz);
            const double dz = pp[a * 3 + 2] - (R[6] * qx + R[7] * qy + R[8] * qz);
            const double d = dx * dx + dy * dy + dz * dz;
            ssq += d;
            mx = fmax(mx, d);
          }
        }
        const double rmsd = sqrt(ssq / (double)Atot), maxdev = sqrt(mx);
        hit = (rmsd < rmsd_thr) && (maxdev < 2.0 * rmsd_thr);
      }
      similar = __any(hit);
    }
    if (!similar) {
      if (lane == 0) kept[n_kept] = s;
      ++n_kept;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    if (lane == 0) accept_out[g * S + s] = similar ? 0 : 1;
  }
}

// ---------------------------------------------------------------------------
size_t tri_group_lds_bytes(int64_t Atot, int U, int S) {
  size_t b = (size_t)U * Atot * 3 * sizeof(double) + (size_t)3 * U * 12 * sizeof(double);
  b += (size_t)3 * U * U * sizeof(int) + (size_t)S * sizeof(int) + (size_t)S;
  return (b + 15) & ~(size_t)15;
}

int launch_tri_embed(const double *const coords_dev[3], const int64_t *const reactive_dev[3],
                     const int64_t A[3], const int64_t nr[3], int64_t J, const int64_t *conf_dev,
                     const double *piv_start_dev, const double *piv_end_dev, const double *vecs_dev,
                     const double *dirs0_dev, const uint8_t *run_dev, const int64_t *rtab_dev,
                     const double *norms_dev, const double *ua_dev, int U, const int32_t *aidx_dev, int S,
                     double thresh, int max_clashes, double rmsd_thr, double *dirs_dev, double *Rt_dev,
                     uint8_t *pass_dev, uint8_t *accept_dev) {
  if (J == 0) return FC_OK;
  MolView m[3];
  for (int i = 0; i < 3; ++i) m[i] = MolView{coords_dev[i], reactive_dev[i], (int)A[i], (int)nr[i]};
  hipLaunchKernelGGL(k_tri_adjust, dim3((unsigned)ceil_div(J, 4)), dim3(256), 0, ctx().stream, m[0], m[1],
                     m[2], J, conf_dev, piv_start_dev, piv_end_dev, vecs_dev, dirs0_dev, run_dev, rtab_dev,
                     norms_dev, dirs_dev);
  FC_TRY(check_launch("k_tri_adjust"));
  const size_t lds = tri_group_lds_bytes(A[0] + A[1] + A[2], U, S);
  if (lds > 64 * 1024) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tri_group),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess)
      return set_error(FC_E_HIP, "hipFuncSetAttribute(LDS=%zu) failed: %s", lds, hipGetErrorString(err));
  }
  hipLaunchKernelGGL(k_tri_group, dim3((unsigned)(J * 8)), dim3(256), lds, ctx().stream, m[0], m[1], m[2], J,
                     conf_dev, piv_start_dev, piv_end_dev, vecs_dev, dirs_dev, run_dev, ua_dev, U, aidx_dev, S,
                     sq_threshold_le(thresh), max_clashes, rmsd_thr, Rt_dev, pass_dev, accept_dev);
  return check_launch("k_tri_group");
}

// fc_warmup(): the first launch from a translation unit makes the runtime load that unit's code object (milliseconds);
// a no-op launch moves that cost out of the first real call
__global__ void k_warm_embed3() {}
int warm_embed3() {
  hipLaunchKernelGGL(k_warm_embed3, dim3(1), dim3(64), 0, ctx().stream);
  return check_launch("k_warm_embed3");
}

}  // namespace fc
