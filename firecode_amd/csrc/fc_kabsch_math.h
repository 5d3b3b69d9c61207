// fc_kabsch_math.h -- per-lane float64 3x3 / 4x4 algebra for the Kabsch
// superposition (register resident, no LDS, no MFMA: SURVEY.md section 7).
//
// Convention (pinned to the in-tree twin firecode/algebra.py:28-49 and the
// call site hypermolecule_class.py:77-84): B[x][y] = sum_a p_a[x] * q_a[y];
// the rotation R returned maximises tr(R^T B) over SO(3), i.e. R rotates q
// onto p and is applied as  q' = R q  ( == (M @ q.T).T row-wise ).
// LAPACK's  u, s, vh = svd(B);  det-fix;  R = u @ vh  gives the same matrix;
// here it is obtained as the dominant eigenvector of Horn's 4x4 quaternion
// matrix by cyclic Jacobi, which never leaves SO(3) and needs no det-fix.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

// One Jacobi rotation on the (P,Q) plane of a symmetric 4x4 `a`, accumulated
// into the eigenvector matrix `v` (columns).  All indices are compile time,
// so both arrays live in VGPRs.
template <int P, int Q>
__host__ __device__ __forceinline__ void jacobi_rot(double (&a)[4][4], double (&v)[4][4], bool late) {
  const double apq = a[P][Q];
  const double g = 100.0 * fabs(apq);
  const double app = a[P][P], aqq = a[Q][Q];
  if (late && (fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq))) {
    a[P][Q] = 0.0;
    a[Q][P] = 0.0;
    return;
  }
  if (apq == 0.0) return;
  const double h = aqq - app;
  double t;
  if (fabs(h) + g == fabs(h)) {
    t = apq / h;
  } else {
    const double theta = 0.5 * h / apq;
    t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
    if (theta < 0.0) t = -t;
  }
  const double c = 1.0 / sqrt(1.0 + t * t);
  const double s = t * c;
  const double tau = s / (1.0 + c);
  const double hh = t * apq;
  a[P][P] = app - hh;
  a[Q][Q] = aqq + hh;
  a[P][Q] = 0.0;
  a[Q][P] = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r != P && r != Q) {
      const double arp = a[r][P], arq = a[r][Q];
      const double np_ = arp - s * (arq + arp * tau);
      const double nq_ = arq + s * (arp - arq * tau);
      a[r][P] = np_;
      a[P][r] = np_;
      a[r][Q] = nq_;
      a[Q][r] = nq_;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double vrp = v[r][P], vrq = v[r][Q];
    v[r][P] = vrp - s * (vrq + vrp * tau);
    v[r][Q] = vrq + s * (vrp - vrq * tau);
  }
}

// Optimal rotation (row-major R[9]) from the covariance B (row-major 3x3).
// Returns the largest eigenvalue of the quaternion matrix
// (= s1 + s2 +/- s3 of B).
__host__ __device__ __forceinline__ double kabsch_rotation(const double (&B)[9], double (&R)[9]) {
  // Horn's matrix for rotating the q-structure onto the p-structure:
  // S_xy = sum q_x p_y = B[y][x]
  const double Sxx = B[0], Sxy = B[3], Sxz = B[6];
  const double Syx = B[1], Syy = B[4], Syz = B[7];
  const double Szx = B[2], Szy = B[5], Szz = B[8];
  double a[4][4], v[4][4];
  a[0][0] = Sxx + Syy + Szz;
  a[0][1] = Syz - Szy;
  a[0][2] = Szx - Sxz;
  a[0][3] = Sxy - Syx;
  a[1][1] = Sxx - Syy - Szz;
  a[1][2] = Sxy + Syx;
  a[1][3] = Szx + Sxz;
  a[2][2] = -Sxx + Syy - Szz;
  a[2][3] = Syz + Szy;
  a[3][3] = -Sxx - Syy + Szz;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) a[i][j] = a[j][i];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;

  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[0][3]) + fabs(a[1][2]) +
                       fabs(a[1][3]) + fabs(a[2][3]);
    if (off == 0.0) break;
    const bool late = sweep > 3;
    jacobi_rot<0, 1>(a, v, late);
    jacobi_rot<0, 2>(a, v, late);
    jacobi_rot<0, 3>(a, v, late);
    jacobi_rot<1, 2>(a, v, late);
    jacobi_rot<1, 3>(a, v, late);
    jacobi_rot<2, 3>(a, v, late);
  }
  // dominant eigenpair (select with compile-time indices only)
  double lam = a[0][0];
  double q0 = v[0][0], q1 = v[1][0], q2 = v[2][0], q3 = v[3][0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const bool better = a[k][k] > lam;
    lam = better ? a[k][k] : lam;
    q0 = better ? v[0][k] : q0;
    q1 = better ? v[1][k] : q1;
    q2 = better ? v[2][k] : q2;
    q3 = better ? v[3][k] : q3;
  }
  const double nrm = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  q0 *= nrm;
  q1 *= nrm;
  q2 *= nrm;
  q3 *= nrm;
  R[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
  R[1] = 2.0 * (q1 * q2 - q0 * q3);
  R[2] = 2.0 * (q1 * q3 + q0 * q2);
  R[3] = 2.0 * (q1 * q2 + q0 * q3);
  R[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
  R[5] = 2.0 * (q2 * q3 - q0 * q1);
  R[6] = 2.0 * (q1 * q3 - q0 * q2);
  R[7] = 2.0 * (q2 * q3 + q0 * q1);
  R[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
  return lam;
}

// Division-free screen used by the all-pairs kernel.  With
// L = (Gp + Gq - A*thr2)/2 the pair has  msd < thr2  iff the largest root of
// the quaternion characteristic polynomial
//     P(x) = x^4 + C2 x^2 + C1 x + C0,
//     C2 = -2 |B|_F^2,  C1 = -8 det B,  C0 = det K
// exceeds L.  All roots are real, so by Budan-Fourier no root exceeds L iff
// P, P', P'', P''' are all >= 0 at L.  Returns true when the pair MAY be
// similar (it is then re-evaluated exactly); false only when it provably
// (with a 1e-12 relative guard on P) is not.
__host__ __device__ __forceinline__ bool kabsch_may_be_below(const double (&B)[9], double GpGq,
                                                     double A_thr2) {
  // the screen has a 1e-6 A^2 margin and a 1e-12 relative guard: fused
  // multiply-adds are welcome here (the file is otherwise built contract=off)
#pragma clang fp contract(fast)
  const double s = 0.5 * GpGq;
  const double L = s - 0.5 * A_thr2;
  const bool tiny = !(A_thr2 < 0.5 * s);  // tiny structure w.r.t. threshold: cannot be screened
  const double Sxx = B[0], Sxy = B[1], Sxz = B[2];
  const double Syx = B[3], Syy = B[4], Syz = B[5];
  const double Szx = B[6], Szy = B[7], Szz = B[8];
  const double n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                    Szx * Szx + Szy * Szy + Szz * Szz;
  // with u = L^2 - n2 the three values to test are
  //   P''/4 = 2 L^2 + u,   P'/4 = u L - 2 det B,   P = u^2 - 4 (|cof B|_F^2 + 2 L det B)
  // (C2 = -2 n2, C1 = -8 det B, C0 = n2^2 - 4 |cof B|_F^2: with s_i the singular values of B
  // the eigenvalues of K are (+-s1 +- s2 +- s3) with an even number of minus signs, s3 signed
  // by det B, whose product is (s1^2+s2^2+s3^2)^2 - 4 (s1^2 s2^2 + s2^2 s3^2 + s3^2 s1^2)).
  // Evaluated without branches: nearly every lane needs all three values anyway, and the
  // early exits cost the wavefront an exec-mask dance per condition.
  const double L2 = L * L;
  const double u = L2 - n2;
  const double c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const double c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const double c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const double detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const double e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                    c20 * c20 + c21 * c21 + c22 * c22;
  const double P2 = 2.0 * L2 + u;
  const double P1 = u * L - 2.0 * detB;
  const double P0 = u * u - 4.0 * (e2 + 2.0 * L * detB);
  const double eps = 1e-12 * (s * s) * (s * s);
  return tiny | (P2 < 0.0) | (P1 < 0.0) | !(P0 > eps);
}


// The same screen on an fp32 covariance (k_simbits_screen_mfma_f32): a pair may be dropped
// only when P, P', P'' at L are PROVABLY non-negative, so each computed value has to clear a
// bound on everything single precision did to it.  In units of s = (Gp+Gq)/2 (b = B/s,
// l = L/s; the test is homogeneous of degree 2, 3, 4): |b|_F <= 1, 0.75 < l <= 1 and, with
// u = 2^-24,
//   db  <= eta + u,  eta = (A4 + 4) u : entry error of b -- inputs rounded to fp32 (2u of
//                                       sum|x||y| <= s), A4 fused accumulations, scaling
//   dl  <= 4u   (G/2 kept as floats: u on s, then L = s - A thr2 / 2 and its conversion)
//   |dn2| <= 2 |b|_F |db|_F <= 6 db,   |d det b| <= |cof b|_F |db|_F <= 1.74 db,
//   |d e2| <= 2 |cof b|_F |d cof|_F <= 2 * 0.58 * 4 db = 4.7 db     (|cof b|_F^2 = e2 <= 1/3)
//   |dP2| <= 6 db + 12 u,   |dP1| <= 9.5 db + 6 u,   |dP0| <= 45 db + 12 u
// plus the roundings of the ~60 fp32 operations below on values <= 3 (<= 160 u for P0, 40 u for
// P1 and P2).  The launcher passes bounds 2x as wide (kabsch_f32_bounds).  A pair inside the
// band goes to the exact fp64 refine like any other candidate.
struct KabschF32Bounds {
  float p0, p1, p2;
};
inline KabschF32Bounds kabsch_f32_bounds(int64_t A4) {
  const double u = 5.9604644775390625e-08, db = ((double)A4 + 4.0) * u + u;
  return {(float)(2.0 * (45.0 * db + 172.0 * u)), (float)(2.0 * (9.5 * db + 46.0 * u)),
          (float)(2.0 * (6.0 * db + 52.0 * u))};
}

// Nothing is divided: with s = (Gp + Gq)/2 in fp32 (the kernel keeps G/2 as floats: one more u on
// s, two more on L, inside the 2x) P'', P', P are compared with p2 s^2, p1 s^3, p0 s^4.  (A form
// scaled by 1/s with s and L in fp64 costs 14 instructions more: 0.495 against 0.474 ms for the
// kernel.  While the SLP vectoriser still packed this polynomial into v_pk_*_f32 the order was
// the other way round.)
// s_bound: the scale the ERROR bounds refer to.  The full test passes s itself.  The subset stage of
// the lean kernel passes the UNcentred half-norm sum of the subset, s_u >= s: its accumulators hold
// uncentred sums (entry error eta * s_u) from which a rank-one centring term is subtracted, so in units
// of s_u: |b|_F <= s / s_u <= 1, l <= 1 and every derivative bound above holds a fortiori.
// tiny_floor: pairs with s at or below it go to the exact path untested (the plain test passes
// 4 * half_A_thr2: a structure that small against the threshold cannot be screened; the split-half kernel
// passes the larger of that and the scale its absolute error term needs).
__device__ __forceinline__ bool kabsch_may_be_below_f32(const float (&B)[9], float s, float half_A_thr2,
                                                        const KabschF32Bounds &bd, float s_bound, float tiny_floor) {
#pragma clang fp contract(fast)
  const float L = s - half_A_thr2;
  const bool tiny = !(tiny_floor < s);
  const float Sxx = B[0], Sxy = B[1], Sxz = B[2];
  const float Syx = B[3], Syy = B[4], Syz = B[5];
  const float Szx = B[6], Szy = B[7], Szz = B[8];
  const float n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                   Szx * Szx + Szy * Szy + Szz * Szz;
  const float L2 = L * L;
  const float uu = L2 - n2;
  const float c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const float c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const float c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const float detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const float e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                   c20 * c20 + c21 * c21 + c22 * c22;
  const float P2 = 2.0f * L2 + uu;
  const float P1 = uu * L - 2.0f * detB;
  const float P0 = uu * uu - 4.0f * (e2 + 2.0f * L * detB);
  const float s2 = s_bound * s_bound;
  // NaN and overflow (inf or NaN coordinates, s^4 beyond fp32) fail every `>`: the pair goes
  // to the exact path; so does an underflow of s^4 to zero
  return tiny | !(P2 > bd.p2 * s2) | !(P1 > bd.p1 * (s2 * s)) | !(P0 > bd.p0 * (s2 * s2));
}
__device__ __forceinline__ bool kabsch_may_be_below_f32(const float (&B)[9], float s, float half_A_thr2,
                                                        const KabschF32Bounds &bd, float s_bound) {
  return kabsch_may_be_below_f32(B, s, half_A_thr2, bd, s_bound, 4.0f * half_A_thr2);
}

// The same decision with TWO sign tests instead of three (the split-half kernel, whose matrix work is so
// cheap that this polynomial is most of its time).  With singular values f1 >= f2 >= |f3| of B (f3 signed by
// det B) the roots of P are  f1+f2+f3, f1-f2-f3, -f1+f2-f3, -f1-f2+f3:  the second largest is <= f1 <= |B|_F.
// So  u = L^2 - |B|_F^2 > 0  (and L > 0: the tiny_floor)  puts three roots below L, P(L) then has the sign of
// L - lambda_max, and  P(L) > 0  alone proves the pair dissimilar -- P' is not needed, P'' is replaced by u,
// which it contains (|du| <= |dP''/4|: bd.p2 covers it).  u <= 0 happens to a dissimilar pair only when both
// structures are nearly rank one (atoms on a line: |B|_F within A thr^2 / 2 of (Gp+Gq)/2): those pairs -- none
// in an ensemble of three-dimensional molecules -- take the three-test form above.
// Branch-free: returns the conservative verdict (u <= 0 counts as "may be similar") and reports in `redo` the
// lanes that should take the three-test form instead (the caller does that behind ONE wave-uniform branch per
// group of pairs: a branch per pair costs the wave a VALU -> SALU round trip each).
__device__ __forceinline__ bool kabsch_may_be_below_f32_2t(const float (&B)[9], float s, float half_A_thr2,
                                                           const KabschF32Bounds &bd, float tiny_floor, bool &redo) {
#pragma clang fp contract(fast)
  const float L = s - half_A_thr2;
  const bool tiny = !(tiny_floor < s);
  const float Sxx = B[0], Sxy = B[1], Sxz = B[2];
  const float Syx = B[3], Syy = B[4], Syz = B[5];
  const float Szx = B[6], Szy = B[7], Szz = B[8];
  const float n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                   Szx * Szx + Szy * Szy + Szz * Szz;
  const float uu = L * L - n2;
  const float c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const float c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const float c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const float detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const float e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                   c20 * c20 + c21 * c21 + c22 * c22;
  const float P0 = uu * uu - 4.0f * (e2 + 2.0f * L * detB);
  const float s2 = s * s;
  const bool u_ok = uu > bd.p2 * s2;
  redo = !u_ok & !tiny;
  return tiny | !u_ok | !(P0 > bd.p0 * (s2 * s2));
}

// The two-test form for a whole wavefront: every comparison is read as a 64-lane mask straight from the compare
// instruction (it writes a scalar register pair anyway) and the verdicts are combined by scalar logic -- no
// per-lane booleans to keep in vector registers across the rare branches (the compiler packed them into bytes:
// ~22 vector instructions per 16 x 16 sub-tile of a kernel bound by vector issue).  Same comparisons, same NaN
// behaviour as kabsch_may_be_below_f32_2t: a failed `>` counts as "may be similar".  All 64 lanes must be active.
__device__ __forceinline__ uint64_t kabsch_may_be_below_f32_2t_wave(const float (&B)[9], float s, float half_A_thr2,
                                                                    const KabschF32Bounds &bd, float tiny_floor,
                                                                    uint64_t &redo) {
#pragma clang fp contract(fast)
  const float L = s - half_A_thr2;
  const float Sxx = B[0], Sxy = B[1], Sxz = B[2];
  const float Syx = B[3], Syy = B[4], Syz = B[5];
  const float Szx = B[6], Szy = B[7], Szz = B[8];
  const float n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                   Szx * Szx + Szy * Szy + Szz * Szz;
  const float uu = L * L - n2;
  const float c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const float c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const float c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const float detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const float e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                   c20 * c20 + c21 * c21 + c22 * c22;
  const float P0 = uu * uu - 4.0f * (e2 + 2.0f * L * detB);
  const float s2 = s * s;
  const uint64_t not_tiny = __builtin_amdgcn_ballot_w64(tiny_floor < s);
  const uint64_t u_ok = __builtin_amdgcn_ballot_w64(uu > bd.p2 * s2);
  const uint64_t p_ok = __builtin_amdgcn_ballot_w64(P0 > bd.p0 * (s2 * s2));
  redo = ~u_ok & not_tiny;
  return ~(not_tiny & u_ok & p_ok);
}

// Bounds for the split-half kernel (k_simbits_screen_mfma_h2): the covariance is accumulated by
// v_mfma_f32_16x16x32_f16 from coordinates held as hi + lo halfs, scaled by a power of two.  In units
// of s (scaled), u = 2^-24, entry error of b = B / s:
//   representation  x = hi + lo + d, |d| <= 2^-22 (1 + 2^-12) |x|  ->  sum (|dp||q| + |p||dq|) <= 8.01 u s;
//                   where lo is subnormal |d| <= 2^-25 instead: <= 2^-25 (sum |p| + sum |q|)
//                   <= 2^-24 sqrt(A s) <= u s  for s >= A  (the kernel's tiny_floor sends smaller s to the exact path)
//   lo lo^T left out: |lo| <= 2^-11 (1 + 2^-11) |x|                     ->  <= 4.01 u s
//   matrix pipe     per instruction <= 66 u (|C| + sum |a b|), a bound that does NOT depend on how the instruction
//                   orders or aligns its sum: the 32 products of two halfs are exact in fp32 (22 significant bits);
//                   they and C are 33 addends; whatever the adder tree, each of its 32 additions loses at most
//                   2 u of its result (a truncating adder; rounding to nearest: u), every partial sum is at most
//                   |C| + sum |a b| in magnitude, and a final rounding adds u: <= (32 * 2 + 1) u, charged as 66.  The
//                   same figure covers an adder that first aligns all 33 addends to the largest and truncates each
//                   below its last place (<= 2 u each).  Measured on gfx950 (tools/ubench_mfma_f16_numerics.hip,
//                   fc_h2_check.hip at first use on every device): worst 5.4 u -- the check is a tripwire at 18 u,
//                   the bound no longer leans on it (round 2 charged 36 u from the measured behaviour).
//                   KS2 hi-hi instructions on an accumulator <= (1 + 2^-9) s, after 2 KS2 cross-term instructions
//                   on one <= 2^-10 (1 + 2^-9) s                                             ->  <= (66.2 KS2 + 0.2 KS2) u s
//   scaling         G/2 to float and its product with 2^2e: inside the `+ u` and the 4 u on L of the f32 analysis
// and from there on the analysis of kabsch_f32_bounds (same polynomial, same evaluation roundings).
constexpr double kH2InstrBound = 66.0;    // u (|C| + sum |a b|) per v_mfma_f32_16x16x32_f16: order-independent (above)
constexpr double kH2InstrTripwire = 18.0;  // fc_h2_check.hip refuses the screen on a device that exceeds this
inline double kabsch_h2_entry_bound(int64_t KS2) { return (8.01 + 1.0 + 4.01 + (kH2InstrBound + 0.4) * (double)KS2) * 5.9604644775390625e-08; }
inline KabschF32Bounds kabsch_h2_bounds(int64_t KS2) {
  const double u = 5.9604644775390625e-08, db = kabsch_h2_entry_bound(KS2) + u;
  return {(float)(2.0 * (45.0 * db + 172.0 * u)), (float)(2.0 * (9.5 * db + 46.0 * u)),
          (float)(2.0 * (6.0 * db + 52.0 * u))};
}

// Largest eigenvalue of the quaternion matrix by Newton's iteration on its
// characteristic polynomial, started from the upper bound (Gp+Gq)/2 (monotone
// convergence from above; Theobald's QCP).  rmsd^2 = (Gp + Gq - 2 lambda) / A.
// Used for all-pairs RMSD *values*: no eigenvector, no rotation.  The caller
// re-evaluates pairs with a tiny rmsd exactly (cancellation in Gp+Gq-2*lambda).
__host__ __device__ __forceinline__ double kabsch_lambda_max(const double (&B)[9], double GpGq) {
#pragma clang fp contract(fast)
  const double Sxx = B[0], Sxy = B[1], Sxz = B[2];
  const double Syx = B[3], Syy = B[4], Syz = B[5];
  const double Szx = B[6], Szy = B[7], Szz = B[8];
  const double n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                    Szx * Szx + Szy * Szy + Szz * Szz;
  const double c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const double c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const double c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const double detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const double e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                    c20 * c20 + c21 * c21 + c22 * c22;
  const double C2 = -2.0 * n2, C1 = -8.0 * detB, C0 = n2 * n2 - 4.0 * e2;
  double x = 0.5 * GpGq;
  for (int it = 0; it < 64; ++it) {
    const double x2 = x * x;
    const double b = (x2 + C2) * x;
    const double a = b + C1;
    const double den = 2.0 * x2 * x + b + a;
    if (den == 0.0) break;
    const double delta = (a * x + C0) / den;
    x -= delta;
    if (fabs(delta) <= 4e-16 * fabs(x)) break;
  }
  return x;
}

// Optimal rotation without the Jacobi sweeps, for pairs whose largest quaternion eigenvalue
// is well separated (candidate pairs of the refine: nearly superposable structures): the
// eigenvalue by Newton on the characteristic polynomial (above), its eigenvector as the
// column of adj(K - lambda I) with the largest diagonal cofactor (for a simple eigenvalue
// adj = c q q^T).  Returns false -- and the caller falls back to kabsch_rotation -- when the
// eigenvalue is not clearly simple or the eigenvector leaves a residual above 1e-12 of the
// matrix scale; ~300 flops against ~7 000 for the sweeps.
__host__ __device__ __forceinline__ bool kabsch_rotation_qcp(const double (&B)[9], double GpGq,
                                                             double (&R)[9]) {
  const double lam = kabsch_lambda_max(B, GpGq);
  const double Sxx = B[0], Sxy = B[3], Sxz = B[6];
  const double Syx = B[1], Syy = B[4], Syz = B[7];
  const double Szx = B[2], Szy = B[5], Szz = B[8];
  // M = K - lambda I (symmetric), same K as kabsch_rotation
  const double m00 = (Sxx + Syy + Szz) - lam, m01 = Syz - Szy, m02 = Szx - Sxz, m03 = Sxy - Syx;
  const double m11 = (Sxx - Syy - Szz) - lam, m12 = Sxy + Syx, m13 = Szx + Sxz;
  const double m22 = (-Sxx + Syy - Szz) - lam, m23 = Syz + Szy;
  const double m33 = (-Sxx - Syy + Szz) - lam;
  // adjugate of the symmetric M: A_ij = (-1)^(i+j) det(M without row i and column j)
  const double m[4][4] = {{m00, m01, m02, m03}, {m01, m11, m12, m13}, {m02, m12, m22, m23}, {m03, m13, m23, m33}};
  auto cof = [&](int i, int j) {
    const int r0 = i == 0 ? 1 : 0, r1 = i <= 1 ? 2 : 1, r2 = i <= 2 ? 3 : 2;
    const int c0 = j == 0 ? 1 : 0, c1 = j <= 1 ? 2 : 1, c2 = j <= 2 ? 3 : 2;
    const double d = m[r0][c0] * (m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1]) -
                     m[r0][c1] * (m[r1][c0] * m[r2][c2] - m[r1][c2] * m[r2][c0]) +
                     m[r0][c2] * (m[r1][c0] * m[r2][c1] - m[r1][c1] * m[r2][c0]);
    return ((i + j) & 1) ? -d : d;
  };
  const double a00 = cof(0, 0), a01 = cof(0, 1), a02 = cof(0, 2), a03 = cof(0, 3);
  const double a11 = cof(1, 1), a12 = cof(1, 2), a13 = cof(1, 3);
  const double a22 = cof(2, 2), a23 = cof(2, 3), a33 = cof(3, 3);
  // column with the largest diagonal entry
  double q0 = a00, q1 = a01, q2 = a02, q3 = a03, best = fabs(a00);
  if (fabs(a11) > best) { best = fabs(a11); q0 = a01; q1 = a11; q2 = a12; q3 = a13; }
  if (fabs(a22) > best) { best = fabs(a22); q0 = a02; q1 = a12; q2 = a22; q3 = a23; }
  if (fabs(a33) > best) { best = fabs(a33); q0 = a03; q1 = a13; q2 = a23; q3 = a33; }
  const double scale = fabs(lam) + fabs(Sxx) + fabs(Syy) + fabs(Szz) + fabs(m01) + fabs(m02) + fabs(m03) +
                       fabs(m12) + fabs(m13) + fabs(m23);
  const double n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
  if (!(best > 2e-3 * scale * scale * scale) || !(n2 > 0.0)) return false;  // eigenvalue not clearly simple
  const double nrm = 1.0 / sqrt(n2);
  q0 *= nrm; q1 *= nrm; q2 *= nrm; q3 *= nrm;
  const double e0 = m00 * q0 + m01 * q1 + m02 * q2 + m03 * q3;
  const double e1 = m01 * q0 + m11 * q1 + m12 * q2 + m13 * q3;
  const double e2 = m02 * q0 + m12 * q1 + m22 * q2 + m23 * q3;
  const double e3 = m03 * q0 + m13 * q1 + m23 * q2 + m33 * q3;
  if (!(fmax(fmax(fabs(e0), fabs(e1)), fmax(fabs(e2), fabs(e3))) <= 1e-12 * scale)) return false;
  R[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
  R[1] = 2.0 * (q1 * q2 - q0 * q3);
  R[2] = 2.0 * (q1 * q3 + q0 * q2);
  R[3] = 2.0 * (q1 * q2 + q0 * q3);
  R[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
  R[5] = 2.0 * (q2 * q3 - q0 * q1);
  R[6] = 2.0 * (q1 * q3 - q0 * q2);
  R[7] = 2.0 * (q2 * q3 + q0 * q1);
  R[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
  return true;
}

// Unit quaternion (w, x, y, z) of the optimal rotation, the form the complete-alignment epilogue of the
// all-pairs kernel keeps per pair (four doubles instead of nine) -- same contract as kabsch_rotation_qcp
// (false: eigenvalue not clearly simple or residual above 1e-12 of the matrix scale -> the caller queues
// the pair for the Jacobi sweeps), a third of its instructions:
//   * Newton's quotient through v_rcp_f64: the iteration's fixed point and its stopping test do not depend
//     on how exactly the step is divided, only the step count does (the fp64 division is ~14 instructions);
//   * the ten cofactors of the symmetric K - lambda I from the twelve 2 x 2 minors of its row pairs (0,1)
//     and (2,3): 3 fused operations each instead of a 3 x 3 determinant;
//   * 1/|q| by v_rsq_f64 and two Newton steps.
// Everything here may fuse (the file is built -ffp-contract=off for the literal cdist arithmetic elsewhere).
__host__ __device__ __forceinline__ double fc_rcp_approx(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return 1.0 / x;
#endif
}
__host__ __device__ __forceinline__ double fc_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
#else
  double y = 1.0 / sqrt(x);  // host build (tools / tests)
#endif
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}

// sqrt of a non-negative finite double to within one unit in the last place: v_rsq_f64 and two coupled
// Newton steps (the compiler's sqrt() is correctly rounded and twice as long); 0 -> 0
__host__ __device__ __forceinline__ double fc_sqrt_nonneg(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(fast)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  const double d2 = fma(-g, g, x);
  g = fma(d2, h, g);
  return x > 0.0 ? g : 0.0;
#else
  return sqrt(x);
#endif
}

__host__ __device__ __forceinline__ bool kabsch_quaternion_qcp(const double (&B)[9], double GpGq, double (&Q)[4],
                                                               int *iterations = nullptr) {
#pragma clang fp contract(fast)
  // B[x][y] = sum p_x q_y; Horn's S_xy = sum q_x p_y = B[y][x]
  const double Sxx = B[0], Sxy = B[3], Sxz = B[6];
  const double Syx = B[1], Syy = B[4], Syz = B[7];
  const double Szx = B[2], Szy = B[5], Szz = B[8];
  const double n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                    Szx * Szx + Szy * Szy + Szz * Szz;
  const double c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const double c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const double c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const double detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const double e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                    c20 * c20 + c21 * c21 + c22 * c22;
  const double C2 = -2.0 * n2, C1 = -8.0 * detB, C0 = n2 * n2 - 4.0 * e2;
  // Newton from the upper bound (Gp + Gq)/2: monotone from above (Theobald's QCP)
  double x = 0.5 * GpGq;
  int it = 0;
  for (; it < 64; ++it) {
    const double x2 = x * x;
    const double b = (x2 + C2) * x;
    const double a = b + C1;
    const double den = 2.0 * x2 * x + b + a;
    if (den == 0.0) break;
    const double delta = (a * x + C0) * fc_rcp_approx(den);
    x -= delta;
    // quadratic convergence: a step below 1e-9 |x| leaves an error below the last place (a multiple root,
    // where that does not hold, fails the residual test below and goes to the Jacobi sweeps)
    if (fabs(delta) <= 1e-9 * fabs(x)) break;
  }
  if (iterations) *iterations = it + 1;
  const double lam = x;
  const double m00 = (Sxx + Syy + Szz) - lam, m01 = Syz - Szy, m02 = Szx - Sxz, m03 = Sxy - Syx;
  const double m11 = (Sxx - Syy - Szz) - lam, m12 = Sxy + Syx, m13 = Szx + Sxz;
  const double m22 = (-Sxx + Syy - Szz) - lam, m23 = Syz + Szy;
  const double m33 = (-Sxx - Syy + Szz) - lam;
  // 2 x 2 minors: L_ab of rows (2, 3), U_ab of rows (0, 1), columns a < b
  const double L01 = m02 * m13 - m12 * m03, L02 = m02 * m23 - m22 * m03, L03 = m02 * m33 - m23 * m03;
  const double L12 = m12 * m23 - m22 * m13, L13 = m12 * m33 - m23 * m13, L23 = m22 * m33 - m23 * m23;
  const double U01 = m00 * m11 - m01 * m01, U02 = m00 * m12 - m02 * m01, U03 = m00 * m13 - m03 * m01;
  const double U12 = m01 * m12 - m02 * m11, U13 = m01 * m13 - m03 * m11, U23 = m02 * m13 - m03 * m12;
  (void)U23;
  // cofactors (adjugate of the symmetric matrix, upper triangle)
  const double a00 = m11 * L23 - m12 * L13 + m13 * L12;
  const double a01 = -(m01 * L23 - m12 * L03 + m13 * L02);
  const double a02 = m01 * L13 - m11 * L03 + m13 * L01;
  const double a03 = -(m01 * L12 - m11 * L02 + m12 * L01);
  const double a11 = m00 * L23 - m02 * L03 + m03 * L02;
  const double a12 = -(m00 * L13 - m01 * L03 + m03 * L01);
  const double a13 = m00 * L12 - m01 * L02 + m02 * L01;
  const double a22 = m03 * U13 - m13 * U03 + m33 * U01;
  const double a23 = -(m03 * U12 - m13 * U02 + m23 * U01);
  const double a33 = m02 * U12 - m12 * U02 + m22 * U01;
  double q0 = a00, q1 = a01, q2 = a02, q3 = a03, best = fabs(a00);
  if (fabs(a11) > best) { best = fabs(a11); q0 = a01; q1 = a11; q2 = a12; q3 = a13; }
  if (fabs(a22) > best) { best = fabs(a22); q0 = a02; q1 = a12; q2 = a22; q3 = a23; }
  if (fabs(a33) > best) { best = fabs(a33); q0 = a03; q1 = a13; q2 = a23; q3 = a33; }
  const double scale = fabs(lam) + fabs(Sxx) + fabs(Syy) + fabs(Szz) + fabs(m01) + fabs(m02) + fabs(m03) +
                       fabs(m12) + fabs(m13) + fabs(m23);
  const double nq = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
  if (!(best > 2e-3 * scale * scale * scale) || !(nq > 0.0)) return false;  // eigenvalue not clearly simple
  const double nrm = fc_rsqrt(nq);
  q0 *= nrm; q1 *= nrm; q2 *= nrm; q3 *= nrm;
  const double r0 = m00 * q0 + m01 * q1 + m02 * q2 + m03 * q3;
  const double r1 = m01 * q0 + m11 * q1 + m12 * q2 + m13 * q3;
  const double r2 = m02 * q0 + m12 * q1 + m22 * q2 + m23 * q3;
  const double r3 = m03 * q0 + m13 * q1 + m23 * q2 + m33 * q3;
  if (!(fmax(fmax(fabs(r0), fabs(r1)), fmax(fabs(r2), fabs(r3))) <= 1e-12 * scale)) return false;
  Q[0] = q0; Q[1] = q1; Q[2] = q2; Q[3] = q3;
  return true;
}

// The form the all-pairs complete-alignment kernel uses (k_simbits_screen_mfma<., 2>; every lane of the wavefront
// active).  Same eigenvalue iteration and adjugate as kabsch_quaternion_qcp; what differs is instruction count --
// that kernel is bound by fp64 issue, 4 rotations per lane and 16 x 16 sub-tile:
//   * the Newton loop has a WAVE-UNIFORM trip count: every lane iterates until all 64 have converged (a lane at
//     its fixed point moves by rounding noise).  The per-lane form cost two v_cndmask, an exec-mask update and a
//     vector counter per iteration for nothing: the masked-off lanes' issue slots are spent either way;
//   * "eigenvalue clearly simple" is tested against (3 |B|_F)^3 -- squared, |B|_F^2 is at hand from the characteristic
//     polynomial -- instead of the cube of the sum of the absolute entries of K - lambda I (|lambda| <= sqrt(3) |B|_F,
//     K's entries are sums and differences of B's: that sum lies in [|B|_F, 12 |B|_F], 2.5 - 4 |B|_F in practice):
//     five multiplications instead of thirteen instructions, and like the original it follows the MATRIX, not
//     (Gp + Gq)/2 (for unrelated structures |B|_F is far below that and a test against it declines every pair);
//   * no residual test: with a converged simple eigenvalue the chosen adjugate column IS the eigenvector up to
//     rounding amplified by 1 / (relative gap product) <= 1 / 2e-3; what the residual test caught beyond that was a
//     Newton iteration that had not converged, which the returned flag now says directly (NaN included).
// tools/qcp_host_check.cpp compares both forms with the Jacobi sweeps.
// The three parts of it, so that a caller with several pairs per lane can run their Newton iterations in ONE loop
// (independent chains side by side instead of one dependent chain at a time):
struct QcpLeanPoly {
  double C2, C1, C0, n2;  // lambda^4 + C2 lambda^2 + C1 lambda + C0;  n2 = |B|_F^2
};
__host__ __device__ __forceinline__ QcpLeanPoly qcp_lean_polynomial(const double (&B)[9]) {
#pragma clang fp contract(fast)
  const double Sxx = B[0], Sxy = B[3], Sxz = B[6];
  const double Syx = B[1], Syy = B[4], Syz = B[7];
  const double Szx = B[2], Szy = B[5], Szz = B[8];
  const double n2 = Sxx * Sxx + Sxy * Sxy + Sxz * Sxz + Syx * Syx + Syy * Syy + Syz * Syz +
                    Szx * Szx + Szy * Szy + Szz * Szz;
  const double c00 = Syy * Szz - Syz * Szy, c01 = Syz * Szx - Syx * Szz, c02 = Syx * Szy - Syy * Szx;
  const double c10 = Sxz * Szy - Sxy * Szz, c11 = Sxx * Szz - Sxz * Szx, c12 = Sxy * Szx - Sxx * Szy;
  const double c20 = Sxy * Syz - Sxz * Syy, c21 = Sxz * Syx - Sxx * Syz, c22 = Sxx * Syy - Sxy * Syx;
  const double detB = Sxx * c00 + Sxy * c01 + Sxz * c02;
  const double e2 = c00 * c00 + c01 * c01 + c02 * c02 + c10 * c10 + c11 * c11 + c12 * c12 +
                    c20 * c20 + c21 * c21 + c22 * c22;
  return QcpLeanPoly{-2.0 * n2, -8.0 * detB, n2 * n2 - 4.0 * e2, n2};
}
// one Newton step from above; returns "this lane is still moving" (false for NaN: den == 0 -- an all-zero padding
// conformer, an exact multiple root -- counts as done here and fails in qcp_lean_quaternion)
__host__ __device__ __forceinline__ bool qcp_lean_step(const QcpLeanPoly &P, double &x, double &delta) {
#pragma clang fp contract(fast)
  const double x2 = x * x;
  const double b = (x2 + P.C2) * x;
  const double a = b + P.C1;
  const double den = 2.0 * x2 * x + b + a;
  delta = (a * x + P.C0) * fc_rcp_approx(den);
  x -= delta;
  return fabs(delta) > 1e-9 * fabs(x);
}
// (the adjugate column as it is, not normalised, and its squared length)
__host__ __device__ __forceinline__ bool qcp_lean_quaternion_raw(const double (&B)[9], const QcpLeanPoly &P, double lam,
                                                                 double delta, double (&Q)[4], double &nq_out) {
#pragma clang fp contract(fast)
  const double Sxx = B[0], Sxy = B[3], Sxz = B[6];
  const double Syx = B[1], Syy = B[4], Syz = B[7];
  const double Szx = B[2], Szy = B[5], Szz = B[8];
  const bool converged = fabs(delta) <= 1e-9 * fabs(lam);  // false for NaN
  const double m00 = (Sxx + Syy + Szz) - lam, m01 = Syz - Szy, m02 = Szx - Sxz, m03 = Sxy - Syx;
  const double m11 = (Sxx - Syy - Szz) - lam, m12 = Sxy + Syx, m13 = Szx + Sxz;
  const double m22 = (-Sxx + Syy - Szz) - lam, m23 = Syz + Szy;
  const double m33 = (-Sxx - Syy + Szz) - lam;
  const double L01 = m02 * m13 - m12 * m03, L02 = m02 * m23 - m22 * m03, L03 = m02 * m33 - m23 * m03;
  const double L12 = m12 * m23 - m22 * m13, L13 = m12 * m33 - m23 * m13, L23 = m22 * m33 - m23 * m23;
  const double U01 = m00 * m11 - m01 * m01, U02 = m00 * m12 - m02 * m01, U03 = m00 * m13 - m03 * m01;
  const double U12 = m01 * m12 - m02 * m11, U13 = m01 * m13 - m03 * m11;
  const double a00 = m11 * L23 - m12 * L13 + m13 * L12;
  const double a01 = -(m01 * L23 - m12 * L03 + m13 * L02);
  const double a02 = m01 * L13 - m11 * L03 + m13 * L01;
  const double a03 = -(m01 * L12 - m11 * L02 + m12 * L01);
  const double a11 = m00 * L23 - m02 * L03 + m03 * L02;
  const double a12 = -(m00 * L13 - m01 * L03 + m03 * L01);
  const double a13 = m00 * L12 - m01 * L02 + m02 * L01;
  const double a22 = m03 * U13 - m13 * U03 + m33 * U01;
  const double a23 = -(m03 * U12 - m13 * U02 + m23 * U01);
  const double a33 = m02 * U12 - m12 * U02 + m22 * U01;
  double q0 = a00, q1 = a01, q2 = a02, q3 = a03, best = fabs(a00);
  if (fabs(a11) > best) { best = fabs(a11); q0 = a01; q1 = a11; q2 = a12; q3 = a13; }
  if (fabs(a22) > best) { best = fabs(a22); q0 = a02; q1 = a12; q2 = a22; q3 = a23; }
  if (fabs(a33) > best) { best = fabs(a33); q0 = a03; q1 = a13; q2 = a23; q3 = a33; }
  const double nq = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
  // best > 2e-3 (3 |B|_F)^3, squared: no square root
  const bool simple = best * best > (2e-3 * 27.0) * (2e-3 * 27.0) * (P.n2 * P.n2) * P.n2;
  Q[0] = q0; Q[1] = q1; Q[2] = q2; Q[3] = q3;
  nq_out = nq;
  return converged && simple && nq > 0.0;
}
__host__ __device__ __forceinline__ bool qcp_lean_quaternion(const double (&B)[9], const QcpLeanPoly &P, double lam,
                                                             double delta, double (&Q)[4]) {
  double nq;
  const bool ok = qcp_lean_quaternion_raw(B, P, lam, delta, Q, nq);
  const double nrm = fc_rsqrt(nq);
  Q[0] *= nrm; Q[1] *= nrm; Q[2] *= nrm; Q[3] *= nrm;
  return ok;
}

__host__ __device__ __forceinline__ bool kabsch_quaternion_qcp_lean(const double (&B)[9], double GpGq, double (&Q)[4],
                                                                    int *iterations = nullptr) {
  const QcpLeanPoly P = qcp_lean_polynomial(B);
  double x = 0.5 * GpGq, delta = 0.0;
  int it = 0;
  for (; it < 64; ++it) {
    const bool moving = qcp_lean_step(P, x, delta);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(moving) == 0ull) break;
#else
    if (!moving) break;
#endif
  }
  if (iterations) *iterations = it + 1;
  return qcp_lean_quaternion(B, P, x, delta, Q);
}

// four pairs of one lane: the Newton iterations of all four in one wave-uniform loop; the quaternions come back
// NOT normalised with their squared lengths (neg_rotation_from_raw_quaternion divides once)
// lam_out (optional): the converged eigenvalues -- sum |p - R q|^2 = (Gp + Gq) - 2 lambda for the optimal rotation
__host__ __device__ __forceinline__ void kabsch_quaternion_qcp_lean4(const double (&B)[4][9], const double (&GpGq)[4],
                                                                     double (&Q)[4][4], double (&nq)[4], bool (&ok)[4],
                                                                     double *lam_out = nullptr) {
  QcpLeanPoly P[4];
  double x[4], delta[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    P[r] = qcp_lean_polynomial(B[r]);
    x[r] = 0.5 * GpGq[r];
  }
  for (int it = 0; it < 64; ++it) {
    bool moving = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) moving |= qcp_lean_step(P[r], x[r], delta[r]);
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(moving) == 0ull) break;
#else
    if (!moving) break;
#endif
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) ok[r] = qcp_lean_quaternion_raw(B[r], P[r], x[r], delta[r], Q[r], nq[r]);
  if (lam_out != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) lam_out[r] = x[r];
  }
}

// R (row-major) from a unit quaternion, same formulas as kabsch_rotation
__host__ __device__ __forceinline__ void rotation_from_quaternion(const double (&Q)[4], double (&R)[9]) {
#pragma clang fp contract(fast)
  const double q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3];
  R[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3;
  R[1] = 2.0 * (q1 * q2 - q0 * q3);
  R[2] = 2.0 * (q1 * q3 + q0 * q2);
  R[3] = 2.0 * (q1 * q2 + q0 * q3);
  R[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3;
  R[5] = 2.0 * (q2 * q3 - q0 * q1);
  R[6] = 2.0 * (q1 * q3 - q0 * q2);
  R[7] = 2.0 * (q2 * q3 + q0 * q1);
  R[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
}

// -R of the same quaternion (the complete-alignment epilogue forms p - R q as fused chains on -R)
__host__ __device__ __forceinline__ void neg_rotation_from_quaternion(const double (&Q)[4], double (&R)[9]) {
#pragma clang fp contract(fast)
  const double q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3];
  R[0] = q2 * q2 + q3 * q3 - q0 * q0 - q1 * q1;
  R[1] = -2.0 * (q1 * q2 - q0 * q3);
  R[2] = -2.0 * (q1 * q3 + q0 * q2);
  R[3] = -2.0 * (q1 * q2 + q0 * q3);
  R[4] = q1 * q1 + q3 * q3 - q0 * q0 - q2 * q2;
  R[5] = -2.0 * (q2 * q3 - q0 * q1);
  R[6] = -2.0 * (q1 * q3 - q0 * q2);
  R[7] = -2.0 * (q2 * q3 + q0 * q1);
  R[8] = q1 * q1 + q2 * q2 - q0 * q0 - q3 * q3;
}

// -R from a quaternion of squared length nq > 0 that is NOT normalised: R = I - (2 / nq) (...), one reciprocal
// (v_rcp_f64 + two Newton steps) instead of the inverse square root and four scalings: 29 instructions against 42
__host__ __device__ __forceinline__ void neg_rotation_from_raw_quaternion(const double (&Q)[4], double nq, double (&R)[9]) {
#pragma clang fp contract(fast)
  const double q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3];
  double inv = fc_rcp_approx(nq);
  inv = fma(fma(-nq, inv, 1.0), inv, inv);
  inv = fma(fma(-nq, inv, 1.0), inv, inv);
  const double t = inv + inv;
  const double p11 = q1 * q1, p22 = q2 * q2;
  R[0] = fma(t, fma(q3, q3, p22), -1.0);
  R[4] = fma(t, fma(q3, q3, p11), -1.0);
  R[8] = fma(t, p11 + p22, -1.0);
  const double a12 = q1 * q2, a13 = q1 * q3, a23 = q2 * q3;
  R[1] = t * fma(q0, q3, -a12);    // -2 (q1 q2 - q0 q3) / nq
  R[3] = -t * fma(q0, q3, a12);    // -2 (q1 q2 + q0 q3) / nq
  R[2] = -t * fma(q0, q2, a13);    // -2 (q1 q3 + q0 q2) / nq
  R[6] = t * fma(q0, q2, -a13);    // -2 (q1 q3 - q0 q2) / nq
  R[5] = t * fma(q0, q1, -a23);    // -2 (q2 q3 - q0 q1) / nq
  R[7] = -t * fma(q0, q1, a23);    // -2 (q2 q3 + q0 q1) / nq
}

}  // namespace fc
