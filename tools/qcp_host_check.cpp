// tools/qcp_host_check.cpp -- host-side check of kabsch_quaternion_qcp (fc_kabsch_math.h) against the
// Jacobi sweeps of kabsch_rotation on random conformer pairs: rotation difference, Newton step counts.
// Build: hipcc -O2 -std=c++17 -I include -I firecode_amd/csrc tools/qcp_host_check.cpp -o /tmp/qcp_host_check
#include <cstdio>
#include <random>
#include <vector>
#include <cmath>
#include <algorithm>
#include "fc_kabsch_math.h"

int main(int argc, char **argv) {
  const int A = argc > 1 ? atoi(argv[1]) : 50;
  const int trials = argc > 2 ? atoi(argv[2]) : 200000;
  const double noise = argc > 3 ? atof(argv[3]) : 0.6;  // per-coordinate displacement between the two structures
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g(0.0, 1.0);
  std::vector<double> p(3 * A), q(3 * A);
  long hist[66] = {0};
  double worst = 0.0, worst_lean = 0.0, worst_raw = 0.0;
  long fail = 0, fail_lean = 0, lean_only_ok = 0;
  for (int t = 0; t < trials; ++t) {
    double cp[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
    // q = random rotation of p + noise
    double M[9];
    for (double &m : M) m = g(rng);
    for (int a = 0; a < A; ++a)
      for (int c = 0; c < 3; ++c) p[a * 3 + c] = 3.0 * g(rng);
    // Gram-Schmidt on M rows -> rotation (sign fixed by the cross product)
    double r0[3] = {M[0], M[1], M[2]}, r1[3] = {M[3], M[4], M[5]}, r2[3];
    double n0 = std::sqrt(r0[0] * r0[0] + r0[1] * r0[1] + r0[2] * r0[2]);
    for (double &v : r0) v /= n0;
    double d = r0[0] * r1[0] + r0[1] * r1[1] + r0[2] * r1[2];
    for (int c = 0; c < 3; ++c) r1[c] -= d * r0[c];
    double n1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    for (double &v : r1) v /= n1;
    r2[0] = r0[1] * r1[2] - r0[2] * r1[1];
    r2[1] = r0[2] * r1[0] - r0[0] * r1[2];
    r2[2] = r0[0] * r1[1] - r0[1] * r1[0];
    for (int a = 0; a < A; ++a) {
      const double *x = &p[a * 3];
      q[a * 3 + 0] = r0[0] * x[0] + r0[1] * x[1] + r0[2] * x[2] + noise * g(rng);
      q[a * 3 + 1] = r1[0] * x[0] + r1[1] * x[1] + r1[2] * x[2] + noise * g(rng);
      q[a * 3 + 2] = r2[0] * x[0] + r2[1] * x[1] + r2[2] * x[2] + noise * g(rng);
    }
    for (int a = 0; a < A; ++a)
      for (int c = 0; c < 3; ++c) { cp[c] += p[a * 3 + c] / A; cq[c] += q[a * 3 + c] / A; }
    double B[9] = {0}, G = 0;
    for (int a = 0; a < A; ++a) {
      double x[3], y[3];
      for (int c = 0; c < 3; ++c) { x[c] = p[a * 3 + c] - cp[c]; y[c] = q[a * 3 + c] - cq[c]; G += x[c] * x[c] + y[c] * y[c]; }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) B[i * 3 + j] += x[i] * y[j];
    }
    double Rj[9], Rq[9], Q[4];
    fc::kabsch_rotation(B, Rj);
    int its = 0;
    const bool ok = fc::kabsch_quaternion_qcp(B, G, Q, &its);
    hist[std::min(its, 65)]++;
    {  // the all-pairs kernel's form: no residual test, gap against (4 S)^3
      double Ql[4], Rl[9];
      const bool okl = fc::kabsch_quaternion_qcp_lean(B, G, Ql);
      if (!okl) ++fail_lean;
      else {
        if (!ok) ++lean_only_ok;
        fc::rotation_from_quaternion(Ql, Rl);
        for (int e = 0; e < 9; ++e) worst_lean = std::max(worst_lean, std::fabs(Rl[e] - Rj[e]));
      }
      // ... as the kernel calls it: four pairs per lane (here four copies), -R from the column that is not normalised
      double B4[4][9], G4[4] = {G, G, G, G}, Q44[4][4], nq4[4], nR[9];
      bool ok4[4];
      for (int r = 0; r < 4; ++r)
        for (int e = 0; e < 9; ++e) B4[r][e] = B[e];
      fc::kabsch_quaternion_qcp_lean4(B4, G4, Q44, nq4, ok4);
      if (ok4[3] != okl) { printf("lean4 and lean disagree on acceptance\n"); return 1; }
      if (okl) {
        fc::neg_rotation_from_raw_quaternion(Q44[3], nq4[3], nR);
        for (int e = 0; e < 9; ++e) worst_raw = std::max(worst_raw, std::fabs(-nR[e] - Rj[e]));
      }
    }
    if (!ok) { ++fail; continue; }
    fc::rotation_from_quaternion(Q, Rq);
    double R9[9];
    const bool ok2 = fc::kabsch_rotation_qcp(B, G, R9);
    for (int e = 0; e < 9; ++e) {
      worst = std::max(worst, std::fabs(Rq[e] - Rj[e]));
      if (ok2) worst = std::max(worst, std::fabs(Rq[e] - R9[e]));
    }
  }
  printf("A=%d trials=%d noise=%.2f: worst |R_new - R_jacobi| = %.3e, not-simple/residual fallbacks = %ld\n", A, trials, noise, worst, fail);
  printf("lean form (kabsch_quaternion_qcp_lean): worst |R - R_jacobi| = %.3e, fallbacks = %ld, accepted where the full form declined = %ld\n",
         worst_lean, fail_lean, lean_only_ok);
  printf("four-at-once form + rotation from the raw column: worst |R - R_jacobi| = %.3e\n", worst_raw);
  printf("Newton steps histogram:");
  for (int i = 0; i < 66; ++i) if (hist[i]) printf(" %d:%ld", i, hist[i]);
  printf("\n");
  return 0;
}
