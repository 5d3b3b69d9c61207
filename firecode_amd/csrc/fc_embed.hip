// fc_embed.hip -- bimolecular rigid ("cyclical") embed on gfx950: pose
// transforms, pre-transformed conformer tables and the all-pose clash grid.
//
// Replaces the pose loop of firecode/embeds.py:597-727
// (_fast_bimol_rigid_cyclical_embed) up to and including the
// compenetration_check of embeds.py:718.
//
// Observation that shapes the kernels: for molecule i the rigid transform of a
// pose depends only on (conformer c, orientation o, own step angle) --
// polygonize() places a segment of the molecule's OWN pivot length on the x
// axis (firecode/utils.py:265-272), align_vec_pair sees only that molecule's
// vectors (embeds.py:677-680) and mean(vec_pair) is the origin.  So the
// n1*n2*2*na1*na2 poses are the cross product of two small tables of
// pre-transformed structures (n_i * 2 * na_i each), and the clash test is an
// all-pairs distance count between the tables, restricted to equal o.
#include "fc_common.h"
#include "fc_kabsch_math.h"

#include <algorithm>

namespace fc {

double sq_threshold_lt(double t);  // fc_clash.hip

// rot_mat_from_pointer (prism_pruner.algebra): scalar-last quaternion -> matrix
__device__ __forceinline__ void rot_axis_angle(double ax, double ay, double az, double angle_deg,
                                               double (&M)[9]) {
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

__device__ __forceinline__ void mat_vec(const double (&M)[9], double x, double y, double z,
                                        double &ox, double &oy, double &oz) {
  ox = (M[0] * x + M[1] * y) + M[2] * z;
  oy = (M[3] * x + M[4] * y) + M[5] * z;
  oz = (M[6] * x + M[7] * y) + M[8] * z;
}

// one lane per (conformer c, orientation o, angle index ai) of molecule `mol`
// (0 or 1): R (9) and t (3) exactly as embeds.py:649-709 builds them.
__global__ void __launch_bounds__(64)
k_embed_mol_transforms(const double *__restrict__ coords, int64_t n, int64_t A,
                       const int64_t *__restrict__ reactive, int nr,
                       const double *__restrict__ piv_start, const double *__restrict__ piv_end,
                       int mol, const double *__restrict__ angles, int64_t na,
                       double *__restrict__ R_out, double *__restrict__ t_out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * 2 * na) return;
  const int64_t ai = g % na;
  const int o = (int)((g / na) % 2);
  const int64_t c = g / (2 * na);
  const double *x = coords + c * A * 3;
  const double sx = piv_start[c * 3], sy = piv_start[c * 3 + 1], sz = piv_start[c * 3 + 2];
  const double ex = piv_end[c * 3], ey = piv_end[c * 3 + 1], ez = piv_end[c * 3 + 2];
  // Pivot.pivot = start - end, Pivot.meanpoint = mean((start, end))  (hypermolecule_class.py:329-333)
  const double pvx = sx - ex, pvy = sy - ey, pvz = sz - ez;
  const double mx = (sx + ex) / 2.0, my = (sy + ey) / 2.0, mz = (sz + ez) / 2.0;
  const double norm = sqrt((pvx * pvx + pvy * pvy) + pvz * pvz);
  // polygonize(): segment (-norm/2,0,0) -> (+norm/2,0,0); orientation 1 flips molecule 1
  const double sgn = (o == 1 && mol == 1) ? -1.0 : 1.0;
  const double esx = sgn * norm;  // end - start, x component (y = z = 0)
  const double diry = (mol == 0) ? 1.0 : -1.0;
  // mean position of the reactive atoms
  double ax = 0.0, ay = 0.0, az = 0.0;
  for (int k = 0; k < nr; ++k) {
    const int64_t r = reactive[k];
    ax += x[r * 3];
    ay += x[r * 3 + 1];
    az += x[r * 3 + 2];
  }
  ax /= (double)nr;
  ay /= (double)nr;
  az /= (double)nr;
  double dx = mx - ax, dy = my - ay, dz = mz - az;
  if (dx == 0.0 && dy == 0.0 && dz == 0.0) {
    dx = mx;
    dy = my;
    dz = mz;
  }
  // align_vec_pair(ref = [end-start, direction], tgt = [pivot, mol_direction])
  double B[9];
  // B[x][y] = sum_j ref[j][x] * tgt[j][y] with ref = [(esx,0,0), (0,diry,0)]
  B[0] = esx * pvx;  B[1] = esx * pvy;  B[2] = esx * pvz;
  B[3] = diry * dx;  B[4] = diry * dy;  B[5] = diry * dz;
  B[6] = 0.0;        B[7] = 0.0;        B[8] = 0.0;
  double Al[9];
  (void)kabsch_rotation(B, Al);
  // axis of the step rotation
  double ux, uy, uz;
  if (nr == 2) {
    const int64_t r0 = reactive[0], r1 = reactive[1];
    mat_vec(Al, x[r0 * 3] - x[r1 * 3], x[r0 * 3 + 1] - x[r1 * 3 + 1], x[r0 * 3 + 2] - x[r1 * 3 + 2],
            ux, uy, uz);
  } else {
    mat_vec(Al, pvx, pvy, pvz, ux, uy, uz);
  }
  double St[9];
  rot_axis_angle(ux, uy, uz, angles[ai], St);
  double cx, cy, cz;
  mat_vec(Al, ax, ay, az, cx, cy, cz);  // center_of_rotation
  double R[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      R[r * 3 + q] = (St[r * 3] * Al[q] + St[r * 3 + 1] * Al[3 + q]) + St[r * 3 + 2] * Al[6 + q];
  double px, py, pz;
  mat_vec(Al, mx, my, mz, px, py, pz);  // pos = mean(vec_pair) - Al @ meanpoint = 0 - ...
  double scx, scy, scz;
  mat_vec(St, cx, cy, cz, scx, scy, scz);
  double *Ro = R_out + g * 9, *to = t_out + g * 3;
#pragma unroll
  for (int k = 0; k < 9; ++k) Ro[k] = R[k];
  to[0] = (cx - scx) + (0.0 - px);
  to[1] = (cy - scy) + (0.0 - py);
  to[2] = (cz - scz) + (0.0 - pz);
}

// table of pre-transformed structures: out[g][a] = (R_g @ x_a) + t_g, g = (c, o, ai)
// aos != 0: out[g][a][3];  else conformer-minor per orientation:
// out[((o*A + a)*3 + k)*S + (c*na + ai)],  S = n*na rounded up to 64
__global__ void __launch_bounds__(256)
k_embed_pretransform(const double *__restrict__ coords, int64_t n, int64_t A, int64_t na,
                     const double *__restrict__ R, const double *__restrict__ t, int aos,
                     int64_t S, double *__restrict__ out) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n * 2 * na * A) return;
  const int64_t a = tix % A;
  const int64_t g = tix / A;
  const int64_t ai = g % na;
  const int o = (int)((g / na) % 2);
  const int64_t c = g / (2 * na);
  const double *x = coords + (c * A + a) * 3;
  const double *r = R + g * 9, *tt = t + g * 3;
  const double ox = ((r[0] * x[0] + r[1] * x[1]) + r[2] * x[2]) + tt[0];
  const double oy = ((r[3] * x[0] + r[4] * x[1]) + r[5] * x[2]) + tt[1];
  const double oz = ((r[6] * x[0] + r[7] * x[1]) + r[8] * x[2]) + tt[2];
  if (aos) {
    out[(g * A + a) * 3] = ox;
    out[(g * A + a) * 3 + 1] = oy;
    out[(g * A + a) * 3 + 2] = oz;
  } else {
    const int64_t s = c * na + ai;
    out[((o * A + a) * 3 + 0) * S + s] = ox;
    out[((o * A + a) * 3 + 1) * S + s] = oy;
    out[((o * A + a) * 3 + 2) * S + s] = oz;
  }
}

// ---------------------------------------------------------------------------
// k_embed_grid_clash: all poses.  Workgroup = one pre-transformed structure of
// molecule 1 (c1, o, a1) staged in LDS x a strip of molecule-2 structures of
// the same orientation; lane = one molecule-2 structure (conformer-minor
// layout: coalesced 512-byte loads per atom coordinate), atoms of molecule 1
// arrive as LDS broadcasts.  A lane stops counting once it exceeds max_clashes.
// pose index (reference loop order, embeds.py:597-647 with cartesian_product's
// "second argument slowest" order):
//   p = ((c2*n1 + c1)*2 + o) * (na1*na2) + (a2*na1 + a1)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_embed_grid_clash_f64(const double *__restrict__ X1, int64_t n1, int A1, int64_t na1,
                   const double *__restrict__ X2s, int64_t n2, int A2, int64_t na2, int64_t S2,
                   double thr2, int max_clashes, int strips, uint8_t *__restrict__ pass,
                   int32_t *__restrict__ counts) {
  extern __shared__ double s[];
  const int tid = threadIdx.x;
  const int64_t g1 = blockIdx.x;  // (c1, o, a1)
  const int64_t a1 = g1 % na1;
  const int o = (int)((g1 / na1) % 2);
  const int64_t c1 = g1 / (2 * na1);
  for (int k = tid; k < A1 * 3; k += 256) s[k] = X1[g1 * (int64_t)A1 * 3 + k];
  __syncthreads();
  const int64_t n_s2 = n2 * na2;
  const double *__restrict__ base2 = X2s + (int64_t)o * A2 * 3 * S2;
  for (int64_t s2 = (int64_t)blockIdx.y * 256 + tid; s2 < S2; s2 += (int64_t)strips * 256) {
    const bool on = s2 < n_s2;
    int cnt = 0;
    for (int b = 0; b < A2; ++b) {
      const double bx = base2[(int64_t)(b * 3 + 0) * S2 + s2];
      const double by = base2[(int64_t)(b * 3 + 1) * S2 + s2];
      const double bz = base2[(int64_t)(b * 3 + 2) * S2 + s2];
      if (cnt <= max_clashes) {
        for (int a = 0; a < A1; ++a) {
          const double dx = bx - s[a * 3], dy = by - s[a * 3 + 1], dz = bz - s[a * 3 + 2];
          const double d2 = ((dx * dx) + dy * dy) + dz * dz;
          cnt += (d2 < thr2) ? 1 : 0;
        }
      }
      // whole wave past the limit: nothing left to learn
      if (__all(cnt > max_clashes || !on)) break;
    }
    if (on) {
      const int64_t c2 = s2 / na2, a2 = s2 % na2;
      const int64_t p = ((c2 * n1 + c1) * 2 + o) * (na1 * na2) + (a2 * na1 + a1);
      pass[p] = (cnt <= max_clashes) ? 1 : 0;
      if (counts) counts[p] = cnt;
    }
  }
}

// The fp64 arithmetic of the reference decides only the pairs an fp32 pass cannot.  Per
// (molecule-2 atom b, molecule-1 structure) the fp32 pass takes the minimum over atoms a of
//   v = |a^|^2 - 2 b^.a^        (a^, b^: coordinates rounded to fp32; = |b^ - a^|^2 - |b^|^2)
// as three fused multiply-adds per pair (two pairs per packed instruction) and compares it with
// tb = T - |b^|^2, where T is thr2 pushed up by a bound on everything fp32 did, for d < thr:
//   conversion:  | |b^-a^|^2 - d^2 | <= 2 thr sqrt(3) 2 e0 + 12 e0^2,   e0 = M 2^-24
//   evaluation:  |v_fp32 - v|        <= 30 M^2 2^-24   (|a^|^2 rounded once, three fma's on
//                                       partial sums below 9 M^2)
// (M = largest |coordinate| of both tables), taken 2x wide.  v >= tb proves "atom b clashes
// with nothing" for the exact arithmetic too; any other atom is recounted exactly as before
// (same operations, same order), so pass flags and counts are bit-identical to the all-fp64
// kernel kept beside it (FC_GRID_F64=1; the tests compare the two and the oracle).  The band
// grows with M^2: for coordinates beyond a few hundred Angstrom most atoms take the exact path
// (slow, still exact).  NaN / inf coordinates: an fp32 minimum ignores NaN like the exact `<`
// does; |coordinates| of 1e15 and more (also +-inf) send every atom down the exact path.
typedef float fc_f2 __attribute__((ext_vector_type(2)));

__global__ void k_to_f32(const double *__restrict__ x, int64_t n, float *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}

__global__ void k_max_abs(const double *__restrict__ x, int64_t n, unsigned long long *__restrict__ out) {
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = fabs(x[i]);
    if (v > m) m = v;  // false for NaN
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(m, off);
    if (o > m) m = o;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

#ifndef FC_GRID_NB
#define FC_GRID_NB 8
#endif
__global__ void __launch_bounds__(256)
k_embed_grid_clash(const double *__restrict__ X1, int64_t n1, int A1, int64_t na1,
                   const double *__restrict__ X2s, int64_t n2, int A2, int64_t na2, int64_t S2,
                   double thr2, int max_clashes, int strips, const unsigned long long *__restrict__ maxabs,
                   const float *__restrict__ X2f, uint8_t *__restrict__ pass, int32_t *__restrict__ counts) {
  constexpr int NB = FC_GRID_NB;  // molecule-2 atoms per trip through molecule 1
  extern __shared__ double s[];  // A1 x 3 doubles, then four arrays of P1 float pairs (-2x, -2y, -2z, |a|^2)
  const int tid = threadIdx.x;
  const int P1 = (A1 + 1) / 2;
  fc_f2 *SX = reinterpret_cast<fc_f2 *>(s + A1 * 3), *SY = SX + P1, *SZ = SY + P1, *SN = SZ + P1;
  const int64_t g1 = blockIdx.x;  // (c1, o, a1)
  const int64_t a1 = g1 % na1;
  const int o = (int)((g1 / na1) % 2);
  const int64_t c1 = g1 / (2 * na1);
  for (int k = tid; k < A1 * 3; k += 256) s[k] = X1[g1 * (int64_t)A1 * 3 + k];
  for (int k = tid; k < P1 * 2; k += 256) {  // -2 a^ and |a^|^2; an odd last pair gets an atom at "infinity"
    const bool real = k < A1;
    const double *src = X1 + g1 * (int64_t)A1 * 3 + (int64_t)k * 3;
    const float x = real ? (float)src[0] : 0.f, y = real ? (float)src[1] : 0.f, z = real ? (float)src[2] : 0.f;
    reinterpret_cast<float *>(SX)[k] = -2.f * x;
    reinterpret_cast<float *>(SY)[k] = -2.f * y;
    reinterpret_cast<float *>(SZ)[k] = -2.f * z;
    // products of floats are exact in double; the sum is rounded to fp32 once (up to 2 ulp(double))
    reinterpret_cast<float *>(SN)[k] = real ? (float)(((double)x * x + (double)y * y) + (double)z * z) : 3.0e37f;
  }
  __syncthreads();
  float Tf;  // the threshold of the fp32 pass before |b^|^2 is taken off
  {
    const double M = __longlong_as_double((long long)maxabs[0]);
    const double thr = sqrt(thr2) * (1.0 + 0x1p-40);
    const double e0 = M * 0x1p-24;
    // conversion + evaluation of v (header) + |b^|^2 in fp32 (three roundings on sums <= 3 M^2)
    // + the subtraction Tf - |b^|^2 (one rounding, covered by the relative factor), all 2x wide
    double T = thr2 + 2.0 * ((2.0 * thr * 3.47 * e0 + 12.0 * e0 * e0) + (30.0 + 16.0) * M * M * 0x1p-24);
    T *= 1.0 + 0x1p-20;
    Tf = nextafterf((float)T, INFINITY);
    if (!(M < 1.0e15)) Tf = INFINITY;  // fp32 products could overflow (or M is NaN): everything takes the exact path
  }
  const int64_t n_s2 = n2 * na2;
  const double *__restrict__ base2 = X2s + (int64_t)o * A2 * 3 * S2;
  const float *__restrict__ base2f = X2f + (int64_t)o * A2 * 3 * S2;
  for (int64_t s2 = (int64_t)blockIdx.y * 256 + tid; s2 < S2; s2 += (int64_t)strips * 256) {
    const bool on = s2 < n_s2;
    int cnt = 0;
    int resume = on ? 0 : A2;  // first molecule-2 atom this lane has not dealt with
    // Scan passes.  In a pass a lane runs the fp32 test over its remaining atoms until one
    // cannot be ruled out (`pending`), then waits; when no lane scans any more, every waiting
    // lane recounts its atom exactly -- ONE divergent recount per pass and wave instead of one
    // per suspicious atom.  With max_clashes = 0 a second pass happens only when a recount finds
    // no clash after all (a distance within ~1e-6 of the threshold).
    for (;;) {
      int pending = -1;
      int first = resume;
      for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
      if (first >= A2) break;  // every lane is done (wave-uniform)
      for (int b0 = first & ~(NB - 1); b0 < A2; b0 += NB) {
        const bool scanning = pending < 0 && resume < A2;
        if (__all(!scanning)) break;
        fc_f2 bx[NB], by[NB], bz[NB];
        float m[NB], tb[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int b = min(b0 + q, A2 - 1);
          const float x = base2f[(int64_t)(b * 3 + 0) * S2 + s2];
          const float y = base2f[(int64_t)(b * 3 + 1) * S2 + s2];
          const float z = base2f[(int64_t)(b * 3 + 2) * S2 + s2];
          bx[q] = fc_f2{x, x}, by[q] = fc_f2{y, y}, bz[q] = fc_f2{z, z};
          tb[q] = Tf - fmaf(x, x, fmaf(y, y, z * z));  // +inf when Tf is: the atom is recounted
          m[q] = 3.0e38f;                              // finite, so that `m >= +inf` is false
        }
        for (int ap = 0; ap < P1; ++ap) {
          const fc_f2 X = SX[ap], Y = SY[ap], Z = SZ[ap], Nn = SN[ap];
#pragma unroll
          for (int q = 0; q < NB; ++q) {
            fc_f2 v = __builtin_elementwise_fma(bz[q], Z, Nn);
            v = __builtin_elementwise_fma(by[q], Y, v);
            v = __builtin_elementwise_fma(bx[q], X, v);
            m[q] = fminf(fminf(m[q], v.x), v.y);
          }
        }
        if (scanning) {
#pragma unroll
          for (int q = NB - 1; q >= 0; --q)  // the FIRST suspicious atom at or behind `resume` wins
            if (b0 + q < A2 && b0 + q >= resume && !(m[q] >= tb[q])) pending = b0 + q;
          if (pending < 0) resume = max(resume, min(b0 + NB, A2));
        }
      }
      if (pending >= 0) {  // exact recount of that atom: the reference's arithmetic and order
        const double ex = base2[(int64_t)(pending * 3 + 0) * S2 + s2];
        const double ey = base2[(int64_t)(pending * 3 + 1) * S2 + s2];
        const double ez = base2[(int64_t)(pending * 3 + 2) * S2 + s2];
        for (int a = 0; a < A1; ++a) {
          const double dx = ex - s[a * 3], dy = ey - s[a * 3 + 1], dz = ez - s[a * 3 + 2];
          const double d2 = ((dx * dx) + dy * dy) + dz * dz;
          cnt += (d2 < thr2) ? 1 : 0;
        }
        resume = (cnt > max_clashes) ? A2 : pending + 1;  // past the limit: nothing left to learn
      }
    }
    if (on) {
      const int64_t c2 = s2 / na2, a2 = s2 % na2;
      const int64_t p = ((c2 * n1 + c1) * 2 + o) * (na1 * na2) + (a2 * na1 + a1);
      pass[p] = (cnt <= max_clashes) ? 1 : 0;
      if (counts) counts[p] = cnt;
    }
  }
}

// ---------------------------------------------------------------------------
// String embed (firecode/embeds.py:51-158): molecule 1 stays put, molecule 2 is
// rotated so that its orbital vector opposes molecule 1's and spun about that
// axis.  One lane per pose builds (R2, t2):
//   R = rotation_matrix_from_vectors(mol_vec, -ref_vec)        (utils.py:224-249)
//   if angle != 0: R = rot_mat_from_pointer(ref_vec, angle) @ R
//   t = p1 - R @ p2
// pose index (reference loop order; cartesian_product: second argument slowest):
//   p = ((c2*n1 + c1) * (K1*K2) + (k2*K1 + k1)) * nA + ia
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_string_transforms(const double *__restrict__ cen1, const double *__restrict__ vec1, int64_t n1,
                    int64_t K1, const double *__restrict__ cen2, const double *__restrict__ vec2,
                    int64_t n2, int64_t K2, const double *__restrict__ angles, int64_t nA,
                    double *__restrict__ R_out, double *__restrict__ t_out, int64_t *__restrict__ c1_out,
                    int64_t *__restrict__ c2_out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t P = n1 * n2 * K1 * K2 * nA;
  if (p >= P) return;
  const int64_t ia = p % nA;
  const int64_t ki = (p / nA) % (K1 * K2);
  const int64_t ci = p / (nA * K1 * K2);
  const int64_t c1 = ci % n1, c2 = ci / n1, k1 = ki % K1, k2 = ki / K1;
  const double *p1 = cen1 + (c1 * K1 + k1) * 3, *rv = vec1 + (c1 * K1 + k1) * 3;
  const double *p2 = cen2 + (c2 * K2 + k2) * 3, *mv = vec2 + (c2 * K2 + k2) * 3;
  // a = mol_vec / |mol_vec|,  b = -ref_vec / |ref_vec|
  const double na = sqrt((mv[0] * mv[0] + mv[1] * mv[1]) + mv[2] * mv[2]);
  const double nb = sqrt((rv[0] * rv[0] + rv[1] * rv[1]) + rv[2] * rv[2]);
  const double ax = mv[0] / na, ay = mv[1] / na, az = mv[2] / na;
  const double bx = -rv[0] / nb, by = -rv[1] / nb, bz = -rv[2] / nb;
  const double vx = ay * bz - az * by, vy = az * bx - ax * bz, vz = ax * by - ay * bx;
  const double nv = sqrt((vx * vx + vy * vy) + vz * vz);
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (nv != 0.0) {
    const double c = (ax * bx + ay * by) + az * bz;
    const double f = (1.0 - c) / (nv * nv);
    // kmat = [[0,-vz,vy],[vz,0,-vx],[-vy,vx,0]];  R = I + kmat + kmat@kmat * f
    const double k2m[9] = {(0.0 * 0.0 + -vz * vz) + vy * -vy, (0.0 * -vz + -vz * 0.0) + vy * vx, (0.0 * vy + -vz * -vx) + vy * 0.0,
                           (vz * 0.0 + 0.0 * vz) + -vx * -vy, (vz * -vz + 0.0 * 0.0) + -vx * vx, (vz * vy + 0.0 * -vx) + -vx * 0.0,
                           (-vy * 0.0 + vx * vz) + 0.0 * -vy, (-vy * -vz + vx * 0.0) + 0.0 * vx, (-vy * vy + vx * -vx) + 0.0 * 0.0};
    const double km[9] = {0.0, -vz, vy, vz, 0.0, -vx, -vy, vx, 0.0};
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = (R[e] + km[e]) + k2m[e] * f;
  } else {
    const double sx = ax + bx, sy = ay + by, sz = az + bz;
    if (sqrt((sx * sx + sy * sy) + sz * sz) == 0.0) rot_axis_angle(0.0, 0.0, 1.0, 180.0, R);
  }
  const double angle = angles[ia];
  if (angle != 0.0) {
    double D[9], RR[9];
    rot_axis_angle(rv[0], rv[1], rv[2], angle, D);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        RR[r * 3 + q] = (D[r * 3] * R[q] + D[r * 3 + 1] * R[3 + q]) + D[r * 3 + 2] * R[6 + q];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = RR[e];
  }
  double rx, ry, rz;
  mat_vec(R, p2[0], p2[1], p2[2], rx, ry, rz);
#pragma unroll
  for (int e = 0; e < 9; ++e) R_out[p * 9 + e] = R[e];
  t_out[p * 3 + 0] = p1[0] - rx;
  t_out[p * 3 + 1] = p1[1] - ry;
  t_out[p * 3 + 2] = p1[2] - rz;
  c1_out[p] = c1;
  c2_out[p] = c2;
}

// torsion fingerprint of a pose that is never materialised: atoms below A1 come
// from molecule 1 as they are, the others from molecule 2 through (R2, t2)
__device__ __forceinline__ double dihedral4(const double (&p)[4][3]) {
  const double b0x = -1.0 * (p[1][0] - p[0][0]), b0y = -1.0 * (p[1][1] - p[0][1]), b0z = -1.0 * (p[1][2] - p[0][2]);
  double b1x = p[2][0] - p[1][0], b1y = p[2][1] - p[1][1], b1z = p[2][2] - p[1][2];
  const double b2x = p[3][0] - p[2][0], b2y = p[3][1] - p[2][1], b2z = p[3][2] - p[2][2];
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
k_pose_fingerprints(const double *__restrict__ m1, int64_t A1, const double *__restrict__ m2, int64_t A2,
                    const int64_t *__restrict__ c1, const int64_t *__restrict__ c2,
                    const double *__restrict__ R2, const double *__restrict__ t2, int64_t P,
                    const int64_t *__restrict__ quads, int Q, const uint8_t *__restrict__ pass,
                    double *__restrict__ tf) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P * Q) return;
  const int64_t p = g / Q;
  const int q = (int)(g % Q);
  if (pass != nullptr && !pass[p]) {
    tf[g] = 0.0;
    return;
  }
  const double *x1 = m1 + c1[p] * A1 * 3, *x2 = m2 + c2[p] * A2 * 3;
  const double *r = R2 + p * 9, *t = t2 + p * 3;
  double pts[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t a = quads[q * 4 + k];
    if (a < A1) {
      pts[k][0] = x1[a * 3];
      pts[k][1] = x1[a * 3 + 1];
      pts[k][2] = x1[a * 3 + 2];
    } else {
      const double *x = x2 + (a - A1) * 3;
      pts[k][0] = ((r[0] * x[0] + r[1] * x[1]) + r[2] * x[2]) + t[0];
      pts[k][1] = ((r[3] * x[0] + r[4] * x[1]) + r[5] * x[2]) + t[1];
      pts[k][2] = ((r[6] * x[0] + r[7] * x[1]) + r[8] * x[2]) + t[2];
    }
  }
  tf[g] = dihedral4(pts);
}

int launch_string_transforms(const double *cen1, const double *vec1, int64_t n1, int64_t K1,
                             const double *cen2, const double *vec2, int64_t n2, int64_t K2,
                             const double *angles, int64_t nA, double *R_dev, double *t_dev,
                             int64_t *c1_dev, int64_t *c2_dev) {
  const int64_t P = n1 * n2 * K1 * K2 * nA;
  if (P == 0) return FC_OK;
  hipLaunchKernelGGL(k_string_transforms, dim3((unsigned)ceil_div(P, 256)), dim3(256), 0, ctx().stream,
                     cen1, vec1, n1, K1, cen2, vec2, n2, K2, angles, nA, R_dev, t_dev, c1_dev, c2_dev);
  return check_launch("k_string_transforms");
}

int launch_pose_fingerprints(const double *m1, int64_t A1, const double *m2, int64_t A2, const int64_t *c1,
                             const int64_t *c2, const double *R2, const double *t2, int64_t P,
                             const int64_t *quads, int64_t Q, const uint8_t *pass, double *tf) {
  if (P * Q == 0) return FC_OK;
  hipLaunchKernelGGL(k_pose_fingerprints, dim3((unsigned)ceil_div(P * Q, 256)), dim3(256), 0, ctx().stream,
                     m1, A1, m2, A2, c1, c2, R2, t2, P, quads, (int)Q, pass, tf);
  return check_launch("k_pose_fingerprints");
}

// ---------------------------------------------------------------------------
// k_embed_group_dedupe: the `rmsd_similarity(embedded_structure, angular_poses,
// rmsd_thr=1)` filter of embeds.py:723 -- inside one (conformer pair,
// orientation) group the poses are visited in angle order and a clash-free pose
// is kept only if no previously kept pose of the group has
// rmsd < thr and maxdev < 2*thr to it (Kabsch without centring, utils.py:494).
// One wavefront per group; the sequential walk over the <= na1*na2 poses stays
// in the wave, the comparison against the poses kept so far is lane-parallel
// (lane = kept pose).  A pose is the concatenation of one pre-transformed
// structure of each molecule, so the 3x3 covariance of two poses is the sum of
// the two per-molecule covariances -- no pose is materialised.
// X1a / X2a: AoS tables [g][atom][3], g = (c*2 + o)*na + a.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_cov(const double *__restrict__ p, const double *__restrict__ q,
                                               int A, double (&B)[9]) {
  for (int a = 0; a < A; ++a) {
    const double px = p[a * 3], py = p[a * 3 + 1], pz = p[a * 3 + 2];
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    B[0] = fma(px, qx, B[0]); B[1] = fma(px, qy, B[1]); B[2] = fma(px, qz, B[2]);
    B[3] = fma(py, qx, B[3]); B[4] = fma(py, qy, B[4]); B[5] = fma(py, qz, B[5]);
    B[6] = fma(pz, qx, B[6]); B[7] = fma(pz, qy, B[7]); B[8] = fma(pz, qz, B[8]);
  }
}
__device__ __forceinline__ void accumulate_dev(const double *__restrict__ p, const double *__restrict__ q,
                                               int A, const double (&R)[9], double &ssq, double &mx) {
  for (int a = 0; a < A; ++a) {
    const double qx = q[a * 3], qy = q[a * 3 + 1], qz = q[a * 3 + 2];
    const double dx = p[a * 3] - (R[0] * qx + R[1] * qy + R[2] * qz);
    const double dy = p[a * 3 + 1] - (R[3] * qx + R[4] * qy + R[5] * qz);
    const double dz = p[a * 3 + 2] - (R[6] * qx + R[7] * qy + R[8] * qz);
    const double s = dx * dx + dy * dy + dz * dz;
    ssq += s;
    mx = fmax(mx, s);
  }
}

__global__ void __launch_bounds__(256)
k_embed_group_dedupe(const double *__restrict__ X1a, int64_t n1, int A1, int64_t na1,
                     const double *__restrict__ X2a, int64_t n2, int A2, int64_t na2, double thr,
                     const uint8_t *__restrict__ pass, uint8_t *__restrict__ accept) {
  extern __shared__ int kept_all[];  // per wave: indices (angle pair) of the poses kept so far
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t n_ang = na1 * na2;
  int *kept = kept_all + (size_t)wv * n_ang;
  const int64_t n_groups = n1 * n2 * 2;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int o = (int)(grp & 1);
    const int64_t ci = grp >> 1;            // = c2*n1 + c1
    const int64_t c1 = ci % n1, c2 = ci / n1;
    const int64_t pbase = grp * n_ang;
    int n_kept = 0;
    for (int64_t ai = 0; ai < n_ang; ++ai) {
      const int64_t p = pbase + ai;
      if (!pass[p]) {  // wave-uniform
        if (lane == 0) accept[p] = 0;
        continue;
      }
      const int64_t a1 = ai % na1, a2 = ai / na1;
      const double *p1 = X1a + (((c1 * 2 + o) * na1 + a1) * (int64_t)A1) * 3;
      const double *p2 = X2a + (((c2 * 2 + o) * na2 + a2) * (int64_t)A2) * 3;
      bool similar = false;
      for (int k0 = 0; k0 < n_kept && !similar; k0 += 64) {
        const int kk = k0 + lane;
        bool hit = false;
        if (kk < n_kept) {
          const int64_t bi = kept[kk];
          const int64_t b1 = bi % na1, b2 = bi / na1;
          const double *q1 = X1a + (((c1 * 2 + o) * na1 + b1) * (int64_t)A1) * 3;
          const double *q2 = X2a + (((c2 * 2 + o) * na2 + b2) * (int64_t)A2) * 3;
          double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          accumulate_cov(p1, q1, A1, B);
          accumulate_cov(p2, q2, A2, B);
          double R[9];
          (void)kabsch_rotation(B, R);
          double ssq = 0.0, mx = 0.0;
          accumulate_dev(p1, q1, A1, R, ssq, mx);
          accumulate_dev(p2, q2, A2, R, ssq, mx);
          const double rmsd = sqrt(ssq / (double)(A1 + A2)), maxdev = sqrt(mx);
          hit = (rmsd < thr) && (maxdev < 2.0 * thr);
        }
        similar = __any(hit);
      }
      if (!similar) {
        if (lane == 0) kept[n_kept] = (int)ai;
        ++n_kept;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
      }
      if (lane == 0) accept[p] = similar ? 0 : 1;
    }
  }
}

int launch_embed_group_dedupe(const double *X1a, int64_t n1, int64_t A1, int64_t na1, const double *X2a,
                              int64_t n2, int64_t A2, int64_t na2, double thr, const uint8_t *pass_dev,
                              uint8_t *accept_dev) {
  const int64_t groups = n1 * n2 * 2;
  if (groups == 0) return FC_OK;
  int64_t blocks = ceil_div(groups, 4);
  const int64_t cap = (int64_t)ctx().n_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_embed_group_dedupe, dim3((unsigned)blocks), dim3(256),
                     (size_t)4 * na1 * na2 * sizeof(int), ctx().stream, X1a, n1, (int)A1, na1, X2a, n2,
                     (int)A2, na2, thr, pass_dev, accept_dev);
  return check_launch("k_embed_group_dedupe");
}

// ---------------------------------------------------------------------------
int launch_embed_mol_transforms(const double *coords_dev, int64_t n, int64_t A,
                                const int64_t *reactive_dev, int nr, const double *ps_dev,
                                const double *pe_dev, int mol, const double *angles_dev, int64_t na,
                                double *R_dev, double *t_dev) {
  const int64_t total = n * 2 * na;
  if (total == 0) return FC_OK;
  hipLaunchKernelGGL(k_embed_mol_transforms, dim3((unsigned)ceil_div(total, 64)), dim3(64), 0,
                     ctx().stream, coords_dev, n, A, reactive_dev, nr, ps_dev, pe_dev, mol,
                     angles_dev, na, R_dev, t_dev);
  return check_launch("k_embed_mol_transforms");
}

int launch_embed_pretransform(const double *coords_dev, int64_t n, int64_t A, int64_t na,
                              const double *R_dev, const double *t_dev, int aos, int64_t S,
                              double *out_dev) {
  const int64_t total = n * 2 * na * A;
  if (total == 0) return FC_OK;
  hipLaunchKernelGGL(k_embed_pretransform, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                     ctx().stream, coords_dev, n, A, na, R_dev, t_dev, aos, S, out_dev);
  return check_launch("k_embed_pretransform");
}

int launch_embed_grid_clash(const double *X1_dev, int64_t n1, int64_t A1, int64_t na1,
                            const double *X2s_dev, int64_t n2, int64_t A2, int64_t na2, int64_t S2,
                            double thresh, int64_t max_clashes, void *scratch, size_t scratch_bytes,
                            uint8_t *pass_dev, int32_t *counts_dev) {
  const int64_t g1 = n1 * 2 * na1;
  if (g1 == 0 || n2 * na2 == 0) return FC_OK;
  const double thr2 = sq_threshold_lt(thresh);
  int64_t strips = ceil_div(S2, 256);
  // enough workgroups to fill the chip, few enough that the LDS stage is amortised
  const int64_t want = ceil_div((int64_t)ctx().n_cu * 16, g1);
  if (strips > want) strips = std::max<int64_t>(want, 1);
  const char *f64_env = getenv("FC_GRID_F64");  // read per call: the tests switch it
  if (f64_env && atoi(f64_env) == 1) {
    hipLaunchKernelGGL(k_embed_grid_clash_f64, dim3((unsigned)g1, (unsigned)strips), dim3(256),
                       (size_t)A1 * 3 * sizeof(double), ctx().stream, X1_dev, n1, (int)A1, na1, X2s_dev,
                       n2, (int)A2, na2, S2, thr2, (int)max_clashes, (int)strips, pass_dev, counts_dev);
    return check_launch("k_embed_grid_clash_f64");
  }
  // scratch: [largest |coordinate| of both tables | fp32 copy of the molecule-2 table]
  const int64_t len1 = g1 * A1 * 3, len2 = 2 * A2 * 3 * S2;
  if (scratch_bytes < 256 + (size_t)len2 * sizeof(float)) return set_error(FC_E_INVALID, "pose grid scratch too small");
  auto *scratch_dev = static_cast<unsigned long long *>(scratch);
  float *X2f_dev = reinterpret_cast<float *>(static_cast<char *>(scratch) + 256);
  FC_HIP_TRY(hipMemsetAsync(scratch_dev, 0, sizeof(unsigned long long), ctx().stream));
  hipLaunchKernelGGL(k_to_f32, dim3((unsigned)ceil_div(len2, 256)), dim3(256), 0, ctx().stream, X2s_dev, len2, X2f_dev);
  hipLaunchKernelGGL(k_max_abs, dim3((unsigned)std::min<int64_t>(ceil_div(len1, 256), 1024)), dim3(256), 0,
                     ctx().stream, X1_dev, len1, scratch_dev);
  hipLaunchKernelGGL(k_max_abs, dim3((unsigned)std::min<int64_t>(ceil_div(len2, 256), 1024)), dim3(256), 0,
                     ctx().stream, X2s_dev, len2, scratch_dev);
  FC_TRY(check_launch("k_max_abs"));
  const size_t lds = (size_t)A1 * 3 * sizeof(double) + (size_t)4 * ((A1 + 1) / 2) * sizeof(fc_f2);
  hipLaunchKernelGGL(k_embed_grid_clash, dim3((unsigned)g1, (unsigned)strips), dim3(256), lds, ctx().stream,
                     X1_dev, n1, (int)A1, na1, X2s_dev, n2, (int)A2, na2, S2, thr2, (int)max_clashes,
                     (int)strips, scratch_dev, X2f_dev, pass_dev, counts_dev);
  return check_launch("k_embed_grid_clash");
}

// fc_warmup(): the first launch from a translation unit makes the runtime load that unit's code object (milliseconds);
// a no-op launch moves that cost out of the first real call
__global__ void k_warm_embed() {}
int warm_embed() {
  hipLaunchKernelGGL(k_warm_embed, dim3(1), dim3(64), 0, ctx().stream);
  return check_launch("k_warm_embed");
}

}  // namespace fc
