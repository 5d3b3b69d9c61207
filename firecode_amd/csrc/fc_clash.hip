// fc_clash.hip -- all-pairs distance (compenetration / clash) counts and rigid
// roto-translation for gfx950 (wave64, float64).
//
// Replaces firecode/algebra.py:52-54 (count_clashes), firecode/utils.py:507-575
// (compenetration_check), firecode/embeds.py:808-817 (get_embed) and the
// clash test of the rigid-embed loop embeds.py:713-722.
//
// Exactness: SciPy's cdist evaluates sqrt(((dx*dx)+dy*dy)+dz*dz) without FMA
// (SURVEY.md Appendix B).  This file is compiled with -ffp-contract=off and
// writes that expression literally up to the square root; the root itself is
// removed without changing a single decision: IEEE sqrt is monotone and
// correctly rounded, so  fl(sqrt(d2)) < t  <=>  d2 < T(t)  with
// T(t) = min{x : fl(sqrt(x)) >= t}, computed once on the host
// (sq_threshold_lt / sq_threshold_le below).
#include "fc_common.h"

#include <cmath>

namespace fc {

// smallest double x with sqrt(x) >= t   ( d < t  <=>  d2 < x )
double sq_threshold_lt(double t) {
  if (!(t > 0.0)) return 0.0;  // sqrt(d2) < t never holds for t <= 0
  double x = t * t;
  while (std::sqrt(x) >= t) x = std::nextafter(x, 0.0);
  while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
  return x;
}
// largest double x with sqrt(x) <= t   ( d <= t  <=>  d2 <= x )
double sq_threshold_le(double t) {
  if (t < 0.0) return -1.0;
  double x = t * t;
  while (std::sqrt(x) <= t) x = std::nextafter(x, INFINITY);
  while (std::sqrt(x) > t) x = std::nextafter(x, 0.0);
  return x;
}

__device__ __forceinline__ double dist2(double ax, double ay, double az, double bx, double by,
                                        double bz) {
  const double dx = ax - bx, dy = ay - by, dz = az - bz;
  return ((dx * dx) + dy * dy) + dz * dz;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------
// count_clashes: ordered pairs with lo < d < hi.  One wavefront per structure,
// structure staged in LDS, lanes own rows of the upper triangle; d(a,b) and
// d(b,a) have identical bits, so each unordered hit counts twice.
// lo2 = largest x with sqrt(x) <= lo (d > lo <=> d2 > lo2), hi2 = T(hi).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_clash_self(const double *__restrict__ coords, int64_t N, int A, double lo2, double hi2,
             int64_t *__restrict__ counts) {
  extern __shared__ double s[];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;
  const double *x = coords + n * (int64_t)A * 3;
  for (int k = lane; k < A * 3; k += 64) s[k] = x[k];
  __syncthreads();
  int cnt = 0;
  for (int a = lane; a < A; a += 64) {
    const double ax = s[a * 3], ay = s[a * 3 + 1], az = s[a * 3 + 2];
    for (int b = a + 1; b < A; ++b) {
      const double d2 = dist2(ax, ay, az, s[b * 3], s[b * 3 + 1], s[b * 3 + 2]);
      cnt += (d2 < hi2 && d2 > lo2) ? 1 : 0;
    }
  }
  cnt = wave_sum(cnt);
  if (lane == 0) counts[n] = 2 * (int64_t)cnt;
}

// ---------------------------------------------------------------------------
// fragment clashes (bimolecular: strict <, trimolecular: <=, cumulative).
// One wavefront per structure; the (row, col) rectangle of each fragment
// pair is flattened over the lanes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rect_count(const double *s, int r0, int nr, int c0, int nc,
                                          double thr2, bool le, int lane) {
  int cnt = 0;
  const int total = nr * nc;
  for (int p = lane; p < total; p += 64) {
    const int r = r0 + p / nc, c = c0 + p % nc;
    const double d2 = dist2(s[r * 3], s[r * 3 + 1], s[r * 3 + 2], s[c * 3], s[c * 3 + 1], s[c * 3 + 2]);
    cnt += (le ? (d2 <= thr2) : (d2 < thr2)) ? 1 : 0;
  }
  return cnt;
}

__global__ void __launch_bounds__(64)
k_clash_fragments(const double *__restrict__ coords, int64_t N, int A, int n0, int n1, int n2,
                  int n_ids, double thr2, int64_t max_clashes, int64_t *__restrict__ counts,
                  uint8_t *__restrict__ pass) {
  extern __shared__ double s[];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;
  const double *x = coords + n * (int64_t)A * 3;
  for (int k = lane; k < A * 3; k += 64) s[k] = x[k];
  __syncthreads();
  int cnt;
  if (n_ids == 2) {
    // cdist(m2, m1) < thresh : m1 = [0, n0), m2 = [n0, A)
    cnt = rect_count(s, n0, A - n0, 0, n0, thr2, false, lane);
  } else {
    cnt = rect_count(s, n0, n1, 0, n0, thr2, true, lane);                 // (m2, m1)
    cnt += rect_count(s, n0 + n1, A - n0 - n1, n0, n1, thr2, true, lane); // (m3, m2)
    cnt += rect_count(s, 0, n0, n0 + n1, A - n0 - n1, thr2, true, lane);  // (m1, m3)
  }
  cnt = wave_sum(cnt);
  if (lane == 0) {
    if (counts) counts[n] = cnt;
    if (pass) pass[n] = ((int64_t)cnt <= max_clashes) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
// compenetration_check(graph=...) -- firecode/utils.py:528-542.  The
// reference walks argwhere(dist < thresh) in row-major order, tests
// `clashes > max_clashes` BEFORE looking at a pair and counts ordered,
// non-bonded, off-diagonal pairs.  The list always ends with the diagonal
// element (A-1, A-1) (distance 0 < thresh), which is never counted, so the
// early-exit test is reached once more after the last counted pair: the
// function returns  count <= max_clashes  -- no off-by-one survives.
// adj: A x A bytes, adj[i*A + j] != 0 when i-j is a bond.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_clash_graph(const double *__restrict__ coords, int64_t N, int A, const uint8_t *__restrict__ adj,
              double thr2, int64_t *__restrict__ counts) {
  extern __shared__ double s[];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;
  const double *x = coords + n * (int64_t)A * 3;
  for (int k = lane; k < A * 3; k += 64) s[k] = x[k];
  __syncthreads();
  int cnt = 0;
  for (int a = lane; a < A; a += 64) {
    const double ax = s[a * 3], ay = s[a * 3 + 1], az = s[a * 3 + 2];
    for (int b = a + 1; b < A; ++b) {
      const double d2 = dist2(ax, ay, az, s[b * 3], s[b * 3 + 1], s[b * 3 + 2]);
      cnt += (d2 < thr2 && !adj[a * A + b]) ? 1 : 0;
    }
  }
  cnt = wave_sum(cnt);
  if (lane == 0) counts[n] = 2 * (int64_t)cnt;  // (i1,i2) and (i2,i1) are both visited
}

// fitness_check (firecode/optimization_methods.py:163-180): sum over constraints
// of (|x_a - x_b| - target) < threshold; one lane per structure.
// pairs (N, C, 2) int64, targets (N, C) with NaN = "None" (skipped).
__global__ void __launch_bounds__(256)
k_fitness(const double *__restrict__ coords, int64_t N, int64_t A, const int64_t *__restrict__ pairs,
          const double *__restrict__ targets, int64_t C, double threshold,
          double *__restrict__ error_out, uint8_t *__restrict__ pass) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double *x = coords + n * A * 3;
  double err = 0.0;
  for (int64_t c = 0; c < C; ++c) {
    const double tgt = targets[n * C + c];
    if (tgt != tgt) continue;  // NaN: no target for this constraint
    const int64_t a = pairs[(n * C + c) * 2], b = pairs[(n * C + c) * 2 + 1];
    const double dx = x[a * 3] - x[b * 3], dy = x[a * 3 + 1] - x[b * 3 + 1], dz = x[a * 3 + 2] - x[b * 3 + 2];
    err += sqrt((dx * dx + dy * dy) + dz * dz) - tgt;
  }
  if (error_out) error_out[n] = err;
  pass[n] = (err < threshold) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// get_embed: out[k][a] = ((R0*x + R1*y) + R2*z) + t  -- one lane per atom,
// R/t of a block are wave-uniform.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rototranslate(const double *__restrict__ coords, int64_t n, int64_t A,
                const double *__restrict__ R, const double *__restrict__ t,
                double *__restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * A) return;
  const int64_t k = g / A;
  const double *r = R + k * 9;
  const double *tt = t + k * 3;
  const double x = coords[g * 3], y = coords[g * 3 + 1], z = coords[g * 3 + 2];
  out[g * 3 + 0] = ((r[0] * x + r[1] * y) + r[2] * z) + tt[0];
  out[g * 3 + 1] = ((r[3] * x + r[4] * y) + r[5] * z) + tt[1];
  out[g * 3 + 2] = ((r[6] * x + r[7] * y) + r[8] * z) + tt[2];
}

// align_by_moi helpers (hypermolecule_class.py:45-86): centre every structure on its plain
// mean (rows added in order, like np.mean(axis=0)); build the "moment vector" arrays
// diag(I_ref), diag(I_n) the reference feeds to get_alignment_matrix.
__global__ void __launch_bounds__(256)
k_center_structures(const double *__restrict__ coords, int64_t N, int64_t A, double *__restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (structure, component)
  if (g >= N * 3) return;
  const int64_t n = g / 3;
  const int c = (int)(g - n * 3);
  const double *x = coords + n * A * 3 + c;
  double s = 0.0;
  for (int64_t a = 0; a < A; ++a) s += x[a * 3];
  const double m = s / (double)A;
  double *o = out + n * A * 3 + c;
  for (int64_t a = 0; a < A; ++a) o[a * 3] = x[a * 3] - m;
}

__global__ void __launch_bounds__(256)
k_moi_diag_pairs(const double *__restrict__ moments, int64_t N, double *__restrict__ P,
                 double *__restrict__ Q) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const bool d = (e == 0 || e == 4 || e == 8);
    P[n * 9 + e] = d ? moments[e / 4] : 0.0;
    Q[n * 9 + e] = d ? moments[n * 3 + e / 4] : 0.0;
  }
}

// the first structure is copied, not rotated (hypermolecule_class.py:60-61)
__global__ void k_set_identity(double *__restrict__ M) {
  if (threadIdx.x < 9) M[threadIdx.x] = (threadIdx.x % 4 == 0) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------
// Rigid-embed poses: transform + clash count fused, pose never written unless
// asked for.  One wavefront per pose (4 per workgroup): molecule 1 of the pose
// is transformed into the wave's LDS slice (A1*3 doubles) and read back as
// broadcasts; every lane keeps one transformed atom of molecule 2 in
// registers (chunks of 64 atoms) and walks the A1 atoms of molecule 1.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_embed_poses_clash(const double *__restrict__ m1, int A1, const double *__restrict__ m2, int A2,
                    const int64_t *__restrict__ c1, const int64_t *__restrict__ c2,
                    const double *__restrict__ R1, const double *__restrict__ t1,
                    const double *__restrict__ R2, const double *__restrict__ t2, int64_t P,
                    double thr2, int64_t max_clashes, int64_t *__restrict__ counts,
                    uint8_t *__restrict__ pass, double *__restrict__ poses) {
  extern __shared__ double s[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  double *sw = s + (size_t)wv * A1 * 3;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t k = wave0; k < P; k += nwaves) {
    const double *x1 = m1 + c1[k] * (int64_t)A1 * 3;
    const double *x2 = m2 + c2[k] * (int64_t)A2 * 3;
    static const double kEye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, kZero[3] = {0, 0, 0};
    // R1 == nullptr: molecule 1 is not moved (string embed); the identity goes through
    // the same expression and reproduces x exactly
    const double *r1 = R1 ? R1 + k * 9 : kEye, *r2 = R2 + k * 9;
    const double *u1 = t1 ? t1 + k * 3 : kZero, *u2 = t2 + k * 3;
    double *po = poses ? poses + k * (int64_t)(A1 + A2) * 3 : nullptr;
    for (int a = lane; a < A1; a += 64) {
      const double x = x1[a * 3], y = x1[a * 3 + 1], z = x1[a * 3 + 2];
      const double ox = ((r1[0] * x + r1[1] * y) + r1[2] * z) + u1[0];
      const double oy = ((r1[3] * x + r1[4] * y) + r1[5] * z) + u1[1];
      const double oz = ((r1[6] * x + r1[7] * y) + r1[8] * z) + u1[2];
      sw[a * 3] = ox;
      sw[a * 3 + 1] = oy;
      sw[a * 3 + 2] = oz;
      if (po) {
        po[a * 3] = ox;
        po[a * 3 + 1] = oy;
        po[a * 3 + 2] = oz;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible
    int cnt = 0;
    for (int b0 = 0; b0 < A2; b0 += 64) {
      const int b = b0 + lane;
      const bool on = b < A2;
      double bx = 0.0, by = 0.0, bz = 0.0;
      if (on) {
        const double x = x2[b * 3], y = x2[b * 3 + 1], z = x2[b * 3 + 2];
        bx = ((r2[0] * x + r2[1] * y) + r2[2] * z) + u2[0];
        by = ((r2[3] * x + r2[4] * y) + r2[5] * z) + u2[1];
        bz = ((r2[6] * x + r2[7] * y) + r2[8] * z) + u2[2];
        if (po) {
          po[(A1 + b) * 3] = bx;
          po[(A1 + b) * 3 + 1] = by;
          po[(A1 + b) * 3 + 2] = bz;
        }
      }
      // cdist(m2, m1)[b][a] = |m2_b - m1_a|
      for (int a = 0; a < A1; ++a) {
        const double d2 = dist2(bx, by, bz, sw[a * 3], sw[a * 3 + 1], sw[a * 3 + 2]);
        cnt += (on && d2 < thr2) ? 1 : 0;
      }
    }
    cnt = wave_sum(cnt);
    if (lane == 0) {
      if (counts) counts[k] = cnt;
      if (pass) pass[k] = ((int64_t)cnt <= max_clashes) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int launch_clash_self(const double *coords_dev, int64_t N, int64_t A, double lo, double hi,
                      int64_t *counts_dev) {
  if (N == 0) return FC_OK;
  const double lo2 = sq_threshold_le(lo);
  const double hi2 = sq_threshold_lt(hi);
  hipLaunchKernelGGL(k_clash_self, dim3((unsigned)N), dim3(64), (size_t)A * 3 * sizeof(double),
                     ctx().stream, coords_dev, N, (int)A, lo2, hi2, counts_dev);
  return check_launch("k_clash_self");
}

int launch_clash_fragments(const double *coords_dev, int64_t N, int64_t A, const int64_t *ids,
                           int64_t n_ids, double thresh, int64_t max_clashes, int64_t *counts_dev,
                           uint8_t *pass_dev) {
  if (N == 0) return FC_OK;
  const double thr2 = (n_ids == 2) ? sq_threshold_lt(thresh) : sq_threshold_le(thresh);
  hipLaunchKernelGGL(k_clash_fragments, dim3((unsigned)N), dim3(64),
                     (size_t)A * 3 * sizeof(double), ctx().stream, coords_dev, N, (int)A,
                     (int)ids[0], (int)ids[1], n_ids == 3 ? (int)ids[2] : 0, (int)n_ids, thr2,
                     max_clashes, counts_dev, pass_dev);
  return check_launch("k_clash_fragments");
}

int launch_clash_graph(const double *coords_dev, int64_t N, int64_t A, const uint8_t *adj_dev,
                       double thresh, int64_t *counts_dev) {
  if (N == 0) return FC_OK;
  hipLaunchKernelGGL(k_clash_graph, dim3((unsigned)N), dim3(64), (size_t)A * 3 * sizeof(double),
                     ctx().stream, coords_dev, N, (int)A, adj_dev, sq_threshold_lt(thresh), counts_dev);
  return check_launch("k_clash_graph");
}

int launch_fitness(const double *coords_dev, int64_t N, int64_t A, const int64_t *pairs_dev,
                   const double *targets_dev, int64_t C, double threshold, double *err_dev,
                   uint8_t *pass_dev) {
  if (N == 0) return FC_OK;
  hipLaunchKernelGGL(k_fitness, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx().stream,
                     coords_dev, N, A, pairs_dev, targets_dev, C, threshold, err_dev, pass_dev);
  return check_launch("k_fitness");
}

int launch_rototranslate(const double *coords_dev, int64_t n, int64_t A, const double *R_dev,
                         const double *t_dev, double *out_dev) {
  if (n * A == 0) return FC_OK;
  hipLaunchKernelGGL(k_rototranslate, dim3((unsigned)ceil_div(n * A, 256)), dim3(256), 0,
                     ctx().stream, coords_dev, n, A, R_dev, t_dev, out_dev);
  return check_launch("k_rototranslate");
}

int launch_center_structures(const double *coords_dev, int64_t N, int64_t A, double *out_dev) {
  if (N == 0) return FC_OK;
  hipLaunchKernelGGL(k_center_structures, dim3((unsigned)ceil_div(N * 3, 256)), dim3(256), 0, ctx().stream,
                     coords_dev, N, A, out_dev);
  return check_launch("k_center_structures");
}

int launch_moi_diag_pairs(const double *moments_dev, int64_t N, double *P_dev, double *Q_dev) {
  if (N == 0) return FC_OK;
  hipLaunchKernelGGL(k_moi_diag_pairs, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx().stream,
                     moments_dev, N, P_dev, Q_dev);
  return check_launch("k_moi_diag_pairs");
}

int launch_set_identity(double *M_dev) {
  hipLaunchKernelGGL(k_set_identity, dim3(1), dim3(64), 0, ctx().stream, M_dev);
  return check_launch("k_set_identity");
}

int launch_embed_poses_clash(const double *m1_dev, int64_t A1, const double *m2_dev, int64_t A2,
                             const int64_t *c1_dev, const int64_t *c2_dev, const double *R1_dev,
                             const double *t1_dev, const double *R2_dev, const double *t2_dev,
                             int64_t P, double thresh, int64_t max_clashes, int64_t *counts_dev,
                             uint8_t *pass_dev, double *poses_dev) {
  if (P == 0) return FC_OK;
  const double thr2 = sq_threshold_lt(thresh);
  int64_t blocks = ceil_div(P, 4);
  const int64_t cap = (int64_t)ctx().n_cu * 32;  // persistent-style grid, waves stride over poses
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(k_embed_poses_clash, dim3((unsigned)blocks), dim3(256),
                     (size_t)4 * A1 * 3 * sizeof(double), ctx().stream, m1_dev, (int)A1, m2_dev,
                     (int)A2, c1_dev, c2_dev, R1_dev, t1_dev, R2_dev, t2_dev, P, thr2, max_clashes,
                     counts_dev, pass_dev, poses_dev);
  return check_launch("k_embed_poses_clash");
}

// fc_warmup(): the first launch from a translation unit makes the runtime load that unit's code object (milliseconds);
// a no-op launch moves that cost out of the first real call
__global__ void k_warm_clash() {}
int warm_clash() {
  hipLaunchKernelGGL(k_warm_clash, dim3(1), dim3(64), 0, ctx().stream);
  return check_launch("k_warm_clash");
}

}  // namespace fc
