// fc_tfd_ladder.hip -- prune_conformers_tfd's k-ladder on the device (firecode/torsion_module.py:973-1041; round 5).
//
// What a chunk rejects depends on the first-match array alone (fc_tfd_host.cpp, tfd_ladder_impl), so the non-last chunks of
// ALL ladder levels that can run are worked out in ONE set of launches over a (level, chunk) index; only the application
// of the levels is sequential (which levels run and where their last chunk ends follows from the number of structures
// still active), and that is a kernel per level over one byte per structure.  No count travels to the host in between:
// array sizes come from upper bounds, every kernel reads what it needs from device memory.
//
//   * chunks of at most kChunkMax structures (tuple set of at most 8192 slots): one WAVEFRONT per chunk, everything in LDS
//     (fc_tfd_core.h chunk_front) -- edge order, roots, member lists; components of up to 18 nodes finished on the spot;
//   * larger chunks: the same steps as passes over global arrays, all such chunks of all levels at once -- the staged
//     priority first-fit of the tuple sets (one launch set per growth stage: k_c_clear / k_c_insert / k_c_positions),
//     pointer jumping with an early exit, member lists through a scan and one atomic per graph node;
//   * components of more than 18 nodes from both paths are records {members, nodes of the chunk's graph}: a wavefront
//     (up to 306 nodes) or a workgroup (up to 4096) each, comp_group_first;
//   * k_apply: one launch per level in ladder order, each working out for itself from the counts of the launches before
//     it whether its level runs; a level whose LAST chunk (which ends at the number of active structures) could hold an
//     edge raises a flag, and so do components above 4096 nodes: the host then applies the levels from the device's flags
//     (tfd_apply_levels_host), with those components and chunks done there.
//
// The CPython orders being re-enacted are stated in fc_tfd_host.cpp's header; tests/test_tfd_ladder_v2.py runs this file's
// per-chunk and per-component routines on the CPU (host group) against the all-host ladder, tests/test_tfd_gpu_graph.py
// the kernels against the same.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fc_common.h"
#include "fc_tfd_core.h"

namespace fc {

using namespace tfd;

int tfd_apply_levels_host(const int64_t *fm, int64_t N, const std::vector<const uint8_t *> &level_flags, int first_level,
                          const uint8_t *first_last_flags, uint8_t *mask_out, int64_t active_known);  // fc_tfd_host.cpp
uint32_t host_component_first_big(const uint32_t *mx, const uint32_t *mp, const uint32_t *ms, int64_t n, uint32_t n_graph);
int tfd_ladder_host_only(const int64_t *fm, int64_t N, uint8_t *mask_out);

namespace {

// ---- groups ---------------------------------------------------------------------------------------------------------------
struct WaveGroup {
  int tid, size;
  __device__ WaveGroup() : tid((int)(threadIdx.x & 63)), size(64) {}
  __device__ void sync() const {  // LDS operations of one wavefront complete in order: only the compiler must not move them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ uint32_t atomic_min(uint32_t *p, uint32_t v) const { return atomicMin(p, v); }
  __device__ uint32_t atomic_add(uint32_t *p, uint32_t v) const { return atomicAdd(p, v); }
  __device__ uint32_t atomic_or(uint32_t *p, uint32_t v) const { return atomicOr(p, v); }
  __device__ uint32_t atomic_cas(uint32_t *p, uint32_t e, uint32_t v) const { return atomicCAS(p, e, v); }
  __device__ uint16_t atomic_add16(uint16_t *base, int idx, uint16_t v) const {
    uint32_t *w = reinterpret_cast<uint32_t *>(base) + (idx >> 1);
    const int sh = (idx & 1) * 16;
    return (uint16_t)((atomicAdd(w, (uint32_t)v << sh) >> sh) & 0xFFFFu);
  }
  __device__ uint32_t scan_excl(uint32_t v, uint32_t &total) const {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (tid >= o) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
  }
  __device__ uint32_t reduce_sum(uint32_t v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  }
  __device__ uint32_t reduce_or(uint32_t v) const { return __ballot(v != 0u) != 0ull ? 1u : 0u; }
  __device__ uint64_t reduce_min64(uint64_t v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t y = __shfl_xor((unsigned long long)v, o);
      v = y < v ? y : v;
    }
    return v;
  }
  __device__ uint32_t bcast(uint32_t v) const { return __shfl(v, 0); }  // lane 0's value
  __device__ bool in_first_wave() const { return true; }
  __device__ WaveGroup first_wave() const { return *this; }
};

struct BlockGroup {  // a whole workgroup; red: 40 uint64 of LDS for the cross-wavefront steps
  int tid, size;
  uint64_t *red;
  __device__ explicit BlockGroup(uint64_t *r) : tid((int)threadIdx.x), size((int)blockDim.x), red(r) {}
  __device__ void sync() const { __syncthreads(); }
  __device__ uint32_t atomic_min(uint32_t *p, uint32_t v) const { return atomicMin(p, v); }
  __device__ uint32_t atomic_add(uint32_t *p, uint32_t v) const { return atomicAdd(p, v); }
  __device__ uint32_t atomic_or(uint32_t *p, uint32_t v) const { return atomicOr(p, v); }
  __device__ uint32_t atomic_cas(uint32_t *p, uint32_t e, uint32_t v) const { return atomicCAS(p, e, v); }
  __device__ uint16_t atomic_add16(uint16_t *base, int idx, uint16_t v) const {
    uint32_t *w = reinterpret_cast<uint32_t *>(base) + (idx >> 1);
    const int sh = (idx & 1) * 16;
    return (uint16_t)((atomicAdd(w, (uint32_t)v << sh) >> sh) & 0xFFFFu);
  }
  __device__ uint32_t scan_excl(uint32_t v, uint32_t &total) const {
    const int lane = tid & 63, wv = tid >> 6, nw = size >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) red[wv] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < nw; ++w) {
      const uint32_t s = (uint32_t)red[w];
      if (w < wv) before += s;
      all += s;
    }
    __syncthreads();
    total = all;
    return before + x - v;
  }
  __device__ uint64_t reduce_min64(uint64_t v) const {
    const int lane = tid & 63, wv = tid >> 6, nw = size >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t y = __shfl_xor((unsigned long long)v, o);
      v = y < v ? y : v;
    }
    if (lane == 0) red[16 + wv] = v;
    __syncthreads();
    uint64_t r = ~0ull;
    for (int w = 0; w < nw; ++w) r = red[16 + w] < r ? red[16 + w] : r;
    __syncthreads();
    return r;
  }
  __device__ uint32_t reduce_sum(uint32_t v) const {
    uint32_t t;
    (void)scan_excl(v, t);
    return t;
  }
  __device__ uint32_t reduce_or(uint32_t v) const { return reduce_sum(v != 0u ? 1u : 0u) != 0u ? 1u : 0u; }
  __device__ bool in_first_wave() const { return tid < 64; }
  __device__ WaveGroup first_wave() const { return WaveGroup(); }
};

// ---- the batched levels ------------------------------------------------------------------------------------------------
constexpr int kMaxEntries = 20;
struct LadTab {        // one entry = the non-last chunks of a ladder level, or the first level's last chunk
  int n;
  int li[kMaxEntries];          // ladder level (index into the k list) the entry's flags belong to
  int64_t k[kMaxEntries];       // that level's k
  int64_t lo[kMaxEntries];      // first structure
  int64_t d[kMaxEntries];       // chunk length
  int64_t nch[kMaxEntries];     // chunks
  int64_t nit[kMaxEntries];     // items = d * nch
  uint32_t ibase[kMaxEntries];  // first item (multiple of 256)
  uint32_t cbase[kMaxEntries];  // first chunk (all entries) ...
  uint32_t ccbase[kMaxEntries]; // ... and among the coarse entries' chunks
  int coarse[kMaxEntries];
  uint32_t smask[kMaxEntries];   // bit s: a chunk of this entry can reach growth stage s of its tuple set (coarse entries)
};

__device__ __forceinline__ int entry_of_block(const LadTab &T, uint32_t t_block) {  // uniform: ibase are multiples of the block size
  int e = 0;
  while (e + 1 < T.n && t_block >= T.ibase[e + 1]) ++e;
  return e;
}

// ---- small chunks: a wavefront each ----------------------------------------------------------------------------------------
template <int DMAX, int TBL, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_chunk_front(const int64_t *__restrict__ fm, int64_t lo, int d, int64_t n_chunks, uint32_t ibase, uint8_t *__restrict__ flags,
              CompRecord *__restrict__ rec, uint32_t *__restrict__ mx, uint32_t *__restrict__ mp, uint32_t *__restrict__ ms) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  const int wv = (int)(threadIdx.x >> 6);
  const int64_t c = (int64_t)blockIdx.x * WPB + wv;
  if (c >= n_chunks) return;  // (wave-uniform; the routine below synchronises wavefronts only)
  constexpr size_t per_wave = (chunk_local_bytes(DMAX, TBL, 64) + 15) & ~(size_t)15;
  ChunkLocal L;
  chunk_local_carve(lds_raw + (size_t)wv * per_wave, DMAX, TBL, L);
  WaveGroup g;
  const uint32_t t0 = ibase + (uint32_t)(c * d);
  const ChunkExport ex{rec + (t0 >> 4), mx + t0, mp + t0, ms + t0, t0};
  (void)chunk_front(g, L, fm, lo + c * d, d, t0, flags + t0, ex);
}

// the same with a WORKGROUP per chunk (chunks of thousands of structures: one wavefront alone leaves its CU idle most of the time)
template <int DMAX, int TBL>
__global__ void __launch_bounds__(256)
k_chunk_front_block(const int64_t *__restrict__ fm, int64_t lo, int d, int64_t n_chunks, uint32_t ibase, uint8_t *__restrict__ flags,
                    CompRecord *__restrict__ rec, uint32_t *__restrict__ mx, uint32_t *__restrict__ mp, uint32_t *__restrict__ ms) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  __shared__ uint64_t red[40];
  const int64_t c = blockIdx.x;
  if (c >= n_chunks) return;
  ChunkLocal L;
  chunk_local_carve(lds_raw, DMAX, TBL, L);
  BlockGroup g(red);
  const uint32_t t0 = ibase + (uint32_t)(c * d);
  const ChunkExport ex{rec + (t0 >> 4), mx + t0, mp + t0, ms + t0, t0};
  (void)chunk_front(g, L, fm, lo + c * d, d, t0, flags + t0, ex);
}

// ---- components of more than kTinyMax nodes: records -> flags --------------------------------------------------------------
// records by size class: class lists of record numbers, so that every wavefront / workgroup of the kernels below draws
// components one at a time (the records of the large chunks lie side by side: walked in place, a few hundred workgroups
// got all the large components and the rest none)
constexpr int kCompClasses = 4;
constexpr int kCompListPer = 8;  // records per thread: ONE global atomic per class and 2 048 records (a wavefront's own
                                 // atomics on the four counters -- 5e4 per launch on four addresses -- were 0.5 of the kernel's 0.63 ms)
__global__ void __launch_bounds__(256)
k_comp_lists(const CompRecord *__restrict__ rec, int64_t n_static, const uint32_t *__restrict__ n_dense, uint32_t *__restrict__ lists,
             int64_t list_cap, uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_cnt[kCompClasses], s_base[kCompClasses];
  const int64_t n_rec = n_static + (int64_t)*n_dense;
  const int lane = (int)(threadIdx.x & 63);
  if (threadIdx.x < kCompClasses) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  int cls[kCompListPer];
  uint32_t at[kCompListPer];  // place inside the workgroup's share of its class
#pragma unroll
  for (int k = 0; k < kCompListPer; ++k) {
    const int64_t ri = ((int64_t)blockIdx.x * kCompListPer + k) * 256 + threadIdx.x;
    const uint32_t n = ri < n_rec ? rec[ri].n : 0u;
    cls[k] = n <= (uint32_t)kTinyMax ? -1 : (n <= 76u ? 0 : (n <= 306u ? 1 : (n <= 1228u ? 2 : 3)));
    at[k] = 0u;
  }
#pragma unroll
  for (int k = 0; k < kCompListPer; ++k)
#pragma unroll
    for (int c = 0; c < kCompClasses; ++c) {
      const unsigned long long b = __ballot(cls[k] == c);
      if (!b) continue;  // (uniform)
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_cnt[c], (uint32_t)__popcll(b));
      base = __shfl(base, 0);
      if (cls[k] == c) at[k] = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    }
  __syncthreads();
  if (threadIdx.x < kCompClasses) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x * 32], s_cnt[threadIdx.x]) : 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kCompListPer; ++k)
    if (cls[k] >= 0) {
      const int64_t ri = ((int64_t)blockIdx.x * kCompListPer + k) * 256 + threadIdx.x;
      lists[(int64_t)cls[k] * list_cap + s_base[cls[k]] + at[k]] = (uint32_t)ri;
    }
}

template <int CAP, int TBL, int CAP2>
__global__ void __launch_bounds__(256)
k_comp_wave(const CompRecord *__restrict__ rec, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count,
            const uint32_t *__restrict__ mx, const uint32_t *__restrict__ mp, const uint32_t *__restrict__ ms,
            uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  constexpr size_t per_wave = (comp_local_bytes(CAP, TBL, CAP2) + 15) & ~(size_t)15;
  const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  CompLocal L;
  comp_local_carve(lds_raw + (size_t)wv * per_wave, CAP, TBL, CAP2, L);
  WaveGroup g;
  const uint32_t todo = *count;
  for (uint32_t q = blockIdx.x * 4 + (uint32_t)wv; q < todo; q += gridDim.x * 4) {  // (wave-uniform)
    const CompRecord R = rec[list[q]];
    uint32_t cap2 = 64;  // the parent look-up's hash map: power of two >= 2 n
    while (cap2 < 2u * R.n) cap2 <<= 1;
    const uint32_t first = comp_group_first(g, L, mx + R.moff, mp + R.moff, ms + R.moff, (int)R.n, R.n_graph, cap2);
    for (uint32_t k = (uint32_t)lane; k < R.n; k += 64) {
      const uint32_t x = mx[R.moff + k];
      if (x != first) flags[R.t0 + x] = 1;
    }
    g.sync();
  }
}

template <int CAP, int TBL, int CAP2>
__global__ void __launch_bounds__(256)
k_comp_block(const CompRecord *__restrict__ rec, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count,
             const uint32_t *__restrict__ mx, const uint32_t *__restrict__ mp, const uint32_t *__restrict__ ms,
             uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  __shared__ uint64_t red[40];
  // The largest class (a few dozen components of thousands of nodes, one per workgroup) ends the component phase: its
  // wavefronts share their SIMDs with the thousands of the other classes' kernels; they go first at the issue port.
  if (CAP > 2048) __builtin_amdgcn_s_setprio(3);
  CompLocal L;
  comp_local_carve(lds_raw, CAP, TBL, CAP2, L);
  BlockGroup g(red);
  const uint32_t todo = *count;
  for (uint32_t q = blockIdx.x; q < todo; q += gridDim.x) {  // (uniform)
    const CompRecord R = rec[list[q]];
    uint32_t cap2 = 64;
    while (cap2 < 2u * R.n) cap2 <<= 1;
    const uint32_t first = comp_group_first(g, L, mx + R.moff, mp + R.moff, ms + R.moff, (int)R.n, R.n_graph, cap2);
    for (uint32_t k = threadIdx.x; k < R.n; k += 256) {
      const uint32_t x = mx[R.moff + k];
      if (x != first) flags[R.t0 + x] = 1;
    }
    __syncthreads();
  }
}

// tiny components of the large chunks: a lane each, from the sharded lists k_c_classify filled (members read in place:
// staged through LDS the kernel took twice as long -- a third of the wavefronts per CU)
constexpr int kTinyBlock = 256;
__global__ void __launch_bounds__(kTinyBlock)
k_tiny_lists(const CompRecord *__restrict__ tiny, const uint32_t *__restrict__ tiny_count, int64_t shard_cap,
             const uint32_t *__restrict__ mx, const uint32_t *__restrict__ mp, const uint32_t *__restrict__ ms,
             uint8_t *__restrict__ flags) {
  __shared__ uint8_t scr[kTinyBlock * kTinyScratch];
  const int64_t q = (int64_t)blockIdx.x * kTinyBlock + threadIdx.x;
  const int64_t shard = q / shard_cap, idx = q - shard * shard_cap;
  if (shard >= 8 || idx >= (int64_t)tiny_count[shard * 32]) return;
  const CompRecord R = tiny[shard * shard_cap + idx];
  struct Acc {
    const uint32_t *a, *b, *c;
    __device__ uint32_t x(int k) const { return a[k]; }
    __device__ uint32_t par(int k) const { return b[k]; }
    __device__ uint32_t slot(int k) const { return c[k]; }
  };
  const Acc A{mx + R.moff, mp + R.moff, ms + R.moff};
  const uint32_t first = tiny_first(A, (int)R.n, R.n_graph, scr + (size_t)threadIdx.x * kTinyScratch);
  for (uint32_t k = 0; k < R.n; ++k) {
    const uint32_t x = mx[R.moff + k];
    if (x != first) flags[R.t0 + x] = 1;
  }
}

// ---- large chunks: passes over global arrays, all chunks of all coarse entries at once ---------------------------------------
constexpr unsigned long long kEmpty64 = ~0ull;

struct StageDesc {  // growth stage s of one coarse chunk's tuple set (n_cur == 0: the chunk does not get that far)
  uint32_t n_prev, n_cur, t_prev, mask;
  uint32_t item_end;  // the keys of arrival rank < n_cur are the valid items of [tstart, item_end)
};
struct Coarse {  // device arrays of the coarse item range [c0, c0 + Ic): index t' = item - c0
  uint32_t c0;
  int64_t Ic;
  uint32_t *root, *slot, *szin, *cfill, *valid, *escan, *moffc;
  uint8_t *gnode;
  int64_t *hash;
  // per coarse chunk
  int64_t *m, *ebase, *toff;
  uint32_t *tstart;  // first item (t') of the chunk
  unsigned long long *table;
  uint16_t *gcid;    // per item: its coarse chunk
  const uint16_t *ylist;  // per growth stage: the chunks that can reach it (by their length), concatenated
  uint32_t *ngraph;  // per coarse chunk: nodes of its graph
  StageDesc *desc;   // [stage][chunk]
  int n_cc;
};

__global__ void __launch_bounds__(256)
k_c_valid(const int64_t *__restrict__ fm, LadTab T, Coarse C) {
  const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tp >= C.Ic) return;
  const uint32_t t = C.c0 + (uint32_t)tp;
  const int e = entry_of_block(T, C.c0 + blockIdx.x * 256u);
  const uint32_t q = t - T.ibase[e], d = (uint32_t)T.d[e];
  if (tp == C.Ic - 1) C.valid[C.Ic] = 0, C.szin[C.Ic] = 0;  // (the scans run over Ic + 1 entries)
  C.szin[tp] = 0;
  C.cfill[tp] = 0;
  C.slot[tp] = kNone;
  if ((int64_t)q >= T.nit[e]) {  // padding between entries
    C.valid[tp] = 0;
    C.root[tp] = (uint32_t)tp;
    C.gcid[tp] = (uint16_t)(T.ccbase[e] + (uint32_t)T.nch[e] - 1u);
    return;
  }
  const uint32_t c = q / d, xr = q - c * d;
  C.gcid[tp] = (uint16_t)(T.ccbase[e] + c);
  const int64_t i = T.lo[e] + q, j = fm[i];
  const bool v = j >= 0 && j < T.lo[e] + (int64_t)(c + 1) * d;
  C.valid[tp] = v ? 1u : 0u;
  C.root[tp] = v ? (uint32_t)(tp + (j - i)) : (uint32_t)tp;
  if (v) {
    C.hash[tp] = tuple2_hash((uint64_t)xr, (uint64_t)(xr + (j - i)));
    C.gnode[tp] = 1;
    C.gnode[tp + (j - i)] = 1;
  }
}

__global__ void __launch_bounds__(256)
k_c_sched(Coarse C, int n_cc, int n_stages) {
  const int gc = blockIdx.x * 256 + threadIdx.x;
  if (gc > n_cc) return;
  if (gc == n_cc) {
    C.m[gc] = 0;  // (the scan of the table sizes runs over n_cc + 1 entries)
    return;
  }
  const uint32_t ts = C.tstart[gc], te = C.tstart[gc + 1];  // (te of the last chunk of an entry: the next entry's start, padding is invalid)
  const uint32_t e0 = C.escan[ts];
  const int64_t m = (int64_t)C.escan[te] - (int64_t)e0;
  C.m[gc] = m;
  C.ebase[gc] = e0;
  for (int s = 0; s < n_stages; ++s) {
    const SetStage st = pyset_stage(m, s);
    StageDesc D{0, 0, 0, 0, ts};
    if (st.exists) {
      D.n_prev = (uint32_t)st.n_prev, D.n_cur = (uint32_t)st.n_cur, D.t_prev = st.t_prev, D.mask = st.mask;
      uint32_t lo = ts, hi = te;  // smallest p with escan[p] - e0 >= n_cur (escan is non-decreasing)
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (C.escan[mid] - e0 >= D.n_cur) hi = mid;
        else lo = mid + 1;
      }
      D.item_end = lo;
    }
    C.desc[(size_t)s * n_cc + gc] = D;
  }
}

// table offsets: exclusive scan of the final table sizes (one workgroup; chunks are few)
__global__ void __launch_bounds__(1024)
k_c_toff(Coarse C, int n_cc) {
  __shared__ int64_t part[1025];
  const int per = (n_cc + 1023) / 1024;
  const int b = threadIdx.x * per;
  int64_t s = 0;
  for (int i = b; i < b + per && i < n_cc; ++i) s += C.m[i] > 0 ? (int64_t)pyset_final_mask(C.m[i]) + 1 : 0;
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int i = 0; i < 1024; ++i) {
      const int64_t v = part[i];
      part[i] = run;
      run += v;
    }
    part[1024] = run;
  }
  __syncthreads();
  int64_t run = part[threadIdx.x];
  for (int i = b; i < b + per && i < n_cc; ++i) {
    C.toff[i] = run;
    run += C.m[i] > 0 ? (int64_t)pyset_final_mask(C.m[i]) + 1 : 0;
  }
  if (threadIdx.x == 0) C.toff[n_cc] = part[1024];
}

__global__ void __launch_bounds__(256)
k_c_clear(Coarse C, int y_off, int y_count, int stage) {
  for (int yi = blockIdx.y; yi < y_count; yi += gridDim.y) {
    const int gc = C.ylist[y_off + yi];
    const StageDesc D = C.desc[(size_t)stage * C.n_cc + gc];
    if (D.n_cur == 0) continue;
    unsigned long long *__restrict__ Tb = C.table + C.toff[gc];
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s <= D.mask; s += gridDim.x * 256) Tb[s] = kEmpty64;
  }
}

// the keys of a stage are the first n_cur edges of their chunk: the launch walks each chunk's items up to item_end only
__global__ void __launch_bounds__(256)
k_c_insert(Coarse C, int y_off, int y_count, int stage) {
  for (int yi = blockIdx.y; yi < y_count; yi += gridDim.y) {
    const int gc = C.ylist[y_off + yi];
    const StageDesc D = C.desc[(size_t)stage * C.n_cc + gc];
    if (D.n_cur == 0) continue;
    unsigned long long *__restrict__ Tb = C.table + C.toff[gc];
    const uint32_t e0 = (uint32_t)C.ebase[gc];
    for (uint32_t tp = C.tstart[gc] + blockIdx.x * 256 + threadIdx.x; tp < D.item_end; tp += gridDim.x * 256) {
      if (!C.valid[tp]) continue;
      const uint32_t r = C.escan[tp] - e0;
      unsigned long long me = ((unsigned long long)(r < D.n_prev ? C.slot[tp] : D.t_prev + r) << 32) | (unsigned long long)tp;
      Probe p;
      p.start(C.hash[tp], D.mask);
      for (;;) {
        const uint32_t s = p.slot();
        const unsigned long long old = atomicMin(&Tb[s], me);
        if (old == kEmpty64) break;
        if (old > me) {
          me = old;
          p.start_at(C.hash[(uint32_t)old], D.mask, s);
        }
        p.next();
      }
    }
  }
}

__global__ void __launch_bounds__(256)
k_c_positions(Coarse C, int y_off, int y_count, int stage) {
  for (int yi = blockIdx.y; yi < y_count; yi += gridDim.y) {
    const int gc = C.ylist[y_off + yi];
    const StageDesc D = C.desc[(size_t)stage * C.n_cc + gc];
    if (D.n_cur == 0) continue;
    const unsigned long long *__restrict__ Tb = C.table + C.toff[gc];
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s <= D.mask; s += gridDim.x * 256) {
      const unsigned long long v = Tb[s];
      if (v != kEmpty64) C.slot[(uint32_t)v] = s;
    }
  }
}

// one round of pointer jumping; changed[r] != 0 when round r moved a pointer (a round that finds the one before it idle returns)
__global__ void __launch_bounds__(256)
k_c_jump(Coarse C, uint32_t *__restrict__ changed, int round) {
  if (round > 0 && changed[round - 1] == 0u) return;
  const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tp >= C.Ic) return;
  const uint32_t p = C.root[tp];  // benign race: a value read is always an ancestor
  uint32_t pp = C.root[p];
  if (pp != p) {
    pp = C.root[C.root[pp]];  // (three more hops: a quarter of the launches of one hop at a time)
    C.root[tp] = pp;
    changed[round] = 1u;
  }
}

__global__ void __launch_bounds__(256)
k_c_sizes(Coarse C) {
  // a block's 256 items lie in at most two chunks (chunks here hold more than kChunkMax structures): one atomic per
  // chunk and block on the chunk's counter, not one per node (8e5 of them on ONE address at the coarsest level)
  __shared__ uint32_t cnt[2];
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int gc0 = C.gcid[(int64_t)blockIdx.x * 256];
  if (tp < C.Ic && C.gnode[tp]) {
    atomicAdd(&C.szin[C.root[tp]], 1u);
    atomicAdd(&cnt[C.gcid[tp] != gc0], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(&C.ngraph[gc0 + (int)threadIdx.x], cnt[threadIdx.x]);
}

struct CoarseOut {
  CompRecord *rec_dense;       // records of components of 19 .. 4096 nodes
  uint32_t *n_dense;
  CompRecord *tiny;            // 8 shards of shard_cap records
  uint32_t *tiny_count;        // [8 * 32] one counter per 128-byte line
  int64_t shard_cap;
  CompRecord *host_list;       // components above kGroupCompMax nodes
  uint32_t *n_host, host_cap;
};

__global__ void __launch_bounds__(256)
k_c_classify(Coarse C, CoarseOut O, uint32_t *__restrict__ err) {
  __shared__ uint32_t n_tiny, n_big, base_tiny, base_big;
  if (threadIdx.x == 0) n_tiny = 0, n_big = 0;
  __syncthreads();
  const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t n = 0;
  if (tp < C.Ic && C.root[tp] == (uint32_t)tp) n = C.szin[tp];
  CompRecord R{0, 0, 0, 0};
  int kind = 0;  // 1 tiny, 2 record, 3 host
  uint32_t my = 0;
  if (n >= 2) {
    const int gc = C.gcid[tp];
    R = CompRecord{C.c0 + C.tstart[gc], C.c0 + C.moffc[tp], n, C.ngraph[gc]};
    kind = n <= (uint32_t)kTinyMax ? 1 : (n <= (uint32_t)kGroupCompMax ? 2 : 3);
    if (kind == 1) my = atomicAdd(&n_tiny, 1u);
    else if (kind == 2) my = atomicAdd(&n_big, 1u);
  }
  __syncthreads();
  const int shard = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    base_tiny = n_tiny ? atomicAdd(&O.tiny_count[shard * 32], n_tiny) : 0u;
    base_big = n_big ? atomicAdd(O.n_dense, n_big) : 0u;
  }
  __syncthreads();
  if (kind == 1) {
    if ((int64_t)base_tiny + my < O.shard_cap) O.tiny[(int64_t)shard * O.shard_cap + base_tiny + my] = R;
    else *err = 1u;  // (cannot happen: a shard holds 256 entries per block dealt to it)
  } else if (kind == 2) {
    O.rec_dense[base_big + my] = R;
  } else if (kind == 3) {
    const uint32_t h = atomicAdd(O.n_host, 1u);
    if (h < O.host_cap) O.host_list[h] = R;
  }
}

__global__ void __launch_bounds__(256)
k_c_members(const int64_t *__restrict__ fm, LadTab T, Coarse C, uint32_t *__restrict__ mx, uint32_t *__restrict__ mp,
            uint32_t *__restrict__ ms) {
  const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tp >= C.Ic || !C.gnode[tp]) return;
  const uint32_t r = C.root[tp];
  const uint32_t at = C.c0 + C.moffc[r] + atomicAdd(&C.cfill[r], 1u);
  const uint32_t xr = (uint32_t)tp - C.tstart[C.gcid[tp]];
  const uint32_t sl = C.slot[tp];
  mx[at] = xr;
  ms[at] = sl;
  if (sl != kNone) {
    const int e = entry_of_block(T, C.c0 + blockIdx.x * 256u);
    const int64_t i = T.lo[e] + ((int64_t)(C.c0 + (uint32_t)tp) - T.ibase[e]);
    mp[at] = (uint32_t)(xr + (fm[i] - i));
  } else {
    mp[at] = xr;
  }
}

// ---- the levels in order -------------------------------------------------------------------------------------------------
struct ApplyState {
  uint32_t removed[kMaxEntries];   // structures the entry's flags newly rejected
  uint32_t needhost;               // a last chunk that ends at the active count could hold an edge
  uint32_t err;                    // a device capacity was exceeded (the host ladder takes over)
  uint32_t active_final;           // filled by k_apply_finish
  uint32_t ran[kMaxEntries];       // (for the log)
};

__device__ __forceinline__ int64_t active_before(const LadTab &T, const ApplyState *S, int e, int64_t N, bool *runs) {
  // the active count at the start of entry e's level, and whether that level runs (torsion_module.py:974-976)
  int64_t act = N;
  int cur = -1;
  int64_t act_in = N;
  bool r = false;
  for (int q = 0; q <= e; ++q) {
    if (T.li[q] != cur) {  // a new level: the entries of the level before are all in `act`
      cur = T.li[q];
      act_in = act;
      r = 5 * T.k[q] < act_in;
    }
    if (q < e && r) act -= (int64_t)S->removed[q];
  }
  *runs = r;
  return act_in;
}

__global__ void __launch_bounds__(256)
k_apply(const int64_t *__restrict__ fm, LadTab T, int e, int64_t N, const uint8_t *__restrict__ flags, uint8_t *__restrict__ mask,
        ApplyState *__restrict__ S, int first_li) {
  __shared__ uint32_t gone_block;
  bool runs;
  const int64_t act_in = active_before(T, S, e, N, &runs);
  if (!runs) return;
  if (threadIdx.x == 0) gone_block = 0;
  __syncthreads();
  uint32_t gone = 0;
  for (int u = 0; u < 4; ++u) {
    const int64_t i = ((int64_t)blockIdx.x * 4 + u) * 256 + threadIdx.x;
    if (i >= N) break;
    const int64_t q = i - T.lo[e];
    if (q >= 0 && q < T.nit[e] && flags[T.ibase[e] + q] && mask[i]) {
      mask[i] = 0;
      ++gone;
    }
    // the level's LAST chunk [d (k - 1), active): empty or edge-free in the common case; anything else is the host's
    if (T.li[e] != first_li && T.lo[e] == 0 && i >= T.nit[e] && i < act_in) {
      const int64_t j = fm[i];
      if (j >= 0 && j < act_in) S->needhost = 1u;
    }
  }
  if (gone) atomicAdd(&gone_block, gone);
  __syncthreads();
  if (threadIdx.x == 0 && gone_block) atomicAdd(&S->removed[e], gone_block);
}

__global__ void k_apply_finish(LadTab T, int64_t N, ApplyState *__restrict__ S) {
  int64_t act = N;
  int cur = -1;
  bool r = false;
  for (int q = 0; q < T.n; ++q) {
    if (T.li[q] != cur) {
      cur = T.li[q];
      r = 5 * T.k[q] < act;
    }
    S->ran[q] = r ? 1u : 0u;
    if (r) act -= (int64_t)S->removed[q];
  }
  S->active_final = (uint32_t)act;
}

__global__ void __launch_bounds__(256) k_fill_u8(uint8_t *p, int64_t n, uint8_t v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
static const double kLadderK[] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};
constexpr int kLadderLevels = (int)(sizeof(kLadderK) / sizeof(kLadderK[0]));

struct LadderPlan {
  LadTab T;
  int first_li = -1;
  int64_t I = 0;          // items (padded)
  int64_t c0 = 0, Ic = 0; // coarse item range
  int64_t n_chunks = 0;   // all entries
  int n_cc = 0;           // coarse chunks
  std::vector<uint32_t> tstart;  // coarse chunk starts (t'), n_cc + 1 entries
  std::vector<uint16_t> ylist;   // per growth stage: the coarse chunks long enough to reach it
  std::vector<int> yoff, ycount;
  int64_t table_slots = 0;       // upper bound of the coarse tuple tables
  int max_stages = 0;
  int64_t d_max = 0;
};

static bool build_plan(int64_t N, LadderPlan &P) {
  LadTab &T = P.T;
  T.n = 0;
  P.first_li = -1;
  struct E { int li; int64_t k, lo, d, nch; };
  std::vector<E> es;
  for (int li = 0; li < kLadderLevels; ++li) {
    const int64_t k = (int64_t)kLadderK[li];
    if (k == 1 || !(5 * k < N)) continue;  // (k = 1: its only chunk is a last chunk, the host's)
    const int64_t d = N / k;
    if (d <= 1) continue;
    if (P.first_li < 0) {
      P.first_li = li;
      const int64_t lo = d * (k - 1);
      if (N - lo >= 2) es.push_back(E{li, k, lo, N - lo, 1});  // the first level's last chunk: the active count is N there
    }
    es.push_back(E{li, k, 0, d, k - 1});
  }
  if (es.empty() || (int)es.size() > kMaxEntries) return false;
  // entries stay in ladder order (k_apply walks them so); items: the small-chunk entries first, then the coarse range
  std::vector<int> order_small, order_coarse;
  for (int q = 0; q < (int)es.size(); ++q) (es[(size_t)q].d > kChunkMax ? order_coarse : order_small).push_back(q);
  int64_t at = 0, cb = 0;
  T.n = (int)es.size();
  for (int q = 0; q < T.n; ++q) {
    const E &x = es[(size_t)q];
    T.li[q] = x.li, T.k[q] = x.k, T.lo[q] = x.lo, T.d[q] = x.d, T.nch[q] = x.nch, T.nit[q] = x.d * x.nch;
    T.coarse[q] = x.d > kChunkMax;
    T.ccbase[q] = 0, T.smask[q] = 0;
  }
  for (int q : order_small) {
    T.ibase[q] = (uint32_t)at, T.cbase[q] = (uint32_t)cb;
    at += (T.nit[q] + 255) / 256 * 256;
    cb += T.nch[q];
  }
  P.c0 = at;
  P.tstart.clear();
  int cc = 0;
  P.table_slots = 0;
  P.d_max = 0;
  for (int q : order_coarse) {
    T.ibase[q] = (uint32_t)at, T.cbase[q] = (uint32_t)cb, T.ccbase[q] = (uint32_t)cc;
    for (int64_t c = 0; c < T.nch[q]; ++c) P.tstart.push_back((uint32_t)(at - P.c0 + c * T.d[q]));
    P.table_slots += T.nch[q] * ((int64_t)pyset_final_mask(T.d[q] - 1) + 1);
    P.d_max = std::max(P.d_max, T.d[q]);
    at += (T.nit[q] + 255) / 256 * 256;
    cb += T.nch[q];
    cc += (int)T.nch[q];
  }
  P.I = at;
  P.Ic = at - P.c0;
  P.n_chunks = cb;
  P.n_cc = cc;
  P.tstart.push_back((uint32_t)P.Ic);
  P.max_stages = P.d_max > 1 ? pyset_stage_count(P.d_max - 1) : 0;
  P.ylist.clear(), P.yoff.assign((size_t)P.max_stages + 1, 0), P.ycount.assign((size_t)P.max_stages + 1, 0);
  for (int q : order_coarse) T.smask[q] = 0;
  for (int s = 0; s < P.max_stages; ++s) {
    P.yoff[(size_t)s] = (int)P.ylist.size();
    for (int q : order_coarse)
      if (pyset_stage(T.d[q] - 1, s).exists) {
        T.smask[q] |= 1u << s;
        for (int64_t c = 0; c < T.nch[q]; ++c) P.ylist.push_back((uint16_t)(T.ccbase[q] + c));
      }
    P.ycount[(size_t)s] = (int)P.ylist.size() - P.yoff[(size_t)s];
  }
  // the order of entry_of_block: ibase ascending in ENTRY order is what the kernels assume within each class; the coarse
  // kernels only ever see coarse items, the small-chunk kernels get their entry's numbers as arguments
  return at < ((int64_t)1 << 31) && cc <= 60000;
}

// entry_of_block walks the entries in index order and needs ibase ascending over the entries it can meet: the coarse
// kernels get a table that holds the coarse entries only
static LadTab coarse_table(const LadTab &T) {
  LadTab C = T;
  C.n = 0;
  for (int q = 0; q < T.n; ++q)
    if (T.coarse[q]) {
      const int o = C.n++;
      C.li[o] = T.li[q], C.k[o] = T.k[q], C.lo[o] = T.lo[q], C.d[o] = T.d[q], C.nch[o] = T.nch[q], C.nit[o] = T.nit[q];
      C.ibase[o] = T.ibase[q], C.cbase[o] = T.cbase[q], C.ccbase[o] = T.ccbase[q], C.coarse[o] = 1, C.smask[o] = T.smask[q];
    }
  return C;
}

template <int DMAX, int TBL, int WPB>
static int launch_chunk_front(hipStream_t st, const int64_t *fm, const LadTab &T, int q, uint8_t *flags, CompRecord *rec,
                              uint32_t *mx, uint32_t *mp, uint32_t *ms) {
  constexpr size_t per_wave = (chunk_local_bytes(DMAX, TBL, 64) + 15) & ~(size_t)15;
  static bool attr_set = false;
  if (!attr_set) {
    FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chunk_front<DMAX, TBL, WPB>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per_wave * WPB)));
    attr_set = true;
  }
  const unsigned grid = (unsigned)ceil_div(T.nch[q], WPB);
  hipLaunchKernelGGL((k_chunk_front<DMAX, TBL, WPB>), dim3(grid), dim3(WPB * 64), per_wave * WPB, st, fm, T.lo[q], (int)T.d[q],
                     T.nch[q], T.ibase[q], flags, rec, mx, mp, ms);
  return check_launch("k_chunk_front");
}

template <int DMAX, int TBL>
static int launch_chunk_front_block(hipStream_t st, const int64_t *fm, const LadTab &T, int q, uint8_t *flags, CompRecord *rec,
                                    uint32_t *mx, uint32_t *mp, uint32_t *ms) {
  constexpr size_t lds = (chunk_local_bytes(DMAX, TBL, 256) + 15) & ~(size_t)15;
  static bool attr_set = false;
  if (!attr_set) {
    FC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chunk_front_block<DMAX, TBL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_chunk_front_block<DMAX, TBL>), dim3((unsigned)T.nch[q]), dim3(256), lds, st, fm, T.lo[q], (int)T.d[q],
                     T.nch[q], T.ibase[q], flags, rec, mx, mp, ms);
  return check_launch("k_chunk_front_block");
}

template <int CAP, int TBL, int CAP2, bool BLOCK>
static int launch_comp(hipStream_t st, const CompRecord *rec, const uint32_t *list, const uint32_t *count, int64_t most,
                       const uint32_t *mx, const uint32_t *mp, const uint32_t *ms, uint8_t *flags) {
  const size_t per = (comp_local_bytes(CAP, TBL, CAP2) + 15) & ~(size_t)15;
  const size_t lds = BLOCK ? per : per * 4;
  static bool attr_set = false;
  const void *fn = BLOCK ? reinterpret_cast<const void *>(&k_comp_block<CAP, TBL, CAP2>)
                         : reinterpret_cast<const void *>(&k_comp_wave<CAP, TBL, CAP2>);
  if (!attr_set) {
    FC_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // as many workgroups as fit the chip at once (by their LDS), or fewer when the class cannot have that many components
  const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)(160 * 1024) / (int64_t)(lds + 1024)));
  const int64_t work = BLOCK ? most : ceil_div(most, 4);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(work, (int64_t)ctx().n_cu * per_cu));
  if (BLOCK) hipLaunchKernelGGL((k_comp_block<CAP, TBL, CAP2>), dim3(grid), dim3(256), lds, st, rec, list, count, mx, mp, ms, flags);
  else hipLaunchKernelGGL((k_comp_wave<CAP, TBL, CAP2>), dim3(grid), dim3(256), lds, st, rec, list, count, mx, mp, ms, flags);
  return check_launch("k_comp");
}

static int scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n, DevBuf &tmp) {
  size_t bytes = 0;
  FC_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, st));
  FC_TRY(tmp.reserve(bytes));
  FC_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, in, out, (int)n, st));
  return FC_OK;
}

// The ladder from the device's first-match array.  fm_host: the same array on the host when the caller has it (nullptr:
// what the host needs is copied down: the first active_final entries for the last level's chunk, all of it on the slow paths).
int tfd_ladder_device(const int64_t *fm_dev, const int64_t *fm_host, int64_t N, uint8_t *mask_out) {
  static const bool debug = getenv("FC_DEBUG") != nullptr;
  const auto t_all = std::chrono::steady_clock::now();
  auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count(); };
  LadderPlan P;
  std::vector<int64_t> fm_copy;
  auto need_fm_host = [&](int64_t upto) -> int {  // fm[0, upto) on the host
    if (fm_host) return FC_OK;
    if ((int64_t)fm_copy.size() >= upto) return FC_OK;
    fm_copy.resize((size_t)upto);
    FC_TRY(d2h(fm_copy.data(), fm_dev, (size_t)upto * sizeof(int64_t)));
    return sync();
  };
  auto fmh = [&]() { return fm_host ? fm_host : fm_copy.data(); };
  auto host_ladder = [&]() -> int {
    FC_TRY(need_fm_host(N));
    return tfd_ladder_host_only(fmh(), N, mask_out);
  };
  if (!build_plan(N, P)) return host_ladder();
  const LadTab &T = P.T;
  hipStream_t st = cur_stream();
  FC_TRY(side_streams());
  // side streams: the small-chunk entries (two streams: the workgroup-per-chunk launches are the long ones) beside the coarse
  // passes, later the component kernels of the four size classes and the tiny components of the large chunks side by side
  hipStream_t sA = ctx().s_lane[0], sB = ctx().s_lane[1], sC = ctx().s_lane[2], sD = ctx().s_screen;
  std::vector<hipEvent_t> &evs = ctx().ev_dep_pool;
  while (evs.size() < 10) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    evs.push_back(e);
  }
  struct JoinSide {  // nothing of this call may still run on a side stream when its buffers go back to the pool (error paths too)
    hipStream_t s[4];
    ~JoinSide() {
      for (hipStream_t x : s) (void)hipStreamSynchronize(x);
    }
  } join_side{{sA, sB, sC, sD}};
  auto order = [&](hipStream_t from, hipStream_t to, int ev) -> int {  // `to` continues behind what `from` holds now
    FC_HIP_TRY(hipEventRecord(evs[(size_t)ev], from));
    FC_HIP_TRY(hipStreamWaitEvent(to, evs[(size_t)ev], 0));
    return FC_OK;
  };
  // ---- buffers (one block) ----
  const int64_t I = P.I, Ic = P.Ic;
  const int64_t n_static = I / 16 + 1, dense_cap = Ic / 19 + 2, shard_cap = (ceil_div(std::max<int64_t>(Ic, 1), 256) + 7) / 8 * 256;
  const uint32_t host_cap = 4096;
  ArenaScope arena;
  {
    size_t total = (size_t)I * (1 + 12) + (size_t)(n_static + dense_cap) * (sizeof(CompRecord) + 16) + (size_t)P.n_chunks * 4 +
                   (size_t)Ic * (4 * 7 + 1 + 8) + (size_t)P.table_slots * 8 + (size_t)8 * shard_cap * sizeof(CompRecord) +
                   (size_t)host_cap * sizeof(CompRecord) + (size_t)P.n_cc * 40 + (size_t)N + ((size_t)64 << 20);
    FC_TRY(arena.begin(total));
  }
  DevBuf d_flags, d_rec, d_mx, d_mp, d_ms, d_ngraph, d_mask, d_state, d_small, d_lists;
  FC_TRY(d_flags.reserve((size_t)I));
  FC_TRY(d_rec.reserve((size_t)(n_static + dense_cap) * sizeof(CompRecord)));
  FC_TRY(d_mx.reserve((size_t)I * 4));
  FC_TRY(d_mp.reserve((size_t)I * 4));
  FC_TRY(d_ms.reserve((size_t)I * 4));
  FC_TRY(d_ngraph.reserve((size_t)(P.n_cc + 2) * 4));
  FC_TRY(d_mask.reserve((size_t)N));
  FC_TRY(d_state.reserve(sizeof(ApplyState)));
  FC_TRY(d_lists.reserve((size_t)kCompClasses * (size_t)(n_static + dense_cap) * 4));
  // small words: [0] n_dense, [32 ..] tiny counters (8 x 32), [512] n_host, [544 ..] jump flags (64)
  FC_TRY(d_small.reserve(1024 * 4));
  uint32_t *small = d_small.as<uint32_t>();
  FC_HIP_TRY(hipMemsetAsync(d_flags.p, 0, (size_t)I, st));
  FC_HIP_TRY(hipMemsetAsync(d_rec.p, 0, (size_t)(n_static + dense_cap) * sizeof(CompRecord), st));
  FC_HIP_TRY(hipMemsetAsync(d_ngraph.p, 0, (size_t)(P.n_cc + 2) * 4, st));
  FC_HIP_TRY(hipMemsetAsync(d_state.p, 0, sizeof(ApplyState), st));
  FC_HIP_TRY(hipMemsetAsync(d_small.p, 0, 1024 * 4, st));
  hipLaunchKernelGGL(k_fill_u8, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, d_mask.as<uint8_t>(), N, (uint8_t)1);
  FC_TRY(check_launch("k_fill_u8"));
  uint8_t *flags = d_flags.as<uint8_t>();
  CompRecord *rec = d_rec.as<CompRecord>();
  uint32_t *mx = d_mx.as<uint32_t>(), *mp = d_mp.as<uint32_t>(), *ms = d_ms.as<uint32_t>();
  // ---- the coarse range on the main stream ----
  DevBuf c_root, c_slot, c_szin, c_cfill, c_valid, c_escan, c_moffc, c_gnode, c_hash, c_m, c_ebase, c_toff, c_tstart, c_table, c_tiny,
      c_host, c_scan_tmp, c_gcid, c_ylist, c_desc;
  Coarse C{};
  CoarseOut O{};
  const LadTab TC = coarse_table(T);
  if (Ic > 0) {
    FC_TRY(c_root.reserve((size_t)Ic * 4));
    FC_TRY(c_slot.reserve((size_t)Ic * 4));
    FC_TRY(c_szin.reserve((size_t)(Ic + 1) * 4));
    FC_TRY(c_cfill.reserve((size_t)Ic * 4));
    FC_TRY(c_valid.reserve((size_t)(Ic + 1) * 4));
    FC_TRY(c_escan.reserve((size_t)(Ic + 1) * 4));
    FC_TRY(c_moffc.reserve((size_t)(Ic + 1) * 4));
    FC_TRY(c_gnode.reserve((size_t)Ic));
    FC_TRY(c_hash.reserve((size_t)Ic * 8));
    FC_TRY(c_m.reserve((size_t)(P.n_cc + 1) * 8));
    FC_TRY(c_ebase.reserve((size_t)(P.n_cc + 1) * 8));
    FC_TRY(c_toff.reserve((size_t)(P.n_cc + 1) * 8));
    FC_TRY(c_tstart.reserve((size_t)(P.n_cc + 1) * 4));
    FC_TRY(c_gcid.reserve((size_t)Ic * 2));
    FC_TRY(c_desc.reserve((size_t)std::max(P.max_stages, 1) * (size_t)(P.n_cc + 1) * sizeof(StageDesc)));
    FC_TRY(c_ylist.reserve(std::max<size_t>(P.ylist.size(), 1) * 2));
    FC_TRY(c_table.reserve((size_t)std::max<int64_t>(P.table_slots, 8) * 8));
    FC_TRY(c_tiny.reserve((size_t)8 * shard_cap * sizeof(CompRecord)));
    FC_TRY(c_host.reserve((size_t)host_cap * sizeof(CompRecord)));
    C.c0 = (uint32_t)P.c0, C.Ic = Ic;
    C.root = c_root.as<uint32_t>(), C.slot = c_slot.as<uint32_t>(), C.szin = c_szin.as<uint32_t>(), C.cfill = c_cfill.as<uint32_t>();
    C.valid = c_valid.as<uint32_t>(), C.escan = c_escan.as<uint32_t>(), C.moffc = c_moffc.as<uint32_t>(), C.gnode = c_gnode.as<uint8_t>();
    C.hash = c_hash.as<int64_t>(), C.m = c_m.as<int64_t>(), C.ebase = c_ebase.as<int64_t>(), C.toff = c_toff.as<int64_t>();
    C.tstart = c_tstart.as<uint32_t>(), C.table = c_table.as<unsigned long long>();
    C.gcid = c_gcid.as<uint16_t>(), C.ylist = c_ylist.as<uint16_t>(), C.ngraph = d_ngraph.as<uint32_t>();
    C.desc = c_desc.as<StageDesc>(), C.n_cc = P.n_cc;
    O.rec_dense = rec + n_static, O.n_dense = small, O.tiny = c_tiny.as<CompRecord>(), O.tiny_count = small + 32, O.shard_cap = shard_cap;
    O.host_list = c_host.as<CompRecord>(), O.n_host = small + 512, O.host_cap = host_cap;
    FC_TRY(pinned_reserve((size_t)(P.n_cc + 1) * 4 + P.ylist.size() * 2 + 4096));
  }
  FC_TRY(order(st, sA, 0));
  FC_TRY(order(st, sB, 1));
  for (int q = 0; q < T.n; ++q) {
    if (T.coarse[q]) continue;
    const int64_t d = T.d[q];
    if (d <= 19) FC_TRY((launch_chunk_front<19, 32, 4>(sB, fm_dev, T, q, flags, rec, mx, mp, ms)));
    else if (d <= 77) FC_TRY((launch_chunk_front<77, 128, 4>(sB, fm_dev, T, q, flags, rec, mx, mp, ms)));
    else if (d <= 307) FC_TRY((launch_chunk_front<307, 512, 4>(sB, fm_dev, T, q, flags, rec, mx, mp, ms)));
    else if (d <= 1229) FC_TRY((launch_chunk_front<1229, 2048, 2>(sB, fm_dev, T, q, flags, rec, mx, mp, ms)));
    else FC_TRY((launch_chunk_front_block<4915, 8192>(sA, fm_dev, T, q, flags, rec, mx, mp, ms)));
  }
  if (Ic > 0) {
    const dim3 block(256), igrid((unsigned)ceil_div(Ic, 256));
    std::memcpy(ctx().pinned, P.tstart.data(), (size_t)(P.n_cc + 1) * 4);
    FC_HIP_TRY(hipMemcpyAsync(C.tstart, ctx().pinned, (size_t)(P.n_cc + 1) * 4, hipMemcpyHostToDevice, st));
    if (!P.ylist.empty()) {
      char *yp = static_cast<char *>(ctx().pinned) + (size_t)(P.n_cc + 1) * 4;
      std::memcpy(yp, P.ylist.data(), P.ylist.size() * 2);
      FC_HIP_TRY(hipMemcpyAsync(c_ylist.p, yp, P.ylist.size() * 2, hipMemcpyHostToDevice, st));
    }
    FC_HIP_TRY(hipMemsetAsync(C.gnode, 0, (size_t)Ic, st));
    hipLaunchKernelGGL(k_c_valid, igrid, block, 0, st, fm_dev, TC, C);
    FC_TRY(check_launch("k_c_valid"));
    FC_TRY(scan_u32(st, C.valid, C.escan, Ic + 1, c_scan_tmp));
    hipLaunchKernelGGL(k_c_sched, dim3((unsigned)ceil_div((int64_t)P.n_cc + 1, 256)), block, 0, st, C, P.n_cc, P.max_stages);
    FC_TRY(check_launch("k_c_sched"));
    hipLaunchKernelGGL(k_c_toff, dim3(1), dim3(1024), 0, st, C, P.n_cc);
    FC_TRY(check_launch("k_c_toff"));
    for (int s = 0; s < P.max_stages; ++s) {
      // the chunks long enough to reach stage s, and the largest table one of them can have in it
      const SetStage top = pyset_stage(P.d_max - 1, s);
      const int yo = P.yoff[(size_t)s], yc = P.ycount[(size_t)s];
      if (yc == 0) continue;
      const unsigned gx = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, ceil_div((int64_t)top.mask + 1, 1024)));
      const unsigned gy = (unsigned)std::min<int>(yc, 60000);
      hipLaunchKernelGGL(k_c_clear, dim3(gx, gy), block, 0, st, C, yo, yc, s);
      FC_TRY(check_launch("k_c_clear"));
      // (a chunk's keys of this stage: about n_cur / density items from its start; the kernel strides over whatever it is)
      const unsigned gi = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(top.n_cur * 2, 1024)));
      hipLaunchKernelGGL(k_c_insert, dim3(gi, gy), block, 0, st, C, yo, yc, s);
      FC_TRY(check_launch("k_c_insert"));
      hipLaunchKernelGGL(k_c_positions, dim3(gx, gy), block, 0, st, C, yo, yc, s);
      FC_TRY(check_launch("k_c_positions"));
    }
    {
      int rounds = 1;  // a pointer covers 4^r edges after r rounds
      while (((int64_t)1 << (2 * rounds)) < P.d_max) ++rounds;
      for (int r = 0; r <= rounds + 1 && r < 60; ++r) {
        hipLaunchKernelGGL(k_c_jump, igrid, block, 0, st, C, small + 544, r);
        FC_TRY(check_launch("k_c_jump"));
      }
    }
    hipLaunchKernelGGL(k_c_sizes, igrid, block, 0, st, C);
    FC_TRY(check_launch("k_c_sizes"));
    FC_TRY(scan_u32(st, C.szin, C.moffc, Ic + 1, c_scan_tmp));
    hipLaunchKernelGGL(k_c_classify, igrid, block, 0, st, C, O, &d_state.as<ApplyState>()->err);
    FC_TRY(check_launch("k_c_classify"));
    hipLaunchKernelGGL(k_c_members, igrid, block, 0, st, fm_dev, TC, C, mx, mp, ms);
    FC_TRY(check_launch("k_c_members"));
    FC_TRY(order(st, sD, 2));
    hipLaunchKernelGGL(k_tiny_lists, dim3((unsigned)ceil_div(8 * shard_cap, kTinyBlock)), dim3(kTinyBlock), 0, sD, O.tiny, O.tiny_count, shard_cap, mx, mp, ms, flags);
    FC_TRY(check_launch("k_tiny_lists"));
  }
  // ---- components of 19 .. 4096 nodes from both paths: the four classes side by side, the largest (longest) first ----
  FC_TRY(order(sA, st, 3));
  FC_TRY(order(sB, st, 4));
  {
    const int64_t n_rec_cap = n_static + dense_cap;
    uint32_t *lists = d_lists.as<uint32_t>(), *lcount = small + 640;  // (one counter per 128-byte line)
    hipLaunchKernelGGL(k_comp_lists, dim3((unsigned)ceil_div(n_rec_cap, 256 * kCompListPer)), dim3(256), 0, st, rec, n_static, small, lists, n_rec_cap, lcount);
    FC_TRY(check_launch("k_comp_lists"));
    FC_HIP_TRY(hipEventRecord(evs[5], st));
    for (hipStream_t s2 : {sA, sB, sC}) FC_HIP_TRY(hipStreamWaitEvent(s2, evs[5], 0));
    FC_TRY((launch_comp<4096, 8192, 8192, true>(sA, rec, lists + 3 * n_rec_cap, lcount + 96, I / 1229 + 1, mx, mp, ms, flags)));
    FC_TRY((launch_comp<1228, 2048, 4096, true>(sB, rec, lists + 2 * n_rec_cap, lcount + 64, I / 307 + 1, mx, mp, ms, flags)));
    FC_TRY((launch_comp<308, 512, 1024, false>(sC, rec, lists + n_rec_cap, lcount + 32, I / 77 + 1, mx, mp, ms, flags)));
    FC_TRY((launch_comp<80, 128, 256, false>(st, rec, lists, lcount, I / 19 + 1, mx, mp, ms, flags)));
    FC_TRY(order(sA, st, 6));
    FC_TRY(order(sB, st, 7));
    FC_TRY(order(sC, st, 8));
    if (Ic > 0) FC_TRY(order(sD, st, 9));
  }
  // ---- the levels in order ----
  ApplyState *S = d_state.as<ApplyState>();
  for (int q = 0; q < T.n; ++q) {
    hipLaunchKernelGGL(k_apply, dim3((unsigned)ceil_div(N, 1024)), dim3(256), 0, st, fm_dev, T, q, N, flags, d_mask.as<uint8_t>(), S, P.first_li);
    FC_TRY(check_launch("k_apply"));
  }
  hipLaunchKernelGGL(k_apply_finish, dim3(1), dim3(1), 0, st, T, N, S);
  FC_TRY(check_launch("k_apply_finish"));
  ApplyState hs;
  uint32_t n_host = 0;
  FC_TRY(pinned_reserve(sizeof(ApplyState) + 64));
  FC_HIP_TRY(hipMemcpyAsync(ctx().pinned, S, sizeof(ApplyState), hipMemcpyDeviceToHost, st));
  FC_HIP_TRY(hipMemcpyAsync(static_cast<char *>(ctx().pinned) + sizeof(ApplyState), small + 512, 4, hipMemcpyDeviceToHost, st));
  FC_TRY(d2h(mask_out, d_mask.p, (size_t)N));
  FC_TRY(sync());
  std::memcpy(&hs, ctx().pinned, sizeof hs);
  std::memcpy(&n_host, static_cast<char *>(ctx().pinned) + sizeof(ApplyState), 4);
  if (debug)
    fprintf(stderr, "[fc] tfd ladder (device): %d entries, %lld items (%lld in %d coarse chunks), %d set stages; active after the levels %u, "
                    "needhost %u, components for the host %u, err %u; %.2f ms\n",
            T.n, (long long)I, (long long)Ic, P.n_cc, P.max_stages, hs.active_final, hs.needhost, n_host, hs.err, ms_since());
  if (hs.err || n_host > host_cap) return host_ladder();
  if (hs.needhost || n_host > 0) {
    // the device's flags, the components it left (above kGroupCompMax nodes) done here, the levels applied on the host
    FC_TRY(need_fm_host(N));
    std::vector<uint8_t> hflags((size_t)I);
    FC_TRY(d2h(hflags.data(), d_flags.p, (size_t)I));
    std::vector<CompRecord> hl(n_host);
    if (n_host) FC_TRY(d2h(hl.data(), c_host.p, (size_t)n_host * sizeof(CompRecord)));
    FC_TRY(sync());
    std::vector<uint32_t> hx, hp, hsl;
    for (const CompRecord &R : hl) {
      hx.resize(R.n), hp.resize(R.n), hsl.resize(R.n);
      FC_TRY(d2h(hx.data(), mx + R.moff, (size_t)R.n * 4));
      FC_TRY(d2h(hp.data(), mp + R.moff, (size_t)R.n * 4));
      FC_TRY(d2h(hsl.data(), ms + R.moff, (size_t)R.n * 4));
      FC_TRY(sync());
      const uint32_t first = host_component_first_big(hx.data(), hp.data(), hsl.data(), R.n, R.n_graph);
      for (uint32_t k = 0; k < R.n; ++k)
        if (hx[k] != first) hflags[(size_t)R.t0 + hx[k]] = 1;
    }
    std::vector<const uint8_t *> lf((size_t)kLadderLevels, nullptr);
    const uint8_t *first_last = nullptr;
    for (int q = 0; q < T.n; ++q) {
      if (T.lo[q] == 0) lf[(size_t)T.li[q]] = hflags.data() + T.ibase[q];
      else first_last = hflags.data() + T.ibase[q];
    }
    return tfd_apply_levels_host(fmh(), N, lf, P.first_li, first_last, mask_out, -1);
  }
  // the last level (k = 1): its only chunk is [0, active), the host's
  const int64_t act = hs.active_final;
  if (act >= 2) {
    FC_TRY(need_fm_host(act));
    std::vector<const uint8_t *> none;
    FC_TRY(tfd_apply_levels_host(fmh(), N, none, -2, nullptr, mask_out, act));  // (-2: only the level k = 1, on the mask as it stands)
  }
  if (debug) fprintf(stderr, "[fc] tfd ladder (device) total %.2f ms\n", ms_since());
  return FC_OK;
}

// test hook: iteration order (indices into `pairs`) of a Python set of n DISTINCT 2-tuples inserted in order, computed
// by the coarse path's staged insertion (one chunk of n items) -- must equal pyset_order_pairs (fc_tfd_host.cpp)
namespace {
__global__ void __launch_bounds__(256)
k_pair_hashes(const int64_t *__restrict__ pairs, int64_t n, Coarse C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    C.valid[n] = 0;
    return;
  }
  C.hash[i] = tuple2_hash((uint64_t)pairs[i * 2], (uint64_t)pairs[i * 2 + 1]);
  C.valid[i] = 1;
  C.slot[i] = kNone;
}
}  // namespace
int pyset_order_pairs_device(const int64_t *pairs_host, int64_t n, int64_t *order_out) {
  if (n == 0) return FC_OK;
  if (n >= (1ll << 30)) return set_error(FC_E_LIMIT, "too many pairs");
  hipStream_t st = cur_stream();
  DevBuf dp, c_slot, c_valid, c_escan, c_hash, c_m, c_ebase, c_toff, c_tstart, c_table, c_gcid, c_ylist, c_desc, tmp;
  FC_TRY(dp.reserve((size_t)n * 2 * sizeof(int64_t)));
  FC_TRY(h2d(dp.p, pairs_host, (size_t)n * 2 * sizeof(int64_t)));
  FC_TRY(c_slot.reserve((size_t)n * 4));
  FC_TRY(c_valid.reserve((size_t)(n + 1) * 4));
  FC_TRY(c_escan.reserve((size_t)(n + 1) * 4));
  FC_TRY(c_hash.reserve((size_t)n * 8));
  FC_TRY(c_m.reserve(2 * 8));
  FC_TRY(c_ebase.reserve(2 * 8));
  FC_TRY(c_toff.reserve(2 * 8));
  FC_TRY(c_tstart.reserve(2 * 4));
  FC_TRY(c_gcid.reserve((size_t)(n + 256) * 2));
  FC_TRY(c_ylist.reserve(64));
  FC_HIP_TRY(hipMemsetAsync(c_gcid.p, 0, (size_t)(n + 256) * 2, st));
  FC_HIP_TRY(hipMemsetAsync(c_ylist.p, 0, 64, st));
  FC_TRY(c_table.reserve(((size_t)pyset_final_mask(n) + 1) * 8));
  Coarse C{};
  C.c0 = 0, C.Ic = n;
  C.slot = c_slot.as<uint32_t>(), C.valid = c_valid.as<uint32_t>(), C.escan = c_escan.as<uint32_t>(), C.hash = c_hash.as<int64_t>();
  C.m = c_m.as<int64_t>(), C.ebase = c_ebase.as<int64_t>(), C.toff = c_toff.as<int64_t>(), C.tstart = c_tstart.as<uint32_t>();
  C.table = c_table.as<unsigned long long>();
  C.gcid = c_gcid.as<uint16_t>(), C.ylist = c_ylist.as<uint16_t>();
  const int stages = pyset_stage_count(n);
  FC_TRY(c_desc.reserve((size_t)(stages + 1) * 2 * sizeof(StageDesc)));
  C.desc = c_desc.as<StageDesc>(), C.n_cc = 1;
  const uint32_t ts[2] = {0u, (uint32_t)n};
  FC_TRY(h2d(C.tstart, ts, sizeof ts));
  LadTab T{};
  T.n = 1, T.d[0] = n, T.nch[0] = 1, T.nit[0] = n, T.ibase[0] = 0, T.ccbase[0] = 0, T.coarse[0] = 1, T.smask[0] = 0xFFFFFFFFu;
  const dim3 block(256), igrid((unsigned)ceil_div(n + 1, 256));
  hipLaunchKernelGGL(k_pair_hashes, igrid, block, 0, st, dp.as<int64_t>(), n, C);
  FC_TRY(check_launch("k_pair_hashes"));
  FC_TRY(scan_u32(st, C.valid, C.escan, n + 1, tmp));
  hipLaunchKernelGGL(k_c_sched, dim3(1), block, 0, st, C, 1, stages);
  FC_TRY(check_launch("k_c_sched"));
  hipLaunchKernelGGL(k_c_toff, dim3(1), dim3(1024), 0, st, C, 1);
  FC_TRY(check_launch("k_c_toff"));
  for (int s = 0; s < stages; ++s) {
    const SetStage top = pyset_stage(n, s);
    const unsigned gx = (unsigned)std::min<int64_t>(512, std::max<int64_t>(1, ceil_div((int64_t)top.mask + 1, 256)));
    hipLaunchKernelGGL(k_c_clear, dim3(gx, 1), block, 0, st, C, 0, 1, s);
    FC_TRY(check_launch("k_c_clear"));
    hipLaunchKernelGGL(k_c_insert, dim3((unsigned)std::min<int64_t>(256, ceil_div(n, 256)), 1), block, 0, st, C, 0, 1, s);
    FC_TRY(check_launch("k_c_insert"));
    hipLaunchKernelGGL(k_c_positions, dim3(gx, 1), block, 0, st, C, 0, 1, s);
    FC_TRY(check_launch("k_c_positions"));
  }
  std::vector<uint32_t> slot((size_t)n);
  FC_TRY(d2h(slot.data(), C.slot, (size_t)n * 4));
  FC_TRY(sync());
  std::vector<int64_t> ord((size_t)n);
  for (int64_t k = 0; k < n; ++k) ord[(size_t)k] = k;
  std::sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return slot[(size_t)a] < slot[(size_t)b]; });
  for (int64_t k = 0; k < n; ++k) order_out[k] = ord[(size_t)k];
  return FC_OK;
}

#if defined(FC_TFD_STAMPS)
extern "C" int fc_debug_cf_stamps(unsigned long long *out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(tfd::g_cf_stamps), 48 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(tfd::g_cf_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int fc_debug_tfd_stamps(unsigned long long *out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(tfd::g_tfd_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(tfd::g_tfd_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif

__global__ void k_warm_tfd_ladder() {}
int warm_tfd_ladder() {
  hipLaunchKernelGGL(k_warm_tfd_ladder, dim3(1), dim3(64), 0, cur_stream());
  return check_launch("k_warm_tfd_ladder");
}

}  // namespace fc
