// fc_tfd_gpu.hip -- the chunk graphs of prune_conformers_tfd's COARSE ladder levels, built on the device
// (firecode/torsion_module.py:985-1030; round 3).
//
// The host ladder (fc_tfd_host.cpp) returns the reference's mask bit for bit by re-enacting, per chunk, what
// CPython and networkx do to the chunk's first-match pairs: (1) `Graph(matches)` sees the edges in the
// iteration order of a Python set of 2-tuples, (2) nodes and neighbour lists come into being in that order,
// (3) every component keeps tuple(g.subgraph(c).nodes)[0].  For the few huge chunks of the levels k <= 20
// (8.4e4 .. 8.4e5 structures each at 1.7 M) step (1) is a walk through hash tables of 2^18 .. 2^21 slots at
// random and step (2) a scatter over 10^6 nodes: 85 ms on one host core for the single chunk of k = 2, the
// critical path of the whole csearch pipeline.  Both steps are data parallel once one sees that
//
//   * an open-addressing table filled by first-fit in a FIXED insertion order is the unique fixed point of
//     "every key sits in the first slot of its probe sequence that holds no key of higher priority": keys can be
//     inserted in ANY order if a key that meets a slot held by a lower-priority key takes it and carries the
//     evicted key on along that key's own probe sequence (one 64-bit atomicMin per probe on (priority, key));
//   * CPython's growth steps depend on the COUNT of keys only (resize when fill * 5 >= mask * 3, to the first
//     power of two above 4 x used, 2 x above 50 000): table s is table s-1 re-inserted in slot order followed by
//     the next keys in arrival order -- priority = old slot, then table size + arrival rank.  Eleven stages for
//     8e5 keys, each one launch over all chunks of a level;
//   * node numbers are ranks of first appearance along the edge order (atomicMin + prefix sum), neighbour
//     lists are a segmented sort of (node, time) keys.
//
// What stays on the host: the per-component part (breadth-first order, the two small Python sets, group[0]),
// which touches only a component's own nodes and is dealt to the host threads -- now from arrays this file
// delivers.  The probe sequence, growth rule and tuple hash are those of fc_tfd_host.cpp (CPython 3.8+
// setobject.c / tupleobject.c); tests/test_pyset_emulation.py checks the host forms against the running
// interpreter and tests/test_tfd_gpu_graph.py this file against the host forms.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "fc_common.h"

// In this file a function that fails half-way drains its stream before it returns: its buffers go back to the pool on the
// way out, and their next user may be a job on ANOTHER stream (the ladder's levels run side by side), which stream
// order would not protect.  On success every function here ends with its own synchronisation.
#undef FC_TRY
#define FC_TRY(expr)                                  \
  do {                                                \
    int _rc = (expr);                                 \
    if (_rc != FC_OK) {                               \
      (void)hipStreamSynchronize(::fc::cur_stream()); \
      return _rc;                                     \
    }                                                 \
  } while (0)
#undef FC_HIP_TRY
#define FC_HIP_TRY(expr)                                                                                          \
  do {                                                                                                            \
    hipError_t _e = (expr);                                                                                       \
    if (_e != hipSuccess) {                                                                                       \
      (void)hipStreamSynchronize(::fc::cur_stream());                                                             \
      return ::fc::set_error(_e == hipErrorOutOfMemory ? FC_E_NOMEM : FC_E_HIP, "%s failed: %s (%s:%d)", #expr,   \
                             hipGetErrorString(_e), __FILE__, __LINE__);                                          \
    }                                                                                                             \
  } while (0)

namespace fc {

int64_t py_tuple2_hash(int64_t a, int64_t b);  // fc_tfd_host.cpp

namespace {

constexpr unsigned long long kEmptySlot = ~0ull;

// one growth stage of one chunk's set: keys of arrival rank < n_cur live in a table of mask + 1 slots that was
// built from the previous table (t_prev slots, keys of rank < n_prev, in slot order) and the later arrivals
struct SetStage {
  int64_t n_prev, n_cur;
  uint32_t t_prev, mask;
};

// the stages a CPython set goes through while it receives m distinct keys one by one
void pyset_schedule(int64_t m, std::vector<SetStage> &st) {
  st.clear();
  if (m <= 0) return;
  uint64_t mask = 7, t_prev = 0;
  int64_t n_prev = 0;
  for (;;) {
    const int64_t trigger = (int64_t)((mask * 3 + 4) / 5);  // smallest fill with fill * 5 >= mask * 3
    if (trigger > m) {
      st.push_back({n_prev, m, (uint32_t)t_prev, (uint32_t)mask});
      return;
    }
    st.push_back({n_prev, trigger, (uint32_t)t_prev, (uint32_t)mask});
    n_prev = trigger;
    t_prev = mask + 1;
    const uint64_t minused = trigger > 50000 ? 2 * (uint64_t)trigger : 4 * (uint64_t)trigger;
    uint64_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
    if (n_prev == m) {  // the last key triggered a growth: the final table is a pure rebuild
      st.push_back({n_prev, m, (uint32_t)t_prev, (uint32_t)mask});
      return;
    }
  }
}

struct ChunkStage {  // device copy: what stage s means for chunk c (n_cur == 0: the chunk has no stage s)
  int64_t n_prev, n_cur;
  uint32_t t_prev, mask;
};

// CPython's probe sequence (setobject.c set_add_entry): slot i, the next LINEAR_PROBES = 9 slots when they fit
// below the table's end, then i = (i * 5 + 1 + (perturb >>= 5)) & mask
struct Probe {
  uint64_t perturb, i;
  uint32_t mask;
  int j, lim;
  __device__ void start(int64_t hash, uint32_t mask_) {
    mask = mask_;
    perturb = (uint64_t)hash;
    i = (uint64_t)hash & mask;
    j = 0;
    lim = (i + 9 <= mask) ? 9 : 0;
  }
  __device__ uint32_t slot() const { return (uint32_t)(i + (uint64_t)j); }
  __device__ void next() {
    if (j < lim) {
      ++j;
      return;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
    j = 0;
    lim = (i + 9 <= mask) ? 9 : 0;
  }
};

// items [0, n_items): valid[i] != 0 marks a key; its chunk is i / d, its arrival rank rank[i], its hash hash[i].
// Keys of rank < n_cur of every chunk that has stage `st` go into table + toff[chunk] (all slots kEmptySlot on
// entry) by priority first-fit; pos[i] = the key's slot in the previous table.
__global__ void __launch_bounds__(256)
k_set_insert(const int32_t *__restrict__ valid, const int32_t *__restrict__ rank, const int64_t *__restrict__ hash,
             const uint32_t *__restrict__ pos, int64_t n_items, int64_t d, const ChunkStage *__restrict__ st,
             const int64_t *__restrict__ toff, unsigned long long *__restrict__ table) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= n_items || !valid[i0]) return;
  const int64_t c = i0 / d;
  const ChunkStage S = st[c];
  const int64_t r = rank[i0];
  if (r >= S.n_cur) return;
  unsigned long long me = ((unsigned long long)(r < S.n_prev ? (uint64_t)pos[i0] : (uint64_t)S.t_prev + (uint64_t)r) << 32) |
                          (unsigned long long)(uint32_t)i0;
  unsigned long long *__restrict__ T = table + toff[c];
  Probe p;
  p.start(hash[i0], S.mask);
  for (;;) {
    const uint32_t s = p.slot();
    const unsigned long long old = atomicMin(&T[s], me);
    if (old == kEmptySlot) return;
    if (old > me) {  // this key outranks the slot's holder: the holder moves on along ITS sequence, behind slot s
      me = old;
      p.start(hash[(uint32_t)old], S.mask);
      while (p.slot() != s) p.next();
    }
    p.next();
  }
}

__global__ void __launch_bounds__(256)
k_set_clear(unsigned long long *__restrict__ table, const int64_t *__restrict__ toff, const ChunkStage *__restrict__ st, int n_chunks) {
  const int c = blockIdx.y;
  if (c >= n_chunks || st[c].n_cur == 0) return;
  const uint32_t size = st[c].mask + 1;
  unsigned long long *__restrict__ T = table + toff[c];
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x) T[s] = kEmptySlot;
}

// pos[key] = slot of every key in the table stage `st` has just built
__global__ void __launch_bounds__(256)
k_set_positions(const unsigned long long *__restrict__ table, const int64_t *__restrict__ toff, const ChunkStage *__restrict__ st,
                int n_chunks, uint32_t *__restrict__ pos) {
  const int c = blockIdx.y;
  if (c >= n_chunks || st[c].n_cur == 0) return;
  const uint32_t size = st[c].mask + 1;
  const unsigned long long *__restrict__ T = table + toff[c];
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x) {
    const unsigned long long v = T[s];
    if (v != kEmptySlot) pos[(uint32_t)v] = s;
  }
}

// flags of the occupied slots of every chunk's FINAL table (buf[fin_buf[c]]), concatenated at toff
__global__ void __launch_bounds__(256)
k_set_occupancy(const unsigned long long *__restrict__ buf0, const unsigned long long *__restrict__ buf1,
                const int64_t *__restrict__ toff, const uint32_t *__restrict__ fin_size, const int32_t *__restrict__ fin_buf,
                int n_chunks, int32_t *__restrict__ occ) {
  const int c = blockIdx.y;
  if (c >= n_chunks) return;
  const uint32_t size = fin_size[c];
  const unsigned long long *__restrict__ T = (fin_buf[c] ? buf1 : buf0) + toff[c];
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x)
    occ[toff[c] + s] = T[s] != kEmptySlot;
}

// qpos[key] = place of the key in its chunk's iteration order; order[ebase[c] + q] = key
__global__ void __launch_bounds__(256)
k_set_order(const unsigned long long *__restrict__ buf0, const unsigned long long *__restrict__ buf1,
            const int64_t *__restrict__ toff, const uint32_t *__restrict__ fin_size, const int32_t *__restrict__ fin_buf,
            int n_chunks, const int32_t *__restrict__ occ_scan, const int64_t *__restrict__ ebase, int32_t *__restrict__ qpos,
            int32_t *__restrict__ order) {
  const int c = blockIdx.y;
  if (c >= n_chunks) return;
  const uint32_t size = fin_size[c];
  const unsigned long long *__restrict__ T = (fin_buf[c] ? buf1 : buf0) + toff[c];
  const int32_t q0 = occ_scan[toff[c]];
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x) {
    const unsigned long long v = T[s];
    if (v == kEmptySlot) continue;
    const int32_t q = occ_scan[toff[c] + s] - q0;
    qpos[(uint32_t)v] = q;
    if (order) order[ebase[c] + q] = (int32_t)(uint32_t)v;
  }
}

struct Scratch {  // temporary storage of the scans / sorts (grow-only)
  DevBuf tmp;
  std::vector<std::unique_ptr<DevBuf>> retired;  // outgrown blocks: kernels in flight may still use them, so they go back
                                                 // to the pool with the Scratch (behind its owner's synchronisation)
  int reserve(size_t n) {
    if (n <= tmp.bytes && tmp.p) return FC_OK;
    if (tmp.p) {
      retired.emplace_back(new DevBuf);
      retired.back()->p = tmp.p, retired.back()->bytes = tmp.bytes, retired.back()->epoch = tmp.epoch, retired.back()->owned = tmp.owned;
      tmp.p = nullptr, tmp.bytes = 0;
    }
    return tmp.reserve(n);
  }
};

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, Scratch &scr) {
  size_t bytes = 0;
  FC_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n, cur_stream()));
  FC_TRY(scr.reserve(bytes));
  FC_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(scr.tmp.p, bytes, in, out, (int)n, cur_stream()));
  return FC_OK;
}

// The iteration order of the Python sets of `n_chunks` chunks of d items each (items [0, n_items), see
// k_set_insert).  m[c] = keys of chunk c (host), ebase[c] = keys before chunk c (device and host).
// Out: qpos[i] (device, n_items int32) and, when order_dev != nullptr, order[ebase[c] + q] = item.
int pyset_orders_device(const int32_t *valid, const int32_t *rank, const int64_t *hash, int64_t n_items, int64_t d,
                        int n_chunks, const std::vector<int64_t> &m, const int64_t *ebase_dev, int32_t *qpos_dev,
                        int32_t *order_dev, Scratch &scr) {
  std::vector<std::vector<SetStage>> sched((size_t)n_chunks);
  size_t n_stages = 0;
  std::vector<int64_t> toff((size_t)n_chunks + 1, 0);
  for (int c = 0; c < n_chunks; ++c) {
    pyset_schedule(m[(size_t)c], sched[(size_t)c]);
    n_stages = std::max(n_stages, sched[(size_t)c].size());
    const int64_t fin = sched[(size_t)c].empty() ? 0 : (int64_t)sched[(size_t)c].back().mask + 1;
    toff[(size_t)c + 1] = toff[(size_t)c] + fin;
  }
  const int64_t total = toff[(size_t)n_chunks];
  if (total == 0 || n_stages == 0) return FC_OK;
  if (total >= (1ll << 31)) return set_error(FC_E_LIMIT, "set tables of %lld slots exceed the 32-bit scan", (long long)total);
  std::vector<ChunkStage> desc(n_stages * (size_t)n_chunks, ChunkStage{0, 0, 0, 0});
  std::vector<uint32_t> fin_size((size_t)n_chunks, 0);
  std::vector<int32_t> fin_buf((size_t)n_chunks, 0);
  for (int c = 0; c < n_chunks; ++c) {
    const auto &sc = sched[(size_t)c];
    for (size_t s = 0; s < sc.size(); ++s) desc[s * (size_t)n_chunks + (size_t)c] = ChunkStage{sc[s].n_prev, sc[s].n_cur, sc[s].t_prev, sc[s].mask};
    if (!sc.empty()) {
      fin_size[(size_t)c] = sc.back().mask + 1;
      fin_buf[(size_t)c] = (int32_t)((sc.size() - 1) & 1);
    }
  }
  DevBuf d_desc, d_toff, d_fin_size, d_fin_buf, d_buf[2], d_pos, d_occ, d_occ_scan;
  FC_TRY(d_desc.reserve(desc.size() * sizeof(ChunkStage)));
  FC_TRY(d_toff.reserve(toff.size() * sizeof(int64_t)));
  FC_TRY(d_fin_size.reserve(fin_size.size() * sizeof(uint32_t)));
  FC_TRY(d_fin_buf.reserve(fin_buf.size() * sizeof(int32_t)));
  FC_TRY(h2d(d_desc.p, desc.data(), desc.size() * sizeof(ChunkStage)));
  FC_TRY(h2d(d_toff.p, toff.data(), toff.size() * sizeof(int64_t)));
  FC_TRY(h2d(d_fin_size.p, fin_size.data(), fin_size.size() * sizeof(uint32_t)));
  FC_TRY(h2d(d_fin_buf.p, fin_buf.data(), fin_buf.size() * sizeof(int32_t)));
  FC_TRY(d_buf[0].reserve((size_t)total * sizeof(unsigned long long)));
  FC_TRY(d_buf[1].reserve((size_t)total * sizeof(unsigned long long)));
  FC_TRY(d_pos.reserve((size_t)n_items * sizeof(uint32_t)));
  FC_TRY(d_occ.reserve((size_t)(total + 1) * sizeof(int32_t)));
  FC_TRY(d_occ_scan.reserve((size_t)(total + 1) * sizeof(int32_t)));
  hipStream_t st = cur_stream();
  const dim3 igrid((unsigned)ceil_div(n_items, 256)), block(256);
  uint32_t max_final = 0;
  for (uint32_t f : fin_size) max_final = std::max(max_final, f);
  for (size_t s = 0; s < n_stages; ++s) {
    // only the slots this stage can use need clearing: the largest table of the stage, per chunk region
    uint32_t stage_max = 0;
    for (int c = 0; c < n_chunks; ++c)
      if (desc[s * (size_t)n_chunks + (size_t)c].n_cur) stage_max = std::max(stage_max, desc[s * (size_t)n_chunks + (size_t)c].mask + 1);
    if (stage_max == 0) continue;
    unsigned long long *buf = d_buf[s & 1].as<unsigned long long>();
    const ChunkStage *dst = d_desc.as<ChunkStage>() + s * (size_t)n_chunks;
    // clear the stage's tables of the chunks that HAVE this stage -- never a whole buffer: a chunk with fewer
    // stages keeps its final table in one of the two buffers while the larger chunks go on
    if (s == 0) {
      FC_HIP_TRY(hipMemsetAsync(buf, 0xff, (size_t)total * sizeof(unsigned long long), st));
    } else {
      const unsigned gxc = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, ceil_div((int64_t)stage_max, 1024)));
      hipLaunchKernelGGL(k_set_clear, dim3(gxc, (unsigned)n_chunks), block, 0, st, buf, d_toff.as<int64_t>(), dst, n_chunks);
      FC_TRY(check_launch("k_set_clear"));
    }
    hipLaunchKernelGGL(k_set_insert, igrid, block, 0, st, valid, rank, hash, d_pos.as<uint32_t>(), n_items, d, dst,
                       d_toff.as<int64_t>(), buf);
    FC_TRY(check_launch("k_set_insert"));
    if (s + 1 < n_stages) {
      const unsigned gx = (unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div((int64_t)stage_max, 256)));
      hipLaunchKernelGGL(k_set_positions, dim3(gx, (unsigned)n_chunks), block, 0, st, buf, d_toff.as<int64_t>(), dst, n_chunks,
                         d_pos.as<uint32_t>());
      FC_TRY(check_launch("k_set_positions"));
    }
  }
  const unsigned gx = (unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div((int64_t)max_final, 256)));
  hipLaunchKernelGGL(k_set_occupancy, dim3(gx, (unsigned)n_chunks), block, 0, st, d_buf[0].as<unsigned long long>(),
                     d_buf[1].as<unsigned long long>(), d_toff.as<int64_t>(), d_fin_size.as<uint32_t>(), d_fin_buf.as<int32_t>(),
                     n_chunks, d_occ.as<int32_t>());
  FC_TRY(check_launch("k_set_occupancy"));
  FC_TRY(exclusive_scan_i32(d_occ.as<int32_t>(), d_occ_scan.as<int32_t>(), total, scr));
  hipLaunchKernelGGL(k_set_order, dim3(gx, (unsigned)n_chunks), block, 0, st, d_buf[0].as<unsigned long long>(),
                     d_buf[1].as<unsigned long long>(), d_toff.as<int64_t>(), d_fin_size.as<uint32_t>(), d_fin_buf.as<int32_t>(),
                     n_chunks, d_occ_scan.as<int32_t>(), ebase_dev, qpos_dev, order_dev);
  FC_TRY(check_launch("k_set_order"));
  // the tables are read by kernels still in flight when this function returns: keep them until the stream has passed
  FC_TRY(sync());
  return FC_OK;
}

__global__ void __launch_bounds__(256)
k_pair_hashes(const int64_t *__restrict__ pairs, int64_t n, int64_t *__restrict__ hash, int32_t *__restrict__ valid,
              int32_t *__restrict__ rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  for (int k = 0; k < 2; ++k) {
    acc += (uint64_t)pairs[i * 2 + k] * P2;
    acc = (acc << 31) | (acc >> 33);
    acc *= P1;
  }
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  hash[i] = acc == (uint64_t)-1 ? 1546275796 : (int64_t)acc;
  valid[i] = 1;
  rank[i] = (int32_t)i;
}

}  // namespace

// ---- one coarse ladder level: chunk graphs from the first-match array ----------------------------------------
namespace {

__device__ __forceinline__ int64_t tuple2_hash_dev(uint64_t a, uint64_t b) {
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  acc += a * P2;
  acc = (acc << 31) | (acc >> 33);
  acc *= P1;
  acc += b * P2;
  acc = (acc << 31) | (acc >> 33);
  acc *= P1;
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  return acc == (uint64_t)-1 ? 1546275796 : (int64_t)acc;
}

// items = structures [0, n_items) of the level's non-last chunks (chunk c = [c d, (c + 1) d)); an item is an edge
// when its first match lies inside its own chunk.  valid has n_items + 1 entries (the last one 0: scan total).
__global__ void __launch_bounds__(256)
k_lvl_valid(const int64_t *__restrict__ fm, int64_t n_items, int64_t d, int32_t *__restrict__ valid, int32_t *__restrict__ par) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_items) return;
  if (i == n_items) {
    valid[i] = 0;
    return;
  }
  const int64_t hi = (i / d + 1) * d, j = fm[i];
  const bool v = j >= 0 && j < hi;
  valid[i] = v;
  par[i] = v ? (int32_t)j : (int32_t)i;
}

__global__ void k_gather_index(const int32_t *__restrict__ scan, const int64_t *__restrict__ index, int64_t mul, int n,
                               int64_t *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[c] = scan[mul * index[c]];
}

__global__ void k_gather_at(const int32_t *__restrict__ scan, int64_t stride, int n, int64_t *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) out[c] = scan[(int64_t)c * stride];
}

__global__ void __launch_bounds__(256)
k_lvl_hash_rank(const int64_t *__restrict__ fm, const int32_t *__restrict__ valid, const int32_t *__restrict__ escan,
                const int64_t *__restrict__ ebase, int64_t n_items, int64_t d, int32_t *__restrict__ rank,
                int64_t *__restrict__ hash, int32_t *__restrict__ first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  first[i] = 0x7fffffff;
  if (!valid[i]) return;
  const int64_t c = i / d, lo = c * d;
  rank[i] = escan[i] - (int32_t)ebase[c];
  hash[i] = tuple2_hash_dev((uint64_t)(i - lo), (uint64_t)(fm[i] - lo));
}

// time of an endpoint = 2 q (the edge's first node) / 2 q + 1 (its second): Graph.add_edge(u, v) order
__global__ void __launch_bounds__(256)
k_lvl_first_seen(const int64_t *__restrict__ fm, const int32_t *__restrict__ valid, const int32_t *__restrict__ qpos,
                 int64_t n_items, int32_t *__restrict__ first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items || !valid[i]) return;
  const int32_t q = qpos[i];
  atomicMin(&first[i], 2 * q);
  atomicMin(&first[fm[i]], 2 * q + 1);
}

__global__ void __launch_bounds__(256)
k_lvl_first_flags(const int64_t *__restrict__ fm, const int32_t *__restrict__ valid, const int32_t *__restrict__ qpos,
                  const int32_t *__restrict__ first, const int64_t *__restrict__ ebase, int64_t n_items, int64_t d,
                  int64_t n_times, int32_t *__restrict__ isfirst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) isfirst[n_times] = 0;
  if (i >= n_items || !valid[i]) return;
  const int32_t t0 = 2 * qpos[i];
  const int64_t at = 2 * ebase[i / d] + t0;
  isfirst[at] = first[i] == t0;
  isfirst[at + 1] = first[fm[i]] == t0 + 1;
}

// node number (local to the chunk) of every structure that is in its chunk's graph; nodes[] = the inverse
__global__ void __launch_bounds__(256)
k_lvl_nodes(const int32_t *__restrict__ first, const int32_t *__restrict__ nscan, const int64_t *__restrict__ ebase,
            int64_t n_items, int64_t d, int32_t *__restrict__ node_of, int32_t *__restrict__ nodes) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_items) return;
  const int32_t f = first[x];
  if (f == 0x7fffffff) {
    node_of[x] = -1;
    return;
  }
  const int64_t c = x / d, tb = 2 * ebase[c];
  const int32_t nb = nscan[tb];
  const int32_t v = nscan[tb + f] - nb;
  node_of[x] = v;
  nodes[nb + v] = (int32_t)(x - c * d);
}

// neighbour records in the COMPONENT-MAJOR node order: key = (new node number << 32 | time), value = the other
// end's new node number (both level-wide)
__global__ void __launch_bounds__(256)
k_lvl_adj_records(const int64_t *__restrict__ fm, const int32_t *__restrict__ valid, const int32_t *__restrict__ qpos,
                  const int32_t *__restrict__ node_of, const int32_t *__restrict__ nscan, const int64_t *__restrict__ ebase,
                  const int32_t *__restrict__ newidx, int64_t n_items, int64_t d, uint64_t *__restrict__ keys,
                  int32_t *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items || !valid[i]) return;
  const int64_t c = i / d, eb = ebase[c];
  const int32_t q = qpos[i];
  const int32_t nb = nscan[2 * eb];
  const int32_t u = newidx[nb + node_of[i]], v = newidx[nb + node_of[fm[i]]];
  const int64_t at = 2 * (eb + q);
  keys[at] = ((uint64_t)(uint32_t)u << 32) | (uint32_t)(2 * q);
  vals[at] = v;
  keys[at + 1] = ((uint64_t)(uint32_t)v << 32) | (uint32_t)(2 * q + 1);
  vals[at + 1] = u;
}

__global__ void __launch_bounds__(256)
k_lvl_adj_heads(const uint64_t *__restrict__ keys_sorted, int64_t n, int64_t n_nodes, int32_t *__restrict__ head) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) head[n_nodes] = (int32_t)n;
  if (p >= n) return;
  const uint32_t g = (uint32_t)(keys_sorted[p] >> 32);
  if (p == 0 || (uint32_t)(keys_sorted[p - 1] >> 32) != g) head[g] = (int32_t)p;
}

__global__ void __launch_bounds__(256)
k_lvl_jump(int32_t *__restrict__ par, int64_t n_items) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_items) par[i] = par[par[i]];  // benign race: a value read is always an ancestor
}

__global__ void __launch_bounds__(256)
k_lvl_comp_min(const int32_t *__restrict__ par, const int32_t *__restrict__ node_of, int64_t n_items, int32_t *__restrict__ comp_min) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_items || node_of[x] < 0) return;
  atomicMin(&comp_min[par[x]], node_of[x]);
}

// sort key of every node: (graph-order number of its component's earliest node << 32 | its own graph-order number),
// both level-wide -- components become contiguous, in the order of their earliest nodes, each led by that node
__global__ void __launch_bounds__(256)
k_lvl_comp_keys(const int32_t *__restrict__ par, const int32_t *__restrict__ node_of, const int32_t *__restrict__ comp_min,
                const int32_t *__restrict__ nscan, const int64_t *__restrict__ ebase, int64_t n_items, int64_t d,
                uint64_t *__restrict__ keys) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_items || node_of[x] < 0) return;
  const uint32_t nb = (uint32_t)nscan[2 * ebase[x / d]];
  const uint32_t g = nb + (uint32_t)node_of[x];
  keys[g] = ((uint64_t)(nb + (uint32_t)comp_min[par[x]]) << 32) | g;
}

// from the sorted keys: newidx[old node] = place, nodes_new[place] = relative index, component starts flagged
__global__ void __launch_bounds__(256)
k_lvl_comp_layout(const uint64_t *__restrict__ keys_sorted, const int32_t *__restrict__ nodes_old, int64_t n_nodes,
                  int32_t *__restrict__ newidx, int32_t *__restrict__ nodes_new, int32_t *__restrict__ is_start) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) is_start[n_nodes] = 0;
  if (p >= n_nodes) return;
  const uint32_t g = (uint32_t)keys_sorted[p];
  newidx[g] = (int32_t)p;
  nodes_new[p] = nodes_old[g];
  is_start[p] = p == 0 || (uint32_t)(keys_sorted[p - 1] >> 32) != (uint32_t)(keys_sorted[p] >> 32);
}

__global__ void __launch_bounds__(256)
k_lvl_comp_starts(const int32_t *__restrict__ is_start, const int32_t *__restrict__ sscan, int64_t n_nodes, int32_t *__restrict__ starts) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_nodes || !is_start[p]) return;
  starts[sscan[p]] = (int32_t)p;
}

// ---- the component phase on the device: ONE THREAD PER COMPONENT ------------------------------------------------
// (networkx's _plain_bfs order over the insertion-ordered neighbour lists, `seen` as a Python set of ints, the
// sub-graph view's set rebuilt from it, its first element: group[0]).  Sequential per component by nature; the
// components of a level are tens of thousands of independent small problems (largest ~4e3 nodes at 1.7 M structures).
__host__ __device__ __forceinline__ uint32_t pyset_final_mask(int64_t n_keys) {
  uint64_t mask = 7;
  for (;;) {
    const int64_t trigger = (int64_t)((mask * 3 + 4) / 5);
    if (trigger > n_keys) return (uint32_t)mask;
    const uint64_t minused = trigger > 50000 ? 2 * (uint64_t)trigger : 4 * (uint64_t)trigger;
    uint64_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
    if (trigger == n_keys) return (uint32_t)mask;
  }
}

struct IntSet {  // a CPython set of distinct non-negative ints < 2^31 (hash(n) == n), slots in global scratch
  int32_t *cur, *other;
  uint32_t mask;
  int64_t fill;
  __device__ void init(int32_t *a, int32_t *b) {
    cur = a;
    other = b;
    mask = 7;
    fill = 0;
    for (int s = 0; s < 8; ++s) cur[s] = -1;
  }
  __device__ static void place(int32_t *t, uint32_t mask, int32_t key) {
    uint64_t perturb = (uint64_t)key, i = (uint64_t)key & mask;
    for (;;) {
      const int lim = (i + 9 <= mask) ? 9 : 0;
      for (int j = 0; j <= lim; ++j)
        if (t[i + j] < 0) {
          t[i + j] = key;
          return;
        }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }
  __device__ void add(int32_t key) {
    place(cur, mask, key);
    ++fill;
    if ((uint64_t)fill * 5 >= (uint64_t)mask * 3) {
      const uint64_t minused = fill > 50000 ? 2 * (uint64_t)fill : 4 * (uint64_t)fill;
      uint64_t newsize = 8;
      while (newsize <= minused) newsize <<= 1;
      const uint32_t nm = (uint32_t)(newsize - 1);
      for (uint32_t s = 0; s <= nm; ++s) other[s] = -1;
      for (uint32_t s = 0; s <= mask; ++s)
        if (cur[s] >= 0) place(other, nm, cur[s]);
      int32_t *t = cur;
      cur = other;
      other = t;
      mask = nm;
    }
  }
};

constexpr int64_t kWaveCompMin = 19;  // (up to 18 members the residues fit one 64-bit word of a single lane)
constexpr int64_t kLdsWalkInts = 2 * 306 + 2 * 512;  // LDS ints per wavefront for the walk of a component with a residue collision
// scratch ints a component needs: queue + marks (size each) + two tables of its sets' final size
__global__ void __launch_bounds__(256)
k_lvl_comp_scratch(const int32_t *__restrict__ starts, int64_t n_src, int64_t n_nodes, int64_t size_cap, int32_t *__restrict__ need,
                   int32_t *__restrict__ mid_list, int32_t *__restrict__ mid_count) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n_src) return;
  if (j == n_src) {
    need[j] = 0;
    return;
  }
  const int64_t s0 = starts[j], s1 = j + 1 < n_src ? starts[j + 1] : n_nodes, size = s1 - s0;
  need[j] = (size <= 4 || size > size_cap) ? 0 : (int32_t)(2 * size + 2 * ((int64_t)pyset_final_mask(size) + 1));
  // components of kWaveCompMin .. size_cap nodes get a wavefront each (k_lvl_components_wave); order does not matter
  if (size >= kWaveCompMin && size <= size_cap) mid_list[atomicAdd(mid_count, 1)] = (int32_t)j;
}

// the reference's orders by one lane: _plain_bfs from the component's earliest node over the insertion-ordered neighbour
// lists into a set, then the set of that set's iteration: its first element.  scratch: queue | marks | two tables
__device__ int32_t component_first_by_walk(const int32_t *__restrict__ nodes, const int32_t *__restrict__ head,
                                           const int32_t *__restrict__ adj, int64_t s0, int64_t size, uint32_t fmask,
                                           int32_t *__restrict__ scr) {
  int32_t *q = scr, *mark = q + size, *A = mark + size, *B = A + ((int64_t)fmask + 1);
  for (int64_t v = 0; v < size; ++v) mark[v] = 0;
  IntSet comp;
  comp.init(A, B);
  q[0] = (int32_t)s0;
  mark[0] = 1;
  comp.add(nodes[s0]);
  int64_t qt = 1;
  for (int64_t qh = 0; qh < qt; ++qh) {
    const int32_t v = q[qh];
    for (int32_t rec = head[v]; rec < head[v + 1]; ++rec) {
      const int32_t x = adj[rec];
      if (!mark[x - s0]) {
        mark[x - s0] = 1;
        q[qt++] = x;
        comp.add(nodes[x]);
      }
    }
  }
  int64_t n_ord = 0;  // the component set's iteration order (over the marks, no longer needed)
  for (uint32_t sl = 0; sl <= comp.mask; ++sl)
    if (comp.cur[sl] >= 0) mark[n_ord++] = comp.cur[sl];
  IntSet view;
  view.init(A, B);
  for (int64_t k = 0; k < n_ord; ++k) view.add(mark[k]);
  for (uint32_t sl = 0; sl <= view.mask; ++sl)
    if (view.cur[sl] >= 0) return view.cur[sl];
  return -1;
}

// One WAVEFRONT per component of kWaveCompMin .. size_cap nodes.  Almost all of them are "clean": no two members share a
// residue modulo the final table size of the set the reference builds, and the kept node is then simply the member with
// the smallest residue (the host's shortcut, fc_tfd_host.cpp graph_component) -- a bitmap of the residues in LDS filled
// with atomicOr by 64 lanes and a minimum over the wavefront, instead of one lane walking up to 1 024 nodes through
// tables in global memory (which made that lane's wavefront the tail of the whole launch: 6.4 ms at k = 5).  A component
// with a collision is walked by lane 0 as before.
__global__ void __launch_bounds__(256)
k_lvl_components_wave(const int32_t *__restrict__ nodes, const int32_t *__restrict__ head, const int32_t *__restrict__ adj,
                      const int32_t *__restrict__ starts, int64_t n_src, int64_t n_nodes, const int64_t *__restrict__ nbase,
                      int n_chunks, int64_t d, const int32_t *__restrict__ soff, int32_t *__restrict__ scratch,
                      uint8_t *__restrict__ flags, const int32_t *__restrict__ mid_list, const int32_t *__restrict__ mid_count,
                      int words_per_wave, int64_t dirty_walk_max, int32_t *__restrict__ left_list, int32_t *__restrict__ left_count) {
  extern __shared__ uint32_t bm_all[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t *bm = bm_all + (size_t)wv * words_per_wave;
  for (int64_t idx = (int64_t)blockIdx.x * 4 + wv; idx < (int64_t)*mid_count; idx += (int64_t)gridDim.x * 4) {  // wave-uniform
    const int64_t j = mid_list[idx];
    const int64_t s0 = starts[j], s1 = j + 1 < n_src ? starts[j + 1] : n_nodes, size = s1 - s0;
    int lo = 0, hi = n_chunks;  // chunk of the component: nbase[c] <= s0 < nbase[c + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (nbase[mid] <= s0) lo = mid;
      else hi = mid;
    }
    const int c = lo;
    const int64_t chunk_nodes = nbase[c + 1] - nbase[c];
    int32_t first = -1;
    if (2 * size >= chunk_nodes) {
      first = nodes[s0];  // FilterAtlas walks the whole graph's nodes: the component's earliest one comes first
    } else {
      const uint32_t fmask = pyset_final_mask(size);
      const int words = (int)(((int64_t)fmask + 32) >> 5);
      for (int w = lane; w < words; w += 64) bm[w] = 0u;
      __builtin_amdgcn_wave_barrier();  // (LDS operations of one wavefront complete in order)
      bool dirty = false;
      unsigned long long best = ~0ull;  // (residue << 32) | node
      for (int64_t v = s0 + lane; v < s1; v += 64) {
        const int32_t node = nodes[v];
        const uint32_t r = (uint32_t)node & fmask, bit = 1u << (r & 31);
        const uint32_t old = atomicOr(&bm[r >> 5], bit);
        dirty = dirty || ((old & bit) != 0u);
        const unsigned long long key = ((unsigned long long)r << 32) | (unsigned long long)(uint32_t)node;
        best = key < best ? key : best;
      }
      if (__ballot(dirty) == 0ull) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long o = __shfl_xor(best, off);
          best = o < best ? o : best;
        }
        first = (int32_t)(uint32_t)(best & 0xffffffffull);
      } else if (size > dirty_walk_max) {
        // a collision in a large component: the walk is sequential, a host core does a node in ~20 ns where a lane here
        // needs ~1 us -- to the host's list (with the components above the cap); nothing flagged here
        if (lane == 0) left_list[atomicAdd(left_count, 1)] = (int32_t)j;
        __builtin_amdgcn_wave_barrier();
        continue;
      } else {
        // the walk's queue, marks and two tables in LDS when they fit (components of up to 306 nodes: tables of 512
        // slots), in global scratch otherwise: a lane's dependent accesses cost ~0.1 us there against ~1 us
        const int64_t need = 2 * size + 2 * ((int64_t)fmask + 1);
        int32_t *scr = need <= kLdsWalkInts ? reinterpret_cast<int32_t *>(bm_all + (size_t)4 * words_per_wave) + (size_t)wv * kLdsWalkInts
                                            : scratch + soff[j];
        if (lane == 0) first = component_first_by_walk(nodes, head, adj, s0, size, fmask, scr);
        first = __shfl(first, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
    uint8_t *__restrict__ f = flags + (int64_t)c * d;
    for (int64_t v = s0 + lane; v < s1; v += 64) {
      const int32_t node = nodes[v];
      if (node != first) f[node] = 1;
    }
  }
}

__global__ void __launch_bounds__(64)
k_lvl_components(const int32_t *__restrict__ nodes, const int32_t *__restrict__ head, const int32_t *__restrict__ adj,
                 const int32_t *__restrict__ starts, int64_t n_src, int64_t n_nodes, const int64_t *__restrict__ nbase,
                 int n_chunks, int64_t d, const int32_t *__restrict__ soff, int32_t *__restrict__ scratch,
                 uint8_t *__restrict__ flags, int64_t size_cap, int32_t *__restrict__ left_list, int32_t *__restrict__ left_count) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_src) return;
  const int64_t s0 = starts[j], s1 = j + 1 < n_src ? starts[j + 1] : n_nodes, size = s1 - s0;
  int lo = 0, hi = n_chunks;  // chunk of the component: nbase[c] <= s0 < nbase[c + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (nbase[mid] <= s0) lo = mid;
    else hi = mid;
  }
  const int c = lo;
  const int64_t chunk_nodes = nbase[c + 1] - nbase[c];
  int32_t first = -1;
  if (size >= kWaveCompMin && size <= size_cap) return;  // k_lvl_components_wave's
  if (2 * size >= chunk_nodes) {
    first = nodes[s0];  // FilterAtlas walks the whole graph's nodes: the component's earliest one comes first
  } else if (size > size_cap) {
    // a long sequential walk: left to a host thread (one lane would hold its wavefront for milliseconds)
    left_list[atomicAdd(left_count, 1)] = (int32_t)j;
    return;
  } else {
    const uint32_t fmask = pyset_final_mask(size);
    bool clean = false;
    if (fmask < 64) {  // up to 18 members: residues in one 64-bit word
      uint64_t seen = 0;
      uint32_t best = 64;
      clean = true;
      for (int64_t v = s0; v < s1; ++v) {
        const uint32_t r = (uint32_t)nodes[v] & fmask;
        if ((seen >> r) & 1ull) {
          clean = false;
          break;
        }
        seen |= 1ull << r;
        if (r < best) {
          best = r;
          first = nodes[v];
        }
      }
    }
    if (!clean) {
      if (size <= 4) {
        // at most four members and a collision: tables of 8 slots, no growth -- in registers
        int32_t t1[8], t2[8], q[4];
        for (int k = 0; k < 8; ++k) t1[k] = t2[k] = -1;
        uint32_t mark = 1;
        int qt = 1;
        q[0] = (int32_t)s0;
        auto put = [](int32_t (&t)[8], int32_t key) {
          uint64_t perturb = (uint64_t)key, i = (uint64_t)key & 7;
          for (;;) {  // a table of 8 has no linear window (i + 9 > mask): one slot per step
            if (t[i] < 0) {
              t[i] = key;
              return;
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & 7;
          }
        };
        put(t1, nodes[s0]);
        for (int qh = 0; qh < qt; ++qh) {
          const int32_t v = q[qh];
          for (int32_t rec = head[v]; rec < head[v + 1]; ++rec) {
            const int32_t x = adj[rec];
            if (!((mark >> (x - s0)) & 1u)) {
              mark |= 1u << (x - s0);
              q[qt++] = x;
              put(t1, nodes[x]);
            }
          }
        }
        for (int k = 0; k < 8; ++k)
          if (t1[k] >= 0) put(t2, t1[k]);
        for (int k = 7; k >= 0; --k)
          if (t2[k] >= 0) first = t2[k];
      } else {
        first = component_first_by_walk(nodes, head, adj, s0, size, fmask, scratch + soff[j]);
      }
    }
  }
  uint8_t *__restrict__ f = flags + (int64_t)c * d;
  for (int64_t v = s0; v < s1; ++v)
    if (nodes[v] != first) f[nodes[v]] = 1;
}

}  // namespace

// flags_out != nullptr: the component phase runs on the device too and flags_out[i] = 1 for every structure a
// non-last chunk of the level rejects (d (k - 1) bytes); the graph arrays are then not downloaded.
// The components the device left to the host (more than size_cap nodes), cut out of the level's arrays: sizes ...
__global__ void __launch_bounds__(256)
k_left_sizes(const int32_t *__restrict__ left, int n_left, const int32_t *__restrict__ starts, int64_t n_src, int64_t n_nodes,
             const int32_t *__restrict__ head, const int64_t *__restrict__ nbase, int n_chunks, int32_t *__restrict__ cnt_nodes,
             int32_t *__restrict__ cnt_adj, int32_t *__restrict__ chunk_of) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j > n_left) return;
  if (j == n_left) {  // (the scans run over n_left + 1 entries)
    cnt_nodes[j] = 0, cnt_adj[j] = 0;
    return;
  }
  const int32_t c = left[j];
  const int64_t s0 = starts[c], s1 = (int64_t)c + 1 < n_src ? (int64_t)starts[c + 1] : n_nodes;
  cnt_nodes[j] = (int32_t)(s1 - s0);
  cnt_adj[j] = head[s1] - head[s0];
  int lo = 0, hi = n_chunks;  // largest chunk with nbase[chunk] <= s0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (nbase[mid] <= s0) lo = mid;
    else hi = mid;
  }
  chunk_of[j] = lo;
}
// ... and contents, in compact numbering: block j copies component j (nodes, neighbour-list heads, neighbours)
__global__ void __launch_bounds__(256)
k_left_gather(const int32_t *__restrict__ left, int n_left, const int32_t *__restrict__ starts, int64_t n_src, int64_t n_nodes,
              const int32_t *__restrict__ nodes, const int32_t *__restrict__ head, const int32_t *__restrict__ adj,
              const int32_t *__restrict__ off_nodes, const int32_t *__restrict__ off_adj, int32_t *__restrict__ c_nodes,
              int32_t *__restrict__ c_head, int32_t *__restrict__ c_adj) {
  const int j = blockIdx.x;
  if (j >= n_left) return;
  const int32_t c = left[j];
  const int64_t s0 = starts[c], s1 = (int64_t)c + 1 < n_src ? (int64_t)starts[c + 1] : n_nodes;
  const int32_t a0 = head[s0], a1 = head[s1];
  const int32_t on = off_nodes[j], oa = off_adj[j];
  for (int64_t i = threadIdx.x; i < s1 - s0; i += 256) {
    c_nodes[on + i] = nodes[s0 + i];
    c_head[on + i] = oa + (head[s0 + i] - a0);
  }
  for (int32_t e = threadIdx.x; e < a1 - a0; e += 256) c_adj[oa + e] = (int32_t)((int64_t)adj[a0 + e] - s0 + on);
  if (threadIdx.x == 0 && j == n_left - 1) c_head[on + (s1 - s0)] = oa + (a1 - a0);  // the closing entry of the last list
}

// One level's chunk graphs (and, with flags_out, its component phase) on cur_stream().  Levels are independent: the
// ladder runs several at a time from threads of its own, each on its own stream (thread_stream_override).
int tfd_level_graph_device(const int64_t *fm_dev, int64_t N, int64_t k, TfdLevelGraph &out, uint8_t *flags_out) {
  const int64_t d = N / k;
  const int n_chunks = (int)(k - 1);
  const int64_t n_items = d * n_chunks;
  out.n_chunks = n_chunks;
  out.d = d;
  out.ebase.assign((size_t)n_chunks + 1, 0);
  out.nbase.assign((size_t)n_chunks + 1, 0);
  out.sbase.assign((size_t)n_chunks + 1, 0);
  out.nodes.clear(); out.adj_head.clear(); out.adj_next.clear(); out.sources.clear();
  out.left.clear(); out.left_chunk.clear();  // (holders are reused: nothing of the previous level may survive an early return)
  out.n_components = 0;
  if (n_chunks <= 0 || n_items <= 0) return FC_OK;
  if (n_items >= (1ll << 30)) return set_error(FC_E_LIMIT, "level too large for the device graph builder");
  hipStream_t st = cur_stream();
  // every temporary of the level from one block: ~200 B per item for the arrays below plus the set tables (24 B per
  // slot, up to 6.7 slots per key); what does not fit comes from the pool as before
  ArenaScope arena;
  FC_TRY(arena.begin((size_t)n_items * 400 + ((size_t)32 << 20)));
  Scratch scr;
  const dim3 block(256), igrid((unsigned)ceil_div(n_items + 1, 256));
  static const bool dbg = getenv("FC_DEBUG") != nullptr && getenv("FC_TFD_LAPS") != nullptr;
  auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg) return;
    (void)hipStreamSynchronize(st);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[fc]     k=%lld %s %.2f ms\n", (long long)k, what, std::chrono::duration<double, std::milli>(t - T0).count());
    T0 = t;
  };
  DevBuf d_valid, d_par, d_escan, d_ebase, d_rank, d_hash, d_first, d_qpos;
  FC_TRY(d_valid.reserve((size_t)(n_items + 1) * sizeof(int32_t)));
  FC_TRY(d_par.reserve((size_t)n_items * sizeof(int32_t)));
  FC_TRY(d_escan.reserve((size_t)(n_items + 1) * sizeof(int32_t)));
  FC_TRY(d_ebase.reserve((size_t)(n_chunks + 1) * sizeof(int64_t)));
  hipLaunchKernelGGL(k_lvl_valid, igrid, block, 0, st, fm_dev, n_items, d, d_valid.as<int32_t>(), d_par.as<int32_t>());
  FC_TRY(check_launch("k_lvl_valid"));
  FC_TRY(exclusive_scan_i32(d_valid.as<int32_t>(), d_escan.as<int32_t>(), n_items + 1, scr));
  hipLaunchKernelGGL(k_gather_at, dim3((unsigned)ceil_div((int64_t)n_chunks + 1, 64)), dim3(64), 0, st, d_escan.as<int32_t>(), d,
                     n_chunks + 1, d_ebase.as<int64_t>());
  FC_TRY(check_launch("k_gather_at"));
  FC_TRY(d2h(out.ebase.data(), d_ebase.p, (size_t)(n_chunks + 1) * sizeof(int64_t)));
  FC_TRY(sync());
  const int64_t M = out.ebase[(size_t)n_chunks];
  if (M == 0) return FC_OK;
  std::vector<int64_t> m((size_t)n_chunks);
  for (int c = 0; c < n_chunks; ++c) m[(size_t)c] = out.ebase[(size_t)c + 1] - out.ebase[(size_t)c];
  FC_TRY(d_rank.reserve((size_t)n_items * sizeof(int32_t)));
  FC_TRY(d_hash.reserve((size_t)n_items * sizeof(int64_t)));
  FC_TRY(d_first.reserve((size_t)n_items * sizeof(int32_t)));
  FC_TRY(d_qpos.reserve((size_t)n_items * sizeof(int32_t)));
  FC_HIP_TRY(hipMemsetAsync(d_qpos.p, 0, (size_t)n_items * sizeof(int32_t), st));  // (defined values whatever happens upstream)
  hipLaunchKernelGGL(k_lvl_hash_rank, igrid, block, 0, st, fm_dev, d_valid.as<int32_t>(), d_escan.as<int32_t>(), d_ebase.as<int64_t>(),
                     n_items, d, d_rank.as<int32_t>(), d_hash.as<int64_t>(), d_first.as<int32_t>());
  FC_TRY(check_launch("k_lvl_hash_rank"));
  // (1) the edge order: iteration order of each chunk's set of (i_rel, j_rel) tuples
  FC_TRY(pyset_orders_device(d_valid.as<int32_t>(), d_rank.as<int32_t>(), d_hash.as<int64_t>(), n_items, d, n_chunks, m,
                             d_ebase.as<int64_t>(), d_qpos.as<int32_t>(), nullptr, scr));
  lap("edges + set orders");
  // (2) node numbers = ranks of first appearance along the edge order
  const int64_t n_times = 2 * M;
  DevBuf d_isfirst, d_nscan, d_node_of, d_nodes;
  FC_TRY(d_isfirst.reserve((size_t)(n_times + 1) * sizeof(int32_t)));
  FC_TRY(d_nscan.reserve((size_t)(n_times + 1) * sizeof(int32_t)));
  FC_TRY(d_node_of.reserve((size_t)n_items * sizeof(int32_t)));
  FC_TRY(d_nodes.reserve((size_t)n_items * sizeof(int32_t)));
  hipLaunchKernelGGL(k_lvl_first_seen, igrid, block, 0, st, fm_dev, d_valid.as<int32_t>(), d_qpos.as<int32_t>(), n_items,
                     d_first.as<int32_t>());
  FC_TRY(check_launch("k_lvl_first_seen"));
  hipLaunchKernelGGL(k_lvl_first_flags, igrid, block, 0, st, fm_dev, d_valid.as<int32_t>(), d_qpos.as<int32_t>(), d_first.as<int32_t>(),
                     d_ebase.as<int64_t>(), n_items, d, n_times, d_isfirst.as<int32_t>());
  FC_TRY(check_launch("k_lvl_first_flags"));
  FC_TRY(exclusive_scan_i32(d_isfirst.as<int32_t>(), d_nscan.as<int32_t>(), n_times + 1, scr));
  hipLaunchKernelGGL(k_lvl_nodes, igrid, block, 0, st, d_first.as<int32_t>(), d_nscan.as<int32_t>(), d_ebase.as<int64_t>(), n_items, d,
                     d_node_of.as<int32_t>(), d_nodes.as<int32_t>());
  FC_TRY(check_launch("k_lvl_nodes"));
  DevBuf d_nbase;
  FC_TRY(d_nbase.reserve((size_t)(n_chunks + 1) * sizeof(int64_t)));
  {  // node bases per chunk: nscan at 2 ebase[c]
    hipLaunchKernelGGL(k_gather_index, dim3((unsigned)ceil_div((int64_t)n_chunks + 1, 64)), dim3(64), 0, st, d_nscan.as<int32_t>(),
                       d_ebase.as<int64_t>(), 2, n_chunks + 1, d_nbase.as<int64_t>());
    FC_TRY(check_launch("k_gather_index"));
    FC_TRY(d2h(out.nbase.data(), d_nbase.p, (size_t)(n_chunks + 1) * sizeof(int64_t)));
    FC_TRY(sync());
  }
  const int64_t n_nodes = out.nbase[(size_t)n_chunks];
  lap("node numbers");
  // (3) components: pointer jumping to the tree roots (a first-match graph is a forest: one out-edge per node, to a
  // later one), the earliest node of every tree, then the COMPONENT-MAJOR node order -- the host's component phase
  // walks each component in one contiguous block instead of hopping through a 10^6-node array in hash order
  DevBuf d_cmin, d_ckeys, d_ckeys2, d_newidx, d_nodes2, d_isstart, d_sscan, d_starts;
  {
    int rounds = 1;
    while ((1ll << rounds) < d) ++rounds;
    for (int r = 0; r < rounds + 1; ++r) {
      hipLaunchKernelGGL(k_lvl_jump, igrid, block, 0, st, d_par.as<int32_t>(), n_items);
      FC_TRY(check_launch("k_lvl_jump"));
    }
  }
  FC_TRY(d_cmin.reserve((size_t)n_items * sizeof(int32_t)));
  FC_HIP_TRY(hipMemsetAsync(d_cmin.p, 0x7f, (size_t)n_items * sizeof(int32_t), st));
  FC_TRY(d_ckeys.reserve((size_t)n_nodes * sizeof(uint64_t)));
  FC_TRY(d_ckeys2.reserve((size_t)n_nodes * sizeof(uint64_t)));
  FC_TRY(d_newidx.reserve((size_t)n_nodes * sizeof(int32_t)));
  FC_TRY(d_nodes2.reserve((size_t)n_nodes * sizeof(int32_t)));
  FC_TRY(d_isstart.reserve((size_t)(n_nodes + 1) * sizeof(int32_t)));
  FC_TRY(d_sscan.reserve((size_t)(n_nodes + 1) * sizeof(int32_t)));
  FC_TRY(d_starts.reserve((size_t)(n_nodes + 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_lvl_comp_min, igrid, block, 0, st, d_par.as<int32_t>(), d_node_of.as<int32_t>(), n_items, d_cmin.as<int32_t>());
  FC_TRY(check_launch("k_lvl_comp_min"));
  hipLaunchKernelGGL(k_lvl_comp_keys, igrid, block, 0, st, d_par.as<int32_t>(), d_node_of.as<int32_t>(), d_cmin.as<int32_t>(),
                     d_nscan.as<int32_t>(), d_ebase.as<int64_t>(), n_items, d, d_ckeys.as<uint64_t>());
  FC_TRY(check_launch("k_lvl_comp_keys"));
  int node_bits = 1;
  while ((1ll << node_bits) < n_nodes + 1) ++node_bits;
  {
    size_t bytes = 0;
    FC_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, d_ckeys.as<uint64_t>(), d_ckeys2.as<uint64_t>(), (int)n_nodes, 0,
                                                 32 + node_bits, st));
    FC_TRY(scr.reserve(bytes));
    FC_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(scr.tmp.p, bytes, d_ckeys.as<uint64_t>(), d_ckeys2.as<uint64_t>(), (int)n_nodes, 0,
                                                 32 + node_bits, st));
  }
  hipLaunchKernelGGL(k_lvl_comp_layout, dim3((unsigned)ceil_div(n_nodes, 256)), block, 0, st, d_ckeys2.as<uint64_t>(), d_nodes.as<int32_t>(),
                     n_nodes, d_newidx.as<int32_t>(), d_nodes2.as<int32_t>(), d_isstart.as<int32_t>());
  FC_TRY(check_launch("k_lvl_comp_layout"));
  FC_TRY(exclusive_scan_i32(d_isstart.as<int32_t>(), d_sscan.as<int32_t>(), n_nodes + 1, scr));
  hipLaunchKernelGGL(k_lvl_comp_starts, dim3((unsigned)ceil_div(n_nodes, 256)), block, 0, st, d_isstart.as<int32_t>(), d_sscan.as<int32_t>(),
                     n_nodes, d_starts.as<int32_t>());
  FC_TRY(check_launch("k_lvl_comp_starts"));
  lap("components, component-major order");
  // (4) neighbour lists in insertion order, in the new node numbers: records sorted by (node, time)
  DevBuf d_keys, d_keys2, d_vals, d_vals2, d_head;
  FC_TRY(d_keys.reserve((size_t)n_times * sizeof(uint64_t)));
  FC_TRY(d_keys2.reserve((size_t)n_times * sizeof(uint64_t)));
  FC_TRY(d_vals.reserve((size_t)n_times * sizeof(int32_t)));
  FC_TRY(d_vals2.reserve((size_t)n_times * sizeof(int32_t)));
  FC_TRY(d_head.reserve((size_t)(n_nodes + 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_lvl_adj_records, igrid, block, 0, st, fm_dev, d_valid.as<int32_t>(), d_qpos.as<int32_t>(), d_node_of.as<int32_t>(),
                     d_nscan.as<int32_t>(), d_ebase.as<int64_t>(), d_newidx.as<int32_t>(), n_items, d, d_keys.as<uint64_t>(),
                     d_vals.as<int32_t>());
  FC_TRY(check_launch("k_lvl_adj_records"));
  {
    size_t bytes = 0;
    FC_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<int32_t>(),
                                                  d_vals2.as<int32_t>(), (int)n_times, 0, 32 + node_bits, st));
    FC_TRY(scr.reserve(bytes));
    FC_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(scr.tmp.p, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<int32_t>(),
                                                  d_vals2.as<int32_t>(), (int)n_times, 0, 32 + node_bits, st));
  }
  hipLaunchKernelGGL(k_lvl_adj_heads, dim3((unsigned)ceil_div(n_times, 256)), block, 0, st, d_keys2.as<uint64_t>(), n_times, n_nodes,
                     d_head.as<int32_t>());
  FC_TRY(check_launch("k_lvl_adj_heads"));
  DevBuf d_sbase;
  FC_TRY(d_sbase.reserve((size_t)(n_chunks + 1) * sizeof(int64_t)));
  {  // components before every chunk
    hipLaunchKernelGGL(k_gather_index, dim3((unsigned)ceil_div((int64_t)n_chunks + 1, 64)), dim3(64), 0, st, d_sscan.as<int32_t>(),
                       d_nbase.as<int64_t>(), 1, n_chunks + 1, d_sbase.as<int64_t>());
    FC_TRY(check_launch("k_gather_index"));
    FC_TRY(d2h(out.sbase.data(), d_sbase.p, (size_t)(n_chunks + 1) * sizeof(int64_t)));
    FC_TRY(sync());
  }
  const int64_t n_src = out.sbase[(size_t)n_chunks];
  lap("adjacency");
  DevBuf d_need, d_soff, d_scratch, d_flags, d_left;  // (function scope: copies from them are still in flight below)
  DevBuf d_lcn, d_lca, d_lon, d_loa, d_lchunk, d_cnodes, d_chead, d_cadj, d_mid;
  if (flags_out != nullptr) {
    static const int64_t size_cap = [] {
      const char *v = getenv("FC_TFD_DEV_COMP_MAX");  // components above this many nodes go to the host threads
      // 256 .. 4096 measured at 1.7 M structures (largest component there: 3 837 nodes): 1024 leaves the three finest
      // device levels nothing to send down (their graph arrays are ~20 MB each, first touched by the first call) at the same
      // steady time as 256; 1536 and above make the longest lane of the component kernel the critical path
      const long long k = v ? std::strtoll(v, nullptr, 10) : 1024;
      return (int64_t)(k >= 4 ? std::min<long long>(k, 16384) : 1024);  // (4 waves x one residue bitmap each in LDS)
    }();
    FC_TRY(d_need.reserve((size_t)(n_src + 1) * sizeof(int32_t)));
    FC_TRY(d_soff.reserve((size_t)(n_src + 1) * sizeof(int32_t)));
    FC_TRY(d_left.reserve((size_t)(n_src + 1) * sizeof(int32_t)));
    FC_HIP_TRY(hipMemsetAsync(d_left.p, 0, sizeof(int32_t), st));  // [0] = count, [1..] = component numbers
    FC_TRY(d_mid.reserve((size_t)(n_nodes / kWaveCompMin + 2) * sizeof(int32_t)));  // [0] = count, [1..] = component numbers
    FC_HIP_TRY(hipMemsetAsync(d_mid.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_lvl_comp_scratch, dim3((unsigned)ceil_div(n_src + 1, 256)), block, 0, st, d_starts.as<int32_t>(), n_src, n_nodes,
                       size_cap, d_need.as<int32_t>(), d_mid.as<int32_t>() + 1, d_mid.as<int32_t>());
    FC_TRY(check_launch("k_lvl_comp_scratch"));
    FC_TRY(exclusive_scan_i32(d_need.as<int32_t>(), d_soff.as<int32_t>(), n_src + 1, scr));
    int32_t total_need = 0;
    FC_TRY(d2h(&total_need, d_soff.as<int32_t>() + n_src, sizeof(int32_t)));
    FC_TRY(sync());
    if (total_need < 0) return set_error(FC_E_LIMIT, "component scratch exceeds 2^31 ints");
    FC_TRY(d_scratch.reserve((size_t)std::max<int64_t>(total_need, 2) * sizeof(int32_t)));
    FC_TRY(d_flags.reserve((size_t)n_items));
    FC_HIP_TRY(hipMemsetAsync(d_flags.p, 0, (size_t)n_items, st));
    hipLaunchKernelGGL(k_lvl_components, dim3((unsigned)ceil_div(n_src, 64)), dim3(64), 0, st, d_nodes2.as<int32_t>(), d_head.as<int32_t>(),
                       d_vals2.as<int32_t>(), d_starts.as<int32_t>(), n_src, n_nodes, d_nbase.as<int64_t>(), n_chunks, d,
                       d_soff.as<int32_t>(), d_scratch.as<int32_t>(), d_flags.as<uint8_t>(), size_cap, d_left.as<int32_t>() + 1,
                       d_left.as<int32_t>());
    FC_TRY(check_launch("k_lvl_components"));
    static const int64_t dirty_walk_max = [] {
      // components with a residue collision above this many nodes go to the host.  At 1.7 M structures 900 - 4 900
      // components per level have a collision; up to 306 nodes their walk runs out of LDS (kLdsWalkInts), the 36 - 218
      // larger ones per level were the tail of this kernel (4.4 - 5.1 -> 2.7 - 3.4 ms per level, + 0.5 - 3 ms on two host
      // threads of the level's helper); a limit of 64 sends thousands down: 4 - 16 ms per level
      const char *v = getenv("FC_TFD_DEV_DIRTY_MAX");
      const long long k = v ? std::strtoll(v, nullptr, 10) : 306;
      return (int64_t)(k >= 19 ? k : 306);
    }();
    {  // the components of kWaveCompMin .. size_cap nodes, a wavefront each (the list's length stays on the device)
      const int words = (int)(((int64_t)pyset_final_mask(size_cap) + 32) >> 5);
      const int64_t most = n_nodes / kWaveCompMin + 1;
      const unsigned wgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(most, 4), (int64_t)ctx().n_cu * 16));
      hipLaunchKernelGGL(k_lvl_components_wave, dim3(wgrid), dim3(256), (size_t)4 * (words + kLdsWalkInts) * sizeof(uint32_t), st, d_nodes2.as<int32_t>(),
                         d_head.as<int32_t>(), d_vals2.as<int32_t>(), d_starts.as<int32_t>(), n_src, n_nodes, d_nbase.as<int64_t>(),
                         n_chunks, d, d_soff.as<int32_t>(), d_scratch.as<int32_t>(), d_flags.as<uint8_t>(), d_mid.as<int32_t>() + 1,
                         d_mid.as<int32_t>(), words, dirty_walk_max, d_left.as<int32_t>() + 1, d_left.as<int32_t>());
      FC_TRY(check_launch("k_lvl_components_wave"));
    }
    int32_t n_left = 0;
    FC_TRY(d2h(flags_out, d_flags.p, (size_t)n_items));
    FC_TRY(d2h(&n_left, d_left.p, sizeof(int32_t)));
    FC_TRY(sync());
    lap("device components + flags down");
    out.left.clear();
    out.left_chunk.clear();
    out.n_components = n_src;
    out.sources.assign(1, 0);
    if (n_left == 0) return FC_OK;
    // ... and the host's walk over those components needs THEIR nodes and neighbour lists only (at 1.7 M structures a
    // few dozen components of up to 3 837 nodes out of 1.6 M nodes: kilobytes instead of the level's 20 MB of arrays)
    FC_TRY(d_lcn.reserve((size_t)(n_left + 1) * sizeof(int32_t)));
    FC_TRY(d_lca.reserve((size_t)(n_left + 1) * sizeof(int32_t)));
    FC_TRY(d_lon.reserve((size_t)(n_left + 1) * sizeof(int32_t)));
    FC_TRY(d_loa.reserve((size_t)(n_left + 1) * sizeof(int32_t)));
    FC_TRY(d_lchunk.reserve((size_t)(n_left + 1) * sizeof(int32_t)));
    hipLaunchKernelGGL(k_left_sizes, dim3((unsigned)ceil_div((int64_t)n_left + 1, 256)), block, 0, st, d_left.as<int32_t>() + 1, (int)n_left,
                       d_starts.as<int32_t>(), n_src, n_nodes, d_head.as<int32_t>(), d_nbase.as<int64_t>(), n_chunks,
                       d_lcn.as<int32_t>(), d_lca.as<int32_t>(), d_lchunk.as<int32_t>());
    FC_TRY(check_launch("k_left_sizes"));
    FC_TRY(exclusive_scan_i32(d_lcn.as<int32_t>(), d_lon.as<int32_t>(), (int64_t)n_left + 1, scr));
    FC_TRY(exclusive_scan_i32(d_lca.as<int32_t>(), d_loa.as<int32_t>(), (int64_t)n_left + 1, scr));
    int32_t tot_nodes = 0, tot_adj = 0;
    FC_TRY(d2h(&tot_nodes, d_lon.as<int32_t>() + n_left, sizeof(int32_t)));
    FC_TRY(d2h(&tot_adj, d_loa.as<int32_t>() + n_left, sizeof(int32_t)));
    FC_TRY(sync());
    FC_TRY(d_cnodes.reserve((size_t)std::max(tot_nodes, 1) * sizeof(int32_t)));
    FC_TRY(d_chead.reserve(((size_t)tot_nodes + 1) * sizeof(int32_t)));
    FC_TRY(d_cadj.reserve((size_t)std::max(tot_adj, 1) * sizeof(int32_t)));
    hipLaunchKernelGGL(k_left_gather, dim3((unsigned)n_left), block, 0, st, d_left.as<int32_t>() + 1, (int)n_left, d_starts.as<int32_t>(),
                       n_src, n_nodes, d_nodes2.as<int32_t>(), d_head.as<int32_t>(), d_vals2.as<int32_t>(), d_lon.as<int32_t>(),
                       d_loa.as<int32_t>(), d_cnodes.as<int32_t>(), d_chead.as<int32_t>(), d_cadj.as<int32_t>());
    FC_TRY(check_launch("k_left_gather"));
    out.left.resize((size_t)n_left);
    for (int32_t j = 0; j < n_left; ++j) out.left[(size_t)j] = j;
    out.left_chunk.resize((size_t)n_left);
    out.nodes.resize((size_t)tot_nodes);
    out.adj_head.resize((size_t)tot_nodes + 1);
    out.adj_next.resize((size_t)tot_adj);
    out.sources.resize((size_t)n_left + 1);
    FC_TRY(d2h(out.left_chunk.data(), d_lchunk.p, (size_t)n_left * sizeof(int32_t)));
    FC_TRY(d2h(out.nodes.data(), d_cnodes.p, (size_t)tot_nodes * sizeof(int32_t)));
    FC_TRY(d2h(out.adj_head.data(), d_chead.p, ((size_t)tot_nodes + 1) * sizeof(int32_t)));
    FC_TRY(d2h(out.adj_next.data(), d_cadj.p, (size_t)tot_adj * sizeof(int32_t)));
    FC_TRY(d2h(out.sources.data(), d_lon.p, ((size_t)n_left + 1) * sizeof(int32_t)));
    FC_TRY(sync());
    lap("left components down");
    return FC_OK;
  }
  out.left_chunk.clear();
  out.n_components = n_src;
  out.nodes.resize((size_t)n_nodes);
  out.adj_head.resize((size_t)n_nodes + 1);
  out.adj_next.resize((size_t)n_times);
  out.sources.resize((size_t)n_src + 1);
  FC_TRY(d2h(out.nodes.data(), d_nodes2.p, (size_t)n_nodes * sizeof(int32_t)));
  FC_TRY(d2h(out.adj_head.data(), d_head.p, (size_t)(n_nodes + 1) * sizeof(int32_t)));
  FC_TRY(d2h(out.adj_next.data(), d_vals2.p, (size_t)n_times * sizeof(int32_t)));
  FC_TRY(d2h(out.sources.data(), d_starts.p, (size_t)n_src * sizeof(int32_t)));
  FC_TRY(sync());
  lap("graph arrays down");
  out.sources[(size_t)n_src] = (int32_t)n_nodes;  // sentinel: component j = nodes [sources[j], sources[j + 1])
  return FC_OK;
}

// test hook: iteration order (indices into `pairs`) of a Python set of n DISTINCT 2-tuples inserted in order,
// computed on the device -- must equal pyset_order_pairs (fc_tfd_host.cpp)
int pyset_order_pairs_device(const int64_t *pairs_host, int64_t n, int64_t *order_out) {
  if (n == 0) return FC_OK;
  if (n >= (1ll << 30)) return set_error(FC_E_LIMIT, "too many pairs");
  DevBuf dp, dh, dv, dr, dq, dord, deb;
  FC_TRY(dp.reserve((size_t)n * 2 * sizeof(int64_t)));
  FC_TRY(h2d(dp.p, pairs_host, (size_t)n * 2 * sizeof(int64_t)));
  FC_TRY(dh.reserve((size_t)n * sizeof(int64_t)));
  FC_TRY(dv.reserve((size_t)n * sizeof(int32_t)));
  FC_TRY(dr.reserve((size_t)n * sizeof(int32_t)));
  FC_TRY(dq.reserve((size_t)n * sizeof(int32_t)));
  FC_TRY(dord.reserve((size_t)n * sizeof(int32_t)));
  FC_TRY(deb.reserve(2 * sizeof(int64_t)));
  const int64_t eb[2] = {0, n};
  FC_TRY(h2d(deb.p, eb, sizeof eb));
  hipLaunchKernelGGL(k_pair_hashes, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, cur_stream(), dp.as<int64_t>(), n,
                     dh.as<int64_t>(), dv.as<int32_t>(), dr.as<int32_t>());
  FC_TRY(check_launch("k_pair_hashes"));
  Scratch scr;
  std::vector<int64_t> m(1, n);
  FC_TRY(pyset_orders_device(dv.as<int32_t>(), dr.as<int32_t>(), dh.as<int64_t>(), n, n, 1, m, deb.as<int64_t>(),
                             dq.as<int32_t>(), dord.as<int32_t>(), scr));
  std::vector<int32_t> ord((size_t)n);
  FC_TRY(d2h(ord.data(), dord.p, (size_t)n * sizeof(int32_t)));
  FC_TRY(sync());
  for (int64_t k = 0; k < n; ++k) order_out[k] = ord[(size_t)k];
  return FC_OK;
}

// fc_warmup(): the first launch from a translation unit makes the runtime load that unit's code object (milliseconds);
// a no-op launch moves that cost out of the first real call
__global__ void k_warm_tfd_gpu() {}
int warm_tfd_gpu() {
  hipLaunchKernelGGL(k_warm_tfd_gpu, dim3(1), dim3(64), 0, cur_stream());
  return check_launch("k_warm_tfd_gpu");
}

}  // namespace fc
