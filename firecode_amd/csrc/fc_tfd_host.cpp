// fc_tfd_host.cpp -- host half of prune_conformers_tfd
// (firecode/torsion_module.py:967-1043): the k-ladder bookkeeping replayed from
// the first-match array the GPU produces (k_tfd_first_match).
//
// The reference keeps, per chunk, a Python `set` of first-match pairs, turns it
// into a networkx Graph, and for every connected component keeps
// `tuple(g.subgraph(c).nodes)[0]`.  Which node that is depends on CPython's set
// iteration order (hash-slot order, not insertion order) at three places:
//   (1) the order in which `Graph(matches)` sees the edges  (set of 2-tuples),
//   (2) the order of the component set built by networkx's BFS  (set of ints),
//   (3) `show_nodes(...).nodes = set(...)`, iterated by the sub-graph view when
//       the component is less than half of the graph (networkx FilterAtlas).
// To return the reference's mask bit for bit this file re-implements exactly
// those semantics: CPython >= 3.8 open-addressing sets (setobject.c:
// LINEAR_PROBES = 9, PERTURB_SHIFT = 5, growth used*4 / used*2 above 50 000,
// rebuild in slot order) and the xxHash-style tuple hash (tupleobject.c).
// tests/test_pyset_emulation.py checks both against the running interpreter.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <new>
#include <system_error>

#include "fc_common.h"
#include "fc_tfd_core.h"

namespace fc {

// Helper threads of this file.  Two rules, both about std::terminate (= SIGABRT in the caller's process): a
// std::thread that is still joinable when it is destroyed ends the process -- so the threads live in a holder whose
// destructor joins, whatever way its scope is left -- and so does an exception that leaves a thread's function -- so
// every thread body catches everything and reports through a flag of its pool.
struct ThreadJoiner {
  std::vector<std::thread> threads;
  template <class F, class... Args>
  bool start(F &&f, Args &&...args) {  // false: the system has no thread to give (the caller's own thread does the work)
    try {
      threads.emplace_back(std::forward<F>(f), std::forward<Args>(args)...);
      return true;
    } catch (const std::system_error &) {
      return false;
    } catch (const std::bad_alloc &) {
      return false;
    }
  }
  void join_all() {
    for (auto &t : threads)
      if (t.joinable()) t.join();
    threads.clear();
  }
  ~ThreadJoiner() { join_all(); }
};

// ---- CPython set emulation ---------------------------------------------------
struct PySetEmu {
  // 16-byte entries (an empty slot has key == kEmpty; keys are non-negative), a spare table that
  // keeps its pages across growths, and prefetches of the home slot: the emulation of a set of
  // ~10^6 tuples is bound by DRAM latency (tables of 2^22 slots), not by arithmetic
  static constexpr int64_t kEmpty = -1;
  struct Entry {
    int64_t key = kEmpty;  // caller-defined id (int value, or index of a pair)
    int64_t hash = 0;
  };
  std::vector<Entry> table, spare;
  size_t mask = 7, fill = 0, used_n = 0;
  PySetEmu() : table(8) {}
  void reset() {  // back to an empty 8-slot set; keeps the allocation
    table.assign(8, Entry{});
    mask = 7;
    fill = used_n = 0;
  }
  void prefetch(int64_t hash) const { __builtin_prefetch(&table[(size_t)hash & mask], 1, 0); }

  static void insert_clean(std::vector<Entry> &t, size_t mask, int64_t key, int64_t hash) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      if (t[e].key == kEmpty) {
        t[e] = {key, hash};
        return;
      }
      if (i + 9 <= mask) {
        for (int j = 0; j < 9; ++j) {
          ++e;
          if (t[e].key == kEmpty) {
            t[e] = {key, hash};
            return;
          }
        }
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }

  void resize(size_t minused) {
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    spare.assign(newsize, Entry{});
    const size_t newmask = newsize - 1;
    constexpr size_t kAhead = 16;  // slots of the old table looked at in advance
    for (size_t s = 0; s <= mask; ++s) {
      if (s + kAhead <= mask && table[s + kAhead].key != kEmpty)
        __builtin_prefetch(&spare[(size_t)table[s + kAhead].hash & newmask], 1, 0);
      if (table[s].key != kEmpty) insert_clean(spare, newmask, table[s].key, table[s].hash);
    }
    table.swap(spare);
    mask = newmask;
    fill = used_n;
  }

  // eq(existing_key, new_key): key equality for entries with equal hash
  template <class Eq>
  bool add(int64_t key, int64_t hash, Eq eq) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      int probes = (i + 9 <= mask) ? 9 : 0;
      do {
        if (table[e].key == kEmpty) {
          table[e] = {key, hash};
          ++fill;
          ++used_n;
          if (fill * 5 >= mask * 3) resize(used_n > 50000 ? used_n * 2 : used_n * 4);
          return true;
        }
        if (table[e].hash == hash && eq(table[e].key, key)) return false;  // already present
        ++e;
      } while (probes--);
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }

  template <class Eq>
  bool contains(int64_t key, int64_t hash, Eq eq) const {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      int probes = (i + 9 <= mask) ? 9 : 0;
      do {
        if (table[e].key == kEmpty) return false;
        if (table[e].hash == hash && eq(table[e].key, key)) return true;
        ++e;
      } while (probes--);
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }

  template <class F>
  void for_each(F f) const {  // iteration order of `for x in the_set`
    for (size_t s = 0; s <= mask; ++s)
      if (table[s].key != kEmpty) f(table[s].key);
  }
};

// The same set for 2-tuples given by index, with 4-byte slots: slot = index of the tuple (-1 = empty), its hash
// read from the caller's array when a probe sequence or a rebuild needs it.  The iteration order of a set of 10^6
// tuples is a walk through a table of 2^22 slots at random: 16 MB of table stay in the last-level cache where the
// 64 MB of 16-byte entries did not.  UNIQUE: the caller guarantees distinct tuples (first-match edges: one per
// row), so an occupied slot is always just a collision -- CPython would compare hash and contents and move on.
struct PyTupleSetEmu {
  std::vector<int32_t> table, spare;
  size_t mask = 7, fill = 0, used_n = 0;
  const int64_t *hashes = nullptr;
  PyTupleSetEmu() : table(8, -1) {}
  void reset(const int64_t *h) {
    table.assign(8, -1);
    mask = 7;
    fill = used_n = 0;
    hashes = h;
  }
  void prefetch(int64_t hash) const { __builtin_prefetch(&table[(size_t)hash & mask], 1, 0); }
  static void insert_clean(std::vector<int32_t> &t, size_t mask, int32_t key, int64_t hash) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      if (t[e] < 0) {
        t[e] = key;
        return;
      }
      if (i + 9 <= mask) {
        for (int j = 0; j < 9; ++j) {
          ++e;
          if (t[e] < 0) {
            t[e] = key;
            return;
          }
        }
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }
  void resize(size_t minused) {
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    spare.assign(newsize, -1);
    const size_t newmask = newsize - 1;
    constexpr size_t kAhead = 16;
    for (size_t s = 0; s <= mask; ++s) {
      if (s + kAhead <= mask && table[s + kAhead] >= 0)
        __builtin_prefetch(&spare[(size_t)hashes[table[s + kAhead]] & newmask], 1, 0);
      if (table[s] >= 0) insert_clean(spare, newmask, table[s], hashes[table[s]]);
    }
    table.swap(spare);
    mask = newmask;
    fill = used_n;
  }
  // eq(existing_index, new_index): contents equal?  (never called when UNIQUE)
  template <bool UNIQUE, class Eq>
  bool add(int32_t key, Eq eq) {
    const int64_t hash = hashes[key];
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      int probes = (i + 9 <= mask) ? 9 : 0;
      do {
        if (table[e] < 0) {
          table[e] = key;
          ++fill;
          ++used_n;
          if (fill * 5 >= mask * 3) resize(used_n > 50000 ? used_n * 2 : used_n * 4);
          return true;
        }
        if (!UNIQUE && hashes[table[e]] == hash && eq((int64_t)table[e], (int64_t)key)) return false;
        ++e;
      } while (probes--);
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }
  template <class F>
  void for_each(F f) const {
    for (size_t s = 0; s <= mask; ++s)
      if (table[s] >= 0) f((int64_t)table[s]);
  }
};

static inline bool int_eq(int64_t a, int64_t b) { return a == b; }

// hash((a, b)) for non-negative Python ints a, b < 2^61 - 1  (hash(n) == n)
int64_t py_tuple2_hash(int64_t a, int64_t b) {
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  for (uint64_t lane : {(uint64_t)a, (uint64_t)b}) {
    acc += lane * P2;
    acc = (acc << 31) | (acc >> 33);
    acc *= P1;
  }
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  if (acc == (uint64_t)-1) return 1546275796;
  return (int64_t)acc;
}

// iteration order of a set built by inserting `keys` (non-negative ints) in order
void pyset_order_ints(const int64_t *keys, int64_t n, std::vector<int64_t> &out) {
  PySetEmu s;
  for (int64_t k = 0; k < n; ++k) s.add(keys[k], keys[k], int_eq);
  out.clear();
  s.for_each([&](int64_t key) { out.push_back(key); });
}

// iteration order (as indices into the input) of a set of 2-tuples inserted in order
void pyset_order_pairs(const int64_t *pairs, int64_t n, std::vector<int64_t> &out) {
  auto eq = [&](int64_t x, int64_t y) {
    return pairs[x * 2] == pairs[y * 2] && pairs[x * 2 + 1] == pairs[y * 2 + 1];
  };
  out.clear();
  if (n < (int64_t)1 << 31) {  // the 4-byte-slot form the chunk graphs use (here with the equality test)
    std::vector<int64_t> hashes((size_t)n);
    for (int64_t k = 0; k < n; ++k) hashes[(size_t)k] = py_tuple2_hash(pairs[k * 2], pairs[k * 2 + 1]);
    PyTupleSetEmu s;
    s.reset(hashes.data());
    for (int64_t k = 0; k < n; ++k) s.add<false>((int32_t)k, eq);
    s.for_each([&](int64_t key) { out.push_back(key); });
    return;
  }
  PySetEmu s;
  for (int64_t k = 0; k < n; ++k) s.add(k, py_tuple2_hash(pairs[k * 2], pairs[k * 2 + 1]), eq);
  s.for_each([&](int64_t key) { out.push_back(key); });
}

// helper threads of the component phase of a huge chunk and the smallest graph that gets them: set by
// tfd_ladder_from_first_match on entry (FC_TFD_COMP_THREADS, FC_TFD_COMP_PAR_MIN: test knobs), read by its workers
static unsigned g_comp_threads = 1;
static int64_t g_comp_par_min = 100000;

// ---- one chunk: matches (i_rel ascending) -> relative indices to reject -------
// Scratch that survives across the ~10^5 chunks of a fine ladder level.
struct ChunkScratch {
  PyTupleSetEmu edge_set;
  PySetEmu comp, view;
  std::vector<int64_t> order, nodes, members, level, next, adj_head, adj_next, adj_to, adj_tail, hashes;
  struct Slot {                    // relative index -> position in `nodes` (valid when the stamp matches): one 8-byte
    int32_t stamp = -1, pos = 0;   // slot per structure, i.e. one cache line per look-up instead of two
  };
  std::vector<Slot> slot;
  std::vector<char> seen;
  int64_t generation = 0;
};

// edges[k] = (i_rel, j_rel) in the order the reference adds them to `matches`.
static void chunk_rejects(const std::vector<int64_t> &edges, int64_t chunk_len, ChunkScratch &w,
                          std::vector<int64_t> &rejects) {
  const bool dbg = chunk_len > 200000 && getenv("FC_DEBUG");
  auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) { if (dbg) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[fc]   chunk %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
  rejects.clear();
  const int64_t m = (int64_t)edges.size() / 2;
  if (m == 0) return;
  // (1) Graph(matches): edges arrive in the iteration order of the set of tuples
  {
    const int64_t *pairs = edges.data();
    auto eq = [&](int64_t, int64_t) { return false; };  // the edges of a chunk are distinct: (i, first_match[i]), one per i
    constexpr int64_t kAhead = 12;
    w.hashes.resize((size_t)m);
    for (int64_t k = 0; k < m; ++k) w.hashes[(size_t)k] = py_tuple2_hash(pairs[k * 2], pairs[k * 2 + 1]);
    w.edge_set.reset(w.hashes.data());
    for (int64_t k = 0; k < m; ++k) {
      if (k + kAhead < m) w.edge_set.prefetch(w.hashes[(size_t)(k + kAhead)]);
      w.edge_set.add<true>((int32_t)k, eq);
    }
    w.order.clear();
    w.edge_set.for_each([&](int64_t key) { w.order.push_back(key); });
  }
  lap("edge set");
  if ((int64_t)w.slot.size() < chunk_len) w.slot.resize((size_t)chunk_len);
  if (w.generation >= 0x7ffffff0) {  // stamps are 32 bits: start over (once per 2*10^9 chunks of one scratch)
    for (auto &sl : w.slot) sl.stamp = -1;
    w.generation = 0;
  }
  const int32_t gen = (int32_t)++w.generation;
  w.nodes.clear();
  auto node_of = [&](int64_t v) {
    ChunkScratch::Slot &sl = w.slot[(size_t)v];
    if (sl.stamp == gen) return (int64_t)sl.pos;
    const int64_t p = (int64_t)w.nodes.size();
    sl.stamp = gen;
    sl.pos = (int32_t)p;
    w.nodes.push_back(v);
    return p;
  };
  // edges arrive in hash order, i.e. at random: two-stage prefetch (the edge, then its endpoints'
  // stamp / position slots) hides most of the DRAM latency of a 10^6-node chunk
  const int64_t n_ord = (int64_t)w.order.size();
  w.adj_to.resize((size_t)n_ord * 2);  // (pu, pv) per edge, in the order Graph() sees them
  for (int64_t q = 0; q < n_ord; ++q) {
    if (q + 16 < n_ord) __builtin_prefetch(&edges[(size_t)w.order[(size_t)(q + 16)] * 2], 0, 0);
    if (q + 8 < n_ord) {
      const int64_t e8 = w.order[(size_t)(q + 8)];
      const int64_t u8 = edges[(size_t)e8 * 2], v8 = edges[(size_t)e8 * 2 + 1];
      __builtin_prefetch(&w.slot[(size_t)u8], 1, 0);
      __builtin_prefetch(&w.slot[(size_t)v8], 1, 0);
    }
    const int64_t e = w.order[(size_t)q];
    w.adj_to[(size_t)q * 2] = node_of(edges[e * 2]);
    w.adj_to[(size_t)q * 2 + 1] = node_of(edges[e * 2 + 1]);
  }
  // neighbour lists in insertion order (what networkx's adjacency dicts iterate), as CSR:
  // matches are unique pairs, so there are no duplicate neighbours
  const int64_t n_nodes_csr = (int64_t)w.nodes.size();
  w.adj_head.assign((size_t)n_nodes_csr + 1, 0);  // start offsets
  for (int64_t q = 0; q < 2 * n_ord; ++q) ++w.adj_head[(size_t)w.adj_to[(size_t)q] + 1];
  for (int64_t v = 0; v < n_nodes_csr; ++v) w.adj_head[(size_t)v + 1] += w.adj_head[(size_t)v];
  w.adj_tail.assign(w.adj_head.begin(), w.adj_head.end() - 1);  // fill cursors
  w.adj_next.resize((size_t)n_ord * 2);                          // neighbour array
  for (int64_t q = 0; q < n_ord; ++q) {
    const int64_t pu = w.adj_to[(size_t)q * 2], pv = w.adj_to[(size_t)q * 2 + 1];
    w.adj_next[(size_t)w.adj_tail[(size_t)pu]++] = pv;
    w.adj_next[(size_t)w.adj_tail[(size_t)pv]++] = pu;
  }
  lap("adjacency");
  const int64_t n_nodes = (int64_t)w.nodes.size();
  w.seen.assign((size_t)n_nodes, 0);
  // One component: BFS from `src` (its earliest node in the graph's node order), the two Python sets, group[0],
  // rejects.  Touches only this component's nodes: components are independent of one another, and networkx's
  // early return of _plain_bfs (len(seen) == n) changes neither the content nor the insertion order of a set.
  struct CompScratch {
    PySetEmu comp, view;
    std::vector<int64_t> members, level, next;
  };
  auto one_component = [&](int64_t src, CompScratch &cs, std::vector<int64_t> &rej) {
    // (2) networkx _plain_bfs: `seen` is a Python set filled in BFS order
    cs.comp.reset();
    cs.members.clear();
    cs.comp.add(w.nodes[(size_t)src], w.nodes[(size_t)src], int_eq);
    cs.members.push_back(src);
    w.seen[(size_t)src] = 1;
    cs.level.assign(1, src);
    while (!cs.level.empty()) {
      cs.next.clear();
      const size_t n_level = cs.level.size();
      for (size_t li = 0; li < n_level; ++li) {
        const int64_t v = cs.level[li];
        if (li + 3 < n_level) {  // neighbours of a node three places ahead: their flags and ids
          const int64_t v3 = cs.level[li + 3];
          for (int64_t rec = w.adj_head[(size_t)v3]; rec < w.adj_head[(size_t)v3 + 1]; ++rec) {
            __builtin_prefetch(&w.seen[(size_t)w.adj_next[(size_t)rec]], 1, 0);
            __builtin_prefetch(&w.nodes[(size_t)w.adj_next[(size_t)rec]], 0, 0);
          }
        }
        for (int64_t rec = w.adj_head[(size_t)v]; rec < w.adj_head[(size_t)v + 1]; ++rec) {
          const int64_t x = w.adj_next[(size_t)rec];
          if (!w.seen[(size_t)x]) {
            w.seen[(size_t)x] = 1;
            __builtin_prefetch(&w.adj_head[(size_t)x], 0, 0);  // read when x is expanded, one level on
            cs.comp.add(w.nodes[(size_t)x], w.nodes[(size_t)x], int_eq);
            cs.members.push_back(x);
            cs.next.push_back(x);
          }
        }
      }
      cs.level.swap(cs.next);
    }
    // (3) group[0] of tuple(g.subgraph(c).nodes)
    int64_t first;
    if (2 * (int64_t)cs.members.size() < n_nodes) {
      cs.view.reset();  // show_nodes.nodes = set(nbunch_iter(c)): rebuilt in c's iteration order
      cs.comp.for_each([&](int64_t key) { cs.view.add(key, key, int_eq); });
      first = -1;
      cs.view.for_each([&](int64_t key) {
        if (first < 0) first = key;
      });
    } else {
      first = w.nodes[(size_t)src];  // atlas order: the BFS source is the component's earliest node
    }
    for (int64_t p : cs.members)
      if (w.nodes[(size_t)p] != first) rej.push_back(w.nodes[(size_t)p]);
  };
  const unsigned comp_threads = g_comp_threads;
  const int64_t comp_par_min = g_comp_par_min;
  if (comp_threads > 1 && n_nodes >= comp_par_min) {
    // A huge chunk (the 840 000-structure chunk of k = 2 at 1.7 M structures is the critical path of the whole
    // ladder): the sources first -- each component's earliest node = the root of a union-find that always
    // hangs the later root under the earlier one -- then the components dealt to helper threads.
    std::vector<int32_t> parent((size_t)n_nodes);
    for (int64_t v = 0; v < n_nodes; ++v) parent[(size_t)v] = (int32_t)v;
    auto find = [&](int32_t v) {
      while (parent[(size_t)v] != v) {
        parent[(size_t)v] = parent[(size_t)parent[(size_t)v]];  // path halving
        v = parent[(size_t)v];
      }
      return v;
    };
    for (int64_t q = 0; q < n_ord; ++q) {
      const int32_t ra = find((int32_t)w.adj_to[(size_t)q * 2]), rb = find((int32_t)w.adj_to[(size_t)q * 2 + 1]);
      if (ra < rb) parent[(size_t)rb] = ra;
      else if (rb < ra) parent[(size_t)ra] = rb;
    }
    std::vector<int32_t> sources;
    for (int64_t v = 0; v < n_nodes; ++v)
      if (parent[(size_t)v] == (int32_t)v) sources.push_back((int32_t)v);
    lap("union-find");
    std::atomic<size_t> next_src{0};
    std::atomic<bool> failed{false};
    std::vector<std::vector<int64_t>> rej_t(comp_threads);
    auto helper = [&](unsigned t) {
      try {
        CompScratch cs;
        constexpr size_t kBatch = 256;  // sources per grab
        while (!failed.load(std::memory_order_relaxed)) {
          const size_t b0 = next_src.fetch_add(kBatch);
          if (b0 >= sources.size()) break;
          const size_t b1 = std::min(sources.size(), b0 + kBatch);
          for (size_t i = b0; i < b1; ++i) one_component((int64_t)sources[i], cs, rej_t[t]);
        }
      } catch (...) {
        failed = true;
      }
    };
    {
      ThreadJoiner pool;
      for (unsigned t = 1; t < comp_threads; ++t)
        if (!pool.start(helper, t)) break;
      helper(0);
    }
    if (failed.load()) throw std::bad_alloc();
    for (auto &r : rej_t) rejects.insert(rejects.end(), r.begin(), r.end());
  } else {
    CompScratch cs;
    for (int64_t src = 0; src < n_nodes; ++src)
      if (!w.seen[(size_t)src]) one_component(src, cs, rejects);
  }
  lap("components");
}

// chunks [step_begin, step_end) of one ladder level; chunks are independent.  The rejects go to `out`
// as absolute indices (the caller applies them when -- and if -- the level runs).
static void level_chunks(const int64_t *fm, int64_t N, int64_t k, int64_t d, int64_t num_active,
                         int64_t step_begin, int64_t step_end, ChunkScratch &scratch, std::vector<int64_t> &out) {
  std::vector<int64_t> edges, rejects;
  for (int64_t step = step_begin; step < step_end; ++step) {
    const int64_t lo = d * step;
    // torsion_module.py:987-990: the LAST chunk ends at num_active_str, not at N
    const int64_t len = (step == k - 1) ? (num_active - lo) : (d * (step + 1) - lo);
    if (len <= 1) continue;
    if (lo >= N) break;
    edges.clear();
    const int64_t hi = lo + len;  // exclusive; <= N because num_active <= N
    for (int64_t i = lo; i < hi && i < N; ++i) {
      const int64_t j = fm[i];
      if (j >= 0 && j < hi) {
        edges.push_back(i - lo);
        edges.push_back(j - lo);
      }
    }
    if (edges.empty()) continue;
    chunk_rejects(edges, len, scratch, rejects);
    for (int64_t r : rejects) out.push_back(r + lo);
  }
}

// ---- the levels in order ---------------------------------------------------------------------------------------------
// level_flags[li] (may be nullptr / missing): one byte per structure of [0, d (k - 1)), 1 = a non-last chunk of level li
// rejects it -- worked out beforehand from first_match alone (here on host threads, or on the device: fc_tfd_ladder.hip).
// first_last_flags: the same for the LAST chunk of level first_level, [d (k - 1), N), known up front because the active
// count is N there.  Every other last chunk ends at the active count of its moment (torsion_module.py:987-990) and is
// done here, on the spot.  first_level == -2: mask_out already holds the mask after every level but k = 1; only that
// level is applied.
static const double kLadderK[] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};
constexpr int kLadderLevels = (int)(sizeof(kLadderK) / sizeof(kLadderK[0]));

int tfd_apply_levels_host(const int64_t *fm, int64_t N, const std::vector<const uint8_t *> &level_flags, int first_level,
                          const uint8_t *first_last_flags, uint8_t *mask_out, int64_t active_known) {
  const bool debug = getenv("FC_DEBUG") != nullptr;
  ChunkScratch scratch;
  std::vector<int64_t> last;
  int64_t num_active = N;  // kept up to date as flags go 1 -> 0
  if (first_level == -2) {
    num_active = active_known;  // (the device counted while it applied the levels)
    if (num_active < 0) {
      num_active = 0;
      for (int64_t i = 0; i < N; ++i) num_active += mask_out[i];
    }
  } else {
    std::memset(mask_out, 1, (size_t)N);
  }
  auto reject = [&](int64_t r) {
    num_active -= mask_out[r];
    mask_out[r] = 0;
  };
  for (int li = 0; li < kLadderLevels; ++li) {
    const int64_t k = (int64_t)kLadderK[li];
    if (first_level == -2 && k != 1) continue;
    const bool runs = (k == 1 || 5 * k < num_active);
    if (!runs) continue;
    const int64_t d = N / k;
    // the last chunk reads nothing but first_match either, so the order of application within
    // the level does not matter: all rejects of a level come from the mask-independent chunk graphs
    last.clear();
    const int64_t active_in = num_active;
    const bool have_first_last = li == first_level && active_in == N && first_last_flags != nullptr;
    if (!have_first_last) level_chunks(fm, N, k, d, active_in, k - 1, k, scratch, last);
    const uint8_t *flags = (size_t)li < level_flags.size() ? level_flags[(size_t)li] : nullptr;
    if (flags != nullptr && k > 1) {  // branch-free over the bytes (vectorised by the compiler): a level rejects up to half of its range
      const size_t n_flags = (size_t)(d * (k - 1));
      unsigned long long gone = 0;
      size_t r = 0;
      for (; r + 32 <= n_flags; r += 32) {  // 32 structures per step; most steps of most levels reject nothing new
        uint64_t m8[4], f8[4];
        std::memcpy(m8, mask_out + r, 32);
        std::memcpy(f8, flags + r, 32);
        const uint64_t h0 = m8[0] & f8[0], h1 = m8[1] & f8[1], h2 = m8[2] & f8[2], h3 = m8[3] & f8[3];
        if ((h0 | h1) | (h2 | h3)) {
          gone += (unsigned long long)(__builtin_popcountll(h0) + __builtin_popcountll(h1) + __builtin_popcountll(h2) +
                                       __builtin_popcountll(h3));
          m8[0] ^= h0, m8[1] ^= h1, m8[2] ^= h2, m8[3] ^= h3;
          std::memcpy(mask_out + r, m8, 32);
        }
      }
      for (; r < n_flags; ++r) {
        const uint8_t hit = (uint8_t)(mask_out[r] & flags[r]);
        gone += hit;
        mask_out[r] = (uint8_t)(mask_out[r] ^ hit);
      }
      num_active -= (int64_t)gone;
    }
    if (have_first_last) {
      const int64_t lo = d * (k - 1);
      for (int64_t r = lo; r < N; ++r)
        if (first_last_flags[r - lo]) reject(r);
    }
    for (int64_t r : last) reject(r);
    if (debug) fprintf(stderr, "[fc] tfd ladder k=%lld: %lld active in\n", (long long)k, (long long)active_in);
  }
  return FC_OK;
}

// The whole ladder on the host: first_match[i] = min{j > i : similar(i, j)} or -1.
//
// What a chunk rejects depends on first_match alone: the reference's inner loops never look at
// final_mask (masked structures keep taking part, torsion_module.py:997-1018).  Only two things depend
// on the levels before: whether a level runs at all (5 k < num_active_str, :976) and the length of
// its LAST chunk (:987-990).  So every non-last chunk of every level that can possibly run is
// worked out up front, all levels at once, on the host threads (largest chunks first), then the levels
// are walked in order (tfd_apply_levels_host).  This is the reference the device ladder (fc_tfd_ladder.hip) is
// tested against, what runs without a device and below kDeviceLadderMin structures, and the fallback when a device
// capacity is exceeded.
static int tfd_ladder_impl(const int64_t *fm, int64_t N, uint8_t *mask_out) {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (hw > 16) hw = 16;
  if (const char *v = getenv("FC_TFD_THREADS")) {  // test knob
    const long t = std::strtol(v, nullptr, 10);
    if (t >= 1 && t <= 64) hw = (unsigned)t;
  }
  g_comp_threads = std::min(hw, 8u);
  if (const char *v = getenv("FC_TFD_COMP_THREADS")) {  // 1: always the sequential walk over the components
    const long t = std::strtol(v, nullptr, 10);
    if (t >= 1 && t <= 64) g_comp_threads = (unsigned)t;
  }
  g_comp_par_min = 100000;
  if (const char *v = getenv("FC_TFD_COMP_PAR_MIN")) g_comp_par_min = (int64_t)std::strtoll(v, nullptr, 10);
  const bool debug = getenv("FC_DEBUG") != nullptr;
  const auto t_all = std::chrono::steady_clock::now();
  // What a task rejects goes into ONE byte per structure and level (chunks of a level are disjoint index
  // ranges, so workers never write the same byte).
  struct Task {
    int level;  // index into kLadderK
    int64_t k, d, step_begin, step_end, cost;
  };
  std::vector<Task> tasks;
  static std::vector<uint8_t> level_rej[kLadderLevels];  // kept across calls (the ladder holds its own lock): fresh pages cost ~7 ms at 1.7 M
  static std::vector<uint8_t> first_last_flags;
  std::vector<const uint8_t *> level_flags((size_t)kLadderLevels, nullptr);
  int first_level = -1;
  for (int li = 0; li < kLadderLevels; ++li) {
    const int64_t k = (int64_t)kLadderK[li];
    if (!(k == 1 || 5 * k < N)) continue;  // num_active <= N: the level can never run
    if (first_level < 0) first_level = li;
    if (k == 1) continue;  // its only chunk is a last chunk
    const int64_t d = N / k;
    if (d <= 1) continue;
    level_rej[li].assign((size_t)(d * (k - 1)), 0);  // the non-last chunks cover [0, d (k - 1))
    level_flags[(size_t)li] = level_rej[li].data();
    // non-last chunks [0, k - 1), cut into tasks of about 2^17 structures (a huge chunk is a task of its own)
    const int64_t per = std::max<int64_t>(1, (int64_t)(131072 / d));
    for (int64_t b = 0; b < k - 1; b += per) {
      const int64_t e = std::min<int64_t>(k - 1, b + per);
      tasks.push_back(Task{li, k, d, b, e, (e - b) * d});
    }
  }
  // the LAST chunk of the first level that can run is known up front as well (num_active = N there)
  std::vector<int64_t> first_last;
  bool have_first_last = false;
  if (first_level >= 0 && (int64_t)kLadderK[first_level] > 1) {
    const int64_t k = (int64_t)kLadderK[first_level], d = N / k;
    tasks.push_back(Task{first_level, k, d, -1, -1, N - d * (k - 1)});
    have_first_last = true;
  }
  std::vector<size_t> order(tasks.size());
  for (size_t t = 0; t < order.size(); ++t) order[t] = t;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tasks[a].cost > tasks[b].cost; });
  {
    std::atomic<size_t> next{0};
    std::atomic<bool> host_failed{false};  // a helper ran out of host memory (the only thing its body can throw)
    auto worker = [&]() {
      try {
        ChunkScratch scratch;
        std::vector<int64_t> rej;
        while (!host_failed.load(std::memory_order_relaxed)) {
          const size_t q = next.fetch_add(1);
          if (q >= order.size()) break;
          const Task &t = tasks[order[q]];
          rej.clear();
          if (t.step_begin < 0) {  // the first level's last chunk (one task: no other writes first_last)
            level_chunks(fm, N, t.k, t.d, N, t.k - 1, t.k, scratch, first_last);
            continue;
          }
          // a non-last chunk never uses num_active: pass N
          level_chunks(fm, N, t.k, t.d, N, t.step_begin, t.step_end, scratch, rej);
          uint8_t *flags = level_rej[t.level].data();
          for (int64_t r : rej) flags[r] = 1;
        }
      } catch (...) {
        host_failed = true;
      }
    };
    const unsigned nthreads = (N >= 2000 && tasks.size() > 1) ? (unsigned)std::min<size_t>(hw, tasks.size()) : 1;
    // (from here to the end of this block helper threads are running: the holder joins in its destructor, so no
    // return, FC_TRY or exception below can leave a joinable std::thread behind -- that would be std::terminate)
    ThreadJoiner pool;
    for (unsigned t = 1; t < nthreads; ++t)
      if (!pool.start(worker)) break;
    worker();
    pool.join_all();
    if (host_failed.load()) return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder's host threads");
  }
  if (debug)
    fprintf(stderr, "[fc] tfd ladder (host): %zu speculative tasks on %u threads, %.1f ms\n", tasks.size(), hw,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
  const uint8_t *flf = nullptr;
  if (have_first_last) {
    const int64_t k = (int64_t)kLadderK[first_level], lo = (N / k) * (k - 1);
    first_last_flags.assign((size_t)(N - lo), 0);
    for (int64_t r : first_last) first_last_flags[(size_t)(r - lo)] = 1;
    flf = first_last_flags.data();
  }
  FC_TRY(tfd_apply_levels_host(fm, N, level_flags, first_level, flf, mask_out, -1));
  if (debug)
    fprintf(stderr, "[fc] tfd ladder (host) total %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
  return FC_OK;
}

// The ladder keeps process-wide state across calls (the per-level flag arrays): one ladder at a time, under a lock of its
// own -- not merely "the callers hold the API lock".  No exception crosses the C ABI: whatever the host side throws
// (std::bad_alloc is the only candidate) becomes an error code.
static std::mutex &ladder_mutex() {
  static std::mutex mu;
  return mu;
}
int tfd_ladder_host_only(const int64_t *fm, int64_t N, uint8_t *mask_out) {
  std::lock_guard<std::mutex> lock(ladder_mutex());
  try {
    return tfd_ladder_impl(fm, N, mask_out);
  } catch (const std::bad_alloc &) {
    return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder (N = %lld)", (long long)N);
  } catch (const std::exception &e) {
    return set_error(FC_E_HIP, "TFD ladder: %s", e.what());
  }
}

int tfd_ladder_device(const int64_t *fm_dev, const int64_t *fm_host, int64_t N, uint8_t *mask_out);  // fc_tfd_ladder.hip
constexpr int64_t kDeviceLadderMin = 20000;  // below: a few hundred microseconds of host work, less than the device's launches

// fm_dev (may be nullptr): the same array on the device -- the ladder then runs there (FC_TFD_GPU=0: on the host anyway)
int tfd_ladder_from_first_match(const int64_t *fm, int64_t N, uint8_t *mask_out, const int64_t *fm_dev) {
  bool use_gpu = fm_dev != nullptr && N >= kDeviceLadderMin;
  if (const char *v = getenv("FC_TFD_GPU")) use_gpu = use_gpu && atoi(v) != 0;
  if (!use_gpu) return tfd_ladder_host_only(fm, N, mask_out);
  try {
    return tfd_ladder_device(fm_dev, fm, N, mask_out);
  } catch (const std::bad_alloc &) {
    return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder (N = %lld)", (long long)N);
  } catch (const std::exception &e) {
    return set_error(FC_E_HIP, "TFD ladder: %s", e.what());
  }
}

// the same for a caller that holds the array on the device only (the csearch pipeline): nothing is copied down that
// the ladder does not need
int tfd_ladder_from_device(const int64_t *fm_dev, int64_t N, uint8_t *mask_out) {
  bool use_gpu = N >= kDeviceLadderMin;
  if (const char *v = getenv("FC_TFD_GPU")) use_gpu = use_gpu && atoi(v) != 0;
  try {
    if (use_gpu) return tfd_ladder_device(fm_dev, nullptr, N, mask_out);
    std::vector<int64_t> fm((size_t)N);
    FC_TRY(d2h(fm.data(), fm_dev, (size_t)N * sizeof(int64_t)));
    FC_TRY(sync());
    return tfd_ladder_host_only(fm.data(), N, mask_out);
  } catch (const std::bad_alloc &) {
    return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder (N = %lld)", (long long)N);
  } catch (const std::exception &e) {
    return set_error(FC_E_HIP, "TFD ladder: %s", e.what());
  }
}

// ---- a component of any size on the host (the device leaves those above tfd::kGroupCompMax nodes) ---------------------------
// mx / mp / ms: relative index, first match (the member itself when it has none in the chunk), slot of the edge in the
// chunk's tuple set.  Same orders as chunk_rejects' one_component.
uint32_t host_component_first_big(const uint32_t *mx, const uint32_t *mp, const uint32_t *ms, int64_t n, uint32_t n_graph) {
  std::vector<int64_t> by_x((size_t)n);
  for (int64_t k = 0; k < n; ++k) by_x[(size_t)k] = k;
  std::sort(by_x.begin(), by_x.end(), [&](int64_t a, int64_t b) { return mx[a] < mx[b]; });
  auto local_of = [&](uint32_t x) {
    int64_t lo = 0, hi = n;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (mx[by_x[(size_t)mid]] <= x) lo = mid;
      else hi = mid;
    }
    return by_x[(size_t)lo];
  };
  int64_t src = 0;
  uint32_t best = 0xFFFFFFFFu;
  std::vector<std::vector<std::pair<uint32_t, int64_t>>> adj((size_t)n);  // (edge slot, neighbour)
  for (int64_t k = 0; k < n; ++k)
    if (mp[k] != mx[k]) {
      if (ms[k] < best) best = ms[k], src = k;
      const int64_t p = local_of(mp[k]);
      adj[(size_t)k].push_back({ms[k], p});
      adj[(size_t)p].push_back({ms[k], k});
    }
  if (2 * (uint64_t)n >= (uint64_t)n_graph) return mx[src];
  for (auto &a : adj) std::sort(a.begin(), a.end());
  PySetEmu comp, view;
  std::vector<char> seen((size_t)n, 0);
  std::vector<int64_t> level(1, src), next;
  seen[(size_t)src] = 1;
  comp.add(mx[src], mx[src], int_eq);
  while (!level.empty()) {
    next.clear();
    for (const int64_t v : level)
      for (const auto &e : adj[(size_t)v])
        if (!seen[(size_t)e.second]) {
          seen[(size_t)e.second] = 1;
          comp.add(mx[e.second], mx[e.second], int_eq);
          next.push_back(e.second);
        }
    level.swap(next);
  }
  comp.for_each([&](int64_t key) { view.add(key, key, int_eq); });
  int64_t first = -1;
  view.for_each([&](int64_t key) {
    if (first < 0) first = key;
  });
  return (uint32_t)first;
}

// ---- the device ladder's routines on the CPU (host group): what tests/test_tfd_ladder_v2.py compares with the ladder above -------
// Chunks of at most kChunkMax structures go through chunk_front exactly as a wavefront runs it; larger ones get their
// tuple-set slots from PyTupleSetEmu (the device's staged insertion is tested against it on its own) and their
// components through the same tiny_first / comp_group_first / host_component_first_big the device's records reach.
int tfd_ladder_emulate_device(const int64_t *fm, int64_t N, uint8_t *mask_out) {
  using namespace tfd;
  struct E { int li; int64_t k, lo, d, nch; };
  std::vector<E> es;
  int first_li = -1;
  for (int li = 0; li < kLadderLevels; ++li) {
    const int64_t k = (int64_t)kLadderK[li];
    if (k == 1 || !(5 * k < N)) continue;
    const int64_t d = N / k;
    if (d <= 1) continue;
    if (first_li < 0) {
      first_li = li;
      if (N - d * (k - 1) >= 2) es.push_back(E{li, k, d * (k - 1), N - d * (k - 1), 1});
    }
    es.push_back(E{li, k, 0, d, k - 1});
  }
  std::vector<std::vector<uint8_t>> eflags(es.size());
  HostGroup g;
  std::vector<uint8_t> local, clocal, tiny_scr((size_t)kTinyScratch);
  std::vector<CompRecord> recs;
  std::vector<uint32_t> mx, mp, ms;
  auto solve_record = [&](const CompRecord &R, const uint32_t *x, const uint32_t *p, const uint32_t *s, uint8_t *flags) {
    uint32_t first;
    if (R.n <= (uint32_t)kTinyMax) {
      struct Acc {
        const uint32_t *a, *b, *c;
        uint32_t x(int k) const { return a[k]; }
        uint32_t par(int k) const { return b[k]; }
        uint32_t slot(int k) const { return c[k]; }
      };
      first = tiny_first(Acc{x, p, s}, (int)R.n, R.n_graph, tiny_scr.data());
    } else if (R.n <= (uint32_t)kGroupCompMax) {
      const size_t cap = ((size_t)R.n + 3) & ~(size_t)3, tbl = (size_t)pyset_final_mask(R.n) + 1;
      uint32_t cap2 = 64;
      while (cap2 < 2u * R.n) cap2 <<= 1;
      clocal.assign(comp_local_bytes(cap, tbl, cap2), 0);
      CompLocal L;
      comp_local_carve(clocal.data(), cap, tbl, cap2, L);
      first = comp_group_first(g, L, x, p, s, (int)R.n, R.n_graph, cap2);
    } else {
      first = host_component_first_big(x, p, s, R.n, R.n_graph);
    }
    for (uint32_t k = 0; k < R.n; ++k)
      if (x[k] != first) flags[x[k]] = 1;
  };
  PyTupleSetEmu edge_set;
  for (size_t q = 0; q < es.size(); ++q) {
    const E &e = es[q];
    eflags[q].assign((size_t)(e.d * e.nch), 0);
    for (int64_t c = 0; c < e.nch; ++c) {
      const int64_t lo = e.lo + c * e.d;
      uint8_t *flags = eflags[q].data() + c * e.d;
      if (e.d <= kChunkMax) {
        const int d = (int)e.d;
        int tbl = (int)pyset_final_mask(d - 1) + 1;
        local.assign(chunk_local_bytes(d, tbl, 1), 0);
        ChunkLocal L;
        chunk_local_carve(local.data(), d, tbl, L);
        recs.assign((size_t)d / 16 + 2, CompRecord{0, 0, 0, 0});
        mx.assign((size_t)d, 0), mp.assign((size_t)d, 0), ms.assign((size_t)d, 0);
        const ChunkExport ex{recs.data(), mx.data(), mp.data(), ms.data(), 0};
        const int nexp = chunk_front(g, L, fm, lo, d, 0u, flags, ex);
        for (int b = 0; b < nexp; ++b) {
          const CompRecord &R = recs[(size_t)b];
          solve_record(R, mx.data() + R.moff, mp.data() + R.moff, ms.data() + R.moff, flags);
        }
        continue;
      }
      // a large chunk: slots from the reference emulation of the tuple set, roots by union-find
      const int64_t d = e.d;
      std::vector<int64_t> hashes;
      std::vector<int32_t> edge_x;
      std::vector<int64_t> par((size_t)d);
      for (int64_t x = 0; x < d; ++x) {
        const int64_t j = fm[lo + x];
        par[(size_t)x] = (j >= 0 && j < lo + d) ? j - lo : x;
        if (par[(size_t)x] != x) {
          edge_x.push_back((int32_t)x);
          hashes.push_back(py_tuple2_hash(x, par[(size_t)x]));
        }
      }
      if (edge_x.empty()) continue;
      edge_set.reset(hashes.data());
      auto never = [](int64_t, int64_t) { return false; };
      for (size_t k = 0; k < edge_x.size(); ++k) edge_set.add<true>((int32_t)k, never);
      std::vector<uint32_t> slot((size_t)d, kNone);
      for (size_t s = 0; s <= edge_set.mask; ++s)
        if (edge_set.table[s] >= 0) slot[(size_t)edge_x[(size_t)edge_set.table[s]]] = (uint32_t)s;
      std::vector<int64_t> root(par);
      for (int64_t x = d - 1; x >= 0; --x) root[(size_t)x] = par[(size_t)x] == x ? x : root[(size_t)par[(size_t)x]];  // parents come later
      std::vector<char> gnode((size_t)d, 0);
      for (int32_t x : edge_x) gnode[(size_t)x] = 1, gnode[(size_t)par[(size_t)x]] = 1;
      std::vector<uint32_t> size((size_t)d, 0), off((size_t)d + 1, 0), fill((size_t)d, 0);
      uint32_t ng = 0;
      for (int64_t x = 0; x < d; ++x)
        if (gnode[(size_t)x]) ++ng, ++size[(size_t)root[(size_t)x]];
      for (int64_t x = 0; x < d; ++x) off[(size_t)x + 1] = off[(size_t)x] + size[(size_t)x];
      mx.assign((size_t)ng, 0), mp.assign((size_t)ng, 0), ms.assign((size_t)ng, 0);
      for (int64_t x = d - 1; x >= 0; --x)  // (any order: the device fills through atomics)
        if (gnode[(size_t)x]) {
          const int64_t r = root[(size_t)x];
          const uint32_t at = off[(size_t)r] + fill[(size_t)r]++;
          mx[at] = (uint32_t)x, mp[at] = (uint32_t)par[(size_t)x], ms[at] = slot[(size_t)x];
        }
      for (int64_t r = 0; r < d; ++r)
        if (size[(size_t)r] >= 2) {
          const CompRecord R{0, off[(size_t)r], size[(size_t)r], ng};
          solve_record(R, mx.data() + R.moff, mp.data() + R.moff, ms.data() + R.moff, flags);
        }
    }
  }
  std::vector<const uint8_t *> lf((size_t)kLadderLevels, nullptr);
  const uint8_t *first_last = nullptr;
  for (size_t q = 0; q < es.size(); ++q) {
    if (es[q].lo == 0) lf[(size_t)es[q].li] = eflags[q].data();
    else first_last = eflags[q].data();
  }
  return tfd_apply_levels_host(fm, N, lf, first_li, first_last, mask_out, -1);
}

}  // namespace fc
