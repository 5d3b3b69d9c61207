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

// ---- component phase of a level whose chunk graphs came from the device (fc_tfd_gpu.hip) --------------------------
// The device delivers the nodes in component-major order: component = one contiguous block led by its earliest node
// (graph order), so most components need neither a search nor a set:
//   * a component of at least half the chunk's graph keeps its earliest node (the atlas branch of networkx's FilterAtlas);
//   * otherwise group[0] is the first element of a Python set of the members' relative indices -- small non-negative
//     ints hash to themselves, so when the members' residues modulo the set's final table size are all different every
//     one sits in its home slot whatever the insertion order, and the first one is the smallest residue;
//   * only components with two members in one home slot need the reference's insertion orders: breadth-first search
//     over the insertion-ordered neighbour lists and the two set emulations, as in one_component above.
// flags[rel] = 1 for every rejected structure of the chunk.
static inline int64_t pyset_final_size(int64_t n_keys) {  // slots of a set that received n_keys distinct keys one by one
  uint64_t mask = 7;
  for (;;) {
    const int64_t trigger = (int64_t)((mask * 3 + 4) / 5);
    if (trigger > n_keys) return (int64_t)mask + 1;
    const uint64_t minused = trigger > 50000 ? 2 * (uint64_t)trigger : 4 * (uint64_t)trigger;
    uint64_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
    if (trigger == n_keys) return (int64_t)mask + 1;
  }
}

struct GraphCompScratch {
  PySetEmu comp, view;
  std::vector<int32_t> members, level, next;
  std::vector<char> seen;
  std::vector<uint64_t> bits;
  int64_t n_search = 0;
};

static void graph_component(const TfdLevelGraph &g, int64_t s0, int64_t s1, int64_t chunk_nodes, GraphCompScratch &cs,
                            uint8_t *flags) {
  const int32_t *nodes = g.nodes.data();
  const int64_t size = s1 - s0;
  int64_t first = -1;
  if (2 * size >= chunk_nodes) {
    first = nodes[s0];
  } else {
    const int64_t T = pyset_final_size(size);
    const size_t words = (size_t)((T + 63) >> 6);
    if (cs.bits.size() < words) cs.bits.resize(words);
    std::fill(cs.bits.begin(), cs.bits.begin() + (std::ptrdiff_t)words, 0ull);
    int64_t best = T;
    bool clean = true;
    for (int64_t v = s0; v < s1; ++v) {
      const int64_t r = (int64_t)nodes[v] & (T - 1);
      uint64_t &w = cs.bits[(size_t)(r >> 6)];
      const uint64_t bit = 1ull << (r & 63);
      if (w & bit) {
        clean = false;
        break;
      }
      w |= bit;
      if (r < best) {
        best = r;
        first = nodes[v];
      }
    }
    if (!clean) {
      // the reference's orders: _plain_bfs from the earliest node over the insertion-ordered neighbour lists ...
      ++cs.n_search;
      const int32_t *head = g.adj_head.data(), *adj = g.adj_next.data();
      cs.seen.assign((size_t)size, 0);
      cs.comp.reset();
      cs.comp.add(nodes[s0], nodes[s0], int_eq);
      cs.seen[0] = 1;
      cs.level.assign(1, (int32_t)s0);
      while (!cs.level.empty()) {
        cs.next.clear();
        for (const int32_t v : cs.level)
          for (int32_t rec = head[v]; rec < head[v + 1]; ++rec) {
            const int32_t x = adj[rec];
            if (!cs.seen[(size_t)(x - s0)]) {
              cs.seen[(size_t)(x - s0)] = 1;
              cs.comp.add(nodes[x], nodes[x], int_eq);
              cs.next.push_back(x);
            }
          }
        cs.level.swap(cs.next);
      }
      // ... then show_nodes.nodes = set(nbunch_iter(c)), iterated: its first element
      cs.view.reset();
      cs.comp.for_each([&](int64_t key) { cs.view.add(key, key, int_eq); });
      first = -1;
      cs.view.for_each([&](int64_t key) {
        if (first < 0) first = key;
      });
    }
  }
  for (int64_t v = s0; v < s1; ++v)
    if (nodes[v] != first) flags[nodes[v]] = 1;
}

// all non-last chunks of one level from the device-built graphs -> level_flags[absolute index] = 1 for rejects
// only_left: just the components the device left over (g.left), one job each
static void level_rejects_from_graph(const TfdLevelGraph &g, unsigned threads, uint8_t *level_flags, int64_t *n_search_out,
                                     bool only_left = false) {
  if (g.nodes.empty()) return;
  struct Job { int chunk; int64_t j0, j1; };
  std::vector<Job> jobs;
  if (only_left) {
    for (const int32_t j : g.left) {
      const int64_t s0 = g.sources[(size_t)j];
      const int c = !g.left_chunk.empty() ? (int)g.left_chunk[(size_t)j]  // (compact arrays: the device says which chunk)
                                          : (int)(std::upper_bound(g.nbase.begin(), g.nbase.end(), s0) - g.nbase.begin()) - 1;
      jobs.push_back(Job{c, j, (int64_t)j + 1});
    }
  } else {
    for (int c = 0; c < g.n_chunks; ++c) {
      const int64_t j0 = g.sbase[(size_t)c], j1 = g.sbase[(size_t)c + 1];
      for (int64_t b = j0; b < j1; b += 2048) jobs.push_back(Job{c, b, std::min(j1, b + 2048)});
    }
  }
  std::atomic<size_t> next{0};
  std::atomic<int64_t> searched{0};
  std::atomic<bool> failed{false};
  auto worker = [&]() {
    try {  // (a thread body must not leak an exception: std::terminate)
      GraphCompScratch cs;
      while (!failed.load(std::memory_order_relaxed)) {
        const size_t q = next.fetch_add(1);
        if (q >= jobs.size()) break;
        const Job &jb = jobs[q];
        const int64_t chunk_nodes = g.nbase[(size_t)jb.chunk + 1] - g.nbase[(size_t)jb.chunk];
        uint8_t *flags = level_flags + (int64_t)jb.chunk * g.d;
        for (int64_t j = jb.j0; j < jb.j1; ++j)
          graph_component(g, g.sources[(size_t)j], g.sources[(size_t)j + 1], chunk_nodes, cs, flags);
      }
      searched += cs.n_search;
    } catch (...) {
      failed = true;
    }
  };
  {
    ThreadJoiner pool;  // joins on every way out of this scope
    if (threads > 1 && jobs.size() > 1)
      for (unsigned t = 1; t < threads; ++t)
        if (!pool.start(worker)) break;  // (no more threads to be had: the ones that exist share the jobs)
    worker();
  }
  if (failed.load()) throw std::bad_alloc();  // out of host memory in a helper: the ladder reports FC_E_NOMEM
  if (n_search_out) *n_search_out = searched.load();
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

// Streams (and graph holders) of the helper threads that run the coarse levels side by side; created on first use or by
// fc_warmup (a stream is a hardware queue: several ms each), dropped with the context they were created in.
constexpr int kLevelStreamsMax = 8;
static hipStream_t g_lvl_stream[kLevelStreamsMax] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static hipEvent_t g_lvl_begin = nullptr;
static uint64_t g_lvl_epoch = 0;
static TfdLevelGraph g_lvl_holders[kLevelStreamsMax];
void tfd_level_streams_teardown() {  // context_teardown: the streams belong to the device being left
  for (auto &st : g_lvl_stream) {
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
    st = nullptr;
  }
  if (g_lvl_begin) (void)hipEventDestroy(g_lvl_begin);
  g_lvl_begin = nullptr;
}
int tfd_level_streams(int n) {
  if (g_lvl_epoch != ctx().epoch) {  // (a new context: context_teardown has destroyed what the old one created)
    for (auto &st : g_lvl_stream) st = nullptr;
    g_lvl_begin = nullptr;
    g_lvl_epoch = ctx().epoch;
  }
  if (!g_lvl_begin) FC_HIP_TRY(hipEventCreateWithFlags(&g_lvl_begin, hipEventDisableTiming));
  for (int w = 0; w < n && w < kLevelStreamsMax; ++w)
    if (!g_lvl_stream[w]) FC_HIP_TRY(hipStreamCreateWithFlags(&g_lvl_stream[w], hipStreamNonBlocking));
  return FC_OK;
}

// The whole ladder: first_match[i] = min{j > i : similar(i, j)} or -1.
//
// What a chunk rejects depends on first_match alone: the reference's inner loops never look at
// final_mask (masked structures keep taking part, torsion_module.py:997-1018).  Only two things depend
// on the levels before: whether a level runs at all (5 k < num_active_str, :976) and the length of
// its LAST chunk (:987-990).  So every non-last chunk of every level that can possibly run is
// worked out up front, all levels at once, on the host threads (largest chunks first: the single
// 840 000-structure chunk of k = 2 at 1.7 M structures is the critical path, everything else fits beside
// it); then the levels are walked in order, applying a level's rejects if it runs and doing its
// last chunk -- usually empty or tiny, since num_active_str has long fallen below its start -- on the spot.
// (One level after the other, threads over the chunks of a level: 0.32 s at 1.7 M structures, of which the
// levels k <= 20 with their few huge chunks took 0.24 s on one to five threads.)
// fm_dev (may be nullptr): the same array on the device -- the chunk graphs of the coarse levels (chunks of at least
// gpu_chunk_min structures) are then built there (fc_tfd_gpu.hip), several levels at a time, while the host threads work on the fine levels.
static int tfd_ladder_impl(const int64_t *fm, int64_t N, uint8_t *mask_out, const int64_t *fm_dev) {
  static const double kl[] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};
  std::memset(mask_out, 1, (size_t)N);
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
  // ranges, so workers never write the same byte): the 1.8*10^7 speculative rejects of 1.7 M structures
  // held as index lists were 209 MB of vectors whose release alone cost 90-100 ms.
  struct Task {
    int level;                 // index into kl
    int64_t k, d, step_begin, step_end, cost;
  };
  std::vector<Task> tasks;
  constexpr int kLevels = (int)(sizeof(kl) / sizeof(kl[0]));
  static std::vector<uint8_t> level_rej[kLevels];  // kept across calls (tfd_ladder_from_first_match holds the ladder's own lock): 17 x N bytes of fresh pages cost ~7 ms at 1.7 M
  std::vector<int> gpu_levels;
  bool use_gpu = fm_dev != nullptr;
  if (const char *v = getenv("FC_TFD_GPU")) use_gpu = use_gpu && atoi(v) != 0;  // 0: everything on the host (A/B, tests)
  bool gpu_components = true;  // FC_TFD_GPU_COMPONENTS=0: the component phase on host threads, from the device's graphs
  if (const char *v = getenv("FC_TFD_GPU_COMPONENTS")) gpu_components = atoi(v) != 0;
  int64_t gpu_chunk_min = 150;  // (1000 .. 30 measured at 1.7 M structures: where the host threads and the device finish together -- 300 before the
                                // wave-per-component kernel, 150 since)
  if (const char *v = getenv("FC_TFD_GPU_CHUNK_MIN")) gpu_chunk_min = std::max<int64_t>(2, std::strtoll(v, nullptr, 10));
  for (int li = 0; li < (int)(sizeof(kl) / sizeof(kl[0])); ++li) {
    const int64_t k = (int64_t)kl[li];
    if (!(k == 1 || 5 * k < N)) continue;  // num_active <= N: the level can never run
    if (k == 1) continue;                  // its only chunk is a last chunk
    const int64_t d = N / k;
    if (d <= 1) continue;
    level_rej[li].assign((size_t)(d * (k - 1)), 0);  // the non-last chunks cover [0, d (k - 1))
    if (use_gpu && d >= gpu_chunk_min) {  // a coarse level: its chunk graphs come from the device
      gpu_levels.push_back(li);
      continue;
    }
    // non-last chunks [0, k - 1), cut into tasks of about 2^17 structures (a huge chunk is a task of its own)
    const int64_t per = std::max<int64_t>(1, (int64_t)(131072 / d));
    for (int64_t b = 0; b < k - 1; b += per) {
      const int64_t e = std::min<int64_t>(k - 1, b + per);
      tasks.push_back(Task{li, k, d, b, e, (e - b) * d});
    }
  }
  // (streams and events of the coarse levels' helpers first: nothing below may return while threads are running)
  TfdLevelGraph *const holders = g_lvl_holders;
  hipStream_t *const lvl_stream = g_lvl_stream;
  int n_lvl_streams = 3;
  if (const char *v = getenv("FC_TFD_GPU_STREAMS")) n_lvl_streams = (int)std::min<long>(kLevelStreamsMax, std::max<long>(1, std::strtol(v, nullptr, 10)));
  n_lvl_streams = (int)std::min<size_t>((size_t)n_lvl_streams, std::max<size_t>(gpu_levels.size(), 1));
  int gpu_rc = FC_OK;
  std::string gpu_err;
  if (!gpu_levels.empty()) {
    FC_TRY(tfd_level_streams(n_lvl_streams));
    FC_HIP_TRY(hipEventRecord(g_lvl_begin, ctx().stream));
    for (int w = 0; w < n_lvl_streams; ++w) FC_HIP_TRY(hipStreamWaitEvent(lvl_stream[w], g_lvl_begin, 0));
  }
  // the LAST chunk of the first level that runs is known up front as well (num_active = N there): at 1.7 M structures
  // it is the remainder N - d (k - 1) = 79 619 structures of k = 200 000, 8 ms on one thread if left to the walk below
  int first_level = -1;
  std::vector<int64_t> first_last;
  for (int li = 0; li < kLevels; ++li) {
    const int64_t k = (int64_t)kl[li];
    if (k == 1 || 5 * k < N) {
      first_level = li;
      break;
    }
  }
  if (first_level >= 0) tasks.push_back(Task{first_level, (int64_t)kl[first_level], N / (int64_t)kl[first_level], -1, -1, N - (N / (int64_t)kl[first_level]) * ((int64_t)kl[first_level] - 1)});
  if (debug)
    fprintf(stderr, "[fc] tfd ladder: set up (flag arrays, tasks, level streams) at %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
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
    // (the helpers of the coarse levels wait for the device by polling: leave them a core each)
    const unsigned hw_pool = gpu_levels.empty() ? hw : std::max(1u, hw > (unsigned)n_lvl_streams ? hw - (unsigned)n_lvl_streams : 1u);
    const unsigned nthreads = (N >= 2000 && tasks.size() > 1) ? std::min<size_t>(hw_pool, tasks.size()) : 1;
    // (from here to the end of this block helper threads are running: both holders join in their destructors, so no
    // return, FC_TRY or exception below can leave a joinable std::thread behind -- that would be std::terminate)
    ThreadJoiner pool;
    bool own_worker = !(nthreads > 1 || !gpu_levels.empty());
    if (!own_worker) {
      for (unsigned t = 0; t < std::max(1u, nthreads); ++t)
        if (!pool.start(worker)) break;
      own_worker = pool.threads.empty();  // not one thread to be had: this thread does the host share itself, below
    } else {
      worker();
      own_worker = false;
    }
    // meanwhile: the coarse levels' graphs from the device.  The levels are independent of each other and each is a
    // chain of ~40 short launches with five host round trips (sizes of the next arrays), so one level at a time leaves
    // the device idle most of the time (4.3-5.2 ms per level, nine levels at 1.7 M structures): kLevelStreams helper
    // threads take levels from a common counter, each enqueueing on a stream of its own (thread_stream_override) that
    // starts behind everything the context's stream holds; a helper also walks the components its level left to the
    // host (the few with more than FC_TFD_DEV_COMP_MAX nodes) before it takes the next level.  The graph holders are
    // kept across calls (their arrays are tens of MB: fresh pages every level cost more than the copies).
    std::atomic<size_t> next_level{0};
    std::mutex err_mu;
    auto level_fail = [&](int rc, const char *msg) {
      std::lock_guard<std::mutex> lock(err_mu);
      if (gpu_rc == FC_OK) gpu_rc = rc, gpu_err = msg;
    };
    const int device = ctx().device;
    auto level_worker = [&](int w) {
     try {
      // HIP's current device is PER THREAD and a fresh thread starts on device 0: without this a helper of a context on
      // device d != 0 (every LOCAL_RANK > 0) would take pool blocks, events and pinned pieces on device 0 and launch
      // on device d's streams over them
      if (w != 0) {
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
          level_fail(FC_E_HIP, (std::string("hipSetDevice in a TFD level helper: ") + hipGetErrorString(e)).c_str());
          return;
        }
      }
      thread_stream_override() = lvl_stream[w];
      struct Restore {
        ~Restore() { thread_stream_override() = nullptr; }
      } restore_stream;
      TfdLevelGraph &g = holders[w];
      while (true) {
        {
          std::lock_guard<std::mutex> lock(err_mu);
          if (gpu_rc != FC_OK) break;  // another level failed: no point in starting the next one
        }
        const size_t q = next_level.fetch_add(1);
        if (q >= gpu_levels.size()) break;
        const int li = gpu_levels[gpu_levels.size() - 1 - q];  // coarsest first: they take longest (largest components), the quick fine levels fill the end
        const auto t_g = std::chrono::steady_clock::now();
        uint8_t *flags = level_rej[li].data();
        const int rc = tfd_level_graph_device(fm_dev, N, (int64_t)kl[li], g, gpu_components ? flags : nullptr);
        if (rc != FC_OK) {
          level_fail(rc, last_error().c_str());  // (the message is the helper thread's own)
          break;
        }
        const auto t_c = std::chrono::steady_clock::now();
        if (!gpu_components) level_rejects_from_graph(g, std::min(g_comp_threads, 4u), flags, nullptr);
        else if (!g.left.empty()) level_rejects_from_graph(g, std::min(g_comp_threads, 2u), flags, nullptr, true);
        int64_t left_nodes = 0, left_max = 0;
        if (debug)
          for (const int32_t j : g.left) {
            const int64_t sz = (int64_t)g.sources[(size_t)j + 1] - g.sources[(size_t)j];
            left_nodes += sz;
            left_max = std::max(left_max, sz);
          }
        if (debug)
          fprintf(stderr, "[fc] tfd ladder k=%lld: %lld nodes in the components left to the host (largest %lld)\n", (long long)kl[li],
                  (long long)left_nodes, (long long)left_max);
        if (debug)
          fprintf(stderr, "[fc] tfd ladder k=%lld on the device (stream %d): %.1f ms + %.1f ms on the host (%zu components, %zu of them left to host threads)%s\n",
                  (long long)kl[li], w, std::chrono::duration<double, std::milli>(t_c - t_g).count(),
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count(),
                  (size_t)g.n_components, g.left.size(),
                  gpu_components ? "" : "; component phase on the host");
      }
     } catch (const std::bad_alloc &) {
      level_fail(FC_E_NOMEM, "out of host memory in a TFD level helper");
     } catch (...) {
      level_fail(FC_E_HIP, "unexpected exception in a TFD level helper");
     }
    };
    {
      ThreadJoiner lvl_pool;
      if (!gpu_levels.empty())
        for (int w = 1; w < n_lvl_streams; ++w)
          if (!lvl_pool.start(level_worker, w)) break;  // (fewer helpers: the levels come from a common counter)
      if (!gpu_levels.empty()) level_worker(0);
      lvl_pool.join_all();
      if (debug && !gpu_levels.empty())
        fprintf(stderr, "[fc] tfd ladder: %zu coarse levels on %d streams done at %.1f ms\n", gpu_levels.size(), n_lvl_streams,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
    }
    if (own_worker) worker();
    pool.join_all();
    if (gpu_rc != FC_OK) {
      // a level that failed half-way may have left kernels on its stream: nothing of this call may outlive it
      for (int w = 0; w < n_lvl_streams; ++w) (void)hipStreamSynchronize(lvl_stream[w]);
      return set_error(gpu_rc, "%s", gpu_err.c_str());
    }
    if (host_failed.load()) return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder's host threads");
  }
  if (debug)
    fprintf(stderr, "[fc] tfd ladder: %zu speculative tasks on %u threads, %.1f ms\n", tasks.size(), hw,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
  // the levels in order
  ChunkScratch scratch;
  std::vector<int64_t> last;
  int64_t num_active = N;  // kept up to date as flags go 1 -> 0
  auto reject = [&](int64_t r) {
    num_active -= mask_out[r];
    mask_out[r] = 0;
  };
  for (int li = 0; li < (int)(sizeof(kl) / sizeof(kl[0])); ++li) {
    const int64_t k = (int64_t)kl[li];
    const bool runs = (k == 1 || 5 * k < num_active);
    if (!runs) continue;
    const int64_t d = N / k;
    // the last chunk first reads nothing but first_match either, so the order of application within
    // the level does not matter: all rejects of a level come from the mask-independent chunk graphs
    last.clear();
    const int64_t active_in = num_active;
    if (li == first_level && active_in == N) last = first_last;
    else level_chunks(fm, N, k, d, active_in, k - 1, k, scratch, last);
    {  // branch-free over the bytes (vectorised by the compiler): a level rejects up to half of its range
      const uint8_t *flags = level_rej[li].data();
      const size_t n_flags = level_rej[li].size();
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
      for (; r + 8 <= n_flags; r += 8) {  // eight structures per step (bytes are 0 / 1: a set bit is a rejected structure)
        uint64_t m8, f8;
        std::memcpy(&m8, mask_out + r, 8);
        std::memcpy(&f8, flags + r, 8);
        const uint64_t hit = m8 & f8;
        if (hit) {
          gone += (unsigned long long)__builtin_popcountll(hit);
          m8 ^= hit;
          std::memcpy(mask_out + r, &m8, 8);
        }
      }
      for (; r < n_flags; ++r) {
        const uint8_t hit = (uint8_t)(mask_out[r] & flags[r]);
        gone += hit;
        mask_out[r] = (uint8_t)(mask_out[r] ^ hit);
      }
      num_active -= (int64_t)gone;
    }
    for (int64_t r : last) reject(r);
    if (debug)
      fprintf(stderr, "[fc] tfd ladder k=%lld: %lld active in\n", (long long)k, (long long)active_in);
  }
  if (debug)
    fprintf(stderr, "[fc] tfd ladder total %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count());
  return FC_OK;
}

// The ladder keeps process-wide state across calls (the per-level flag arrays, the helpers' graph holders and streams):
// one ladder at a time, under a lock of its own -- not merely "the callers hold the API lock".  No exception crosses
// the C ABI: whatever the host side throws (std::bad_alloc is the only candidate) becomes an error code.
int tfd_ladder_from_first_match(const int64_t *fm, int64_t N, uint8_t *mask_out, const int64_t *fm_dev) {
  static std::mutex ladder_mu;
  std::lock_guard<std::mutex> lock(ladder_mu);
  try {
    return tfd_ladder_impl(fm, N, mask_out, fm_dev);
  } catch (const std::bad_alloc &) {
    return set_error(FC_E_NOMEM, "out of host memory in the TFD ladder (N = %lld)", (long long)N);
  } catch (const std::exception &e) {
    return set_error(FC_E_HIP, "TFD ladder: %s", e.what());
  }
}

}  // namespace fc
