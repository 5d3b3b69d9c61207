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
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "fc_common.h"

namespace fc {

// ---- CPython set emulation ---------------------------------------------------
struct PySetEmu {
  struct Entry {
    int64_t key = 0;   // caller-defined id (int value, or index of a pair)
    int64_t hash = 0;
    bool used = false;
  };
  std::vector<Entry> table;
  size_t mask = 7, fill = 0, used_n = 0;
  PySetEmu() : table(8) {}

  static void insert_clean(std::vector<Entry> &t, size_t mask, int64_t key, int64_t hash) {
    size_t perturb = (size_t)hash;
    size_t i = (size_t)hash & mask;
    while (true) {
      size_t e = i;
      if (!t[e].used) {
        t[e] = {key, hash, true};
        return;
      }
      if (i + 9 <= mask) {
        for (int j = 0; j < 9; ++j) {
          ++e;
          if (!t[e].used) {
            t[e] = {key, hash, true};
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
    std::vector<Entry> nt(newsize);
    const size_t newmask = newsize - 1;
    for (size_t s = 0; s <= mask; ++s)
      if (table[s].used) insert_clean(nt, newmask, table[s].key, table[s].hash);
    table.swap(nt);
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
        if (!table[e].used) {
          table[e] = {key, hash, true};
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
        if (!table[e].used) return false;
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
      if (table[s].used) f(table[s].key);
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
  PySetEmu s;
  auto eq = [&](int64_t x, int64_t y) {
    return pairs[x * 2] == pairs[y * 2] && pairs[x * 2 + 1] == pairs[y * 2 + 1];
  };
  for (int64_t k = 0; k < n; ++k) s.add(k, py_tuple2_hash(pairs[k * 2], pairs[k * 2 + 1]), eq);
  out.clear();
  s.for_each([&](int64_t key) { out.push_back(key); });
}

// ---- one chunk: matches (i_rel ascending) -> relative indices to reject -------
// edges[k] = (i_rel, j_rel) in the order the reference adds them to `matches`.
static void chunk_rejects(const std::vector<int64_t> &edges, std::vector<int64_t> &rejects) {
  rejects.clear();
  const int64_t m = (int64_t)edges.size() / 2;
  if (m == 0) return;
  // (1) Graph(matches): edges arrive in the iteration order of the set of tuples
  std::vector<int64_t> order;
  pyset_order_pairs(edges.data(), m, order);
  std::vector<int64_t> nodes;                      // insertion order of g._node
  std::unordered_map<int64_t, int64_t> idx;        // node value -> position in `nodes`
  std::vector<std::vector<int64_t>> adj;           // neighbour lists in insertion order
  auto node_of = [&](int64_t v) {
    auto it = idx.find(v);
    if (it != idx.end()) return it->second;
    const int64_t p = (int64_t)nodes.size();
    idx.emplace(v, p);
    nodes.push_back(v);
    adj.emplace_back();
    return p;
  };
  for (int64_t e : order) {
    const int64_t u = edges[e * 2], v = edges[e * 2 + 1];
    const int64_t pu = node_of(u), pv = node_of(v);
    // dict semantics: re-adding an existing neighbour keeps its position
    bool have = false;
    for (int64_t w : adj[pu]) have = have || (w == pv);
    if (!have) adj[pu].push_back(pv);
    have = false;
    for (int64_t w : adj[pv]) have = have || (w == pu);
    if (!have) adj[pv].push_back(pu);
  }
  const int64_t n_nodes = (int64_t)nodes.size();
  std::vector<char> seen_all(n_nodes, 0);
  int64_t n_seen = 0;
  for (int64_t src = 0; src < n_nodes; ++src) {
    if (seen_all[src]) continue;
    // (2) networkx _plain_bfs: `seen` is a Python set filled in BFS order
    PySetEmu comp;
    std::vector<int64_t> members;  // BFS discovery order (positions)
    const int64_t limit = n_nodes - n_seen;
    comp.add(nodes[src], nodes[src], int_eq);
    members.push_back(src);
    seen_all[src] = 1;
    std::vector<int64_t> level{src}, next;
    bool full = (int64_t)members.size() == limit;
    while (!level.empty() && !full) {
      next.clear();
      for (int64_t v : level) {
        for (int64_t w : adj[v]) {
          if (!seen_all[w]) {
            seen_all[w] = 1;
            comp.add(nodes[w], nodes[w], int_eq);
            members.push_back(w);
            next.push_back(w);
          }
        }
        if ((int64_t)members.size() == limit) {
          full = true;
          break;
        }
      }
      level.swap(next);
    }
    n_seen += (int64_t)members.size();
    // (3) group[0] of tuple(g.subgraph(c).nodes)
    int64_t first;
    if (2 * (int64_t)members.size() < n_nodes) {
      PySetEmu view;  // show_nodes.nodes = set(nbunch_iter(c)): rebuilt in c's iteration order
      comp.for_each([&](int64_t key) { view.add(key, key, int_eq); });
      first = -1;
      view.for_each([&](int64_t key) {
        if (first < 0) first = key;
      });
    } else {
      first = nodes[src];  // atlas order: the BFS source is the component's earliest node
    }
    for (int64_t p : members)
      if (nodes[p] != first) rejects.push_back(nodes[p]);
  }
}

// the whole ladder: first_match[i] = min{j > i : similar(i, j)} or -1
int tfd_ladder_from_first_match(const int64_t *fm, int64_t N, uint8_t *mask_out) {
  static const double kl[] = {5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1};
  for (int64_t i = 0; i < N; ++i) mask_out[i] = 1;
  std::vector<int64_t> edges, rejects;
  for (double kd : kl) {
    const int64_t k = (int64_t)kd;
    int64_t num_active = 0;
    for (int64_t i = 0; i < N; ++i) num_active += mask_out[i];
    if (!(k == 1 || 5 * k < num_active)) continue;
    const int64_t d = N / k;
    for (int64_t step = 0; step < k; ++step) {
      const int64_t lo = d * step;
      // torsion_module.py:987-990: the LAST chunk ends at num_active_str, not at N
      int64_t len = (step == k - 1) ? (num_active - lo) : (d * (step + 1) - lo);
      if (len <= 1) continue;
      if (lo >= N) break;
      edges.clear();
      const int64_t hi = lo + len;  // exclusive; may exceed N only through num_active <= N: never
      for (int64_t i = lo; i < hi && i < N; ++i) {
        const int64_t j = fm[i];
        if (j >= 0 && j < hi) {
          edges.push_back(i - lo);
          edges.push_back(j - lo);
        }
      }
      if (edges.empty()) continue;
      chunk_rejects(edges, rejects);
      for (int64_t r : rejects) mask_out[r + lo] = 0;
    }
  }
  return FC_OK;
}

}  // namespace fc
