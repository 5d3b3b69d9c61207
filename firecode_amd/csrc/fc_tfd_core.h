// fc_tfd_core.h -- the per-chunk and per-component routines of prune_conformers_tfd's k-ladder
// (firecode/torsion_module.py:985-1041), written ONCE for the device and for the host.
//
// What the reference does per chunk: a Python set of first-match pairs (i_rel, j_rel) -> networkx.Graph(matches) ->
// connected_components -> keep tuple(g.subgraph(c).nodes)[0] of every component.  Which node that is follows from
// three CPython / networkx orders (fc_tfd_host.cpp's header); this file states them as data-parallel steps over a
// GROUP of threads (one wavefront, one workgroup -- or one host thread, for the CPU tests and the fallbacks):
//
//   pyset_build     final slot of every key of a CPython set filled in a given arrival order (staged priority
//                   first-fit, one stage per growth step: see fc_tfd_ladder.hip's header);
//   chunk_front     one chunk of at most kChunkMax structures, everything in the group's local memory: tuple-set
//                   slots of the edges, tree roots, component sizes and member lists; components of up to
//                   kTinyMax nodes are finished on the spot (one lane each), larger ones are exported;
//   tiny_first      group[0] of a component of up to kTinyMax nodes by one lane;
//   comp_group_first  group[0] of a component of any size by a whole group: neighbour lists in edge order by
//                   counting, breadth-first order level by level through prefix sums (a first-match graph is a
//                   forest: every neighbour but the one a node was reached from is new), the two int sets by
//                   pyset_build.
//
// Every loop over elements is `for (x = g.tid; x < n; x += g.size)`, every hand-over a g.sync(): with the host group
// (tid 0, size 1) the same source runs sequentially -- tests/test_tfd_ladder_v2.py runs the whole ladder that way on the
// CPU against the all-host ladder of fc_tfd_host.cpp.
#pragma once

#include <cstdint>
#include <type_traits>
#include <cstdlib>

#if defined(__HIPCC__)
#define FC_HD __host__ __device__ __forceinline__
#define FC_HDC __host__ __device__ constexpr
#else
#define FC_HD inline
#define FC_HDC constexpr
#endif

namespace fc {
namespace tfd {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint16_t kNone16 = 0xFFFFu;
constexpr int kTinyMax = 18;       // components of up to this many nodes: sets of at most 32 slots, one lane each
constexpr int kChunkMax = 4915;    // chunks of up to this many structures: tuple set of at most 8192 slots in local memory
constexpr int kGroupCompMax = 4096;  // components of up to this many nodes: one workgroup (larger ones: the host)
constexpr int kTinyScratch = 104;  // bytes of private scratch of tiny_first

// number of slots - 1 of a CPython set that received n_keys distinct keys one by one (setobject.c: resize when
// fill * 5 >= mask * 3, to the first power of two above 4 x used, 2 x above 50 000)
FC_HD uint32_t pyset_final_mask(int64_t n_keys) {
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

// growth stage s of a set that receives m keys: keys of arrival rank < n_cur live in a table of mask + 1 slots built
// from the previous table (t_prev slots, keys of rank < n_prev, in slot order) followed by the later arrivals
struct SetStage {
  int64_t n_prev, n_cur;
  uint32_t t_prev, mask;
  bool exists;
};
FC_HD SetStage pyset_stage(int64_t m, int s) {
  uint64_t mask = 7, t_prev = 0;
  int64_t n_prev = 0;
  for (int i = 0;; ++i) {
    const int64_t trigger = (int64_t)((mask * 3 + 4) / 5);
    if (i == s) return SetStage{n_prev, trigger > m ? m : trigger, (uint32_t)t_prev, (uint32_t)mask, m > 0};
    if (trigger > m) return SetStage{0, 0, 0, 0, false};
    n_prev = trigger;
    t_prev = mask + 1;
    const uint64_t minused = trigger > 50000 ? 2 * (uint64_t)trigger : 4 * (uint64_t)trigger;
    uint64_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
  }
}
FC_HD int pyset_stage_count(int64_t m) {
  int s = 0;
  while (pyset_stage(m, s).exists) ++s;
  return s;
}

// hash((a, b)) of two non-negative Python ints (tupleobject.c, xxHash-style)
FC_HD int64_t tuple2_hash(uint64_t a, uint64_t b) {
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

// CPython's probe sequence (setobject.c set_add_entry): slot i, the next LINEAR_PROBES = 9 slots when they fit below
// the table's end, then i = (i * 5 + 1 + (perturb >>= 5)) & mask
struct Probe {
  uint64_t perturb, i;
  uint32_t mask;
  int j, lim;
  FC_HD void start(int64_t hash, uint32_t mask_) {
    mask = mask_;
    perturb = (uint64_t)hash;
    i = (uint64_t)hash & mask;
    j = 0;
    lim = (i + 9 <= mask) ? 9 : 0;
  }
  FC_HD uint32_t slot() const { return (uint32_t)(i + (uint64_t)j); }
  // the sequence of `hash` positioned AT slot s, which a key of that hash holds: a key sits at the first place of its
  // sequence that it won (holders only ever get stronger), and that is nearly always inside its first linear window
  FC_HD void start_at(int64_t hash, uint32_t mask_, uint32_t s) {
    start(hash, mask_);
    if ((uint64_t)s >= i && (uint64_t)s <= i + (uint64_t)lim) {
      j = (int)((uint64_t)s - i);
      return;
    }
    while (slot() != s) next();
  }
  FC_HD void next() {
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

// ---- the host group: one thread stands for the whole group ---------------------------------------------------------
struct HostGroup {
  int tid = 0, size = 1;
  void sync() const {}
  uint32_t atomic_min(uint32_t *p, uint32_t v) const {
    const uint32_t o = *p;
    if (v < o) *p = v;
    return o;
  }
  uint32_t atomic_add(uint32_t *p, uint32_t v) const {
    const uint32_t o = *p;
    *p = o + v;
    return o;
  }
  uint32_t atomic_or(uint32_t *p, uint32_t v) const {
    const uint32_t o = *p;
    *p = o | v;
    return o;
  }
  uint32_t atomic_cas(uint32_t *p, uint32_t expect, uint32_t v) const {
    const uint32_t o = *p;
    if (o == expect) *p = v;
    return o;
  }
  uint16_t atomic_add16(uint16_t *base, int idx, uint16_t v) const {
    // (the device forms work on the 32-bit word that holds the counter: the array must start on a 4-byte boundary --
    // checked here, where the CPU tests see it)
    if (reinterpret_cast<uintptr_t>(base) & 3u) abort();
    const uint16_t o = base[idx];
    base[idx] = (uint16_t)(o + v);
    return o;
  }
  uint32_t scan_excl(uint32_t v, uint32_t &total) const {
    total = v;
    return 0;
  }
  uint32_t reduce_sum(uint32_t v) const { return v; }
  uint32_t reduce_or(uint32_t v) const { return v; }
  uint64_t reduce_min64(uint64_t v) const { return v; }
  uint32_t bcast(uint32_t v) const { return v; }
  bool in_first_wave() const { return true; }
  HostGroup first_wave() const { return *this; }
};

// ---- a CPython set's final layout ---------------------------------------------------------------------------------------
// ids [0, n_ids); rank[id] = arrival rank of the key (kNone16: not a key), m keys in all, hash(id) = its Python hash.
// Out: pos[id] = the key's slot in the final table.  An open-addressing table filled by first-fit in a FIXED order is
// the unique fixed point of "every key sits in the first slot of its probe sequence not held by a key of higher
// priority", so the keys of a stage may be inserted in ANY order if a key that meets a slot held by a lower-priority
// key takes it and carries the evicted key on along THAT key's sequence: one atomic minimum on (priority << 13 | id)
// per probe.  Priorities of stage s: the slot in the table of stage s - 1 for the keys that were in it (a rebuild
// re-inserts in slot order), table size + arrival rank for the rest.  ids < 8192, table of at most 8192 slots.
template <class G, class HashF>
FC_HD void pyset_build(G &g, int n_ids, const uint16_t *rank, int m, HashF hash, uint32_t *table, uint16_t *pos) {
  uint32_t mask = 7, t_prev = 0;
  int n_prev = 0;
  for (;;) {
    const int trigger = (int)((mask * 3u + 4u) / 5u);
    const int n_cur = trigger > m ? m : trigger;
    for (uint32_t s = (uint32_t)g.tid; s <= mask; s += (uint32_t)g.size) table[s] = kNone;
    g.sync();
    for (int x = g.tid; x < n_ids; x += g.size) {
      const uint32_t r = rank[x];
      if (r >= (uint32_t)n_cur) continue;
      uint32_t me = ((r < (uint32_t)n_prev ? (uint32_t)pos[x] : t_prev + r) << 13) | (uint32_t)x;
      Probe p;
      p.start(hash(x), mask);
      for (;;) {
        const uint32_t s = p.slot();
        const uint32_t old = g.atomic_min(&table[s], me);
        if (old == kNone) break;
        if (old > me) {  // this key outranks the slot's holder: the holder moves on along ITS sequence, behind slot s
          me = old;
          p.start_at(hash((int)(old & 0x1FFFu)), mask, s);
        }
        p.next();
      }
    }
    g.sync();
    for (uint32_t s = (uint32_t)g.tid; s <= mask; s += (uint32_t)g.size) {
      const uint32_t v = table[s];
      if (v != kNone) pos[v & 0x1FFFu] = (uint16_t)s;
    }
    g.sync();
    if (trigger > m) break;
    n_prev = trigger;
    t_prev = mask + 1;
    const uint32_t minused = 4u * (uint32_t)trigger;  // (at most 4915 keys here: never the 2 x rule of large sets)
    uint32_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
  }
}

// The same layout with the keys inserted IN PRIORITY ORDER, a batch of g.size at a time.  Any order gives the same table,
// but not at the same price: a key that arrives before one it must yield to is evicted later, and the evicted key may
// evict the next one along -- members that are runs of consecutive integers (chains i -> i + 1: every key at home, one
// slot apart) get shifted by ONE early key through a cascade as long as the run, carried step by step by one thread
// (0.6e6 cycles per set for a 3 837-node component).  In priority order a key only ever meets stronger holders; what
// is left are the inversions inside one batch.  byrank[r] = the key of arrival rank r (r < m); byord: m entries of
// scratch, on return the keys in the final table's slot order (the set's iteration).
template <class G, class HashF>
FC_HD void pyset_build_ordered(G &g, int m, const uint16_t *byrank, HashF hash, uint32_t *table, uint16_t *pos, uint16_t *byord) {
  uint32_t mask = 7;
  int n_prev = 0;
  auto insert = [&](int x, uint32_t prio) {
    uint32_t me = (prio << 13) | (uint32_t)x;
    Probe p;
    p.start(hash(x), mask);
    for (;;) {
      const uint32_t s = p.slot();
      const uint32_t old = g.atomic_min(&table[s], me);
      if (old == kNone) break;
      if (old > me) {
        me = old;
        p.start_at(hash((int)(old & 0x1FFFu)), mask, s);
      }
      p.next();
    }
  };
  for (;;) {
    const int trigger = (int)((mask * 3u + 4u) / 5u);
    const int n_cur = trigger > m ? m : trigger;
    for (uint32_t s = (uint32_t)g.tid; s <= mask; s += (uint32_t)g.size) table[s] = kNone;
    g.sync();
    for (int base = 0; base < n_prev; base += g.size) {  // the keys of the table before, in its slot order
      const int q = base + g.tid;
      if (q < n_prev) insert(byord[q], (uint32_t)q);
      g.sync();
    }
    for (int base = n_prev; base < n_cur; base += g.size) {  // then the arrivals
      const int r = base + g.tid;
      if (r < n_cur) insert(byrank[r], (uint32_t)r);
      g.sync();
    }
    {  // positions, and the keys in slot order: every thread a contiguous piece of the table, one prefix sum
      const uint32_t size = mask + 1, per = (size + (uint32_t)g.size - 1) / (uint32_t)g.size;
      const uint32_t s0 = (uint32_t)g.tid * per, s1 = s0 + per < size ? s0 + per : size;
      uint32_t cnt = 0;
      for (uint32_t s = s0; s < s1; ++s) cnt += table[s] != kNone;
      uint32_t tot;
      uint32_t at = g.scan_excl(s0 < size ? cnt : 0u, tot);
      for (uint32_t s = s0; s < s1; ++s) {
        const uint32_t v = table[s];
        if (v != kNone) {
          pos[v & 0x1FFFu] = (uint16_t)s;
          byord[at++] = (uint16_t)(v & 0x1FFFu);
        }
      }
    }
    g.sync();
    if (trigger > m) break;
    n_prev = trigger;
    const uint32_t minused = 4u * (uint32_t)trigger;
    uint32_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    mask = newsize - 1;
  }
}

// ---- group[0] of a component of at most kTinyMax nodes, by ONE lane -------------------------------------------------------
// A.x(k): the member's index relative to its chunk (the Python int the reference's sets hold); A.par(k): the relative
// index of its first match when that lies in the chunk, else A.x(k) itself; A.slot(k): the slot of the edge
// (x, par) in the chunk's tuple set (any order-preserving number).  n_graph: nodes of the chunk's whole graph.
template <class Acc>
FC_HD uint32_t tiny_first(const Acc &A, int n, uint32_t n_graph, uint8_t *scr) {
  int src = 0;
  uint32_t best = kNone;
#pragma unroll 6
  for (int k = 0; k < n; ++k) {
    const uint32_t pk = A.par(k), xk = A.x(k), sk = A.slot(k);
    const bool take = pk != xk && sk < best;
    best = take ? sk : best;
    src = take ? k : src;
  }
  // networkx FilterAtlas: a component of at least half the graph is listed in the GRAPH's node order -- its earliest
  // node is the first endpoint of its earliest edge
  if (2u * (uint32_t)n >= n_graph) return A.x(src);
  const uint32_t fmask = n <= 4 ? 7u : 31u;
  {  // no two members in one home slot: every member sits at home whatever the order, the first is the smallest residue
    uint32_t seen = 0, twice = 0, bestr = 64, bx = 0;
#pragma unroll 6
    for (int k = 0; k < n; ++k) {
      const uint32_t xk = A.x(k), r = xk & fmask;
      twice |= seen & (1u << r);
      seen |= 1u << r;
      const bool take = r < bestr;
      bestr = take ? r : bestr;
      bx = take ? xk : bx;
    }
    if (twice == 0u) return bx;
  }
  // the reference's orders: _plain_bfs from the earliest node over the neighbour lists in edge order ...
  uint8_t *q = scr, *frm = scr + 18, *tS = scr + 36, *tV = scr + 68;
  uint32_t *keybuf = reinterpret_cast<uint32_t *>(scr + 36);  // (over tS / tV, which are filled behind the walk: 17 keys)
  q[0] = (uint8_t)src;
  frm[0] = 0xFF;
  int qt = 1;
  for (int qh = 0; qh < qt; ++qh) {
    const int v = q[qh], f = frm[qh];
    const uint32_t xv = A.x(v), pv = A.par(v), sv = A.slot(v);
    // ONE pass over the members per node: its neighbours go straight behind the queue's end, kept in the order of their
    // edges by insertion (a node of a first-match forest has its one first match and a few children).  Every member's
    // three numbers are requested whatever they turn out to be: with the look-ups behind branches -- and a pass per
    // neighbour found -- the walk was a chain of dependent LDS round trips, ~350 cycles per member and pass on a
    // wavefront that has its SIMD to itself (tools/cf_stamps.py).
    int c = 0;
#pragma unroll 6
    for (int k = 0; k < n; ++k) {
      const uint32_t pk = A.par(k), xk = A.x(k), sk = A.slot(k);
      const bool child = pk == xv;                     // a child of v: the edge is the child's
      const bool up = !child && pv != xv && xk == pv;  // v's own first match
      if ((child || up) && k != v && k != f) {
        const uint32_t key = child ? sk : sv;
        int p = c;
        while (p > 0 && keybuf[p - 1] > key) {
          keybuf[p] = keybuf[p - 1];
          q[qt + p] = q[qt + p - 1];
          --p;
        }
        keybuf[p] = key;
        q[qt + p] = (uint8_t)k;
        ++c;
      }
    }
    for (int j = 0; j < c; ++j) frm[qt + j] = (uint8_t)v;
    qt += c;
  }
  // ... into a set of ints (hash(n) == n), then the set of that set's iteration (show_nodes), its first element.
  // Which slots are taken is a 32-bit mask in a register (a set of at most 18 ints has 8 or 32 slots): a probe is a few
  // bit operations, not a byte read from LDS and a wait; the table itself is only written, and read once in slot order.
  auto place = [&](uint8_t *t, uint32_t &occ, uint32_t mask, int k) {
    const uint64_t h = A.x(k);
    uint64_t perturb = h, i = h & mask;
    for (;;) {
      const uint32_t w = (i + 9 <= mask) ? 0x3FFu : 1u;  // the linear window: ten slots, or this one
      const uint32_t open = ~(occ >> (uint32_t)i) & w;
      if (open != 0u) {
        const uint32_t s = (uint32_t)i + (uint32_t)__builtin_ctz(open);
        occ |= 1u << s;
        t[s] = (uint8_t)k;
        return;
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  };
  auto add = [&](uint8_t *t, uint32_t &occ, uint32_t &mask, int &fill, int k) {
    place(t, occ, mask, k);
    ++fill;
    if ((uint32_t)fill * 5u >= mask * 3u) {  // 8 -> 32 slots at the fifth key; 32 slots hold 18
      uint8_t old[8];
      for (int s = 0; s < 8; ++s) old[s] = t[s];
      const uint32_t was = occ;
      occ = 0u;
      mask = 31;
      for (int s = 0; s < 8; ++s)
        if ((was >> s) & 1u) place(t, occ, mask, old[s]);
    }
  };
  uint32_t maskS = 7, maskV = 7, occS = 0u, occV = 0u;
  int fillS = 0, fillV = 0;
  for (int i = 0; i < qt; ++i) add(tS, occS, maskS, fillS, q[i]);
  for (uint32_t m = occS; m != 0u; m &= m - 1u) add(tV, occV, maskV, fillV, tS[__builtin_ctz(m)]);
  if (occV != 0u) return A.x(tV[__builtin_ctz(occV)]);
  return A.x(src);
}

// ---- group[0] of a component of any size, by a whole group ----------------------------------------------------------------
// Local memory of comp_group_first for a component of up to `cap` nodes (cap2 = the power of two >= 2 cap, tbl = slots of
// the sets' final table): see comp_local_bytes.
struct CompLocal {
  uint32_t *X, *S;          // [cap] relative index, edge slot of every member
  uint16_t *pl, *deg, *cur; // [cap] local parent (kNone16: none), degree, fill cursor (later: where a node was reached from)
  uint16_t *head;           // [cap + 2] neighbour list offsets
  uint16_t *adjU, *adjS;    // [2 cap] neighbours as they came / in edge order (adjU later: breadth-first order + ranks)
  uint32_t *table;          // [max(tbl, cap2)] the parent look-up's hash map, then the sets' table
};
FC_HDC size_t comp_local_bytes(size_t cap, size_t tbl, size_t cap2) {
  return cap * 8 + cap * 6 + (cap + 2) * 2 + cap * 8 + (tbl > cap2 ? tbl : cap2) * 4 + 64;
}
FC_HD void comp_local_carve(void *mem, size_t cap, size_t tbl, size_t cap2, CompLocal &L) {
  const size_t t = tbl > cap2 ? tbl : cap2;
  uint8_t *p = static_cast<uint8_t *>(mem);
  L.table = reinterpret_cast<uint32_t *>(p), p += t * 4;
  L.X = reinterpret_cast<uint32_t *>(p), p += cap * 4;
  L.S = reinterpret_cast<uint32_t *>(p), p += cap * 4;
  L.adjU = reinterpret_cast<uint16_t *>(p), p += cap * 4;
  L.adjS = reinterpret_cast<uint16_t *>(p), p += cap * 4;
  L.pl = reinterpret_cast<uint16_t *>(p), p += cap * 2;
  L.deg = reinterpret_cast<uint16_t *>(p), p += cap * 2;
  L.cur = reinterpret_cast<uint16_t *>(p), p += cap * 2;
  L.head = reinterpret_cast<uint16_t *>(p);
}

// mx / mp / ms: the members' relative index, first match (relative; the member itself when it has none in the chunk)
// and edge slot, in any order.  cap2: power of two >= 2 n (the hash map's size).  Returns the kept member's relative index.

// Breadth-first positions of a tree WITHOUT walking it level by level (trees of thousands of levels a few nodes wide: a
// walk is a chain of dependent steps by one lane, however it is coded).  Root the tree at src and list every node's
// children in the order of its neighbour list; then nodes of one depth stand in breadth-first order exactly as in
// depth-first PREORDER (both are the lexicographic order of the child-index paths from the root), so
//   position(w) = #{nodes of smaller depth} + #{nodes of the same depth and smaller preorder number}.
// Depth and preorder number of every node are prefix sums along the Euler tour of the tree (+1 / -1 and 1 / 0 per arc
// down / up): the tour is cut at about one arc per thread, every thread sums its piece, the pieces are chained, and a
// second pass hands each node its numbers.  Local memory: nxt = 2 n uint16 in the table, depth / pre in L.S, the pieces'
// sums over L.pl and L.deg (5 (g.size + 1) uint16: components of more than kWalkMax nodes only), rank out.
constexpr int kWalkMax = 306;
template <class G>
FC_HD void bfs_rank_euler(G &g, const CompLocal &L, int n, int src, uint16_t *rank) {
  uint16_t *up = L.cur, *nxt = reinterpret_cast<uint16_t *>(L.table);
  uint16_t *depth = reinterpret_cast<uint16_t *>(L.S), *pre = depth + n;
  const int narcs = 2 * (n - 1);
  // (1) the parent of every node as seen from src: its first match, except along the path src -> old root, which turns round
  for (int v = g.tid; v < n; v += g.size) up[v] = L.pl[v];
  g.sync();
  if (g.tid == 0) {
    int v = src, prev = kNone16;
    for (;;) {
      const int nx = L.pl[v];
      up[v] = (uint16_t)prev;
      if (nx == kNone16) break;
      prev = v;
      v = nx;
    }
  }
  g.sync();
  // (2) the Euler tour: successor of every arc (v -> w), bit 15 = the arc goes down (w is a child of v)
  const int a0 = L.head[src];
  for (int a = g.tid; a < narcs; a += g.size) {
    int lo = 0, hi = n;  // the arc's owner: head[v] <= a < head[v + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)L.head[mid] <= a) lo = mid;
      else hi = mid;
    }
    const int v = lo, w = L.adjS[a];
    const int w0 = L.head[w], w1 = L.head[w + 1], uw = up[w];
    const bool down = uw == v;
    int succ = -1;
    if (down) {  // into w from its parent: on to w's first child, or straight back up from a leaf
      for (int b = w0; b < w1; ++b)
        if (L.adjS[b] != v) {
          succ = b;
          break;
        }
      if (succ < 0) succ = w0;
    } else {  // back in w from its child v: on to the next child, or up to w's parent, or the tour ends (w is src)
      int t = w0;
      while (L.adjS[t] != v) ++t;
      for (int b = t + 1; b < w1; ++b)
        if (L.adjS[b] != uw) {
          succ = b;
          break;
        }
      if (succ < 0) {
        if (uw == kNone16) succ = a;  // (the last arc: it points to itself)
        else {
          succ = w0;
          while (L.adjS[succ] != uw) ++succ;
        }
      }
    }
    nxt[a] = (uint16_t)(succ | (down ? 0x8000 : 0));
  }
  g.sync();
  // (3) pieces: a piece starts at a0 and at every arc whose number is a multiple of `stride`
  const int stride = narcs > g.size ? (narcs + g.size - 1) / g.size : 1;
  const int K = (narcs + stride - 1) / stride;  // regular pieces; the piece of a0 (when a0 is not one of them) is number K
  const bool a0_extra = (a0 % stride) != 0;
  int16_t *tot_d = reinterpret_cast<int16_t *>(L.pl), *off_d = tot_d + (g.size + 1);
  uint16_t *tot_p = reinterpret_cast<uint16_t *>(off_d + (g.size + 1)), *off_p = tot_p + (g.size + 1), *nxts = off_p + (g.size + 1);
  auto piece_of = [&](int a) { return a == a0 && a0_extra ? K : a / stride; };
  auto starts_piece = [&](int a) { return a == a0 || a % stride == 0; };
  auto walk = [&](int k, int a, bool second) {
    int d = second ? (int)off_d[k] : 0, pp = second ? (int)off_p[k] : 0;
    int nextp = kNone16;
    for (;;) {
      const int e = nxt[a], b = e & 0x7FFF;
      if (e & 0x8000) {
        ++d, ++pp;
        if (second) {
          const int w = L.adjS[a];
          depth[w] = (uint16_t)d, pre[w] = (uint16_t)pp;
        }
      } else {
        --d;
      }
      if (b == a) break;
      a = b;
      if (starts_piece(a)) {
        nextp = piece_of(a);
        break;
      }
    }
    if (!second) tot_d[k] = (int16_t)d, tot_p[k] = (uint16_t)pp, nxts[k] = (uint16_t)nextp;
  };
  for (int k = g.tid; k < K; k += g.size) walk(k, k * stride, false);
  if (g.tid == 0 && a0_extra) walk(K, a0, false);
  g.sync();
  if (g.tid == 0) {
    int d = 0, pp = 0;
    for (int k = piece_of(a0); k != kNone16; k = nxts[k]) {
      off_d[k] = (int16_t)d, off_p[k] = (uint16_t)pp;
      d += tot_d[k], pp += tot_p[k];
    }
    depth[src] = 0, pre[src] = 0;
  }
  g.sync();
  for (int k = g.tid; k < K; k += g.size) walk(k, k * stride, true);
  if (g.tid == 0 && a0_extra) walk(K, a0, true);
  g.sync();
  // (4) positions: nodes by depth (counting sort), inside a depth by preorder number
  const int np = (n + 3) & ~1;  // (even: atomic_add16 works on the 32-bit word that holds the counter)
  uint16_t *start = nxt, *fillc = start + np, *bucket = fillc + np;  // (the tour is no longer needed)
  for (int d = g.tid; d < n + 2; d += g.size) start[d] = 0, fillc[d] = 0;
  g.sync();
  for (int v = g.tid; v < n; v += g.size) g.atomic_add16(fillc, depth[v], 1);
  g.sync();
  {
    uint32_t carry = 0;
    for (int base = 0; base < n + 1; base += g.size) {
      const int d = base + g.tid;
      const uint32_t c = d < n + 1 ? fillc[d] : 0u;
      uint32_t tot;
      const uint32_t prefix = g.scan_excl(c, tot);
      if (d < n + 1) start[d] = (uint16_t)(carry + prefix);
      carry += tot;
    }
  }
  g.sync();
  for (int d = g.tid; d < n + 1; d += g.size) fillc[d] = 0;
  g.sync();
  for (int v = g.tid; v < n; v += g.size) bucket[start[depth[v]] + g.atomic_add16(fillc, depth[v], 1)] = (uint16_t)v;
  g.sync();
  for (int v = g.tid; v < n; v += g.size) {
    const int d = depth[v], b0 = start[d], b1 = start[d + 1], pv = pre[v];
    int before = 0;
    for (int b = b0; b < b1; ++b) before += pre[bucket[b]] < pv;
    rank[v] = (uint16_t)(b0 + before);
  }
  g.sync();
}

#if defined(FC_TFD_STAMPS) && defined(__HIPCC__)
// tuning build: cycles of the phases of the LARGEST component seen (tools/ladder_stamps.py reads g_tfd_stamps)
static __device__ unsigned long long g_tfd_stamps[16];
#endif
#if defined(FC_TFD_STAMPS) && defined(__HIP_DEVICE_COMPILE__) && defined(FC_TFD_STAMPS_SMALL)
// (second form: the phases of the components of 19 ... 76 nodes, one in 64 sampled, ticks of 10 ns SUMMED; slot 11 counts them)
#define FC_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (g.tid == 0 && n <= 76 && (mx[0] & 63u) == 0u) {                                              \
      const unsigned long long now_ = wall_clock64();                                                \
      atomicAdd(&g_tfd_stamps[i], now_ - t_prev_);                                                   \
      if ((i) == 0) atomicAdd(&g_tfd_stamps[11], 1ull);                                              \
      t_prev_ = now_;                                                                                \
    }                                                                                                \
  } while (0)
#define FC_STAMP_BEGIN unsigned long long t_prev_ = wall_clock64()
#define FC_COUNT(i, v) do { } while (0)
#elif defined(FC_TFD_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define FC_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (g.tid == 0 && n >= 3000) {                                                                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                  \
      atomicMax(&g_tfd_stamps[i], now_ - t_prev_);                                                   \
      t_prev_ = now_;                                                                                \
    }                                                                                                \
  } while (0)
#define FC_STAMP_BEGIN unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#define FC_COUNT(i, v) do { if (n >= 3000) atomicAdd(&g_tfd_stamps[i], (unsigned long long)(v)); } while (0)
#else
#define FC_COUNT(i, v) do { } while (0)
#define FC_STAMP(i) do { } while (0)
#define FC_STAMP_BEGIN do { } while (0)
#endif

template <class G>
FC_HD uint32_t comp_group_first(G &g, const CompLocal &L, const uint32_t *mx, const uint32_t *mp, const uint32_t *ms, int n,
                                uint32_t n_graph, uint32_t cap2) {
  FC_STAMP_BEGIN;
  for (int k = g.tid; k < n; k += g.size) L.X[k] = mx[k];
  g.sync();
  // the earliest edge's first endpoint: the component's earliest node in the graph's order, the search's source
  uint64_t mine = ~0ull;
  for (int k = g.tid; k < n; k += g.size)
    if (mp[k] != mx[k]) {
      const uint64_t key = ((uint64_t)ms[k] << 16) | (uint64_t)k;
      mine = key < mine ? key : mine;
    }
  const int src = (int)(g.reduce_min64(mine) & 0xFFFFull);
  if (2u * (uint32_t)n >= n_graph) return L.X[src];
  const uint32_t fmask = pyset_final_mask(n);
  {  // residues all different: the smallest one's member
    const uint32_t words = (fmask + 32u) >> 5;
    for (uint32_t w = (uint32_t)g.tid; w < words; w += (uint32_t)g.size) L.table[w] = 0u;
    g.sync();
    uint32_t dirty = 0;
    uint64_t bestr = ~0ull;
    for (int k = g.tid; k < n; k += g.size) {
      const uint32_t r = L.X[k] & fmask, bit = 1u << (r & 31);
      dirty |= (g.atomic_or(&L.table[r >> 5], bit) & bit) != 0u;
      const uint64_t key = ((uint64_t)r << 32) | (uint64_t)L.X[k];
      bestr = key < bestr ? key : bestr;
    }
    dirty = g.reduce_or(dirty);
    bestr = g.reduce_min64(bestr);
    g.sync();
    if (!dirty) return (uint32_t)(bestr & 0xFFFFFFFFull);
  }
  FC_STAMP(0);
  // local parent of every member through a hash map relative index -> member
  const uint32_t hmask = cap2 - 1;
  for (uint32_t s = (uint32_t)g.tid; s < cap2; s += (uint32_t)g.size) L.table[s] = 0u;
  g.sync();
  for (int k = g.tid; k < n; k += g.size) {
    uint32_t h = (L.X[k] * 2654435761u) & hmask;
    while (g.atomic_cas(&L.table[h], 0u, (uint32_t)k + 1u) != 0u) h = (h + 1) & hmask;
    L.deg[k] = 0;
    L.cur[k] = 0;
  }
  g.sync();
  for (int k = g.tid; k < n; k += g.size) {
    L.S[k] = ms[k];
    uint16_t p = kNone16;
    if (mp[k] != mx[k]) {
      uint32_t h = (mp[k] * 2654435761u) & hmask;
      for (;;) {
        const uint32_t e = L.table[h];
        if (e == 0u) break;  // (cannot happen: a first match inside the chunk is a node of the same tree)
        if (L.X[e - 1] == mp[k]) {
          p = (uint16_t)(e - 1);
          break;
        }
        h = (h + 1) & hmask;
      }
    }
    L.pl[k] = p;
  }
  g.sync();
  for (int k = g.tid; k < n; k += g.size)
    if (L.pl[k] != kNone16) {
      g.atomic_add16(L.deg, k, 1);
      g.atomic_add16(L.deg, L.pl[k], 1);
    }
  g.sync();
  FC_STAMP(1);
  {  // neighbour list offsets
    uint32_t carry = 0;
    for (int base = 0; base < n; base += g.size) {
      const int k = base + g.tid;
      const uint32_t v = k < n ? L.deg[k] : 0u;
      uint32_t tot;
      const uint32_t pre = g.scan_excl(v, tot);
      if (k < n) L.head[k] = (uint16_t)(carry + pre);
      carry += tot;
    }
    if (g.tid == 0) L.head[n] = (uint16_t)carry;
  }
  g.sync();
  for (int k = g.tid; k < n; k += g.size)
    if (L.pl[k] != kNone16) {
      const int p = L.pl[k];
      L.adjU[L.head[k] + g.atomic_add16(L.cur, k, 1)] = (uint16_t)p;
      L.adjU[L.head[p] + g.atomic_add16(L.cur, p, 1)] = (uint16_t)k;
    }
  g.sync();
  FC_STAMP(2);
  // neighbour lists in edge order (what networkx's adjacency dicts iterate): place = edges of the node with a smaller slot
  const int n_rec = 2 * (n - 1);
  for (int rec = g.tid; rec < n_rec; rec += g.size) {
    int lo = 0, hi = n;  // the node of record rec: head[v] <= rec < head[v + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)L.head[mid] <= rec) lo = mid;
      else hi = mid;
    }
    const int v = lo, w = L.adjU[rec];
    const uint32_t key = (L.pl[v] == w) ? L.S[v] : L.S[w];
    int place = 0;
    for (int a = L.head[v]; a < (int)L.head[v + 1]; ++a) {
      const int w2 = L.adjU[a];
      place += ((L.pl[v] == w2) ? L.S[v] : L.S[w2]) < key;
    }
    L.adjS[L.head[v] + place] = (uint16_t)w;
  }
  g.sync();
  FC_STAMP(3);
  // breadth-first order, level by level: every neighbour of a node except the one it was reached from is new.  The
  // levels of these trees are many and narrow (hundreds of levels of a few nodes): ONE wavefront walks them (its
  // hand-overs cost a few cycles where a workgroup barrier per level costs a microsecond), the others wait
  uint16_t *bfs = L.adjU, *rank = L.adjU + n, *frm = L.cur, *pos = reinterpret_cast<uint16_t *>(L.S);
  if (n > kWalkMax) {
    bfs_rank_euler(g, L, n, src, rank);
    FC_STAMP(4);
  } else {
    uint64_t *nb = reinterpret_cast<uint64_t *>(L.table);  // (the table is free between the parent look-up and the sets)
    for (int v = g.tid; v < n; v += g.size) {
      const int a0 = L.head[v], dg = (int)L.head[v + 1] - a0;
      uint64_t word = 0;
      for (int q = 0; q < 4; ++q) word |= (uint64_t)(q < dg ? L.adjS[a0 + q] : (uint16_t)0xFFFFu) << (16 * q);
      if (dg > 4) word = (word & 0x0000FFFFFFFFFFFFull) | (0xFFFEull << 48);
      nb[v] = word;
    }
    if (g.tid == 0) bfs[0] = (uint16_t)src, frm[src] = kNone16;
    g.sync();
    FC_STAMP(7);
    if (g.in_first_wave()) {
      auto w = g.first_wave();
      int lo = 0, hi = 1;
      while (lo < hi) {
        if (hi - lo <= 4) {
          // narrow levels (most of them, in trees that are thousands of levels deep): lane 0 walks on by itself until a
          // level is wider, the level's nodes in registers and a node's first four neighbours in ONE 8-byte word -- one
          // round trip to local memory per LEVEL, against a wave-wide prefix sum and a hand-over
          if (w.tid == 0) {
            int cnt = hi - lo;
            uint16_t cv[4], cf[4];
            for (int i = 0; i < 4; ++i)
              if (i < cnt) cv[i] = bfs[lo + i], cf[i] = frm[cv[i]];
            while (cnt > 0 && cnt <= 4) {
              if (cnt == 1 && (uint16_t)(nb[cv[0]] >> 48) == 0xFFFFu) {
                // a single node with at most three neighbours (the links of a chain): nothing to index dynamically
                const int v = cv[0], f = cf[0];
                const uint64_t word = nb[v];
                const int x0 = (int)(word & 0xFFFFu), x1 = (int)((word >> 16) & 0xFFFFu), x2 = (int)((word >> 32) & 0xFFFFu);
                int o = hi;
                if (x0 != 0xFFFF && x0 != f) bfs[o++] = (uint16_t)x0, frm[x0] = (uint16_t)v;
                if (x1 != 0xFFFF && x1 != f) bfs[o++] = (uint16_t)x1, frm[x1] = (uint16_t)v;
                if (x2 != 0xFFFF && x2 != f) bfs[o++] = (uint16_t)x2, frm[x2] = (uint16_t)v;
                FC_COUNT(8, 1);
                const int c = o - hi;
                lo = hi;
                hi = o;
                cnt = c;
                if (x0 != 0xFFFF && x0 != f) cv[0] = (uint16_t)x0;  // (the next level in order: at most three nodes)
                else if (x1 != 0xFFFF && x1 != f) cv[0] = (uint16_t)x1;
                else cv[0] = (uint16_t)x2;
                if (c >= 2) cv[1] = bfs[lo + 1];
                if (c >= 3) cv[2] = bfs[lo + 2];
                cf[0] = cf[1] = cf[2] = (uint16_t)v;
                continue;
              }
              FC_COUNT(9, 1);
              uint64_t nbv[4];
              for (int i = 0; i < 4; ++i)
                if (i < cnt) nbv[i] = nb[cv[i]];
              int o = hi, ncnt = 0;
              uint16_t nv[4], nf[4];
              for (int i = 0; i < 4; ++i) {
                if (i >= cnt) break;
                const int v = cv[i], f = cf[i];
                auto reach = [&](int x) {
                  if (x == f) return;
                  bfs[o++] = (uint16_t)x, frm[x] = (uint16_t)v;
                  if (ncnt < 4) nv[ncnt] = (uint16_t)x, nf[ncnt] = (uint16_t)v;
                  ++ncnt;
                };
                if ((uint16_t)(nbv[i] >> 48) == 0xFFFEu) {  // more than four neighbours: the list itself
                  for (int a = L.head[v]; a < (int)L.head[v + 1]; ++a) reach(L.adjS[a]);
                } else {
                  for (int q = 0; q < 4; ++q) {
                    const int x = (int)((nbv[i] >> (16 * q)) & 0xFFFFu);
                    if (x == 0xFFFF) break;
                    reach(x);
                  }
                }
              }
              lo = hi;
              hi = o;
              cnt = ncnt;
              for (int i = 0; i < 4; ++i) cv[i] = nv[i], cf[i] = nf[i];
            }
          }
          w.sync();
          lo = (int)w.bcast((uint32_t)lo);
          hi = (int)w.bcast((uint32_t)hi);
          continue;
        }
        if (w.tid == 0) FC_COUNT(10, 1);
        uint32_t carry = (uint32_t)hi;
        for (int base = lo; base < hi; base += w.size) {
          const int p = base + w.tid;
          int v = 0, f = 0;
          uint32_t c = 0;
          if (p < hi) {
            v = bfs[p];
            f = frm[v];
            c = (uint32_t)L.deg[v] - (f != kNone16 ? 1u : 0u);
          }
          uint32_t tot;
          uint32_t o = carry + w.scan_excl(c, tot);
          if (p < hi)
            for (int a = L.head[v]; a < (int)L.head[v + 1]; ++a) {
              const int x = L.adjS[a];
              if (x != f) bfs[o++] = (uint16_t)x, frm[x] = (uint16_t)v;
            }
          carry += tot;
        }
        w.sync();
        lo = hi;
        hi = (int)carry;
      }
    }
    g.sync();
    FC_STAMP(4);
    for (int p = g.tid; p < n; p += g.size) rank[bfs[p]] = (uint16_t)p;
    g.sync();
  }
  // the set the search fills (keys in breadth-first order) ...
  auto hash = [&](int k) { return (int64_t)L.X[k]; };
  uint16_t *byord = L.deg;
  if (n > kWalkMax) {  // (the walk leaves the order itself in bfs)
    for (int v = g.tid; v < n; v += g.size) bfs[rank[v]] = (uint16_t)v;
    g.sync();
  }
  pyset_build_ordered(g, n, bfs, hash, L.table, pos, byord);
  FC_STAMP(5);
  // ... and the set made from its iteration (slot order): its first element
  for (int q = g.tid; q < n; q += g.size) bfs[q] = byord[q];
  g.sync();
  pyset_build_ordered(g, n, bfs, hash, L.table, pos, byord);
  FC_STAMP(6);
  return L.X[byord[0]];
}

// ---- one chunk of at most kChunkMax structures ---------------------------------------------------------------------------
struct ChunkLocal {
  uint16_t *jr, *root, *pos, *rank;  // [dpad] each
  uint32_t *table;                   // [tbl]; later three uint16 arrays of dpad
  uint8_t *mark;                     // [2 * ceil(dpad / 2)] graph-node marks; later the list of roots (uint16)
  uint8_t *tiny;                     // [group size * kTinyScratch]
};
FC_HDC int chunk_dpad(int dmax) { return (dmax + 3) & ~3; }
FC_HDC size_t chunk_local_bytes(int dmax, int tbl, int group) {
  return (size_t)chunk_dpad(dmax) * 8 +
         ((size_t)tbl * 4 < (size_t)chunk_dpad(dmax) * 6 ? (size_t)chunk_dpad(dmax) * 6 : (size_t)tbl * 4) +
         (size_t)chunk_dpad(dmax) + (size_t)group * kTinyScratch + 16;
}
FC_HD void chunk_local_carve(void *mem, int dmax, int tbl, ChunkLocal &L) {
  const size_t dp = (size_t)chunk_dpad(dmax);
  size_t t = (size_t)tbl * 4;
  if (t < dp * 6) t = dp * 6;
  uint8_t *p = static_cast<uint8_t *>(mem);
  L.table = reinterpret_cast<uint32_t *>(p), p += t;
  L.jr = reinterpret_cast<uint16_t *>(p), p += dp * 2;
  L.root = reinterpret_cast<uint16_t *>(p), p += dp * 2;
  L.pos = reinterpret_cast<uint16_t *>(p), p += dp * 2;
  L.rank = reinterpret_cast<uint16_t *>(p), p += dp * 2;
  L.mark = p, p += dp;
  L.tiny = p;
}

// where chunk_front leaves the components it does not finish itself
struct CompRecord {
  uint32_t t0;       // first item of the chunk (flags[t0 + relative index])
  uint32_t moff;     // first member in mx / mp / ms
  uint32_t n;        // members (0: empty record)
  uint32_t n_graph;  // nodes of the chunk's graph
};
struct ChunkExport {
  CompRecord *rec;         // this chunk's records: at most d / 19 of them
  uint32_t *mx, *mp, *ms;  // this chunk's region of the member arrays: d entries each
  uint32_t moff0;          // absolute index of mx[0]
};

// fm: first-match array (absolute indices, -1: none); the chunk = structures [lo, lo + d); flags[x] = 1 for every
// structure the chunk rejects (tiny components); the others go to ex.  Returns the number of exported components.
#if defined(FC_TFD_STAMPS) && defined(__HIPCC__)
static __device__ unsigned long long g_cf_stamps[48];
#endif
#if defined(FC_TFD_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
// tuning build: ticks per phase of chunk_front, one chunk in 64 sampled, slots [16 + 8 e + phase] by chunk-length class e
#define FC_CF_BEGIN const bool cf_on_ = g.tid == 0 && ((t0 / (uint32_t)(d > 0 ? d : 1)) & 63u) == 0u; unsigned long long cf_t_ = cf_on_ ? wall_clock64() : 0ull; \
  const int cf_e_ = d <= 19 ? 0 : d <= 77 ? 1 : d <= 307 ? 2 : d <= 1229 ? 3 : 4
#define FC_CF(i) do { if (cf_on_) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_cf_stamps[cf_e_ * 8 + (i)], n_ - cf_t_); cf_t_ = n_; } } while (0)
#define FC_CF_N(i, v) do { if (cf_on_) atomicAdd(&g_cf_stamps[cf_e_ * 8 + (i)], (unsigned long long)(v)); } while (0)
#else
#define FC_CF_BEGIN do { } while (0)
#define FC_CF(i) do { } while (0)
#define FC_CF_N(i, v) do { } while (0)
#endif
template <class G>
FC_HD int chunk_front(G &g, const ChunkLocal &L, const int64_t *fm, int64_t lo, int d, uint32_t t0, uint8_t *flags,
                      const ChunkExport &ex) {
  const int dp = chunk_dpad(d);
  int m = 0;
  FC_CF_BEGIN;
  {
    uint32_t carry = 0;
    for (int base = 0; base < d; base += g.size) {
      const int x = base + g.tid;
      uint32_t v = 0;
      if (x < d) {
        const int64_t j = fm[lo + x];
        v = (j >= 0 && j < lo + d) ? 1u : 0u;
        L.jr[x] = v ? (uint16_t)(j - lo) : (uint16_t)x;
      }
      uint32_t tot;
      const uint32_t pre = g.scan_excl(v, tot);
      if (x < d) L.rank[x] = v ? (uint16_t)(carry + pre) : kNone16;
      carry += tot;
    }
    m = (int)carry;
  }
  g.sync();
  FC_CF(0);
  FC_CF_N(7, 1);
  if (m == 0) return 0;
  // (1) the edge order: slots of the chunk's set of (i_rel, j_rel) tuples, filled in ascending i_rel
  if (m >= 2) {
    pyset_build(g, d, L.rank, m, [&](int x) { return tuple2_hash((uint64_t)x, (uint64_t)L.jr[x]); }, L.table, L.pos);
  } else {
    for (int x = g.tid; x < d; x += g.size) L.pos[x] = 0;
    g.sync();
  }
  FC_CF(1);
  // (2) tree roots by pointer jumping (a first-match graph is a forest: one edge per node, to a later one)
  for (int x = g.tid; x < d; x += g.size) L.root[x] = L.jr[x];
  g.sync();
  {
    int rounds = 1;
    while ((1 << rounds) < d) ++rounds;
    for (int r = 0; r <= rounds; ++r) {
      for (int x = g.tid; x < d; x += g.size) L.root[x] = L.root[L.root[x]];
      g.sync();
    }
  }
  FC_CF(2);
  // (3) graph nodes, component sizes, members grouped by component
  uint16_t *size = reinterpret_cast<uint16_t *>(L.table), *off = size + dp, *fill = off + dp, *lmem = L.rank;
  uint16_t *rlist = reinterpret_cast<uint16_t *>(L.mark);
  for (int x = g.tid; x < dp; x += g.size) L.mark[x] = 0, size[x] = 0, fill[x] = 0;
  g.sync();
  for (int x = g.tid; x < d; x += g.size)
    if (L.jr[x] != x) L.mark[x] = 1, L.mark[L.jr[x]] = 1;
  g.sync();
  uint32_t ng = 0;
  for (int x = g.tid; x < d; x += g.size)
    if (L.mark[x]) {
      ++ng;
      g.atomic_add16(size, L.root[x], 1);
    }
  ng = g.reduce_sum(ng);
  g.sync();
  {
    uint32_t carry = 0;
    for (int base = 0; base < d; base += g.size) {
      const int x = base + g.tid;
      const bool is = x < d && L.root[x] == x && size[x] >= 2;
      uint32_t tot;
      const uint32_t pre = g.scan_excl(is ? (uint32_t)size[x] : 0u, tot);
      if (is) off[x] = (uint16_t)(carry + pre);
      carry += tot;
    }
  }
  g.sync();
  for (int x = g.tid; x < d; x += g.size)
    if (L.mark[x]) {
      const int r = L.root[x];
      lmem[off[r] + g.atomic_add16(fill, r, 1)] = (uint16_t)x;
    }
  g.sync();
  int nr = 0;
  {
    uint32_t carry = 0;
    for (int base = 0; base < d; base += g.size) {
      const int x = base + g.tid;
      const bool is = x < d && L.root[x] == x && size[x] >= 2;
      uint32_t tot;
      const uint32_t pre = g.scan_excl(is ? 1u : 0u, tot);
      // (the marks of the structures behind this sweep are still to be read by nobody: the member lists are complete)
      if (is) fill[carry + pre] = (uint16_t)x;  // (fill is free again; rlist would overwrite marks this sweep has not passed)
      carry += tot;
    }
    nr = (int)carry;
  }
  g.sync();
  for (int c = g.tid; c < nr; c += g.size) rlist[c] = fill[c];
  g.sync();
  FC_CF(3);
  // (4) components: the tiny ones here, one lane each; the others exported
  struct Acc {
    const uint16_t *lm, *jr, *pos;
    FC_HD uint32_t x(int k) const { return lm[k]; }
    FC_HD uint32_t par(int k) const { return jr[lm[k]]; }
    FC_HD uint32_t slot(int k) const { return pos[lm[k]]; }
  };
  // The components in the order of their size, largest first (a counting sort over 2 .. kTinyMax nodes and "larger"):
  // a lane's walk costs ~n^2 steps and a wavefront waits for its slowest lane -- taken as they come, every wavefront of
  // a chunk of thousands had a component of a dozen nodes among its 64 and the three in four that have 2 or 3 nodes
  // waited for it (tools/cf_stamps.py: this phase was 73 % of the workgroup-per-chunk kernel, half of the others').
  // sorted = root's array (the roots are in size / off / the member lists by now); the large ones come first: they are
  // the export list.
  // bins[0 .. kTinyMax]: count per size (index 0: larger), then the cursors -- in the first lane's scratch, which no walk uses yet
  static_assert(kTinyScratch >= 2 * (2 * (kTinyMax + 1) + 2), "the size bins live in one lane's scratch");
  uint16_t *sorted = L.root, *bins = reinterpret_cast<uint16_t *>(L.tiny);
  constexpr int kBins = kTinyMax + 1;       // (index n for 2 <= n <= kTinyMax; 0: more than kTinyMax)
  for (int k = g.tid; k < 2 * kBins + 2; k += g.size) bins[k] = 0;
  g.sync();
  for (int ci = g.tid; ci < nr; ci += g.size) {
    const int n = size[rlist[ci]];
    g.atomic_add16(bins, n > kTinyMax ? 0 : n, 1);
  }
  g.sync();
  int nexp = bins[0];
  for (int ci = g.tid; ci < nr; ci += g.size) {
    const int r = rlist[ci], n = size[r], bin = n > kTinyMax ? 0 : n;
    int start = 0;  // components in front of this size: the larger ones, then the sizes above
    if (bin != 0) {
      start = bins[0];
      for (int q = kTinyMax; q > bin; --q) start += bins[q];
    }
    sorted[start + (int)g.atomic_add16(bins + kBins + 1, bin, 1)] = (uint16_t)r;
  }
  g.sync();
  uint16_t *biglist = sorted;
  for (int ci = nexp + g.tid; ci < nr; ci += g.size) {
    const int r = sorted[ci], n = size[r], o = off[r];
    const Acc A{lmem + o, L.jr, L.pos};
    const uint32_t first = tiny_first(A, n, ng, L.tiny + (size_t)g.tid * kTinyScratch);
    for (int k = 0; k < n; ++k)
      if (lmem[o + k] != first) flags[lmem[o + k]] = 1;
  }
  g.sync();
  FC_CF(4);
  for (int b = 0; b < nexp; ++b) {
    const int r = biglist[b], n = size[r], o = off[r];
    for (int k = g.tid; k < n; k += g.size) {
      const int x = lmem[o + k];
      ex.mx[o + k] = (uint32_t)x;
      ex.mp[o + k] = (uint32_t)L.jr[x];
      ex.ms[o + k] = (uint32_t)L.pos[x];
    }
    if (g.tid == 0) ex.rec[b] = CompRecord{t0, ex.moff0 + (uint32_t)o, (uint32_t)n, ng};
  }
  FC_CF(5);
  return nexp;
}

}  // namespace tfd
}  // namespace fc
