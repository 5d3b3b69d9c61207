"""The host half of prune_conformers_tfd re-implements CPython's set iteration
order and tuple hash (csrc/fc_tfd_host.cpp).  These CPU-only tests check the
emulation against the running interpreter, and the whole ladder against the
reference's own outputs (golden) given exact first-match arrays."""

import ctypes as C

import numpy as np
import pytest

from firecode_amd import _lib as L
from oracle import cpu_ref as o


def _order_ints(keys):
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    out = np.zeros(len(keys), dtype=np.int64)
    n = C.c_int64(0)
    L.call("fc_debug_pyset_order_ints", L.pi(keys), len(keys), L.pi(out), C.byref(n))
    return out[: n.value].tolist()


def _order_pairs(pairs):
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    out = np.zeros(len(pairs), dtype=np.int64)
    n = C.c_int64(0)
    L.call("fc_debug_pyset_order_pairs", L.pi(pairs), len(pairs), L.pi(out), C.byref(n))
    return [tuple(int(x) for x in pairs[i]) for i in out[: n.value]]


def _py_set_in_order(items):
    s = set()
    for x in items:
        s.add(x)
    return list(s)


@pytest.mark.parametrize("seed", range(12))
def test_int_set_iteration_order(seed):
    rng = np.random.default_rng(seed)
    for n, hi in ((1, 10), (2, 20), (3, 9), (5, 40), (8, 64), (9, 1000), (33, 200), (200, 5000),
                  (1500, 10**6), (60000, 10**7)):
        keys = rng.integers(0, hi, size=n)
        assert _order_ints(keys) == _py_set_in_order(int(k) for k in keys)
    # collisions on purpose: many keys sharing the low bits
    keys = (rng.integers(0, 50, size=300) * 8 + 5)
    assert _order_ints(keys) == _py_set_in_order(int(k) for k in keys)
    keys = rng.permutation(70)
    assert _order_ints(keys) == _py_set_in_order(int(k) for k in keys)


@pytest.mark.parametrize("seed", range(8))
def test_pair_set_iteration_order(seed):
    rng = np.random.default_rng(100 + seed)
    for n, hi in ((1, 5), (4, 30), (7, 100), (40, 400), (700, 3000), (30000, 10**5)):
        i = rng.integers(0, hi, size=n)
        j = i + 1 + rng.integers(0, hi, size=n)
        pairs = np.stack([i, j], axis=1)
        assert _order_pairs(pairs) == _py_set_in_order((int(a), int(b)) for a, b in pairs)


def _first_match_from_tf(tf, thresh=10):
    n = len(tf)
    fm = np.full(n, -1, dtype=np.int64)
    for i in range(n):
        d = np.abs(tf[i + 1:] - tf[i])
        d = np.abs(d - (d > 180) * 360)
        sums = np.array([np.sum(row) for row in d]) if len(d) else np.zeros(0)
        hit = np.flatnonzero(sums < thresh)
        if len(hit):
            fm[i] = i + 1 + hit[0]
    return fm


@pytest.mark.parametrize("name", ["tfdp_small", "tfdp_mid", "tfdp_big", "tfdp_dense", "tfdp_q8", "tfdp_q11", "tfdp_q19"])
def test_ladder_from_first_match_matches_reference(golden, name):
    tf = golden[name + "_tf"]
    fm = _first_match_from_tf(tf)
    mask = np.zeros(len(tf), dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), len(fm), L.pb(mask))
    assert np.array_equal(mask.astype(bool), golden[name + "_mask"])


def test_ladder_random_graphs_against_networkx():
    """random first-match arrays (not from geometry): the C++ replay against the
    oracle's literal networkx loop"""
    rng = np.random.default_rng(7)
    for n in (50, 400, 3000):
        for dens in (0.05, 0.3, 0.9):
            fm = np.full(n, -1, dtype=np.int64)
            for i in range(n - 1):
                if rng.random() < dens:
                    fm[i] = rng.integers(i + 1, min(n, i + 1 + rng.integers(1, 60)))
            # a similarity relation whose first match per row is fm: S[i, fm[i]] only
            tf = None
            mask = np.zeros(n, dtype=np.uint8)
            L.call("fc_tfd_ladder_from_first_match", L.pi(fm), n, L.pb(mask))
            ref = _oracle_ladder_from_fm(fm)
            assert np.array_equal(mask.astype(bool), ref)


def _oracle_ladder_from_fm(fm):
    from networkx import Graph, connected_components

    n = len(fm)
    final_mask = np.ones(n, dtype=bool)
    for k in (5e5, 2e5, 1e5, 5e4, 2e4, 1e4, 5000, 2000, 1000, 500, 200, 100, 50, 20, 10, 5, 2, 1):
        num_active_str = np.count_nonzero(final_mask)
        if k == 1 or 5 * k < num_active_str:
            d = int(n // k)
            for step in range(int(k)):
                if step == k - 1:
                    _l = len(range(d * step, num_active_str))
                else:
                    _l = len(range(d * step, int(d * (step + 1))))
                matches = set()
                for i_rel in range(_l):
                    j = fm[i_rel + d * step] if i_rel + d * step < n else -1
                    if j >= 0 and j - d * step < _l:
                        matches.add((i_rel, int(j - d * step)))
                g = Graph(matches)
                for c in connected_components(g):
                    group = tuple(g.subgraph(c).nodes)
                    for i in set(group) - {group[0]}:
                        final_mask[i + d * step] = 0
    return final_mask


def test_threaded_ladder_equals_serial(monkeypatch):
    """above 10^5 structures every ladder level runs its chunks on host threads: same mask as
    one thread (chunks own disjoint row ranges), on a first-match array with long runs,
    hubs and isolated structures"""
    rng = np.random.default_rng(11)
    n = 150_000
    fm = np.full(n, -1, dtype=np.int64)
    idx = np.arange(n)
    step = rng.integers(1, 40, size=n)
    has = rng.random(n) < 0.8
    tgt = idx + step
    ok = has & (tgt < n)
    fm[ok] = tgt[ok]
    hub = rng.integers(0, n, size=50)  # many structures whose first match is the same later one
    for h in hub:
        lo = max(0, h - 200)
        sel = np.arange(lo, h)[rng.random(h - lo) < 0.5]
        fm[sel] = h
    masks = []
    for threads in ("1", "7"):
        monkeypatch.setenv("FC_TFD_THREADS", threads)
        mask = np.zeros(n, dtype=np.uint8)
        L.call("fc_tfd_ladder_from_first_match", L.pi(fm), n, L.pb(mask))
        masks.append(mask)
    assert np.array_equal(masks[0], masks[1])
    assert 0 < masks[0].sum() < n
