"""The TFD ladder on the device (csrc/fc_tfd_ladder.hip + fc_tfd_core.h) against the host emulation of CPython's sets
and networkx's orders (csrc/fc_tfd_host.cpp, itself checked against the running interpreter in
tests/test_pyset_emulation.py and against the reference's loop by the golden masks): the iteration order of a set of
2-tuples by staged priority first-fit, and whole ladders."""

import numpy as np
import pytest

from firecode_amd import _lib

pytestmark = pytest.mark.gpu


def _orders(pairs):
    pairs = np.ascontiguousarray(pairs, dtype=np.int64)
    n = len(pairs)
    host = np.zeros(n, dtype=np.int64)
    n_out = np.zeros(1, dtype=np.int64)
    _lib.call("fc_debug_pyset_order_pairs", _lib.pi(pairs), n, _lib.pi(host), _lib.pi(n_out))
    assert n_out[0] == n
    dev = np.zeros(n, dtype=np.int64)
    _lib.call("fc_debug_pyset_order_pairs_device", _lib.pi(pairs), n, _lib.pi(dev))
    return host, dev


@pytest.mark.parametrize("n", [1, 4, 5, 6, 18, 19, 20, 76, 77, 78, 307, 5000, 49999, 50000, 50001, 78643, 78644, 200000, 840000])
def test_tuple_set_order_first_match_like_pairs(fc, n):
    """(i, j > i) pairs, one per i, as the chunk graphs have them -- sizes on both sides of every growth step of a
    small set (5, 19, 77, ...) and of the 50 000 rule"""
    rng = np.random.default_rng(n)
    i = np.arange(n, dtype=np.int64)
    gap = np.where(rng.random(n) < 0.7, rng.integers(1, 40, n), rng.integers(1, max(2, n), n))
    pairs = np.stack([i, i + gap], axis=1)
    host, dev = _orders(pairs)
    assert np.array_equal(host, dev)


def test_tuple_set_order_adversarial_hashes(fc):
    """pairs chosen so that many share their low hash bits (long probe chains, perturbation jumps) and a strided
    arrival order"""
    from firecode_amd import _lib as L

    rng = np.random.default_rng(5)
    cand = np.stack([rng.integers(0, 1 << 20, 400000), rng.integers(0, 1 << 20, 400000)], axis=1).astype(np.int64)
    cand = np.unique(cand, axis=0)
    # keep the pairs whose tuple hash falls into 1/64 of the residues mod 2^16: heavy clustering in every table size
    P1, P2, P5 = 11400714785074694791, 14029467366897019727, 2870177450012600261
    M = (1 << 64) - 1
    acc = np.full(len(cand), P5, dtype=np.uint64)
    for k in range(2):
        acc = (acc + cand[:, k].astype(np.uint64) * np.uint64(P2)) & np.uint64(M)
        acc = ((acc << np.uint64(31)) | (acc >> np.uint64(33))) & np.uint64(M)
        acc = (acc * np.uint64(P1)) & np.uint64(M)
    acc = (acc + np.uint64(2 ^ (P5 ^ 3527539))) & np.uint64(M)
    keep = cand[(acc & np.uint64(0xFC00)) == 0]
    assert len(keep) > 3000
    rng.shuffle(keep)
    host, dev = _orders(keep)
    assert np.array_equal(host, dev)


def _random_first_match(rng, n, kind):
    """first_match[i] = -1 or some j > i, in the shapes the ladder meets: neighbours, long jumps, chains, stars"""
    i = np.arange(n, dtype=np.int64)
    if kind == "near":
        j = i + rng.integers(1, 40, n)
    elif kind == "mixed":
        j = i + np.where(rng.random(n) < 0.6, rng.integers(1, 50, n), rng.integers(1, n, n))
    elif kind == "stars":   # many structures share their match: large in-degree
        hubs = np.sort(rng.choice(n, size=max(2, n // 300), replace=False))
        j = hubs[np.minimum(np.searchsorted(hubs, i + 1), len(hubs) - 1)]
    else:                    # "chain": i -> i + 1 in long runs
        j = i + 1
    fm = np.where((j > i) & (j < n) & (rng.random(n) < 0.97), j, -1).astype(np.int64)
    fm[-1] = -1
    return fm


def _host_and_device(monkeypatch, fm):
    n = len(fm)
    out = {}
    for label, env in (("host", "0"), ("device", "1")):
        monkeypatch.setenv("FC_TFD_GPU", env)
        m = np.zeros(n, dtype=np.uint8)
        _lib.call("fc_tfd_ladder_from_first_match", _lib.pi(fm), n, _lib.pb(m))
        out[label] = m
    monkeypatch.delenv("FC_TFD_GPU", raising=False)
    return out["host"], out["device"]


@pytest.mark.parametrize("n,kind,seed", [(25000, "near", 9), (140000, "near", 1), (300000, "mixed", 2), (700000, "stars", 3),
                                          (262144, "chain", 4), (1679611, "mixed", 5), (131072, "mixed", 6), (1000003, "near", 7),
                                          (60000, "stars", 8)])
def test_ladder_mask_is_the_same_on_the_device(fc, monkeypatch, n, kind, seed):
    """fc_tfd_ladder_from_first_match on the device (default with a GPU from 20 000 structures on) == the all-host ladder
    (FC_TFD_GPU=0), whose group[0] bookkeeping is pinned to CPython / networkx by the golden masks and
    tests/test_pyset_emulation.py.  'chain' and 'stars' hold components above 4 096 nodes: the device leaves those to
    the host and the levels are then applied there from the device's flags."""
    rng = np.random.default_rng(seed)
    fm = _random_first_match(rng, n, kind)
    host, dev = _host_and_device(monkeypatch, fm)
    assert np.array_equal(host, dev)
    assert 0 < host.sum() < n


def test_ladder_components_of_every_device_size_class(fc, monkeypatch):
    """Directed shapes for every size class of the device's component phase in ONE first-match array: a chain of 5 001
    nodes (above 4 096: the host's), a star with in-degree 1 600 and a chain of 3 001 nodes (a workgroup each, tables
    of 8 192 slots), a 401-node component with residue collisions (workgroup, 2 048 slots), a 200-node one with
    collisions and a 60-node clean one (a wavefront each), a 30-node one with collisions, 3- and 4-node components with
    collisions (one lane each), a star that is MORE than half of its chunk at a fine level (keeps its earliest node)
    -- all == the all-host ladder."""
    n = 262144
    rng = np.random.default_rng(11)
    i = np.arange(n, dtype=np.int64)
    fm = np.where(rng.random(n) < 0.5, i + rng.integers(1, 30, n), -1).astype(np.int64)
    fm[fm >= n] = -1

    def clear(lo, hi):
        fm[lo:hi] = -1
        back = (fm >= lo) & (fm < hi)
        fm[back] = -1

    clear(1000, 2700)
    fm[1000:2600] = 2600                       # star, in-degree 1 600
    clear(10000, 13100)
    fm[10000:13000] = np.arange(10001, 13001)  # chain of 3 001 nodes
    clear(30000, 30000 + 128 * 402)
    fm[30000:30000 + 128 * 400:128] = 30000 + 128 * 400   # 401 nodes, residues collide modulo 2 048
    clear(100000, 100000 + 128 * 201)
    fm[100000:100000 + 128 * 199:128] = 100000 + 128 * 199  # 200 nodes, residues collide modulo 512
    clear(140000, 140100)
    fm[140000:140059] = 140059                 # 60 nodes, consecutive: clean
    clear(145000, 145000 + 128 * 31)
    fm[145000:145000 + 128 * 29:128] = 145000 + 128 * 29    # 30 nodes, residues collide modulo 128
    clear(150000, 150100)
    fm[150000] = 150008
    fm[150008] = 150016                        # 3 nodes, residues 0 modulo 8
    fm[150040] = 150048
    fm[150048] = 150056
    fm[150056] = 150064                        # 4 nodes
    clear(200000, 200260)
    fm[200000:200255] = 200255                 # 256 nodes inside one chunk of the k = 1000 level (d = 262): > half
    clear(210000, 215100)
    fm[210000:215000] = np.arange(210001, 215001)  # chain of 5 001 nodes
    assert np.all((fm == -1) | ((fm > i) & (fm < n)))
    host, dev = _host_and_device(monkeypatch, fm)
    assert np.array_equal(host, dev)
    assert 0 < host.sum() < n
    # the same array without the 5 001-node chain: nothing is left to the host, the levels are applied on the device
    clear(210000, 215100)
    host, dev = _host_and_device(monkeypatch, fm)
    assert np.array_equal(host, dev)


def test_ladder_whose_last_chunks_hold_edges(fc, monkeypatch):
    """Few structures are rejected, so the LAST chunk of every level (which ends at the active count of its moment,
    torsion_module.py:987-990) is long and holds edges: the device notices and the levels are applied on the host."""
    n = 200000
    rng = np.random.default_rng(21)
    i = np.arange(n, dtype=np.int64)
    fm = np.where(rng.random(n) < 0.02, i + rng.integers(1, 2000, n), -1).astype(np.int64)
    fm[fm >= n] = -1
    host, dev = _host_and_device(monkeypatch, fm)
    assert np.array_equal(host, dev)
    assert 0.9 * n < host.sum() < n


def test_ladder_on_a_device_other_than_zero(fc):
    """Needs a second GPU: a child interpreter runs fc.init(1) and a device ladder of 3e5 structures, mask == the
    all-host ladder."""
    import os
    import subprocess
    import sys

    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import numpy as np\n"
        "import firecode_amd as fc\n"
        "from firecode_amd import _lib\n"
        "from test_tfd_gpu_graph import _random_first_match\n"
        "fc.init(1)\n"
        "fm = _random_first_match(np.random.default_rng(2), 300000, 'mixed')\n"
        "out = {}\n"
        "for label, env in (('host', '0'), ('device', '1')):\n"
        "    os.environ['FC_TFD_GPU'] = env\n"
        "    m = np.zeros(len(fm), dtype=np.uint8)\n"
        "    _lib.call('fc_tfd_ladder_from_first_match', _lib.pi(fm), len(fm), _lib.pb(m))\n"
        "    out[label] = m\n"
        "assert np.array_equal(out['host'], out['device'])\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
