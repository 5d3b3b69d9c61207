"""The device ladder's per-chunk and per-component routines (csrc/fc_tfd_core.h: chunk_front, tiny_first,
comp_group_first, pyset_build) run on the CPU -- one host thread standing for a wavefront, fc_debug_tfd_ladder_emulate --
against the all-host ladder of csrc/fc_tfd_host.cpp (pinned to the reference's loop, firecode/torsion_module.py:957-1043,
by tests/test_tfd_ladder_host.py and the golden masks).  Same source as the kernels of csrc/fc_tfd_ladder.hip; what the CPU
cannot show (barriers, atomics between lanes) is tests/test_tfd_gpu_graph.py's."""

import numpy as np
import pytest

from firecode_amd import _lib as L
from oracle import cpu_ref as o


def _fm(n, rng, kind):
    i = np.arange(n, dtype=np.int64)
    if kind == "near":
        j = i + rng.geometric(1.0 / rng.choice([2, 8, 40, 300, 5000]), size=n)
    elif kind == "pow6":        # cfg3's distances: multiples of 8 and 32 are frequent, residues collide in the small sets
        j = i + rng.choice([1, 6, 36, 216, 1296, 7776], size=n)
    elif kind == "stars":
        hub = ((i // 300) + 1) * 300
        j = np.where(rng.random(n) < 0.7, hub, i + 1 + rng.integers(0, 50, size=n))
        j = np.where(j <= i, i + 1, j)
    elif kind == "chains":
        j = i + 1
    else:
        j = i + 1 + rng.integers(0, n, size=n)
    ok = (j < n) & (rng.random(n) < rng.choice([0.3, 0.8, 0.98]))
    return np.where(ok, j, -1).astype(np.int64)


def _both(fm):
    n = len(fm)
    a = np.zeros(n, dtype=np.uint8)
    b = np.zeros(n, dtype=np.uint8)
    L.call("fc_tfd_ladder_from_first_match", L.pi(fm), n, L.pb(a))
    L.call("fc_debug_tfd_ladder_emulate", L.pi(fm), n, L.pb(b))
    return a, b


@pytest.mark.parametrize("kind", ["near", "pow6", "stars", "chains", "far"])
@pytest.mark.parametrize("n", [30, 300, 2500, 5200, 30000, 120000])
def test_emulated_device_ladder_equals_the_host_ladder(n, kind):
    rng = np.random.default_rng(n + len(kind))
    for _ in range(3 if n <= 30000 else 1):
        fm = _fm(n, rng, kind)
        a, b = _both(fm)
        assert np.array_equal(a, b), (n, kind)


def test_emulated_device_ladder_equals_the_reference_loop():
    """... and the oracle's literal restatement of the reference loop, from fingerprints"""
    rng = np.random.default_rng(4)
    n, q = 2600, 3
    centres = rng.uniform(-180, 180, size=(n // 6, q))
    tf = centres[rng.integers(0, len(centres), n)] + rng.normal(scale=2.0, size=(n, q))
    tf = (tf + 180) % 360 - 180
    fm = np.full(n, -1, dtype=np.int64)
    for i in range(n - 1):
        d = np.abs(tf[i + 1:] - tf[i])
        d = np.abs(d - (d > 180) * 360)
        hit = np.flatnonzero(d.sum(axis=1) < 10)
        if len(hit):
            fm[i] = i + 1 + hit[0]
    ref = o.prune_tfd_from_tf_mat(tf, 10)
    b = np.zeros(n, dtype=np.uint8)
    L.call("fc_debug_tfd_ladder_emulate", L.pi(fm), n, L.pb(b))
    assert np.array_equal(b.astype(bool), ref)


def test_directed_size_classes():
    """one array with components of every class the device distinguishes (1 lane: <= 18 nodes; a wavefront: <= 306; a
    workgroup: <= 4096; the host above), each with colliding residues, and a component of more than half its chunk"""
    n = 262144
    rng = np.random.default_rng(11)
    i = np.arange(n, dtype=np.int64)
    fm = np.where(rng.random(n) < 0.5, i + rng.integers(1, 30, n), -1).astype(np.int64)
    fm[fm >= n] = -1

    def clear(lo, hi):
        fm[lo:hi] = -1
        fm[(fm >= lo) & (fm < hi)] = -1

    clear(1000, 2700)
    fm[1000:2600] = 2600
    clear(10000, 13100)
    fm[10000:13000] = np.arange(10001, 13001)
    clear(30000, 30000 + 128 * 402)
    fm[30000:30000 + 128 * 400:128] = 30000 + 128 * 400
    clear(100000, 100000 + 128 * 201)
    fm[100000:100000 + 128 * 199:128] = 100000 + 128 * 199
    clear(145000, 145000 + 128 * 31)
    fm[145000:145000 + 128 * 29:128] = 145000 + 128 * 29
    clear(150000, 150100)
    fm[150000], fm[150008] = 150008, 150016
    fm[150040], fm[150048], fm[150056] = 150048, 150056, 150064
    clear(200000, 200260)
    fm[200000:200255] = 200255
    clear(210000, 215100)
    fm[210000:215000] = np.arange(210001, 215001)
    a, b = _both(fm)
    assert np.array_equal(a, b)
