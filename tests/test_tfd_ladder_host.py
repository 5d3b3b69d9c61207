"""The host half of prune_conformers_tfd (csrc/fc_tfd_host.cpp, pure host code: no device needed): the
reference's k-ladder / match-graph / "keep group[0]" bookkeeping replayed from the first-match array,
all levels' mask-independent chunks worked out concurrently on host threads.  Checked against the oracle's
literal restatement of firecode/torsion_module.py:957-1043 (itself pinned to the reference's own function
by the golden vectors) for one and several threads."""

import numpy as np
import pytest

from firecode_amd import _lib as L
from oracle import cpu_ref as o


def _first_match(tf, thr=10):
    n = len(tf)
    fm = np.full(n, -1, dtype=np.int64)
    for i in range(n - 1):
        d = np.abs(tf[i + 1:] - tf[i])
        d = np.abs(d - (d > 180) * 360)
        hit = np.flatnonzero(d.sum(axis=1) < thr)
        if len(hit):
            fm[i] = i + 1 + hit[0]
    return fm


@pytest.mark.parametrize("n,q,seed", [(300, 3, 1), (2500, 4, 2), (5200, 3, 3)])
def test_ladder_from_first_match_equals_the_reference_loop(monkeypatch, n, q, seed):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-180, 180, size=(max(n // 6, 1), q))
    tf = centres[rng.integers(0, len(centres), n)] + rng.normal(scale=2.0, size=(n, q))
    tf = (tf + 180) % 360 - 180
    fm = _first_match(tf)
    ref = o.prune_tfd_from_tf_mat(tf, 10)
    assert 0 < ref.sum() < n
    # threads over the chunks of a level; helper threads over the connected components inside a chunk (taken for
    # huge chunks only: FC_TFD_COMP_PAR_MIN=0 sends every chunk down that path)
    for threads, comp_threads, par_min in (("1", "1", None), ("3", "1", None), ("16", "8", None), ("1", "4", "0"), ("4", "3", "0")):
        monkeypatch.setenv("FC_TFD_THREADS", threads)
        monkeypatch.setenv("FC_TFD_COMP_THREADS", comp_threads)
        if par_min is None:
            monkeypatch.delenv("FC_TFD_COMP_PAR_MIN", raising=False)
        else:
            monkeypatch.setenv("FC_TFD_COMP_PAR_MIN", par_min)
        m = np.zeros(n, dtype=np.uint8)
        L.call("fc_tfd_ladder_from_first_match", L.pi(fm), n, L.pb(m))
        assert np.array_equal(m.astype(bool), ref), (threads, comp_threads, par_min)
