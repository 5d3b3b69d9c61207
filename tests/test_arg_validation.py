"""Argument checks of the C ABI happen on the host BEFORE any device is touched, so they
are testable without a GPU: a bad shape / index / threshold must come back as FC_E_INVALID
(FirecodeHipInputError), never reach a kernel."""

import ctypes as C

import numpy as np
import pytest

import firecode_amd as fc
from firecode_amd import _lib as L


def _tri_args(J=1, U=2, S=3, A=(4, 5, 3)):
    coords = [np.zeros((2, a, 3)) for a in A]
    reactive = [np.array([0, 1], dtype=np.int64) for _ in A]
    cptr = (C.POINTER(C.c_double) * 3)(*[L.pf(c) for c in coords])
    rptr = (C.POINTER(C.c_int64) * 3)(*[L.pi(r) for r in reactive])
    keep = (coords, reactive)
    a = dict(
        cptr=cptr, nconf=L.i64([2, 2, 2]), natm=L.i64(list(A)), rptr=rptr, nreact=L.i64([2, 2, 2]), J=J,
        conf=np.zeros((J, 3), dtype=np.int64), ps=np.zeros((J, 3, 3)), pe=np.ones((J, 3, 3)),
        vecs=np.zeros((J, 8, 3, 2, 3)), d0=np.zeros((J, 3, 3)), run=np.ones((J, 8), dtype=np.uint8),
        rtab=np.zeros((J, 8, 3, 3), dtype=np.int64), norms=np.ones((J, 3)), ua=np.zeros((3, U)), U=U,
        aidx=np.zeros((S, 3), dtype=np.int32), S=S, thresh=1.5, mc=0, rthr=1.0, dirs=np.zeros((J, 8, 3, 3)),
        Rt=np.zeros((J, 8, 3, U, 12)), ok=np.zeros((J, 8, S), dtype=np.uint8), acc=np.zeros((J, 8, S), dtype=np.uint8))
    return a, keep


def _call_tri(a):
    L.call("fc_embed_trimolecular", a["cptr"], L.pi(a["nconf"]), L.pi(a["natm"]), a["rptr"], L.pi(a["nreact"]),
           a["J"], L.pi(a["conf"]), L.pf(a["ps"]), L.pf(a["pe"]), L.pf(a["vecs"]), L.pf(a["d0"]), L.pb(a["run"]),
           L.pi(a["rtab"]), L.pf(a["norms"]), L.pf(a["ua"]), a["U"], a["aidx"].ctypes.data_as(C.POINTER(C.c_int32)),
           a["S"], a["thresh"], a["mc"], a["rthr"], L.pf(a["dirs"]), L.pf(a["Rt"]), L.pb(a["ok"]), L.pb(a["acc"]))


@pytest.mark.parametrize("what", ["conf", "rtab", "aidx", "thresh", "reactive"])
def test_trimolecular_rejects_out_of_range_indices(what):
    a, keep = _tri_args()
    if what == "conf":
        a["conf"][0, 1] = 2  # only 2 conformers
    elif what == "rtab":
        a["rtab"][0, 5, 2, 0] = 3  # molecule 2 has 3 atoms
    elif what == "aidx":
        a["aidx"][1, 2] = 2  # U = 2
    elif what == "thresh":
        a["thresh"] = 0.0
    else:
        keep[1][0][1] = 9
    with pytest.raises(fc.FirecodeHipInputError):
        _call_tri(a)


def test_rot_corr_rejects_bad_torsions_and_masks():
    X = np.zeros((3, 5, 3))
    heavy = np.ones(5, dtype=np.uint8)
    mask = np.zeros(3, dtype=np.uint8)
    ang = np.zeros((1, 3))
    na = np.array([3], dtype=np.int32)
    rot = np.zeros((1, 5), dtype=np.uint8)

    def call(tors, heavy=heavy, na=na, max_rmsd=0.25):
        tors = np.array(tors, dtype=np.int64).reshape(-1, 4)
        L.call("fc_prune_rmsd_rot_corr", L.pf(X), 3, 5, L.pb(heavy), L.pi(tors), len(tors), L.pb(rot), L.pf(ang),
               na.ctypes.data_as(C.POINTER(C.c_int32)), 3, max_rmsd, 0.5, None, 0.0, 20, L.pb(mask), None)

    with pytest.raises(fc.FirecodeHipInputError):
        call([[0, 1, 2, 5]])  # atom index == A
    with pytest.raises(fc.FirecodeHipInputError):
        call([[0, 1, 2, 3]], heavy=np.zeros(5, dtype=np.uint8))  # no heavy atom selected
    with pytest.raises(fc.FirecodeHipInputError):
        call([[0, 1, 2, 3]], na=np.array([4], dtype=np.int32))  # more angles than max_angles
    with pytest.raises(fc.FirecodeHipInputError):
        call([[0, 1, 2, 3]], max_rmsd=-1.0)


def test_python_wrappers_reject_bad_shapes():
    with pytest.raises(fc.FirecodeHipInputError):
        fc.pruner.prune_by_rmsd(np.zeros((4, 5)), ["C"] * 5)
    with pytest.raises(fc.FirecodeHipInputError):
        fc.pruner.prune_by_rmsd(np.zeros((4, 5, 3)), ["C"] * 4)
    with pytest.raises(fc.FirecodeHipInputError):
        fc.pruner.prune_by_rmsd_rot_corr(np.zeros((4, 5, 3)), ["C"] * 5, None, torsions=[(0, 1, 2, 3, 3)])
    with pytest.raises(fc.FirecodeHipInputError):
        fc.hypermolecule_class.align_by_moi(["C"] * 4, np.zeros((2, 5, 3)))
    with pytest.raises(fc.FirecodeHipInputError):
        fc.embeds.rototranslate(np.zeros((2, 5, 3)), np.zeros((3, 3, 3)), np.zeros((2, 3)))
    with pytest.raises(fc.FirecodeHipInputError):
        fc.embeds.cyclical_embed_trimolecular([{}, {}], np.zeros((1, 3)))
    with pytest.raises(fc.FirecodeHipInputError):
        fc.torsion_module.torsion_scan(np.zeros((5, 3)), [[0, 1, 2, 3]], np.zeros((1, 4), dtype=bool), [[0]])


def test_c_entry_points_reject_null_and_bad_sizes():
    out = np.zeros((2, 5, 3))
    with pytest.raises(fc.FirecodeHipInputError):
        L.call("fc_align_by_moi", None, 2, 5, L.pf(np.ones(5)), L.pf(out))
    with pytest.raises(fc.FirecodeHipInputError):
        L.call("fc_align_by_moi", L.pf(out), 2, 0, L.pf(np.ones(5)), L.pf(out))
    with pytest.raises(fc.FirecodeHipInputError):
        L.call("fc_rototranslate", L.pf(out), 2, 5, None, None, L.pf(out))
    with pytest.raises(fc.FirecodeHipInputError):
        L.call("fc_prune_export_pairs_dev", None, None, 4)


def test_prune_many_rejects_bad_lists():
    one = (C.c_void_p * 1)(None)
    masks = (C.POINTER(C.c_uint8) * 1)()
    with pytest.raises(fc.FirecodeHipInputError):  # n out of range
        L.call("fc_prune_rmsd_many", one, 5000, 0.5, 1.0, 20, masks, None)
    with pytest.raises(fc.FirecodeHipInputError):  # NULL list
        L.call("fc_prune_rmsd_many", None, 1, 0.5, 1.0, 20, masks, None)
    with pytest.raises(fc.FirecodeHipInputError):  # threshold
        L.call("fc_prune_rmsd_many", one, 1, 0.0, 1.0, 20, masks, None)
    L.call("fc_prune_rmsd_many", None, 0, 0.5, 1.0, 20, None, None)  # empty queue: nothing to do
