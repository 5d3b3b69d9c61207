"""The library's .xyz writer/reader against the Python formatting/parsing they
replace (CPU only: host code of libfc_hip.so)."""

import numpy as np
import pytest

from firecode_amd import _lib as L
from firecode_amd import ensemble as fe
from oracle import cpu_ref as o


def _tricky_values(rng, n):
    base = rng.normal(scale=30.0, size=n)
    ties = (rng.integers(-10**9, 10**9, size=n) + 0.5) * 1e-8          # decimal ties at 8 places (inexact in binary)
    ties6 = (rng.integers(-10**7, 10**7, size=n) + 0.5) * 1e-6
    exact = rng.integers(-2**20, 2**20, size=n) / 2.0**rng.integers(0, 30, size=n)  # exactly representable, some true ties
    small = rng.normal(size=n) * 10.0 ** rng.integers(-14, -5, size=n)
    big = rng.normal(size=n) * 10.0 ** rng.integers(3, 12, size=n)
    special = np.array([0.0, -0.0, 1e-9, -1e-9, 5e-9, -5e-9, 0.5e-8, 1.5e-8, 2.5e-8, 0.125, -0.375, 99999999.99999999,
                        -12345678.123456785, 3.9e7, 4.1e7, 1e15, -1e22, 0.000000005, 0.000000015])
    return np.concatenate([base, ties, ties6, exact, small, big, special])


def test_writer_is_byte_identical_to_python(tmp_path):
    rng = np.random.default_rng(0)
    vals = _tricky_values(rng, 4000)
    vals = vals[: len(vals) // 3 * 3].reshape(-1, 3)
    A = 7
    vals = vals[: len(vals) // A * A].reshape(-1, A, 3)
    atoms = np.array(["C", "H", "Cl", "N", "O", "Br", "Si"])
    path = tmp_path / "w.xyz"
    L.xyz_write(path, atoms, vals, label="golden", mode=0)
    assert path.read_text() == o.ensemble_to_xyz_text(atoms, vals, "golden")
    # utils.write_xyz format, one block per conformer
    L.xyz_write(path, atoms, vals[:50], label="temp", mode=1)
    expected = ""
    for c in vals[:50]:
        expected += str(len(c)) + "\ntemp\n"
        for atom, xyz in zip(atoms, c):
            expected += "%s     % .6f % .6f % .6f\n" % (atom, xyz[0], xyz[1], xyz[2])
    assert path.read_text() == expected


def test_reader_matches_python_float(tmp_path, golden):
    rng = np.random.default_rng(1)
    A, N = 5, 300
    lines = []
    truth = np.empty((N, A, 3))
    fmts = ["%.8f", "%.15g", "%.17g", "%e", "%.3f", "%d"]
    for n in range(N):
        lines.append(f"  {A} ")
        lines.append(f"comment {n} energy -12.5")
        for a in range(A):
            toks = []
            for c in range(3):
                v = rng.normal(scale=50.0) * 10.0 ** rng.integers(-6, 6)
                t = fmts[rng.integers(len(fmts))] % (int(v) if fmts[-1] == "%d" and False else v)
                if rng.random() < 0.1:
                    t = "%d" % int(v)
                toks.append(t)
                truth[n, a, c] = float(t)
            lines.append(("  " if a % 2 else "") + "C" + str(a) + "   " + "  ".join(toks) + ("  extra col" if a == 0 else ""))
        if n % 7 == 0:
            lines.append("")
    path = tmp_path / "r.xyz"
    path.write_text("\n".join(lines) + "\n")
    atoms, coords = L.xyz_read(path)
    assert np.array_equal(coords, truth)  # bit-identical to float()
    assert atoms.tolist() == [f"C{a}" for a in range(A)]
    # the reference's own writer output, read back (golden from the reference)
    p2 = tmp_path / "g.xyz"
    p2.write_text(str(golden["ens_text"]))
    a2, c2 = L.xyz_read(p2)
    assert np.array_equal(c2, golden["ens_back_coords"]) and np.array_equal(a2, golden["ens_back_atoms"])


def test_truncated_file_drops_partial_conformer(tmp_path):
    path = tmp_path / "t.xyz"
    path.write_text("2\nc\nH 0 0 0\nH 1 0 0\n2\nc\nH 0 0 1\n")
    atoms, coords = L.xyz_read(path)
    assert coords.shape == (1, 2, 3)
    with pytest.raises(L.FirecodeHipInputError):
        bad = tmp_path / "b.xyz"
        bad.write_text("2\nc\nH 0 0\nH 1 0 0\n")
        L.xyz_read(bad)


def test_ensemble_roundtrip_and_energies(tmp_path):
    rng = np.random.default_rng(2)
    X = rng.normal(scale=4.0, size=(6, 4, 3))
    ens = fe.Ensemble(atoms=np.array(["C", "H", "O", "H"]), coords=X, basename="rt", logfunction=None)
    p = tmp_path / "rt.xyz"
    ens.to_xyz(p)
    back = fe.Ensemble.from_xyz(p)
    assert np.array_equal(back.coords, np.array([[[float(f"{v:15.8f}") for v in row] for row in c] for c in X]))
    assert back.atomnos.tolist() == [6, 1, 8, 1]
    p.write_text("\n".join(f"2\nE = -{n}.250 Eh\nH 0 0 0\nH 0 0 {n}" for n in range(1, 4)))
    e = fe.Ensemble.from_xyz(p, read_energies=True)
    assert e.energies.tolist() == [-1.25, -2.25, -3.25] and e.coords.shape == (3, 2, 3)


def test_reads_the_reference_fixture_files_like_the_reference(golden, tmp_path):
    """the library's reader on the reference's own test data files (firecode/tests/**.xyz) against
    what the reference's Ensemble.from_xyz made of them; and its writer round-trips them"""
    for name in golden["fx_names"]:
        path = tmp_path / f"{name}.xyz"
        path.write_text(str(golden[f"fx_{name}_text"]))
        atoms, coords = L.xyz_read(str(path))
        assert list(atoms) == list(golden[f"fx_{name}_atoms"])
        assert np.array_equal(coords, golden[f"fx_{name}_coords"])
        out = tmp_path / f"{name}_out.xyz"
        L.xyz_write(str(out), atoms, coords, label=name, mode=0)
        atoms2, coords2 = L.xyz_read(str(out))
        assert list(atoms2) == list(atoms) and np.abs(coords2 - coords).max() < 1e-8
