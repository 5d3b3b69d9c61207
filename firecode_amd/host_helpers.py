"""O(1) host-side geometry of the embed set-up, with the reference's names.

These build the handful of 3-vectors / 3x3 matrices a pose grid is *described*
by (one call per molecule or per pivot pair, never per pose) -- the per-pose and
per-pair arithmetic is in the GPU kernels.  They are plain NumPy like their
reference counterparts and are not a fallback for anything batched."""

import numpy as np


def rot_mat_from_pointer(pointer, angle):
    """prism_pruner.algebra.rot_mat_from_pointer (call sites embeds.py:539,694;
    utils.py:246): rotation by ``angle`` degrees about ``pointer``."""
    pointer = np.asarray(pointer, dtype=np.float64)
    a2 = angle / 2 * np.pi / 180
    q1, q2, q3 = np.sin(a2) * pointer / np.linalg.norm(pointer)
    q0 = np.cos(a2)
    return np.array([
        [2 * (q0 * q0 + q1 * q1) - 1, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
        [2 * (q1 * q2 + q0 * q3), 2 * (q0 * q0 + q2 * q2) - 1, 2 * (q2 * q3 - q0 * q1)],
        [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 2 * (q0 * q0 + q3 * q3) - 1],
    ])


def rotation_matrix_from_vectors(vec1, vec2):
    """firecode/utils.py:224-249."""
    vec1, vec2 = np.asarray(vec1, dtype=np.float64), np.asarray(vec2, dtype=np.float64)
    assert vec1.shape == (3,) and vec2.shape == (3,)
    a, b = vec1 / np.linalg.norm(vec1), vec2 / np.linalg.norm(vec2)
    v = np.cross(a, b)
    if np.linalg.norm(v) != 0:
        c, s = np.dot(a, b), np.linalg.norm(v)
        kmat = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        return np.eye(3) + kmat + kmat.dot(kmat) * ((1 - c) / (s**2))
    if np.linalg.norm(a + b) == 0:
        return rot_mat_from_pointer(np.array([0, 0, 1]), 180)
    return np.eye(3)


def polygonize(lengths):
    """firecode/utils.py:252-312: start/end points of the pivot vectors
    (bimolecular: two centred segments, second orientation flips molecule 2;
    trimolecular: the eight triangle arrangements)."""
    assert len(lengths) in (2, 3)
    arr = np.zeros((len(lengths), 2, 3))
    if len(lengths) == 2:
        arr[0, 0] = [-lengths[0] / 2, 0, 0]
        arr[0, 1] = [+lengths[0] / 2, 0, 0]
        arr[1, 0] = [-lengths[1] / 2, 0, 0]
        arr[1, 1] = [+lengths[1] / 2, 0, 0]
        out = np.vstack(([arr], [arr]))
        out[1, 1] *= -1
        return out
    if not all(lengths[i] < lengths[i - 1] + lengths[i - 2] for i in (0, 1, 2)):
        raise ValueError(f"Impossible to build a triangle with sides {lengths}")
    arr[0, 1] = [lengths[0], 0, 0]
    arr[1, 0] = [lengths[0], 0, 0]
    a, b, c = (np.power(lengths[i], 2) for i in range(3))
    x = (a - b + c) / (2 * a**0.5)
    y = (c - x**2) ** 0.5
    arr[1, 1] = [x, y, 0]
    arr[2, 0] = [x, y, 0]
    out = np.vstack([[arr]] * 8)
    for t, v in [(1, 2), (2, 1), (3, 1), (3, 2), (4, 0), (5, 0), (5, 1), (6, 0), (6, 2), (7, 0), (7, 1), (7, 2)]:
        out[t, v][[0, 1]] = out[t, v][[1, 0]]
    return out


def systematic_angles(rotation_steps=5, rotation_range=45.0):
    """The per-molecule step-angle grid of a cyclical embed
    (firecode/embedder.py:1090-1098): ``steps + 1`` values in [-range, +range];
    the reference's angle *pairs* are ``cartesian_product`` of this grid with
    itself, i.e. the (a2, a1) axes of ``embeds.embed_grid_clash``."""
    return np.arange(rotation_steps + 1) * 2 * rotation_range / rotation_steps - rotation_range


def cyclical_reactive_indices(cum_ids_1, cum_ids_2, orientation):
    """``_get_cyclical_reactive_indices`` (embeds.py:753-784), bimolecular branch:
    cumulative atom indices (start, end) of each molecule's pivot -> the two
    index couples to constrain for polygon orientation 0 / 1."""
    swaps = [(0, 0), (0, 1)]
    o1 = list(reversed(cum_ids_1)) if swaps[orientation][0] else list(cum_ids_1)
    o2 = list(reversed(cum_ids_2)) if swaps[orientation][1] else list(cum_ids_2)
    return [(o1[0], o2[0]), (o1[1], o2[1])]


def _vec_angle(v1, v2):
    """prism_pruner.algebra.vec_angle: degrees between two vectors."""
    u1, u2 = np.asarray(v1, float), np.asarray(v2, float)
    u1, u2 = u1 / np.linalg.norm(u1), u2 / np.linalg.norm(u2)
    return float(np.degrees(np.arccos(np.clip(np.dot(u1, u2), -1.0, 1.0))))


def triangle_directions(norms):
    """``_get_directions`` of ``cyclical_embed`` for three molecules (embeds.py:187-260):
    unit vectors, in the plane of the pivot triangle, from the middle of each side towards
    the circumcentre (sign flipped for the two sides adjacent to an obtuse angle's opposite
    vertex, as the reference does).  May nudge ``norms[0]`` by 1e-5 in place for a right
    triangle (:236), like the reference -- the caller's later uses see the nudged value."""
    n0 = norms[0]
    a, b, c = n0**2, norms[1] ** 2, norms[2] ** 2
    x = (a - b + c) / (2 * a**0.5)
    y = (c - x**2) ** 0.5
    v0, v1, v2 = np.zeros(2), np.array([n0, 0.0]), np.array([x, y])
    # circumcentre of (0,0), (n0,0), (x,y)
    cc = np.array([n0 / 2, (x**2 + y**2 - n0 * x) / (2 * y)])
    dirs = [cc - np.mean((v0, v1), axis=0), cc - np.mean((v1, v2), axis=0), cc - np.mean((v2, v0), axis=0)]
    if any(np.all(d == 0) for d in dirs):
        norms[0] += 1e-5
        dirs = [t[:-1] for t in triangle_directions(norms)]
    obtuse0 = _vec_angle(v1 - v0, v2 - v0) > 90
    obtuse1 = _vec_angle(v0 - v1, v2 - v1) > 90
    obtuse2 = _vec_angle(v0 - v2, v1 - v2) > 90
    flips = (obtuse2, obtuse0, obtuse1)
    out = np.zeros((3, 3))
    for k in range(3):
        d = -dirs[k] if flips[k] else dirs[k]
        d3 = np.array([d[0], d[1], 0.0])
        out[k] = d3 / np.linalg.norm(d3)
    return out


_TRI_SWAPS = ((0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1))


def cyclical_reactive_indices_tri(cum_ids, orientation):
    """``_get_cyclical_reactive_indices`` (embeds.py:753-784), trimolecular branch:
    ``cum_ids`` = three (start, end) cumulative atom numbers -> three sorted couples."""
    oriented = [tuple(reversed(ids)) if _TRI_SWAPS[orientation][i] else tuple(ids) for i, ids in enumerate(cum_ids)]
    couples = ((oriented[0][1], oriented[1][0]), (oriented[1][1], oriented[2][0]), (oriented[2][1], oriented[0][0]))
    return [tuple(sorted(int(x) for x in c)) for c in couples]
