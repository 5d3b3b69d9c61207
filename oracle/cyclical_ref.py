"""CPU restatement of the cyclical embed drivers (trimolecular ``cyclical_embed`` and the
bimolecular ``_fast_bimol_rigid_cyclical_embed``, embeds.py:588-750) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module (same rule as ``oracle/cpu_ref.py``); the
product path is the HIP library and never routes through here.

Follows ``cyclical_embed`` (firecode/embeds.py:180-585) for three molecules,
line by line, with the Embedder / Hypermolecule objects replaced by the plain
data they carry.  In-tree logic (loop structure, ``_get_directions`` :187-260,
``_adjust_directions`` :262-407, the pose loop :465-569,
``_get_cyclical_reactive_indices`` :753-784, ``polygonize``, ``align_vec_pair``,
the trimolecular ``compenetration_check``) is literal.  Third-party pieces the
tree only imports -- ``rot_mat_from_pointer``, ``vec_angle``, ``normalize``
(prism_pruner.algebra 0.0.7), ``rmsd_and_max`` -- are restated from their
published form: PARITY UNPINNED for those, as in cpu_ref.py.
"""

from copy import deepcopy

import numpy as np

from oracle import cpu_ref as o


def normalize(v):
    """prism_pruner.algebra.normalize: v / |v|."""
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def vec_angle(v1, v2):
    """prism_pruner.algebra.vec_angle: angle in degrees between two vectors,
    arccos of the clipped dot product of the unit vectors."""
    return float(np.degrees(np.arccos(np.clip(np.dot(normalize(v1), normalize(v2)), -1.0, 1.0))))


class Pivot:
    """firecode/hypermolecule_class.py:300-333: ``pivot = start - end``,
    ``meanpoint = mean((start, end))``; ``start_cumnum`` / ``end_cumnum`` stand for
    ``start_atom.cumnum`` / ``end_atom.cumnum``."""

    def __init__(self, start, end, start_cumnum, end_cumnum):
        self.start = np.asarray(start, dtype=np.float64)
        self.end = np.asarray(end, dtype=np.float64)
        self.start_cumnum = int(start_cumnum)
        self.end_cumnum = int(end_cumnum)
        self.pivot = self.start - self.end
        self.meanpoint = np.mean((self.start, self.end), axis=0)


class Mol:
    """What the embed reads from a Hypermolecule: ``coords`` (n_conf, A, 3),
    ``reactive_indices`` (atom indices), ``pivots[conf]`` (list of Pivot),
    ``reactive_cumnums`` = {atom index: cumnum} of ``reactive_atoms_classes_dict[0]``."""

    def __init__(self, coords, reactive_indices, pivots, reactive_cumnums):
        self.coords = np.asarray(coords, dtype=np.float64)
        self.reactive_indices = np.asarray(reactive_indices, dtype=np.int64)
        self.pivots = pivots
        self.reactive_cumnums = dict(reactive_cumnums)
        self.rotation = np.eye(3)
        self.position = np.zeros(3)


def get_cyclical_reactive_indices(pivots, n):
    """embeds.py:753-784, trimolecular branch."""
    cumulative_pivots_ids = [[p.start_cumnum, p.end_cumnum] for p in pivots]
    swaps = [(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)]

    def orient(i, ids, n):
        if swaps[n][i]:
            return list(reversed(ids))
        return ids

    oriented = [orient(i, ids, n) for i, ids in enumerate(cumulative_pivots_ids)]
    couples = [
        (oriented[0][1], oriented[1][0]),
        (oriented[1][1], oriented[2][0]),
        (oriented[2][1], oriented[0][0]),
    ]
    return [tuple(sorted(c)) for c in couples]


def get_directions(norms):
    """embeds.py:187-260 (three norms).  May perturb ``norms[0]`` in place (:236)."""
    vertices = np.zeros((3, 2))
    vertices[1] = np.array([norms[0], 0])
    a = np.power(norms[0], 2)
    b = np.power(norms[1], 2)
    c = np.power(norms[2], 2)
    x = (a - b + c) / (2 * a**0.5)
    y = (c - x**2) ** 0.5
    vertices[2] = np.array([x, y])

    a = vertices[1, 0]
    b = vertices[2, 0]
    c = vertices[2, 1]
    x = a / 2
    y = (b**2 + c**2 - a * b) / (2 * c)
    cc = np.array([x, y])
    v0, v1, v2 = vertices
    meanpoint1 = np.mean((v0, v1), axis=0)
    meanpoint2 = np.mean((v1, v2), axis=0)
    meanpoint3 = np.mean((v2, v0), axis=0)
    dir1 = cc - meanpoint1
    dir2 = cc - meanpoint2
    dir3 = cc - meanpoint3
    if np.any([np.all(d == 0) for d in (dir1, dir2, dir3)]):
        norms[0] += 1e-5
        dir1, dir2, dir3 = [t[:-1] for t in get_directions(norms)]
    angle0_obtuse = vec_angle(v1 - v0, v2 - v0) > 90
    angle1_obtuse = vec_angle(v0 - v1, v2 - v1) > 90
    angle2_obtuse = vec_angle(v0 - v2, v1 - v2) > 90
    dir1 = -dir1 if angle2_obtuse else dir1
    dir2 = -dir2 if angle0_obtuse else dir2
    dir3 = -dir3 if angle1_obtuse else dir3
    dir1 = normalize(np.concatenate((dir1, [0])))
    dir2 = normalize(np.concatenate((dir2, [0])))
    dir3 = normalize(np.concatenate((dir3, [0])))
    return np.vstack((dir1, dir2, dir3))


def adjust_directions(objects, norms, directions, constrained_indices, triangle_vectors, pivots, conf_ids):
    """embeds.py:262-407.  ``norms`` is the enclosing scope's variable there."""
    assert directions.shape[0] == 3
    mols = deepcopy(objects)
    p0, p1, p2 = [end - start for start, end in triangle_vectors]
    p0_mean, p1_mean, p2_mean = [np.mean((end, start), axis=0) for start, end in triangle_vectors]

    vertices = np.zeros((3, 2))
    vertices[1] = np.array([norms[0], 0])
    a = np.power(norms[0], 2)
    b = np.power(norms[1], 2)
    c = np.power(norms[2], 2)
    x = (a - b + c) / (2 * a**0.5)
    y = (c - x**2) ** 0.5
    vertices[2] = np.array([x, y])
    v0, v1, v2 = vertices
    v0 = np.concatenate((v0, [0]))
    v1 = np.concatenate((v1, [0]))
    v2 = np.concatenate((v2, [0]))

    for i in (0, 1, 2):
        start, end = triangle_vectors[i]
        mol_direction = pivots[i].meanpoint - np.mean(
            objects[i].coords[conf_ids[i]][objects[i].reactive_indices], axis=0
        )
        if np.all(mol_direction == 0.0):
            mol_direction = pivots[i].meanpoint
        mols[i].rotation = o.align_vec_pair(
            np.array([end - start, directions[i]]), np.array([pivots[i].pivot, mol_direction])
        )
        mols[i].position = np.mean(triangle_vectors[i], axis=0) - mols[i].rotation @ pivots[i].meanpoint

    pairings = [[(-1, -1), (-1, -1)] for _ in constrained_indices]
    for i, c in enumerate(constrained_indices):
        for m, mol in enumerate(objects):
            for index, cumnum in mol.reactive_cumnums.items():
                if cumnum == c[0]:
                    pairings[i][0] = (m, index)
                if cumnum == c[1]:
                    pairings[i][1] = (m, index)
    r = np.zeros((3, 3), dtype=int)
    for first, second in pairings:
        r[first[0], second[0]] = first[1]
        r[second[0], first[0]] = second[1]

    mol0, mol1, mol2 = mols
    a01 = mol0.rotation @ mol0.coords[0][r[0, 1]] + mol0.position
    a02 = mol0.rotation @ mol0.coords[0][r[0, 2]] + mol0.position
    a10 = mol1.rotation @ mol1.coords[0][r[1, 0]] + mol1.position
    a12 = mol1.rotation @ mol1.coords[0][r[1, 2]] + mol1.position
    a20 = mol2.rotation @ mol2.coords[0][r[2, 0]] + mol2.position
    a21 = mol2.rotation @ mol2.coords[0][r[2, 1]] + mol2.position

    steps = 6
    angle_range = 30
    step_angle = 2 * angle_range / steps
    angles_list = o.cartesian_product(*[range(steps + 1) for _ in range(3)]) * step_angle - angle_range

    candidates = []
    for angles in angles_list:
        rot0 = o.rot_mat_from_pointer(p0, angles[0])
        new_a01 = rot0 @ a01
        new_a02 = rot0 @ a02
        d0 = p0_mean - np.mean((new_a01, new_a02), axis=0)
        rot1 = o.rot_mat_from_pointer(p1, angles[1])
        new_a10 = rot1 @ a10
        new_a12 = rot1 @ a12
        d1 = p1_mean - np.mean((new_a10, new_a12), axis=0)
        rot2 = o.rot_mat_from_pointer(p2, angles[2])
        new_a20 = rot2 @ a20
        new_a21 = rot2 @ a21
        d2 = p2_mean - np.mean((new_a20, new_a21), axis=0)
        cost = 0
        cost += vec_angle(v0 - new_a02, new_a20 - v0)
        cost += vec_angle(v1 - new_a01, new_a10 - v1)
        cost += vec_angle(v2 - new_a21, new_a12 - v2)
        candidates.append((cost, angles, (d0, d1, d2)))
    cost, angles, directions = sorted(candidates, key=lambda x: x[0])[0]
    return np.array(directions)


def cyclical_embed_trimolecular(objects, systematic_angles, pairings_table=None, internal_constraints=(),
                                clash_thresh=1.5, trace=None):
    """embeds.py:409-585 for ``len(embedder.objects) == 3``.
    Returns (poses (P, A1+A2+A3, 3), constrained_indices (P, 3, 2) int).  ``trace``, if a
    list, receives one (conf_ids, pivot_ids, v, directions, passed, accepted) per group."""
    ids_atoms = [m.coords.shape[1] for m in objects]
    conf_number = [len(mol.coords) for mol in objects]
    conf_indices = o.cartesian_product(*[np.array(range(i)) for i in conf_number])
    poses, constrained_indices = [], []
    for conf_ids in conf_indices:
        pivots_indices = o.cartesian_product(
            *[range(len(mol.pivots[conf_ids[i]])) for i, mol in enumerate(objects)]
        )
        for pi in pivots_indices:
            pivots = [objects[m].pivots[conf_ids[m]][pi[m]] for m, _ in enumerate(objects)]
            norms = np.linalg.norm(np.array([p.pivot for p in pivots]), axis=1)
            if all([norms[i] < norms[i - 1] + norms[i - 2] for i in (0, 1, 2)]):
                polygon_vectors = o.polygonize(norms)
            else:
                continue
            directions = get_directions(norms)
            for v, vecs in enumerate(polygon_vectors):
                ids = get_cyclical_reactive_indices(pivots, v)
                if not pairings_table or all(
                    (pair in ids) or (pair in list(internal_constraints)) for pair in pairings_table.values()
                ):
                    angular_poses = []
                    directions = adjust_directions(objects, norms, directions, ids, vecs, pivots, conf_ids)
                    passed, accepted = [], []
                    for angles in systematic_angles:
                        for i, vec_pair in enumerate(vecs):
                            start, end = vec_pair
                            angle = angles[i]
                            reactive_coords = objects[i].coords[conf_ids[i]][objects[i].reactive_indices]
                            atomic_pivot_mean = np.mean(reactive_coords, axis=0)
                            mol_direction = pivots[i].meanpoint - atomic_pivot_mean
                            if np.all(mol_direction == 0.0):
                                mol_direction = pivots[i].meanpoint
                            alignment_rotation = o.align_vec_pair(
                                np.array([end - start, directions[i]]),
                                np.array([pivots[i].pivot, mol_direction]),
                            )
                            if len(reactive_coords) == 2:
                                axis_of_step_rotation = alignment_rotation @ (reactive_coords[0] - reactive_coords[1])
                            else:
                                axis_of_step_rotation = alignment_rotation @ pivots[i].pivot
                            step_rotation = o.rot_mat_from_pointer(axis_of_step_rotation, angle)
                            center_of_rotation = alignment_rotation @ atomic_pivot_mean
                            objects[i].rotation = step_rotation @ alignment_rotation
                            pos = np.mean(vec_pair, axis=0) - alignment_rotation @ pivots[i].meanpoint
                            objects[i].position = center_of_rotation - step_rotation @ center_of_rotation + pos
                        embedded_structure = o.get_embed(
                            [m.coords[c] for m, c in zip(objects, conf_ids)],
                            [m.rotation for m in objects], [m.position for m in objects])
                        ok = o.compenetration_check(embedded_structure, ids=ids_atoms, thresh=clash_thresh)
                        new = False
                        if ok:
                            if not o.rmsd_similarity(embedded_structure, np.array(angular_poses), rmsd_thr=1):
                                poses.append(embedded_structure)
                                angular_poses.append(embedded_structure)
                                constrained_indices.append(ids)
                                new = True
                        passed.append(ok)
                        accepted.append(new)
                    if trace is not None:
                        trace.append((tuple(int(c) for c in conf_ids), tuple(int(p) for p in pi), v,
                                      np.array(directions), np.array(passed), np.array(accepted)))
    n_atoms = sum(ids_atoms)
    return (np.array(poses) if poses else np.empty((0, n_atoms, 3))), np.array(constrained_indices, dtype=np.int64).reshape(-1, 3, 2)


def get_cyclical_reactive_indices_bimol(pivots, n):
    """embeds.py:753-773, bimolecular branch (couples are NOT sorted here)."""
    cumulative_pivots_ids = [[p.start_cumnum, p.end_cumnum] for p in pivots]
    swaps = [(0, 0), (0, 1)]
    oriented = [list(reversed(ids)) if swaps[n][i] else ids for i, ids in enumerate(cumulative_pivots_ids)]
    return [(oriented[0][0], oriented[1][0]), (oriented[0][1], oriented[1][1])]


def cyclical_embed_bimolecular(objects, systematic_angles, pairings_table=None, internal_constraints=(),
                               clash_thresh=1.5, max_norm_delta=10.0, trace=None):
    """``_fast_bimol_rigid_cyclical_embed`` (embeds.py:588-750), literal, with plain data
    for the Embedder.  Returns (poses, constrained_indices (P, 2, 2))."""
    ids_atoms = [m.coords.shape[1] for m in objects]
    conf_number = [len(mol.coords) for mol in objects]
    conf_indices = o.cartesian_product(*[np.array(range(i)) for i in conf_number])
    poses, constrained_indices = [], []
    for conf_ids in conf_indices:
        pivots_indices = o.cartesian_product(
            *[range(len(mol.pivots[conf_ids[i]])) for i, mol in enumerate(objects)]
        )
        for pi in pivots_indices:
            pivots = [objects[m].pivots[conf_ids[m]][pi[m]] for m, _ in enumerate(objects)]
            norms = np.linalg.norm(np.array([p.pivot for p in pivots]), axis=1)
            if abs(norms[0] - norms[1]) > max_norm_delta:
                continue
            polygon_vectors = o.polygonize(norms)
            directions = np.array([[0, 1, 0], [0, -1, 0]])
            for v, vecs in enumerate(polygon_vectors):
                ids = get_cyclical_reactive_indices_bimol(pivots, v)
                if not pairings_table or all(
                    (pair in ids) or (pair in list(internal_constraints)) for pair in pairings_table.values()
                ):
                    angular_poses = []
                    n_acc = 0
                    for angles in systematic_angles:
                        for i, vec_pair in enumerate(vecs):
                            start, end = vec_pair
                            angle = angles[i]
                            reactive_coords = objects[i].coords[conf_ids[i]][objects[i].reactive_indices]
                            atomic_pivot_mean = np.mean(reactive_coords, axis=0)
                            mol_direction = pivots[i].meanpoint - atomic_pivot_mean
                            if np.all(mol_direction == 0.0):
                                mol_direction = pivots[i].meanpoint
                            alignment_rotation = o.align_vec_pair(
                                np.array([end - start, directions[i]]),
                                np.array([pivots[i].pivot, mol_direction]),
                            )
                            if len(reactive_coords) == 2:
                                axis_of_step_rotation = alignment_rotation @ (reactive_coords[0] - reactive_coords[1])
                            else:
                                axis_of_step_rotation = alignment_rotation @ pivots[i].pivot
                            step_rotation = o.rot_mat_from_pointer(axis_of_step_rotation, angle)
                            center_of_rotation = alignment_rotation @ atomic_pivot_mean
                            objects[i].rotation = step_rotation @ alignment_rotation
                            pos = np.mean(vec_pair, axis=0) - alignment_rotation @ pivots[i].meanpoint
                            objects[i].position = center_of_rotation - step_rotation @ center_of_rotation + pos
                        embedded_structure = o.get_embed(
                            [m.coords[c] for m, c in zip(objects, conf_ids)],
                            [m.rotation for m in objects], [m.position for m in objects])
                        if o.compenetration_check(embedded_structure, ids=ids_atoms, thresh=clash_thresh):
                            if not o.rmsd_similarity(embedded_structure, np.array(angular_poses), rmsd_thr=1):
                                poses.append(embedded_structure)
                                angular_poses.append(embedded_structure)
                                constrained_indices.append(ids)
                                n_acc += 1
                    if trace is not None:
                        trace.append((tuple(int(c) for c in conf_ids), tuple(int(p) for p in pi), v, n_acc))
    n_atoms = sum(ids_atoms)
    return ((np.array(poses) if poses else np.empty((0, n_atoms, 3))),
            np.array(constrained_indices, dtype=np.int64).reshape(-1, 2, 2))
