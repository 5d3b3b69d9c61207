"""ORACLE (test infrastructure -- only tests/ may import this; the product is
firecode_amd/torsion_perception.py and never does): the torsion perception of
firecode/torsion_module.py restated LITERALLY, as the checker the product's own algorithm must equal.

* IN-TREE logic, statement by statement: ``Torsion`` (/root/reference/firecode/torsion_module.py:69-160:
  ``in_cycle``, ``is_rotable``, ``get_n_fold`` with ``mode="symmetry"``, ``get_angles``), ``_is_free``
  (:163-189), ``_is_nondummy`` (:192-269: ``deepcopy`` of the graph, edge removal, connected components,
  pairwise ``networkx.is_isomorphic``), ``get_quadruplets`` (:385-408: ``minimum_spanning_tree`` edge order,
  first-neighbour choice), ``_get_torsions`` (:411-433) and ``is_sp_n`` (graph_manipulations.py:109-114).
* THIRD-PARTY helpers those functions call -- ``get_sp_n``, ``is_amide_n``, ``is_ester_o``, ``get_phenyl_ids``
  (prism_pruner.graph_manipulations 0.0.7) and ``get_double_bonds_indices`` (prism_pruner.utils): not in the
  reference tree, PARITY UNPINNED, restated from what their call sites require.  The product carries its own
  table-driven form of the same rules; both are pinned together by ``tests/golden/make_golden_prism.py``
  whenever the package is importable.
"""

from copy import deepcopy

import numpy as np

__all__ = ["Torsion", "get_quadruplets", "get_torsions", "symmetric_torsions", "get_sp_n", "is_sp_n",
           "is_amide_n", "is_ester_o", "get_phenyl_ids", "get_double_bonds_indices", "graphize", "d_min_bond"]

# covalent radii in Angstrom (Cordero et al. 2008) for the elements organic / organometallic inputs hold
RADII_TABLE = {"H": 0.31, "B": 0.84, "C": 0.76, "N": 0.71, "O": 0.66, "F": 0.57, "Si": 1.11, "P": 1.07, "S": 1.05,
               "Cl": 1.02, "Br": 1.20, "I": 1.39, "Li": 1.28, "Na": 1.66, "K": 2.03, "Mg": 1.41, "Al": 1.21,
               "Se": 1.20, "Zn": 1.22, "Cu": 1.32, "Ni": 1.24, "Pd": 1.39, "Pt": 1.36, "Fe": 1.32, "Ru": 1.46,
               "Rh": 1.42, "Ir": 1.41, "Au": 1.36, "Ag": 1.45, "Co": 1.26, "Mn": 1.39, "Ti": 1.60, "Sn": 1.39}


def d_min_bond(e1, e2, factor=1.2):
    """prism_pruner.graph_manipulations.d_min_bond (call site firecode/utils.py:588): ``factor`` times
    the sum of the two covalent radii.  PARITY UNPINNED (radii table and default factor)."""
    return factor * (RADII_TABLE.get(str(e1), 1.5) + RADII_TABLE.get(str(e2), 1.5))


def graphize(atoms, coords, mask=None):
    """prism_pruner.graph_manipulations.graphize (call site firecode/ensemble.py:250): the bond graph
    of one structure -- an edge wherever two atoms are closer than ``d_min_bond`` -- with the element
    symbol as node attribute ``"atoms"`` (read at torsion_module.py:111,176).  PARITY UNPINNED."""
    import networkx as nx
    from scipy.spatial.distance import cdist

    atoms, coords = np.asarray(atoms), np.asarray(coords, dtype=np.float64)
    keep = np.ones(len(atoms), dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
    g = nx.Graph()
    for i in range(len(atoms)):
        g.add_node(i, atoms=str(atoms[i]))
    d = cdist(coords, coords)
    for i in range(len(atoms)):
        for j in range(i + 1, len(atoms)):
            if keep[i] and keep[j] and d[i, j] < d_min_bond(atoms[i], atoms[j]):
                g.add_edge(i, j)
    return g


# ------------------------------------------------------------------------------------------
# third-party helpers (prism_pruner.graph_manipulations / .utils) -- PARITY UNPINNED
# ------------------------------------------------------------------------------------------
def _symbol(graph, i):
    return graph.nodes[i]["atoms"]


def get_sp_n(index, graph):
    """Hybridisation number of a C / N / O / S atom from its coordination: 3 (sp3), 2 (sp2),
    1 (sp), None for anything else -- the domain the call sites need (``3 == sp_n_i2 == sp_n_i3``,
    ``sp_n_i3 or 2``, torsion_module.py:121-132; ``is_sp_n(index, graph, 2)``, :178)."""
    sym = _symbol(graph, index)
    n = len([x for x in graph.neighbors(index) if x != index])
    table = {"C": {4: 3, 3: 2, 2: 1}, "N": {4: 3, 3: 3, 2: 2, 1: 1}, "O": {2: 3, 1: 2}, "S": {4: 3, 3: 3, 2: 3, 1: 2},
             "P": {4: 3, 3: 3}, "Si": {4: 3}, "B": {3: 2, 4: 3}}
    return table.get(sym, {}).get(n)


def is_sp_n(index, graph, n):
    """firecode/graph_manipulations.py:109-114 (in-tree)."""
    return get_sp_n(index, graph) == n


def _carbonyl_neighbours(index, graph):
    """sp2 carbons bonded to ``index`` that carry a terminal oxygen (C=O)"""
    out = []
    for c in graph.neighbors(index):
        if c != index and _symbol(graph, c) == "C" and is_sp_n(c, graph, 2):
            if any(_symbol(graph, o) == "O" and len([x for x in graph.neighbors(o) if x != o]) == 1
                   for o in graph.neighbors(c)):
                out.append(c)
    return out


def is_amide_n(index, graph, mode=-1):
    """Nitrogen bonded to a carbonyl carbon.  mode 0: primary (CONH2), 1: secondary (CONHR),
    2: tertiary (CONR2), -1: any -- ``mode=1`` blocks rotation about the CO-NHR bond
    (torsion_module.py:183), ``mode=2`` makes tertiary amides 2-fold (:116)."""
    if _symbol(graph, index) != "N" or not _carbonyl_neighbours(index, graph):
        return False
    n_h = sum(1 for x in graph.neighbors(index) if x != index and _symbol(graph, x) == "H")
    if mode == -1:
        return True
    return {0: n_h == 2, 1: n_h == 1, 2: n_h == 0}.get(mode, False)


def is_ester_o(index, graph):
    """Bridging oxygen of an ester / carboxylic acid: two neighbours, one of them a carbonyl carbon."""
    if _symbol(graph, index) != "O":
        return False
    nb = [x for x in graph.neighbors(index) if x != index]
    return len(nb) == 2 and bool(_carbonyl_neighbours(index, graph))


def get_phenyl_ids(i, graph):
    """If atom ``i`` (the ipso position: two ring neighbours besides the torsion's root) sits on a
    six-membered ring of sp2 C / N atoms, the ring indices (i1..i6) starting at ``i`` and going
    around; None otherwise (torsion_module.py:220-224 unpacks six indices)."""
    import networkx as nx

    for cycle in nx.cycle_basis(graph, i):
        if len(cycle) == 6 and i in cycle and all(
                _symbol(graph, a) in ("C", "N") and get_sp_n(a, graph) == 2 for a in cycle):
            k = cycle.index(i)
            return tuple(cycle[k:] + cycle[:k])
    return None


_DOUBLE_BOND_MAX = {frozenset(("C", "C")): 1.40, frozenset(("C", "N")): 1.34, frozenset(("C", "O")): 1.28,
                    frozenset(("N", "N")): 1.30, frozenset(("N", "O")): 1.26, frozenset(("C", "S")): 1.68}


def get_double_bonds_indices(coords, atoms):
    """Sorted index pairs of bonds short enough to be double (C=C, C=N, C=O, N=N, N=O, C=S), the
    pairs ``_get_torsions`` refuses to rotate (torsion_module.py:680-683, :424)."""
    from scipy.spatial.distance import cdist

    coords, atoms = np.asarray(coords, dtype=np.float64), np.asarray(atoms)
    d = cdist(coords, coords)
    out = []
    for a in range(len(atoms)):
        for b in range(a + 1, len(atoms)):
            lim = _DOUBLE_BOND_MAX.get(frozenset((str(atoms[a]), str(atoms[b]))))
            if lim is not None and 0.0 < d[a, b] < lim:
                out.append((a, b))
    return out


# ------------------------------------------------------------------------------------------
# in-tree logic (firecode/torsion_module.py), restated
# ------------------------------------------------------------------------------------------
class Torsion:
    """firecode/torsion_module.py:69-160."""

    def __init__(self, i1, i2, i3, i4, mode=None):
        self.i1, self.i2, self.i3, self.i4 = int(i1), int(i2), int(i3), int(i4)
        self.torsion = (self.i1, self.i2, self.i3, self.i4)
        self.mode = mode

    def __repr__(self):
        if hasattr(self, "n_fold"):
            return f"Torsion({self.i1}, {self.i2}, {self.i3}, {self.i4}; {self.n_fold}-fold)"
        return f"Torsion({self.i1}, {self.i2}, {self.i3}, {self.i4})"

    def in_cycle(self, graph):
        from networkx import has_path

        graph.remove_edge(self.i2, self.i3)
        cyclical = bool(has_path(graph, self.i1, self.i4))
        graph.add_edge(self.i2, self.i3)
        return cyclical

    def is_rotable(self, graph, hydrogen_bonds, keepdummy=False):
        if tuple(sorted((self.i2, self.i3))) in hydrogen_bonds:
            return False
        if _is_free(self.i2, graph) or _is_free(self.i3, graph):
            if keepdummy or (_is_nondummy(self.i2, self.i3, graph) and _is_nondummy(self.i3, self.i2, graph)):
                self.n_fold = self.get_n_fold(graph)
                return True
        return False

    def get_n_fold(self, graph):
        symbols = (_symbol(graph, self.i2), _symbol(graph, self.i3))
        if "H" in symbols:
            return 6
        if is_amide_n(self.i2, graph, mode=2) or is_amide_n(self.i3, graph, mode=2):
            return 2
        if ("C" in symbols) or ("N" in symbols) or ("S" in symbols):
            sp_n_i2 = get_sp_n(self.i2, graph)
            sp_n_i3 = get_sp_n(self.i3, graph)
            if 3 == sp_n_i2 == sp_n_i3:
                return 3
            if 3 in (sp_n_i2, sp_n_i3):
                if self.mode == "csearch":
                    return 3
                elif self.mode == "symmetry":
                    return sp_n_i3 or 2
            if 2 in (sp_n_i2, sp_n_i3):
                return 2
        return 4

    def get_angles(self):
        return {2: (0, 180), 3: (0, 120, 240), 4: (0, 90, 180, 270), 6: (0, 60, 120, 180, 240, 300)}.get(self.n_fold)


def _is_free(index, graph):
    """firecode/torsion_module.py:163-189."""
    if all((_symbol(graph, index) == "C", is_sp_n(index, graph, 2),
            "O" in (_symbol(graph, n) for n in graph.neighbors(index)))):
        return False
    if is_amide_n(index, graph, mode=1):
        return False
    if is_ester_o(index, graph):
        return False
    return True


def _is_nondummy(i, root, graph):
    """firecode/torsion_module.py:192-269: False when rotating about (*, root, i, *) only permutes
    identical substituents of ``i`` (methyl, CF3, tBu, phenyl-like rings)."""
    from networkx import connected_components, is_isomorphic, subgraph

    if _symbol(graph, i) not in ("C", "N"):
        return True
    G = deepcopy(graph)
    nb = list(G.neighbors(i))
    if len(nb) == 1:
        # the reference calls len() on a neighbour iterator here (:214) -- a TypeError whenever the
        # branch is reached; an atom with a single bond cannot be the centre of a torsion built by
        # get_quadruplets, so the branch is dead and is kept only as the rule it states
        if len(list(G.neighbors(nb[0]))) == 2:
            return False
    if len(nb) == 2:
        phenyl_indices = get_phenyl_ids(i, G)
        if phenyl_indices is not None:
            i1, i2, i3, i4, i5, i6 = phenyl_indices
            G.remove_edge(i3, i4)
            G.remove_edge(i4, i5)
            G.remove_edge(i1, i2)
            G.remove_edge(i1, i6)
            subgraphs = [subgraph(G, _set) for _set in connected_components(G) if i2 in _set or i6 in _set]
            if len(subgraphs) == 2:
                return not is_isomorphic(subgraphs[0], subgraphs[1],
                                         node_match=lambda n1, n2: n1["atoms"] == n2["atoms"])
            return True
    for n in nb:
        G.remove_edge(i, n)
    subgraphs_nodes = [_set for _set in connected_components(G)
                       if root not in _set and any(n in _set for n in nb)]
    if len(subgraphs_nodes) == 1:
        return True
    subgraphs = [subgraph(G, s) for s in subgraphs_nodes]
    for sub in subgraphs[1:]:
        if not is_isomorphic(subgraphs[0], sub, node_match=lambda n1, n2: n1["atoms"] == n2["atoms"]):
            return True
    return False


def get_quadruplets(graph):
    """firecode/torsion_module.py:385-408."""
    from networkx import minimum_spanning_tree

    spanning_tree = minimum_spanning_tree(graph)
    dihedrals = []
    for i, j in spanning_tree.edges():
        i_neighbors = [n for n in graph.neighbors(i) if n not in (i, j)]
        j_neighbors = [n for n in graph.neighbors(j) if n not in (i, j)]
        if len(i_neighbors) > 0 and len(j_neighbors) > 0:
            dihedrals.append((i_neighbors[0], i, j, j_neighbors[0]))
    return np.array(dihedrals)


def get_torsions(graph, hydrogen_bonds=(), double_bonds=(), keepdummy=False, mode="csearch"):
    """``_get_torsions`` (firecode/torsion_module.py:411-433)."""
    torsions = []
    double_bonds = {tuple(sorted(b)) for b in double_bonds}
    for path in get_quadruplets(graph):
        _, i2, i3, _ = path
        if tuple(sorted((int(i2), int(i3)))) not in double_bonds:
            t = Torsion(*path, mode=mode)
            if (not t.in_cycle(graph)) and t.is_rotable(graph, hydrogen_bonds, keepdummy=keepdummy):
                torsions.append(t)
    return torsions


def symmetric_torsions(graph, coords=None, atoms=None):
    """The locally symmetric torsions of a molecule as ``(i1, i2, i3, i4, n_fold)``: rotatable,
    non-ring torsions (``keepdummy=True, mode="symmetry"``) about which at least one end only
    permutes identical substituents -- tBu, CF3, NMe2 (3-fold through an sp3 centre), phenyl-like
    rings and carboxylates (2-fold) -- oriented so that the symmetric end is the one that rotates
    (i4 side).  These are the groups docs/introduction.rst:105 names for ``prune_by_rmsd_rot_corr``;
    which torsions prism_pruner itself selects is not visible from the tree (PARITY UNPINNED)."""
    double_bonds = get_double_bonds_indices(coords, atoms) if coords is not None and atoms is not None else ()
    out = []
    for t in get_torsions(graph, (), double_bonds, keepdummy=True, mode="symmetry"):
        dummy_i3 = not _is_nondummy(t.i3, t.i2, graph)
        dummy_i2 = not _is_nondummy(t.i2, t.i3, graph)
        if not (dummy_i2 or dummy_i3):
            continue
        quad = t.torsion if dummy_i3 else tuple(reversed(t.torsion))
        n_fold = Torsion(*quad, mode="symmetry").get_n_fold(graph)
        if n_fold not in (2, 3, 4, 6):  # Torsion.get_angles knows these (:134-140)
            continue
        out.append((*quad, int(n_fold)))
    return out
