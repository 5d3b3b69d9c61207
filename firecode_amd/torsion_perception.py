"""Host-side perception of the locally symmetric torsions a rotationally corrected RMSD prune
needs, so that ``prune_by_rmsd_rot_corr(structures, atoms, graph, max_rmsd=...)`` works with
exactly the arguments the reference passes (firecode/ensemble.py:253-260, embedder.py:1489-1496).

What has to come out is fixed by the reference (firecode/torsion_module.py:69-269, 385-433: which
bonds are torsions, which are rotatable, which ends are "dummy", the n-fold of each); HOW it is
computed here is this package's own:

* the molecule is analysed ONCE into flat per-atom tables (``MolGraph``: symbols, adjacency lists in
  the graph's own order, coordination-derived hybridisation, carbonyl / amide / ester flags, the set
  of bridge bonds from one low-link depth-first search) instead of per-call graph copies;
* "rotating about this bond only permutes identical substituents" is decided by comparing the
  branches hanging off the axis atom through COLOUR REFINEMENT of the labelled branch graphs
  (1-dimensional Weisfeiler-Leman, complete for the tree-shaped branches of ordinary molecules) with
  an individualise-and-refine search behind it for branches that contain rings -- an exact
  isomorphism test without copying or mutating the caller's graph;
* ring membership of a bond is bridge detection (a bond is on a cycle iff it is not a bridge);
* the spanning-tree walk that defines the order of the torsions (an ordering contract, SURVEY
  Appendix C) is a union-find pass over the edges in the graph's iteration order;
* the n-fold is a table lookup on (hybridisation of i2, hybridisation of i3, mode).

``oracle/torsion_perception_ref.py`` holds the literal restatement of the reference's functions;
``tests/test_torsion_perception.py`` checks that both agree on the reference's fixture molecules and
on random molecular graphs.  The chemistry helpers the reference imports from ``prism_pruner``
(``get_sp_n``, ``is_amide_n``, ``is_ester_o``, ``get_phenyl_ids``, ``get_double_bonds_indices``,
``graphize``, ``d_min_bond``) are not in the reference tree: PARITY UNPINNED, restated from what
their call sites require.

Graph work on one molecule (tens of atoms, once per prune) stays on the host, like the reference --
nothing here is on the data-parallel path.
"""

import numpy as np

__all__ = ["MolGraph", "Torsion", "get_quadruplets", "get_torsions", "symmetric_torsions", "get_sp_n", "is_sp_n",
           "is_amide_n", "is_ester_o", "get_phenyl_ids", "get_double_bonds_indices", "graphize", "d_min_bond"]

# covalent radii in Angstrom (Cordero et al. 2008) for the elements organic / organometallic inputs hold
RADII_TABLE = {"H": 0.31, "B": 0.84, "C": 0.76, "N": 0.71, "O": 0.66, "F": 0.57, "Si": 1.11, "P": 1.07, "S": 1.05,
               "Cl": 1.02, "Br": 1.20, "I": 1.39, "Li": 1.28, "Na": 1.66, "K": 2.03, "Mg": 1.41, "Al": 1.21,
               "Se": 1.20, "Zn": 1.22, "Cu": 1.32, "Ni": 1.24, "Pd": 1.39, "Pt": 1.36, "Fe": 1.32, "Ru": 1.46,
               "Rh": 1.42, "Ir": 1.41, "Au": 1.36, "Ag": 1.45, "Co": 1.26, "Mn": 1.39, "Ti": 1.60, "Sn": 1.39}

# hybridisation number by element and coordination: 3 (sp3), 2 (sp2), 1 (sp); anything else None -- the
# domain the call sites need (``3 == sp_n_i2 == sp_n_i3``, ``sp_n_i3 or 2``, torsion_module.py:121-132)
_SP_TABLE = {"C": {4: 3, 3: 2, 2: 1}, "N": {4: 3, 3: 3, 2: 2, 1: 1}, "O": {2: 3, 1: 2}, "S": {4: 3, 3: 3, 2: 3, 1: 2},
             "P": {4: 3, 3: 3}, "Si": {4: 3}, "B": {3: 2, 4: 3}}

_DOUBLE_BOND_MAX = {frozenset(("C", "C")): 1.40, frozenset(("C", "N")): 1.34, frozenset(("C", "O")): 1.28,
                    frozenset(("N", "N")): 1.30, frozenset(("N", "O")): 1.26, frozenset(("C", "S")): 1.68}

_ANGLES = {2: (0, 180), 3: (0, 120, 240), 4: (0, 90, 180, 270), 6: (0, 60, 120, 180, 240, 300)}


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
    radii = np.array([RADII_TABLE.get(str(a), 1.5) for a in atoms])
    bonded = cdist(coords, coords) < 1.2 * (radii[:, None] + radii[None, :])
    bonded &= keep[:, None] & keep[None, :]
    g = nx.Graph()
    g.add_nodes_from((i, {"atoms": str(a)}) for i, a in enumerate(atoms))
    g.add_edges_from(zip(*np.nonzero(np.triu(bonded, 1))))
    return g


def get_double_bonds_indices(coords, atoms):
    """Sorted index pairs of bonds short enough to be double (C=C, C=N, C=O, N=N, N=O, C=S), the
    pairs ``_get_torsions`` refuses to rotate (torsion_module.py:680-683, :424).  PARITY UNPINNED."""
    from scipy.spatial.distance import cdist

    coords = np.asarray(coords, dtype=np.float64)
    atoms = [str(a) for a in np.asarray(atoms)]
    limit = np.array([[_DOUBLE_BOND_MAX.get(frozenset((a, b)), 0.0) for b in atoms] for a in atoms])
    d = cdist(coords, coords)
    ia, ib = np.nonzero(np.triu((d > 0.0) & (d < limit), 1))
    return [(int(a), int(b)) for a, b in zip(ia, ib)]


# ------------------------------------------------------------------------------------------
# the molecule as flat tables
# ------------------------------------------------------------------------------------------
class MolGraph:
    """One pass over a bond graph (anything with networkx's ``nodes`` / ``adj`` mappings and an
    ``"atoms"`` node attribute) -> per-atom tables every question below is answered from.  Node keys
    are kept as they are (the reference's graphs use 0..A-1); adjacency lists keep the graph's own
    neighbour order, which the torsion order depends on."""

    def __init__(self, graph):
        self.nodes = list(graph.nodes)
        self.sym = {n: str(graph.nodes[n]["atoms"]) for n in self.nodes}
        self.adj = {n: list(graph.adj[n]) for n in self.nodes}           # raw: a self-loop stays listed
        self.nbr = {n: [x for x in self.adj[n] if x != n] for n in self.nodes}
        self.sp = {n: _SP_TABLE.get(self.sym[n], {}).get(len(self.nbr[n])) for n in self.nodes}
        sym, nbr, sp = self.sym, self.nbr, self.sp
        # sp2 carbons that carry a terminal oxygen (C=O)
        carbonyl_c = {c for c in self.nodes if sym[c] == "C" and sp[c] == 2
                      and any(sym[o] == "O" and len(nbr[o]) == 1 for o in self.adj[c])}
        self.on_carbonyl = {n: any(c in carbonyl_c for c in nbr[n]) for n in self.nodes}
        self.n_h = {n: sum(1 for x in nbr[n] if sym[x] == "H") for n in self.nodes}
        self._bridges = None

    # ---- chemistry flags (prism_pruner.graph_manipulations helpers; PARITY UNPINNED) ----
    def amide_n(self, n, mode=-1):
        if self.sym[n] != "N" or not self.on_carbonyl[n]:
            return False
        return True if mode == -1 else {0: 2, 1: 1, 2: 0}.get(mode, -1) == self.n_h[n]

    def ester_o(self, n):
        return self.sym[n] == "O" and len(self.nbr[n]) == 2 and self.on_carbonyl[n]

    def free(self, n):
        """firecode/torsion_module.py:163-189: carbonyl-like sp2 carbons, secondary amide nitrogens and
        ester oxygens do not rotate freely."""
        if self.sym[n] == "C" and self.sp[n] == 2 and any(self.sym[x] == "O" for x in self.adj[n]):
            return False
        return not (self.amide_n(n, 1) or self.ester_o(n))

    # ---- topology ----
    def bridges(self):
        """Bonds on no cycle, by one low-link depth-first search (iterative): an edge (p, v) of the search
        tree is a bridge iff nothing below v reaches p or above."""
        if self._bridges is not None:
            return self._bridges
        order, low, out = {}, {}, set()
        for root in self.nodes:
            if root in order:
                continue
            order[root] = low[root] = len(order)
            stack = [(root, None, iter(self.nbr[root]))]
            while stack:
                v, parent, it = stack[-1]
                advanced = False
                for w in it:
                    if w not in order:
                        order[w] = low[w] = len(order)
                        stack.append((w, v, iter(self.nbr[w])))
                        advanced = True
                        break
                    if w != parent:
                        low[v] = min(low[v], order[w])
                if advanced:
                    continue
                stack.pop()
                if parent is not None:
                    low[parent] = min(low[parent], low[v])
                    if low[v] > order[parent]:
                        out.add(frozenset((parent, v)))
        self._bridges = out
        return out

    def components(self, cut_atom=None, cut_edges=()):
        """Connected components (node lists, in node order) of the graph without the bonds of ``cut_atom``
        and without ``cut_edges`` (frozensets)."""
        seen, comps = set(), []
        for s in self.nodes:
            if s in seen:
                continue
            comp, todo = [], [s]
            seen.add(s)
            while todo:
                v = todo.pop()
                comp.append(v)
                if v == cut_atom:
                    continue
                for w in self.nbr[v]:
                    if w == cut_atom or w in seen or frozenset((v, w)) in cut_edges:
                        continue
                    seen.add(w)
                    todo.append(w)
            comps.append(comp)
        return comps

    def edge_order(self, adj=None, nodes=None):
        """Edges in the order an undirected networkx graph reports them: node by node, each node's
        neighbours in adjacency order, an edge once (at its first end)."""
        adj = self.adj if adj is None else adj
        done, out = set(), []
        for u in (self.nodes if nodes is None else nodes):
            for v in adj[u]:
                if v not in done:
                    out.append((u, v))
            done.add(u)
        return out

    # ---- branch comparison ----
    def _initial_colour(self, v):
        return (self.sym[v], v in self.adj[v])

    def isomorphic(self, part_a, part_b, cut_edges=()):
        """Exact isomorphism of the element-labelled subgraphs induced on two disjoint node sets (bonds in
        ``cut_edges`` do not exist): colour refinement on their union; where it leaves classes with more
        than one atom per side, individualise one atom and refine again, trying each candidate image."""
        if len(part_a) != len(part_b):
            return False
        in_a, in_b = set(part_a), set(part_b)
        verts = list(part_a) + list(part_b)
        inside = {v: in_a if v in in_a else in_b for v in verts}
        local = {v: [w for w in self.nbr[v] if w in inside[v] and frozenset((v, w)) not in cut_edges] for v in verts}
        if sum(len(local[v]) for v in part_a) != sum(len(local[v]) for v in part_b):
            return False
        palette = {}
        colour = {v: palette.setdefault(self._initial_colour(v), len(palette)) for v in verts}
        return self._match(colour, local, part_a, part_b)

    @staticmethod
    def _refine(colour, local):
        n_classes = len(set(colour.values()))
        while True:
            sig = {v: (colour[v], tuple(sorted(colour[w] for w in local[v]))) for v in colour}
            rank = {s: k for k, s in enumerate(sorted(set(sig.values())))}
            colour = {v: rank[sig[v]] for v in colour}
            if len(rank) == n_classes:
                return colour
            n_classes = len(rank)

    def _match(self, colour, local, part_a, part_b):
        colour = self._refine(colour, local)
        cells_a, cells_b = {}, {}
        for v in part_a:
            cells_a.setdefault(colour[v], []).append(v)
        for v in part_b:
            cells_b.setdefault(colour[v], []).append(v)
        if {c: len(m) for c, m in cells_a.items()} != {c: len(m) for c, m in cells_b.items()}:
            return False
        open_cells = [c for c, m in cells_a.items() if len(m) > 1]
        if not open_cells:  # discrete: the colouring IS the map -- check that it carries bonds to bonds
            image = {cells_a[c][0]: cells_b[c][0] for c in cells_a}
            return all(sorted(image[w] for w in local[v]) == sorted(local[image[v]]) for v in part_a)
        cell = min(open_cells, key=lambda c: (len(cells_a[c]), c))
        fresh = max(colour.values()) + 1
        pivot = cells_a[cell][0]
        for cand in cells_b[cell]:
            trial = dict(colour)
            trial[pivot] = trial[cand] = fresh
            if self._match(trial, local, part_a, part_b):
                return True
        return False

    def six_ring(self, i):
        """A six-membered ring of sp2 C / N atoms through ``i`` as (i, i2, .., i6) going around, or None
        (prism_pruner's get_phenyl_ids as torsion_module.py:220-224 unpacks it).  PARITY UNPINNED."""
        def aromatic(a):
            return self.sym[a] in ("C", "N") and self.sp[a] == 2

        if not aromatic(i):
            return None
        path = [i]

        def walk(v):
            if len(path) == 6:
                return i in self.nbr[v]
            for w in self.nbr[v]:
                if w not in path and aromatic(w):
                    path.append(w)
                    if walk(w):
                        return True
                    path.pop()
            return False

        return tuple(path) if walk(i) else None

    def nondummy(self, i, root):
        """firecode/torsion_module.py:192-269: False when rotating about (*, root, i, *) only permutes
        identical substituents of ``i`` (methyl, CF3, tBu, NMe2, ring halves)."""
        if self.sym[i] not in ("C", "N"):
            return True
        bonds = self.adj[i]  # the root stays among them (the reference's `# nb.remove(root)`, :209)
        if len(bonds) == 1 and len(self.adj[bonds[0]]) == 2:
            return False
        if len(bonds) == 2:
            ring = self.six_ring(i)
            if ring is not None:
                i1, i2, i3, i4, i5, i6 = ring
                cut = {frozenset(e) for e in ((i3, i4), (i4, i5), (i1, i2), (i1, i6))}
                halves = [c for c in self.components(cut_edges=cut) if i2 in c or i6 in c]
                return not self.isomorphic(halves[0], halves[1], cut) if len(halves) == 2 else True
        branches = [c for c in self.components(cut_atom=i) if root not in c and any(n in c for n in bonds)]
        if len(branches) == 1:
            return True
        return any(not self.isomorphic(branches[0], b) for b in branches[1:])


def _mol(graph):
    return graph if isinstance(graph, MolGraph) else MolGraph(graph)


# ---- the reference's free-standing helper names, answered from the tables ------------------------
def get_sp_n(index, graph):
    """Hybridisation number of an atom from its coordination: 3, 2, 1 or None (PARITY UNPINNED)."""
    return _mol(graph).sp[index]


def is_sp_n(index, graph, n):
    """firecode/graph_manipulations.py:109-114."""
    return _mol(graph).sp[index] == n


def is_amide_n(index, graph, mode=-1):
    """Nitrogen bonded to a carbonyl carbon.  mode 0: primary (CONH2), 1: secondary (CONHR), 2: tertiary
    (CONR2), -1: any -- ``mode=1`` blocks rotation about the CO-NHR bond (torsion_module.py:183),
    ``mode=2`` makes tertiary amides 2-fold (:116).  PARITY UNPINNED."""
    return _mol(graph).amide_n(index, mode)


def is_ester_o(index, graph):
    """Bridging oxygen of an ester / carboxylic acid.  PARITY UNPINNED."""
    return _mol(graph).ester_o(index)


def get_phenyl_ids(i, graph):
    return _mol(graph).six_ring(i)


def _is_free(index, graph):
    return _mol(graph).free(index)


def _is_nondummy(i, root, graph):
    return _mol(graph).nondummy(i, root)


# ------------------------------------------------------------------------------------------
# torsions
# ------------------------------------------------------------------------------------------
def _n_fold_table(mode):
    """(sp_n of i2, sp_n of i3) -> n-fold for bonds between C / N / S atoms (torsion_module.py:118-132)."""
    table = {}
    for a in (None, 1, 2, 3):
        for b in (None, 1, 2, 3):
            if a == 3 and b == 3:
                n = 3
            elif 3 in (a, b) and mode == "csearch":
                n = 3
            elif 3 in (a, b) and mode == "symmetry":
                n = b or 2
            elif 2 in (a, b):
                n = 2
            else:
                n = 4
            table[(a, b)] = n
    return table


_N_FOLD = {mode: _n_fold_table(mode) for mode in ("csearch", "symmetry", None)}


class Torsion:
    """The record firecode/torsion_module.py:69-160 passes around: four atom indices, the n-fold once known,
    the scan angles.  The questions it answers are looked up in a ``MolGraph`` (a networkx graph is
    analysed on the fly)."""

    def __init__(self, i1, i2, i3, i4, mode=None):
        self.i1, self.i2, self.i3, self.i4 = int(i1), int(i2), int(i3), int(i4)
        self.torsion = (self.i1, self.i2, self.i3, self.i4)
        self.mode = mode

    def __repr__(self):
        fold = f"; {self.n_fold}-fold" if hasattr(self, "n_fold") else ""
        return f"Torsion({self.i1}, {self.i2}, {self.i3}, {self.i4}{fold})"

    def in_cycle(self, graph):
        return frozenset((self.i2, self.i3)) not in _mol(graph).bridges()

    def is_rotable(self, graph, hydrogen_bonds, keepdummy=False):
        m = _mol(graph)
        if tuple(sorted((self.i2, self.i3))) in hydrogen_bonds:
            return False
        if not (m.free(self.i2) or m.free(self.i3)):
            return False
        if not keepdummy and not (m.nondummy(self.i2, self.i3) and m.nondummy(self.i3, self.i2)):
            return False
        self.n_fold = self.get_n_fold(m)
        return True

    def get_n_fold(self, graph):
        m = _mol(graph)
        ends = {m.sym[self.i2], m.sym[self.i3]}
        if "H" in ends:
            return 6
        if m.amide_n(self.i2, 2) or m.amide_n(self.i3, 2):
            return 2
        if ends & {"C", "N", "S"}:
            return _N_FOLD.get(self.mode, _N_FOLD[None])[(m.sp[self.i2], m.sp[self.i3])]
        return 4

    def get_angles(self):
        return _ANGLES.get(self.n_fold)


def get_quadruplets(graph):
    """firecode/torsion_module.py:385-408: one dihedral per bond of a spanning tree whose both ends have a
    further neighbour.  The spanning tree and its edge order are those networkx's Kruskal gives on equal
    weights: bonds accepted in the graph's edge order while they join two trees, then reported node by
    node."""
    m = _mol(graph)
    parent = {n: n for n in m.nodes}

    def find(v):
        while parent[v] != v:
            parent[v] = parent[parent[v]]
            v = parent[v]
        return v

    tree = {n: [] for n in m.nodes}
    for u, v in m.edge_order():
        ru, rv = find(u), find(v)
        if ru != rv:
            parent[ru] = rv
            tree[u].append(v)
            tree[v].append(u)
    out = []
    for i, j in m.edge_order(adj=tree):
        left = next((n for n in m.adj[i] if n not in (i, j)), None)
        right = next((n for n in m.adj[j] if n not in (i, j)), None)
        if left is not None and right is not None:
            out.append((left, i, j, right))
    return np.array(out)


def get_torsions(graph, hydrogen_bonds=(), double_bonds=(), keepdummy=False, mode="csearch"):
    """``_get_torsions`` (firecode/torsion_module.py:411-433): the rotatable, non-ring, non-double bonds."""
    m = _mol(graph)
    double_bonds = {tuple(sorted(b)) for b in double_bonds}
    out = []
    for quad in get_quadruplets(m):
        if tuple(sorted((int(quad[1]), int(quad[2])))) in double_bonds:
            continue
        t = Torsion(*quad, mode=mode)
        if not t.in_cycle(m) and t.is_rotable(m, hydrogen_bonds, keepdummy=keepdummy):
            out.append(t)
    return out


def symmetric_torsions(graph, coords=None, atoms=None):
    """The locally symmetric torsions of a molecule as ``(i1, i2, i3, i4, n_fold)``: rotatable,
    non-ring torsions (``keepdummy=True, mode="symmetry"``) about which at least one end only
    permutes identical substituents -- tBu, CF3, NMe2 (3-fold through an sp3 centre), phenyl-like
    rings and carboxylates (2-fold) -- oriented so that the symmetric end is the one that rotates
    (i4 side).  These are the groups docs/introduction.rst:105 names for ``prune_by_rmsd_rot_corr``;
    which torsions prism_pruner itself selects is not visible from the tree (PARITY UNPINNED)."""
    m = _mol(graph)
    double_bonds = get_double_bonds_indices(coords, atoms) if coords is not None and atoms is not None else ()
    out = []
    for t in get_torsions(m, (), double_bonds, keepdummy=True, mode="symmetry"):
        if not m.nondummy(t.i3, t.i2):
            quad = t.torsion
        elif not m.nondummy(t.i2, t.i3):
            quad = tuple(reversed(t.torsion))
        else:
            continue
        n_fold = Torsion(*quad, mode="symmetry").get_n_fold(m)
        if n_fold in _ANGLES:  # Torsion.get_angles knows these (:134-140)
            out.append((*quad, int(n_fold)))
    return out
