"""A ``gpu_prune>`` operator with FIRECODE's operator contract
(firecode/operators.py:63-135: ``f(filename: str, embedder) -> str``: read
``embedder.mols[filename]``, log through ``embedder.log``, write an ``.xyz``,
return its name) -- SURVEY.md section 8f rank 3.

Wiring it into the reference is one ``case`` in ``operate``::

    case "gpu_prune":
        outname = firecode_amd.operators.gpu_prune_operator(filename, embedder)
"""

from time import perf_counter

import numpy as np

from firecode_amd.pruner import prune_by_rmsd_rot_corr, prune_similarity
from firecode_amd.utils import write_xyz


def gpu_prune_operator(filename, embedder, moi=True, rmsd=True, rmsd_rot_corr=True):
    """Similarity-prune the ensemble of ``filename`` on the GPU with the triplet the search operators end
    with (firecode/operators.py:613-632): moment of inertia -> heavy-atom RMSD at ``options.rmsd`` -- both on
    ONE upload of the coordinates (``prune_similarity``), without the reference's 5e4 cap on the RMSD stage --
    then, below 1000 structures as there and when the molecule carries its bond graph (``mol.graph``), the
    symmetry-corrected RMSD prune.  Writes ``<basename>_gpu_pruned.xyz``."""
    data = embedder.mols[filename]
    coords = np.asarray(data.coords, dtype=np.float64)
    embedder.log(f"--> GPU similarity pruning on {filename} ({len(coords)} structures)")
    t0 = perf_counter()
    before = len(coords)
    debug = getattr(embedder, "debuglog", None)
    max_rmsd = embedder.options.rmsd if getattr(embedder.options, "rmsd", None) else 0.25
    if moi or rmsd:
        _, keep, counts = prune_similarity(coords, data.atoms, moi=moi, rmsd=rmsd, max_rmsd=max_rmsd)
        coords = coords[keep]
        if debug is not None:
            debug(f"DEBUG: gpu_prune - MOI {int(counts[0])} -> {int(counts[1])}, RMSD -> {int(counts[2])} structures")
    graph = getattr(data, "graph", None)
    if rmsd and rmsd_rot_corr and graph is not None and len(coords) < 1e3:
        coords, _ = prune_by_rmsd_rot_corr(coords, data.atoms, graph, max_rmsd=max_rmsd, debugfunction=debug)
    embedder.log(f"  Discarded {before - len(coords)} RMSD-similar structures ({len(coords)} left, "
                 f"{perf_counter() - t0:.3f} s)\n")
    outname = data.basename + "_gpu_pruned.xyz"
    write_xyz(data.atoms, coords, outname, title="GPU-pruned conformer")
    return outname


def operate(filename, operator, embedder):
    """Dispatch mirror of ``operate`` (operators.py:63-135) for the operators this
    package provides; anything else is not ours to run."""
    if getattr(embedder.options, "dryrun", False):
        embedder.log(f'--> Dry run requested: skipping operator "{operator}"')
        return filename
    if operator == "gpu_prune":
        return gpu_prune_operator(filename, embedder)
    raise KeyError(f'operator "{operator}" is not provided by firecode_amd')
