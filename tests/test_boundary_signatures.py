"""The drop-in boundary, mechanically: every call the reference makes to a function of the hot path (SURVEY.md 8a;
extracted with `ast` from the reference's sources by tests/golden/make_callsites.py into tests/golden/callsites_v1.json)
must bind -- same positional count, same keyword names -- onto the firecode_amd callable of the same name, and every
parameter of the reference's own definitions of those functions must be accepted under the same name and position.
CPU only: signatures are inspected, nothing is called."""

import importlib
import inspect
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SITES = json.load(open(os.path.join(ROOT, "tests", "golden", "callsites_v1.json")))

# where FIRECODE's callers would import each name from after the swap
HOME = {
    "prune_by_rmsd": "firecode_amd.pruner", "prune_by_moment_of_inertia": "firecode_amd.pruner",
    "prune_by_rmsd_rot_corr": "firecode_amd.pruner", "rmsd_and_max": "firecode_amd.rmsd",
    "get_alignment_matrix": "firecode_amd.rmsd", "align_structures": "firecode_amd.utils",
    "align_by_moi": "firecode_amd.hypermolecule_class", "get_inertia_moments": "firecode_amd.algebra",
    "align_vec_pair": "firecode_amd.algebra", "count_clashes": "firecode_amd.algebra",
    "compenetration_check": "firecode_amd.utils", "get_embed": "firecode_amd.embeds",
    "cartesian_product": "firecode_amd.utils", "rotate_dihedral": "firecode_amd.utils",
    "torsion_comp_check": "firecode_amd.torsion_module", "_get_rotation_mask": "firecode_amd.torsion_module",
    "prune_conformers_tfd": "firecode_amd.torsion_module", "get_torsion_fingerprint": "firecode_amd.torsion_module",
    "_get_tf_mat": "firecode_amd.torsion_module", "rmsd_similarity": "firecode_amd.utils",
    "clustered_csearch": "firecode_amd.torsion_module", "random_csearch": "firecode_amd.torsion_module",
    "most_diverse_conformers": "firecode_amd.torsion_module", "string_embed": "firecode_amd.embeds",
    "cyclical_embed": "firecode_amd.embeds", "fitness_check": "firecode_amd.utils",
}


def _callable(name):
    if name == "similarity_pruning":
        from firecode_amd.ensemble import Ensemble

        return Ensemble.similarity_pruning, True
    mod = importlib.import_module(HOME[name])
    return getattr(mod, name), False


def test_every_extracted_name_has_a_home():
    names = {c["name"] for c in SITES["calls"]} | {d["name"] for d in SITES["definitions"]}
    assert names - {"similarity_pruning"} <= set(HOME), sorted(names - set(HOME))
    assert len(SITES["calls"]) >= 80


@pytest.mark.parametrize("site", SITES["calls"], ids=lambda c: f"{c['file']}:{c['line']}:{c['name']}")
def test_reference_call_binds_onto_the_product_callable(site):
    fn, is_method = _callable(site["name"])
    sig = inspect.signature(fn)
    args = [object()] * (site["n_positional"] + (1 if is_method else 0))
    kwargs = {k: object() for k in site["keywords"]}
    if site["star_args"]:
        # f(*arrays): the callable must take a variable number of positional arguments (cartesian_product)
        assert any(p.kind is p.VAR_POSITIONAL for p in sig.parameters.values()), f"{site['name']} must accept *args"
    sig.bind(*args, **kwargs)  # raises TypeError when the reference's call does not fit


@pytest.mark.parametrize("d", SITES["definitions"], ids=lambda d: f"{d['file']}:{d['line']}:{d['name']}")
def test_reference_definition_parameters_are_accepted(d):
    """positional parameters in the reference's order, keyword use of every one of them"""
    fn, _ = _callable(d["name"])
    sig = inspect.signature(fn)
    params = list(sig.parameters.values())
    if any(p.kind is p.VAR_POSITIONAL for p in params) and not d["params"]:
        return  # cartesian_product(*arrays)
    names = [p.name for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    n_required = len(d["params"]) - d["n_defaults"]
    assert names[: len(d["params"])] == d["params"], f"{d['name']}: {names} vs the reference's {d['params']}"
    sig.bind(*[object()] * n_required)  # the reference's required arguments suffice
    sig.bind(**{k: object() for k in d["params"]})
