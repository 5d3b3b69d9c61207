#!/usr/bin/env python
"""Extract, with `ast`, every call the reference makes to a function of the hot path and write
them to tests/golden/callsites_v1.json: file, line, name, number of positional arguments, keyword names.

Build-container tool (like make_golden.py): it READS the reference's sources as text -- nothing is imported or
executed -- and nothing of it travels to the GPU box; tests/test_boundary_signatures.py binds every recorded call
onto the firecode_amd callable of the same name (inspect.signature(...).bind) from the committed JSON alone.

    python tests/golden/make_callsites.py [/root/reference]
"""
import ast
import json
import os
import sys

# the names FIRECODE's callers use for the rows of SURVEY 8a (functions; Ensemble's methods are checked apart)
NAMES = {
    "prune_by_rmsd", "prune_by_moment_of_inertia", "prune_by_rmsd_rot_corr", "rmsd_and_max", "get_alignment_matrix",
    "align_structures", "align_by_moi", "get_inertia_moments", "align_vec_pair", "count_clashes", "compenetration_check",
    "get_embed", "cartesian_product", "rotate_dihedral", "torsion_comp_check", "_get_rotation_mask", "prune_conformers_tfd",
    "get_torsion_fingerprint", "_get_tf_mat", "rmsd_similarity", "clustered_csearch", "random_csearch",
    "most_diverse_conformers", "string_embed", "cyclical_embed", "fitness_check", "similarity_pruning",
}
# calls of these names on objects that are NOT the hot path's (ase Atoms.rotate_dihedral ...)
SKIP = {("atropisomer_module.py", "rotate_dihedral")}


def parse(path):
    """ast of a reference file.  The reference targets Python 3.12 (backslashes inside f-string expressions); on an older
    interpreter such a literal is re-read as a plain string -- a call inside one is lost to the record, nothing else."""
    src = open(path).read()
    try:
        return ast.parse(src, filename=path)
    except SyntaxError:
        import io
        import re
        import tokenize

        toks = []
        for t in tokenize.generate_tokens(io.StringIO(src).readline):
            if t.type == tokenize.STRING and re.match(r"(?i)^[rb]*f[rb]*['\"]", t.string) and "\\" in t.string:
                t = t._replace(string=re.sub(r"(?i)^([rb]*)f([rb]*)", r"\1\2", t.string, count=1))
            toks.append(t)
        return ast.parse(tokenize.untokenize(toks), filename=path)


def calls_in(path, rel):
    tree = parse(path)
    out = []
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call):
            continue
        f = node.func
        name = f.id if isinstance(f, ast.Name) else f.attr if isinstance(f, ast.Attribute) else None
        if name not in NAMES or (os.path.basename(rel), name) in SKIP:
            continue
        method = isinstance(f, ast.Attribute)
        out.append({
            "file": rel, "line": node.lineno, "name": name, "method_call": method,
            "n_positional": sum(not isinstance(a, ast.Starred) for a in node.args),
            "star_args": any(isinstance(a, ast.Starred) for a in node.args),
            "keywords": [k.arg for k in node.keywords if k.arg is not None],
            "star_kwargs": any(k.arg is None for k in node.keywords),
        })
    return out


def definitions_in(path, rel):
    """the reference's own definitions of those names: parameter names in order, defaults, for the record"""
    tree = parse(path)
    out = []
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and node.name in NAMES:
            a = node.args
            out.append({"file": rel, "line": node.lineno, "name": node.name,
                        "params": [x.arg for x in a.posonlyargs + a.args], "n_defaults": len(a.defaults),
                        "kwonly": [x.arg for x in a.kwonlyargs]})
    return out


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    pkg = os.path.join(root, "firecode")
    calls, defs = [], []
    for dirpath, _, files in sorted(os.walk(pkg)):
        if os.sep + "tests" in dirpath:
            continue
        for f in sorted(files):
            if f.endswith(".py"):
                p = os.path.join(dirpath, f)
                rel = os.path.relpath(p, root)
                calls += calls_in(p, rel)
                defs += definitions_in(p, rel)
    calls.sort(key=lambda c: (c["file"], c["line"], c["name"]))
    defs.sort(key=lambda c: (c["file"], c["line"]))
    out = {"source": "ntampellini/FIRECODE (reference tree), extracted with ast by tests/golden/make_callsites.py",
           "calls": calls, "definitions": defs}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "callsites_v1.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{len(calls)} calls, {len(defs)} definitions -> {dst}")


if __name__ == "__main__":
    main()
