#!/usr/bin/env python
"""Secondary workloads of BASELINE.json (configs[2..4]) on one MI355X.
bench.py stays the driver's contract (configs[1]); this script measures the
other named shapes and prints one JSON line per workload.

  python tools/bench_workloads.py embed      # cfg5: 500x500 conformer pairs x 512 rigid transforms
  python tools/bench_workloads.py csearch    # cfg3: 8 torsions x 6-fold = 1 679 616 angle-sets
  python tools/bench_workloads.py prune80    # cfg4 shape on ONE GPU at reduced N (A = 80)
  python tools/bench_workloads.py queue      # many resident ensembles: one call each vs all in flight
"""

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import firecode_amd as fc  # noqa: E402
from firecode_amd import synthetic as syn  # noqa: E402


def embed():
    rng = np.random.default_rng(5)
    n, A = 500, 40
    def mol(seed):
        X, _, _ = syn.synthetic_ensemble(n, A, seed=seed, cluster_size=1, sigma_cluster=0.25)
        X = X - X.reshape(-1, 3).mean(axis=0)  # hypermolecule_class.py:152-156
        r = np.array([3, 7])
        pv = np.stack([X[:, 3] * 1.5, X[:, 7] * 1.5], axis=1)
        return X, r, pv
    m1, r1, pv1 = mol(51)
    m2, r2, pv2 = mol(52)
    steps, rr = 15, 45.0
    angles = np.arange(steps + 1) * 2 * rr / steps - rr
    t0 = time.perf_counter()
    ok, ms = fc.embeds.embed_grid_clash(m1, r1, pv1, m2, r2, pv2, angles, thresh=1.5, max_clashes=0)
    wall = time.perf_counter() - t0
    P = ok.size
    bytes_per_pose = (A + A) * 24 + 2 * 96 + 1
    print(json.dumps({
        "workload": "cfg5 bimolecular rigid embed: 500x500 conformer pairs x 512 rototranslations, clash 1.5 A",
        "poses": P, "kernel_ms": ms, "poses_per_s_kernel": P / (ms * 1e-3),
        "wall_s_host_in_mask_out": wall, "poses_per_s_wall": P / wall, "passed": int(ok.sum()),
        "algorithmic_bytes_per_pose": bytes_per_pose,
        "roofline_hbm_frac": P * bytes_per_pose / (ms * 1e-3) / 8e12,
    }))


def csearch():
    rng = np.random.default_rng(3)
    A, T = 50, 8
    base = syn.synthetic_skeleton(A, rng)
    centres = np.linspace(3, A - 6, T).astype(int)
    torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
    masks = np.zeros((T, A), dtype=bool)
    for t, c in enumerate(centres):
        masks[t, c + 2:] = True
    S = 6 ** T
    atoms = np.array(["C"] * A)
    runs = []
    if os.environ.get("FC_CSEARCH_WARMUP", "1") != "0":
        fc._lib.warmup()  # fc_warmup: device code of every translation unit loaded, buffer pool primed (FC_CSEARCH_WARMUP=0: cold)
    for run in range(int(os.environ.get("FC_CSEARCH_RUNS", "2"))):  # the first run pays the one-time costs (device allocations, first touch of the host buffers)
        t0 = time.perf_counter()
        values = [(0, 60, 120, 180, 240, 300)] * T
        if os.environ.get("FC_CSEARCH_HOST_GRID") == "1":  # A/B: the grid built on the host and sent (round 2)
            angles = fc.utils.cartesian_product(*values)  # the grid is part of the search (:822)
            t_grid = time.perf_counter() - t0
            rot, keep = fc.torsion_module.torsion_scan_tfd(base, torsions, masks, angles, torsions, thresh=1.5, tfd_thresh=10)
        else:
            # the grid of angle-sets is generated on the device (fc_torsion_scan_tfd_grid, what clustered_csearch calls);
            # scan with the fingerprints taken inside the kernel + TFD prune of [base] + [rotated conformers]; the
            # fingerprints stay on the device between the two
            t_grid = 0.0
            rot, keep = fc.torsion_module.torsion_scan_tfd_grid(base, torsions, masks, values, torsions, thresh=1.5, tfd_thresh=10)
        t_scan_tfd = time.perf_counter() - t0
        t1 = time.perf_counter()
        rows = np.flatnonzero(keep[1:])  # (as clustered_csearch_core does it: the starting structure in front, the survivors written behind it)
        first = 1 if keep[0] else 0
        surv = np.empty((first + len(rows),) + base.shape)
        if first:
            surv[0] = base
        fc.torsion_module.torsion_scan(base, torsions, masks, fc.utils.cartesian_rows_at(values, rows), thresh=1.5, out=surv[first:])
        t_regen = time.perf_counter() - t1
        t3 = time.perf_counter()
        _, rmask = fc.pruner.prune_by_rmsd(surv, atoms, 0.5)
        t_rmsd = time.perf_counter() - t3
        runs.append({"s_angle_grid": t_grid, "s_scan_fingerprints_tfd_prune": t_scan_tfd - t_grid, "s_rescan_survivors": t_regen, "s_rmsd_prune": t_rmsd,
                     "s_total": time.perf_counter() - t0})
    wall = runs[1]["s_total"]
    print(json.dumps({
        "workload": "cfg3 csearch: 8 rotatable bonds x 6-fold = 1 679 616 angle-sets (grid generated on the device), clash 1.5 A, back-off 5 deg, "
                    "fingerprints taken inside the scan kernel and TFD-pruned (10 deg) without leaving the device, "
                    "survivors re-scanned, RMSD prune (0.5 A)",
        "angle_sets": S, "kept_after_scan": 1 + int(np.count_nonzero(rot)), "after_tfd": int(keep.sum()), "after_rmsd": int(rmask.sum()),
        "first_run": runs[0], "second_run": runs[1], "all_runs": runs if len(runs) > 2 else None,
        "s_total": wall, "conformers_per_s_total": S / wall,
        "algorithmic_bytes_per_conformer": 2 * A * 24 + T * 4 + 1,
    }))


def prune80():
    n, A = 30000, 80
    X, atoms, asg = syn.synthetic_ensemble(n, A, seed=4)
    ens = fc.DeviceEnsemble(X, center=True)
    ens.bench_prune(0.5, 1.0, reps=1, want_mask=False)
    tk, ts, mask, stats = ens.bench_prune(0.5, 1.0, reps=3, want_mask=True)
    pairs = n * (n - 1) // 2
    print(json.dumps({
        "workload": f"cfg4 shape on one GPU: {n} conformers x {A} atoms all-pairs RMSD prune",
        "pairs": pairs, "kernel_ms": tk, "step_ms": ts, "alignments_per_s": pairs / (ts * 1e-3),
        "survivors": int(mask.sum()), "expected": int(len(np.unique(asg))),
        "screen": {16: "f16x2 MFMA (split-half)", 32: "fp32 MFMA"}.get(fc._lib.screen_last_kind(), "fp64 MFMA"),
        # flops issued to the matrix pipe per pair: split-half = three f16 products over atoms padded to 32, fp32 / fp64 = one
        # product over the atoms; peaks 2.5 PFLOP/s f16, 157.3 TFLOP/s fp32, 78.6 TFLOP/s fp64
        "roofline_mfma_frac": (pairs * 3 * 2 * 9 * 96 / (tk * 1e-3) / 2.5e15 if fc._lib.screen_last_kind() == 16 else
                               pairs * 2 * 9 * 80 / (tk * 1e-3) / (157.3e12 if fc._lib.screen_last_kind() == 32 else 78.6e12)),
    }))


def pcie():
    """host arrays in -> mask out (what a FIRECODE caller of prune_by_rmsd sees)"""
    X, atoms, asg = syn.synthetic_ensemble(10000, 50, seed=2)
    fc.pruner.prune_by_rmsd(X[:2000], atoms, 0.5)  # warm up (context, allocations)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        _, mask = fc.pruner.prune_by_rmsd(X, atoms, 0.5)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    with fc.DeviceEnsemble(X, center=True) as ens:
        ens.prune(0.5, 1.0)
        t0 = time.perf_counter()
        for _ in range(5):
            ens.prune(0.5, 1.0)
        t_res = (time.perf_counter() - t0) / 5
    pairs = 10000 * 9999 // 2
    print(json.dumps({
        "workload": "cfg2 through the drop-in function: prune_by_rmsd(host (10000,50,3) float64) -> (structures[mask], mask)",
        "s_host_in_mask_out": t, "alignments_per_s_pcie_inclusive": pairs / t, "ensembles_per_s_pcie_inclusive": 1 / t,
        "s_resident_prune_call": t_res, "survivors": int(mask.sum()),
        "note": "includes H2D of 12 MB, the prep kernel, workspace allocation (bits, queues) and the output fancy-index copy",
    }))


def values():
    """all-pairs RMSD VALUES (not just decisions) for cfg2"""
    X, atoms, asg = syn.synthetic_ensemble(10000, 50, seed=2)
    with fc.DeviceEnsemble(X, center=True) as ens:
        ens.rmsd_values(want_matrix=False)
        ms = min(ens.rmsd_values(want_matrix=False)[1] for _ in range(5))
    pairs = 10000 * 9999 // 2
    print(json.dumps({
        "workload": "cfg2 all-pairs RMSD values: MFMA covariance + Newton eigenvalue + exact fix-up below 0.02 A, "
                    "(N, N) float64 matrix left in HBM",
        "pairs": pairs, "kernel_ms": ms, "rmsd_values_per_s": pairs / (ms * 1e-3),
    }))


def tri():
    """trimolecular cyclical embed: 12 x 12 x 12 conformer triples x 8 orientations x 216 poses"""
    mols = syn.synthetic_trimolecular(n_conf=(12, 12, 12), n_atoms=(30, 34, 28), seed=9, pivots_per_conf=(1, 1, 1),
                                      sep=(4, 5, 4))
    angles = fc.utils.cartesian_product(*[range(6)] * 3) * 2 * 45 / 5 - 45
    fc.embeds.cyclical_embed_trimolecular(mols, angles, clash_thresh=1.2)  # warm-up
    t0 = time.perf_counter()
    poses, ci, det = fc.embeds.cyclical_embed_trimolecular(mols, angles, clash_thresh=1.2, return_details=True)
    dt = time.perf_counter() - t0
    n_poses = int(det["run"].sum()) * len(angles)
    print(json.dumps({"workload": "trimolecular cyclical embed, 12x12x12 conformers (30+34+28 atoms), 8 orientations, "
                                  "216 step-angle triples; host set-up + adjust + clash + accept filter + pose build",
                      "jobs": len(det["jobs"]), "poses": n_poses, "passed": int(det["passed"].sum()),
                      "accepted": int(len(poses)), "seconds": dt, "poses_per_s": n_poses / dt}))


def small():
    """latency of one-structure calls through the drop-in names (what FIRECODE's Python loops issue)"""
    rng = np.random.default_rng(3)
    c = rng.normal(scale=2.0, size=(36, 3))
    p, q = rng.normal(size=(30, 3)), rng.normal(size=(30, 3))
    X = rng.normal(scale=2.0, size=(200, 30, 3))
    atoms = np.array(["C"] * 30)

    def lat(fn, n=200):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e6

    print(json.dumps({
        "workload": "per-call latency (us), host arrays in -> result out, pool FC_POOL_MB=%s" % os.environ.get("FC_POOL_MB", "default"),
        "compenetration_check_36_atoms": lat(lambda: fc.utils.compenetration_check(c, ids=[20, 16], thresh=1.5)),
        "rmsd_and_max_30_atoms": lat(lambda: fc.rmsd.rmsd_and_max(p, q)),
        "count_clashes_36_atoms": lat(lambda: fc.algebra.count_clashes(c)),
        "prune_by_rmsd_200x30": lat(lambda: fc.pruner.prune_by_rmsd(X, atoms, 0.5), n=50),
    }))


def queue():
    """a queue of resident ensembles pruned one call each vs. all in flight (fc_prune_rmsd_many)"""
    from firecode_amd import _lib as L

    out = {"workload": "queue of resident ensembles, prune at 0.5 A: one fc_prune_rmsd per ensemble vs fc_prune_rmsd_many"}
    for label, count, n, a in [("64x(1000x30)", 64, 1000, 30), ("32x(3000x40)", 32, 3000, 40), ("16x(10000x50)", 16, 10000, 50)]:
        ens = []
        for q in range(count):
            X, _, _ = syn.synthetic_ensemble(n, a, seed=100 + q)
            ens.append(fc.DeviceEnsemble(X, center=True))
        ref = [e.prune(0.5, 1.0)[0] for e in ens]  # also warms every workspace
        best = {"loop": 1e9, "many": 1e9}
        for _ in range(5):
            t0 = time.perf_counter()
            for e in ens:
                e.prune(0.5, 1.0)
            best["loop"] = min(best["loop"], time.perf_counter() - t0)
            t0 = time.perf_counter()
            masks, _ = L.prune_many(ens, 0.5, 1.0)
            best["many"] = min(best["many"], time.perf_counter() - t0)
        assert all(np.array_equal(m, r) for m, r in zip(masks, ref))
        out[label] = {"loop_ms_per_ensemble": best["loop"] / count * 1e3, "many_ms_per_ensemble": best["many"] / count * 1e3}
        for e in ens:
            e.close()
    print(json.dumps(out))


def cfg4():
    """BASELINE configs[3] whole (100 000 x 80): each of the 8 ranks of the sharded prune run on ONE GPU"""
    X, atoms, asg = syn.synthetic_ensemble(100_000, 80, seed=6)
    with fc.DeviceEnsemble(X, center=True) as ens:
        per_rank = {}
        for rank in range(8):
            ens.prune_begin(0.5, 1.0, rank, 8)
            t0 = time.perf_counter()
            st = ens.prune_begin(0.5, 1.0, rank, 8)
            per_rank[rank] = {"screen_ms": st[4] * 1e-6, "begin_call_ms": (time.perf_counter() - t0) * 1e3,
                              "owned_pairs": int(st[0]), "similar": int(st[2])}
    worst = max(v["begin_call_ms"] for v in per_rank.values())
    pairs = 100_000 * 99_999 // 2
    print(json.dumps({"workload": "cfg4 whole: 100 000 conformers x 80 atoms, the similarity stage of each of 8 ranks "
                                  "(row blocks in snake order) timed on one GPU; the exchange + ladder add ~0.2 ms",
                      "pairs": pairs, "per_rank": per_rank, "slowest_rank_ms": worst,
                      "projected_alignments_per_s_8_gpus": pairs / ((worst + 0.2) * 1e-3)}))


if __name__ == "__main__":
    fc.init(0)
    for w in sys.argv[1:] or ["embed", "csearch", "prune80"]:
        {"embed": embed, "csearch": csearch, "prune80": prune80, "pcie": pcie, "values": values, "tri": tri, "small": small, "cfg4": cfg4, "queue": queue}[w]()
