#!/usr/bin/env python
"""bench.py -- conformer-pair RMSD alignments/s (+ pruned ensembles/s) on MI355X.

Workloads (BASELINE.json `configs`), chosen with --workload:

  cfg2  configs[1]: 10 000-conformer x 50-atom float64 ensemble, all-pairs Kabsch RMSD + 0.5 A
        greedy prune.  THE single-GPU workload (default when --gpus 1).  A step = one full pruning
        pass over the HBM-resident ensemble: all-pairs similarity decisions (screen + exact fp64
        refine) and the k-ladder replay -> survivor mask.  On N GPUs the conformer count grows as
        sqrt(N) (weak scaling: pairs per GPU constant).
  cfg4  configs[3]: 100 000-conformer x 80-atom ensemble sharded over the GPUs, one RCCL
        all-gather (default when --gpus > 1).  Weak-scaling family through the named point:
        n_conf = 100 000 * sqrt(N / 8), so every GPU owns 6.25e8 pairs at any N and N = 8 IS
        configs[3]; the line also carries the family's N = 1 member measured on rank 0 in the
        same run, which is what a scaling efficiency has to be computed against.
  cfg5  configs[4]: bimolecular rigid embed, 500 x (500 N / 8) conformer pairs x 512
        rototranslations, compenetration check; poses sharded by molecule-2 conformer, one
        all-gather of the packed pass mask.

What `value` counts (cfg2 / cfg4): PAIR DECISIONS per second -- every conformer pair of the step
is decided exactly as the reference's fp64 Kabsch would decide it (bit-identical mask), but most
pairs are ruled out by a conservative screen and only the candidates get a full fp64 alignment.
The stricter readings are in the same line: `fp64_path` (the same step with the fp64 screen: the
reference's arithmetic in every kernel), `alignments_complete_per_s` (rmsd AND max deviation of
every pair from the explicit rotated difference, the a4 contract) and `rmsd_values_per_s`.

N > 1: one process per GPU (`python -m torch.distributed.run ... bench.py --gpus N`; the ranks read
RANK / LOCAL_RANK / WORLD_SIZE and never import torch), RCCL behind the library's C ABI
(firecode_amd.dist.comm_init_from_env).  Every rank keeps the whole ensemble resident and owns the
row blocks of the similarity matrix dealt in snake order; ONE all-gather of the ranks'
similar-pair lists per prune, the k-ladder replayed on every rank.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAX_RMSD = 0.5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PEAK_F32_MFMA, PEAK_F64_MFMA = 157.3, 78.6  # TFLOP/s dense (same guide)
PEAK_F16_MFMA = 2500.0  # TFLOP/s dense, f16 / bf16 (same guide: ~2.5 PF)
FLOPS_PER_ALIGNMENT = lambda a: 53 * a + 600  # noqa: E731  SURVEY 8d "algorithmic flops" of one complete alignment


# ----------------------------------------------------------------------------------------------
# CPU baselines (oracle = the build's NumPy restatement, "port"); rank 0, N = 1 only
# ----------------------------------------------------------------------------------------------
def cpu_baseline(coords, budget_s=float(os.environ.get("FC_BENCH_CPU_SECONDS", "20"))):
    """Oracle ('port' of the reference's per-pair NumPy path) on a bounded
    sample: all pairs of the first n0 conformers, one core."""
    from oracle import cpu_ref as o

    n0 = 1400
    X = coords[:n0] - coords[:n0].mean(axis=1, keepdims=True)
    iu, ju = np.triu_indices(n0, 1)
    t0 = time.perf_counter()
    done = 0
    for a, b in zip(iu, ju):
        o.rmsd_and_max(X[a], X[b])
        done += 1
        if (done & 1023) == 0 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"{done} conformer pairs among the first {n0} conformers of the workload, "
                      f"oracle rmsd_and_max (NumPy, LAPACK 3x3 SVD per pair), {dt:.1f} s; single process, "
                      f"{len(os.sched_getaffinity(0))} host cores visible (3x3 LAPACK calls do not thread)"}


def cpu_baseline_other_configs(budget_s=8.0):
    """Oracle rates for the secondary configurations of BASELINE.json on bounded samples
    ("CPU baseline beside it"): cfg3 angle-sets/s, cfg5 poses/s.  CPU only."""
    from firecode_amd import synthetic as syn
    from oracle import cpu_ref as o

    out = {}
    rng = np.random.default_rng(3)
    A, T = 50, 8
    base = syn.synthetic_skeleton(A, rng)
    centres = np.linspace(3, A - 6, T).astype(int)
    torsions = np.array([(c - 1, c, c + 1, c + 2) for c in centres])
    masks = np.zeros((T, A), dtype=bool)
    for t, c in enumerate(centres):
        masks[t, c + 2:] = True
    angles = o.cartesian_product(*[(0, 60, 120, 180, 240, 300)] * T)
    pick = np.random.default_rng(0).choice(len(angles), size=4000, replace=False)
    t0, done = time.perf_counter(), 0
    for k in range(0, len(pick), 20):
        o.torsion_scan(base, torsions, masks, angles[pick[k: k + 20]])
        done += 20
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out["cfg3_torsion_scan"] = {"value": done / dt, "unit": "angle-sets/s", "cores": 1, "kind": "port",
                                "sample": f"{done} random angle-sets of the 1 679 616, oracle torsion_scan, {dt:.1f} s"}
    out["cfg5_pose_clash"] = cpu_baseline_poses(budget_s)
    return out


def cpu_baseline_poses(budget_s=8.0):
    from oracle import cpu_ref as o

    rng = np.random.default_rng(5)
    m1 = rng.normal(scale=2.5, size=(4, 40, 3))
    m2 = rng.normal(scale=2.5, size=(4, 40, 3))
    r1, r2 = np.array([0, 7]), np.array([3, 11])
    p1 = np.stack([m1[:, 0] + 0.9, m1[:, 7] - 0.8], axis=1)
    p2 = np.stack([m2[:, 3] + 0.7, m2[:, 11] - 1.0], axis=1)
    ang = np.arange(16) * 2 * 45 / 15 - 45
    t0, done = time.perf_counter(), 0
    for c1 in range(4):
        for c2 in range(4):
            for orient in (0, 1):
                for a1 in ang:
                    for a2 in ang[::4]:
                        R1, t1, R2, t2 = o.bimol_pose_transforms(m1[c1], m2[c2], r1, r2, p1[c1], p2[c2], (a1, a2), orient)
                        pose = o.get_embed([m1[c1], m2[c2]], [R1, R2], [t1, t2])
                        o.compenetration_check(pose, ids=[40, 40], thresh=1.5)
                        done += 1
                if time.perf_counter() - t0 > budget_s:
                    break
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "poses/s", "cores": 1, "kind": "port",
            "sample": f"{done} poses (40+40 atoms): oracle transforms + get_embed + compenetration_check, {dt:.1f} s"}


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def mask_checks(mask, assign):
    """The synthetic ensemble's answer is known: one survivor per cluster, and it is the cluster's
    LAST member (a structure falls to any later similar one)."""
    K = len(np.unique(assign))
    last = np.zeros(K, dtype=np.int64)
    last[assign] = np.arange(len(assign))
    return {"survivors": int(mask.sum()), "survivors_expected": K, "survivor_count_ok": int(mask.sum()) == K,
            "survivors_are_last_cluster_members": bool(np.array_equal(np.flatnonzero(mask), np.sort(last)))}


def screen_roofline(_lib, kernel_ms, owned_pairs, n_atoms, traffic_file=True, world=1, stats=None):
    """Dominant kernel of the prune = the all-pairs screen, on the matrix pipe.
    achieved = MFMA flops of the all-pairs covariance (9 entries x K = atoms padded to 4, per pair) /
    HIP-event time of the screen kernels.  The lean fp32 screen runs in two stages (subset of the
    atoms for every 16 x 32 unit, the remaining work only for the units the subset cannot rule out):
    `achieved` keeps counting the full covariance per pair -- the work a one-stage screen does --
    and `executed_frac` gives what was actually issued to the matrix pipe."""
    kind = _lib.screen_last_kind()
    if kind == 16:
        # split-half screen: three f16 products (hi hi^T, hi lo^T, lo hi^T) over atoms padded to 32 give the
        # covariance at single-precision accuracy; the work issued to the f16 matrix pipe is what `achieved` counts
        a32 = (n_atoms + 31) // 32 * 32
        flops = 3 * 2 * 9 * a32
        tflops = owned_pairs * flops / (kernel_ms * 1e-3) / 1e12
        eq = owned_pairs * 2 * 9 * ((n_atoms + 3) // 4 * 4) / (kernel_ms * 1e-3) / 1e12
        traffic, src = None, None
        pmc = os.path.join(ROOT, "profiles", "r02_pmc_screen_h2.json")
        if traffic_file and world == 1 and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
            src = "from_file: " + os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc passes of an earlier run of this kernel, not of this run)"
        return {"bound": "mfma", "kernel": "k_simbits_screen_mfma_h2", "achieved": tflops, "peak": PEAK_F16_MFMA,
                "unit": "TFLOP/s", "frac": tflops / PEAK_F16_MFMA, "traffic": traffic, "traffic_source": src,
                "kernel_ms": kernel_ms, "flops_per_pair": flops, "dtype": "f16x2",
                "fp32_equivalent": {"what": "the same covariance counted as ONE fp32 product per pair (atoms padded to 4), "
                                            "against the fp32 matrix-pipe peak the round-1 kernel was priced on",
                                    "achieved": eq, "peak": PEAK_F32_MFMA, "frac": eq / PEAK_F32_MFMA},
                "valu_share": {"what": "the bounded fp32 polynomial (~75 vector instructions per pair) is what is left beside "
                                       "the matrix work: lane-instructions per second against the fp32 vector FMA rate",
                               "achieved_tera_lane_instr_per_s": owned_pairs * 75 / (kernel_ms * 1e-3) / 1e12,
                               "peak": PEAK_F32_MFMA / 2, "frac": owned_pairs * 75 / (kernel_ms * 1e-3) / 1e12 / (PEAK_F32_MFMA / 2)},
                "two_stage": None,
                "note": "frac = f16 flops issued to the matrix pipe (3 products x 2 x 9 x atoms padded to 32 per pair) / time / "
                        "2.5 PFLOP/s; the kernel is bound by the sum of that and of the vector epilogue (valu_share)"}
    f32 = kind == 32
    peak = PEAK_F32_MFMA if f32 else PEAK_F64_MFMA
    a4 = (n_atoms + 3) // 4 * 4
    flops = 2 * 9 * a4
    tflops = owned_pairs * flops / (kernel_ms * 1e-3) / 1e12
    staged = None
    if f32 and stats is not None and len(stats) > 7 and stats[6] > 0 and stats[7] == 0:
        ks = a4 // 4
        ks1 = (ks + 1) // 2
        executed = owned_pairs * 2 * 9 * 4 * ks1 + int(stats[6]) * 512 * 2 * 9 * 4 * ks
        staged = {"units_queued_for_the_full_test": int(stats[6]), "units_total_about": int(owned_pairs // 512),
                  "executed_mfma_tflops": executed / (kernel_ms * 1e-3) / 1e12,
                  "executed_frac": executed / (kernel_ms * 1e-3) / 1e12 / peak,
                  "kernels": "k_simbits_screen_mfma_f32<4, false, true> (subset stage, sample + rest) + "
                             "k_screen_density_verdict + k_screen_units_f32 (full test per queued unit)"}
    traffic, src = None, None
    pmc = os.path.join(ROOT, "profiles", "r02_pmc_screen_f32.json" if f32 else "r02_pmc_screen_f64.json")
    if traffic_file and world == 1 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        src = "from_file: " + os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc passes of an earlier run of this kernel, not of this run)"
    return {"bound": "mfma", "kernel": "k_simbits_screen_mfma_f32" if f32 else "k_simbits_screen_mfma",
            "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak, "traffic": traffic,
            "traffic_source": src, "kernel_ms": kernel_ms, "flops_per_pair": flops, "dtype": "f32" if f32 else "f64",
            "two_stage": staged,
            "note": ("frac = algorithmic covariance flops (the full K loop per pair, what a one-stage screen executes) / time / peak; "
                     "the two-stage screen issues only two_stage.executed_frac of the peak to the matrix pipe -- the rest of the "
                     "gap to 1 is work it avoids, not pipe utilisation") if staged else None}


def timed_prunes(ens, steps, warmup, sharded, overlap=True):
    """W untimed + K timed stream-ordered prunes (one host wait per batch) -> (elapsed s, mean screen
    kernel ms, mask, stats).  The caller brackets this with its barrier."""
    def run(n):
        if sharded:
            return ens.bench_prune_sharded(MAX_RMSD, 2 * MAX_RMSD, reps=n, overlap=overlap)
        return ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=n, want_mask=True)

    run(2)  # set-up outside the timed region, whatever W is: second workspace, streams, operand copy
    # ... and the chip at its steady clocks: a dozen 0.5 ms steps end before the device has ramped up
    # (measured: 6 % longer steps at K = 20 than at K = 200); ~0.15 s of the same prunes, untimed
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < float(os.environ.get("FC_BENCH_SETTLE_S", "0.15")):
        run(64 if not sharded else 8)
    done = 0
    while done < warmup:
        n = min(512, warmup - done)
        run(n)
        done += n
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.25 s of GPU time; a dozen steps end before the device reaches its steady clocks
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("auto", "cfg2", "cfg4", "cfg5"), default="auto",
                    help="auto = cfg2 on one GPU (BASELINE configs[1]), cfg4 on several (configs[3] at 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region (profile runs): no fp64_path / complete / secondary / host-in legs")
    ap.add_argument("--cpu-baseline-configs", action="store_true",
                    help="only time the CPU oracle on samples of BASELINE configs 3 and 5 (no GPU needed) and exit")
    args = ap.parse_args()
    if args.cpu_baseline_configs:
        print(json.dumps({"cpu_baseline_other_configs": cpu_baseline_other_configs(),
                          "host_cores_visible": len(os.sched_getaffinity(0))}))
        return

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL, before the HSA runtime starts
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` "
                     "(any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE)")
        args.gpus = world
    workload = args.workload if args.workload != "auto" else ("cfg2" if world == 1 else "cfg4")

    # RCCL writes a version banner to fd 1 when a communicator is created; the contract is ONE JSON
    # line on stdout, so everything but that line goes to stderr
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import firecode_amd as fc
    from firecode_amd import _lib
    from firecode_amd import dist as fdist
    from firecode_amd import synthetic as syn

    # FC_BENCH_FORCE_SHARDED=1: the multi-GPU code path (RCCL communicator of one rank) on a single GPU
    sharded = world > 1 or workload == "cfg4" or os.environ.get("FC_BENCH_FORCE_SHARDED") == "1"
    if sharded or workload == "cfg5":
        rank, world, local_rank = fdist.comm_init_from_env()
    else:
        rank, local_rank = 0, 0
        fc.init(0)

    def barrier():
        if sharded or workload == "cfg5":
            _lib.comm_barrier()  # a 1-byte all-gather + device synchronisation on every rank

    def max_over_ranks(x):
        if not (sharded or workload == "cfg5") or world == 1:
            return float(x)
        g = _lib.allgather_mask(np.array([x], dtype=np.float64).view(np.uint8))
        return float(g.view(np.float64).max())

    if workload == "cfg5":
        out = run_cfg5(args, fc, _lib, fdist, syn, rank, world, barrier, max_over_ranks)
    else:
        out = run_prune(args, workload, fc, _lib, fdist, syn, rank, world, sharded, barrier, max_over_ranks)

    if rank == 0:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if sharded or workload == "cfg5":
        _lib.comm_barrier()
        _lib.comm_destroy()


# ----------------------------------------------------------------------------------------------
# cfg2 / cfg4: all-pairs RMSD prune
# ----------------------------------------------------------------------------------------------
def run_prune(args, workload, fc, _lib, fdist, syn, rank, world, sharded, barrier, max_over_ranks):
    if workload == "cfg2":
        n_atoms, seed = 50, 2
        n_conf = int(round(10000 * np.sqrt(world)))
        steps = 200 if args.steps is None else args.steps
        warmup = 20 if args.warmup is None else args.warmup
        what = (f"{n_conf}-conformer x {n_atoms}-atom ensemble, all-pairs Kabsch RMSD + {MAX_RMSD} A prune "
                "(BASELINE configs[1]" + ("" if world == 1 else "; conformers scaled by sqrt(n_gpus): pairs per GPU constant") + ")")
    else:
        n_atoms, seed = 80, 6
        n_conf = int(round(100000 * np.sqrt(world / 8.0)))
        steps = 20 if args.steps is None else args.steps
        warmup = 3 if args.warmup is None else args.warmup
        what = (f"{n_conf}-conformer x {n_atoms}-atom ensemble sharded over {world} GPU(s), all-pairs Kabsch RMSD + "
                f"{MAX_RMSD} A prune, one RCCL all-gather per prune (BASELINE configs[3] is the n_gpus = 8 member of "
                "this weak-scaling family: n_conf = 100 000 * sqrt(n_gpus / 8), 6.25e8 pairs per GPU)")
    coords, atoms, assign = syn.synthetic_ensemble(n_conf, n_atoms, seed=seed)
    ens = fc.DeviceEnsemble(coords, center=True)  # resident in HBM from here on
    pairs_total = n_conf * (n_conf - 1) // 2

    run = timed_prunes(ens, steps, warmup, sharded)
    barrier()
    t0 = time.perf_counter()
    tk, done, owned = 0.0, 0, pairs_total
    while done < steps:  # one host wait per batch of stream-ordered steps
        n = min(512, steps - done)
        k_ms, _, mask, stats = run(n)
        tk += k_ms * n
        done += n
        owned = int(stats[0])
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    t_kernel_ms = tk / steps
    if rank != 0:
        if workload == "cfg4" and world > 1:
            barrier()  # rank 0 measures the family's single-GPU member meanwhile
        return None

    out = {
        "metric": "conformer-pair RMSD alignments/s",
        "value": pairs_total * steps / elapsed,
        "unit": "alignments/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": what, "baseline_config": "configs[1]" if workload == "cfg2" else "configs[3]",
                   "n_conformers": n_conf, "n_atoms": n_atoms, "max_rmsd": MAX_RMSD, "pairs_per_step": pairs_total,
                   "value_counts": "pair decisions: every pair of the step decided as the reference's fp64 Kabsch "
                                   "decides it (bit-identical mask); a conservative all-pairs screen rules most pairs "
                                   "out, candidates get the full fp64 alignment (rmsd + max deviation).  Stricter "
                                   "readings: fp64_path, alignments_complete_per_s, rmsd_values_per_s",
                   "sharding": (f"row blocks of 128 dealt in snake order over {world} rank(s); one all-gather of "
                                "similar-pair lists, ladder replayed on every rank") if sharded else "none (single GPU)",
                   "comm": "RCCL through libfc_hip.so's C ABI (fc_comm_init / ncclAllGather), no PyTorch in the ranks"
                           if sharded else "none",
                   "host_sync": "once per batch of <= 512 stream-ordered steps (one pinned result slot each)",
                   "untimed_before_warmup": "set-up (second workspace, streams, fp32 operand copy) and "
                                            + os.environ.get("FC_BENCH_SETTLE_S", "0.15") + " s of the same prunes so that the "
                                            "W + K steps run at the device's steady clocks",
                   "step_overlap": "screens in order on one stream; refine" + (" + export + all-gather" if sharded else "")
                                   + " + ladder + result copy of step r run beside the screen of step r+1 "
                                     "(two workspaces over the same resident coordinates)"},
        "pruned_ensembles_per_s": steps / elapsed,
    }
    out.update(mask_checks(mask, assign))
    out["roofline"] = screen_roofline(_lib, t_kernel_ms, owned, n_atoms, world=world, stats=stats)
    out["roofline"]["kernel_ms_source"] = ("HIP events on the kernel's stream around every %sth launch of the timed region "
                                           "(an event pair costs the stream ~14 us; FC_BENCH_EVENT_STRIDE=1 times all)"
                                           % os.environ.get("FC_BENCH_EVENT_STRIDE", "8")) + ("" if world == 1 else "; rank 0's launches")
    out["dtype"] = {"f32": "f32 screen + f64 exact refine", "f16x2": "f16x2 screen (split-half, fp32-accurate) + f64 exact refine",
                    "f64": "f64"}[out["roofline"]["dtype"]]
    bytes_per_alignment = 2 * n_atoms * 24 + 16
    achieved = owned * bytes_per_alignment / (t_kernel_ms * 1e-3) / 1e9
    # the north star's view: algorithmic bytes (two conformers in, rmsd + maxdev out) against the
    # 8 TB/s HBM roof; > 1 because a staged tile serves 64-256 partners
    out["roofline_hbm"] = {"bound": "hbm", "kernel": out["roofline"]["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": out["roofline"]["traffic"],
                           "algorithmic_bytes_per_alignment": bytes_per_alignment,
                           "compulsory_bytes": n_conf * n_atoms * 24 + n_conf * ((n_conf + 63) // 64) * 8}

    if workload == "cfg4" and world > 1:
        # the family's n_gpus = 1 member (same pairs per GPU), on rank 0 alone, outside the timed region
        n1 = int(round(100000 * np.sqrt(1 / 8.0)))
        c1, _, a1 = syn.synthetic_ensemble(n1, n_atoms, seed=seed)
        with fc.DeviceEnsemble(c1, center=True) as e1:
            e1.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=2, want_mask=False)
            _, s_ms, m1, _ = e1.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=min(steps, 20), want_mask=True)
        p1 = n1 * (n1 - 1) // 2
        out["scaling_family_n1"] = {"n_conformers": n1, "pairs_per_step": p1, "ms_per_step": s_ms,
                                    "value": p1 / (s_ms * 1e-3), "survivor_count_ok": mask_checks(m1, a1)["survivor_count_ok"],
                                    "note": "same kernels without the exchange, measured on rank 0 while the other ranks wait; "
                                            "scaling efficiency of this line = value / (n_gpus * scaling_family_n1.value)"}
        barrier()

    if not args.no_extras and not sharded:
        extras_single_gpu(args, out, fc, _lib, syn, ens, coords, atoms, n_conf, n_atoms, pairs_total, steps, warmup)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(coords)
    return out


def extras_single_gpu(args, out, fc, _lib, syn, ens, coords, atoms, n_conf, n_atoms, pairs_total, steps, warmup):
    """Everything below is OUTSIDE the timed region of `value`; each leg has its own device timing."""
    # (a) the same step with the fp64 screen: the reference's arithmetic in every kernel
    _lib.screen_select(64)
    try:
        run = timed_prunes(ens, steps, warmup, False)
        t0 = time.perf_counter()
        tk, done = 0.0, 0
        while done < steps:
            n = min(512, steps - done)
            k_ms, _, mask64, _ = run(n)
            tk += k_ms * n
            done += n
        el = time.perf_counter() - t0
        r64 = screen_roofline(_lib, tk / steps, pairs_total, n_atoms)
        out["fp64_path"] = {"what": "the same K steps with fc_screen_select(64): fp64 MFMA screen, fp64 refine, ladder",
                            "dtype": "f64", "ms_per_step": 1e3 * el / steps, "value": pairs_total * steps / el,
                            "unit": "alignments/s (pair decisions, all arithmetic fp64)",
                            "pruned_ensembles_per_s": steps / el, "mask_equals_default_path": None, "roofline": r64}
    finally:
        _lib.screen_select(0)
    _, _, mask32, _ = ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=1, want_mask=True)
    out["fp64_path"]["mask_equals_default_path"] = bool(np.array_equal(mask32, mask64))
    # (b) the a4 contract for every pair: (rmsd, maxdev) from the explicit rotated difference
    ens.rmsd_and_max_all(want_matrices=False)
    ms_c = min(ens.rmsd_and_max_all(want_matrices=False)[2] for _ in range(3))
    fl = FLOPS_PER_ALIGNMENT(n_atoms)
    out["alignments_complete_per_s"] = pairs_total / (ms_c * 1e-3)
    out["alignments_complete"] = {
        "what": "rmsd_and_max of ALL pairs (firecode/utils.py:499 contract): two conformers in -> (rmsd, max deviation) out, "
                "fp64: covariance tiles on the fp64 matrix pipe, rotation + explicit rotated difference per pair; "
                "dense (N, N) outputs stay in HBM",
        "kernel": "k_simbits_screen_mfma<., 2> + k_rmsd_fix_small", "kernel_ms": ms_c, "dtype": "f64",
        "roofline": {"bound": "mfma", "achieved": pairs_total * fl / (ms_c * 1e-3) / 1e12, "peak": PEAK_F64_MFMA,
                     "unit": "TFLOP/s", "frac": pairs_total * fl / (ms_c * 1e-3) / 1e12 / PEAK_F64_MFMA,
                     "flops_per_alignment": fl, "note": "SURVEY 8d algorithmic flops (53 A + 600) against the fp64 peak"},
        "roofline_hbm": {"bound": "hbm", "achieved": pairs_total * (2 * n_atoms * 24 + 16) / (ms_c * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": pairs_total * (2 * n_atoms * 24 + 16) / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8d algorithmic 2*A*24+16 B per alignment; > 1 = tile reuse",
                         "output_bytes_written": 2 * 8 * pairs_total}}
    # (c) RMSD values only (no rotation)
    ens.rmsd_values(want_matrix=False)
    values_ms = min(ens.rmsd_values(want_matrix=False)[1] for _ in range(3))
    out["rmsd_values_per_s"] = pairs_total / (values_ms * 1e-3)
    # (d) an ensemble WITHOUT cluster structure: continuous RMSD distribution across the threshold
    Xc = syn.continuous_ensemble(n_conf, n_atoms, seed=11, thr=MAX_RMSD)
    with fc.DeviceEnsemble(Xc, center=True) as ec:
        ec.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=2, want_mask=False)
        kc, sc, mc, stc = ec.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=max(2, min(steps, 20)), want_mask=True)
        kind_c = _lib.screen_last_kind()
    out["config"]["secondary"] = {
        "workload": f"{n_conf} x {n_atoms}, continuous RMSD distribution (6 collective modes, ~1.5 % of the pairs below "
                    f"{MAX_RMSD} A, smooth density across the threshold): the case the clustered ensemble does not exercise",
        "ms_per_step": sc, "value": pairs_total / (sc * 1e-3), "screen_kernel_ms": kc,
        "screen": {16: "f16x2", 32: "f32"}.get(kind_c, "f64"), "candidates_refined": int(stc[1]), "similar_pairs": int(stc[2]),
        "survivors": int(mc.sum())}
    # (e) BASELINE's second metric as SURVEY 8d words it: host arrays in -> mask out, H2D / D2H included
    fc.pruner.prune_by_rmsd(coords[:2000], atoms, MAX_RMSD)
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        fc.pruner.prune_by_rmsd(coords, atoms, MAX_RMSD)
        ts.append(time.perf_counter() - t1)
    out["pruned_ensembles_per_s_host_in_mask_out"] = 1.0 / min(ts)
    out["host_in_mask_out_ms"] = 1e3 * min(ts)
    # ... and the drivers' MOI -> RMSD sequence on one upload (fc_prune_similarity)
    fc.pruner.prune_similarity(coords[:2000], atoms, max_rmsd=MAX_RMSD)
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        _, m_both, counts = fc.pruner.prune_similarity(coords, atoms, max_rmsd=MAX_RMSD)
        ts.append(time.perf_counter() - t1)
    out["similarity_pipeline_host_in_mask_out"] = {
        "what": "prune_by_moment_of_inertia -> prune_by_rmsd as Ensemble.similarity_pruning runs them "
                "(firecode/ensemble.py:205-235), host arrays in -> mask out, ONE upload of the coordinates",
        "ms": 1e3 * min(ts), "structures": [int(c) for c in counts]}


# ----------------------------------------------------------------------------------------------
# cfg5: bimolecular rigid embed pose grid
# ----------------------------------------------------------------------------------------------
def run_cfg5(args, fc, _lib, fdist, syn, rank, world, barrier, max_over_ranks):
    n1, A = 500, 40
    n2 = max(1, int(round(500 * world / 8.0)))
    steps = 5 if args.steps is None else args.steps
    warmup = 1 if args.warmup is None else args.warmup

    def mol(n, seed):
        X, _, _ = syn.synthetic_ensemble(n, A, seed=seed, cluster_size=1, sigma_cluster=0.25)
        X = X - X.reshape(-1, 3).mean(axis=0)  # hypermolecule_class.py:152-156
        return X, np.array([3, 7]), np.stack([X[:, 3] * 1.5, X[:, 7] * 1.5], axis=1)

    m1, r1, pv1 = mol(n1, 51)
    m2, r2, pv2 = mol(500, 52)
    m2, pv2 = m2[:n2], pv2[:n2]
    angles = np.arange(16) * 2 * 45.0 / 15 - 45.0
    gather = fdist.rccl_allgather()
    kernel_ms = []

    def grid_fn(*a, **k):
        ok, ms = fc.embeds.embed_grid_clash(*a, **k)
        kernel_ms.append(ms)
        return ok, ms

    def step():
        return fdist.embed_grid_clash_sharded(m1, r1, pv1, m2, r2, pv2, angles, rank=rank, world=world,
                                              allgather_fn=gather, thresh=1.5, max_clashes=0, grid_fn=grid_fn)

    for _ in range(warmup):
        step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ok = step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return None
    P = int(ok.size)
    k_ms = float(np.mean(kernel_ms))
    local_poses = P // world
    bytes_per_pose = (A + A) * 24 + 2 * 96 + 1
    out = {"metric": "clash-checked embed poses/s", "value": P * steps / elapsed, "unit": "poses/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 screen + f64 exact recount", "data": "synthetic",
           "config": {"workload": f"bimolecular rigid embed: {n1} x {n2} conformer pairs x 512 rototranslations (2 orientations x "
                                  f"16 x 16 step angles), 40 + 40 atoms, compenetration check at 1.5 A (BASELINE configs[4] is "
                                  "the n_gpus = 8 member: 500 x 500); poses sharded by molecule-2 conformer, one all-gather of "
                                  "the packed pass mask; a step = host tables in -> gathered pass mask out",
                      "baseline_config": "configs[4]", "poses_per_step": P, "passed": int(ok.sum()),
                      "comm": "RCCL through libfc_hip.so's C ABI (fc_allgather_mask)"},
           "roofline": {"bound": "hbm", "kernel": "k_embed_grid_clash", "achieved": local_poses * bytes_per_pose / (k_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": local_poses * bytes_per_pose / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "kernel_ms": k_ms, "algorithmic_bytes_per_pose": bytes_per_pose,
                        "note": "rank 0's kernel (HIP events); > 1 = the pre-transformed structures are reused from LDS / L2"}}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_poses()
    return out


if __name__ == "__main__":
    main()
