#!/usr/bin/env python
"""bench.py -- conformer-pair RMSD alignments/s (+ pruned ensembles/s) on MI355X.

Workload (BASELINE.json configs[1]): synthetic 10 000-conformer
x 50-atom float64 ensemble, all-pairs Kabsch RMSD + 0.5 A greedy prune.
A step = one full pruning pass over the HBM-resident ensemble: all-pairs
similarity bits (screen + exact refine) and the k-ladder replay -> survivor mask.

N GPUs (weak scaling): every rank keeps the whole ensemble resident and owns a
block-cyclic share of the bit-matrix rows; the conformer count grows as
sqrt(N) so that the pairs per GPU stay constant; masks are exchanged with one
RCCL all-gather per ladder level.
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

N_CONF, N_ATOMS, MAX_RMSD = 10000, 50, 0.5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


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
    out["cfg5_pose_clash"] = {"value": done / dt, "unit": "poses/s", "cores": 1, "kind": "port",
                              "sample": f"{done} poses (40+40 atoms): oracle transforms + get_embed + compenetration_check, {dt:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.25 s of GPU time; a dozen steps end before the device reaches its steady clocks
    # (measured: 1.040 ms per step at K = 10, W = 2; 0.980 ms at K = 200, W = 20 and at K = 1000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-configs", action="store_true",
                    help="only time the CPU oracle on samples of BASELINE configs 3 and 5 (no GPU needed) and exit")
    args = ap.parse_args()
    if args.cpu_baseline_configs:
        print(json.dumps({"cpu_baseline_other_configs": cpu_baseline_other_configs(),
                          "host_cores_visible": len(os.sched_getaffinity(0))}))
        return

    # dmabuf IPC for RCCL; must be in the environment before the HSA runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world

    import firecode_amd as fc
    from firecode_amd import _lib
    from firecode_amd import dist as fdist
    from firecode_amd import synthetic as syn

    # FC_BENCH_BACKEND=gloo rehearses the multi-rank flow on ONE GPU (all ranks on
    # device 0, masks exchanged through gloo); the driver's runs use RCCL.
    backend = os.environ.get("FC_BENCH_BACKEND", "nccl")
    dev_index = 0 if backend == "gloo" else local_rank
    fc.init(dev_index)

    allgather = None
    tdist = None
    # FC_BENCH_FORCE_SHARDED=1: take the multi-GPU code path (RCCL group of one rank) on a
    # single GPU -- measures what the exchange costs over the resident single-GPU step
    sharded = world > 1 or os.environ.get("FC_BENCH_FORCE_SHARDED") == "1"
    stdout_fd = None
    if sharded:
        # RCCL writes a version banner to fd 1 when the communicator is created; the contract
        # is ONE JSON line on stdout, so everything but that line goes to stderr
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch
        import torch.distributed as tdist

        if backend == "gloo":
            tdist.init_process_group(backend="gloo")
            allgather = fdist.torch_allgather()
        else:
            torch.cuda.set_device(local_rank)
            # the screens fill every workgroup slot of the chip and run at the lowest stream priority;
            # RCCL's own stream gets the highest, like the lanes that feed it (firecode_amd/dist.py)
            pg_opts = tdist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            if world == 1 and "RANK" not in os.environ:
                tdist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29655", rank=0,
                                         world_size=1, device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
            else:
                tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
            allgather = fdist.torch_allgather(device=torch.device("cuda", local_rank))

    n_conf = int(round(N_CONF * np.sqrt(world)))
    coords, atoms, assign = syn.synthetic_ensemble(n_conf, N_ATOMS, seed=2)
    ens = fc.DeviceEnsemble(coords, center=True)  # resident in HBM from here on
    pairs_total = n_conf * (n_conf - 1) // 2

    def barrier():
        if sharded:
            import torch

            if backend != "gloo":
                torch.cuda.synchronize()
            tdist.barrier()
            if backend != "gloo":
                torch.cuda.synchronize()

    t_kernel_ms = None
    if not sharded:
        # the K steps are enqueued back to back by one library call (fc_bench_prune_rmsd): every step
        # is the whole pass -- counters reset, screen, refine, level buckets, ladder, survivor words
        # + counters copied to the step's own pinned host slot -- and the host waits once for all K
        # (the barrier + synchronisation the contract asks for, not one per step)
        # set-up outside the timed region, whatever W is: second workspace, streams, fp32 operand copy
        ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=2, want_mask=False)
        if args.warmup:
            ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=min(args.warmup, 1024), want_mask=True)
        barrier()
        t0 = time.perf_counter()
        tk, done = 0.0, 0
        while done < args.steps:  # one host wait per batch of at most 1024 stream-ordered steps
            n = min(1024, args.steps - done)
            k_ms, s_ms, _, stats = ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=n, want_mask=True)
            tk += k_ms * n
            done += n
        elapsed = time.perf_counter() - t0
        t_kernel_ms = tk / args.steps
        _, _, mask, stats = ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=1, want_mask=True)
        # outside the timed region: the stricter reading of "alignment" -- an RMSD VALUE per pair
        ens.rmsd_values(want_matrix=False)
        values_ms = min(ens.rmsd_values(want_matrix=False)[1] for _ in range(3))
        # ... and BASELINE's second metric as SURVEY 8d words it: one full prune through the drop-in
        # function, host arrays in -> mask out, H2D / D2H included
        host_in_out_s = []
        if not args.no_cpu_baseline:  # (skipped with the other side measurements in profile runs)
            fc.pruner.prune_by_rmsd(coords[:2000], atoms, MAX_RMSD)
        for _ in range(0 if args.no_cpu_baseline else 5):
            t1 = time.perf_counter()
            fc.pruner.prune_by_rmsd(coords, atoms, MAX_RMSD)
            host_in_out_s.append(time.perf_counter() - t1)
    else:
        if backend == "gloo":  # rehearsal: host exchange through gloo
            def step():
                return fdist.prune_by_rmsd_sharded(ens, MAX_RMSD, rank=rank, world=world, allgather_fn=allgather)
        else:  # everything between the screen and the mask stays in HBM, one RCCL all-gather
            def step():
                return fdist.prune_by_rmsd_sharded_device(ens, MAX_RMSD, rank=rank, world=world,
                                                          device=torch.device("cuda", local_rank))
        for _ in range(args.warmup):
            step()
        if backend != "gloo":  # set-up of the overlapped batch path outside the timed region, whatever W is:
            # second workspace and its operand copy, lane streams, the two message buffer pairs
            fdist.prune_steps_sharded_device(ens, 2, MAX_RMSD, rank=rank, world=world,
                                             device=torch.device("cuda", local_rank))
        barrier()
        t0 = time.perf_counter()
        tk_ns, owned = 0, 0
        if backend == "gloo":
            for _ in range(args.steps):
                mask, stats = step()
                tk_ns += int(stats[4])
                owned = int(stats[0])
        else:
            # the K steps are stream-ordered: every step's screen, refine, export, all-gather and
            # ladder are enqueued behind the previous step's, the host waits once for all K
            done = 0
            while done < args.steps:  # batches of at most 64 stream-ordered steps (64 collectives in flight)
                n = min(64, args.steps - done)
                for mask, stats in fdist.prune_steps_sharded_device(ens, n, MAX_RMSD, rank=rank, world=world,
                                                                    device=torch.device("cuda", local_rank)):
                    tk_ns += int(stats[4])
                    owned = int(stats[0])
                done += n
        barrier()
        elapsed = time.perf_counter() - t0
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
        t_kernel_ms = tk_ns / args.steps * 1e-6  # rank 0's screen kernel, its own row blocks
        owned_pairs_rank0 = owned

    survivors = int(mask.sum())
    expected = len(np.unique(assign))
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = pairs_total * args.steps / elapsed
        bytes_per_alignment = 2 * N_ATOMS * 3 * 8 + 16
        out = {
            "metric": "conformer-pair RMSD alignments/s",
            "value": value,
            "unit": "alignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n_conf}-conformer x {N_ATOMS}-atom ensemble, all-pairs Kabsch RMSD "
                                   f"+ {MAX_RMSD} A prune (BASELINE configs[1]; conformers scale as sqrt(n_gpus))",
                       "n_conformers": n_conf, "n_atoms": N_ATOMS, "max_rmsd": MAX_RMSD,
                       "pairs_per_step": pairs_total,
                       "sharding": f"row blocks of 128 dealt in snake order over {world} rank(s); one all-gather "
                                   "of similar-pair lists, ladder replayed on every rank",
                       "host_sync": ("once for the K steps (stream-ordered steps, one pinned result slot each)"
                                     if (not sharded or backend != "gloo") else "once per step"),
                       "step_overlap": ("screens in order on one stream; refine + ladder + result copy of step r run "
                                        "beside the screen of step r+1 (two workspaces over the same resident "
                                        "coordinates; FC_BENCH_LANES=1 turns it off)"
                                        if (not sharded and os.environ.get("FC_BENCH_LANES") != "1") else
                                        "screens in order on one stream; refine + export + all-gather + ladder of step r "
                                        "run beside the screen of step r+1 (ensemble and twin workspace, two message "
                                        "buffers; FC_SHARD_LANES=1 turns it off)"
                                        if (sharded and backend != "gloo" and os.environ.get("FC_SHARD_LANES") != "1")
                                        else "none"),
                       "exchange": ("none (single GPU, resident step)" if not sharded else
                                    "host lists through gloo" if backend == "gloo" else
                                    "device-resident: export kernel -> RCCL all_gather_into_tensor -> ladder, "
                                    "no host sync inside a batch of 64 steps")},
            "pruned_ensembles_per_s": args.steps / elapsed,
            "rmsd_values_per_s": (pairs_total / (values_ms * 1e-3)) if not sharded else None,
            "pruned_ensembles_per_s_host_in_mask_out": (1.0 / min(host_in_out_s)) if (not sharded and host_in_out_s) else None,
            "survivors": survivors,
            "survivors_expected": expected,
            "mask_ok": survivors == expected,
        }
        if t_kernel_ms is not None:
            # dominant kernel: k_simbits_screen, HIP events on the library's stream
            owned_pairs = pairs_total if not sharded else owned_pairs_rank0  # pairs of the timed launch
            achieved = owned_pairs * bytes_per_alignment / (t_kernel_ms * 1e-3) / 1e9
            # The bound that binds: the screen kernel runs its contraction on the matrix pipe
            # (DESIGN.md section 5).  achieved = executed MFMA flops per launch (9 covariance
            # entries x K = atoms padded to 4, per pair) / HIP-event kernel time.  Default screen:
            # fp32 MFMA + fp32 polynomial against proven bounds, every pair it lets through is
            # decided by the exact fp64 refine (peak 157.3 TFLOP/s dense fp32, f32-input MFMA);
            # FC_SCREEN_F32=0: the fp64 screen (peak 78.6 TFLOP/s dense fp64).
            kind = _lib.screen_last_kind()
            f32 = kind == 32
            peak = 157.3 if f32 else 78.6
            kname = "k_simbits_screen_mfma_f32" if f32 else "k_simbits_screen_mfma"
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_screen_f32.json" if f32 else "r01_pmc_screen_final.json")
            traffic, traffic_src = None, None
            if world == 1 and n_conf == N_CONF and os.path.exists(pmc):
                traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
                traffic_src = os.path.relpath(pmc, ROOT)
            a4 = (N_ATOMS + 3) // 4 * 4
            flops_per_alignment = 2 * 9 * a4
            tflops = owned_pairs * flops_per_alignment / (t_kernel_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": kname, "achieved": tflops, "peak": peak,
                "unit": "TFLOP/s", "frac": tflops / peak, "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": t_kernel_ms, "flops_per_alignment": flops_per_alignment,
                "kernel_ms_source": ("HIP events on the kernel's stream around every %sth launch of the timed region "
                                     "(an event pair costs the stream ~14 us; FC_BENCH_EVENT_STRIDE=1 times all)"
                                     % os.environ.get("FC_BENCH_EVENT_STRIDE", "8")) if not sharded else
                                    "HIP events around rank 0's launches",
                "dtype": "f32" if f32 else "f64",
            }
            out["dtype"] = "f32 screen + f64 exact refine" if f32 else "f64"
            # the north star's view: algorithmic bytes (two conformers in, rmsd + maxdev out)
            # against the 8 TB/s HBM roof; > 1 because a staged tile serves 64-256 partners
            out["roofline_hbm"] = {
                "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_alignment": bytes_per_alignment,
                "compulsory_bytes": n_conf * N_ATOMS * 24 + n_conf * ((n_conf + 63) // 64) * 8,
            }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(coords)
        sys.stdout.flush()
        if stdout_fd is not None:
            os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        if stdout_fd is not None:
            os.dup2(2, 1)
    if sharded:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
