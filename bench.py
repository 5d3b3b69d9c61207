#!/usr/bin/env python
"""bench.py -- conformer-pair RMSD alignments/s (+ pruned ensembles/s) on MI355X.

The line's top-level fields are the reference-precision measurement in SURVEY.md 8d's unit of work:

  value / ms_per_step / dtype "f64" / roofline
        COMPLETE fp64 alignments per second: a step = one pass of `rmsd_and_max` (firecode/utils.py:499:
        two conformers in -> Kabsch RMSD (from the largest eigenvalue) AND max per-atom deviation of the explicit rotated difference out)
        over EVERY pair of the HBM-resident ensemble (fc_bench_rmsd_and_max_all: covariance tiles on the fp64
        matrix pipe, rotation and deviation pass per pair, two dense (N, N) fp64 outputs that stay in HBM).
        K steps enqueued back to back, one host wait, bracketed by barrier + device synchronisation.
        `roofline` = that pass's dominant kernel, k_simbits_screen_mfma<4, 2, 64, true>, HIP events on its stream around
        every launch of the timed region, against the fp64 peak by 8d's 53 A + 600 flop per alignment.

Extra blocks of the same line (each with its own device timing, outside the timed region):

  prune_path   the pruning step the reference's prune_by_rmsd performs on the same ensemble -- all-pairs
               similarity DECISIONS (split-half f16 screen, fp64-exact refine of the candidates) + k-ladder ->
               survivor mask, bit-identical to the fp64 oracle's: ms_per_step, pair_decisions_per_s,
               pruned_ensembles_per_s, roofline of its screen kernel.  (Rounds 1-2 reported this as `value`.)
  fp64_path    the same pruning step with the fp64 screen (the reference's arithmetic in every kernel).
  config.secondary  an ensemble without cluster structure; its refine kernel has its own HBM roofline.
  cpu_baseline the oracle's rmsd_and_max on a bounded sample, one host core.

Workloads (BASELINE.json `configs`), chosen with --workload:

  cfg2  configs[1]: 10 000-conformer x 50-atom float64 ensemble.  THE workload of `value` at every N
        (default): on N GPUs the conformer count grows as sqrt(N) -- weak scaling, pairs per GPU constant,
        rows of the pair matrix dealt to the ranks in snake order, no exchange for the alignments; the
        prune_path of N > 1 exchanges the similar-pair lists with ONE RCCL all-gather.
  cfg4  configs[3]: 100 000-conformer x 80-atom ensemble sharded over the GPUs, one RCCL all-gather:
        weak-scaling family n_conf = 100 000 * sqrt(N / 8) (N = 8 IS configs[3]); reported as the block
        `cfg4_family` of every N > 1 line, or alone with --workload cfg4.
  cfg5  configs[4]: bimolecular rigid embed, 500 x (500 N / 8) conformer pairs x 512 rototranslations.

N > 1: `python bench.py --gpus N` starts N fresh rank processes itself (the parent never loads the library
or touches a GPU) and relays rank 0's line; under an external launcher that sets RANK / LOCAL_RANK /
WORLD_SIZE (python -m torch.distributed.run ...) the process is a rank.  The ranks never import torch: RCCL
sits behind the library's C ABI (firecode_amd.dist.comm_init_from_env).
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
    # ... and the same arithmetic through the oracle's VECTORISED form (stacked einsum + stacked 3x3 SVD over blocks of
    # 20 000 pairs): what a NumPy user who batches the pairs gets from one core -- the per-pair figure above is bound
    # by Python / LAPACK call overhead, not by arithmetic
    t1, done_v, block = time.perf_counter(), 0, 20000
    while time.perf_counter() - t1 < min(8.0, budget_s) and done_v + block <= len(iu):
        o.rmsd_and_max_batch(X[iu[done_v:done_v + block]], X[ju[done_v:done_v + block]])
        done_v += block
    dtv = time.perf_counter() - t1
    return {"value": done / dt, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"{done} conformer pairs among the first {n0} conformers of the workload, "
                      f"oracle rmsd_and_max (NumPy, LAPACK 3x3 SVD per pair), {dt:.1f} s; single process, "
                      f"{len(os.sched_getaffinity(0))} host cores visible (3x3 LAPACK calls do not thread)",
            "vectorised": {"value": done_v / dtv if done_v else None, "unit": "alignments/s", "cores": 1, "kind": "port",
                           "sample": f"{done_v} pairs of the same set through oracle.rmsd_and_max_batch (stacked NumPy einsum + "
                                     f"SVD, blocks of {block}), {dtv:.1f} s"}}


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


def pmc_traffic(files, n_conf, n_atoms, world=1):
    """`roofline.traffic` from a kept rocprofv3 --pmc summary -- only when that file's workload IS the line's
    (same conformers x atoms, one GPU); otherwise null."""
    if world != 1 or n_conf is None:
        return None, None
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            rec = json.load(open(path))
            if rec.get("n_conformers") == n_conf and rec.get("n_atoms") == n_atoms:
                return rec["traffic_bytes_per_launch"], ("from_file: profiles/" + name + " (rocprofv3 --pmc passes of an earlier run of "
                                                          "this kernel on this workload, not of this run)")
    return None, None


def pmc_held_clock(files, n_conf, n_atoms, frac, world=1):
    """The clock the chip held under this kernel in the kept counter run (GRBM_GUI_ACTIVE / 8 / kernel time: MI355X_MICROARCH.md,
    DVFS give-back) and what `frac` -- priced, as the rules say, at the 2.4 GHz peak -- comes to at that clock; None when
    there is no such record for this workload."""
    if world != 1 or n_conf is None:
        return None
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            rec = json.load(open(path))
            if rec.get("n_conformers") == n_conf and rec.get("n_atoms") == n_atoms and rec.get("effective_clock_GHz"):
                ghz = float(rec["effective_clock_GHz"])
                return {"held_clock_GHz": ghz, "frac_of_the_peak_at_that_clock": frac * 2.4 / ghz,
                        "source": "profiles/" + name + " (GRBM_GUI_ACTIVE / 8 / mean kernel duration of the kernel-trace run of the same "
                                  "command; an earlier profiled run, not this one; reads high on dispatches shorter than ~0.3 ms)"}
    return None


def pmc_issue_cycles(files, n_conf, n_atoms, world=1):
    """Issue-model cycles per 16 x 16 sub-tile and wave of the split-half screen from the kept counter run: MFMAs per sub-tile
    = 27 per k-step of 32 atoms; the other vector instructions = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / sub-tiles."""
    if world != 1 or n_conf is None:
        return None
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            rec = json.load(open(path))
            if rec.get("n_conformers") == n_conf and rec.get("n_atoms") == n_atoms and rec.get("SQ_INSTS_MFMA"):
                per = 27 * ((n_atoms + 31) // 32)
                subtiles = rec["SQ_INSTS_MFMA"] / per
                valu = (rec["SQ_INSTS_VALU"] - rec["SQ_INSTS_MFMA"]) / subtiles
                return {"mfma_per_subtile": per, "other_vector_instructions_per_subtile": valu, "matrix_pipe": per * 16,
                        "issue": per * 8 + valu * 4, "source": "profiles/" + name}
    return None


def screen_roofline(_lib, kernel_ms, owned_pairs, n_atoms, traffic_file=True, world=1, stats=None, n_conf=None):
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
        traffic, src = pmc_traffic(("r05_pmc_screen_h2.json", "r05_pmc_screen_h2_cfg4_member.json", "r03_pmc_screen_h2.json", "r03_pmc_screen_h2_cfg4_member.json"), n_conf if traffic_file else None,
                                   n_atoms, world)
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
                "clock": pmc_held_clock(("r05_pmc_screen_h2.json", "r05_pmc_screen_h2_cfg4_member.json", "r03_pmc_screen_h2.json", "r03_pmc_screen_h2_cfg4_member.json"), n_conf if traffic_file else None,
                                        n_atoms, tflops / PEAK_F16_MFMA, world),
                "issue_model": {"what": "tools/ubench_issue_model.hip (profiles/r03_issue_model_f16mfma_valu*.txt): one SIMD issues a "
                                        "v_mfma_f32_16x16x32_f16 in 8 and an fp32 vector instruction in ~4 of its cycles, whichever "
                                        "waves they come from and however they are interleaved; the matrix pipe is busy 16 per MFMA",
                                "cycles_per_16x16_subtile": pmc_issue_cycles(("r05_pmc_screen_h2.json", "r05_pmc_screen_h2_cfg4_member.json", "r03_pmc_screen_h2.json", "r03_pmc_screen_h2_cfg4_member.json"),
                                                                             n_conf if traffic_file else None, n_atoms, world)},
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
    traffic, src = None, None  # (no round-3 counter passes of the fp32 / fp64 screens: profiles/r02_pmc_screen_f32|f64.json are round 2's)
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


def rehearsal_scale():
    """FC_BENCH_REHEARSAL_SCALE (0 < s <= 1, default 1): conformer counts of every family times s -- a REHEARSAL of the
    N-rank launch on one device (FC_BENCH_SAME_DEVICE=1), stated in the line as config.rehearsal_scale; its numbers are
    not measurements."""
    try:
        s = float(os.environ.get("FC_BENCH_REHEARSAL_SCALE", "1"))
    except ValueError:
        s = 1.0
    return s if 0.0 < s <= 1.0 else 1.0


def fail_here(rank, where):
    """FC_BENCH_FAIL_RANK="k:before" | "k:after": rank k raises before / after the communicator is up (the launcher's
    exit codes under a failing rank are a test of their own)."""
    spec = os.environ.get("FC_BENCH_FAIL_RANK")
    if spec and spec == f"{rank}:{where}":
        raise RuntimeError(f"FC_BENCH_FAIL_RANK: rank {rank} fails {where} the communicator is created")


def spawn_ranks(n):
    """`python bench.py --gpus N` typed as is: start N fresh rank processes (this parent never imports the
    library or touches a GPU), relay rank 0's single JSON line, exit non-zero if any rank does."""
    import subprocess
    import tempfile
    import uuid

    idfile = os.path.join(tempfile.gettempdir(), f"fc_comm_{uuid.uuid4().hex}.id")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=rank_env(os.environ, r, n, idfile),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    line, failed, open_out = b"", None, True
    try:
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        while True:
            if open_out and sel.select(timeout=0.2):
                chunk = os.read(procs[0].stdout.fileno(), 1 << 16)
                if chunk:
                    line += chunk
                else:
                    open_out = False
                    sel.unregister(procs[0].stdout)
            elif not open_out:
                time.sleep(0.05)
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes) and not open_out:
                break
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None and failed is not None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        try:
            os.remove(idfile)
        except OSError:
            pass
    if failed is not None and open_out:  # whatever rank 0 had already printed survives a failing rank
        try:
            while True:
                chunk = os.read(procs[0].stdout.fileno(), 1 << 16)
                if not chunk:
                    break
                line += chunk
        except OSError:
            pass
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        sys.exit(failed[1] if 0 < failed[1] < 256 else 1)


def start_extras_guard(emit, rank, headline, seconds):
    """Several ranks, blocks that exchange data over a communicator no round has seen on more than one device: if they do
    not come back within `seconds`, every rank's own timer ends its process -- rank 0 after printing the line with the
    headline measurement it already holds (`emit`).  Returns the timer (cancel it when the blocks are through)."""
    import threading

    def fire():
        if rank == 0:
            d = dict(headline)
            d["extras_error"] = f"the exchanging extra blocks did not finish within {seconds} s; the ranks ended themselves"
            emit(d)
        sys.stderr.write(f"bench.py: rank {rank}: extra blocks timed out after {seconds} s\n")
        sys.stderr.flush()
        os._exit(4)  # a hang is a failure: the launcher sees it, the line (with extras_error) is already out

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def rank_env(base, rank, world, idfile):
    """Environment of rank `rank` of a self-spawned launch (what an external launcher would set)."""
    env = dict(base)
    # FC_BENCH_SAME_DEVICE=1 (rehearsal on a one-GPU box): every rank on device 0 -- the launch, the rendezvous and
    # ncclCommInitRank run for real, RCCL refuses the duplicate device, the ranks go on without a communicator
    local = "0" if base.get("FC_BENCH_SAME_DEVICE") == "1" else str(rank)
    env.update(RANK=str(rank), LOCAL_RANK=local, WORLD_SIZE=str(world), FC_COMM_ID_FILE=idfile,
               FC_BENCH_SPAWNED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.pop("FC_BENCH_FORCE_SPAWN", None)
    return env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("auto", "cfg2", "cfg4", "cfg5"), default="auto",
                    help="auto = cfg2 (BASELINE configs[1]; conformers x sqrt(n_gpus)) with the cfg4 family as an extra block on several GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region (profile runs): no prune_path / fp64_path / secondary / host-in legs")
    ap.add_argument("--cpu-baseline-configs", action="store_true",
                    help="only time the CPU oracle on samples of BASELINE configs 3 and 5 (no GPU needed) and exit")
    args = ap.parse_args()
    if args.cpu_baseline_configs:
        print(json.dumps({"cpu_baseline_other_configs": cpu_baseline_other_configs(),
                          "host_cores_visible": len(os.sched_getaffinity(0))}))
        return

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("FC_BENCH_FORCE_SPAWN") == "1"):
        spawn_ranks(args.gpus)
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL, before the HSA runtime starts
    if os.environ.get("FC_BENCH_SAME_DEVICE") == "1":  # rehearsal of a launcher's N ranks on a one-GPU box (see rank_env)
        os.environ["LOCAL_RANK"] = "0"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world  # under a launcher the launch decides
    workload = args.workload if args.workload != "auto" else "cfg2"

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
    comm = (world > 1 or workload in ("cfg4", "cfg5") or os.environ.get("FC_BENCH_FORCE_SHARDED") == "1"
            or os.environ.get("FC_BENCH_SPAWNED") == "1")
    # Rank coordination of the timed regions (barrier, max over ranks): the RCCL communicator where there is one.  The
    # headline path has no data-path collective, so a communicator that cannot be created on some node must not take the
    # measurement with it: the ranks then agree through files (firecode_amd.dist.HostRendezvous), shard by logical rank,
    # and the blocks that DO exchange data (sharded prune, cfg4 / cfg5 families) are reported as skipped with the reason.
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    rv = fdist.HostRendezvous(rank, world) if world > 1 else None
    comm_error = None
    fail_here(rank, "before")
    if comm:
        fc.init(local_rank)  # a missing device is not a communicator problem: it ends the launch here, with its own message
        try:
            rank, world, local_rank = fdist.comm_init_from_env()
        except Exception as exc:  # noqa: BLE001 -- whatever it was, every rank has to learn of it
            if world == 1:
                raise
            comm_error = f"{type(exc).__name__}: {exc}"
        if rv is not None:
            oks = rv.allgather(b"\x01" if comm_error is None else b"\x00" + comm_error.encode()[:400])
            bad = [(r, o[1:].decode(errors="replace")) for r, o in enumerate(oks) if o[:1] != b"\x01"]
            if bad:
                if comm_error is None:
                    _lib.comm_destroy()
                comm_error = "; ".join(f"rank {r}: {m}" for r, m in bad)
                comm = False
                fc.init(local_rank)  # (a missing device is not a communicator problem: this raises and the launch fails)
                if workload in ("cfg4", "cfg5"):
                    raise RuntimeError("the cfg4 / cfg5 lines exchange data between the ranks and need the RCCL communicator: " + comm_error)
                _lib.call("fc_debug_comm_loopback", rank, world)  # logical rank / world for the row-block dealing, no exchange
                sys.stderr.write(f"bench.py: rank {rank}: no RCCL communicator ({comm_error}); ranks coordinated through files\n")
    else:
        fc.init(0)
    fail_here(rank, "after")

    def barrier():
        # several ranks: through the files, whether or not there is a communicator -- the timed regions bracket device work
        # that every bench hook has already synchronised, and the headline must not depend on a collective
        if rv is not None:
            rv.barrier()
        elif comm:
            _lib.comm_barrier()  # a 1-byte all-gather + device synchronisation (the 1-rank communicator of the forced modes)

    def max_over_ranks(x):
        return float(x) if rv is None else rv.max(x)

    import threading

    emit_lock, emitted = threading.Lock(), [False]

    def emit(line_dict):
        """the ONE JSON line, once (rank 0)"""
        with emit_lock:
            if emitted[0] or rank != 0:
                return
            emitted[0] = True
            if rehearsal_scale() != 1.0 and isinstance(line_dict.get("config"), dict):
                line_dict["config"]["rehearsal_scale"] = rehearsal_scale()
                line_dict["config"]["rehearsal_note"] = "conformer counts scaled down for a launch rehearsal: NOT a measurement"
            if world > 1:
                line_dict["rank_coordination"] = ("barrier / max over ranks through files (firecode_amd.dist.HostRendezvous); RCCL communicator " +
                                                  ("up: used by the exchanging blocks" if comm else "NOT created: " + str(comm_error)))
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
            print(json.dumps(line_dict), flush=True)
            os.dup2(2, 1)

    def guard_extras(headline, seconds):
        return start_extras_guard(emit, rank, headline, seconds)

    ctx = dict(args=args, fc=fc, _lib=_lib, fdist=fdist, syn=syn, rank=rank, world=world, comm=comm,
               barrier=barrier, max_over_ranks=max_over_ranks, comm_error=comm_error, guard_extras=guard_extras)
    if workload == "cfg5":
        out = run_cfg5(args, fc, _lib, fdist, syn, rank, world, barrier, max_over_ranks)
    elif workload == "cfg4":
        out = run_prune_line(ctx, "cfg4")
    else:
        out = run_alignments(ctx)

    if rank == 0:
        emit(out)
    if comm:
        _lib.comm_barrier()
        _lib.comm_destroy()
    if rv is not None:
        rv.close()
    if ctx.get("value_check_failed"):
        sys.exit(3)  # the line is out (with value_check.ok false); a wrong number must not look like a measurement


# ----------------------------------------------------------------------------------------------
# the headline: complete fp64 alignments of all pairs (cfg2 family)
# ----------------------------------------------------------------------------------------------
def complete_roofline(kernel_ms, owned_pairs, n_conf, n_atoms, world):
    fl = FLOPS_PER_ALIGNMENT(n_atoms)
    tflops = owned_pairs * fl / (kernel_ms * 1e-3) / 1e12
    traffic, src = pmc_traffic(("r05_pmc_complete.json", "r04_pmc_complete.json", "r03_pmc_complete.json"), n_conf, n_atoms, world)
    return {"bound": "mfma", "kernel": "k_simbits_screen_mfma<4, 2, 64, true>", "achieved": tflops, "peak": PEAK_F64_MFMA,
            "unit": "TFLOP/s", "frac": tflops / PEAK_F64_MFMA, "traffic": traffic, "traffic_source": src,
            "kernel_ms": kernel_ms, "flops_per_alignment": fl, "dtype": "f64",
            "clock": pmc_held_clock(("r05_pmc_complete.json", "r04_pmc_complete.json", "r03_pmc_complete.json"), n_conf, n_atoms, tflops / PEAK_F64_MFMA, world),
            "note": "achieved = SURVEY 8d's algorithmic flops of one complete alignment (53 A + 600) x pairs of one launch / the "
                    "kernel's mean HIP-event duration; peak = the fp64 rate of the matrix pipe, which on this chip is also the "
                    "fp64 vector rate -- the kernel runs its covariance on the first and rotation + deviation pass on the second; "
                    "the flops are the algorithmic ones whatever the kernel issues (round 5: the rmsd comes from the eigenvalue, "
                    "one add per atom and pair less than 8d counts)",
            "kernel_ms_source": "HIP events on the kernel's stream around every launch of the timed region"
                                + ("" if world == 1 else "; rank 0's launches")}


def value_check_pairs(n_conf, rank, world, fdist, n_pairs=2048, seed=1234):
    """Pairs (i < j) of the rows THIS rank owns (row blocks of 128 in snake order): random ones, plus the corners of the
    launch -- first / last owned rows, both sides of 16- / 128-row boundaries, the last (partial) column tile, the tip of
    the triangle where the item table ends in half-row-block items."""
    rng = np.random.default_rng(seed + rank)
    rows = np.flatnonzero(fdist.owner_of_rows(n_conf, world, 128) == rank)
    rows = rows[rows < n_conf - 1]
    i = rng.choice(rows, n_pairs - 448)
    j = rng.integers(i + 1, n_conf)
    corner_rows = np.concatenate([rows[:16], rows[-16:], rows[(rows % 128 == 0) | (rows % 128 == 127)][:32],
                                  rows[rows >= n_conf - 3000][::max(1, len(rows[rows >= n_conf - 3000]) // 48)][:48]])
    ci = rng.choice(corner_rows, 448)
    cj = np.where(rng.random(448) < 0.3, np.maximum(ci + 1, n_conf - 1 - rng.integers(0, 16, 448)),
                  np.minimum(ci + 1 + rng.integers(0, 130, 448), n_conf - 1))
    return np.concatenate([i, ci]).astype(np.int64), np.concatenate([j, cj]).astype(np.int64)


def value_check(coords, iu, ju, r, d, tol=1e-10):
    """The sampled elements of the last timed pass's outputs against the oracle's rmsd_and_max (the checker, after the
    clock has stopped): rmsd within 1e-10; max deviation within 1e-10 + the pair's conditioning bound."""
    from oracle import cpu_ref as o

    r0, d0 = o.rmsd_and_max_batch(coords[iu], coords[ju], center=True)
    bound = o.rotation_error_bound_batch(coords[iu], coords[ju], center=True)
    err_r, err_d = np.abs(r - r0), np.abs(d - d0)
    ok = bool(np.all(np.isfinite(r)) and np.all(np.isfinite(d)) and err_r.max() < tol and np.all(err_d <= tol + bound))
    return {"pairs": int(len(iu)), "max_abs_err_rmsd": float(err_r.max()), "max_abs_err_maxdev": float(err_d.max()),
            "tolerance": tol, "largest_conditioning_allowance": float(bound.max()), "ok": ok,
            "what": "elements of the two (N, N) outputs as the LAST timed pass left them (fc_bench_rmsd_and_max_all_sampled: "
                    "gathered on the device behind that pass, inside the timed region) against oracle.rmsd_and_max_batch "
                    "(firecode/utils.py:494-504), compared after the clock stopped; rows owned by this rank incl. first / last "
                    "rows, 16- / 128-row boundaries, the last column tile and the half-item tail of the item table"}


def run_alignments(ctx):
    args, fc, _lib, syn = ctx["args"], ctx["fc"], ctx["_lib"], ctx["syn"]
    rank, world, barrier, max_over_ranks = ctx["rank"], ctx["world"], ctx["barrier"], ctx["max_over_ranks"]
    n_atoms, seed = 50, 2
    n_conf = int(round(10000 * np.sqrt(world) * rehearsal_scale()))
    steps = 50 if args.steps is None else args.steps
    warmup = 5 if args.warmup is None else args.warmup
    what = (f"{n_conf}-conformer x {n_atoms}-atom ensemble, complete Kabsch alignment (rmsd + max deviation, fp64) of all pairs "
            "(BASELINE configs[1]" + ("" if world == 1 else "; conformers scaled by sqrt(n_gpus): pairs per GPU constant, rows of "
                                      "the pair matrix dealt to the ranks, no exchange") + ")")
    coords, atoms, assign = syn.synthetic_ensemble(n_conf, n_atoms, seed=seed)
    ens = fc.DeviceEnsemble(coords, center=True)  # resident in HBM from here on
    pairs_total = n_conf * (n_conf - 1) // 2

    chk_i, chk_j = value_check_pairs(n_conf, rank, world, ctx["fdist"])
    sampled = [None, None]

    def run(n, sample=False):
        k = t = 0.0
        st = None
        done = 0
        while done < n:
            m = min(1024, n - done)
            if sample and done + m == n:  # the last batch: its last pass's outputs are read back at the sampled pairs
                km, tm, st, sampled[0], sampled[1] = ens.bench_rmsd_and_max_all_sampled(chk_i, chk_j, m)
            else:
                km, tm, st = ens.bench_rmsd_and_max_all(m)
            k += km * m
            t += tm
            done += m
        return k / n, t, st

    run(2, sample=True)  # output matrices, item table, event pool, sample buffers: outside the timed region whatever W is
    if warmup:
        run(warmup)
    barrier()
    t0 = time.perf_counter()
    k_ms, dev_ms, st = run(steps, sample=True)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    owned = int(st[0])
    check = value_check(coords, chk_i, chk_j, sampled[0], sampled[1])
    ranks_failed = int(round(max_over_ranks(0.0 if check["ok"] else 1.0)))
    if not check["ok"]:
        sys.stderr.write(f"bench.py: rank {rank}: value_check FAILED: {json.dumps(check)}\n")
    ctx["value_check_failed"] = bool(ranks_failed)

    out = None
    if rank == 0:
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
            "config": {"workload": what, "baseline_config": "configs[1]", "n_conformers": n_conf, "n_atoms": n_atoms,
                       "pairs_per_step": pairs_total, "pairs_per_step_rank0": owned,
                       "value_counts": "complete alignments: for EVERY conformer pair of the step the optimal rotation (Kabsch), the "
                                       "max per-atom deviation from the explicit rotated difference and the RMSD from the pair's largest "
                                       "eigenvalue ((Gp + Gq) - 2 lambda: the sum of squares of that difference under the optimal rotation; "
                                       "pairs closer than ~1e-3 A from the explicit sum), all fp64 -- "
                                       "the (rmsd, maxdev) contract of rmsd_and_max (firecode/utils.py:499); outputs: two dense "
                                       "(N, N) fp64 matrices in HBM",
                       "sharding": (f"row blocks of 128 of the pair matrix dealt in snake order over {world} ranks, every rank "
                                    "keeps the whole ensemble and its rows of the outputs; no exchange") if world > 1 else "none (single GPU)",
                       "host_sync": "once per batch of <= 1024 stream-ordered steps",
                       "device_ms_per_step": dev_ms / steps},
            "fixup_pairs_last_step": int(st[1]),
            "value_check": dict(check, all_ranks_ok=not ranks_failed),
        }
        out["roofline"] = complete_roofline(k_ms, owned, n_conf, n_atoms, world)
        bytes_per_alignment = 2 * n_atoms * 24 + 16
        achieved = owned * bytes_per_alignment / (k_ms * 1e-3) / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_simbits_screen_mfma<4, 2, 64, true>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": out["roofline"]["traffic"],
                               "algorithmic_bytes_per_alignment": bytes_per_alignment,
                               "compulsory_bytes": n_conf * n_atoms * 24 + 16 * owned,
                               "note": "the north star's view: SURVEY 8d's 2*A*24+16 B per alignment against the 8 TB/s roof; > 1 = a "
                                       "staged tile of 64 conformers serves 128 partners; the compulsory traffic is the ensemble "
                                       "once + the two output matrices"}

    if world > 1:
        # the family's n_gpus = 1 member (same pairs per GPU), on rank 0 alone, outside the timed region
        if rank == 0:
            c1, _, _ = syn.synthetic_ensemble(int(round(10000 * rehearsal_scale())), n_atoms, seed=seed)
            _lib.call("fc_debug_comm_loopback", 0, 1)  # rank 0 computes every row, as a single GPU does
            try:
                with fc.DeviceEnsemble(c1, center=True) as e1:
                    e1.bench_rmsd_and_max_all(2)
                    k1, t1, s1 = e1.bench_rmsd_and_max_all(min(steps, 20))
            finally:
                if ctx["comm"]:
                    _lib.call("fc_debug_comm_loopback", -1, 0)
                else:
                    _lib.call("fc_debug_comm_loopback", rank, world)  # (no communicator: back to this rank's logical share)
            v1 = int(s1[0]) * min(steps, 20) / (t1 * 1e-3)
            out["scaling_family_n1"] = {"n_conformers": int(round(10000 * rehearsal_scale())), "pairs_per_step": int(s1[0]), "kernel_ms": k1,
                                        "ms_per_step": t1 / min(steps, 20), "value": v1,
                                        "note": "the N = 1 member of the family (BASELINE configs[1] itself), measured on rank 0 in "
                                                "this run while the other ranks wait (device time, first launch to last)"}
            out["efficiency"] = out["value"] / (world * v1)
        barrier()
    if ctx["comm"]:
        r_, w_ = _lib.comm_info()
        if rank == 0:
            out["ranks_seen"] = int(w_)
    elif world > 1 and rank == 0:
        out["ranks_seen"] = world  # (every rank answered the file rendezvous)

    if not args.no_extras and world > 1 and not ctx["comm"]:
        if rank == 0:
            out["prune_path"] = {"skipped": "the sharded prune exchanges its similar-pair lists over RCCL; " + str(ctx["comm_error"])}
    elif not args.no_extras and world > 1:
        # the blocks that exchange data between the ranks: whatever goes wrong in them -- the same on every rank, the calls
        # are collective -- must not take the headline measurement above with it
        timer = ctx["guard_extras"](out if rank == 0 else {}, int(os.environ.get("FC_BENCH_EXTRAS_TIMEOUT_S", "300")))
        try:
            blk = prune_block(ctx, "cfg2", ens, coords, atoms, assign, n_conf, n_atoms, steps_default=200, warmup_default=20)
            if rank == 0:
                out["prune_path"] = blk
                out["pruned_ensembles_per_s"] = blk["pruned_ensembles_per_s"]
            blk4 = run_prune_line(ctx, "cfg4", as_block=True)
            if rank == 0:
                out["cfg4_family"] = blk4
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write(f"bench.py: rank {rank}: extra blocks failed: {type(exc).__name__}: {exc}\n")
            if rank == 0:
                out["extras_error"] = f"{type(exc).__name__}: {exc}"
        finally:
            timer.cancel()
    elif not args.no_extras:
        blk = prune_block(ctx, "cfg2", ens, coords, atoms, assign, n_conf, n_atoms, steps_default=200, warmup_default=20)
        if rank == 0:
            out["prune_path"] = blk
            out["pruned_ensembles_per_s"] = blk["pruned_ensembles_per_s"]
            extras_single_gpu(args, out, fc, _lib, syn, ens, coords, atoms, n_conf, n_atoms, pairs_total)
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(coords)
        out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


# ----------------------------------------------------------------------------------------------
# cfg2 / cfg4: all-pairs RMSD prune (pair decisions + ladder -> survivor mask)
# ----------------------------------------------------------------------------------------------
def prune_block(ctx, workload, ens, coords, atoms, assign, n_conf, n_atoms, steps_default, warmup_default, steps=None, warmup=None):
    """K stream-ordered prunes of the resident ensemble, bracketed like the main region -> dict (rank 0) / None."""
    _lib, rank, world = ctx["_lib"], ctx["rank"], ctx["world"]
    barrier, max_over_ranks = ctx["barrier"], ctx["max_over_ranks"]
    sharded = world > 1 or workload == "cfg4" or os.environ.get("FC_BENCH_FORCE_SHARDED") == "1"
    steps = steps_default if steps is None else steps
    warmup = warmup_default if warmup is None else warmup
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
        return None
    roof = screen_roofline(_lib, t_kernel_ms, owned, n_atoms, world=world, stats=stats, n_conf=n_conf)
    roof["kernel_ms_source"] = ("HIP events on the kernel's stream around every %sth launch of the timed region "
                                "(an event pair costs the stream ~14 us; FC_BENCH_EVENT_STRIDE=1 times all)"
                                % os.environ.get("FC_BENCH_EVENT_STRIDE", "8")) + ("" if world == 1 else "; rank 0's launches")
    blk = {
        "what": "prune_by_rmsd's step on the resident ensemble: all-pairs similarity decisions (conservative screen + exact fp64 "
                "refine of the candidates) + k-ladder replay -> survivor mask, bit-identical to the fp64 oracle's",
        "dtype": {"f32": "f32 screen + f64 exact refine", "f16x2": "f16x2 screen (split-half, fp32-accurate) + f64 exact refine",
                  "f64": "f64"}[roof["dtype"]],
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "pair_decisions_per_s": pairs_total * steps / elapsed, "pruned_ensembles_per_s": steps / elapsed,
        "n_conformers": n_conf, "n_atoms": n_atoms, "max_rmsd": MAX_RMSD, "pairs_per_step": pairs_total,
        "candidates_refined": int(stats[1]), "similar_pairs": int(stats[2]),
        "sharding": (f"row blocks of 128 dealt in snake order over {world} rank(s); one all-gather of similar-pair lists, "
                     "ladder replayed on every rank") if sharded else "none (single GPU)",
        "comm": "RCCL through libfc_hip.so's C ABI (fc_comm_init / ncclAllGather), no PyTorch in the ranks" if sharded else "none",
        "step_overlap": "screens in order on one stream; refine" + (" + export + all-gather" if sharded else "")
                        + " + ladder + result copy of step r run beside the screen of step r+1 (two workspaces over the same "
                          "resident coordinates); one host wait per batch of <= 512 steps",
        "roofline": roof,
    }
    blk.update(mask_checks(mask, assign))
    return blk


def run_prune_line(ctx, workload, as_block=False):
    """The cfg4 family (100 000 x sqrt(n_gpus / 8) conformers x 80 atoms, sharded prune with one all-gather): a line of
    its own with --workload cfg4, the block `cfg4_family` of every N > 1 line otherwise."""
    args, fc, _lib, syn = ctx["args"], ctx["fc"], ctx["_lib"], ctx["syn"]
    rank, world, barrier = ctx["rank"], ctx["world"], ctx["barrier"]
    n_atoms, seed = 80, 6
    n_conf = int(round(100000 * np.sqrt(world / 8.0) * rehearsal_scale()))
    steps = 20 if (args.steps is None or as_block) else args.steps
    warmup = 3 if (args.warmup is None or as_block) else args.warmup
    what = (f"{n_conf}-conformer x {n_atoms}-atom ensemble sharded over {world} GPU(s), all-pairs Kabsch RMSD + "
            f"{MAX_RMSD} A prune, one RCCL all-gather per prune (BASELINE configs[3] is the n_gpus = 8 member of "
            "this weak-scaling family: n_conf = 100 000 * sqrt(n_gpus / 8), 6.25e8 pairs per GPU)")
    coords, atoms, assign = syn.synthetic_ensemble(n_conf, n_atoms, seed=seed)
    with fc.DeviceEnsemble(coords, center=True) as ens:
        blk = prune_block(ctx, "cfg4", ens, coords, atoms, assign, n_conf, n_atoms, 20, 3, steps=steps, warmup=warmup)
    if rank == 0:
        blk["workload"] = what
    if world > 1:
        if rank == 0:
            # the family's n_gpus = 1 member (same pairs per GPU), on rank 0 alone
            n1 = int(round(100000 * np.sqrt(1 / 8.0) * rehearsal_scale()))
            c1, _, a1 = syn.synthetic_ensemble(n1, n_atoms, seed=seed)
            with fc.DeviceEnsemble(c1, center=True) as e1:
                e1.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=2, want_mask=False)
                _, s_ms, m1, _ = e1.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=min(steps, 20), want_mask=True)
            p1 = n1 * (n1 - 1) // 2
            blk["scaling_family_n1"] = {"n_conformers": n1, "pairs_per_step": p1, "ms_per_step": s_ms,
                                        "pair_decisions_per_s": p1 / (s_ms * 1e-3),
                                        "survivor_count_ok": mask_checks(m1, a1)["survivor_count_ok"],
                                        "note": "same kernels without the exchange, measured on rank 0 while the other ranks wait"}
            blk["efficiency"] = blk["pair_decisions_per_s"] / (world * blk["scaling_family_n1"]["pair_decisions_per_s"])
        barrier()
    if rank != 0:
        return None
    if as_block:
        return blk
    out = {"metric": "conformer-pair RMSD pair decisions/s (prune_by_rmsd step)", "value": blk["pair_decisions_per_s"],
           "unit": "pair decisions/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": blk["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": blk["dtype"], "data": "synthetic",
           "config": {"workload": what, "baseline_config": "configs[3]", "n_conformers": n_conf, "n_atoms": n_atoms,
                      "max_rmsd": MAX_RMSD, "pairs_per_step": blk["pairs_per_step"],
                      "value_counts": "pair decisions: every pair of the step decided as the reference's fp64 Kabsch decides it "
                                      "(bit-identical mask); NOT complete alignments -- the default line (--workload cfg2) reports those"},
           "pruned_ensembles_per_s": blk["pruned_ensembles_per_s"], "roofline": blk.pop("roofline"), "prune_path": blk}
    if ctx["comm"]:
        out["ranks_seen"] = int(_lib.comm_info()[1])
    if "efficiency" in blk:
        out["efficiency"] = blk["efficiency"]
    return out


def extras_single_gpu(args, out, fc, _lib, syn, ens, coords, atoms, n_conf, n_atoms, pairs_total):
    """Everything below is OUTSIDE the timed region of `value`; each leg has its own device timing."""
    steps, warmup = 200, 20
    # (a) the pruning step with the fp64 screen: the reference's arithmetic in every kernel
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
        r64 = screen_roofline(_lib, tk / steps, pairs_total, n_atoms, traffic_file=False)
        out["fp64_path"] = {"what": "prune_path's K steps with fc_screen_select(64): fp64 MFMA screen, fp64 refine, ladder",
                            "dtype": "f64", "ms_per_step": 1e3 * el / steps, "pair_decisions_per_s": pairs_total * steps / el,
                            "pruned_ensembles_per_s": steps / el, "mask_equals_default_path": None, "roofline": r64}
    finally:
        _lib.screen_select(0)
    _, _, mask32, _ = ens.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=1, want_mask=True)
    out["fp64_path"]["mask_equals_default_path"] = bool(np.array_equal(mask32, mask64))
    # (b) RMSD values only (no rotation, no max deviation)
    ens.rmsd_values(want_matrix=False)
    values_ms = min(ens.rmsd_values(want_matrix=False)[1] for _ in range(3))
    out["rmsd_values_per_s"] = pairs_total / (values_ms * 1e-3)
    # (c) an ensemble WITHOUT cluster structure: continuous RMSD distribution across the threshold
    Xc = syn.continuous_ensemble(n_conf, n_atoms, seed=11, thr=MAX_RMSD)
    with fc.DeviceEnsemble(Xc, center=True) as ec:
        ec.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=2, want_mask=False)
        kc, sc, mc, stc = ec.bench_prune(MAX_RMSD, 2 * MAX_RMSD, reps=20, want_mask=True)
        kind_c = _lib.screen_last_kind()
        refine = ec.bench_refine(MAX_RMSD, 2 * MAX_RMSD, reps=20)
    sec = {
        "workload": f"{n_conf} x {n_atoms}, continuous RMSD distribution (6 collective modes, ~1.5 % of the pairs below "
                    f"{MAX_RMSD} A, smooth density across the threshold): the case the clustered ensemble does not exercise",
        "ms_per_step": sc, "pair_decisions_per_s": pairs_total / (sc * 1e-3), "screen_kernel_ms": kc,
        "screen": {16: "f16x2", 32: "f32"}.get(kind_c, "f64"), "candidates_refined": int(stc[1]), "similar_pairs": int(stc[2]),
        "survivors": int(mc.sum())}
    if refine is not None:
        r_ms, n_cand = refine
        bpa = 2 * n_atoms * 24 + 16
        gbs = n_cand * bpa / (r_ms * 1e-3) / 1e9
        sec["refine"] = {"kernel": "k_bucket_count + k_bucket_scan + k_bucket_scatter + k_refine_buckets (the whole refine of a long queue)",
                         "kernel_ms": r_ms, "candidates": int(n_cand),
                         "alignments_per_s": n_cand / (r_ms * 1e-3),
                         # bound: the fp64 VECTOR peak (78.6 TFLOP/s on this chip, the same figure as the matrix pipe's) with
                         # SURVEY 8d's flops of one complete alignment -- not the HBM roof: the 8d bytes of a candidate never leave
                         # the L2 since the queue is ordered by bucket (`traffic`: what crosses the fabric, 12 x fewer), and an HBM
                         # fraction computed from them (0.97 in round 4's line) says nothing
                         "roofline": {"bound": "fp64", "achieved": n_cand * (53 * n_atoms + 600) / (r_ms * 1e-3) / 1e12,
                                      "peak": PEAK_F64_MFMA, "unit": "TFLOP/s",
                                      "frac": n_cand * (53 * n_atoms + 600) / (r_ms * 1e-3) / 1e12 / PEAK_F64_MFMA,
                                      "flops_per_alignment": 53 * n_atoms + 600,
                                      "algorithmic_bytes_per_alignment": bpa, "algorithmic_GBps": gbs,
                                      "traffic": pmc_traffic(("r05_pmc_refine.json", "r04_pmc_refine.json"), n_conf, n_atoms)[0],
                                      "traffic_source": pmc_traffic(("r05_pmc_refine.json", "r04_pmc_refine.json"), n_conf, n_atoms)[1],
                                      "note": "one exact fp64 alignment (rotation, rmsd, max deviation) per queued candidate pair, a lane per "
                                              "pair, the queue ordered by (1024-row, 64-column) bucket with the column tile in LDS.  What holds "
                                              "it at this fraction is the CU's vector-memory path, not arithmetic and not HBM: every lane gathers "
                                              "its row conformer twice (2.5 KB per pair through the L1: ~14 cycles per 64-lane 8-byte load, "
                                              "tuning build -DFC_RB_TIMELINE: per item of 512 pairs 8.7 us covariance pass, 8.3 us deviation "
                                              "pass with twice the multiply-adds); DESIGN.md section 9"}}
    out["config"]["secondary"] = sec
    # (d) BASELINE's second metric as SURVEY 8d words it: host arrays in -> mask out, H2D / D2H included
    fc.pruner.prune_by_rmsd(coords[:2000], atoms, MAX_RMSD)
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        fc.pruner.prune_by_rmsd(coords, atoms, MAX_RMSD)
        ts.append(time.perf_counter() - t1)
    out["pruned_ensembles_per_s_host_in_mask_out"] = 1.0 / min(ts)
    out["host_in_mask_out_ms"] = 1e3 * min(ts)
    # ... and the same call on an ensemble that lives in page-locked host memory (firecode_amd.pinned_empty: an integrator that
    # can choose where its coordinate arrays live; the upload is then a direct DMA, the staging copy is skipped).  Reported
    # BESIDE host_in_mask_out_ms, which stays the ordinary-array call FIRECODE makes today
    try:
        cp = fc.pinned_empty(coords.shape)
        cp[...] = coords
        fc.pruner.prune_by_rmsd(cp[:2000], atoms, MAX_RMSD)
        tp = []
        for _ in range(5):
            t1 = time.perf_counter()
            _, m_p = fc.pruner.prune_by_rmsd(cp, atoms, MAX_RMSD)
            tp.append(time.perf_counter() - t1)
        out["host_in_mask_out_pinned_ms"] = 1e3 * min(tp)
        del cp
    except Exception as ex:  # (reported, never fatal: the contract line does not depend on it)
        out["host_in_mask_out_pinned_ms"] = None
        out["host_in_mask_out_pinned_error"] = str(ex)[:200]
    # where that time goes (the same three steps prune_by_rmsd performs, timed apart; min / median of 7)
    legs = {"create_ms": [], "prune_ms": [], "index_ms": []}
    for _ in range(7):
        t1 = time.perf_counter()
        e2 = fc.DeviceEnsemble(coords, center=True)
        t2 = time.perf_counter()
        m2, _ = e2.prune(MAX_RMSD, 2 * MAX_RMSD)
        t3 = time.perf_counter()
        coords[m2]
        t4 = time.perf_counter()
        e2.close()
        legs["create_ms"].append(1e3 * (t2 - t1))
        legs["prune_ms"].append(1e3 * (t3 - t2))
        legs["index_ms"].append(1e3 * (t4 - t3))
    out["host_in_mask_out_breakdown"] = {
        "what": "create = 12 MB through the library's pinned pieces (memmove 0.24 ms + DMA 0.22 ms at 54 GB/s, overlapped piece by "
                "piece: tools/attic/pin_probe.py) + the preparation kernel + one host wait; prune = one synchronous prune of the resident "
                "ensemble (operand conversion, screen, refine, ladder, mask copy, host wait: nothing overlaps in a single call); "
                "index = structures[mask] on the host.  min / median over 7 calls: the spread between runs and boxes is in the first leg "
                "(host memory bandwidth, PCIe), not in the kernels",
        **{k: {"min": min(v), "median": sorted(v)[len(v) // 2]} for k, v in legs.items()},
        "whole_call_ms": {"min": 1e3 * min(ts), "median": 1e3 * sorted(ts)[len(ts) // 2]}}
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
