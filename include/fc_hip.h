/* fc_hip.h -- C ABI of libfc_hip.so: the MI355X (gfx950) ensemble-geometry
 * hot path of FIRECODE behind plain pointers and sizes.
 *
 * The reference (ntampellini/FIRECODE v2.0.4, paths relative to
 * /root/reference) has no FFI: its hot path is NumPy/SciPy calls made from
 * Python.  Each entry point below names the reference call it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives
 * in firecode_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a HOST pointer to a caller-owned buffer unless the
 *     parameter name ends in _dev; coordinates are float64, C-order
 *     (N, A, 3) -- firecode/typing_.py:6-8; masks are 1 byte per element
 *     (np.bool_, typing_.py:12); index arrays are int64.
 *   - return value: 0 = OK, < 0 = error (FC_E_*); the message is kept per
 *     thread and read with fc_last_error().  No exceptions, no callbacks and
 *     no stdout writes cross this boundary.  Calls block until results are in
 *     the output buffers.
 *   - the HIP context is created lazily, per process, on the first call
 *     (the reference may call from spawn()ed pool workers, embedder.py:116).
 *   - threads: the library keeps ONE context (device, streams, staging buffers) per
 *     process.  Entry points may be called from several host threads; each holds the
 *     library's lock for its whole duration, so calls run one after the other (the
 *     reference's hot path is single-threaded per process).  fc_last_error() is per thread.
 *     fc_shutdown / fc_init(other device) end the context: ensembles created before are
 *     refused afterwards (FC_E_INVALID) and may only be destroyed.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point returns FC_E_NODEVICE.
 */
#ifndef FC_HIP_H
#define FC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_OK 0
#define FC_E_INVALID (-1)   /* bad argument (shape, NULL, range)              */
#define FC_E_NODEVICE (-2)  /* no HIP device / HIP runtime error at init      */
#define FC_E_HIP (-3)       /* HIP runtime error (message has the hipError)   */
#define FC_E_NOMEM (-4)     /* device allocation failed                       */
#define FC_E_LIMIT (-5)     /* size outside what a kernel supports            */
/* FC_E_LIMIT as CONTRACT -- where the reference has no size regime and this library refuses (everything else is served at
 * any size, more slowly beyond the tiled kernels: complete alignments above 104 atoms and prune screens above 192 atoms
 * leave the matrix pipes, tests/test_gpu_parity.py checks them at 105 ... 320 atoms):
 *   - per-structure clash / rotation / fitness kernels, torsion scan, embed pose grids (fc_clash_*, fc_rototranslate,
 *     fc_fitness, fc_torsion_scan*, fc_embed_*): one structure (or pose table) lives in a workgroup's 160 KB of LDS --
 *     A <= 1706 atoms per structure (4 x A x 24 B), for the pose grids the per-group bytes the message states;
 *   - TFD (fc_tfd_simbits, fc_tfd_first_match, fc_torsion_scan_tfd*): Q <= 128 fingerprint entries -- np.sum switches to
 *     pairwise summation above 128 addends (firecode/torsion_module.py:1064) and the kernels reproduce the plain order;
 *   - cyclical embed: at most 85 distinct step angles per molecule and 2^31 poses per call (split the jobs);
 *   - A <= 32767 atoms per conformer, bit matrices below 2^32 words, and the queue capacities stated with the entry
 *     points that have one (they name the entry point to use instead).
 * The TFD ladder has no size limit of its own: what exceeds a device capacity (components above 4096 nodes, last chunks
 * that hold edges) is finished on the host from the device's flags (fc_tfd_ladder_from_first_match). */

/* ---- lifecycle --------------------------------------------------------- */
int fc_abi_version(void);
int fc_device_count(void);
int fc_init(int device);
int fc_shutdown(void);
const char *fc_last_error(void);
/* Optional: pays the one-time costs of a process now instead of inside its first calls -- the HIP runtime loads a
 * translation unit's device code at that unit's first launch (a few ms each, ~0.1 s over the library) and the first
 * large call takes its device buffers from the runtime one by one.  For long-running callers and for timing runs;
 * fc_init stays lazy and cheap (FIRECODE calls from short-lived pool workers, embedder.py:116-120). */
int fc_warmup(void);
/* Enqueue everything on the caller's HIP stream (a hipStream_t, e.g. the stream a
 * collective library orders itself against) instead of the library's own non-blocking
 * stream; NULL switches back.  The previous stream is drained first.  The legacy null
 * stream cannot be named this way (NULL means "own stream"). */
int fc_stream_set(void *hip_stream);
/* The same switch WITHOUT draining the previous stream: for a caller that orders its streams
 * with events itself (the overlapped multi-GPU steps of firecode_amd/dist.py). */
int fc_stream_use(void *hip_stream);
/* Device temporaries of the entry points come from a caching pool (released blocks are kept
 * and reused; FC_POOL_MB caps what is kept, default 8192, 0 disables).  fc_memory_trim returns
 * the kept blocks to the HIP runtime; fc_shutdown does the same. */
int fc_memory_trim(void);
/* Page-locked host memory for a caller that can choose where its coordinate arrays live (FIRECODE's Ensemble holds one
 * (N, A, 3) float64 array per ensemble: firecode/ensemble.py:58-98).  An upload from such memory is a direct DMA -- the
 * library's staging copy for pageable arrays (0.24 ms per 12 MB) is skipped, nothing else changes:
 * every entry point takes either kind of pointer.  fc_host_free_pinned releases a block; blocks still alive at
 * fc_shutdown stay valid host memory and must still be freed by their owner. */
int fc_host_alloc_pinned(int64_t bytes, void **out);
int fc_host_free_pinned(void *p);
/* name, CU count and bytes of HBM of the active device (diagnostics) */
int fc_device_info(char *name, int64_t name_len, int64_t *n_cu, int64_t *hbm_bytes);

/* ---- resident ensemble -------------------------------------------------
 * Uploads (N, A, 3) coordinates once and keeps the prepared layout in HBM:
 * the atoms selected by atom_mask (NULL = all; the "heavy atoms" of
 * prism_pruner's RMSD pruner), optionally centred on their centroid, stored
 * conformer-minor  Xs[(a*3+c)*Npad + n]  so that a wavefront reads 64
 * conformers of one coordinate with one coalesced 512-byte load, plus
 * G[n] = sum |x|^2.  Everything below that takes an fc_ensemble works on
 * HBM-resident data only. */
typedef struct fc_ensemble fc_ensemble;
int fc_ensemble_create(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask,
                       int center, fc_ensemble **out);
int fc_ensemble_destroy(fc_ensemble *ens);
int fc_ensemble_shape(const fc_ensemble *ens, int64_t *N, int64_t *A_selected);

/* ---- a4: rmsd_and_max(p, q, center) -- prism_pruner.rmsd; call sites
 * firecode/utils.py:499, embedder.py:1784, ase_manipulations.py:1384.
 * P pairs (pair_i[k], pair_j[k]) of one coordinate block. */
int fc_kabsch_rmsd_pairs(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask,
                         const int64_t *pair_i, const int64_t *pair_j, int64_t P, int center,
                         double *rmsd_out, double *maxdev_out);
/* same over a resident ensemble */
int fc_ensemble_rmsd_pairs(fc_ensemble *ens, const int64_t *pair_i, const int64_t *pair_j,
                           int64_t P, double *rmsd_out, double *maxdev_out);
/* all pairs: rmsd_out / maxdev_out are (N, N) row-major, symmetric, 0 diagonal */
int fc_ensemble_rmsd_matrix(fc_ensemble *ens, double *rmsd_out, double *maxdev_out);
/* The complete alignment of ALL pairs, timed on the device: rmsd_and_max's two outputs (RMSD and
 * max per-atom deviation; firecode/utils.py:499) for every i < j -- covariance tiles on the fp64
 * matrix pipe, rotation (Newton eigenvalue + adjugate column; Jacobi sweeps where the eigenvalue is
 * not clearly simple) and one atom pass in the epilogue.  The max deviation is taken from the explicit
 * rotated difference p - R q; the rmsd from the largest eigenvalue of the pair's quaternion matrix,
 * sqrt(((Gp + Gq) - 2 lambda) / A) -- the sum of squares of that same difference under the optimal
 * rotation, equal to it within 1e-11 on everything tested (tests/test_complete_eig_bound.py pins the
 * bound) -- except for pairs closer than ~1e-3 A, where that difference of large numbers no longer
 * holds 1e-10: those, like pairs with a declined rotation, are recomputed by a fix-up kernel from the
 * explicit sum.  FC_COMPLETE_EIG=0: the explicit running sum for every pair (the form of rounds 2-4).
 * rmsd_out / maxdev_out (N, N) symmetric with 0 diagonal, either may be NULL (timing only);
 * ms_kernel (may be NULL) = HIP-event time of the kernels.  fc_ensemble_rmsd_matrix is this
 * call with both outputs required. */
int fc_ensemble_rmsd_and_max_all(fc_ensemble *ens, double *rmsd_out, double *maxdev_out, double *ms_kernel);
/* all-pairs RMSD VALUES on the fp64 matrix pipe: covariance by MFMA, largest
 * quaternion eigenvalue by Newton (QCP), rmsd = sqrt((Gp+Gq-2*lambda)/A); pairs
 * below 0.02 A are re-evaluated with the explicit rotated difference.  No max
 * deviation (that needs the rotation: fc_ensemble_rmsd_matrix).  rmsd_out (N, N)
 * symmetric, may be NULL (timing only); ms_kernel (may be NULL) = HIP-event time
 * of the two kernels. */
int fc_ensemble_rmsd_values(fc_ensemble *ens, double *rmsd_out, double *ms_kernel);
/* a9: get_alignment_matrix(p, q) -- prism_pruner.rmsd; call site
 * hypermolecule_class.py:77.  M (3,3) row-major, applied as (M @ q.T).T */
int fc_alignment_matrices(const double *p, const double *q, int64_t n_pairs, int64_t A,
                          double *M_out);

/* ---- a5: prune_by_rmsd(structures, atoms, max_rmsd, energies=, max_dE=)
 * -- prism_pruner.pruner; call sites firecode/ensemble.py:230-235,
 * embedder.py:1472-1474, operators.py:619-624.
 *
 * fc_rmsd_simbits: similarity bits of rows [row_begin,row_end) against all
 * later conformers: bit j of row i (j > i) = rmsd(i,j) < max_rmsd &&
 * maxdev(i,j) < max_dev [&& |E_i-E_j| < max_dE when energies != NULL].
 * bits_out: (row_end-row_begin) rows of W = ceil(N/64) uint64 words.
 * n_grey (may be NULL): pairs within 1e-9 of either threshold.
 *
 * fc_prune_rmsd: the whole stage on the GPU -- similarity bits, then the
 * k-ladder greedy replay -- conformers taken in the order given (the host
 * pre-sorts by energy as the reference does).  mask_out: N bytes.
 * stats (may be NULL): [0]=pairs evaluated, [1]=candidate pairs refined,
 * [2]=similar pairs, [3]=grey pairs, [4]=ladder levels run, [5]=survivors. */
int fc_rmsd_simbits(fc_ensemble *ens, double max_rmsd, double max_dev, const double *energies,
                    double max_dE, int64_t row_begin, int64_t row_end, uint64_t *bits_out,
                    int64_t *n_grey);
int fc_prune_rmsd(fc_ensemble *ens, double max_rmsd, double max_dev, const double *energies,
                  double max_dE, int64_t min_per_group, uint8_t *mask_out, int64_t *stats);
/* prism_pruner.pruner.prune_by_rmsd as FIRECODE calls it (firecode/ensemble.py:230-235, firecode/embedder.py:1472-1474:
 * host arrays in, mask out) in ONE call: coords (N, A, 3), atom_mask (A bytes or NULL = all atoms), center as in
 * fc_ensemble_create; the other arguments, mask_out (N bytes) and stats (6 values or NULL) as in fc_prune_rmsd.  Same
 * mask as fc_ensemble_create + fc_prune_rmsd + fc_ensemble_destroy; nothing stays resident. */
int fc_prune_rmsd_host(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask, int center, double max_rmsd,
                       double max_dev, const double *energies, double max_dE, int64_t min_per_group, uint8_t *mask_out,
                       int64_t *stats);
/* One of the conventions of prism_pruner's pruner that the reference tree does not show (SURVEY.md
 * Appendix A) as a switch: 0 (default) = inside a chunk a structure is removed at the first LATER
 * similar one; 1 = the mirror rule (a structure falls to any earlier similar one of its chunk).
 * The other conventions need no kernel support: "<" against "<=" is one ulp on the threshold
 * (firecode_amd.pruner.CONVENTIONS).  Process-wide; 1 is served by the pair ladder only
 * (FC_E_INVALID from a prune that needs the bit-matrix levels). */
int fc_prune_conventions(int drop_later);
/* greedy k-ladder replay over caller-supplied bits (N rows x ceil(N/64) words) */
int fc_greedy_prune_from_bits(const uint64_t *bits, int64_t N, int64_t min_per_group,
                              uint8_t *mask_out);

/* sharded form (conformer rows dealt block-cyclically to ranks):
 * fc_prune_rmsd_begin computes this rank's rows of the bit matrix;
 * fc_prune_level applies one ladder level to this rank's rows given the
 * GLOBAL mask of the previous level (mask_in, N bytes) and returns the rows'
 * new flags in mask_out (N bytes, only owned rows written, others copied
 * from mask_in) -- the caller all-gathers between levels. */
/* stats of fc_prune_rmsd_begin: [0] pairs owned by this rank, [1] refined, [2] similar,
 * [3] grey, [4] duration of this rank's screen kernel in ns (HIP events), [5] 0 */
int fc_prune_rmsd_begin(fc_ensemble *ens, double max_rmsd, double max_dev,
                        const double *energies, double max_dE, int64_t rank, int64_t world,
                        int64_t row_block, int64_t *stats);
int fc_prune_level(fc_ensemble *ens, int64_t k, const uint8_t *mask_in, uint8_t *mask_out);
/* Preferred exchange when similar pairs are sparse (the usual case): after
 * fc_prune_rmsd_begin, fc_prune_similar_pairs returns this rank's exactly-
 * similar pairs as (i << 32) | j in processing order indices (n_out = count;
 * FC_E_LIMIT when the candidate queue overflowed -- use fc_prune_level then).
 * The ranks all-gather the lists ONCE and every rank replays the whole ladder
 * locally with fc_prune_from_pairs (pairs from all ranks, any order). */
int fc_prune_similar_pairs(fc_ensemble *ens, uint64_t *pairs_out, int64_t capacity, int64_t *n_out);
int fc_prune_from_pairs(fc_ensemble *ens, const uint64_t *pairs, int64_t n_pairs,
                        int64_t min_per_group, uint8_t *mask_out);
/* The same exchange without leaving the device (multi-GPU path of bench.py):
 *   fc_prune_rmsd_begin_async   enqueue screen + refine of this rank's row blocks, no sync;
 *   fc_prune_export_pairs_dev   write this rank's message into a CALLER-OWNED DEVICE buffer of
 *                               cap+1 words: [count | pairs ... | padding]; count is
 *                               0xFFFFFFFFFFFFFFFE when the candidate queue overflowed;
 *   (caller: one all-gather of the cap+1 words -- RCCL on the stream given to fc_stream_set)
 *   fc_prune_from_gathered_dev  compact the world*(cap+1) gathered words (device pointer), replay
 *                               the whole ladder, ONE sync, mask to the host.  FC_E_LIMIT when a
 *                               rank's list was missing or longer than cap: every rank sees the same
 *                               words, so every rank falls back to the host exchange above together.
 * stats (6): [0] pairs owned, [1] refined, [2] similar (this rank), [3] grey, [4] screen ns,
 * [5] survivors. */
int fc_prune_rmsd_begin_async(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t rank,
                              int64_t world, int64_t row_block);
int fc_prune_export_pairs_dev(fc_ensemble *ens, uint64_t *dev_out, int64_t cap);
int fc_prune_from_gathered_dev(fc_ensemble *ens, const uint64_t *dev_gathered, int64_t world,
                               int64_t cap, int64_t min_per_group, uint8_t *mask_out, int64_t *stats);
/* Stream-ordered form for a batch of n_slots prunes: ..._enqueue(slot) returns without waiting
 * (begin_async, export, all-gather and this call of prune k+1 may be issued while prune k still
 * runs); fc_prune_collect(slot) waits for the stream and delivers that prune's mask and stats
 * (FC_E_LIMIT: its ladder declined -- redo that prune through fc_prune_from_gathered_dev or the
 * host exchange).  Slot 0 must be enqueued first: it sizes the pinned result area of the batch. */
int fc_prune_from_gathered_dev_enqueue(fc_ensemble *ens, const uint64_t *dev_gathered, int64_t world,
                                       int64_t cap, int64_t min_per_group, int64_t slot, int64_t n_slots);
int fc_prune_collect(fc_ensemble *ens, int64_t slot, int64_t n_slots, uint8_t *mask_out, int64_t *stats);
/* Overlap of consecutive sharded prunes of ONE resident ensemble:
 *   fc_ensemble_twin                  a second prune workspace (bit rows, queues, counters, ladder
 *                                     words) over the same coordinates; owned by `ens`, destroyed
 *                                     with it; takes every prune call an ensemble takes;
 *   fc_prune_rmsd_begin_split_async   begin_async with the screen on `screen_stream` (all screens of a
 *                                     batch go there, in order) and counters reset + refine on the
 *                                     current stream (fc_stream_use), tied together with events;
 *                                     timed != 0: HIP events around the screen kernel (what stats[4]
 *                                     of fc_prune_collect reports; ~14 us of stream time).
 * Prune k on workspace k&1 and stream k&1 of two: export, all-gather and ladder of prune k run
 * beside the screen of prune k+1. */
int fc_ensemble_twin(fc_ensemble *ens, fc_ensemble **twin_out);
int fc_prune_rmsd_begin_split_async(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t rank,
                                    int64_t world, int64_t row_block, void *screen_stream, int timed);

/* ---- the exchange of the sharded path: RCCL over xGMI behind the C ABI (no PyTorch) ---------
 * "ensembles shard by conformer across the GPUs of one node with a single RCCL
 * all-gather to reassemble the surviving-conformer mask".  One process per GPU, started by any
 * launcher; librccl.so is bound at run time (dlopen) by the first of these calls, so single-GPU
 * users never load it.  Rank 0 calls fc_comm_unique_id and passes the 128 bytes to the other
 * ranks out of band (firecode_amd/dist.py: a file next to MASTER_PORT, or FC_COMM_ID);
 * every rank then calls fc_comm_init (collective; after fc_init(device of this rank)).
 * Without a communicator the calls below behave as a group of one rank.
 *   fc_allgather_mask     host buffers, blocking: mask_global = world x n_local bytes, rank-major
 *                         (the per-level mask exchange and the pass masks of the pose grid / scan);
 *   fc_allgather_u8_dev   device buffers, enqueued behind the library's current stream;
 *   fc_comm_barrier       all ranks have arrived (a 1-byte all-gather + wait). */
#define FC_COMM_ID_BYTES 128
int fc_comm_unique_id(uint8_t *id_out);
int fc_comm_init(int64_t rank, int64_t world, const uint8_t *id);
int fc_comm_destroy(void);
int fc_comm_info(int64_t *rank, int64_t *world);
int fc_allgather_mask(const uint8_t *mask_local, int64_t n_local, uint8_t *mask_global);
int fc_allgather_u8_dev(const uint8_t *send_dev, uint8_t *recv_dev, int64_t bytes_per_rank);
int fc_comm_barrier(void);
/* test hook: act as `rank` of `world` without a communicator -- the all-gather writes only this
 * rank's slot and leaves the others as earlier calls left them, so the logical ranks of one GPU can
 * be run one after the other (tests/test_gpu_parity.py); world = 0 switches it off */
int fc_debug_comm_loopback(int64_t rank, int64_t world);
/* prune_by_rmsd over the ranks of the communicator: every rank holds the whole ensemble and
 * calls this with the same arguments; rows of the similarity matrix are dealt in snake order
 * (row_block rows at a time, <= 0: default), each rank screens + refines its own, ONE all-gather of
 * the ranks' exactly-similar pair lists (cap = 1024 + 4N/world pairs each), the whole k-ladder
 * replayed on every rank: identical mask_out (N bytes) everywhere.  Dense similarity (a list that
 * does not fit): one fc_allgather_mask per ladder level instead, decided alike on every rank.
 * stats (6, may be NULL): [0] pairs owned, [1] refined, [2] similar (this rank), [3] grey,
 * [4] screen kernel ns, [5] survivors. */
int fc_prune_rmsd_sharded(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t min_per_group,
                          int64_t row_block, uint8_t *mask_out, int64_t *stats);

/* ---- a7: prune_by_rmsd_rot_corr(structures, atoms, graph, max_rmsd=, energies=, max_dE=) --
 * prism_pruner.pruner (NOT in the reference tree); call sites firecode/ensemble.py:253-260,
 * embedder.py:1489-1496, operators.py:626-632.  PARITY UNPINNED: restated after the predecessor's
 * published routine (see the kernel's header).  The caller does the graph perception and passes
 * the locally symmetric torsions (T,4), the rotation mask of each (T,A: atoms that move with i4,
 * firecode/torsion_module.py:354-382), and per torsion its n-fold trial angles in degrees
 * (angles (T,max_angles), n_angles (T); 0 first, as Torsion.get_angles gives them).
 * Structures are centred on their plain mean; similarity = rmsd_and_max over heavy_mask atoms
 * after the torsional correction; then the same k-ladder as fc_prune_rmsd.  coords in processing
 * order (energy-sorted by the caller when energies are given).  bits_out: optional (N, W). */
int fc_prune_rmsd_rot_corr(const double *coords, int64_t N, int64_t A, const uint8_t *heavy_mask,
                           const int64_t *torsions, int64_t T, const uint8_t *rotation_masks,
                           const double *angles, const int32_t *n_angles, int64_t max_angles,
                           double max_rmsd, double max_dev, const double *energies, double max_dE,
                           int64_t min_per_group, uint8_t *mask_out, uint64_t *bits_out);

/* ---- a9: align_by_moi(atoms, structures) -- firecode/hypermolecule_class.py:45-86.
 * Every structure centred on its plain mean; M = get_alignment_matrix(diag(I_ref), diag(I_n))
 * applied as (M @ X.T).T; structure 0 is copied unrotated.  masses (A,), out (N, A, 3). */
int fc_align_by_moi(const double *coords, int64_t N, int64_t A, const double *masses, double *out);

/* ---- a6: prune_by_moment_of_inertia -- prism_pruner.pruner; call sites
 * firecode/ensemble.py:211-216, embedder.py:1452-1454.
 * fc_inertia_moments: get_inertia_moments(coords, masses) for N conformers,
 * moments_out (N,3) ascending. */
int fc_inertia_moments(const double *coords, int64_t N, int64_t A, const double *masses,
                       double *moments_out);
int fc_prune_moi(const double *coords, int64_t N, int64_t A, const double *masses,
                 double max_deviation, const double *energies, double max_dE,
                 int64_t min_per_group, uint8_t *mask_out);

/* ---- a2/a3: the similarity stages of the drivers on ONE upload -- Ensemble.similarity_pruning
 * (firecode/ensemble.py:205-235) and Embedder.similarity_refining (embedder.py:1445-1474) run
 * prune_by_moment_of_inertia, apply its mask, then prune_by_rmsd on the survivors.  coords (N, A, 3) in
 * processing order (energy-sorted by the caller when energies are given) are uploaded once; the MOI
 * stage (do_moi; all atoms, `masses`, relative tolerance moi_tol) works on them in HBM, its survivors
 * are gathered on the device into the RMSD stage's layout (do_rmsd; atoms of heavy_mask, NULL = all),
 * the masks are composed on the way out.  Each stage's result is the stand-alone entry point's.
 * mask_moi_out (N bytes, may be NULL): after the MOI stage; mask_out (N bytes): after both;
 * counts (3, may be NULL): structures in, after MOI, after RMSD. */
int fc_prune_similarity(const double *coords, int64_t N, int64_t A, const uint8_t *heavy_mask, const double *masses,
                        int do_moi, double moi_tol, int do_rmsd, double max_rmsd, double max_dev,
                        const double *energies, double max_dE, int64_t min_per_group, uint8_t *mask_moi_out,
                        uint8_t *mask_out, int64_t *counts);

/* ---- a8: align_structures(structures, indices) -- prism_pruner.utils;
 * call sites embedder.py:1704,1910,2218,2300; operators.py:262.
 * out (N,A,3): every conformer superposed on conformer 0 using the n_idx
 * atoms in idx (NULL = all). */
int fc_align_to_first(const double *coords, int64_t N, int64_t A, const int64_t *idx,
                      int64_t n_idx, double *out);

/* ---- a13: get_embed -- firecode/embeds.py:808-817: out[k] = (R_k @ X_k.T).T + t_k
 * for n blocks of A atoms (R (n,3,3), t (n,3)). */
int fc_rototranslate(const double *coords, int64_t n, int64_t A, const double *R,
                     const double *t, double *out);

/* ---- a11/a12: count_clashes (firecode/algebra.py:52-54) and
 * compenetration_check (firecode/utils.py:507-575), batched over N structures.
 * fc_clash_self: ordered pairs with lo < d < hi (reference: 0 < d < 0.5).
 * fc_clash_fragments: ids = fragment lengths (2 or 3 entries); bimolecular
 * counts d < thresh, trimolecular counts d <= thresh cumulatively over
 * (m2,m1),(m3,m2),(m1,m3); pass_out[n] = 1 when within max_clashes. */
int fc_clash_self(const double *coords, int64_t N, int64_t A, double lo, double hi,
                  int64_t *counts_out);
int fc_clash_fragments(const double *coords, int64_t N, int64_t A, const int64_t *ids,
                       int64_t n_ids, double thresh, int64_t max_clashes,
                       int64_t *counts_out, uint8_t *pass_out);

/* graph mode of compenetration_check (utils.py:528-542): ordered, off-diagonal,
 * NON-BONDED pairs with d < thresh; adj is an (A, A) byte adjacency matrix. */
int fc_clash_graph(const double *coords, int64_t N, int64_t A, const uint8_t *adj, double thresh,
                   int64_t *counts_out);
/* a21: fitness_check (optimization_methods.py:163-180) for N structures with C
 * constraints each: pairs (N, C, 2), targets (N, C) (NaN = no target);
 * pass_out[n] = sum(|x_a - x_b| - target) < threshold; error_out may be NULL. */
int fc_fitness_check(const double *coords, int64_t N, int64_t A, const int64_t *pairs,
                     const double *targets, int64_t C, double threshold, double *error_out,
                     uint8_t *pass_out);

/* ---- a14: bimolecular rigid embed poses -- firecode/embeds.py:713-722:
 * pose k uses conformer c1[k] of m1 (n1,A1,3) moved by (R1[k], t1[k]) and
 * conformer c2[k] of m2 (n2,A2,3) moved by (R2[k], t2[k]); counts atom pairs
 * closer than thresh (strict <) without materialising the pose;
 * pass_out[k] = count <= max_clashes.  poses_out (may be NULL): (P, A1+A2, 3). */
int fc_embed_poses_clash(const double *m1, int64_t n1, int64_t A1, const double *m2,
                         int64_t n2, int64_t A2, const int64_t *c1, const int64_t *c2,
                         const double *R1, const double *t1, const double *R2,
                         const double *t2, int64_t P, double thresh, int64_t max_clashes,
                         int64_t *counts_out, uint8_t *pass_out, double *poses_out);

/* Whole bimolecular rigid-embed grid (embeds.py:597-722 for one pivot per
 * conformer): molecule i has n_i conformers of A_i atoms, nr_i (1 or 2)
 * reactive atom indices, one pivot (start, end 3-vectors) per conformer and
 * na_i step angles (degrees).  fc_embed_mol_transforms returns the per-
 * molecule tables R (n,2,na,3,3), t (n,2,na,3) over (conformer, orientation,
 * angle); fc_embed_grid_clash tests every pose
 *   p = ((c2*n1 + c1)*2 + o)*(na1*na2) + (a2*na1 + a1)
 * (the reference's loop order) and writes pass_out[p] = count(d < thresh) <=
 * max_clashes; counts_out (may be NULL) saturates just above max_clashes. */
int fc_embed_mol_transforms(const double *coords, int64_t n, int64_t A, const int64_t *reactive,
                            int64_t nr, const double *pivot_start, const double *pivot_end,
                            int64_t mol, const double *angles, int64_t na, double *R_out,
                            double *t_out);
int fc_embed_grid_clash(const double *m1, int64_t n1, int64_t A1, const int64_t *reactive1,
                        int64_t nr1, const double *ps1, const double *pe1, const double *m2,
                        int64_t n2, int64_t A2, const int64_t *reactive2, int64_t nr2,
                        const double *ps2, const double *pe2, const double *angles1, int64_t na1,
                        const double *angles2, int64_t na2, double thresh, int64_t max_clashes,
                        uint8_t *pass_out, int32_t *counts_out, double *ms_kernel);
/* same grid followed by the in-group de-duplication of embeds.py:723
 * (rmsd_similarity(pose, poses kept so far in this (conformer pair, orientation)
 * group, rmsd_thr)): accept_out[p] = pass[p] && the pose is new. */
int fc_embed_grid_dedupe(const double *m1, int64_t n1, int64_t A1, const int64_t *reactive1,
                         int64_t nr1, const double *ps1, const double *pe1, const double *m2,
                         int64_t n2, int64_t A2, const int64_t *reactive2, int64_t nr2,
                         const double *ps2, const double *pe2, const double *angles1, int64_t na1,
                         const double *angles2, int64_t na2, double thresh, int64_t max_clashes,
                         double rmsd_thr, uint8_t *pass_out, uint8_t *accept_out);

/* ---- a14, three molecules: the body of cyclical_embed, firecode/embeds.py:409-585.
 * A "job" is one (conformer triple, pivot triple); the host enumerates jobs in the reference's
 * loop order and supplies, per job: conf (J,3); the three pivots' start / end points
 * (J,3,3 each; Pivot.pivot = start - end); polygonize(norms) as vecs (J,8,3,2,3);
 * _get_directions(norms) as dirs0 (J,3,3); run (J,8) = the orientation passes the pairings filter
 * (:471-475); rtab (J,8,3,3): atom index of molecule m's reactive atom facing molecule k
 * (the r table of :338-352); norms (J,3).  ua (3,U): distinct step angles (degrees) per molecule,
 * aidx (S,3) int32: pose s of embedder.systematic_angles -> indices into ua.
 * Per job the 8 orientations run in order: _adjust_directions (:262-407, 343 candidates, first
 * minimum) whose result also seeds the next orientation; then the S poses: R, t per molecule
 * (:488-546), trimolecular compenetration_check (utils.py:556-575, d <= thresh, sum over the three
 * fragment pairs <= max_clashes) and the rmsd_similarity(rmsd_thr) filter against the poses
 * already accepted in the group (:553-566).
 * Out: dirs_out (J,8,3,3) directions used by each orientation; Rt_out (J,8,3,U,12): R (9, row
 * major) and t (3) of molecule i at step angle u; pass_out / accept_out (J,8,S) uint8. */
int fc_embed_trimolecular(const double *const coords[3], const int64_t n_conf[3], const int64_t n_atoms[3],
                          const int64_t *const reactive[3], const int64_t n_reactive[3], int64_t J,
                          const int64_t *conf, const double *piv_start, const double *piv_end,
                          const double *vecs, const double *dirs0, const uint8_t *run, const int64_t *rtab,
                          const double *norms, const double *ua, int64_t U, const int32_t *aidx, int64_t S,
                          double thresh, int64_t max_clashes, double rmsd_thr, double *dirs_out,
                          double *Rt_out, uint8_t *pass_out, uint8_t *accept_out);

/* String embed (firecode/embeds.py:51-158) for two molecules with one reactive atom
 * each: molecule i has n_i conformers, K_i orbital centres per conformer
 * (centers_i, orbvecs_i: (n_i, K_i, 3) -- RAtom.center / .orb_vecs) and the run
 * has nA rotation angles.  Pose p = ((c2*n1 + c1)*(K1*K2) + (k2*K1 + k1))*nA + ia
 * (the reference's loop order).  pass_out[p]: compenetration_check of the pose;
 * accept_out[p]: passed AND its torsion fingerprint over quads (Q,4; atom indices
 * in the concatenated pose) is not TFD-similar (tfd_thresh) to any pose accepted
 * before it.  R2_out (P,3,3) / t2_out (P,3) (may be NULL): molecule 2's transform
 * (molecule 1 is not moved). */
int fc_string_embed(const double *m1, int64_t n1, int64_t A1, const double *centers1,
                    const double *orbvecs1, int64_t K1, const double *m2, int64_t n2, int64_t A2,
                    const double *centers2, const double *orbvecs2, int64_t K2,
                    const double *angles, int64_t nA, const int64_t *quads, int64_t Q,
                    double thresh, int64_t max_clashes, double tfd_thresh, uint8_t *pass_out,
                    uint8_t *accept_out, double *R2_out, double *t2_out);

/* ---- a17-a19: torsion scan -- firecode/torsion_module.py:812-856
 * (clustered_csearch inner loops) with rotate_dihedral (prism_pruner.utils)
 * and torsion_comp_check (torsion_module.py:894-918).
 * base (A,3); torsions (T,4) int64; rotmasks (T,A) bytes; angles (S,T) int64
 * degrees.  coords_out (S,A,3); rotated_bonds_out (S) int64. */
int fc_torsion_scan(const double *base, int64_t A, const int64_t *torsions, int64_t T,
                    const uint8_t *rotmasks, const int64_t *angles, int64_t S, double thresh,
                    int64_t backoff_deg, double *coords_out, int64_t *rotated_bonds_out);
/* The same scan with the torsion fingerprint of every generated conformer taken while it is
 * still in LDS (get_torsion_fingerprint, torsion_module.py:1070-1077): tf_out (S, Q) degrees for
 * quads (Q,4).  coords_out may be NULL -- for 1.7 M angle-sets the conformers are 2 GB of which
 * prune_conformers_tfd keeps a few thousand: fingerprint all, prune, re-scan only the survivors. */
int fc_torsion_scan_fingerprints(const double *base, int64_t A, const int64_t *torsions, int64_t T,
                                 const uint8_t *rotmasks, const int64_t *angles, int64_t S, double thresh,
                                 int64_t backoff_deg, const int64_t *quads, int64_t Q, double *tf_out,
                                 int64_t *rotated_bonds_out, double *coords_out);
/* fc_torsion_scan_fingerprints followed by the TFD prune of `[starting structure] + [scanned conformers
 * with at least one rotated bond]`, the list clustered_csearch hands to prune_conformers_tfd
 * (firecode/torsion_module.py:858-870, 957-1043) -- with the fingerprints never leaving the device
 * (at 1.7 M angle-sets they are 107 MB each way).  keep_out: S + 1 bytes, [0] = the starting structure,
 * [1 + s] = 1 when conformer s rotated a bond AND survived the prune.  Same mask as fc_tfd_prune on the
 * same rows. */
int fc_torsion_scan_tfd(const double *base, int64_t A, const int64_t *torsions, int64_t T, const uint8_t *rotmasks,
                        const int64_t *angles, int64_t S, double thresh, int64_t backoff_deg, const int64_t *quads,
                        int64_t Q, double tfd_thresh, int64_t *rotated_bonds_out, uint8_t *keep_out);
/* fc_torsion_scan_tfd over the whole n-fold grid of clustered_csearch (firecode/torsion_module.py:822: `for angles in
 * cartesian_product(*rotations)`): values = the T value lists one after the other (counts[t] entries each, degrees), the
 * S = prod counts angle-sets are the rows of the reference's cartesian_product (firecode/utils.py:219-221, array #2
 * slowest, then #1, #3 ... #T) and are generated on the device -- at 8 x 6-fold the grid is 107 MB that otherwise is built
 * on the host and sent.  rotated_bonds_out: S, keep_out: S + 1, row numbers = row numbers of that product. */
int fc_torsion_scan_tfd_grid(const double *base, int64_t A, const int64_t *torsions, int64_t T, const uint8_t *rotmasks,
                             const int64_t *values, const int64_t *counts, double thresh, int64_t backoff_deg,
                             const int64_t *quads, int64_t Q, double tfd_thresh, int64_t *rotated_bonds_out, uint8_t *keep_out);

/* ---- a20: torsion fingerprints and TFD similarity bits --
 * firecode/torsion_module.py:1046-1076.
 * tf_out (N,Q) degrees in (-180,180].  fc_tfd_simbits: bit j of row i (all
 * j != i) = sum |wrap(tf_i - tf_j)| < thresh; rows [row_begin,row_end). */
int fc_torsion_fingerprint(const double *coords, int64_t N, int64_t A, const int64_t *quads,
                           int64_t Q, double *tf_out);
int fc_tfd_simbits(const double *tf, int64_t N, int64_t Q, double thresh, int64_t row_begin,
                   int64_t row_end, uint64_t *bits_out);

/* prune_conformers_tfd at any N (firecode/torsion_module.py:957-1043):
 * fc_tfd_first_match: first_out[i] = min{ j > i : TFD-similar(i, j) } or -1 (GPU).  From 65 536 structures on
 * (and angles within +-270 degrees, all finite: the reference's delta is the circular distance exactly there) the
 * pair tests run on 16-bit angles with the fp64 sum in NumPy's order for every pair within 0.055 degrees of the
 * threshold -- same array as the fp32-filtered kernels, which serve every other input (FC_TFD_U16=0: always);
 * fc_tfd_ladder_from_first_match: the reference's k-ladder / match-graph /
 * "keep group[0]" bookkeeping replayed from that array -- bit-identical to the
 * reference under CPython >= 3.8 + networkx 3.x because it reproduces their set
 * iteration order.  With an initialised device and N >= 20000 the ladder runs
 * there (fc_tfd_ladder.hip; FC_TFD_GPU=0: on the host anyway), otherwise it is
 * pure host code (no device needed);
 * fc_tfd_prune = both. */
int fc_tfd_first_match(const double *tf, int64_t N, int64_t Q, double thresh, int64_t *first_out);
int fc_tfd_ladder_from_first_match(const int64_t *first_match, int64_t N, uint8_t *mask_out);
int fc_tfd_prune(const double *tf, int64_t N, int64_t Q, double thresh, uint8_t *mask_out);
/* test hooks for the CPython set-order emulation used by the ladder: iteration
 * order of set(keys) (non-negative ints inserted in the given order) and of a
 * set of 2-tuples (returned as indices into the input) */
int fc_debug_pyset_order_ints(const int64_t *keys, int64_t n, int64_t *order_out, int64_t *n_out);
int fc_debug_pyset_order_pairs(const int64_t *pairs, int64_t n, int64_t *order_out, int64_t *n_out);
/* the same order for n DISTINCT pairs computed on the device (fc_tfd_ladder.hip: staged priority first-fit, what the
 * coarse levels of the TFD ladder use); order_out: n indices.  Test hook. */
int fc_debug_pyset_order_pairs_device(const int64_t *pairs, int64_t n, int64_t *order_out);
/* the device ladder's per-chunk and per-component routines (fc_tfd_core.h) run on the CPU by one host thread standing
 * for a wavefront: same mask as fc_tfd_ladder_from_first_match.  Pure host code.  Test hook. */
int fc_debug_tfd_ladder_emulate(const int64_t *first_match, int64_t N, uint8_t *mask_out);

/* ---- a1: the .xyz wire format (host code; firecode/ensemble.py:58-98, 284-297;
 * firecode/utils.py:105-116).  atoms: A C strings.  mode 0 = Ensemble.to_xyz text
 * (label = basename), mode 1 = utils.write_xyz text repeated per conformer
 * (label = title).  Byte-identical to the Python writers.
 * fc_xyz_scan: conformer count and atoms per conformer of a file;
 * fc_xyz_read: atoms_out A x 8 bytes (NUL padded symbols of the first conformer),
 * coords_out (N, A, 3) parsed exactly like Python's float(). */
int fc_xyz_write(const char *path, const char *const *atoms, int64_t A, const double *coords,
                 int64_t N, const char *label, int mode);
int fc_xyz_scan(const char *path, int64_t *N_out, int64_t *A_out);
/* ---- cartesian_product (firecode/utils.py:219-221; callers torsion_module.py:484, 822) -------------------
 * Rows of np.stack(np.meshgrid(*arrays), -1).reshape(-1, T): `values` = the T arrays one after the other
 * (counts[t] entries each), out = (prod counts) x T, array #2 varying slowest, then #1, then #3 ... #T.
 * Host code (threads), no device needed: the angle grid of a conformational search (1 679 616 x 8 at cfg3)
 * is an INPUT of the scan kernels and costs NumPy 1 s.  FC_E_INVALID: T < 1, negative length. */
int fc_cartesian_product_i64(const int64_t *values, const int64_t *counts, int64_t T, int64_t *out);
int fc_cartesian_product_f64(const double *values, const int64_t *counts, int64_t T, double *out);
int fc_xyz_read(const char *path, int64_t N, int64_t A, char *atoms_out, double *coords_out);

/* ---- many ensembles in flight -------------------------------------------
 * fc_prune_rmsd_many: fc_prune_rmsd (no energy window) for `n` distinct resident ensembles,
 * enqueued together and waited for once.  What a caller with a queue of ensembles gains:
 * no host round trip between two prunes, and the small kernels of one prune (refine, level
 * buckets, ladder, result copy) run beside the all-pairs screen of the next one on other
 * streams (FC_PRUNE_LANES=1: strictly one after another).  Results are those of n calls of
 * fc_prune_rmsd: the reference has no batched form, a maintainer's loop over
 * prune_by_rmsd (firecode/ensemble.py:230-235 in a loop over ensembles) maps to one call.
 * mask_out[r]: N_r bytes; survivors_out (may be NULL): n counts.
 * FC_E_INVALID: NULL or repeated ensemble, n outside 0..4096, thresholds <= 0. */
int fc_prune_rmsd_many(fc_ensemble *const *ens, int64_t n, double max_rmsd, double max_dev,
                       int64_t min_per_group, uint8_t *const *mask_out, int64_t *survivors_out);

/* ---- bench / profiling hooks (resident data, device-side timing) --------
 * Runs the all-pairs similarity stage + greedy replay `reps` times on the
 * resident ensemble and returns HIP-event times (ms, per rep) of the
 * dominant kernel and of the whole step measured on the library's stream.  The kernel time is
 * the mean over every 8th prune (FC_BENCH_EVENT_STRIDE): the event pair costs the stream ~14 us. */
/* Arithmetic of the all-pairs screen the last prune launched: 16 = single-precision covariance on the
 * half-precision matrix pipe (coordinates split into two halfs, three f16 products) + bounded fp32
 * polynomial (default where its undecidable band is narrow and the structure has at most 128 atoms),
 * 32 = the same on the fp32 matrix pipe (FC_SCREEN_H2=0, more than 128 atoms), 64 = fp64 matrix pipe
 * (FC_SCREEN_F32=0, large structures with tight thresholds), 1 = VALU kernel, 0 = none yet.  Results do
 * not depend on it: every pair a screen lets through is decided by the exact fp64 refine. */
int fc_screen_last_kind(void);
/* Choice of the all-pairs screen for the prunes that follow: 0 = automatic (default), 16 / 32 = that
 * single-precision screen whatever its band (16: FC_E_INVALID from the prune when it does not apply),
 * 64 = the fp64 screen (the reference's arithmetic in every kernel of the step).  Process-wide;
 * results never depend on it. */
int fc_screen_select(int kind);
/* Checks of what the split-half screen (kind 16) assumes, for the tests -- no FIRECODE call maps to them.
 * fc_debug_mfma_f16_model: runs the model check of v_mfma_f32_16x16x32_f16 the library runs itself before it
 * uses that screen: flags_out[0..7] = 1 where the pattern behaves as assumed (subnormal inputs honoured, one
 * rounding to nearest per instruction, the largest addend's last place kept for the others, exact large
 * products); worst_out = largest |D - exact| / (2^-24 (|C| + sum |a b|)) over `trials` random and adversarial
 * 16 x 16 x 32 products -- the error bounds charge 66 per instruction (an order-independent bound of a 33-addend
 * fp32 sum); the library refuses the screen on a device where this probe exceeds 18.
 * fc_debug_h2_covariance: the 16 x 16 tile of covariances (rows ib.., columns jb.., multiples of 16) exactly
 * as that screen accumulates them: B_out[(row*16 + col)*9 + 3x + y] = scale^2 sum_a p_ax q_ay; entry_bound_out
 * = the bound on |B_out / scale^2 - B| / s the screen's polynomial bounds start from (s = (Gp + Gq)/2).
 * scale_out = 0 (B_out untouched): the screen does not apply to this ensemble. */
int fc_debug_mfma_f16_model(int64_t trials, int64_t *flags_out, double *worst_out);
int fc_debug_h2_covariance(fc_ensemble *ens, int64_t ib, int64_t jb, float *B_out, double *scale_out,
                           double *entry_bound_out);
int fc_bench_prune_rmsd(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps,
                        double *ms_simbits_kernel, double *ms_step, uint8_t *mask_out,
                        int64_t *stats);
/* The exact refine of the prune alone: one all-pairs screen fills the candidate-pair queue of the resident
 * ensemble, then `reps` launches of the refine over that queue (k_refine_pairs: one exact fp64 alignment --
 * covariance, rotation, rmsd, max deviation, decision -- per queued pair), HIP events around each.
 * ms_refine: mean; n_candidates: pairs per launch.  FC_E_LIMIT when the pair queue overflowed. */
int fc_bench_refine(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps, double *ms_refine,
                    int64_t *n_candidates);
/* `reps` complete all-pairs alignment passes (fc_ensemble_rmsd_and_max_all: the a4 contract of
 * firecode/utils.py:499 for every pair, fp64) over the resident ensemble, enqueued back to back, the two
 * dense (N, N) outputs staying in HBM; one host wait.  ms_kernel_mean: HIP events on the kernel's stream
 * around the tiled kernel of every pass; ms_total: first launch to the end of the last pass (fix-up
 * kernels included).  stats[3]: pairs this rank computed per pass, pairs the last pass sent to the Jacobi
 * fix-up, 1 when the tiled kernel ran (0: the one-wave-per-row kernel for structures beyond its LDS tile).
 * Under a communicator (fc_comm_init) a rank computes the rows of the row blocks (of 128) dealt to it in
 * snake order -- the units shard by row with no exchange -- and keeps its rows of the matrices.
 * FC_E_LIMIT: more degenerate pairs than the fix-up queue holds. */
int fc_bench_rmsd_and_max_all(fc_ensemble *ens, int64_t reps, double *ms_kernel_mean, double *ms_total,
                              int64_t *stats);
/* The same passes, and behind them (inside the call's wall-clock time -- about 0.1 ms -- but outside the HIP-event kernel
 * times it reports) the elements (pair_i[p], pair_j[p]),
 * p < P, of the two output matrices as the LAST pass left them: what bench.py's `value_check` and the
 * full-size parity tests compare with the oracle's rmsd_and_max (firecode/utils.py:494-504: `(rmsd,
 * maxdev)` per pair).  Under a communicator only rows this rank owns hold values.  (i > j reads (j, i).) */
int fc_bench_rmsd_and_max_all_sampled(fc_ensemble *ens, int64_t reps, const int64_t *pair_i, const int64_t *pair_j,
                                      int64_t P, double *rmsd_out, double *maxdev_out, double *ms_kernel_mean,
                                      double *ms_total, int64_t *stats);
/* (fc_bench_prune_rmsd writes EIGHT stats: [6] = 16 x 32-pair units the subset stage of the lean fp32
 * screen queued for the full test in the last prune, [7] = 1 when its sample found similarity dense
 * and the single-stage kernel did the launch) */
/* the same for fc_prune_rmsd_sharded: `reps` sharded prunes enqueued back to back, one host wait;
 * overlap != 0: prune k on workspace / lane k&1, its refine + export + all-gather + ladder beside
 * the screen of prune k+1 */
int fc_bench_prune_rmsd_sharded(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps, int overlap,
                                double *ms_screen_kernel, double *ms_step, uint8_t *mask_out, int64_t *stats);

#ifdef __cplusplus
}
#endif
#endif /* FC_HIP_H */
