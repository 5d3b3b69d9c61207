// fc_common.h -- context, error plumbing and device buffers shared by the
// translation units of libfc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/fc_hip.h"
#include "fc_tuning.h"

namespace fc {

// ---- error state (per thread) ---------------------------------------------
std::string &last_error();
int set_error(int code, const char *fmt, ...);

#define FC_HIP_TRY(expr)                                                              \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess)                                                             \
      return ::fc::set_error(_e == hipErrorOutOfMemory ? FC_E_NOMEM : FC_E_HIP,       \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                             __FILE__, __LINE__);                                     \
  } while (0)

#define FC_TRY(expr)           \
  do {                         \
    int _rc = (expr);          \
    if (_rc != FC_OK) return _rc; \
  } while (0)

#define FC_REQUIRE(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return ::fc::set_error(FC_E_INVALID, __VA_ARGS__); \
  } while (0)

// ---- context ----------------------------------------------------------------
struct Context {
  bool ready = false;
  int device = -1;
  // what the last finished prune on this context saw (shape, candidate pairs, similar pairs): a FRESH ensemble of the same
  // atom count and a comparable size starts from it, scaled by the number of pairs, instead of from "unknown" -- a caller
  // that prunes one new ensemble after another (prune_by_rmsd on host arrays: a new fc_ensemble per call) would otherwise
  // never get the long-queue forms of the refine and the ladder.  Speed only: every form gates itself on the device.
  int64_t hint_A = -1, hint_N = 0, hint_candidates = 0, hint_similar = 0;
  hipStream_t stream = nullptr;      // the stream every launch and copy goes to
  hipStream_t own_stream = nullptr;  // the library's own; `stream` differs after fc_stream_set
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  // when set, launch_simbits_screen records it right behind its main kernel (the fp32 screen is
  // followed by a verdict kernel and a gated fp64 screen that a timing of the kernel must not include)
  hipEvent_t mark_after_screen = nullptr;
  // pipelined prunes: what follows the main screen kernel of a launch (verdict, gated fp64 screen) goes to this stream,
  // ordered behind the kernel by this event, so that the screen stream holds nothing but screens
  hipStream_t after_main_stream = nullptr;
  // pipelined prunes: a speculative screen whose verdict is "redo in fp64" is not redone in place (the gated fp64
  // launch would queue its 6 753 workgroups behind the next prune's screen: 170 us on the lane) -- the verdict empties
  // the queues and sets counters[12], the pair ladder declines, and the caller redoes that prune synchronously
  bool optimistic_screen = false;
  hipEvent_t after_main_event = nullptr;
  int n_cu = 0;
  size_t hbm = 0;
  char name[128] = {0};
  // pinned host staging for small device->host results (truly asynchronous copies)
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  int64_t *ladder_all = nullptr;  // the 18 ladder values on the device, once per context (an ensemble's list is a suffix of them)
  void *pinned_word = nullptr;  // 64 page-locked bytes: the largest G of a fresh ensemble comes back here without a host wait
  // pinned pieces + events of the staged copies (d2h_staged / h2d_staged): one set per concurrent caller (the TFD
  // ladder's helper threads copy side by side), handed out under stage_mu
  struct StageSet {
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};  // one per pinned piece
    bool busy = false;
  };
  static constexpr int kStageSets = 8;
  StageSet stage[kStageSets];
  std::mutex stage_mu;
  // side streams and events of the pipelined prunes (prune_pipeline, the sharded steps): created on
  // first use by side_streams(), destroyed by fc_shutdown and when fc_init moves to another device
  hipStream_t s_screen = nullptr, s_lane[3] = {nullptr, nullptr, nullptr}, s_comm = nullptr;
  std::vector<hipEvent_t> ev_pool;
  std::vector<hipEvent_t> ev_dep_pool;  // ordering only (hipEventDisableTiming)
  hipEvent_t ev_reset = nullptr, ev_screened = nullptr, ev_comm[2] = {nullptr, nullptr};
  // incremented by every (re)initialisation: an fc_ensemble remembers the epoch it was built in and
  // is refused afterwards (its buffers and workspaces belong to the context that is gone)
  uint64_t epoch = 0;
};
// Threading contract (include/fc_hip.h): every extern "C" entry point that touches the context holds
// this lock for its whole duration -- two host threads may call the library, their calls run one
// after the other.  Recursive: entry points are also used as building blocks of others.
std::recursive_mutex &api_mutex();
#define FC_API_LOCK std::lock_guard<std::recursive_mutex> fc_api_lock_guard(::fc::api_mutex())
int side_streams();  // creates Context::s_screen / s_lane / s_comm and the ordering events once
int pinned_reserve(size_t bytes);  // grows ctx().pinned
// blocking copies through pinned pieces of the library's own (see h2d / d2h below); ordered behind what `st` holds
int d2h_staged(void *dst, const void *src_dev, size_t n, hipStream_t st);
Context &ctx();
int ensure_init();  // lazy init on device 0 (or the one given to fc_init)

// ---- device memory: grow-only RAII buffers over a small caching pool ------------------
// hipMalloc / hipFree cost ~0.1-0.3 ms each and hipFree synchronises the device; a drop-in
// call (host arrays in, mask out) makes 5-15 temporaries.  Released blocks are therefore
// kept (size classes: powers of two up to 1 MiB, multiples of 2 MiB above) and handed out
// again.  Reuse is safe because a block's next user is ordered behind its last one: entry points
// enqueue on ONE stream (fc_stream_set drains the old one), and every multi-stream region
// (prune_pipeline, the sharded steps) starts behind an event recorded on that stream, ends with a
// wait on all its side streams, and sizes its grow-only buffers before it forks.
// FC_POOL_MB caps what is kept (default 8192 of the 288 GB, 0 = no caching); fc_memory_trim() empties it.
void *pool_take(size_t n, size_t *capacity, bool any_larger = false);  // nullptr when the device is out of memory
void pool_give(void *p, size_t capacity);
void pool_trim();

// One block that a job carves its many temporaries from (bump pointer, 256-byte pieces) instead of taking each from the
// pool: a cold pool pays a hipMalloc per buffer (0.2 - 1 ms; a TFD ladder level has ~45), a block is one.  Active for
// the calling thread between ArenaScope's constructor and destructor; DevBuf::reserve falls back to the pool when the
// block is used up.  Pieces are views (never given back one by one): the block returns to the pool with the scope,
// which the job ends only behind its last kernel.
struct DevArena {
  char *base = nullptr;
  size_t size = 0, used = 0;
  void *take(size_t n) {
    const size_t need = (n + 255) & ~(size_t)255;
    if (!base || used + need > size) return nullptr;
    void *q = base + used;
    used += need;
    return q;
  }
};
inline DevArena *&thread_arena() {
  static thread_local DevArena *a = nullptr;
  return a;
}

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  uint64_t epoch = 0;  // context epoch the block was taken in: blocks of a context that is gone are freed, not pooled
  bool owned = true;  // false: a view of another DevBuf's block (see alias())
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) {
      if (ctx().ready && epoch == ctx().epoch) pool_give(p, bytes);
      else (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  // read-only view of `o`; the caller keeps `o` alive for as long as the view is used
  void alias(const DevBuf &o) {
    release();
    p = o.p;
    bytes = o.bytes;
    epoch = o.epoch;
    owned = false;
  }
  // grow-only allocation; contents are NOT preserved.  any_larger: a transient block (ArenaScope) takes the smallest
  // cached block that fits, however large, before a new hipMalloc
  int reserve(size_t n, bool any_larger = false) {
    if (n <= bytes && p) return FC_OK;
    release();
    if (n == 0) n = 8;
    if (DevArena *a = thread_arena()) {
      if (void *q = a->take(n)) {
        p = q;
        bytes = (n + 255) & ~(size_t)255;
        epoch = ctx().epoch;
        owned = false;
        return FC_OK;
      }
    }
    size_t cap = 0;
    p = pool_take(n, &cap, any_larger);
    if (!p) return set_error(FC_E_NOMEM, "device allocation of %zu bytes failed", n);
    bytes = cap;
    epoch = ctx().epoch;
    return FC_OK;
  }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
};

struct ArenaScope {  // declare BEFORE the buffers that are to come from it (destroyed after them)
  DevBuf block;
  DevArena arena;
  DevArena *prev = nullptr;
  bool active = false;
  int begin(size_t bytes) {
    prev = thread_arena();
    thread_arena() = nullptr;  // the block itself comes from the pool
    // (a level's block is held for a few ms: in the FIRST ladder of a process the coarsest levels come first and leave
    // blocks more than twice as large as the finer levels ask for -- with the pool's usual "at most 2 x" rule every
    // level of the first search paid its own hipMalloc: 52 against 25 ms for the ladder)
    const int rc = block.reserve(bytes, true);
    if (rc != FC_OK) {
      thread_arena() = prev;
      return rc;
    }
    arena.base = static_cast<char *>(block.p), arena.size = block.bytes, arena.used = 0;
    thread_arena() = &arena;
    active = true;
    return FC_OK;
  }
  ~ArenaScope();  // (below cur_stream())
};

// The stream a helper thread enqueues on (nullptr: the context's).  Set only by code that runs several independent
// device jobs from threads of its own inside ONE API call (the TFD ladder's coarse levels): each job's stream starts
// behind an event on the context's stream, and a job synchronises its stream before it releases a pool block.
inline hipStream_t &thread_stream_override() {
  static thread_local hipStream_t s = nullptr;
  return s;
}
inline hipStream_t cur_stream() {
  hipStream_t s = thread_stream_override();
  return s ? s : ctx().stream;
}
// Jobs run side by side on streams of their own (the TFD ladder's levels), so "the next taker of this block is ordered
// behind me on the one stream" no longer holds: the block goes back to the pool only when everything the owning
// thread enqueued has finished -- also on the early-return (error) paths, where kernels may still be writing to it.
inline ArenaScope::~ArenaScope() {
  if (!active) return;
  thread_arena() = prev;
  if (ctx().ready) (void)hipStreamSynchronize(cur_stream());
}

// Host <-> device copies.  A copy of kStagedCopyMin bytes or more from or to PAGEABLE host memory goes through pinned
// pieces of the library's own (h2d_staged / d2h_staged: blocking): the runtime's path for such copies registers the
// caller's pages with the driver, and when the caller later FREES that memory (a NumPy array, a std::vector: an munmap)
// the process's queues stand still for 10-25 ms at a moment that has nothing to do with the copy (measured: every other
// cfg3 search lost 22 ms in the kernel of the re-scan behind it; tools/attic/rescan_probe.py).  Downloads showed it first, uploads
// less often (one run in three); FC_STAGED_UPLOADS=0 puts uploads back on the runtime's path (0.12 ms per 12 MB faster).
// Pinned host memory (the pipelines' result slots) and small copies stay asynchronous.
constexpr size_t kStagedCopyMin = (size_t)64 << 10;
// column tile of the long-queue refine's buckets: 2^kBucketColShift conformers (fc_kabsch.hip; ensemble_shard sizes the
// bucket tables with it)
#ifndef FC_RB_COLSHIFT
#define FC_RB_COLSHIFT 6
#endif
constexpr int kBucketColShift = FC_RB_COLSHIFT;
#ifndef FC_RB_ROWS
#define FC_RB_ROWS 1024
#endif
constexpr int kBucketRows = FC_RB_ROWS;  // rows of a bucket
bool host_memory_is_pinned(const void *p);
bool staged_uploads();  // FC_STAGED_UPLOADS (default below)
int h2d_staged(void *dst_dev, const void *src, size_t n, hipStream_t st);
inline int h2d(void *dst, const void *src, size_t n) {
  if (n == 0) return FC_OK;
  if (n >= kStagedCopyMin && staged_uploads() && !host_memory_is_pinned(src)) return h2d_staged(dst, src, n, cur_stream());
  FC_HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, cur_stream()));
  return FC_OK;
}
inline int d2h(void *dst, const void *src, size_t n) {
  if (n == 0) return FC_OK;
  if (n >= kStagedCopyMin && !host_memory_is_pinned(dst)) return d2h_staged(dst, src, n, cur_stream());
  FC_HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, cur_stream()));
  return FC_OK;
}
// (Round 4, built and taken out again: hipHostRegister -> DMA in place -> hipHostUnregister for copies of 4 MB and more.
// 12 MB arrive in 0.29 ms instead of 0.37 and prune_by_rmsd(host arrays) takes 0.68-0.71 instead of 0.79 ms -- but memory
// that has EVER been registered keeps the property that made the library copy in the first place: when the caller frees
// it with an munmap, the next kernels wait 10-25 ms.  The cfg3 search, whose survivors' coordinates (4.6 MB) are a fresh
// array per search, lost 12-25 ms in the re-scan of three searches out of ten; tools/attic/hostin_fresh_probe.py does not show
// it only because glibc stops returning a repeatedly allocated 12 MB block to the system.  tools/attic/pin_probe_fresh.py.)
inline int sync() {
  FC_HIP_TRY(hipStreamSynchronize(cur_stream()));
  return FC_OK;
}
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(FC_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return FC_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A^2 added to max_rmsd^2 in the all-pairs screens (and in the refine's own early exit)
constexpr double kScreenMargin = 1e-6;
// candidate-pair queues longer than this go to the one-lane-per-pair refine kernels (k_refine_buckets / k_refine_pairs);
// shorter ones are a latency problem and stay with the 8-lanes-per-pair walk of k_simbits_refine
constexpr unsigned long long kRefineLanesMin = 1ull << 17;

// uint64 words of fc_ensemble::counters: [0..10] queue lengths and flags, [32] / [33] mask roles and tickets of the per-level pair
// ladder, [34] / [35] passed samples and tickets of the screen verdict, [11] "the fp64 screen has to run again" (k_screen_verdict),
// [13] units queued by the subset stage of the lean fp32 screen, [15] its density verdict (1 = dense),
// [63] scratch of the screen launcher; [64 + 16 l] fill level of the pair ladder's bucket of level l -- ONE PER
// 128-BYTE LINE: atomics on words of one line serialise at the memory side (18 level counters in three lines cost
// k_pair_buckets 131 us on 7.6e5 similar pairs, one wave-wide atomic instruction per 64 pairs)
constexpr size_t kCounters = 64 + 20 * 16;
constexpr int kCntLevel = 64, kCntLevelStride = 16;

// Row blocks of the bit matrix are dealt to ranks in snake order (0..W-1,
// W-1..0, 0..W-1, ...): the work of a row block falls linearly with its index,
// so pairs of consecutive cycles carry equal work on every rank.
__host__ __device__ inline int64_t global_block(int64_t local_block, int64_t rank, int64_t world) {
  return local_block * world + ((local_block & 1) ? (world - 1 - rank) : rank);
}
inline int64_t local_block_count(int64_t n_gblocks, int64_t rank, int64_t world) {
  int64_t n = 0;
  while (global_block(n, rank, world) < n_gblocks) ++n;  // strictly increasing in n
  return n;
}

}  // namespace fc

// ---- resident ensemble -----------------------------------------------------------
struct fc_ensemble {
  uint64_t epoch = 0;          // fc::Context::epoch at creation
  int64_t N = 0, A = 0;        // conformers, selected atoms
  int64_t Npad = 0, W = 0;     // N rounded up to 64; words per bit row
  fc::DevBuf Xs;               // [(a*3+c)*Npad + n] doubles, zero padded
  fc::DevBuf Xa;               // [n][a][c] doubles: same (centred, selected) atoms, conformer-major
  fc::DevBuf G;                // [Npad] sum of squares per conformer
  fc::DevBuf Xsf;              // fp32 copy of Xs for the single-precision screen (made on first use)
  fc::DevBuf sub;              // [Npad][8] floats: stage-1 subset statistics per conformer (made with Xsf)
  fc::DevBuf unitq;            // queue of the 16 x 32 units the subset stage could not rule out
  bool xsf_valid = false;
  fc::DevBuf Xh;               // split-half copy of Xs for the f16-matrix-pipe screen (made on first use)
  bool xh_valid = false;
  double xh_scale = 0.0;       // the power of two Xh was made with
  double g_max = -1.0;         // largest G (host copy, found on first use): sizes the fp32 screen's band
  bool g_max_pending = false;  // the copy of the largest G into Context::pinned_word is still in flight (ensemble_build_finish)
  std::vector<int32_t> sel_host;  // the atom selection: source of an asynchronous upload
  // prune workspace (allocated on first use, kept for later calls)
  fc::DevBuf bits;             // rows_local * W uint64
  fc::DevBuf cand;             // rows_local * W uint32: queue of words to refine
  fc::DevBuf pairq;            // pairq_cap x uint64: queue of candidate pairs (i<<32 | j)
  int64_t pairq_cap = 0;
  fc::DevBuf simq;             // pairq_cap x uint64: exactly-similar pairs found by the refine
  // the long-queue refine orders the candidates by (128-row block, 64-column tile) bucket first (fc_kabsch.hip,
  // k_bucket_* / k_refine_buckets): control words + counts + scatter cursors | offsets | non-empty buckets | sorted queue
  fc::DevBuf bk, bk_off, bk_list, sortq;
  int64_t bk_buckets = 0;      // 0: too many buckets for the single-workgroup scan (the straight walk is used)
  // Which long-queue refine to enqueue is a HOST decision made without a host wait: the bucket path is four launches and
  // a memset that all return at once on a short queue -- 0.054 ms per step on the lanes of the pipelined prune of a
  // clustered ensemble (0.190 -> 0.244 ms) -- so it is taken only when the last prune of these coordinates whose counters
  // reached the host had a long queue.  Either choice is correct for any queue (the kernels gate themselves on the
  // device-side length); -1: nothing seen yet.
  int64_t last_candidates = -1;
  int64_t last_similar = -1;   // ... and the same for the ladder's form (one workgroup / a launch per level): similar pairs last seen
  fc::DevBuf bits_full;        // N x W uint64: whole bit matrix rebuilt from gathered pairs
  fc::DevBuf item_table;       // screen items (lb << 32 | jt) that touch the upper triangle
  int64_t item_key[4] = {-1, -1, -1, -1}, item_total = 0;
  std::vector<uint64_t> item_host;  // source of the (asynchronous) upload: lives as long as the ensemble
  fc::DevBuf gathered;         // world x cap uint64: all ranks' similar pairs, compacted (device exchange)
  fc::DevBuf msg_send, msg_recv;  // this workspace's message of the all-gather: cap + 1 / world x (cap + 1) uint64
  fc::DevBuf energies;         // N doubles (optional)
  fc::DevBuf maskA, maskB;     // N bytes each
  fc::DevBuf mbits;            // W uint64 active-flag words
  fc::DevBuf ladder;           // (levels+1) x W mask words of the fused single-GPU ladder
  fc::DevBuf ladder_k;         // the ladder values that can apply at this N (int64), for k_ladder_pairs
  fc::DevBuf levelmask;        // per similar pair: ladder levels at which it shares a chunk
  std::vector<int64_t> ladder_k_host;  // source of ladder_k's upload
  int ladder_k_n = -1;
  int64_t ladder_k_mpg = -1;
  fc::DevBuf counters;         // 8 x uint64
  // sharding of the bit matrix rows (block-cyclic)
  int64_t rank = 0, world = 1, row_block = 64;
  int64_t rows_local = 0;
  bool bits_valid = false;
  // lean prune: only the candidate / similar PAIR lists are produced (no bit matrix, no word queue);
  // set by the entry points whose consumer is the pair ladder or the exchange, which repeat the prune
  // with lean = false when the pair queue overflowed
  bool lean = false;
  // second prune workspace over the same coordinates (Xs/Xa/G are views): lets the refine and
  // ladder of one prune run beside the screen of the next one (fc_bench_prune_rmsd)
  fc_ensemble *twin = nullptr;
  fc_ensemble *head = nullptr;  // twins: the ensemble that owns the coordinates (nullptr on that ensemble itself)
  fc_ensemble() = default;
  fc_ensemble(const fc_ensemble &) = delete;
  fc_ensemble &operator=(const fc_ensemble &) = delete;
  ~fc_ensemble() { delete twin; }
};
