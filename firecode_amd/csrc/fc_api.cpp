// fc_api.cpp -- the extern "C" surface of libfc_hip.so (include/fc_hip.h):
// argument checks, host<->HBM staging, kernel sequencing.  No compute here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>

#include <map>
#include <mutex>
#include <thread>
#include "fc_common.h"
#include "fc_kabsch_math.h"

namespace fc {
struct LadderKs {
  int n;
  int64_t k[24];
};
__global__ void k_store_ladder_ks(LadderKs a, int64_t *__restrict__ out) {
  if ((int)threadIdx.x < a.n) out[threadIdx.x] = a.k[threadIdx.x];
}
}  // namespace fc

namespace fc {

// ---- launchers implemented in the .hip translation units --------------------
int launch_prep(const double *, int64_t, int64_t, const int32_t *, int64_t, int, fc_ensemble *, const int32_t *);
int launch_prep_begin(fc_ensemble *);
int launch_prep_body(const double *, int64_t, int64_t, const int32_t *, int64_t, int, fc_ensemble *, const int32_t *);
int prebuild_screen_items(fc_ensemble *);
bool prep_by_tiles(int64_t);
int launch_prep_tiles(const double *, int64_t, int64_t, const int32_t *, int64_t, int, fc_ensemble *, const int32_t *, int64_t,
                      int64_t);
int launch_pairs_exact(const fc_ensemble *, const int64_t *, const int64_t *, int64_t, double *,
                       double *, double *);
int launch_matrix_exact(const fc_ensemble *, double *, double *);
int launch_rmsd_values(fc_ensemble *, double, double *, double *, int64_t rank = 0, int64_t world = 1, bool explicit_sum = false);
int launch_mirror_upper(double *m_dev, int64_t N);
int launch_gather_matrix_pairs(const double *, const double *, int64_t, const int64_t *, const int64_t *, int64_t,
                               double *, double *);
void screen_select(int);
int launch_simbits_screen(fc_ensemble *, double);
int ensure_h2_operands(fc_ensemble *, double *);
int launch_h2_cov_tile(const fc_ensemble *, int64_t, int64_t, float *);
int h2_model_report(int64_t, int64_t *, double *);
int launch_simbits_refine(fc_ensemble *, double, double, const double *, double);
int launch_align_to_first(const double *, int64_t, int64_t, const int64_t *, int64_t, double *);
int launch_alignment_matrices(const double *, const double *, int64_t, int64_t, double *);
int launch_pack_mask(const uint8_t *, int64_t, uint64_t *, int64_t, unsigned long long *);
int launch_level(const uint64_t *, int64_t, const uint64_t *, const uint8_t *, uint8_t *, int64_t,
                 int64_t, int64_t, int64_t, int64_t, int64_t);
int launch_copy_bytes(const uint8_t *, uint8_t *, int64_t);
int launch_mask_init(uint64_t *, int64_t, int64_t, int64_t);
int launch_ladder_pairs_many(const uint64_t *, uint64_t *, const unsigned long long *, const unsigned long long *,
                             unsigned long long, unsigned long long, int64_t, int64_t, int64_t, const int64_t *, const int64_t *,
                             int, uint64_t *, uint64_t *, unsigned long long *);
int launch_ladder_pairs(const uint64_t *, uint64_t *, const unsigned long long *, const unsigned long long *,
                        unsigned long long, unsigned long long, int64_t, int64_t, int64_t, const int64_t *,
                        int, uint64_t *, unsigned long long *);
int launch_scatter_pairs(const uint64_t *, int64_t, int64_t, int64_t, uint64_t *);
int launch_rotcorr_simbits(const double *, int64_t, int64_t, const uint8_t *, const int64_t *, int64_t,
                           const uint8_t *, const double *, const int32_t *, int, double, double, const double *,
                           double, uint64_t *, int64_t);
int launch_center_structures(const double *, int64_t, int64_t, double *);
int launch_moi_diag_pairs(const double *, int64_t, double *, double *);
int launch_set_identity(double *);
size_t tri_group_lds_bytes(int64_t, int, int);
int launch_tri_embed(const double *const[3], const int64_t *const[3], const int64_t[3], const int64_t[3], int64_t,
                     const int64_t *, const double *, const double *, const double *, const double *,
                     const uint8_t *, const int64_t *, const double *, const double *, int, const int32_t *, int,
                     double, int, double, double *, double *, uint8_t *, uint8_t *);
int launch_export_pairs(const uint64_t *, const unsigned long long *, unsigned long long, int64_t, uint64_t *);
int launch_compact_gathered(const uint64_t *, int, int64_t, uint64_t *, unsigned long long *);
int launch_level_fused(const uint64_t *, int64_t, const uint64_t *, uint64_t *, int64_t, int64_t,
                       int64_t, unsigned long long *);
int launch_inertia_moments(const double *, int64_t, int64_t, const double *, double *);
int launch_moi_simbits(const double *, int64_t, double, const double *, double, uint64_t *, int64_t);
int launch_tfd_simbits(const double *, int64_t, int64_t, double, int64_t, int64_t, uint64_t *,
                       int64_t);
int launch_clash_self(const double *, int64_t, int64_t, double, double, int64_t *);
int launch_clash_fragments(const double *, int64_t, int64_t, const int64_t *, int64_t, double,
                           int64_t, int64_t *, uint8_t *);
int launch_rototranslate(const double *, int64_t, int64_t, const double *, const double *, double *);
int launch_clash_graph(const double *, int64_t, int64_t, const uint8_t *, double, int64_t *);
int launch_fitness(const double *, int64_t, int64_t, const int64_t *, const double *, int64_t, double,
                   double *, uint8_t *);
int launch_embed_poses_clash(const double *, int64_t, const double *, int64_t, const int64_t *,
                             const int64_t *, const double *, const double *, const double *,
                             const double *, int64_t, double, int64_t, int64_t *, uint8_t *,
                             double *);
int launch_torsion_scan(const double *, int64_t, const int64_t *, int64_t, const uint8_t *,
                        const int16_t *, const int16_t *, const int32_t *, const int32_t *,
                        const int64_t *, int64_t, double, int64_t, double *, int64_t *, const int64_t *, int64_t, double *);
int launch_angle_grid(const int64_t *, const int64_t *, const int64_t *, int64_t, int64_t, int64_t *);
int launch_select_rotated(const int64_t *, int64_t, int64_t *, int64_t *, DevBuf &);
int launch_rows_to_sets(const int64_t *, const int64_t *, int64_t, int64_t *);
int launch_torsion_fingerprint(const double *, int64_t, int64_t, const int64_t *, int64_t, double *);
int launch_tfd_first_match(const double *, int64_t, int64_t, int64_t, double, int64_t *, float *);
int launch_transpose_pad(const double *, int64_t, int64_t, int64_t, double *);
int launch_gather_transpose_pad(const double *, const double *, const int64_t *, int64_t, int64_t, int64_t, double *);
int tfd_ladder_from_first_match(const int64_t *, int64_t, uint8_t *, const int64_t *fm_dev = nullptr);
int tfd_ladder_from_device(const int64_t *fm_dev, int64_t, uint8_t *);
void pyset_order_ints(const int64_t *, int64_t, std::vector<int64_t> &);
void pyset_order_pairs(const int64_t *, int64_t, std::vector<int64_t> &);
int pyset_order_pairs_device(const int64_t *, int64_t, int64_t *);
int warm_clash();
int warm_embed();
int warm_embed3();
int warm_torsion();
int warm_prune();
int warm_h2_check();
int warm_kabsch();
int warm_tfd_ladder();
int tfd_ladder_emulate_device(const int64_t *, int64_t, uint8_t *);
int xyz_write(const char *, const char *const *, int64_t, const double *, int64_t, const char *, int);
int xyz_read(const char *, int64_t *, int64_t *, char *, double *);
int launch_embed_mol_transforms(const double *, int64_t, int64_t, const int64_t *, int, const double *,
                                const double *, int, const double *, int64_t, double *, double *);
int launch_embed_pretransform(const double *, int64_t, int64_t, int64_t, const double *,
                              const double *, int, int64_t, double *);
int last_screen_kind();
void prune_conventions_set(int);
int prune_drop_later();
void comm_teardown();  // fc_comm.cpp
int comm_rank();
int comm_world();
bool comm_ready();
int comm_allgather_dev(const void *, void *, size_t, int);
int launch_embed_grid_clash(const double *, int64_t, int64_t, int64_t, const double *, int64_t, int64_t,
                            int64_t, int64_t, double, int64_t, void *, size_t, uint8_t *, int32_t *);
int launch_string_transforms(const double *, const double *, int64_t, int64_t, const double *, const double *,
                             int64_t, int64_t, const double *, int64_t, double *, double *, int64_t *, int64_t *);
int launch_pose_fingerprints(const double *, int64_t, const double *, int64_t, const int64_t *, const int64_t *,
                             const double *, const double *, int64_t, const int64_t *, int64_t, const uint8_t *,
                             double *);
int launch_leader_chunk(const double *, int64_t, int64_t, int64_t, const uint8_t *, double *, int64_t,
                        unsigned long long *, double, uint8_t *, uint8_t *);
int launch_embed_group_dedupe(const double *, int64_t, int64_t, int64_t, const double *, int64_t, int64_t,
                              int64_t, double, const uint8_t *, uint8_t *);

// ---- error state / context -----------------------------------------------------
std::string &last_error() {
  static thread_local std::string e;
  return e;
}

int set_error(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

Context &ctx() {
  static Context c;
  return c;
}

std::recursive_mutex &api_mutex() {
  static std::recursive_mutex *m = new std::recursive_mutex;  // never destroyed (see pool())
  return *m;
}

// ---- caching pool behind DevBuf (fc_common.h) ---------------------------------------------
namespace {
struct Pool {
  std::mutex mu;
  std::multimap<size_t, void *> free_blocks;  // capacity -> block
  size_t cached = 0;
  size_t limit = (size_t)8192 << 20;
  bool limit_read = false;
};
Pool &pool() {
  static Pool *p = new Pool;  // never destroyed: DevBufs with static storage may outlive it otherwise
  return *p;
}
size_t size_class(size_t n) {
  if (n <= 256) return 256;
  if (n <= ((size_t)1 << 20)) {
    size_t c = 256;
    while (c < n) c <<= 1;
    return c;
  }
  const size_t step = (size_t)2 << 20;
  return (n + step - 1) / step * step;
}
}  // namespace

void *pool_take(size_t n, size_t *capacity, bool any_larger) {
  Pool &P = pool();
  const size_t want = size_class(n);
  {
    std::lock_guard<std::mutex> lock(P.mu);
    if (!P.limit_read) {
      if (const char *v = getenv("FC_POOL_MB")) P.limit = (size_t)std::strtoull(v, nullptr, 10) << 20;
      P.limit_read = true;
    }
    auto it = P.free_blocks.lower_bound(want);
    if (it != P.free_blocks.end() && (any_larger || it->first <= 2 * want)) {  // never hand a huge block to a small LASTING request
      void *p = it->second;
      *capacity = it->first;
      P.cached -= it->first;
      P.free_blocks.erase(it);
      return p;
    }
  }
  void *p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    pool_trim();  // give the cached blocks back and try once more
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  *capacity = want;
  return p;
}

void pool_give(void *p, size_t capacity) {
  if (!p) return;
  Pool &P = pool();
  {
    std::lock_guard<std::mutex> lock(P.mu);
    if (ctx().ready && P.cached + capacity <= P.limit) {
      P.free_blocks.emplace(capacity, p);
      P.cached += capacity;
      return;
    }
  }
  (void)hipFree(p);
}

void pool_trim() {
  Pool &P = pool();
  std::multimap<size_t, void *> blocks;
  {
    std::lock_guard<std::mutex> lock(P.mu);
    blocks.swap(P.free_blocks);
    P.cached = 0;
  }
  for (auto &b : blocks) (void)hipFree(b.second);
}


// everything the context owns on its device: streams, events, pinned staging, cached blocks
static void context_teardown() {
  Context &c = ctx();
  if (!c.ready) return;
  (void)hipStreamSynchronize(c.stream);
  for (hipStream_t s : {c.s_screen, c.s_lane[0], c.s_lane[1], c.s_lane[2], c.s_comm})
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  c.s_screen = c.s_lane[0] = c.s_lane[1] = c.s_lane[2] = c.s_comm = nullptr;
  for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
  c.ev_pool.clear();
  for (hipEvent_t e : c.ev_dep_pool) (void)hipEventDestroy(e);
  c.ev_dep_pool.clear();
  for (hipEvent_t *e : {&c.ev0, &c.ev1, &c.ev2, &c.ev3, &c.ev_reset, &c.ev_screened, &c.ev_comm[0], &c.ev_comm[1]})
    if (*e) {
      (void)hipEventDestroy(*e);
      *e = nullptr;
    }
  pool_trim();  // cached blocks belong to the device being left
  (void)hipStreamDestroy(c.own_stream);
  c.stream = c.own_stream = nullptr;
  if (c.pinned) (void)hipHostFree(c.pinned);
  c.pinned = nullptr;
  if (c.pinned_word) (void)hipHostFree(c.pinned_word);
  c.pinned_word = nullptr;
  if (c.ladder_all) (void)hipFree(c.ladder_all);
  c.ladder_all = nullptr;
  c.pinned_bytes = 0;
  for (auto &set : c.stage)
    for (int b = 0; b < 2; ++b) {
      if (set.pin[b]) (void)hipHostFree(set.pin[b]);
      if (set.ev[b]) (void)hipEventDestroy(set.ev[b]);
      set.pin[b] = nullptr;
      set.ev[b] = nullptr;
      set.busy = false;
    }
  c.mark_after_screen = nullptr;
  c.ready = false;
}

static int do_init(int device) {
  Context &c = ctx();
  if (c.ready && c.device == device) return FC_OK;
  context_teardown();  // a device switch: nothing of the old context survives (ensembles are refused by epoch)
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(FC_E_NODEVICE, "no HIP device available (%s); libfc_hip has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= n)
    return set_error(FC_E_INVALID, "device %d out of range (have %d)", device, n);
  if (hipSetDevice(device) != hipSuccess)
    return set_error(FC_E_NODEVICE, "hipSetDevice(%d) failed", device);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return set_error(FC_E_NODEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(FC_E_NODEVICE, "device %d is %s; this library is built for gfx950 only",
                     device, prop.gcnArchName);
  FC_HIP_TRY(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
  c.stream = c.own_stream;
  FC_HIP_TRY(hipEventCreate(&c.ev0));
  FC_HIP_TRY(hipEventCreate(&c.ev1));
  FC_HIP_TRY(hipEventCreate(&c.ev2));
  FC_HIP_TRY(hipEventCreate(&c.ev3));
  c.device = device;
  c.n_cu = prop.multiProcessorCount;
  c.hbm = prop.totalGlobalMem;
  std::snprintf(c.name, sizeof c.name, "%s (%s)", prop.name, prop.gcnArchName);
  ++c.epoch;
  c.ready = true;
  return FC_OK;
}

int side_streams() {
  Context &c = ctx();
  if (c.s_screen) return FC_OK;
  // The screen fills every workgroup slot of the chip (three per CU): the small kernels of the
  // previous prune (and the collective behind them) get compute units only if the dispatcher
  // prefers them, so their streams have the highest priority and the screens' the lowest.
  // (Without: a refine launched beside a screen took 450 us instead of 45 and the prune after
  // next waited for it.)
  int least = 0, greatest = 0;
  FC_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  FC_HIP_TRY(hipStreamCreateWithPriority(&c.s_screen, hipStreamNonBlocking, least));
  FC_HIP_TRY(hipStreamCreateWithPriority(&c.s_lane[0], hipStreamNonBlocking, greatest));
  FC_HIP_TRY(hipStreamCreateWithPriority(&c.s_lane[1], hipStreamNonBlocking, greatest));
  FC_HIP_TRY(hipStreamCreateWithPriority(&c.s_lane[2], hipStreamNonBlocking, greatest));
  FC_HIP_TRY(hipStreamCreateWithPriority(&c.s_comm, hipStreamNonBlocking, greatest));
  FC_HIP_TRY(hipEventCreateWithFlags(&c.ev_reset, hipEventDisableTiming));
  FC_HIP_TRY(hipEventCreateWithFlags(&c.ev_screened, hipEventDisableTiming));
  FC_HIP_TRY(hipEventCreateWithFlags(&c.ev_comm[0], hipEventDisableTiming));
  FC_HIP_TRY(hipEventCreateWithFlags(&c.ev_comm[1], hipEventDisableTiming));
  return FC_OK;
}

namespace {
constexpr size_t kStagePiece = (size_t)4 << 20;
struct StageLease {  // one set of pinned pieces for the duration of one staged copy
  Context::StageSet *set = nullptr;
  int acquire() {
    Context &c = ctx();
    {
      std::lock_guard<std::mutex> lock(c.stage_mu);
      for (auto &cand : c.stage)
        if (!cand.busy) {
          cand.busy = true;
          set = &cand;
          break;
        }
    }
    if (!set) return set_error(FC_E_LIMIT, "more than %d staged copies at once", Context::kStageSets);
    for (int b = 0; b < 2; ++b) {
      if (!set->pin[b] && hipHostMalloc(&set->pin[b], kStagePiece, hipHostMallocDefault) != hipSuccess)
        return set_error(FC_E_NOMEM, "pinned staging memory: hipHostMalloc failed");
    }
    for (int b = 0; b < 2; ++b)
      if (!set->ev[b] && hipEventCreateWithFlags(&set->ev[b], hipEventDisableTiming) != hipSuccess)
        return set_error(FC_E_HIP, "hipEventCreate failed");
    return FC_OK;
  }
  ~StageLease() {
    if (set) {
      std::lock_guard<std::mutex> lock(ctx().stage_mu);
      set->busy = false;
    }
  }
};
}  // namespace

bool staged_uploads() {
  static const bool on = [] {
    // measured (cfg3 search, ten runs per process, tools/attic/rescan_probe.py): with uploads on the runtime's own path one run
    // in three still lost 9-22 ms in the kernel behind the free of an uploaded array; with the pinned detour none did.
    // Price: memcpy + DMA instead of DMA from the caller's pages, +0.12 ms per 12 MB (FC_STAGED_UPLOADS=0: direct)
    const char *v = getenv("FC_STAGED_UPLOADS");
    return v ? atoi(v) != 0 : true;
  }();
  return on;
}

bool host_memory_is_pinned(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary host pointer: not an error of ours
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int d2h_staged(void *dst, const void *src_dev, size_t n, hipStream_t st) {
  if (n == 0) return FC_OK;
  StageLease lease;
  FC_TRY(lease.acquire());
  Context::StageSet &S = *lease.set;
  const char *src = static_cast<const char *>(src_dev);
  char *out = static_cast<char *>(dst);
  size_t issued = 0, copied = 0;
  while (copied < n) {
    // two pieces in flight: request the next one(s), then collect the oldest
    while (issued < n && issued - copied < 2 * kStagePiece) {
      const int b = (int)((issued / kStagePiece) & 1);
      const size_t len = std::min(kStagePiece, n - issued);
      FC_HIP_TRY(hipMemcpyAsync(S.pin[b], src + issued, len, hipMemcpyDeviceToHost, st));
      FC_HIP_TRY(hipEventRecord(S.ev[b], st));
      issued += len;
    }
    const int b = (int)((copied / kStagePiece) & 1);
    const size_t len = std::min(kStagePiece, n - copied);
    FC_HIP_TRY(hipEventSynchronize(S.ev[b]));
    std::memcpy(out + copied, S.pin[b], len);
    copied += len;
  }
  return FC_OK;
}

int h2d_staged(void *dst_dev, const void *src, size_t n, hipStream_t st) {
  if (n == 0) return FC_OK;
  StageLease lease;
  FC_TRY(lease.acquire());
  Context::StageSet &S = *lease.set;
  const char *in = static_cast<const char *>(src);
  char *dst = static_cast<char *>(dst_dev);
  size_t done = 0;
  int64_t piece = 0;
  bool used[2] = {false, false};
  while (done < n) {
    const int b = (int)(piece & 1);
    const size_t len = std::min(kStagePiece, n - done);
    if (used[b]) FC_HIP_TRY(hipEventSynchronize(S.ev[b]));  // the DMA out of this piece two turns ago
    std::memcpy(S.pin[b], in + done, len);
    FC_HIP_TRY(hipMemcpyAsync(dst + done, S.pin[b], len, hipMemcpyHostToDevice, st));
    FC_HIP_TRY(hipEventRecord(S.ev[b], st));
    used[b] = true;
    done += len;
    ++piece;
  }
  for (int b = 0; b < 2; ++b)
    if (used[b]) FC_HIP_TRY(hipEventSynchronize(S.ev[b]));  // the pieces go back to the pool idle
  return FC_OK;
}

int pinned_reserve(size_t bytes) {
  Context &c = ctx();
  if (bytes <= c.pinned_bytes) return FC_OK;
  if (c.pinned) (void)hipHostFree(c.pinned);
  c.pinned = nullptr;
  c.pinned_bytes = 0;
  const size_t want = std::max<size_t>(bytes, 1 << 20);
  FC_HIP_TRY(hipHostMalloc(&c.pinned, want, hipHostMallocDefault));
  c.pinned_bytes = want;
  return FC_OK;
}

int ensure_init() {
  if (ctx().ready) {
    // the calling thread may differ from the one that initialised
    if (hipSetDevice(ctx().device) != hipSuccess)
      return set_error(FC_E_NODEVICE, "hipSetDevice(%d) failed", ctx().device);
    return FC_OK;
  }
  return do_init(0);
}

// upload helper: host array -> fresh device buffer
template <class T>
static int upload(DevBuf &b, const T *host, size_t count) {
  FC_TRY(b.reserve(count * sizeof(T)));
  return h2d(b.p, host, count * sizeof(T));
}

static int make_selection(const uint8_t *atom_mask, int64_t A_all, std::vector<int32_t> &sel) {
  sel.clear();
  for (int64_t a = 0; a < A_all; ++a)
    if (atom_mask == nullptr || atom_mask[a]) sel.push_back((int32_t)a);
  if (sel.empty()) return set_error(FC_E_INVALID, "atom_mask selects no atom");
  return FC_OK;
}

// prepared layout of N conformers taken from device-resident raw coordinates (all atoms, AoS):
// conformer n is raw[conf_idx[n]] (conf_idx_dev == nullptr: raw[n])
// defer_wait: return with the preparation kernel and the copy of the largest G still in flight -- the caller does host
// work that does not need them (fc_prune_rmsd_host: the prune's reserves and item table) and then calls
// ensemble_build_finish
static int ensemble_build_finish(fc_ensemble *e) {
  if (!e->g_max_pending) return FC_OK;
  e->g_max_pending = false;
  FC_TRY(sync());
  std::memcpy(&e->g_max, ctx().pinned_word, sizeof(double));
  auto *gmax_dev = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
  FC_HIP_TRY(hipMemsetAsync(gmax_dev, 0, sizeof(unsigned long long), ctx().stream));
  return FC_OK;
}

// The largest G of a SMALL ensemble from the caller's array, on the host: the preparation kernel's arithmetic in its order
// (running sums over the selected atoms, the division, x*x + y*y + z*z per atom; nothing is fused on either side), so that
// fc_prune_rmsd_host needs no wait between the preparation and the prune -- at the sizes of FIRECODE's own runs (hundreds of
// conformers) the call is a chain of waits and launches, not of kernels (0.16 ms at 100 conformers).  Used a hair larger
// than computed: the band rule and the split-half scale are both conservative in a LARGER value.
static double host_largest_g(const double *coords, int64_t N, int64_t A_all, const std::vector<int32_t> &sel, int center) {
  const int64_t A = (int64_t)sel.size();
  double gmax = 0.0;
  for (int64_t n = 0; n < N; ++n) {
    const double *t = coords + n * A_all * 3;
    double cx = 0.0, cy = 0.0, cz = 0.0;
    if (center) {
      for (int64_t a = 0; a < A; ++a) {
        const double *r = t + (int64_t)sel[(size_t)a] * 3;
        cx += r[0];
        cy += r[1];
        cz += r[2];
      }
      cx /= (double)A;
      cy /= (double)A;
      cz /= (double)A;
    }
    double g = 0.0;
    for (int64_t a = 0; a < A; ++a) {
      const double *r = t + (int64_t)sel[(size_t)a] * 3;
      const double x = r[0] - cx, y = r[1] - cy, z = r[2] - cz;
      g += x * x + y * y + z * z;
    }
    if (g == g && g > gmax) gmax = g;
  }
  return gmax;
}
// N * A * 3 up to which the host pass beats the wait: measured at 50 atoms -- 100 conformers 0.148 against 0.161 ms per call,
// 300 conformers even, 1 000 conformers 0.247 against 0.228 (the pass is three dependent chains of additions per conformer)
constexpr int64_t kHostGmaxDoubles = 36000;

// Two halves: what does not need the coordinates on the device (selection, reserves, the selection's upload, the reset of the
// largest-G word) and the preparation launch behind them -- ensemble_build issues the first half in FRONT of the upload, so
// that nothing but the launch itself stands between the last piece's DMA and the kernel (the trace of one
// prune_by_rmsd(host arrays) call showed 39 us there).
static int ensemble_build_prepare(int64_t N, int64_t A_all, const uint8_t *atom_mask, fc_ensemble *e, DevBuf &dsel) {
  std::vector<int32_t> &sel = e->sel_host;
  FC_TRY(make_selection(atom_mask, A_all, sel));
  e->N = N;
  e->A = (int64_t)sel.size();
  {  // (Context::hint_*: the last prune of an ensemble of this shape, scaled to this one's number of pairs)
    const Context &c = ctx();
    if (c.hint_A == e->A && c.hint_N >= 2 && N >= c.hint_N / 2 && N <= 2 * c.hint_N) {
      const double scale = ((double)N * (double)N) / ((double)c.hint_N * (double)c.hint_N);
      e->last_candidates = (int64_t)((double)c.hint_candidates * scale);
      e->last_similar = (int64_t)((double)c.hint_similar * scale);
    }
  }
  e->Npad = ceil_div(std::max<int64_t>(N, 1), 64) * 64;
  e->W = e->Npad / 64;
  FC_TRY(e->Xs.reserve((size_t)((e->A + 3) / 4 * 4) * 3 * e->Npad * sizeof(double)));
  FC_TRY(e->G.reserve((size_t)e->Npad * sizeof(double)));
  FC_TRY(e->Xa.reserve((size_t)std::max<int64_t>(N, 1) * e->A * 3 * sizeof(double)));
  FC_TRY(e->counters.reserve(kCounters * sizeof(uint64_t)));
  FC_TRY(upload(dsel, sel.data(), sel.size()));
  return launch_prep_begin(e);
}

static int ensemble_build_launch(const double *raw_dev, int64_t N, int64_t A_all, int center, const int32_t *conf_idx_dev,
                                 fc_ensemble *e, DevBuf &dsel, bool defer_wait, bool host_gmax = false) {
  FC_TRY(launch_prep_body(raw_dev, N, A_all, dsel.as<int32_t>(), e->A, center, e, conf_idx_dev));
  if (host_gmax) return FC_OK;  // (the caller computes the largest G from its array: fc_prune_rmsd_host on small ensembles)
  // the largest G (left by the prep kernel in the last counter word) comes back behind the same wait
  unsigned long long gmax_bits = 0;
  auto *gmax_dev = reinterpret_cast<unsigned long long *>(e->counters.p) + (kCounters - 1);
  if (defer_wait) {
    Context &c = ctx();
    if (!c.pinned_word && hipHostMalloc(&c.pinned_word, 64, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      c.pinned_word = nullptr;
    }
    if (c.pinned_word) {
      FC_HIP_TRY(hipMemcpyAsync(c.pinned_word, gmax_dev, sizeof gmax_bits, hipMemcpyDeviceToHost, c.stream));
      e->g_max_pending = true;
      return FC_OK;  // (dsel returns to the pool: its next user is ordered behind the kernel on this stream)
    }
  }
  FC_TRY(d2h(&gmax_bits, gmax_dev, sizeof gmax_bits));
  FC_TRY(sync());  // (also keeps `sel` / dsel alive until the kernel has read them)
  std::memcpy(&e->g_max, &gmax_bits, sizeof(double));
  FC_HIP_TRY(hipMemsetAsync(gmax_dev, 0, sizeof(unsigned long long), ctx().stream));
  return FC_OK;
}

static int ensemble_build_dev(const double *raw_dev, int64_t N, int64_t A_all, const uint8_t *atom_mask,
                              int center, const int32_t *conf_idx_dev, fc_ensemble *e, bool defer_wait = false) {
  DevBuf dsel;
  FC_TRY(ensemble_build_prepare(N, A_all, atom_mask, e, dsel));
  return ensemble_build_launch(raw_dev, N, A_all, center, conf_idx_dev, e, dsel, defer_wait);
}

// Host arrays in (the drop-in call prune_by_rmsd(structures, ...)): the coordinates go through the pinned pieces like every
// large upload from pageable memory (h2d_staged; fc_common.h says why the caller's pages are not handed to the runtime,
// nor registered by the library for the duration of the copy).
// What that costs and what was tried against it (round 4; tools/attic/pin_probe.py, tools/hostin_breakdown.py; 12 MB):
// DMA from pinned memory 0.22 ms (54 GB/s), from the caller's pageable pages THE SAME 0.22 ms (the driver maps them; that
// mapping is what later stalls the queues when the caller frees the array), memmove into pinned memory 0.24 ms on one
// core -- piece by piece (3 x 4 MB, copy of piece k + 1 beside the DMA of piece k) 0.44 ms per ensemble with the
// preparation kernel.  Built and not kept: ~2 MB pieces of whole 64-conformer tiles, each followed by its own preparation
// launch -- on one stream every copy -> kernel -> copy hand-over between the copy engine and the compute queue costs
// ~25 us (0.67 ms); with the copies on a stream of their own every extra piece costs ~10 us and every event ~7 us
// (0.53 ms); the copy of each 4 MB piece split between this thread and a helper thread (0.41 ms: a thread per call
// for 0.03 ms).
static int ensemble_build(const double *coords, int64_t N, int64_t A_all, const uint8_t *atom_mask,
                          int center, fc_ensemble *e, bool defer_wait = false, bool host_gmax = false) {
  DevBuf dsel, raw;
  FC_TRY(ensemble_build_prepare(N, A_all, atom_mask, e, dsel));
  FC_TRY(upload(raw, coords, (size_t)N * A_all * 3));
  return ensemble_build_launch(raw.as<double>(), N, A_all, center, nullptr, e, dsel, defer_wait, host_gmax);
}

// (re)shape the bit-matrix workspace for a given sharding
static int ensemble_shard(fc_ensemble *e, int64_t rank, int64_t world, int64_t row_block) {
  FC_REQUIRE(e->epoch == 0 || e->epoch == ctx().epoch,
             "this ensemble was created before fc_shutdown / a device switch: create it again");
  FC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world %lld/%lld", (long long)rank,
             (long long)world);
  FC_REQUIRE(row_block >= 32 && row_block % 32 == 0 && row_block <= 4096,
             "row_block must be a multiple of 32 in [32, 4096]");
  e->rank = rank;
  e->world = world;
  e->row_block = row_block;
  const int64_t n_gblocks = ceil_div(e->N, row_block);
  const int64_t n_lblocks = local_block_count(n_gblocks, rank, world);
  e->rows_local = n_lblocks * row_block;
  if ((uint64_t)e->rows_local * (uint64_t)e->W >= (1ull << 32))
    return set_error(FC_E_LIMIT, "bit matrix of %lld x %lld words exceeds the 32-bit word index",
                     (long long)e->rows_local, (long long)e->W);
  FC_TRY(e->bits.reserve(std::max<size_t>((size_t)e->rows_local * e->W * sizeof(uint64_t), 8)));
  FC_TRY(e->cand.reserve(std::max<size_t>((size_t)e->rows_local * e->W * sizeof(uint32_t), 8)));
  // candidate-pair queue: 256 entries per conformer, 1M..64M entries
  e->pairq_cap = std::min<int64_t>(std::max<int64_t>(256 * e->N, 1 << 20), 1 << 26);
  if (const char *v = getenv("FC_PAIRQ_CAP")) {  // test knob: force the word-queue fallback
    const long long c = std::strtoll(v, nullptr, 10);
    if (c >= 1 && c <= (1ll << 26)) e->pairq_cap = c;
  }
  FC_TRY(e->pairq.reserve((size_t)e->pairq_cap * sizeof(uint64_t)));
  FC_TRY(e->simq.reserve((size_t)e->pairq_cap * sizeof(uint64_t)));
  {  // buckets of the long-queue refine: (row buckets of 1 024) x (column tiles of 64), numbered supertile by supertile
     // (8 column tiles of one row bucket; fc_kabsch.hip); sized here, before any pipeline forks
    const int64_t n_rb = ceil_div(e->N, (int64_t)kBucketRows), n_sc = ceil_div(e->Npad >> kBucketColShift, (int64_t)8);
    const int64_t n_st = n_rb * n_sc, nb = n_st * 8;
    e->bk_buckets = nb <= ((int64_t)1 << 22) ? nb : 0;  // one workgroup scans the counts; work items are bucket | piece << 24
    if (e->bk_buckets > 0) {
      FC_TRY(e->bk.reserve((size_t)(512 + 2 * nb + 1) * sizeof(int)));
      FC_TRY(e->bk_off.reserve((size_t)((nb + 1) + (n_st + 1) + 8 * ((n_st + 7) / 8 + 1)) * sizeof(int)));  // offsets | supertile starts | per-XCD prefixes
      FC_TRY(e->bk_list.reserve((size_t)(nb + e->pairq_cap / 256 + 1) * sizeof(int)));  // pieces of <= 512 pairs (256 in tuning builds)
      FC_TRY(e->sortq.reserve((size_t)e->pairq_cap * sizeof(uint64_t)));
    }
  }
  FC_TRY(e->maskA.reserve((size_t)e->Npad));
  FC_TRY(e->maskB.reserve((size_t)e->Npad));
  FC_TRY(e->mbits.reserve((size_t)e->W * sizeof(uint64_t)));
  e->bits_valid = false;
  return FC_OK;
}


// rows of the bit matrix per workgroup (tuning knob FC_ROW_BLOCK, multiple of 128)
static int64_t default_row_block() {
  const char *v = getenv("FC_ROW_BLOCK");
  if (v) {
    const long r = std::strtol(v, nullptr, 10);
    if (r >= 64 && r <= 4096 && r % 64 == 0) return r;
  }
  return 128;  // measured best on cfg2 (tail and balance beat the extra LDS fills)
}

// similarity bits of this rank's rows: screen + exact refine; counters[1..3]
static int simbits_local(fc_ensemble *e, double max_rmsd, double max_dev, const double *energies,
                         double max_dE, bool zero_counters, bool lean = false) {
  e->lean = lean;
  const double *en_dev = nullptr;
  if (energies != nullptr) {
    FC_TRY(upload(e->energies, energies, (size_t)e->N));
    en_dev = e->energies.as<double>();
  }
  if (zero_counters)
    FC_HIP_TRY(hipMemsetAsync(e->counters.p, 0, kCounters * sizeof(uint64_t), ctx().stream));
  // HIP events bracket the screen kernel (the dominant one) on the library's stream
  FC_HIP_TRY(hipEventRecord(ctx().ev2, ctx().stream));
  ctx().mark_after_screen = ctx().ev3;  // recorded by the launcher right behind the screen kernel
  const int rc_screen = launch_simbits_screen(e, max_rmsd * max_rmsd + kScreenMargin);
  ctx().mark_after_screen = nullptr;
  FC_TRY(rc_screen);
  FC_TRY(launch_simbits_refine(e, max_rmsd, max_dev, en_dev, max_dE));
  e->bits_valid = true;
  return FC_OK;
}

// duration of the last screen kernel enqueued by simbits_local, after a sync
static int64_t last_screen_ns() {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, ctx().ev2, ctx().ev3) != hipSuccess) return 0;
  return (int64_t)(ms * 1e6);
}

static const int64_t kLadder[] = {500000, 200000, 100000, 50000, 20000, 10000, 5000, 2000, 1000,
                                  500,    200,    100,    50,    20,    10,    5,    2,    1};

// whole ladder on one device (world == 1), enqueued without host round trips:
// one fused launch per ladder value that can still apply (the first ones are
// ruled out on the host from N alone), one sync at the end.
// pairs_dev != nullptr: a device list of the exactly-similar pairs whose length is the device
// counter counters[2] (refine output; complete iff counters[6] <= pairq_cap) or, with
// pairs_are_final, counters[2] alone (list uploaded by the caller).  Then the one-launch
// k_ladder_pairs runs first and the bit-matrix levels behind it return at once.
static const unsigned long long kPairLadderCap = 1ull << 20;

static int ladder_single(fc_ensemble *e, const uint64_t *bits_dev, int64_t min_per_group,
                         uint8_t *mask_out, int64_t *levels, int64_t *survivors,
                         unsigned long long *counters_out = nullptr, const uint64_t *pairs_dev = nullptr,
                         bool pairs_are_final = false, bool counters_zeroed = false, int64_t defer_slot = -1,
                         int64_t slot_stride = 0) {
  // defer_slot >= 0: enqueue the pair ladder and the copy of its result into slot `defer_slot` of
  // the pinned staging area and return WITHOUT waiting (the caller synchronises once for many
  // prunes and reads the slots with ladder_collect; pinned memory for all slots is the caller's;
  // slot_stride: words per slot when the prunes differ in size, default W + 16)
  const int64_t N = e->N, W = e->W;
  const int n_ladder = (int)(sizeof(kLadder) / sizeof(kLadder[0]));
  FC_TRY(e->ladder.reserve(((size_t)(n_ladder + 1) * W + 16) * sizeof(uint64_t)));
  if (defer_slot < 0) FC_TRY(pinned_reserve((size_t)(W + 16) * sizeof(uint64_t)));
  uint64_t *mb = e->ladder.as<uint64_t>();
  auto *cnt = reinterpret_cast<unsigned long long *>(e->counters.p);
  // ladder values that can ever run (n_active <= N) -- decided here, the rest on the device
  std::vector<int64_t> ks;
  for (int64_t k : kLadder)
    if (k == 1 || min_per_group * k < N) ks.push_back(k);
  const int n_lv = (int)ks.size();
  // counters[8] = levels run, counters[9] = "k_ladder_pairs produced the mask"
  // [8], [9]: ladder flags; [10], [11]: spare; [16 ..): bucket fill levels of the pair ladder
  if (!counters_zeroed) FC_HIP_TRY(hipMemsetAsync(cnt + 8, 0, (kCounters - 8) * sizeof(uint64_t), ctx().stream));
  uint64_t *words = static_cast<uint64_t *>(ctx().pinned) + (defer_slot > 0 ? (size_t)defer_slot * (size_t)(slot_stride > 0 ? slot_stride : W + 16) : 0);
  uint64_t *cnt_host = words + W;
  bool have_mask = false;
  const bool lds_ok = (size_t)2 * W * sizeof(uint64_t) <= 60 * 1024;
  if (pairs_dev != nullptr && lds_ok) {
    // sparse similarity (the usual case): the whole ladder is ONE launch over the pair list
    // the values that can run at this N are a suffix of kLadder: they sit on the device once per context (a launch of
    // k_store_ladder_ks per fresh ensemble -- every drop-in call -- was 5 us of an otherwise empty device)
    const int64_t *ks_dev = nullptr;
    {
      bool suffix = n_lv >= 1 && n_lv <= n_ladder;
      for (int q = 0; suffix && q < n_lv; ++q) suffix = ks[(size_t)q] == kLadder[n_ladder - n_lv + q];
      Context &c = ctx();
      if (suffix && c.ladder_all == nullptr) {
        if (hipMalloc(reinterpret_cast<void **>(&c.ladder_all), sizeof kLadder) == hipSuccess) {
          // (a blocking copy, once per context: the first ladder may run on a lane of a pipeline, whose other lanes must not
          // find the values still in flight)
          if (hipMemcpy(c.ladder_all, kLadder, sizeof kLadder, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(c.ladder_all);
            c.ladder_all = nullptr;
          }
        } else {
          (void)hipGetLastError();
          c.ladder_all = nullptr;
        }
      }
      if (suffix && c.ladder_all != nullptr) ks_dev = c.ladder_all + (n_ladder - n_lv);
    }
    if (ks_dev == nullptr && (e->ladder_k_n != n_lv || e->ladder_k_mpg != min_per_group)) {
      // the source of the (asynchronous) copy lives with the ensemble: no host wait here -- a wait at this point sat
      // between the refine and the ladder of every drop-in call (each creates its ensemble) and cost it ~45 us
      // (the values travel as kernel arguments: an asynchronous copy out of a pageable vector could be overtaken by the next
      // prune's reassignment of that vector, or by the ensemble's destruction)
      e->ladder_k_host = ks;
      FC_TRY(e->ladder_k.reserve(ks.size() * sizeof(int64_t)));
      LadderKs args{};
      args.n = (int)std::min<size_t>(ks.size(), 24);
      for (int q = 0; q < args.n; ++q) args.k[q] = ks[(size_t)q];
      hipLaunchKernelGGL(k_store_ladder_ks, dim3(1), dim3(32), 0, ctx().stream, args, e->ladder_k.as<int64_t>());
      FC_TRY(check_launch("k_store_ladder_ks"));
      e->ladder_k_n = n_lv;
      e->ladder_k_mpg = min_per_group;
    }
    // level buckets: one per level, each as long as the longest list the ladder accepts
    const unsigned long long ladder_cap =
        std::min<unsigned long long>(kPairLadderCap, std::max<unsigned long long>(1024, (unsigned long long)N * (N - 1) / 2));
    FC_TRY(e->levelmask.reserve((size_t)ladder_cap * (size_t)n_lv * sizeof(uint64_t)));
    // long lists: a launch per level over the whole chip instead of one workgroup (fc_prune.hip); a host decision from
    // the last similar-pair count seen for these coordinates -- either form is correct for any list
    static const bool many_ok = [] {
      const char *v = getenv("FC_LADDER_MANY");  // 0: always the one-workgroup ladder
      return !(v && atoi(v) == 0);
    }();
    if (many_ok && n_lv >= 2 && e->last_similar > ((int64_t)1 << 17))
      FC_TRY(launch_ladder_pairs_many(pairs_dev, e->levelmask.as<uint64_t>(), cnt + 2, pairs_are_final ? nullptr : cnt + 6,
                                      (unsigned long long)e->pairq_cap, ladder_cap, N, W, min_per_group,
                                      ks_dev ? ks_dev : e->ladder_k.as<int64_t>(), ks.data(), n_lv, mb, mb + (size_t)n_lv * W, cnt));
    else
      FC_TRY(launch_ladder_pairs(pairs_dev, e->levelmask.as<uint64_t>(), cnt + 2,
                                 pairs_are_final ? nullptr : cnt + 6,
                                 (unsigned long long)e->pairq_cap, ladder_cap, N, W, min_per_group,
                                 ks_dev ? ks_dev : e->ladder_k.as<int64_t>(), n_lv, mb + (size_t)n_lv * W, cnt));
    // mask words and the 16 counters behind them (written by the kernel): one copy
    FC_TRY(d2h(words, mb + (size_t)n_lv * W, (size_t)(W + 16) * sizeof(uint64_t)));
    if (defer_slot >= 0) return FC_OK;
    FC_TRY(sync());
    have_mask = cnt_host[9] != 0;
  }
  if (defer_slot >= 0) return set_error(FC_E_INVALID, "deferred ladder needs the pair list");
  if (!have_mask) {
    // dense similarity / no pair list: one fused launch per level over the bit matrix
    if (prune_drop_later())
      return set_error(FC_E_INVALID, "fc_prune_conventions(drop_later = 1) is implemented by the pair ladder only; "
                       "this prune needs the bit-matrix levels (dense similarity or no pair list)");
    if (bits_dev == nullptr) return set_error(FC_E_LIMIT, "pair list too long for the one-launch ladder and no bit matrix given");
    FC_TRY(launch_mask_init(mb, N, W, (int64_t)(n_lv + 1) * W));
    int cur = 0;
    for (int64_t k : ks) {
      FC_TRY(launch_level_fused(bits_dev, W, mb + (size_t)cur * W, mb + (size_t)(cur + 1) * W, N, k,
                                min_per_group, cnt));
      ++cur;
    }
    FC_TRY(d2h(words, mb + (size_t)n_lv * W, (size_t)W * sizeof(uint64_t)));
    FC_TRY(d2h(cnt_host, cnt, 16 * sizeof(uint64_t)));
    FC_TRY(sync());
  }
  int64_t alive = 0;
  for (int64_t w = 0; w < W; ++w) alive += __builtin_popcountll(words[w]);
  if (mask_out)
    for (int64_t i = 0; i < N; ++i) mask_out[i] = (uint8_t)((words[(size_t)(i >> 6)] >> (i & 63)) & 1ull);
  if (levels) *levels = (int64_t)cnt_host[8];
  if (survivors) *survivors = alive;
  if (counters_out)
    for (int k = 0; k < 8; ++k) counters_out[k] = cnt_host[k];
  return FC_OK;
}

// result of a deferred pair ladder (after the caller's synchronisation); false: the kernel
// declined (queue overflow / list too long) and the prune has to be redone synchronously
// what a finished prune says about the length of this ensemble's candidate queue -> every workspace over its coordinates
static void note_candidates(fc_ensemble *e, unsigned long long refined, unsigned long long similar) {
  // (every workspace of the chain, whichever lane the prune ran on: the choice of the refine's and the ladder's form must not
  // depend on which lane finished last)
  int guard = 0;
  for (fc_ensemble *w = e->head ? e->head : e; w != nullptr && guard < 8; w = w->twin, ++guard) {
    w->last_candidates = (int64_t)refined;
    w->last_similar = (int64_t)similar;
  }
  Context &c = ctx();
  c.hint_A = e->A, c.hint_N = e->N, c.hint_candidates = (int64_t)refined, c.hint_similar = (int64_t)similar;
}

static bool ladder_collect(fc_ensemble *e, int64_t slot, uint8_t *mask_out, int64_t *levels,
                           int64_t *survivors, unsigned long long *counters_out, int64_t slot_stride = 0) {
  const int64_t N = e->N, W = e->W;
  const uint64_t *words = static_cast<const uint64_t *>(ctx().pinned) +
                          (size_t)slot * (size_t)(slot_stride > 0 ? slot_stride : W + 16);
  const uint64_t *cnt_host = words + W;
  note_candidates(e, cnt_host[6], cnt_host[2]);  // (pairs the screen queued; valid also when the ladder declined)
  if (cnt_host[9] == 0) return false;
  int64_t alive = 0;
  for (int64_t w = 0; w < W; ++w) alive += __builtin_popcountll(words[w]);
  if (mask_out)
    for (int64_t i = 0; i < N; ++i) mask_out[i] = (uint8_t)((words[(size_t)(i >> 6)] >> (i & 63)) & 1ull);
  if (levels) *levels = (int64_t)cnt_host[8];
  if (survivors) *survivors = alive;
  if (counters_out)
    for (int k = 0; k < 8; ++k) counters_out[k] = cnt_host[k];
  return true;
}

// second prune workspace over the coordinates of `ens` (created once, destroyed with it)
static int ensemble_twin(fc_ensemble *ens, fc_ensemble **out) {
  if (!ens->twin) {
    std::unique_ptr<fc_ensemble> t(new (std::nothrow) fc_ensemble);
    if (!t) return set_error(FC_E_NOMEM, "host allocation failed");
    t->epoch = ens->epoch;
    t->N = ens->N, t->A = ens->A, t->Npad = ens->Npad, t->W = ens->W;
    t->Xs.alias(ens->Xs), t->Xa.alias(ens->Xa), t->G.alias(ens->G);
    if (ens->xsf_valid) t->Xsf.alias(ens->Xsf), t->sub.alias(ens->sub), t->xsf_valid = true;
    if (ens->xh_valid) t->Xh.alias(ens->Xh), t->xh_valid = true, t->xh_scale = ens->xh_scale;
    t->g_max = ens->g_max;
    t->head = ens->head ? ens->head : ens;
    t->last_candidates = ens->last_candidates, t->last_similar = ens->last_similar;
    FC_TRY(t->counters.reserve(kCounters * sizeof(uint64_t)));
    ens->twin = t.release();
  }
  *out = ens->twin;
  return FC_OK;
}

// rows of np.stack(np.meshgrid(*arrays), -1).reshape(-1, T) (firecode/utils.py:219-221): with the default
// 'xy' indexing array #2 varies slowest, then #1, then #3 ... #T (fastest).  Written row by row, once, on
// host threads: the NumPy expression makes T strided passes over the whole output (1.0-1.3 s for the
// 1 679 616 x 8 grid of cfg3 -- five times the GPU pipeline it feeds).
template <class V>
static void cartesian_rows(const V *values, const int64_t *counts, int64_t T, V *out) {
  std::vector<int64_t> first((size_t)T, 0);  // offset of array t in `values`
  for (int64_t t = 1; t < T; ++t) first[(size_t)t] = first[(size_t)t - 1] + counts[t - 1];
  // digit order, slowest first: 1, 0, 2, 3, ... (T == 1: just 0)
  std::vector<int64_t> ord;
  if (T >= 2) ord = {1, 0};
  else ord = {0};
  for (int64_t t = 2; t < T; ++t) ord.push_back(t);
  int64_t rows = 1;
  for (int64_t t = 0; t < T; ++t) rows *= counts[t];
  if (rows == 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  const unsigned nthreads = (unsigned)std::min<int64_t>(std::min<unsigned>(hw, 16u), std::max<int64_t>(1, rows / 65536));
  auto fill = [&](int64_t r0, int64_t r1) {
    std::vector<int64_t> digit((size_t)T, 0);
    int64_t rem = r0;
    for (int64_t p = T - 1; p >= 0; --p) {  // mixed-radix digits of r0 in the order `ord`
      const int64_t t = ord[(size_t)p];
      digit[(size_t)t] = rem % counts[t];
      rem /= counts[t];
    }
    for (int64_t r = r0; r < r1; ++r) {
      V *row = out + r * T;
      for (int64_t t = 0; t < T; ++t) row[t] = values[first[(size_t)t] + digit[(size_t)t]];
      for (int64_t p = T - 1; p >= 0; --p) {  // + 1
        const int64_t t = ord[(size_t)p];
        if (++digit[(size_t)t] < counts[t]) break;
        digit[(size_t)t] = 0;
      }
    }
  };
  if (nthreads <= 1) {
    fill(0, rows);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (rows + nthreads - 1) / nthreads;
  for (unsigned k = 0; k < nthreads; ++k) {
    const int64_t r0 = (int64_t)k * per, r1 = std::min<int64_t>(rows, r0 + per);
    if (r0 < r1) pool.emplace_back(fill, r0, r1);
  }
  for (auto &th : pool) th.join();
}

}  // namespace fc

using namespace fc;

// =============================================================================
extern "C" {

int fc_abi_version(void) { return 1; }

int fc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fc_init(int device) {
  FC_API_LOCK;
  return do_init(device);
}

int fc_shutdown(void) {
  FC_API_LOCK;
  comm_teardown();
  context_teardown();
  return FC_OK;
}

const char *fc_last_error(void) { return last_error().c_str(); }

int fc_warmup(void) {
  FC_API_LOCK;
  FC_TRY(ensure_init());
  FC_TRY(warm_clash());
  FC_TRY(warm_embed());
  FC_TRY(warm_embed3());
  FC_TRY(warm_torsion());
  FC_TRY(warm_prune());
  FC_TRY(warm_h2_check());
  FC_TRY(warm_kabsch());
  FC_TRY(warm_tfd_ladder());
  FC_TRY(side_streams());  // the pipelines' streams and ordering events
  // the buffers a first large call would otherwise take from the runtime one by one (0.2 - 1 ms each): through the pool once
  {
    DevBuf warm[6];
    for (DevBuf &b : warm) FC_TRY(b.reserve((size_t)64 << 20));
  }
  return sync();
}

int fc_memory_trim(void) {
  FC_API_LOCK;
  if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
  pool_trim();
  return FC_OK;
}

int fc_host_alloc_pinned(int64_t bytes, void **out) {
  FC_API_LOCK;
  FC_REQUIRE(out != nullptr && bytes >= 0, "bad arguments");
  *out = nullptr;
  if (bytes == 0) return FC_OK;
  FC_TRY(ensure_init());
  void *p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return set_error(FC_E_NOMEM, "hipHostMalloc of %lld bytes failed", (long long)bytes);
  }
  *out = p;
  return FC_OK;
}

int fc_host_free_pinned(void *p) {
  FC_API_LOCK;
  if (p == nullptr) return FC_OK;
  if (hipHostFree(p) != hipSuccess) return set_error(FC_E_HIP, "hipHostFree failed: %s", hipGetErrorString(hipGetLastError()));
  return FC_OK;
}

int fc_stream_set(void *hip_stream) {
  FC_API_LOCK;
  FC_TRY(ensure_init());
  Context &c = ctx();
  FC_HIP_TRY(hipStreamSynchronize(c.stream));  // nothing of ours may still be queued on the old one
  c.stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c.own_stream;
  return FC_OK;
}

int fc_stream_use(void *hip_stream) {
  FC_API_LOCK;
  FC_TRY(ensure_init());
  Context &c = ctx();
  c.stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c.own_stream;
  return FC_OK;
}

int fc_device_info(char *name, int64_t name_len, int64_t *n_cu, int64_t *hbm_bytes) {
  FC_API_LOCK;
  FC_TRY(ensure_init());
  if (name && name_len > 0) std::snprintf(name, (size_t)name_len, "%s", ctx().name);
  if (n_cu) *n_cu = ctx().n_cu;
  if (hbm_bytes) *hbm_bytes = (int64_t)ctx().hbm;
  return FC_OK;
}

// ---- ensemble ------------------------------------------------------------------
int fc_ensemble_create(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask,
                       int center, fc_ensemble **out) {
  FC_API_LOCK;
  FC_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  FC_REQUIRE(coords != nullptr || N == 0, "coords is NULL");
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape N=%lld A=%lld", (long long)N, (long long)A);
  FC_REQUIRE(A <= 32767, "A=%lld exceeds 32767 atoms", (long long)A);
  FC_TRY(ensure_init());
  std::unique_ptr<fc_ensemble> e(new (std::nothrow) fc_ensemble);
  if (!e) return set_error(FC_E_NOMEM, "host allocation failed");
  e->epoch = ctx().epoch;
  FC_TRY(ensemble_build(coords, N, A, atom_mask, center, e.get()));
  *out = e.release();
  return FC_OK;
}

int fc_ensemble_destroy(fc_ensemble *ens) {
  FC_API_LOCK;
  delete ens;
  return FC_OK;
}

int fc_ensemble_shape(const fc_ensemble *ens, int64_t *N, int64_t *A_selected) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  if (N) *N = ens->N;
  if (A_selected) *A_selected = ens->A;
  return FC_OK;
}

// ---- a4 ------------------------------------------------------------------------
int fc_ensemble_rmsd_pairs(fc_ensemble *ens, const int64_t *pair_i, const int64_t *pair_j,
                           int64_t P, double *rmsd_out, double *maxdev_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  FC_REQUIRE(ens->epoch == ctx().epoch, "this ensemble was created before fc_shutdown / a device switch: create it again");
  FC_REQUIRE(P >= 0, "P < 0");
  if (P == 0) return FC_OK;
  FC_REQUIRE(pair_i && pair_j && rmsd_out && maxdev_out, "NULL pointer argument");
  for (int64_t k = 0; k < P; ++k)
    FC_REQUIRE(pair_i[k] >= 0 && pair_i[k] < ens->N && pair_j[k] >= 0 && pair_j[k] < ens->N,
               "pair %lld = (%lld, %lld) out of range [0, %lld)", (long long)k,
               (long long)pair_i[k], (long long)pair_j[k], (long long)ens->N);
  FC_TRY(ensure_init());
  DevBuf di, dj, dr, dm;
  FC_TRY(upload(di, pair_i, (size_t)P));
  FC_TRY(upload(dj, pair_j, (size_t)P));
  FC_TRY(dr.reserve((size_t)P * sizeof(double)));
  FC_TRY(dm.reserve((size_t)P * sizeof(double)));
  FC_TRY(launch_pairs_exact(ens, di.as<int64_t>(), dj.as<int64_t>(), P, dr.as<double>(),
                            dm.as<double>(), nullptr));
  FC_TRY(d2h(rmsd_out, dr.p, (size_t)P * sizeof(double)));
  FC_TRY(d2h(maxdev_out, dm.p, (size_t)P * sizeof(double)));
  return sync();
}

int fc_kabsch_rmsd_pairs(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask,
                         const int64_t *pair_i, const int64_t *pair_j, int64_t P, int center,
                         double *rmsd_out, double *maxdev_out) {
  FC_API_LOCK;
  fc_ensemble *e = nullptr;
  FC_TRY(fc_ensemble_create(coords, N, A, atom_mask, center, &e));
  const int rc = fc_ensemble_rmsd_pairs(e, pair_i, pair_j, P, rmsd_out, maxdev_out);
  fc_ensemble_destroy(e);
  return rc;
}

// all pairs, both outputs: covariance tiles on the fp64 matrix pipe, rotation + explicit rotated
// difference in the epilogue (k_simbits_screen_mfma<.., 2>); structures beyond the LDS column tile
// (A > 104) take the one-wave-per-row kernel.  Outputs may both be NULL (timing only).
static bool rmsd_and_max_tiled(const fc_ensemble *ens) {
  // (a 64-column tile up to 104 atoms, 32 columns up to 208, 16 up to 416: launch_rmsd_values picks)
  const size_t lds_m = ((size_t)(((ens->A + 3) / 4 + 1) / 2) * 384 + 16 + 128) * sizeof(double) + 1024;
  return lds_m <= (size_t)160 * 1024 && (uint64_t)((ens->A + 3) / 4 * 4) * 3 * (uint64_t)ens->Npad < (1ull << 32);
}

static int rmsd_and_max_all(fc_ensemble *ens, double *rmsd_out, double *maxdev_out, double *ms_kernel) {
  FC_TRY(ensure_init());
  const int64_t N = ens->N;
  if (N == 0) return FC_OK;
  Context &c = ctx();
  DevBuf dr, dm;
  const size_t bytes = (size_t)N * N * sizeof(double);
  FC_TRY(dr.reserve(bytes));
  FC_TRY(dm.reserve(bytes));
  const bool tiled = rmsd_and_max_tiled(ens);
  unsigned long long cnt[16] = {0};
  if (tiled) {
    // the tiled kernel writes every (i, j >= i), exact zeros on the diagonal; the lower triangle is
    // mirrored on the device below: nothing to clear (2 x 800 MB of memset per call at 10^4 conformers)
    FC_TRY(ensemble_shard(ens, 0, 1, 256));  // sizes the pair queue of the fix-up
    FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, 16 * sizeof(uint64_t), c.stream));
  } else {
    FC_HIP_TRY(hipMemsetAsync(dr.p, 0, bytes, c.stream));
    FC_HIP_TRY(hipMemsetAsync(dm.p, 0, bytes, c.stream));
  }
  FC_HIP_TRY(hipEventRecord(c.ev0, c.stream));
  if (tiled) FC_TRY(launch_rmsd_values(ens, 0.0, dr.as<double>(), dm.as<double>()));
  else FC_TRY(launch_matrix_exact(ens, dr.as<double>(), dm.as<double>()));
  FC_HIP_TRY(hipEventRecord(c.ev1, c.stream));
  if (tiled) FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  FC_TRY(sync());
  if (ms_kernel) {
    float ms = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *ms_kernel = ms;
  }
  if (tiled && cnt[6] > (unsigned long long)ens->pairq_cap) {
    // more pairs for the fix-up than its queue holds.  Near-duplicates first (the eigenvalue form of the rmsd declines pairs
    // closer than ~1e-3 A): the same kernel with the running sum
    FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, 16 * sizeof(uint64_t), c.stream));
    FC_TRY(launch_rmsd_values(ens, 0.0, dr.as<double>(), dm.as<double>(), 0, 1, /*explicit_sum=*/true));
    FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
    FC_TRY(sync());
  }
  if (tiled && cnt[6] > (unsigned long long)ens->pairq_cap) {
    // more degenerate pairs than the fix-up queue holds (planar or collinear structures: every pair): the plain kernel
    // redoes the matrix
    FC_HIP_TRY(hipMemsetAsync(dr.p, 0, bytes, c.stream));
    FC_HIP_TRY(hipMemsetAsync(dm.p, 0, bytes, c.stream));
    FC_TRY(launch_matrix_exact(ens, dr.as<double>(), dm.as<double>()));
  }
  // both kernels write the upper triangle and the diagonal: the lower one is mirrored on the device, then the matrices
  // travel (the host's element loop took ~0.1 s per call at 10^4 conformers)
  if (rmsd_out) FC_TRY(launch_mirror_upper(dr.as<double>(), N));
  if (maxdev_out) FC_TRY(launch_mirror_upper(dm.as<double>(), N));
  if (rmsd_out) FC_TRY(d2h(rmsd_out, dr.p, bytes));
  if (maxdev_out) FC_TRY(d2h(maxdev_out, dm.p, bytes));
  return sync();
}

// bench hook: `reps` complete all-pairs alignment passes over the resident ensemble, enqueued back to
// back on the library's stream (outputs: two dense (N, N) matrices that stay in HBM), one host wait.
// ms_kernel_mean: HIP events around the dominant kernel (k_simbits_screen_mfma<., 2>) of every launch;
// ms_total: first launch to the end of the last fix-up kernel.  stats[0] = pairs per pass,
// stats[1] = pairs the last pass queued for the Jacobi fix-up, stats[2] = 1 when the tiled kernel ran.
// sample_*: P elements (i, j) of the LAST pass's two output matrices, read back for the caller's checker
static int bench_rmsd_and_max_all(fc_ensemble *ens, int64_t reps, double *ms_kernel_mean, double *ms_total,
                                  int64_t *stats, const int64_t *sample_i = nullptr, const int64_t *sample_j = nullptr,
                                  int64_t P = 0, double *sample_rmsd = nullptr, double *sample_maxdev = nullptr) {
  FC_TRY(ensure_init());
  const int64_t N = ens->N;
  FC_REQUIRE(N >= 2, "needs at least two conformers");
  Context &c = ctx();
  DevBuf dr, dm;
  const size_t bytes = (size_t)N * N * sizeof(double);
  FC_TRY(dr.reserve(bytes));
  FC_TRY(dm.reserve(bytes));
  const bool tiled = rmsd_and_max_tiled(ens);
  // under a communicator (or the loopback hook) a rank computes the rows dealt to it: the units shard with
  // no exchange; every rank keeps its rows of the two matrices
  const int64_t rank = comm_rank(), world = comm_world();
  FC_REQUIRE(tiled || world == 1, "structures beyond the tiled kernel are not sharded");
  if (tiled) FC_TRY(ensemble_shard(ens, 0, 1, 256));
  std::vector<hipEvent_t> &ev = c.ev_pool;
  while ((int64_t)ev.size() < 2 * reps + 2) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreate(&e));
    ev.push_back(e);
  }
  FC_HIP_TRY(hipEventRecord(ev[2 * reps], c.stream));
  for (int64_t r = 0; r < reps; ++r) {
    if (tiled) {
      FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, 16 * sizeof(uint64_t), c.stream));
      FC_HIP_TRY(hipEventRecord(ev[2 * r], c.stream));
      c.mark_after_screen = ev[2 * r + 1];  // recorded right behind the tiled kernel, in front of the fix-up
      const int rc = launch_rmsd_values(ens, 0.0, dr.as<double>(), dm.as<double>(), rank, world);
      c.mark_after_screen = nullptr;
      FC_TRY(rc);
    } else {
      FC_HIP_TRY(hipEventRecord(ev[2 * r], c.stream));
      FC_TRY(launch_matrix_exact(ens, dr.as<double>(), dm.as<double>()));
      FC_HIP_TRY(hipEventRecord(ev[2 * r + 1], c.stream));
    }
  }
  FC_HIP_TRY(hipEventRecord(ev[2 * reps + 1], c.stream));
  unsigned long long cnt[16] = {0};
  if (tiled) FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  FC_TRY(sync());
  double sum = 0.0;
  for (int64_t r = 0; r < reps; ++r) {
    float ms = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
    sum += ms;
  }
  float tot = 0.f;
  FC_HIP_TRY(hipEventElapsedTime(&tot, ev[2 * reps], ev[2 * reps + 1]));
  if (ms_kernel_mean) *ms_kernel_mean = sum / (double)reps;
  if (ms_total) *ms_total = tot;
  if (stats) {
    // pairs this rank computed: rows of its blocks of 128, columns right of the diagonal
    int64_t own = 0;
    const int64_t nb = ceil_div(N, (int64_t)128);
    for (int64_t lb = 0, b; (b = global_block(lb, rank, world)) < nb; ++lb)
      for (int64_t i = b * 128; i < std::min<int64_t>(N, (b + 1) * 128); ++i) own += N - 1 - i;
    stats[0] = own;
    stats[1] = (int64_t)cnt[6];
    stats[2] = tiled ? 1 : 0;
  }
  if (tiled && cnt[6] > (unsigned long long)ens->pairq_cap)
    return set_error(FC_E_LIMIT, "%llu degenerate pairs exceed the fix-up queue (%lld)", cnt[6], (long long)ens->pairq_cap);
  if (P > 0) {  // behind the timed passes and their events: what the last pass left in the two matrices
    DevBuf di, dj, sr, sm;
    FC_TRY(upload(di, sample_i, (size_t)P));
    FC_TRY(upload(dj, sample_j, (size_t)P));
    FC_TRY(sr.reserve((size_t)P * sizeof(double)));
    FC_TRY(sm.reserve((size_t)P * sizeof(double)));
    FC_TRY(launch_gather_matrix_pairs(dr.as<double>(), dm.as<double>(), N, di.as<int64_t>(), dj.as<int64_t>(), P,
                                      sr.as<double>(), sm.as<double>()));
    FC_TRY(d2h(sample_rmsd, sr.p, (size_t)P * sizeof(double)));
    FC_TRY(d2h(sample_maxdev, sm.p, (size_t)P * sizeof(double)));
    FC_TRY(sync());
  }
  return FC_OK;
}

int fc_ensemble_rmsd_matrix(fc_ensemble *ens, double *rmsd_out, double *maxdev_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && rmsd_out && maxdev_out, "NULL pointer argument");
  return rmsd_and_max_all(ens, rmsd_out, maxdev_out, nullptr);
}

int fc_ensemble_rmsd_and_max_all(fc_ensemble *ens, double *rmsd_out, double *maxdev_out, double *ms_kernel) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  return rmsd_and_max_all(ens, rmsd_out, maxdev_out, ms_kernel);
}

int fc_bench_rmsd_and_max_all(fc_ensemble *ens, int64_t reps, double *ms_kernel_mean, double *ms_total,
                              int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr && reps >= 1 && reps <= 4096, "bad arguments");
  return bench_rmsd_and_max_all(ens, reps, ms_kernel_mean, ms_total, stats);
}

int fc_bench_rmsd_and_max_all_sampled(fc_ensemble *ens, int64_t reps, const int64_t *pair_i, const int64_t *pair_j,
                                      int64_t P, double *rmsd_out, double *maxdev_out, double *ms_kernel_mean,
                                      double *ms_total, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr && reps >= 1 && reps <= 4096, "bad arguments");
  FC_REQUIRE(P >= 0 && (P == 0 || (pair_i && pair_j && rmsd_out && maxdev_out)), "NULL sample arrays");
  for (int64_t p = 0; p < P; ++p)
    FC_REQUIRE(pair_i[p] >= 0 && pair_i[p] < ens->N && pair_j[p] >= 0 && pair_j[p] < ens->N,
               "sample pair %lld out of range", (long long)p);
  return bench_rmsd_and_max_all(ens, reps, ms_kernel_mean, ms_total, stats, pair_i, pair_j, P, rmsd_out, maxdev_out);
}

int fc_ensemble_rmsd_values(fc_ensemble *ens, double *rmsd_out, double *ms_kernel) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  FC_TRY(ensure_init());
  const int64_t N = ens->N;
  if (N == 0) return FC_OK;
  FC_TRY(ensemble_shard(ens, 0, 1, 256));  // sizes the pair queue
  Context &c = ctx();
  DevBuf dr;
  const size_t bytes = (size_t)N * N * sizeof(double);
  FC_TRY(dr.reserve(bytes));
  FC_HIP_TRY(hipMemsetAsync(dr.p, 0, bytes, c.stream));
  FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, 16 * sizeof(uint64_t), c.stream));
  FC_HIP_TRY(hipEventRecord(c.ev0, c.stream));
  FC_TRY(launch_rmsd_values(ens, 0.02, dr.as<double>(), nullptr));
  FC_HIP_TRY(hipEventRecord(c.ev1, c.stream));
  unsigned long long cnt[16];
  FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  if (rmsd_out) {  // (the lower triangle mirrored on the device, then one copy)
    FC_TRY(launch_mirror_upper(dr.as<double>(), N));
    FC_TRY(d2h(rmsd_out, dr.p, bytes));
  }
  FC_TRY(sync());
  if (cnt[6] > (unsigned long long)ens->pairq_cap)
    return set_error(FC_E_LIMIT, "%llu pairs closer than 0.02 A exceed the fix-up queue (%lld): "
                     "use fc_ensemble_rmsd_matrix", cnt[6], (long long)ens->pairq_cap);
  if (ms_kernel) {
    float ms = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *ms_kernel = ms;
  }
  return FC_OK;
}

int fc_alignment_matrices(const double *p, const double *q, int64_t n_pairs, int64_t A,
                          double *M_out) {
  FC_API_LOCK;
  FC_REQUIRE(n_pairs >= 0 && A >= 1, "bad shape");
  if (n_pairs == 0) return FC_OK;
  FC_REQUIRE(p && q && M_out, "NULL pointer argument");
  FC_TRY(ensure_init());
  DevBuf dp, dq, dM;
  FC_TRY(upload(dp, p, (size_t)n_pairs * A * 3));
  FC_TRY(upload(dq, q, (size_t)n_pairs * A * 3));
  FC_TRY(dM.reserve((size_t)n_pairs * 9 * sizeof(double)));
  FC_TRY(launch_alignment_matrices(dp.as<double>(), dq.as<double>(), n_pairs, A, dM.as<double>()));
  FC_TRY(d2h(M_out, dM.p, (size_t)n_pairs * 9 * sizeof(double)));
  return sync();
}

// ---- a9: align_by_moi (firecode/hypermolecule_class.py:45-86) ------------------------------
int fc_align_by_moi(const double *coords, int64_t N, int64_t A, const double *masses, double *out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && masses && out, "NULL pointer argument");
  FC_TRY(ensure_init());
  DevBuf dc, dm, dcen, dmom, dP, dQ, dM, dt, dout;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(dm, masses, (size_t)A));
  FC_TRY(dcen.reserve((size_t)N * A * 3 * sizeof(double)));
  FC_TRY(dmom.reserve((size_t)N * 3 * sizeof(double)));
  FC_TRY(dP.reserve((size_t)N * 9 * sizeof(double)));
  FC_TRY(dQ.reserve((size_t)N * 9 * sizeof(double)));
  FC_TRY(dM.reserve((size_t)N * 9 * sizeof(double)));
  FC_TRY(dt.reserve((size_t)N * 3 * sizeof(double)));
  FC_TRY(dout.reserve((size_t)N * A * 3 * sizeof(double)));
  FC_TRY(launch_center_structures(dc.as<double>(), N, A, dcen.as<double>()));
  FC_TRY(launch_inertia_moments(dcen.as<double>(), N, A, dm.as<double>(), dmom.as<double>()));
  FC_TRY(launch_moi_diag_pairs(dmom.as<double>(), N, dP.as<double>(), dQ.as<double>()));
  FC_TRY(launch_alignment_matrices(dP.as<double>(), dQ.as<double>(), N, 3, dM.as<double>()));
  FC_TRY(launch_set_identity(dM.as<double>()));
  FC_HIP_TRY(hipMemsetAsync(dt.p, 0, (size_t)N * 3 * sizeof(double), ctx().stream));
  FC_TRY(launch_rototranslate(dcen.as<double>(), N, A, dM.as<double>(), dt.as<double>(), dout.as<double>()));
  FC_TRY(d2h(out, dout.p, (size_t)N * A * 3 * sizeof(double)));
  return sync();
}

// ---- a5 ------------------------------------------------------------------------
int fc_rmsd_simbits(fc_ensemble *ens, double max_rmsd, double max_dev, const double *energies,
                    double max_dE, int64_t row_begin, int64_t row_end, uint64_t *bits_out,
                    int64_t *n_grey) {
  FC_API_LOCK;
  FC_REQUIRE(ens && bits_out, "NULL pointer argument");
  FC_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= ens->N, "bad row range");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_TRY(ensure_init());
  if (ens->N == 0) return FC_OK;
  FC_TRY(ensemble_shard(ens, 0, 1, default_row_block()));
  FC_TRY(simbits_local(ens, max_rmsd, max_dev, energies, max_dE, true));
  const int64_t W = ens->W;
  std::vector<uint64_t> all((size_t)ens->rows_local * W);
  unsigned long long cnt[8];
  FC_TRY(d2h(all.data(), ens->bits.p, all.size() * sizeof(uint64_t)));
  FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  FC_TRY(sync());
  note_candidates(ens, cnt[6], cnt[2]);
  // words at or below the diagonal were never produced: define them as 0
  for (int64_t i = row_begin; i < row_end; ++i)
    for (int64_t w = 0; w < W; ++w)
      bits_out[(i - row_begin) * W + w] = (w * 64 + 63 > i) ? all[(size_t)i * W + w] : 0ull;
  if (n_grey) *n_grey = (int64_t)cnt[3];
  return FC_OK;
}

int fc_prune_rmsd(fc_ensemble *ens, double max_rmsd, double max_dev, const double *energies,
                  double max_dE, int64_t min_per_group, uint8_t *mask_out, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && mask_out, "NULL pointer argument");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_REQUIRE(min_per_group >= 1, "min_per_group must be >= 1");
  FC_TRY(ensure_init());
  if (ens->N == 0) return FC_OK;
  FC_TRY(ensemble_shard(ens, 0, 1, default_row_block()));
  // lean first: only the pair lists (no bit matrix); the pair ladder declines when the candidate
  // queue overflowed or the list is too long for it -- then the same prune again with the bits
  FC_TRY(simbits_local(ens, max_rmsd, max_dev, energies, max_dE, true, /*lean=*/true));
  unsigned long long cnt[8];
  int64_t levels = 0, survivors = 0;
  int rc = ladder_single(ens, nullptr, min_per_group, mask_out, &levels, &survivors, cnt,
                         ens->simq.as<uint64_t>(), false, true);
  if (rc == FC_E_LIMIT) {
    FC_TRY(simbits_local(ens, max_rmsd, max_dev, energies, max_dE, true, /*lean=*/false));
    rc = ladder_single(ens, ens->bits.as<uint64_t>(), min_per_group, mask_out, &levels, &survivors, cnt,
                       ens->simq.as<uint64_t>(), false, true);
  }
  FC_TRY(rc);
  note_candidates(ens, cnt[6], cnt[2]);
  if (stats) {
    stats[0] = ens->N * (ens->N - 1) / 2;
    stats[1] = (int64_t)cnt[1];
    stats[2] = (int64_t)cnt[2];
    stats[3] = (int64_t)cnt[3];
    stats[4] = levels;
    stats[5] = survivors;
  }
  return FC_OK;
}

// prune_by_rmsd(host arrays) as ONE call (firecode/ensemble.py:230-235, firecode/embedder.py:1472-1474): upload, preparation,
// prune, mask -- what a caller got from fc_ensemble_create + fc_prune_rmsd + fc_ensemble_destroy, under one lock and with
// one crossing of the language boundary
int fc_prune_rmsd_host(const double *coords, int64_t N, int64_t A, const uint8_t *atom_mask, int center, double max_rmsd,
                       double max_dev, const double *energies, double max_dE, int64_t min_per_group, uint8_t *mask_out,
                       int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape N=%lld A=%lld", (long long)N, (long long)A);
  FC_REQUIRE(A <= 32767, "A=%lld exceeds 32767 atoms", (long long)A);
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_REQUIRE(min_per_group >= 1, "min_per_group must be >= 1");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords != nullptr && mask_out != nullptr, "NULL pointer argument");
  FC_TRY(ensure_init());
  fc_ensemble e;
  e.epoch = ctx().epoch;
  // the last piece's DMA and the preparation kernel are still running when ensemble_build returns: the prune's reserves and
  // the screen's item table (host work + one small copy) go under them instead of behind the wait for the largest G
  // a small ensemble: the largest G from the caller's array, no copy of the device's and no wait for it (the counters are
  // reset in front of the screen)
  const bool host_gmax = N * A * 3 <= kHostGmaxDoubles;
  int rc = ensemble_build(coords, N, A, atom_mask, center, &e, /*defer_wait=*/true, host_gmax);
  if (rc == FC_OK && host_gmax) {
    const double g = host_largest_g(coords, N, A, e.sel_host, center);
    e.g_max = std::isfinite(g) ? g * (1.0 + 1e-12) : g;  // (not finite: every screen that needs it declines, as with the device's)
  }
  if (rc == FC_OK) rc = ensemble_shard(&e, 0, 1, default_row_block());
  if (rc == FC_OK) rc = prebuild_screen_items(&e);
  const int rc_fin = ensemble_build_finish(&e);  // (always: nothing of `e` may be in flight when it goes out of scope)
  if (rc != FC_OK || rc_fin != FC_OK) {
    (void)hipStreamSynchronize(cur_stream());
    return rc != FC_OK ? rc : rc_fin;
  }
  return fc_prune_rmsd(&e, max_rmsd, max_dev, energies, max_dE, min_per_group, mask_out, stats);
}

int fc_greedy_prune_from_bits(const uint64_t *bits, int64_t N, int64_t min_per_group,
                              uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && min_per_group >= 1, "bad arguments");
  if (N == 0) return FC_OK;
  FC_REQUIRE(bits && mask_out, "NULL pointer argument");
  FC_TRY(ensure_init());
  fc_ensemble e;
  e.N = N;
  e.Npad = ceil_div(N, 64) * 64;
  e.W = e.Npad / 64;
  e.row_block = 64;
  FC_TRY(e.maskA.reserve((size_t)e.Npad));
  FC_TRY(e.maskB.reserve((size_t)e.Npad));
  FC_TRY(e.mbits.reserve((size_t)e.W * sizeof(uint64_t)));
  FC_TRY(e.counters.reserve(kCounters * sizeof(uint64_t)));
  // rows padded to a multiple of the row block so k_level's local row == global row
  const int64_t rows = ceil_div(N, e.row_block) * e.row_block;
  FC_TRY(e.bits.reserve((size_t)rows * e.W * sizeof(uint64_t)));
  FC_TRY(h2d(e.bits.p, bits, (size_t)N * e.W * sizeof(uint64_t)));
  return ladder_single(&e, e.bits.as<uint64_t>(), min_per_group, mask_out, nullptr, nullptr);
}

int fc_prune_rmsd_begin(fc_ensemble *ens, double max_rmsd, double max_dev, const double *energies,
                        double max_dE, int64_t rank, int64_t world, int64_t row_block,
                        int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_TRY(ensure_init());
  FC_TRY(ensemble_shard(ens, rank, world, row_block));
  if (ens->N == 0) return FC_OK;
  FC_TRY(simbits_local(ens, max_rmsd, max_dev, energies, max_dE, true));
  unsigned long long cnt[8];
  FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  FC_TRY(sync());
  if (stats) {
    stats[0] = 0;  // pairs owned by this rank: rows i owned, columns j > i
    const int64_t nb = ceil_div(ens->N, row_block);
    for (int64_t lb = 0, b = rank; (b = global_block(lb, rank, world)) < nb; ++lb)
      for (int64_t i = b * row_block; i < std::min(ens->N, (b + 1) * row_block); ++i)
        stats[0] += ens->N - 1 - i;
    stats[1] = (int64_t)cnt[1];
    stats[2] = (int64_t)cnt[2];
    stats[3] = (int64_t)cnt[3];
    stats[4] = last_screen_ns();  // screen-kernel duration (ns) of this rank
    stats[5] = 0;
  }
  return FC_OK;
}

int fc_prune_level(fc_ensemble *ens, int64_t k, const uint8_t *mask_in, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && mask_in && mask_out, "NULL pointer argument");
  FC_REQUIRE(ens->bits_valid && !ens->lean, "fc_prune_rmsd_begin has not been called on this ensemble");
  FC_REQUIRE(k >= 1, "k must be >= 1");
  FC_TRY(ensure_init());
  const int64_t N = ens->N;
  if (N == 0) return FC_OK;
  auto *cnt = reinterpret_cast<unsigned long long *>(ens->counters.p);
  FC_TRY(h2d(ens->maskA.p, mask_in, (size_t)N));
  FC_HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint64_t), ctx().stream));
  FC_TRY(launch_pack_mask(ens->maskA.as<uint8_t>(), N, ens->mbits.as<uint64_t>(), ens->W, cnt));
  FC_TRY(launch_copy_bytes(ens->maskA.as<uint8_t>(), ens->maskB.as<uint8_t>(), N));
  FC_TRY(launch_level(ens->bits.as<uint64_t>(), ens->W, ens->mbits.as<uint64_t>(),
                      ens->maskA.as<uint8_t>(), ens->maskB.as<uint8_t>(), N, k, ens->row_block,
                      ens->rank, ens->world, ens->rows_local));
  FC_TRY(d2h(mask_out, ens->maskB.p, (size_t)N));
  return sync();
}

int fc_prune_similar_pairs(fc_ensemble *ens, uint64_t *pairs_out, int64_t capacity, int64_t *n_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && n_out, "NULL pointer argument");
  FC_REQUIRE(ens->bits_valid, "fc_prune_rmsd_begin has not been called on this ensemble");
  FC_TRY(ensure_init());
  *n_out = 0;
  if (ens->N == 0) return FC_OK;
  unsigned long long cnt[8];
  FC_TRY(d2h(cnt, ens->counters.p, sizeof cnt));
  FC_TRY(sync());
  if ((int64_t)cnt[6] > ens->pairq_cap)
    return set_error(FC_E_LIMIT, "candidate queue overflow (%llu > %lld): similar-pair list unavailable",
                     cnt[6], (long long)ens->pairq_cap);
  *n_out = (int64_t)cnt[2];
  if (pairs_out == nullptr) return FC_OK;  // size query
  FC_REQUIRE(capacity >= (int64_t)cnt[2], "pairs_out holds %lld entries, %llu needed", (long long)capacity, cnt[2]);
  FC_TRY(d2h(pairs_out, ens->simq.p, (size_t)cnt[2] * sizeof(uint64_t)));
  return sync();
}

int fc_prune_from_pairs(fc_ensemble *ens, const uint64_t *pairs, int64_t n_pairs,
                        int64_t min_per_group, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && mask_out && (pairs || n_pairs == 0), "NULL pointer argument");
  FC_REQUIRE(n_pairs >= 0 && min_per_group >= 1, "bad arguments");
  FC_TRY(ensure_init());
  const int64_t N = ens->N, W = ens->W;
  if (N == 0) return FC_OK;
  FC_TRY(ens->counters.reserve(kCounters * sizeof(uint64_t)));
  auto *cnt = reinterpret_cast<unsigned long long *>(ens->counters.p);
  DevBuf dp;
  FC_TRY(upload(dp, pairs, (size_t)n_pairs));
  const unsigned long long np = (unsigned long long)n_pairs;
  FC_TRY(h2d(cnt + 2, &np, sizeof np));  // k_ladder_pairs reads the list length from counters[2]
  const bool sparse = np <= kPairLadderCap && (size_t)2 * W * sizeof(uint64_t) <= 60 * 1024;
  if (sparse)  // one launch over the pair list; no bit matrix needed
    return ladder_single(ens, nullptr, min_per_group, mask_out, nullptr, nullptr, nullptr,
                         dp.as<uint64_t>(), true);
  // dense: rebuild the whole bit matrix (rows padded like ladder_single expects) and run the levels
  const int64_t rb = ens->row_block > 0 ? ens->row_block : 64;
  const size_t bytes = (size_t)(ceil_div(N, rb) * rb) * W * sizeof(uint64_t);
  FC_TRY(ens->bits_full.reserve(bytes));
  FC_HIP_TRY(hipMemsetAsync(ens->bits_full.p, 0, bytes, ctx().stream));
  FC_TRY(launch_scatter_pairs(dp.as<uint64_t>(), n_pairs, N, W, ens->bits_full.as<uint64_t>()));
  return ladder_single(ens, ens->bits_full.as<uint64_t>(), min_per_group, mask_out, nullptr, nullptr);
}

// ---- device-resident exchange (no host round trip between screen and mask) --------------
int fc_prune_rmsd_begin_async(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t rank,
                              int64_t world, int64_t row_block) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_TRY(ensure_init());
  FC_TRY(ensemble_shard(ens, rank, world, row_block));
  if (ens->N == 0) return FC_OK;
  return simbits_local(ens, max_rmsd, max_dev, nullptr, 0.0, true, /*lean=*/true);  // consumer: the exported pair list
}

int fc_ensemble_twin(fc_ensemble *ens, fc_ensemble **twin_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && twin_out, "NULL pointer argument");
  FC_TRY(ensure_init());
  return ensemble_twin(ens, twin_out);
}

// counters reset on the current stream (behind the last user of this workspace), the screen on
// `scr` (behind the previous screen), the refine back on the current stream; ev_a / ev_b (may be
// null): timing events recorded right around the main screen kernel
static int begin_split(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t rank, int64_t world,
                       int64_t row_block, hipStream_t scr, hipEvent_t ev_a, hipEvent_t ev_b) {
  FC_TRY(ensemble_shard(ens, rank, world, row_block));
  if (ens->N == 0) return FC_OK;
  Context &c = ctx();
  FC_TRY(side_streams());
  hipStream_t const tail = c.stream;
  struct Restore {
    Context &c;
    hipStream_t s;
    ~Restore() { c.stream = s; }
  } restore{c, tail};
  FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, kCounters * sizeof(uint64_t), tail));
  FC_HIP_TRY(hipEventRecord(c.ev_reset, tail));
  FC_HIP_TRY(hipStreamWaitEvent(scr, c.ev_reset, 0));
  ens->lean = true;  // consumer: the exported pair list
  c.stream = scr;
  if (ev_a) FC_HIP_TRY(hipEventRecord(ev_a, scr));  // the pair costs the stream ~14 us: not every step needs it
  c.mark_after_screen = ev_a ? ev_b : nullptr;
  const int rc_screen = launch_simbits_screen(ens, max_rmsd * max_rmsd + kScreenMargin);
  c.mark_after_screen = nullptr;
  FC_TRY(rc_screen);
  FC_HIP_TRY(hipEventRecord(c.ev_screened, scr));  // behind the verdict and the gated fp64 screen, too
  FC_HIP_TRY(hipStreamWaitEvent(tail, c.ev_screened, 0));
  c.stream = tail;
  FC_TRY(launch_simbits_refine(ens, max_rmsd, max_dev, nullptr, 0.0));
  ens->bits_valid = true;
  return FC_OK;
}

int fc_prune_rmsd_begin_split_async(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t rank,
                                    int64_t world, int64_t row_block, void *screen_stream, int timed) {
  FC_API_LOCK;
  FC_REQUIRE(ens != nullptr, "ens is NULL");
  FC_REQUIRE(screen_stream != nullptr, "screen_stream is NULL");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_TRY(ensure_init());
  return begin_split(ens, max_rmsd, max_dev, rank, world, row_block, static_cast<hipStream_t>(screen_stream),
                     timed ? ctx().ev2 : nullptr, timed ? ctx().ev3 : nullptr);
}

int fc_prune_export_pairs_dev(fc_ensemble *ens, uint64_t *dev_out, int64_t cap) {
  FC_API_LOCK;
  FC_REQUIRE(ens && dev_out, "NULL pointer argument");
  FC_REQUIRE(cap >= 0, "cap must be >= 0");
  FC_REQUIRE(ens->bits_valid, "fc_prune_rmsd_begin has not been called on this ensemble");
  FC_TRY(ensure_init());
  return launch_export_pairs(ens->simq.as<uint64_t>(),
                             reinterpret_cast<const unsigned long long *>(ens->counters.p),
                             (unsigned long long)ens->pairq_cap, cap, dev_out);
}

int fc_prune_from_gathered_dev(fc_ensemble *ens, const uint64_t *dev_gathered, int64_t world, int64_t cap,
                               int64_t min_per_group, uint8_t *mask_out, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && dev_gathered && mask_out, "NULL pointer argument");
  FC_REQUIRE(world >= 1 && world <= 64 && cap >= 0 && min_per_group >= 1, "bad arguments");
  FC_TRY(ensure_init());
  const int64_t N = ens->N, W = ens->W;
  if (N == 0) return FC_OK;
  if ((uint64_t)world * (uint64_t)cap > kPairLadderCap || (size_t)2 * W * sizeof(uint64_t) > 60 * 1024)
    return set_error(FC_E_LIMIT, "exchange too large for the one-launch ladder (world*cap = %lld, N = %lld)",
                     (long long)(world * cap), (long long)N);
  FC_TRY(ens->counters.reserve(kCounters * sizeof(uint64_t)));
  FC_TRY(ens->gathered.reserve((size_t)std::max<int64_t>(world * cap, 1) * sizeof(uint64_t)));
  auto *cnt = reinterpret_cast<unsigned long long *>(ens->counters.p);
  unsigned long long local[8];
  FC_TRY(pinned_reserve((size_t)(W + 32) * sizeof(uint64_t)));
  // this rank's own counters (candidates, similar, grey) before counters[2] becomes the global length
  unsigned long long *local_host = static_cast<unsigned long long *>(ctx().pinned) + W + 16;
  FC_TRY(d2h(local_host, cnt, 8 * sizeof(uint64_t)));
  FC_TRY(launch_compact_gathered(dev_gathered, (int)world, cap, ens->gathered.as<uint64_t>(), cnt));
  int64_t survivors = 0;
  const int rc = ladder_single(ens, nullptr, min_per_group, mask_out, nullptr, &survivors, nullptr,
                               ens->gathered.as<uint64_t>(), true);
  if (rc != FC_OK) return rc;  // FC_E_LIMIT: some rank's list was missing or longer than cap
  for (int k = 0; k < 8; ++k) local[k] = local_host[k];
  if (stats) {
    stats[0] = 0;
    const int64_t nb = ceil_div(N, ens->row_block);
    for (int64_t lb = 0, b; (b = global_block(lb, ens->rank, ens->world)) < nb; ++lb)
      for (int64_t i = b * ens->row_block; i < std::min(N, (b + 1) * ens->row_block); ++i)
        stats[0] += N - 1 - i;
    stats[1] = (int64_t)local[1];
    stats[2] = (int64_t)local[2];
    stats[3] = (int64_t)local[3];
    stats[4] = last_screen_ns();
    stats[5] = survivors;
  }
  return FC_OK;
}

// The same, stream-ordered: enqueue prune number `slot` of `n_slots` and return without waiting;
// fc_prune_collect waits for the stream and reads that prune's result.  Lets a caller keep the
// GPU busy across prunes (the next screen starts while the host is still in Python).
int fc_prune_from_gathered_dev_enqueue(fc_ensemble *ens, const uint64_t *dev_gathered, int64_t world,
                                       int64_t cap, int64_t min_per_group, int64_t slot, int64_t n_slots) {
  FC_API_LOCK;
  FC_REQUIRE(ens && dev_gathered, "NULL pointer argument");
  FC_REQUIRE(world >= 1 && world <= 64 && cap >= 0 && min_per_group >= 1, "bad arguments");
  FC_REQUIRE(n_slots >= 1 && n_slots <= 4096 && slot >= 0 && slot < n_slots, "bad slot %lld of %lld", (long long)slot,
             (long long)n_slots);
  FC_TRY(ensure_init());
  const int64_t N = ens->N, W = ens->W;
  FC_REQUIRE(N > 0, "empty ensemble");
  if ((uint64_t)world * (uint64_t)cap > kPairLadderCap || (size_t)2 * W * sizeof(uint64_t) > 60 * 1024)
    return set_error(FC_E_LIMIT, "exchange too large for the one-launch ladder (world*cap = %lld, N = %lld)",
                     (long long)(world * cap), (long long)N);
  FC_TRY(ens->counters.reserve(kCounters * sizeof(uint64_t)));
  FC_TRY(ens->gathered.reserve((size_t)std::max<int64_t>(world * cap, 1) * sizeof(uint64_t)));
  auto *cnt = reinterpret_cast<unsigned long long *>(ens->counters.p);
  // pinned layout: n_slots x (W + 16) ladder results, then n_slots x 8 local counters; sized by
  // the first prune of a batch (growing it later would move results that are still in flight)
  const size_t need = ((size_t)n_slots * (size_t)(W + 16) + (size_t)n_slots * 8) * sizeof(uint64_t);
  if (slot == 0) FC_TRY(pinned_reserve(need));
  FC_REQUIRE(ctx().pinned_bytes >= need, "slot 0 of this batch has not been enqueued");
  unsigned long long *local_host =
      static_cast<unsigned long long *>(ctx().pinned) + (size_t)n_slots * (size_t)(W + 16) + (size_t)slot * 8;
  FC_TRY(d2h(local_host, cnt, 8 * sizeof(uint64_t)));
  FC_TRY(launch_compact_gathered(dev_gathered, (int)world, cap, ens->gathered.as<uint64_t>(), cnt));
  return ladder_single(ens, nullptr, min_per_group, nullptr, nullptr, nullptr, nullptr,
                       ens->gathered.as<uint64_t>(), true, false, slot);
}

int fc_prune_collect(fc_ensemble *ens, int64_t slot, int64_t n_slots, uint8_t *mask_out, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && mask_out, "NULL pointer argument");
  FC_REQUIRE(n_slots >= 1 && slot >= 0 && slot < n_slots, "bad slot");
  FC_TRY(ensure_init());
  FC_TRY(sync());
  const int64_t N = ens->N, W = ens->W;
  int64_t survivors = 0;
  if (!ladder_collect(ens, slot, mask_out, nullptr, &survivors, nullptr))
    return set_error(FC_E_LIMIT, "prune %lld: a rank's pair list was missing or too long for the device ladder",
                     (long long)slot);
  if (stats) {
    const unsigned long long *local =
        static_cast<const unsigned long long *>(ctx().pinned) + (size_t)n_slots * (size_t)(W + 16) + (size_t)slot * 8;
    stats[0] = 0;
    const int64_t nb = ceil_div(N, ens->row_block);
    for (int64_t lb = 0, b; (b = global_block(lb, ens->rank, ens->world)) < nb; ++lb)
      for (int64_t i = b * ens->row_block; i < std::min(N, (b + 1) * ens->row_block); ++i)
        stats[0] += N - 1 - i;
    stats[1] = (int64_t)local[1];
    stats[2] = (int64_t)local[2];
    stats[3] = (int64_t)local[3];
    stats[4] = last_screen_ns();  // the most recent screen kernel of this rank
    stats[5] = survivors;
  }
  return FC_OK;
}

// ---- a6 ------------------------------------------------------------------------
int fc_inertia_moments(const double *coords, int64_t N, int64_t A, const double *masses,
                       double *moments_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && masses && moments_out, "NULL pointer argument");
  FC_TRY(ensure_init());
  DevBuf dc, dm, dout;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(dm, masses, (size_t)A));
  FC_TRY(dout.reserve((size_t)N * 3 * sizeof(double)));
  FC_TRY(launch_inertia_moments(dc.as<double>(), N, A, dm.as<double>(), dout.as<double>()));
  FC_TRY(d2h(moments_out, dout.p, (size_t)N * 3 * sizeof(double)));
  return sync();
}

int fc_prune_moi(const double *coords, int64_t N, int64_t A, const double *masses,
                 double max_deviation, const double *energies, double max_dE,
                 int64_t min_per_group, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1 && min_per_group >= 1, "bad arguments");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && masses && mask_out, "NULL pointer argument");
  FC_TRY(ensure_init());
  fc_ensemble e;
  e.N = N;
  e.Npad = ceil_div(N, 64) * 64;
  e.W = e.Npad / 64;
  e.row_block = 64;
  FC_TRY(e.maskA.reserve((size_t)e.Npad));
  FC_TRY(e.maskB.reserve((size_t)e.Npad));
  FC_TRY(e.mbits.reserve((size_t)e.W * sizeof(uint64_t)));
  FC_TRY(e.counters.reserve(kCounters * sizeof(uint64_t)));
  const int64_t rows = ceil_div(N, e.row_block) * e.row_block;
  FC_TRY(e.bits.reserve((size_t)rows * e.W * sizeof(uint64_t)));
  DevBuf dc, dm, dmom;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(dm, masses, (size_t)A));
  FC_TRY(dmom.reserve((size_t)N * 3 * sizeof(double)));
  const double *en_dev = nullptr;
  if (energies) {
    FC_TRY(upload(e.energies, energies, (size_t)N));
    en_dev = e.energies.as<double>();
  }
  FC_TRY(launch_inertia_moments(dc.as<double>(), N, A, dm.as<double>(), dmom.as<double>()));
  FC_TRY(launch_moi_simbits(dmom.as<double>(), N, max_deviation, en_dev, max_dE,
                            e.bits.as<uint64_t>(), e.W));
  return ladder_single(&e, e.bits.as<uint64_t>(), min_per_group, mask_out, nullptr, nullptr);
}

// ---- the similarity stages of the drivers on ONE upload --------------------------
// firecode/ensemble.py:205-235 and embedder.py:1445-1474 run prune_by_moment_of_inertia, apply its mask,
// then prune_by_rmsd on the survivors; each call of the reference re-reads host arrays.  Here the
// coordinates go to HBM once: the MOI stage works on them, the survivors are GATHERED on the device
// into the RMSD stage's prepared layout (k_prep_tile with an index list), and the stage masks are
// composed on the way out.  Per stage the result equals the stand-alone entry point's.
int fc_prune_similarity(const double *coords, int64_t N, int64_t A, const uint8_t *heavy_mask, const double *masses,
                        int do_moi, double moi_tol, int do_rmsd, double max_rmsd, double max_dev,
                        const double *energies, double max_dE, int64_t min_per_group, uint8_t *mask_moi_out,
                        uint8_t *mask_out, int64_t *counts) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1 && A <= 32767 && min_per_group >= 1, "bad arguments");
  if (counts) counts[0] = N, counts[1] = N, counts[2] = N;
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && mask_out, "NULL pointer argument");
  FC_REQUIRE(!do_moi || (masses != nullptr && moi_tol > 0.0), "the MOI stage needs masses and a positive tolerance");
  FC_REQUIRE(!do_rmsd || (max_rmsd > 0.0 && max_dev > 0.0), "thresholds must be positive");
  FC_TRY(ensure_init());
  DevBuf raw;
  FC_TRY(upload(raw, coords, (size_t)N * A * 3));  // the one upload of the coordinates
  std::vector<uint8_t> m1((size_t)N, 1);
  if (do_moi) {
    fc_ensemble e;
    e.N = N;
    e.Npad = ceil_div(N, 64) * 64;
    e.W = e.Npad / 64;
    e.row_block = 64;
    FC_TRY(e.maskA.reserve((size_t)e.Npad));
    FC_TRY(e.maskB.reserve((size_t)e.Npad));
    FC_TRY(e.mbits.reserve((size_t)e.W * sizeof(uint64_t)));
    FC_TRY(e.counters.reserve(kCounters * sizeof(uint64_t)));
    const int64_t rows = ceil_div(N, e.row_block) * e.row_block;
    FC_TRY(e.bits.reserve((size_t)rows * e.W * sizeof(uint64_t)));
    DevBuf dm, dmom;
    FC_TRY(upload(dm, masses, (size_t)A));
    FC_TRY(dmom.reserve((size_t)N * 3 * sizeof(double)));
    const double *en_dev = nullptr;
    if (energies) {
      FC_TRY(upload(e.energies, energies, (size_t)N));
      en_dev = e.energies.as<double>();
    }
    FC_TRY(launch_inertia_moments(raw.as<double>(), N, A, dm.as<double>(), dmom.as<double>()));
    FC_TRY(launch_moi_simbits(dmom.as<double>(), N, moi_tol, en_dev, max_dE, e.bits.as<uint64_t>(), e.W));
    FC_TRY(ladder_single(&e, e.bits.as<uint64_t>(), min_per_group, m1.data(), nullptr, nullptr));
  }
  if (mask_moi_out) std::memcpy(mask_moi_out, m1.data(), (size_t)N);
  std::vector<int32_t> idx;
  idx.reserve((size_t)N);
  for (int64_t i = 0; i < N; ++i)
    if (m1[(size_t)i]) idx.push_back((int32_t)i);
  const int64_t N1 = (int64_t)idx.size();
  if (counts) counts[1] = N1, counts[2] = N1;
  if (!do_rmsd || N1 == 0) {
    std::memcpy(mask_out, m1.data(), (size_t)N);
    return FC_OK;
  }
  // survivors gathered into the RMSD stage's layout on the device (no second upload of coordinates)
  DevBuf didx;
  const bool gather = N1 != N;
  if (gather) FC_TRY(upload(didx, idx.data(), idx.size()));
  std::unique_ptr<fc_ensemble> e2(new (std::nothrow) fc_ensemble);
  if (!e2) return set_error(FC_E_NOMEM, "host allocation failed");
  e2->epoch = ctx().epoch;
  FC_TRY(ensemble_build_dev(raw.as<double>(), N1, A, heavy_mask, 1, gather ? didx.as<int32_t>() : nullptr, e2.get()));
  std::vector<double> en1;
  if (energies) {
    en1.resize((size_t)N1);
    for (int64_t k = 0; k < N1; ++k) en1[(size_t)k] = energies[idx[(size_t)k]];
  }
  std::vector<uint8_t> m2((size_t)N1, 0);
  int64_t st[6] = {0};
  FC_TRY(fc_prune_rmsd(e2.get(), max_rmsd, max_dev, energies ? en1.data() : nullptr, max_dE, min_per_group, m2.data(), st));
  std::memset(mask_out, 0, (size_t)N);
  int64_t alive = 0;
  for (int64_t k = 0; k < N1; ++k)
    if (m2[(size_t)k]) {
      mask_out[idx[(size_t)k]] = 1;
      ++alive;
    }
  if (counts) counts[2] = alive;
  return FC_OK;
}

// ---- a7: prune_by_rmsd_rot_corr (prism_pruner.pruner; firecode/ensemble.py:253-260) ----------
int fc_prune_rmsd_rot_corr(const double *coords, int64_t N, int64_t A, const uint8_t *heavy_mask,
                           const int64_t *torsions, int64_t T, const uint8_t *rotation_masks,
                           const double *angles, const int32_t *n_angles, int64_t max_angles,
                           double max_rmsd, double max_dev, const double *energies, double max_dE,
                           int64_t min_per_group, uint8_t *mask_out, uint64_t *bits_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1 && T >= 0 && min_per_group >= 1, "bad arguments");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && heavy_mask && mask_out, "NULL pointer argument");
  FC_REQUIRE(T == 0 || (torsions && rotation_masks && angles && n_angles), "NULL torsion arrays");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_REQUIRE(max_angles >= 1 && max_angles <= 64, "1..64 trial angles per torsion");
  int64_t n_heavy = 0;
  for (int64_t a = 0; a < A; ++a) n_heavy += heavy_mask[a] ? 1 : 0;
  FC_REQUIRE(n_heavy >= 1, "the heavy-atom mask selects no atom");
  for (int64_t t = 0; t < T; ++t) {
    for (int k = 0; k < 4; ++k)
      FC_REQUIRE(torsions[t * 4 + k] >= 0 && torsions[t * 4 + k] < A, "torsion %lld: atom index out of range", (long long)t);
    FC_REQUIRE(n_angles[t] >= 1 && n_angles[t] <= max_angles, "torsion %lld: bad angle count", (long long)t);
  }
  if ((size_t)4 * A * 24 > 160 * 1024) return set_error(FC_E_LIMIT, "A=%lld too large for the LDS slice", (long long)A);
  FC_TRY(ensure_init());
  fc_ensemble e;
  e.N = N;
  e.Npad = ceil_div(N, 64) * 64;
  e.W = e.Npad / 64;
  e.row_block = 64;
  FC_TRY(e.maskA.reserve((size_t)e.Npad));
  FC_TRY(e.maskB.reserve((size_t)e.Npad));
  FC_TRY(e.mbits.reserve((size_t)e.W * sizeof(uint64_t)));
  FC_TRY(e.counters.reserve(kCounters * sizeof(uint64_t)));
  const int64_t rows = ceil_div(N, e.row_block) * e.row_block;
  const size_t bits_bytes = (size_t)rows * e.W * sizeof(uint64_t);
  FC_TRY(e.bits.reserve(bits_bytes));
  FC_HIP_TRY(hipMemsetAsync(e.bits.p, 0, bits_bytes, ctx().stream));
  DevBuf dc, dcen, dh, dt, dm, da, dn;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(dcen.reserve((size_t)N * A * 3 * sizeof(double)));
  FC_TRY(upload(dh, heavy_mask, (size_t)A));
  if (T > 0) {
    FC_TRY(upload(dt, torsions, (size_t)T * 4));
    FC_TRY(upload(dm, rotation_masks, (size_t)T * A));
    FC_TRY(upload(da, angles, (size_t)T * max_angles));
    FC_TRY(upload(dn, n_angles, (size_t)T));
  }
  const double *en_dev = nullptr;
  if (energies) {
    FC_TRY(upload(e.energies, energies, (size_t)N));
    en_dev = e.energies.as<double>();
  }
  FC_TRY(launch_center_structures(dc.as<double>(), N, A, dcen.as<double>()));
  FC_TRY(launch_rotcorr_simbits(dcen.as<double>(), N, A, dh.as<uint8_t>(), dt.as<int64_t>(), T, dm.as<uint8_t>(),
                                da.as<double>(), dn.as<int32_t>(), (int)max_angles, max_rmsd, max_dev, en_dev,
                                max_dE, e.bits.as<uint64_t>(), e.W));
  if (bits_out) FC_TRY(d2h(bits_out, e.bits.p, (size_t)N * e.W * sizeof(uint64_t)));
  return ladder_single(&e, e.bits.as<uint64_t>(), min_per_group, mask_out, nullptr, nullptr);
}

// ---- a8 / a13 --------------------------------------------------------------------
int fc_align_to_first(const double *coords, int64_t N, int64_t A, const int64_t *idx,
                      int64_t n_idx, double *out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && out, "NULL pointer argument");
  if (idx == nullptr) n_idx = A;
  FC_REQUIRE(n_idx >= 1, "n_idx must be >= 1");
  if (idx)
    for (int64_t k = 0; k < n_idx; ++k)
      FC_REQUIRE(idx[k] >= 0 && idx[k] < A, "idx[%lld] out of range", (long long)k);
  FC_TRY(ensure_init());
  DevBuf dc, di, dout;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  if (idx) FC_TRY(upload(di, idx, (size_t)n_idx));
  FC_TRY(dout.reserve((size_t)N * A * 3 * sizeof(double)));
  FC_TRY(launch_align_to_first(dc.as<double>(), N, A, idx ? di.as<int64_t>() : nullptr, n_idx,
                               dout.as<double>()));
  FC_TRY(d2h(out, dout.p, (size_t)N * A * 3 * sizeof(double)));
  return sync();
}

int fc_rototranslate(const double *coords, int64_t n, int64_t A, const double *R, const double *t,
                     double *out) {
  FC_API_LOCK;
  FC_REQUIRE(n >= 0 && A >= 1, "bad shape");
  if (n == 0) return FC_OK;
  FC_REQUIRE(coords && R && t && out, "NULL pointer argument");
  FC_TRY(ensure_init());
  DevBuf dc, dR, dt, dout;
  FC_TRY(upload(dc, coords, (size_t)n * A * 3));
  FC_TRY(upload(dR, R, (size_t)n * 9));
  FC_TRY(upload(dt, t, (size_t)n * 3));
  FC_TRY(dout.reserve((size_t)n * A * 3 * sizeof(double)));
  FC_TRY(launch_rototranslate(dc.as<double>(), n, A, dR.as<double>(), dt.as<double>(),
                              dout.as<double>()));
  FC_TRY(d2h(out, dout.p, (size_t)n * A * 3 * sizeof(double)));
  return sync();
}

// ---- a11 / a12 ---------------------------------------------------------------------
static const int64_t kMaxLdsAtoms = 160 * 1024 / 24;  // one structure per wavefront in LDS

int fc_clash_self(const double *coords, int64_t N, int64_t A, double lo, double hi,
                  int64_t *counts_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && counts_out, "NULL pointer argument");
  if (A > kMaxLdsAtoms) return set_error(FC_E_LIMIT, "A=%lld exceeds %lld atoms", (long long)A, (long long)kMaxLdsAtoms);
  FC_TRY(ensure_init());
  DevBuf dc, dn;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(dn.reserve((size_t)N * sizeof(int64_t)));
  FC_TRY(launch_clash_self(dc.as<double>(), N, A, lo, hi, dn.as<int64_t>()));
  FC_TRY(d2h(counts_out, dn.p, (size_t)N * sizeof(int64_t)));
  return sync();
}

int fc_clash_fragments(const double *coords, int64_t N, int64_t A, const int64_t *ids,
                       int64_t n_ids, double thresh, int64_t max_clashes, int64_t *counts_out,
                       uint8_t *pass_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  FC_REQUIRE(ids != nullptr && (n_ids == 2 || n_ids == 3), "ids must hold 2 or 3 fragment lengths");
  int64_t tot = 0;
  for (int64_t k = 0; k < n_ids; ++k) {
    FC_REQUIRE(ids[k] >= 0, "negative fragment length");
    tot += ids[k];
  }
  // reference slices m_last = coords[sum(ids[:-1]):] -- the last fragment takes the rest
  FC_REQUIRE(tot - ids[n_ids - 1] <= A, "fragment lengths exceed A");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && (counts_out || pass_out), "NULL pointer argument");
  if (A > kMaxLdsAtoms) return set_error(FC_E_LIMIT, "A=%lld exceeds %lld atoms", (long long)A, (long long)kMaxLdsAtoms);
  FC_TRY(ensure_init());
  DevBuf dc, dn, dp;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(dn.reserve((size_t)N * sizeof(int64_t)));
  FC_TRY(dp.reserve((size_t)N));
  FC_TRY(launch_clash_fragments(dc.as<double>(), N, A, ids, n_ids, thresh, max_clashes,
                                dn.as<int64_t>(), dp.as<uint8_t>()));
  if (counts_out) FC_TRY(d2h(counts_out, dn.p, (size_t)N * sizeof(int64_t)));
  if (pass_out) FC_TRY(d2h(pass_out, dp.p, (size_t)N));
  return sync();
}

int fc_clash_graph(const double *coords, int64_t N, int64_t A, const uint8_t *adj, double thresh,
                   int64_t *counts_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && adj && counts_out, "NULL pointer argument");
  FC_REQUIRE(thresh > 0.0, "thresh must be positive");
  if (A > kMaxLdsAtoms) return set_error(FC_E_LIMIT, "A=%lld exceeds %lld atoms", (long long)A, (long long)kMaxLdsAtoms);
  FC_TRY(ensure_init());
  DevBuf dc, da, dn;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(da, adj, (size_t)A * A));
  FC_TRY(dn.reserve((size_t)N * sizeof(int64_t)));
  FC_TRY(launch_clash_graph(dc.as<double>(), N, A, da.as<uint8_t>(), thresh, dn.as<int64_t>()));
  FC_TRY(d2h(counts_out, dn.p, (size_t)N * sizeof(int64_t)));
  return sync();
}

int fc_fitness_check(const double *coords, int64_t N, int64_t A, const int64_t *pairs,
                     const double *targets, int64_t C, double threshold, double *error_out,
                     uint8_t *pass_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1 && C >= 0, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(coords && pass_out && (C == 0 || (pairs && targets)), "NULL pointer argument");
  for (int64_t k = 0; k < N * C * 2; ++k) FC_REQUIRE(pairs[k] >= 0 && pairs[k] < A, "constraint index out of range");
  FC_TRY(ensure_init());
  DevBuf dc, dp, dt, de, dm;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(dp, pairs, (size_t)N * C * 2));
  FC_TRY(upload(dt, targets, (size_t)N * C));
  FC_TRY(de.reserve((size_t)N * sizeof(double)));
  FC_TRY(dm.reserve((size_t)N));
  FC_TRY(launch_fitness(dc.as<double>(), N, A, dp.as<int64_t>(), dt.as<double>(), C, threshold,
                        de.as<double>(), dm.as<uint8_t>()));
  if (error_out) FC_TRY(d2h(error_out, de.p, (size_t)N * sizeof(double)));
  FC_TRY(d2h(pass_out, dm.p, (size_t)N));
  return sync();
}

// ---- a14 -------------------------------------------------------------------------
int fc_embed_poses_clash(const double *m1, int64_t n1, int64_t A1, const double *m2, int64_t n2,
                         int64_t A2, const int64_t *c1, const int64_t *c2, const double *R1,
                         const double *t1, const double *R2, const double *t2, int64_t P,
                         double thresh, int64_t max_clashes, int64_t *counts_out,
                         uint8_t *pass_out, double *poses_out) {
  FC_API_LOCK;
  FC_REQUIRE(n1 >= 1 && n2 >= 1 && A1 >= 1 && A2 >= 1 && P >= 0, "bad shape");
  if (P == 0) return FC_OK;
  FC_REQUIRE(m1 && m2 && c1 && c2 && R1 && t1 && R2 && t2, "NULL pointer argument");
  FC_REQUIRE(counts_out || pass_out || poses_out, "no output requested");
  if (4 * A1 * 24 > 160 * 1024) return set_error(FC_E_LIMIT, "A1=%lld too large for the LDS slice", (long long)A1);
  for (int64_t k = 0; k < P; ++k)
    FC_REQUIRE(c1[k] >= 0 && c1[k] < n1 && c2[k] >= 0 && c2[k] < n2, "conformer id out of range at pose %lld", (long long)k);
  FC_TRY(ensure_init());
  DevBuf dm1, dm2, dc1, dc2, dR1, dt1, dR2, dt2, dn, dp, dposes;
  FC_TRY(upload(dm1, m1, (size_t)n1 * A1 * 3));
  FC_TRY(upload(dm2, m2, (size_t)n2 * A2 * 3));
  FC_TRY(upload(dc1, c1, (size_t)P));
  FC_TRY(upload(dc2, c2, (size_t)P));
  FC_TRY(upload(dR1, R1, (size_t)P * 9));
  FC_TRY(upload(dt1, t1, (size_t)P * 3));
  FC_TRY(upload(dR2, R2, (size_t)P * 9));
  FC_TRY(upload(dt2, t2, (size_t)P * 3));
  FC_TRY(dn.reserve((size_t)P * sizeof(int64_t)));
  FC_TRY(dp.reserve((size_t)P));
  if (poses_out) FC_TRY(dposes.reserve((size_t)P * (A1 + A2) * 3 * sizeof(double)));
  FC_TRY(launch_embed_poses_clash(dm1.as<double>(), A1, dm2.as<double>(), A2, dc1.as<int64_t>(),
                                  dc2.as<int64_t>(), dR1.as<double>(), dt1.as<double>(),
                                  dR2.as<double>(), dt2.as<double>(), P, thresh, max_clashes,
                                  dn.as<int64_t>(), dp.as<uint8_t>(),
                                  poses_out ? dposes.as<double>() : nullptr));
  if (counts_out) FC_TRY(d2h(counts_out, dn.p, (size_t)P * sizeof(int64_t)));
  if (pass_out) FC_TRY(d2h(pass_out, dp.p, (size_t)P));
  if (poses_out) FC_TRY(d2h(poses_out, dposes.p, (size_t)P * (A1 + A2) * 3 * sizeof(double)));
  return sync();
}

static int check_embed_mol(const double *m, int64_t n, int64_t A, const int64_t *reactive, int64_t nr,
                           const double *ps, const double *pe, const double *angles, int64_t na) {
  FC_REQUIRE(m && reactive && ps && pe && angles, "NULL pointer argument");
  FC_REQUIRE(n >= 1 && A >= 1 && na >= 1, "bad shape");
  FC_REQUIRE(nr == 1 || nr == 2, "a molecule has 1 or 2 reactive atoms");
  for (int64_t k = 0; k < nr; ++k) FC_REQUIRE(reactive[k] >= 0 && reactive[k] < A, "reactive index out of range");
  return FC_OK;
}

int fc_embed_mol_transforms(const double *coords, int64_t n, int64_t A, const int64_t *reactive,
                            int64_t nr, const double *pivot_start, const double *pivot_end,
                            int64_t mol, const double *angles, int64_t na, double *R_out,
                            double *t_out) {
  FC_API_LOCK;
  FC_TRY(check_embed_mol(coords, n, A, reactive, nr, pivot_start, pivot_end, angles, na));
  FC_REQUIRE((mol == 0 || mol == 1) && R_out && t_out, "bad arguments");
  FC_TRY(ensure_init());
  DevBuf dc, dr, dps, dpe, da, dR, dt;
  FC_TRY(upload(dc, coords, (size_t)n * A * 3));
  FC_TRY(upload(dr, reactive, (size_t)nr));
  FC_TRY(upload(dps, pivot_start, (size_t)n * 3));
  FC_TRY(upload(dpe, pivot_end, (size_t)n * 3));
  FC_TRY(upload(da, angles, (size_t)na));
  const size_t G = (size_t)n * 2 * na;
  FC_TRY(dR.reserve(G * 9 * sizeof(double)));
  FC_TRY(dt.reserve(G * 3 * sizeof(double)));
  FC_TRY(launch_embed_mol_transforms(dc.as<double>(), n, A, dr.as<int64_t>(), (int)nr, dps.as<double>(),
                                     dpe.as<double>(), (int)mol, da.as<double>(), na, dR.as<double>(),
                                     dt.as<double>()));
  FC_TRY(d2h(R_out, dR.p, G * 9 * sizeof(double)));
  FC_TRY(d2h(t_out, dt.p, G * 3 * sizeof(double)));
  return sync();
}

static int embed_grid(const double *m1, int64_t n1, int64_t A1, const int64_t *reactive1,
                      int64_t nr1, const double *ps1, const double *pe1, const double *m2,
                      int64_t n2, int64_t A2, const int64_t *reactive2, int64_t nr2,
                      const double *ps2, const double *pe2, const double *angles1, int64_t na1,
                      const double *angles2, int64_t na2, double thresh, int64_t max_clashes,
                      uint8_t *pass_out, int32_t *counts_out, double *ms_kernel, double rmsd_thr,
                      uint8_t *accept_out) {
  FC_TRY(check_embed_mol(m1, n1, A1, reactive1, nr1, ps1, pe1, angles1, na1));
  FC_TRY(check_embed_mol(m2, n2, A2, reactive2, nr2, ps2, pe2, angles2, na2));
  FC_REQUIRE(pass_out != nullptr && max_clashes >= 0, "bad arguments");
  if (A1 * 40 + 64 > 64 * 1024) return set_error(FC_E_LIMIT, "A1=%lld too large for the LDS stage", (long long)A1);
  FC_TRY(ensure_init());
  Context &c = ctx();
  const int64_t P = n1 * n2 * 2 * na1 * na2;
  const int64_t S2 = ceil_div(n2 * na2, 64) * 64;
  DevBuf d1, d2, r1, r2, s1, e1, s2, e2, a1, a2, R1, t1, R2, t2, X1, X2s, dpass, dcnt, dmax;
  FC_TRY(upload(d1, m1, (size_t)n1 * A1 * 3));
  FC_TRY(upload(d2, m2, (size_t)n2 * A2 * 3));
  FC_TRY(upload(r1, reactive1, (size_t)nr1));
  FC_TRY(upload(r2, reactive2, (size_t)nr2));
  FC_TRY(upload(s1, ps1, (size_t)n1 * 3));
  FC_TRY(upload(e1, pe1, (size_t)n1 * 3));
  FC_TRY(upload(s2, ps2, (size_t)n2 * 3));
  FC_TRY(upload(e2, pe2, (size_t)n2 * 3));
  FC_TRY(upload(a1, angles1, (size_t)na1));
  FC_TRY(upload(a2, angles2, (size_t)na2));
  const size_t G1 = (size_t)n1 * 2 * na1, G2 = (size_t)n2 * 2 * na2;
  FC_TRY(R1.reserve(G1 * 9 * sizeof(double)));
  FC_TRY(t1.reserve(G1 * 3 * sizeof(double)));
  FC_TRY(R2.reserve(G2 * 9 * sizeof(double)));
  FC_TRY(t2.reserve(G2 * 3 * sizeof(double)));
  FC_TRY(X1.reserve(G1 * A1 * 3 * sizeof(double)));
  FC_TRY(X2s.reserve((size_t)2 * A2 * 3 * S2 * sizeof(double)));
  FC_TRY(dmax.reserve(256 + (size_t)2 * A2 * 3 * S2 * sizeof(float)));  // scratch of the clash kernel
  FC_TRY(dpass.reserve((size_t)P));
  if (counts_out) FC_TRY(dcnt.reserve((size_t)P * sizeof(int32_t)));
  FC_HIP_TRY(hipMemsetAsync(X2s.p, 0, (size_t)2 * A2 * 3 * S2 * sizeof(double), c.stream));
  FC_TRY(launch_embed_mol_transforms(d1.as<double>(), n1, A1, r1.as<int64_t>(), (int)nr1, s1.as<double>(),
                                     e1.as<double>(), 0, a1.as<double>(), na1, R1.as<double>(), t1.as<double>()));
  FC_TRY(launch_embed_mol_transforms(d2.as<double>(), n2, A2, r2.as<int64_t>(), (int)nr2, s2.as<double>(),
                                     e2.as<double>(), 1, a2.as<double>(), na2, R2.as<double>(), t2.as<double>()));
  FC_TRY(launch_embed_pretransform(d1.as<double>(), n1, A1, na1, R1.as<double>(), t1.as<double>(), 1, 0,
                                   X1.as<double>()));
  FC_TRY(launch_embed_pretransform(d2.as<double>(), n2, A2, na2, R2.as<double>(), t2.as<double>(), 0, S2,
                                   X2s.as<double>()));
  FC_HIP_TRY(hipEventRecord(c.ev0, c.stream));
  FC_TRY(launch_embed_grid_clash(X1.as<double>(), n1, A1, na1, X2s.as<double>(), n2, A2, na2, S2, thresh,
                                 max_clashes, dmax.p, dmax.bytes, dpass.as<uint8_t>(),
                                 counts_out ? dcnt.as<int32_t>() : nullptr));
  FC_HIP_TRY(hipEventRecord(c.ev1, c.stream));
  DevBuf X2a, dacc;
  if (accept_out != nullptr) {
    FC_REQUIRE(rmsd_thr > 0.0, "rmsd_thr must be positive");
    FC_REQUIRE(na1 * na2 * 4 * (int64_t)sizeof(int) <= 64 * 1024, "too many angle pairs per group for the LDS list");
    FC_TRY(X2a.reserve(G2 * A2 * 3 * sizeof(double)));
    FC_TRY(dacc.reserve((size_t)P));
    FC_TRY(launch_embed_pretransform(d2.as<double>(), n2, A2, na2, R2.as<double>(), t2.as<double>(), 1, 0,
                                     X2a.as<double>()));
    FC_TRY(launch_embed_group_dedupe(X1.as<double>(), n1, A1, na1, X2a.as<double>(), n2, A2, na2, rmsd_thr,
                                     dpass.as<uint8_t>(), dacc.as<uint8_t>()));
    FC_TRY(d2h(accept_out, dacc.p, (size_t)P));
  }
  FC_TRY(d2h(pass_out, dpass.p, (size_t)P));
  if (counts_out) FC_TRY(d2h(counts_out, dcnt.p, (size_t)P * sizeof(int32_t)));
  FC_TRY(sync());
  if (ms_kernel) {
    float ms = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
    *ms_kernel = ms;
  }
  return FC_OK;
}

int fc_embed_grid_clash(const double *m1, int64_t n1, int64_t A1, const int64_t *reactive1,
                        int64_t nr1, const double *ps1, const double *pe1, const double *m2,
                        int64_t n2, int64_t A2, const int64_t *reactive2, int64_t nr2,
                        const double *ps2, const double *pe2, const double *angles1, int64_t na1,
                        const double *angles2, int64_t na2, double thresh, int64_t max_clashes,
                        uint8_t *pass_out, int32_t *counts_out, double *ms_kernel) {
  FC_API_LOCK;
  return embed_grid(m1, n1, A1, reactive1, nr1, ps1, pe1, m2, n2, A2, reactive2, nr2, ps2, pe2, angles1,
                    na1, angles2, na2, thresh, max_clashes, pass_out, counts_out, ms_kernel, 0.0, nullptr);
}

int fc_embed_grid_dedupe(const double *m1, int64_t n1, int64_t A1, const int64_t *reactive1,
                         int64_t nr1, const double *ps1, const double *pe1, const double *m2,
                         int64_t n2, int64_t A2, const int64_t *reactive2, int64_t nr2,
                         const double *ps2, const double *pe2, const double *angles1, int64_t na1,
                         const double *angles2, int64_t na2, double thresh, int64_t max_clashes,
                         double rmsd_thr, uint8_t *pass_out, uint8_t *accept_out) {
  FC_API_LOCK;
  FC_REQUIRE(accept_out != nullptr, "accept_out is NULL");
  return embed_grid(m1, n1, A1, reactive1, nr1, ps1, pe1, m2, n2, A2, reactive2, nr2, ps2, pe2, angles1,
                    na1, angles2, na2, thresh, max_clashes, pass_out, nullptr, nullptr, rmsd_thr, accept_out);
}

// ---- a14, three molecules: cyclical_embed (firecode/embeds.py:409-585) -----------------------
int fc_embed_trimolecular(const double *const coords[3], const int64_t n_conf[3], const int64_t n_atoms[3],
                          const int64_t *const reactive[3], const int64_t n_reactive[3], int64_t J,
                          const int64_t *conf, const double *piv_start, const double *piv_end,
                          const double *vecs, const double *dirs0, const uint8_t *run, const int64_t *rtab,
                          const double *norms, const double *ua, int64_t U, const int32_t *aidx, int64_t S,
                          double thresh, int64_t max_clashes, double rmsd_thr, double *dirs_out,
                          double *Rt_out, uint8_t *pass_out, uint8_t *accept_out) {
  FC_API_LOCK;
  FC_REQUIRE(coords && n_conf && n_atoms && reactive && n_reactive, "NULL pointer argument");
  FC_REQUIRE(J >= 0 && U >= 1 && S >= 1, "bad shape");
  if (J == 0) return FC_OK;
  FC_REQUIRE(conf && piv_start && piv_end && vecs && dirs0 && run && rtab && norms && ua && aidx && dirs_out &&
                 Rt_out && pass_out && accept_out,
             "NULL pointer argument");
  FC_REQUIRE(thresh > 0.0 && rmsd_thr > 0.0 && max_clashes >= 0, "thresholds must be positive");
  int64_t Atot = 0;
  for (int i = 0; i < 3; ++i) {
    FC_REQUIRE(coords[i] && reactive[i] && n_conf[i] >= 1 && n_atoms[i] >= 1, "bad molecule %d", i);
    FC_REQUIRE(n_reactive[i] >= 1 && n_reactive[i] <= 2, "molecule %d: 1 or 2 reactive atoms expected", i);
    for (int64_t r = 0; r < n_reactive[i]; ++r)
      FC_REQUIRE(reactive[i][r] >= 0 && reactive[i][r] < n_atoms[i], "reactive index out of range");
    Atot += n_atoms[i];
  }
  // every index a kernel dereferences is checked here
  for (int64_t j = 0; j < J; ++j)
    for (int i = 0; i < 3; ++i) {
      FC_REQUIRE(conf[j * 3 + i] >= 0 && conf[j * 3 + i] < n_conf[i], "conformer index out of range (job %lld)",
                 (long long)j);
      for (int v = 0; v < 8; ++v)
        for (int k = 0; k < 3; ++k) {
          const int64_t r = rtab[((j * 8 + v) * 3 + i) * 3 + k];
          FC_REQUIRE(r >= 0 && r < n_atoms[i], "reactive-pair table entry out of range (job %lld)", (long long)j);
        }
    }
  for (int64_t s = 0; s < S * 3; ++s) FC_REQUIRE(aidx[s] >= 0 && aidx[s] < U, "angle index out of range");
  if (3 * U > 256) return set_error(FC_E_LIMIT, "U=%lld distinct step angles per molecule exceed 85", (long long)U);
  if (J * 8 * S >= (1ll << 31)) return set_error(FC_E_LIMIT, "too many poses in one call: split the jobs");
  const size_t lds = tri_group_lds_bytes(Atot, (int)U, (int)S);
  if (lds > 160 * 1024)
    return set_error(FC_E_LIMIT, "%zu bytes of LDS per group (atoms %lld x angles %lld, %lld poses) exceed 160 KB",
                     lds, (long long)Atot, (long long)U, (long long)S);
  FC_TRY(ensure_init());
  DevBuf dc[3], dr[3], dconf, dps, dpe, dvecs, dd0, drun, drt, dn, dua, daidx, ddirs, dRt, dpass, dacc;
  const double *cdev[3];
  const int64_t *rdev[3];
  for (int i = 0; i < 3; ++i) {
    FC_TRY(upload(dc[i], coords[i], (size_t)n_conf[i] * n_atoms[i] * 3));
    FC_TRY(upload(dr[i], reactive[i], (size_t)n_reactive[i]));
    cdev[i] = dc[i].as<double>();
    rdev[i] = dr[i].as<int64_t>();
  }
  FC_TRY(upload(dconf, conf, (size_t)J * 3));
  FC_TRY(upload(dps, piv_start, (size_t)J * 9));
  FC_TRY(upload(dpe, piv_end, (size_t)J * 9));
  FC_TRY(upload(dvecs, vecs, (size_t)J * 8 * 18));
  FC_TRY(upload(dd0, dirs0, (size_t)J * 9));
  FC_TRY(upload(drun, run, (size_t)J * 8));
  FC_TRY(upload(drt, rtab, (size_t)J * 8 * 9));
  FC_TRY(upload(dn, norms, (size_t)J * 3));
  FC_TRY(upload(dua, ua, (size_t)3 * U));
  FC_TRY(upload(daidx, aidx, (size_t)S * 3));
  FC_TRY(ddirs.reserve((size_t)J * 8 * 9 * sizeof(double)));
  FC_TRY(dRt.reserve((size_t)J * 8 * 3 * U * 12 * sizeof(double)));
  FC_TRY(dpass.reserve((size_t)J * 8 * S));
  FC_TRY(dacc.reserve((size_t)J * 8 * S));
  FC_HIP_TRY(hipMemsetAsync(dRt.p, 0, (size_t)J * 8 * 3 * U * 12 * sizeof(double), ctx().stream));
  FC_TRY(launch_tri_embed(cdev, rdev, n_atoms, n_reactive, J, dconf.as<int64_t>(), dps.as<double>(),
                          dpe.as<double>(), dvecs.as<double>(), dd0.as<double>(), drun.as<uint8_t>(),
                          drt.as<int64_t>(), dn.as<double>(), dua.as<double>(), (int)U, daidx.as<int32_t>(),
                          (int)S, thresh, (int)max_clashes, rmsd_thr, ddirs.as<double>(), dRt.as<double>(),
                          dpass.as<uint8_t>(), dacc.as<uint8_t>()));
  FC_TRY(d2h(dirs_out, ddirs.p, (size_t)J * 8 * 9 * sizeof(double)));
  FC_TRY(d2h(Rt_out, dRt.p, (size_t)J * 8 * 3 * U * 12 * sizeof(double)));
  FC_TRY(d2h(pass_out, dpass.p, (size_t)J * 8 * S));
  FC_TRY(d2h(accept_out, dacc.p, (size_t)J * 8 * S));
  return sync();
}

int fc_string_embed(const double *m1, int64_t n1, int64_t A1, const double *centers1,
                    const double *orbvecs1, int64_t K1, const double *m2, int64_t n2, int64_t A2,
                    const double *centers2, const double *orbvecs2, int64_t K2,
                    const double *angles, int64_t nA, const int64_t *quads, int64_t Q,
                    double thresh, int64_t max_clashes, double tfd_thresh, uint8_t *pass_out,
                    uint8_t *accept_out, double *R2_out, double *t2_out) {
  FC_API_LOCK;
  FC_REQUIRE(m1 && m2 && centers1 && orbvecs1 && centers2 && orbvecs2 && angles && pass_out && accept_out,
             "NULL pointer argument");
  FC_REQUIRE(n1 >= 1 && n2 >= 1 && A1 >= 1 && A2 >= 1 && K1 >= 1 && K2 >= 1 && nA >= 1 && Q >= 0, "bad shape");
  FC_REQUIRE(quads || Q == 0, "quads is NULL");
  if (Q > 128) return set_error(FC_E_LIMIT, "Q=%lld fingerprints exceed 128", (long long)Q);
  for (int64_t k = 0; k < Q * 4; ++k) FC_REQUIRE(quads[k] >= 0 && quads[k] < A1 + A2, "quadruplet index out of range");
  if (4 * A1 * 24 > 160 * 1024) return set_error(FC_E_LIMIT, "A1=%lld too large for the LDS slice", (long long)A1);
  FC_TRY(ensure_init());
  const int64_t P = n1 * n2 * K1 * K2 * nA;
  DevBuf d1, d2, dc1, dv1, dc2, dv2, da, dq, dR, dt, di1, di2, dpass, dacc, drej, dtf, daccT, dn;
  FC_TRY(upload(d1, m1, (size_t)n1 * A1 * 3));
  FC_TRY(upload(d2, m2, (size_t)n2 * A2 * 3));
  FC_TRY(upload(dc1, centers1, (size_t)n1 * K1 * 3));
  FC_TRY(upload(dv1, orbvecs1, (size_t)n1 * K1 * 3));
  FC_TRY(upload(dc2, centers2, (size_t)n2 * K2 * 3));
  FC_TRY(upload(dv2, orbvecs2, (size_t)n2 * K2 * 3));
  FC_TRY(upload(da, angles, (size_t)nA));
  FC_TRY(upload(dq, quads, (size_t)Q * 4));
  FC_TRY(dR.reserve((size_t)P * 9 * sizeof(double)));
  FC_TRY(dt.reserve((size_t)P * 3 * sizeof(double)));
  FC_TRY(di1.reserve((size_t)P * sizeof(int64_t)));
  FC_TRY(di2.reserve((size_t)P * sizeof(int64_t)));
  FC_TRY(dpass.reserve((size_t)P));
  FC_TRY(dacc.reserve((size_t)P));
  FC_TRY(drej.reserve(256));
  FC_TRY(dtf.reserve((size_t)P * std::max<int64_t>(Q, 1) * sizeof(double)));
  FC_TRY(daccT.reserve((size_t)P * std::max<int64_t>(Q, 1) * sizeof(double)));
  FC_TRY(dn.reserve(sizeof(uint64_t)));
  FC_HIP_TRY(hipMemsetAsync(dn.p, 0, sizeof(uint64_t), ctx().stream));
  FC_HIP_TRY(hipMemsetAsync(drej.p, 0, 256, ctx().stream));
  FC_TRY(launch_string_transforms(dc1.as<double>(), dv1.as<double>(), n1, K1, dc2.as<double>(), dv2.as<double>(),
                                  n2, K2, da.as<double>(), nA, dR.as<double>(), dt.as<double>(),
                                  di1.as<int64_t>(), di2.as<int64_t>()));
  FC_TRY(launch_embed_poses_clash(d1.as<double>(), A1, d2.as<double>(), A2, di1.as<int64_t>(), di2.as<int64_t>(),
                                  nullptr, nullptr, dR.as<double>(), dt.as<double>(), P, thresh, max_clashes,
                                  nullptr, dpass.as<uint8_t>(), nullptr));
  FC_TRY(launch_pose_fingerprints(d1.as<double>(), A1, d2.as<double>(), A2, di1.as<int64_t>(), di2.as<int64_t>(),
                                  dR.as<double>(), dt.as<double>(), P, dq.as<int64_t>(), Q, dpass.as<uint8_t>(),
                                  dtf.as<double>()));
  for (int64_t c0 = 0; c0 < P; c0 += 256)
    FC_TRY(launch_leader_chunk(dtf.as<double>(), Q, c0, P, dpass.as<uint8_t>(), daccT.as<double>(), P,
                               reinterpret_cast<unsigned long long *>(dn.p), tfd_thresh, drej.as<uint8_t>(),
                               dacc.as<uint8_t>()));
  FC_TRY(d2h(pass_out, dpass.p, (size_t)P));
  FC_TRY(d2h(accept_out, dacc.p, (size_t)P));
  if (R2_out) FC_TRY(d2h(R2_out, dR.p, (size_t)P * 9 * sizeof(double)));
  if (t2_out) FC_TRY(d2h(t2_out, dt.p, (size_t)P * 3 * sizeof(double)));
  return sync();
}

// ---- a17-a20 -----------------------------------------------------------------------
// tfd_keep_out != nullptr (fc_torsion_scan_tfd): the fingerprints never leave the device -- the list
// [starting structure] + [scanned conformers with at least one rotated bond] is TFD-pruned at once.
static int torsion_scan_impl(const double *base, int64_t A, const int64_t *torsions, int64_t T,
                             const uint8_t *rotmasks, const int64_t *angles, int64_t S, double thresh,
                             int64_t backoff_deg, const int64_t *quads, int64_t Q, double *coords_out,
                             int64_t *rotated_bonds_out, double *tf_out, double tfd_thresh = 0.0,
                             uint8_t *tfd_keep_out = nullptr, const int64_t *grid_values = nullptr,
                             const int64_t *grid_counts = nullptr) {
  // grid_values / grid_counts (with angles == nullptr): the angle-sets are the rows of cartesian_product over the T
  // value lists, generated on the device (k_angle_grid)
  FC_REQUIRE(A >= 2 && T >= 1 && S >= 0 && Q >= 0, "bad shape");
  FC_REQUIRE(backoff_deg != 0, "backoff_deg must be non-zero");
  if (S == 0) return FC_OK;
  FC_REQUIRE(base && torsions && rotmasks && (angles || (grid_values && grid_counts)) && rotated_bonds_out, "NULL pointer argument");
  FC_REQUIRE(coords_out || tf_out || tfd_keep_out, "nothing to compute: coords_out and tf_out are both NULL");
  const bool want_tf = tf_out != nullptr || tfd_keep_out != nullptr;
  FC_REQUIRE(!want_tf || (quads != nullptr && Q >= 1), "fingerprints need quadruplets");
  if (want_tf)
    for (int64_t k = 0; k < Q * 4; ++k) FC_REQUIRE(quads[k] >= 0 && quads[k] < A, "quadruplet index out of range");
  if (tfd_keep_out && Q > 128) return set_error(FC_E_LIMIT, "Q=%lld fingerprints exceed 128 (NumPy's summation order changes there)", (long long)Q);
  if (4 * A * 24 > 160 * 1024 || A > 32767)
    return set_error(FC_E_LIMIT, "A=%lld too large for the LDS slice", (long long)A);
  // moving / rest index lists per torsion (torsion_module.py:907-915)
  std::vector<int16_t> mv((size_t)T * A, 0), rs((size_t)T * A, 0);
  std::vector<int32_t> nmv((size_t)T, 0), nrs((size_t)T, 0);
  for (int64_t t = 0; t < T; ++t) {
    for (int k = 0; k < 4; ++k)
      FC_REQUIRE(torsions[t * 4 + k] >= 0 && torsions[t * 4 + k] < A, "torsion %lld index out of range", (long long)t);
    const int64_t i2 = torsions[t * 4 + 1], i3 = torsions[t * 4 + 2];
    for (int64_t a = 0; a < A; ++a) {
      if (rotmasks[t * A + a]) mv[(size_t)t * A + nmv[t]++] = (int16_t)a;
      else if (a != i2 && a != i3) rs[(size_t)t * A + nrs[t]++] = (int16_t)a;
    }
  }
  FC_TRY(ensure_init());
  static const bool dbg_laps = getenv("FC_DEBUG") != nullptr && getenv("FC_SCAN_LAPS") != nullptr;
  auto lap_t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg_laps) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[fc]   scan S=%lld %s %.2f ms\n", (long long)S, what, std::chrono::duration<double, std::milli>(t - lap_t0).count());
    lap_t0 = t;
  };
  DevBuf db, dt, dmk, dmv, drs, dnm, dnr, da, dout, drot, dq, dtf;
  FC_TRY(upload(db, base, (size_t)A * 3));
  FC_TRY(upload(dt, torsions, (size_t)T * 4));
  FC_TRY(upload(dmk, rotmasks, (size_t)T * A));
  FC_TRY(upload(dmv, mv.data(), mv.size()));
  FC_TRY(upload(drs, rs.data(), rs.size()));
  FC_TRY(upload(dnm, nmv.data(), nmv.size()));
  FC_TRY(upload(dnr, nrs.data(), nrs.size()));
  if (angles) {
    FC_TRY(upload(da, angles, (size_t)S * T));
  } else {
    std::vector<int64_t> first((size_t)T, 0);
    for (int64_t t = 1; t < T; ++t) first[(size_t)t] = first[(size_t)t - 1] + grid_counts[t - 1];
    DevBuf dval, dfirst, dcnt;
    FC_TRY(upload(dval, grid_values, (size_t)(first[(size_t)T - 1] + grid_counts[T - 1])));
    FC_TRY(upload(dfirst, first.data(), (size_t)T));
    FC_TRY(upload(dcnt, grid_counts, (size_t)T));
    FC_TRY(da.reserve((size_t)S * T * sizeof(int64_t)));
    FC_TRY(launch_angle_grid(dval.as<int64_t>(), dfirst.as<int64_t>(), dcnt.as<int64_t>(), T, S, da.as<int64_t>()));
    FC_TRY(sync());  // (`first` and the three small buffers end here)
  }
  lap("uploads enqueued");
  if (coords_out) FC_TRY(dout.reserve((size_t)S * A * 3 * sizeof(double)));
  FC_TRY(drot.reserve((size_t)S * sizeof(int64_t)));
  lap("output buffers");
  if (want_tf) {
    FC_TRY(upload(dq, quads, (size_t)Q * 4));
    FC_TRY(dtf.reserve((size_t)S * Q * sizeof(double)));
  }
  FC_TRY(launch_torsion_scan(db.as<double>(), A, dt.as<int64_t>(), T, dmk.as<uint8_t>(),
                             dmv.as<int16_t>(), drs.as<int16_t>(), dnm.as<int32_t>(),
                             dnr.as<int32_t>(), da.as<int64_t>(), S, thresh, backoff_deg,
                             coords_out ? dout.as<double>() : nullptr, drot.as<int64_t>(),
                             want_tf ? dq.as<int64_t>() : nullptr, Q, want_tf ? dtf.as<double>() : nullptr));
  lap("scan launched");
  if (dbg_laps) {
    (void)hipStreamSynchronize(ctx().stream);
    lap("scan kernels done");
  }
  if (coords_out) FC_TRY(d2h(coords_out, dout.p, (size_t)S * A * 3 * sizeof(double)));
  lap("coords down");
  if (tf_out) FC_TRY(d2h(tf_out, dtf.p, (size_t)S * Q * sizeof(double)));
  if (!tfd_keep_out) {
    FC_TRY(d2h(rotated_bonds_out, drot.p, (size_t)S * sizeof(int64_t)));
    const int rc_sync = sync();
    lap("synchronised");
    return rc_sync;
  }
  // rows of the TFD problem: the starting structure, then the scanned conformers that rotated a bond -- selected on the
  // device (the counts are 13 MB at 1.7 M angle-sets: down, through a host loop and up again cost 6 ms in front of the
  // first-match kernels; now the counts travel down BESIDE those kernels, on another stream)
  DevBuf dtf0, didx, dT, dfm, dcount, dseltmp;
  FC_TRY(didx.reserve((size_t)S * sizeof(int64_t)));
  FC_TRY(dcount.reserve(sizeof(int64_t)));
  FC_TRY(launch_select_rotated(drot.as<int64_t>(), S, didx.as<int64_t>(), dcount.as<int64_t>(), dseltmp));
  int64_t M = 0;
  FC_TRY(d2h(&M, dcount.p, sizeof(int64_t)));
  FC_TRY(sync());  // (the scan is complete here)
  lap("scan done, rows selected");
  const int64_t N = M + 1, Npad = ceil_div(N, 64) * 64;
  FC_TRY(dtf0.reserve((size_t)Q * sizeof(double)));
  FC_TRY(launch_torsion_fingerprint(db.as<double>(), 1, A, dq.as<int64_t>(), Q, dtf0.as<double>()));
  FC_TRY(dT.reserve((size_t)Q * Npad * sizeof(double)));
  FC_TRY(launch_gather_transpose_pad(dtf.as<double>(), dtf0.as<double>(), didx.as<int64_t>(), M, Q, Npad, dT.as<double>()));
  FC_TRY(dfm.reserve((size_t)N * sizeof(int64_t)));
  DevBuf dtfF;
  FC_TRY(dtfF.reserve((size_t)std::min<int64_t>(Q, 8) * Npad * sizeof(float)));
  FC_TRY(launch_tfd_first_match(dT.as<double>(), N, Npad, Q, tfd_thresh, dfm.as<int64_t>(), dtfF.as<float>()));
  {  // the counts, while the first-match walk runs (the copy into the caller's pageable array keeps this thread busy for
     // 0.3 ms when the array's pages are in place and 1.3 ms when they are fresh from the kernel -- which of the two
     // depends on the caller's allocator, not on this library: a pre-faulting helper thread was built and brought
     // nothing measurable, FC_PREFAULT A/B over 4 x 10 searches)
    FC_TRY(side_streams());
    std::memset(tfd_keep_out, 0, (size_t)S + 1);
    FC_TRY(d2h_staged(rotated_bonds_out, drot.p, (size_t)S * sizeof(int64_t), ctx().s_comm));
  }
  lap("first match enqueued, counts down");
  std::vector<uint8_t> mask((size_t)N);
  FC_TRY(tfd_ladder_from_device(dfm.as<int64_t>(), N, mask.data()));
  lap("ladder");
  // keep flags: row r >= 1 of the TFD problem is the (r - 1)-th angle-set that rotated a bond, and the device still holds
  // that list (didx).  The survivors are few (thousands of 1.7 M): their rows go up, their angle-sets come down -- a
  // walk over all S counts on the host cost 0.7 ms
  tfd_keep_out[0] = mask[0];
  std::vector<int64_t> rows;
  {
    const uint8_t *mb = mask.data();
    int64_t r = 1;
    for (; r + 8 <= N; r += 8) {
      uint64_t w;
      std::memcpy(&w, mb + r, 8);
      if (!w) continue;
      for (int b = 0; b < 8; ++b)
        if (mb[r + b]) rows.push_back(r + b);
    }
    for (; r < N; ++r)
      if (mb[r]) rows.push_back(r);
  }
  if (!rows.empty()) {
    DevBuf drows, dsets;
    FC_TRY(upload(drows, rows.data(), rows.size()));
    FC_TRY(dsets.reserve(rows.size() * sizeof(int64_t)));
    FC_TRY(launch_rows_to_sets(didx.as<int64_t>(), drows.as<int64_t>(), (int64_t)rows.size(), dsets.as<int64_t>()));
    std::vector<int64_t> sets(rows.size());
    FC_TRY(d2h(sets.data(), dsets.p, rows.size() * sizeof(int64_t)));
    FC_TRY(sync());
    for (const int64_t sidx : sets) {
      if (sidx < 0 || sidx >= S) return set_error(FC_E_HIP, "internal: device selection returned angle-set %lld of %lld", (long long)sidx, (long long)S);
      tfd_keep_out[1 + sidx] = 1;
    }
  }
  lap("keep mask assembled");
  return FC_OK;
}

int fc_torsion_scan(const double *base, int64_t A, const int64_t *torsions, int64_t T,
                    const uint8_t *rotmasks, const int64_t *angles, int64_t S, double thresh,
                    int64_t backoff_deg, double *coords_out, int64_t *rotated_bonds_out) {
  FC_API_LOCK;
  FC_REQUIRE(S == 0 || coords_out != nullptr, "NULL pointer argument");
  return torsion_scan_impl(base, A, torsions, T, rotmasks, angles, S, thresh, backoff_deg, nullptr, 0, coords_out,
                           rotated_bonds_out, nullptr);
}

int fc_torsion_scan_fingerprints(const double *base, int64_t A, const int64_t *torsions, int64_t T,
                                 const uint8_t *rotmasks, const int64_t *angles, int64_t S, double thresh,
                                 int64_t backoff_deg, const int64_t *quads, int64_t Q, double *tf_out,
                                 int64_t *rotated_bonds_out, double *coords_out) {
  FC_API_LOCK;
  FC_REQUIRE(S == 0 || tf_out != nullptr, "NULL pointer argument");
  return torsion_scan_impl(base, A, torsions, T, rotmasks, angles, S, thresh, backoff_deg, quads, Q, coords_out,
                           rotated_bonds_out, tf_out);
}

int fc_torsion_scan_tfd(const double *base, int64_t A, const int64_t *torsions, int64_t T, const uint8_t *rotmasks,
                        const int64_t *angles, int64_t S, double thresh, int64_t backoff_deg, const int64_t *quads,
                        int64_t Q, double tfd_thresh, int64_t *rotated_bonds_out, uint8_t *keep_out) {
  FC_API_LOCK;
  FC_REQUIRE(keep_out != nullptr, "NULL pointer argument");
  if (S == 0) {  // the starting structure alone: nothing to compare it with
    keep_out[0] = 1;
    return FC_OK;
  }
  return torsion_scan_impl(base, A, torsions, T, rotmasks, angles, S, thresh, backoff_deg, quads, Q, nullptr,
                           rotated_bonds_out, nullptr, tfd_thresh, keep_out);
}

int fc_torsion_scan_tfd_grid(const double *base, int64_t A, const int64_t *torsions, int64_t T, const uint8_t *rotmasks,
                             const int64_t *values, const int64_t *counts, double thresh, int64_t backoff_deg,
                             const int64_t *quads, int64_t Q, double tfd_thresh, int64_t *rotated_bonds_out, uint8_t *keep_out) {
  FC_API_LOCK;
  FC_REQUIRE(keep_out != nullptr && counts != nullptr && values != nullptr && T >= 1, "NULL pointer argument");
  int64_t S = 1;
  for (int64_t t = 0; t < T; ++t) {
    FC_REQUIRE(counts[t] >= 0, "counts[%lld] is negative", (long long)t);
    FC_REQUIRE(counts[t] == 0 || S <= ((int64_t)1 << 40) / std::max<int64_t>(counts[t], 1), "the grid has more than 2^40 rows");
    S *= counts[t];
  }
  if (S == 0) {  // the starting structure alone: nothing to compare it with
    keep_out[0] = 1;
    return FC_OK;
  }
  return torsion_scan_impl(base, A, torsions, T, rotmasks, nullptr, S, thresh, backoff_deg, quads, Q, nullptr,
                           rotated_bonds_out, nullptr, tfd_thresh, keep_out, values, counts);
}

int fc_torsion_fingerprint(const double *coords, int64_t N, int64_t A, const int64_t *quads,
                           int64_t Q, double *tf_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && A >= 1 && Q >= 0, "bad shape");
  if (N == 0 || Q == 0) return FC_OK;
  FC_REQUIRE(coords && quads && tf_out, "NULL pointer argument");
  for (int64_t k = 0; k < Q * 4; ++k) FC_REQUIRE(quads[k] >= 0 && quads[k] < A, "quadruplet index out of range");
  FC_TRY(ensure_init());
  DevBuf dc, dq, dtf;
  FC_TRY(upload(dc, coords, (size_t)N * A * 3));
  FC_TRY(upload(dq, quads, (size_t)Q * 4));
  FC_TRY(dtf.reserve((size_t)N * Q * sizeof(double)));
  FC_TRY(launch_torsion_fingerprint(dc.as<double>(), N, A, dq.as<int64_t>(), Q, dtf.as<double>()));
  FC_TRY(d2h(tf_out, dtf.p, (size_t)N * Q * sizeof(double)));
  return sync();
}

int fc_tfd_simbits(const double *tf, int64_t N, int64_t Q, double thresh, int64_t row_begin,
                   int64_t row_end, uint64_t *bits_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && Q >= 0, "bad shape");
  FC_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= N, "bad row range");
  if (row_end == row_begin) return FC_OK;
  FC_REQUIRE(bits_out && (tf || Q == 0), "NULL pointer argument");
  FC_TRY(ensure_init());
  const int64_t W = ceil_div(N, 64);
  DevBuf dtf, dbits;
  FC_TRY(upload(dtf, tf, (size_t)N * Q));
  const size_t bytes = (size_t)(row_end - row_begin) * W * sizeof(uint64_t);
  FC_TRY(dbits.reserve(bytes));
  FC_TRY(launch_tfd_simbits(dtf.as<double>(), N, Q, thresh, row_begin, row_end,
                            dbits.as<uint64_t>(), W));
  FC_TRY(d2h(bits_out, dbits.p, bytes));
  return sync();
}

int fc_tfd_first_match(const double *tf, int64_t N, int64_t Q, double thresh, int64_t *first_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && Q >= 0, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(first_out && (tf || Q == 0), "NULL pointer argument");
  if (Q > 128) return set_error(FC_E_LIMIT, "Q=%lld fingerprints exceed 128 (NumPy's summation order changes there)", (long long)Q);
  FC_TRY(ensure_init());
  const int64_t Npad = ceil_div(N, 64) * 64;
  // fingerprint-major copy so that consecutive columns are contiguous: made on the device (on the
  // host the 13 M strided stores of a 1.7 M x 8 matrix cost more than the first-match kernel)
  DevBuf draw, dT, dfm;
  FC_TRY(upload(draw, tf, (size_t)N * (size_t)std::max<int64_t>(Q, 1)));
  FC_TRY(dT.reserve((size_t)std::max<int64_t>(Q, 1) * Npad * sizeof(double)));
  FC_TRY(launch_transpose_pad(draw.as<double>(), N, Q, Npad, dT.as<double>()));
  FC_TRY(dfm.reserve((size_t)N * sizeof(int64_t)));
  DevBuf dtfF;
  FC_TRY(dtfF.reserve((size_t)std::max<int64_t>(std::min<int64_t>(Q, 8), 1) * Npad * sizeof(float)));
  FC_TRY(launch_tfd_first_match(dT.as<double>(), N, Npad, Q, thresh, dfm.as<int64_t>(), dtfF.as<float>()));
  FC_TRY(d2h(first_out, dfm.p, (size_t)N * sizeof(int64_t)));
  return sync();
}

int fc_tfd_ladder_from_first_match(const int64_t *first_match, int64_t N, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(first_match && mask_out, "NULL pointer argument");
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < N; ++i)
    FC_REQUIRE(first_match[i] == -1 || (first_match[i] > i && first_match[i] < N), "first_match[%lld] invalid", (long long)i);
  const auto t1 = std::chrono::steady_clock::now();
  // with a device at hand the ladder runs there (a pure host function otherwise: the CPU tests call it without a GPU)
  DevBuf dfm;
  const int64_t *fm_dev = nullptr;
  if (ctx().ready && N >= 20000) {
    FC_TRY(ensure_init());  // (the calling thread's current device: HIP keeps it per thread)
    FC_TRY(upload(dfm, first_match, (size_t)N));
    fm_dev = dfm.as<int64_t>();
  }
  const int rc = tfd_ladder_from_first_match(first_match, N, mask_out, fm_dev);
  if (getenv("FC_DEBUG"))
    fprintf(stderr, "[fc] fc_tfd_ladder_from_first_match: validation %.1f ms, ladder incl. tear-down %.1f ms\n",
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return rc;
}

int fc_tfd_prune(const double *tf, int64_t N, int64_t Q, double thresh, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0 && Q >= 0, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(mask_out != nullptr, "NULL pointer argument");
  std::vector<int64_t> fm((size_t)N);
  FC_TRY(fc_tfd_first_match(tf, N, Q, thresh, fm.data()));
  return fc_tfd_ladder_from_first_match(fm.data(), N, mask_out);
}

int fc_debug_pyset_order_ints(const int64_t *keys, int64_t n, int64_t *order_out, int64_t *n_out) {
  FC_API_LOCK;
  FC_REQUIRE(n >= 0 && (keys || n == 0) && order_out && n_out, "bad arguments");
  for (int64_t k = 0; k < n; ++k) FC_REQUIRE(keys[k] >= 0, "keys must be non-negative");
  std::vector<int64_t> o;
  pyset_order_ints(keys, n, o);
  for (size_t k = 0; k < o.size(); ++k) order_out[k] = o[k];
  *n_out = (int64_t)o.size();
  return FC_OK;
}

int fc_debug_tfd_ladder_emulate(const int64_t *first_match, int64_t N, uint8_t *mask_out) {
  FC_API_LOCK;
  FC_REQUIRE(N >= 0, "bad shape");
  if (N == 0) return FC_OK;
  FC_REQUIRE(first_match && mask_out, "NULL pointer argument");
  for (int64_t i = 0; i < N; ++i)
    FC_REQUIRE(first_match[i] == -1 || (first_match[i] > i && first_match[i] < N), "first_match[%lld] invalid", (long long)i);
  try {
    return tfd_ladder_emulate_device(first_match, N, mask_out);
  } catch (const std::bad_alloc &) {
    return set_error(FC_E_NOMEM, "out of host memory");
  }
}

int fc_debug_pyset_order_pairs_device(const int64_t *pairs, int64_t n, int64_t *order_out) {
  FC_API_LOCK;
  FC_REQUIRE(n >= 0 && (n == 0 || (pairs && order_out)), "bad arguments");
  FC_TRY(ensure_init());
  return pyset_order_pairs_device(pairs, n, order_out);
}

int fc_debug_pyset_order_pairs(const int64_t *pairs, int64_t n, int64_t *order_out, int64_t *n_out) {
  FC_API_LOCK;
  FC_REQUIRE(n >= 0 && (pairs || n == 0) && order_out && n_out, "bad arguments");
  std::vector<int64_t> o;
  pyset_order_pairs(pairs, n, o);
  for (size_t k = 0; k < o.size(); ++k) order_out[k] = o[k];
  *n_out = (int64_t)o.size();
  return FC_OK;
}

int fc_cartesian_product_i64(const int64_t *values, const int64_t *counts, int64_t T, int64_t *out) {
  FC_REQUIRE(T >= 1 && counts != nullptr, "at least one array");
  int64_t rows = 1, total = 0;
  for (int64_t t = 0; t < T; ++t) {
    FC_REQUIRE(counts[t] >= 0, "negative length");
    FC_REQUIRE(counts[t] == 0 || rows <= (int64_t)1 << 40, "product of the lengths too large");
    rows *= counts[t];
    total += counts[t];
  }
  if (rows == 0) return FC_OK;
  FC_REQUIRE(values != nullptr && out != nullptr && total > 0, "NULL pointer argument");
  cartesian_rows<int64_t>(values, counts, T, out);
  return FC_OK;
}

int fc_cartesian_product_f64(const double *values, const int64_t *counts, int64_t T, double *out) {
  FC_REQUIRE(T >= 1 && counts != nullptr, "at least one array");
  int64_t rows = 1, total = 0;
  for (int64_t t = 0; t < T; ++t) {
    FC_REQUIRE(counts[t] >= 0, "negative length");
    FC_REQUIRE(counts[t] == 0 || rows <= (int64_t)1 << 40, "product of the lengths too large");
    rows *= counts[t];
    total += counts[t];
  }
  if (rows == 0) return FC_OK;
  FC_REQUIRE(values != nullptr && out != nullptr && total > 0, "NULL pointer argument");
  cartesian_rows<double>(values, counts, T, out);
  return FC_OK;
}

int fc_xyz_write(const char *path, const char *const *atoms, int64_t A, const double *coords,
                 int64_t N, const char *label, int mode) {
  FC_API_LOCK;
  FC_REQUIRE(path && atoms && label && (coords || N == 0), "NULL pointer argument");
  FC_REQUIRE(A >= 1 && N >= 0 && (mode == 0 || mode == 1), "bad arguments");
  for (int64_t a = 0; a < A; ++a) FC_REQUIRE(atoms[a] != nullptr, "atoms[%lld] is NULL", (long long)a);
  return xyz_write(path, atoms, A, coords, N, label, mode);
}

int fc_xyz_scan(const char *path, int64_t *N_out, int64_t *A_out) {
  FC_API_LOCK;
  FC_REQUIRE(path && N_out && A_out, "NULL pointer argument");
  return xyz_read(path, N_out, A_out, nullptr, nullptr);
}

int fc_xyz_read(const char *path, int64_t N, int64_t A, char *atoms_out, double *coords_out) {
  FC_API_LOCK;
  FC_REQUIRE(path && atoms_out && coords_out, "NULL pointer argument");
  FC_REQUIRE(N >= 0 && A >= 0, "bad shape");
  return xyz_read(path, &N, &A, atoms_out, coords_out);
}

// ---- many prunes in flight -----------------------------------------------------------
// Enqueues the prunes work[0..n) -- counters reset, screen, refine, level buckets, ladder, copy
// of the survivor words + counters into pinned slot r -- and waits ONCE.  Every prune runs in
// full and delivers its mask words to host memory; what goes away is the host round trip
// between two prunes (~45 us of sync wake-up and launch latency on an idle GPU).
//
// overlap: all screens go, in order, to one stream, so two screens never share the chip and
// their event durations stay those of a kernel that has the matrix pipes to itself; verdict,
// refine, level buckets, ladder and result copy of prune r (eight small launches: ~110 us alone,
// up to 250 us beside a screen that holds every workgroup slot) go to stream r % kPruneLanes of
// three others and run beside the screens of prunes r+1 and r+2 -- with two lanes that chain,
// not the screen, set the pace (0.202 ms per prune at a 0.165 ms screen).  A workspace may
// therefore appear again only a multiple of kPruneLanes places later (same lane: ordered on that
// lane's stream); fc_prune_rmsd_many passes distinct ensembles, the bench hook cycles through an
// ensemble and its twins.
// The caller reads the slots with ladder_collect(work[r], r, ..., stride).
constexpr int kPruneLanes = 3;
static int prune_pipeline(fc_ensemble *const *work, int64_t n, double max_rmsd, double max_dev,
                          int64_t min_per_group, bool overlap, int64_t stride, double *screen_ms_sum,
                          double *total_ms) {
  Context &c = ctx();
  std::vector<hipEvent_t> &ev = c.ev_pool;  // 4 per prune: around the screen kernel, counters reset, screen phase done
  while ((int64_t)ev.size() < 4 * n + 2 + kPruneLanes) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreate(&e));
    ev.push_back(e);
  }
  hipEvent_t const ev_begin = ev[4 * n], ev_end = ev[4 * n + 1];
  std::vector<hipEvent_t> &dep = c.ev_dep_pool;  // 2 per prune: counters reset -> screen, screen -> rest of the prune
  while ((int64_t)dep.size() < 2 * n) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    dep.push_back(e);
  }
  FC_TRY(pinned_reserve((size_t)n * (size_t)stride * sizeof(uint64_t)));
  static const int64_t stride_ev = [] {
    const char *v = getenv("FC_BENCH_EVENT_STRIDE");
    const long k = v ? std::strtol(v, nullptr, 10) : 8;
    return (int64_t)(k >= 1 && k <= 4096 ? k : 8);
  }();
  hipStream_t const home = c.stream;
  struct Restore {
    Context &c;
    hipStream_t s;
    ~Restore() { c.stream = s; }
  } restore{c, home};
  const bool lanes = overlap && n > 1;
  if (lanes) FC_TRY(side_streams());
  hipStream_t const s_screen = c.s_screen, s_lane[kPruneLanes] = {c.s_lane[0], c.s_lane[1], c.s_lane[2]};
  // everything enqueued here is ordered behind what the home stream already holds (also what
  // makes pool blocks released by earlier calls safe to reuse on the other streams)
  FC_HIP_TRY(hipEventRecord(ev_begin, home));
  if (lanes)
    for (hipStream_t s : {s_screen, s_lane[0], s_lane[1], s_lane[2]}) FC_HIP_TRY(hipStreamWaitEvent(s, ev_begin, 0));
  auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_enqueue0 = now_s();
  for (int64_t r = 0; r < n; ++r) {
    fc_ensemble *e = work[r];
    hipStream_t const tail = lanes ? s_lane[r % kPruneLanes] : home, scr = lanes ? s_screen : home;
    c.stream = tail;
    FC_TRY(ensemble_shard(e, 0, 1, default_row_block()));
    FC_HIP_TRY(hipMemsetAsync(e->counters.p, 0, kCounters * sizeof(uint64_t), tail));
    if (lanes) {
      FC_HIP_TRY(hipEventRecord(dep[2 * r], tail));
      FC_HIP_TRY(hipStreamWaitEvent(scr, dep[2 * r], 0));
    }
    c.stream = scr;
    // timing events around the screen kernel of every `stride`-th prune only: the pair costs the
    // screen stream ~14 us (0.542 -> 0.528 ms per step when all 200 prunes carry it)
    e->lean = true;  // consumer: the one-launch pair ladder (a prune it declines is redone by the caller)
    const bool timed = screen_ms_sum != nullptr && r % stride_ev == 0;
    if (timed) FC_HIP_TRY(hipEventRecord(ev[4 * r], scr));
    c.mark_after_screen = timed ? ev[4 * r + 1] : nullptr;  // the launcher records it right behind the screen kernel
    // with lanes the launcher itself moves to the tail stream behind its main kernel: verdict and gated fp64 screen
    // run there (27 us between two screens on the screen stream otherwise: tools/attic/step_gaps.py)
    c.after_main_stream = lanes ? tail : nullptr;
    c.after_main_event = lanes ? dep[2 * r + 1] : nullptr;
    c.optimistic_screen = lanes;
    const int rc_screen = launch_simbits_screen(e, max_rmsd * max_rmsd + kScreenMargin);
    c.mark_after_screen = nullptr;
    c.after_main_stream = nullptr;
    c.after_main_event = nullptr;
    c.optimistic_screen = false;
    const bool moved = c.stream == tail;
    c.stream = scr;
    FC_TRY(rc_screen);
    if (lanes && !moved) {  // (a launch that ended before its main kernel: nothing to move)
      FC_HIP_TRY(hipEventRecord(dep[2 * r + 1], scr));
      FC_HIP_TRY(hipStreamWaitEvent(tail, dep[2 * r + 1], 0));
    }
    c.stream = tail;
    FC_TRY(launch_simbits_refine(e, max_rmsd, max_dev, nullptr, 0.0));
    e->bits_valid = true;
    FC_TRY(ladder_single(e, nullptr, min_per_group, nullptr, nullptr, nullptr, nullptr,
                         e->simq.as<uint64_t>(), false, true, r, stride));
  }
  c.stream = home;
  if (getenv("FC_DEBUG") && n > 8)
    fprintf(stderr, "[fc] prune_pipeline: %lld prunes enqueued in %.3f ms of host time (%.1f us each)\n", (long long)n,
            1e3 * (now_s() - t_enqueue0), 1e6 * (now_s() - t_enqueue0) / (double)n);
  if (lanes)  // the home stream ends behind the last prune of every lane
    for (int l = 0; l < kPruneLanes; ++l) {
      FC_HIP_TRY(hipEventRecord(ev[4 * n + 2 + l], s_lane[l]));
      FC_HIP_TRY(hipStreamWaitEvent(home, ev[4 * n + 2 + l], 0));
    }
  FC_HIP_TRY(hipEventRecord(ev_end, home));
  FC_HIP_TRY(hipEventSynchronize(ev_end));
  if (total_ms) {
    float t = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&t, ev_begin, ev_end));
    *total_ms = t;
  }
  if (screen_ms_sum) {  // mean over the timed prunes
    *screen_ms_sum = 0.0;
    int64_t n_timed = 0;
    for (int64_t r = 0; r < n; r += stride_ev) {
      float a = 0.f;
      FC_HIP_TRY(hipEventElapsedTime(&a, ev[4 * r], ev[4 * r + 1]));
      *screen_ms_sum += a;
      ++n_timed;
    }
    *screen_ms_sum /= (double)std::max<int64_t>(n_timed, 1);
  }
  return FC_OK;
}

int fc_prune_rmsd_many(fc_ensemble *const *ens, int64_t n, double max_rmsd, double max_dev,
                       int64_t min_per_group, uint8_t *const *mask_out, int64_t *survivors_out) {
  FC_API_LOCK;
  FC_REQUIRE(n >= 0 && n <= 4096, "n=%lld outside 0..4096", (long long)n);
  if (n == 0) return FC_OK;
  FC_REQUIRE(ens && mask_out, "NULL pointer argument");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_REQUIRE(min_per_group >= 1, "min_per_group must be >= 1");
  FC_TRY(ensure_init());
  std::vector<fc_ensemble *> sorted(ens, ens + n);
  std::sort(sorted.begin(), sorted.end());
  for (int64_t r = 0; r < n; ++r) {
    FC_REQUIRE(sorted[r] != nullptr, "NULL ensemble in the list");
    FC_REQUIRE(r == 0 || sorted[r] != sorted[r - 1], "the same ensemble appears twice in the list");
    FC_REQUIRE(mask_out[r] != nullptr || ens[r]->N == 0, "mask_out[%lld] is NULL", (long long)r);
  }
  // in flight together: ensembles the one-launch ladder can take; the rest one by one behind them
  std::vector<fc_ensemble *> work;
  std::vector<int64_t> where;
  int64_t stride = 0;
  for (int64_t r = 0; r < n; ++r)
    if (ens[r]->N >= 2 && (size_t)2 * ens[r]->W * sizeof(uint64_t) <= 60 * 1024) {
      work.push_back(ens[r]);
      where.push_back(r);
      stride = std::max(stride, ens[r]->W + 16);
    }
  static const bool overlap = [] {
    const char *v = getenv("FC_PRUNE_LANES");
    return !(v && atoi(v) == 1);
  }();
  std::vector<char> done((size_t)n, 0);
  if (!work.empty()) {
    FC_TRY(prune_pipeline(work.data(), (int64_t)work.size(), max_rmsd, max_dev, min_per_group, overlap,
                          stride, nullptr, nullptr));
    for (size_t q = 0; q < work.size(); ++q) {
      int64_t levels = 0, alive = 0;
      const int64_t r = where[q];
      if (ladder_collect(work[q], (int64_t)q, mask_out[r], &levels, &alive, nullptr, stride)) {
        done[(size_t)r] = 1;
        if (survivors_out) survivors_out[r] = alive;
      }
    }
  }
  for (int64_t r = 0; r < n; ++r) {
    if (done[(size_t)r]) continue;
    int64_t st[6] = {0};
    if (ens[r]->N > 0) FC_TRY(fc_prune_rmsd(ens[r], max_rmsd, max_dev, nullptr, 0.0, min_per_group, mask_out[r], st));
    if (survivors_out) survivors_out[r] = st[5];
  }
  return FC_OK;
}

}  // extern "C"

// ---- sharded prune over the ranks of the communicator (fc_comm.cpp), no host round trips -------
// Prune k works on workspace k&1 (the ensemble and its twin) and on lane k&1: counters reset,
// refine, export of the rank's similar-pair list, the all-gather (RCCL, on its own stream between
// two events) and the ladder replay; all screens go, in order, to the screen stream.  So the ~0.13 ms
// behind a screen run beside the next screen.  overlap = false: everything on the current stream.
static int sharded_pipeline(fc_ensemble *ens, int64_t steps, double max_rmsd, double max_dev,
                            int64_t min_per_group, int64_t row_block, bool overlap, double *screen_ms_mean,
                            double *total_ms) {
  Context &c = ctx();
  const int64_t rank = comm_rank(), world = comm_world();
  const int64_t cap = 1024 + 4 * ens->N / world;  // pairs a rank can send in the fixed-size message
  FC_TRY(side_streams());
  overlap = overlap && steps > 1;
  fc_ensemble *work[2] = {ens, ens};
  if (overlap) FC_TRY(ensemble_twin(ens, &work[1]));
  for (fc_ensemble *w : work) {  // every grow-only buffer reaches its size before the streams fork
    FC_TRY(ensemble_shard(w, rank, world, row_block));
    FC_TRY(w->msg_send.reserve((size_t)(cap + 1) * sizeof(uint64_t)));
    const size_t recv_bytes = (size_t)world * (size_t)(cap + 1) * sizeof(uint64_t);
    if (w->msg_recv.bytes < recv_bytes || !w->msg_recv.p) {  // a new block starts as "every rank sent an empty list"
      FC_TRY(w->msg_recv.reserve(recv_bytes));
      FC_HIP_TRY(hipMemsetAsync(w->msg_recv.p, 0, w->msg_recv.bytes, c.stream));
    }
    FC_TRY(w->gathered.reserve((size_t)std::max<int64_t>(world * cap, 1) * sizeof(uint64_t)));
  }
  std::vector<hipEvent_t> &ev = c.ev_pool;
  while ((int64_t)ev.size() < 2 * steps + 4) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreate(&e));
    ev.push_back(e);
  }
  static const int64_t stride_ev = [] {
    const char *v = getenv("FC_BENCH_EVENT_STRIDE");
    const long k = v ? std::strtol(v, nullptr, 10) : 8;
    return (int64_t)(k >= 1 && k <= 4096 ? k : 8);
  }();
  hipEvent_t const ev_begin = ev[2 * steps], ev_end = ev[2 * steps + 1];
  hipStream_t const home = c.stream;
  struct Restore {
    Context &c;
    hipStream_t s;
    ~Restore() { c.stream = s; }
  } restore{c, home};
  FC_HIP_TRY(hipEventRecord(ev_begin, home));
  for (hipStream_t s : {c.s_screen, c.s_lane[0], c.s_lane[1], c.s_comm}) FC_HIP_TRY(hipStreamWaitEvent(s, ev_begin, 0));
  for (int64_t k = 0; k < steps; ++k) {
    fc_ensemble *e = work[k & 1];
    c.stream = overlap ? c.s_lane[k & 1] : home;
    const bool timed = k % stride_ev == 0;
    FC_TRY(begin_split(e, max_rmsd, max_dev, rank, world, row_block, overlap ? c.s_screen : c.stream,
                       timed ? ev[2 * k] : nullptr, timed ? ev[2 * k + 1] : nullptr));
    FC_TRY(fc_prune_export_pairs_dev(e, e->msg_send.as<uint64_t>(), cap));
    FC_TRY(comm_allgather_dev(e->msg_send.p, e->msg_recv.p, (size_t)(cap + 1) * sizeof(uint64_t), (int)(k & 1)));
    FC_TRY(fc_prune_from_gathered_dev_enqueue(e, e->msg_recv.as<uint64_t>(), world, cap, min_per_group, k, steps));
  }
  c.stream = home;
  if (overlap)
    for (hipStream_t s : {c.s_lane[0], c.s_lane[1], c.s_screen}) {
      FC_HIP_TRY(hipEventRecord(ev[2 * steps + 2], s));
      FC_HIP_TRY(hipStreamWaitEvent(home, ev[2 * steps + 2], 0));
    }
  FC_HIP_TRY(hipEventRecord(ev_end, home));
  FC_HIP_TRY(hipEventSynchronize(ev_end));
  if (total_ms) {
    float t = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&t, ev_begin, ev_end));
    *total_ms = t;
  }
  if (screen_ms_mean) {
    double sum = 0.0;
    int64_t n_timed = 0;
    for (int64_t k = 0; k < steps; k += stride_ev) {
      float a = 0.f;
      FC_HIP_TRY(hipEventElapsedTime(&a, ev[2 * k], ev[2 * k + 1]));
      sum += a;
      ++n_timed;
    }
    *screen_ms_mean = sum / (double)std::max<int64_t>(n_timed, 1);
  }
  return FC_OK;
}

static int64_t owned_pairs(const fc_ensemble *ens) {
  int64_t n = 0;
  const int64_t nb = ceil_div(ens->N, ens->row_block);
  for (int64_t lb = 0, b; (b = global_block(lb, ens->rank, ens->world)) < nb; ++lb)
    for (int64_t i = b * ens->row_block; i < std::min(ens->N, (b + 1) * ens->row_block); ++i) n += ens->N - 1 - i;
  return n;
}

// dense similarity (a rank's candidate queue overflowed or its list did not fit the message): one
// all-gather of the (N,) mask per ladder level; every rank takes this path together because every
// rank saw the same gathered headers
static int sharded_levels_fallback(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t min_per_group,
                                   int64_t row_block, uint8_t *mask_out, int64_t *stats) {
  const int64_t rank = comm_rank(), world = comm_world(), N = ens->N;
  int64_t st[6] = {0};
  FC_TRY(fc_prune_rmsd_begin(ens, max_rmsd, max_dev, nullptr, 0.0, rank, world, row_block, st));
  std::vector<uint8_t> mask((size_t)N, 1), mine((size_t)N), all((size_t)N * (size_t)world);
  for (int64_t k : kLadder) {
    int64_t alive = 0;
    for (uint8_t m : mask) alive += m;
    if (!(k == 1 || min_per_group * k < alive)) continue;
    FC_TRY(fc_prune_level(ens, k, mask.data(), mine.data()));
    FC_TRY(fc_allgather_mask(mine.data(), N, all.data()));
    for (int64_t i = 0; i < N; ++i) {
      uint8_t m = 1;
      for (int64_t r = 0; r < world; ++r) m = std::min(m, all[(size_t)r * N + i]);
      mask[(size_t)i] = m;
    }
  }
  int64_t alive = 0;
  for (int64_t i = 0; i < N; ++i) {
    mask_out[i] = mask[(size_t)i];
    alive += mask[(size_t)i];
  }
  if (stats) {
    for (int k = 0; k < 5; ++k) stats[k] = st[k];
    stats[5] = alive;
  }
  return FC_OK;
}

static int sharded_collect(fc_ensemble *ens, int64_t steps, bool overlap, uint8_t *mask_out, int64_t *stats,
                           bool *all_ok, int64_t *units = nullptr) {
  const int64_t W = ens->W;
  *all_ok = true;
  int64_t survivors = 0;
  for (int64_t k = 0; k < steps; ++k)
    if (!ladder_collect(ens, k, k == steps - 1 ? mask_out : nullptr, nullptr, &survivors, nullptr)) *all_ok = false;
  if (*all_ok && stats) {
    const unsigned long long *local = static_cast<const unsigned long long *>(ctx().pinned) +
                                      (size_t)steps * (size_t)(W + 16) + (size_t)(steps - 1) * 8;
    stats[0] = owned_pairs(ens);
    stats[1] = (int64_t)local[1];
    stats[2] = (int64_t)local[2];
    stats[3] = (int64_t)local[3];
    stats[4] = 0;
    stats[5] = survivors;
  }
  (void)overlap;
  if (*all_ok && units) {  // the subset stage of this rank's lean fp32 screen in the last prune (counters ride behind the mask words)
    const uint64_t *cnt_last = static_cast<const uint64_t *>(ctx().pinned) + (size_t)(steps - 1) * (size_t)(W + 16) + W;
    units[0] = (int64_t)cnt_last[13];
    units[1] = (int64_t)cnt_last[15];
  }
  return FC_OK;
}

extern "C" {

int fc_prune_rmsd_sharded(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t min_per_group,
                          int64_t row_block, uint8_t *mask_out, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && mask_out, "NULL pointer argument");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0 && min_per_group >= 1, "bad arguments");
  FC_TRY(ensure_init());
  if (ens->N == 0) return FC_OK;
  if (row_block <= 0) row_block = default_row_block();
  bool ok = false;
  const int64_t world = comm_world();
  const int64_t cap = 1024 + 4 * ens->N / world;
  const bool device_path = (uint64_t)world * (uint64_t)cap <= kPairLadderCap && (size_t)2 * ens->W * sizeof(uint64_t) <= 60 * 1024;
  if (device_path) {
    double ms = 0.0;
    FC_TRY(sharded_pipeline(ens, 1, max_rmsd, max_dev, min_per_group, row_block, false, &ms, nullptr));
    FC_TRY(sharded_collect(ens, 1, false, mask_out, stats, &ok));
    if (ok && stats) stats[4] = (int64_t)(ms * 1e6);
  }
  if (!ok) FC_TRY(sharded_levels_fallback(ens, max_rmsd, max_dev, min_per_group, row_block, mask_out, stats));
  return FC_OK;
}

int fc_bench_prune_rmsd_sharded(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps, int overlap,
                                double *ms_screen_kernel, double *ms_step, uint8_t *mask_out, int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && reps >= 1 && reps <= 1024, "bad arguments");
  FC_REQUIRE(max_rmsd > 0.0 && max_dev > 0.0, "thresholds must be positive");
  FC_TRY(ensure_init());
  FC_REQUIRE(ens->N > 0, "empty ensemble");
  const int64_t row_block = default_row_block();
  double ms = 0.0, total = 0.0;
  FC_TRY(sharded_pipeline(ens, reps, max_rmsd, max_dev, 20, row_block, overlap != 0, &ms, &total));
  bool ok = false;
  std::vector<uint8_t> scratch;
  if (!mask_out) {
    scratch.resize((size_t)ens->N);
    mask_out = scratch.data();
  }
  if (stats) stats[6] = stats[7] = 0;  // EIGHT stats, as fc_bench_prune_rmsd
  FC_TRY(sharded_collect(ens, reps, overlap != 0, mask_out, stats, &ok, stats ? stats + 6 : nullptr));
  if (!ok) FC_TRY(sharded_levels_fallback(ens, max_rmsd, max_dev, 20, row_block, mask_out, stats));
  if (ms_screen_kernel) *ms_screen_kernel = ms;
  if (ms_step) *ms_step = total / (double)reps;
  return FC_OK;
}

}  // extern "C"

extern "C" {

// ---- bench hook ----------------------------------------------------------------------
int fc_screen_last_kind(void) { return last_screen_kind(); }

int fc_prune_conventions(int drop_later) {
  FC_API_LOCK;
  prune_conventions_set(drop_later);
  return FC_OK;
}

int fc_debug_mfma_f16_model(int64_t trials, int64_t *flags_out, double *worst_out) {
  FC_API_LOCK;
  FC_REQUIRE(flags_out && worst_out && trials >= 0 && trials <= (1 << 24), "NULL output or trials outside 0..2^24");
  FC_TRY(ensure_init());
  return h2_model_report(trials, flags_out, worst_out);
}

int fc_debug_h2_covariance(fc_ensemble *ens, int64_t ib, int64_t jb, float *B_out, double *scale_out, double *entry_bound_out) {
  FC_API_LOCK;
  FC_REQUIRE(ens && B_out && scale_out && entry_bound_out, "NULL argument");
  FC_REQUIRE(ib >= 0 && jb >= 0 && ib % 16 == 0 && jb % 16 == 0 && ib + 16 <= ens->Npad && jb + 16 <= ens->Npad,
             "tile origin must be a multiple of 16 inside the padded ensemble");
  FC_TRY(ensure_init());
  FC_REQUIRE(ens->epoch == ctx().epoch, "ensemble belongs to a context that was shut down");
  double scale = 0.0;
  FC_TRY(ensure_h2_operands(ens, &scale));
  *scale_out = scale;
  const int64_t KS2 = (ens->A + 31) / 32;
  *entry_bound_out = kabsch_h2_entry_bound(KS2);
  if (scale == 0.0) return FC_OK;  // does not apply (more than 128 atoms, degenerate norms)
  DevBuf out;
  FC_TRY(out.reserve(256 * 9 * sizeof(float)));
  FC_TRY(launch_h2_cov_tile(ens, ib, jb, out.as<float>()));
  FC_TRY(d2h(B_out, out.p, 256 * 9 * sizeof(float)));
  return sync();
}

int fc_screen_select(int kind) {
  FC_API_LOCK;
  FC_REQUIRE(kind == 0 || kind == 16 || kind == 32 || kind == 64, "kind must be 0 (automatic), 16, 32 or 64");
  screen_select(kind);
  return FC_OK;
}

int fc_bench_refine(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps, double *ms_refine,
                    int64_t *n_candidates) {
  FC_API_LOCK;
  FC_REQUIRE(ens && reps >= 1 && reps <= 4096, "bad arguments");
  FC_TRY(ensure_init());
  FC_TRY(ensemble_shard(ens, 0, 1, default_row_block()));
  Context &c = ctx();
  // one screen fills the candidate-pair queue; then the exact refine alone, `reps` times over that queue
  ens->lean = true;
  FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, kCounters * sizeof(uint64_t), c.stream));
  FC_TRY(launch_simbits_screen(ens, max_rmsd * max_rmsd + kScreenMargin));
  {  // the launcher chooses the long-queue kernels from what the host last saw of this ensemble's queue
    unsigned long long h0[8] = {0};
    FC_TRY(d2h(h0, ens->counters.p, sizeof h0));
    FC_TRY(sync());
    note_candidates(ens, h0[6], ens->last_similar < 0 ? 0 : (unsigned long long)ens->last_similar);
  }
  std::vector<hipEvent_t> &ev = c.ev_pool;
  while ((int64_t)ev.size() < 2 * reps) {
    hipEvent_t e = nullptr;
    FC_HIP_TRY(hipEventCreate(&e));
    ev.push_back(e);
  }
  auto *cnt = reinterpret_cast<unsigned long long *>(ens->counters.p);
  for (int64_t r = 0; r < reps; ++r) {
    FC_HIP_TRY(hipMemsetAsync(cnt + 1, 0, 3 * sizeof(uint64_t), c.stream));  // refined / similar / grey
    FC_HIP_TRY(hipEventRecord(ev[2 * r], c.stream));
    FC_TRY(launch_simbits_refine(ens, max_rmsd, max_dev, nullptr, 0.0));
    FC_HIP_TRY(hipEventRecord(ev[2 * r + 1], c.stream));
  }
  unsigned long long h[32] = {0};
  FC_TRY(d2h(h, ens->counters.p, sizeof h));
  FC_TRY(sync());
  if (getenv("FC_DEBUG") && h[22])  // (tuning build FC_RB_TIMELINE: sums over all launches since the counters were cleared)
    fprintf(stderr, "[fc] refine buckets: %llu wave-items, %llu with pairs; mean ticks (100 MHz) staging %.1f, compute %.1f\n", h[22], h[23],
            (double)h[20] / (double)h[22], (double)h[21] / (double)h[22]);
  if (getenv("FC_DEBUG") && h[27])
    fprintf(stderr, "[fc] refine buckets, wave 0 per item (%llu rounds): covariance pass %.1f ticks, polynomial + rotation %.1f, + deviation pass %.1f\n",
            h[27], (double)h[26] / (double)h[27], (double)h[24] / (double)h[27], (double)h[25] / (double)h[27]);
  double sum = 0.0;
  for (int64_t r = 0; r < reps; ++r) {
    float ms = 0.f;
    FC_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
    sum += ms;
  }
  if (ms_refine) *ms_refine = sum / (double)reps;
  if (n_candidates) *n_candidates = (int64_t)h[1];
  if (h[6] > (unsigned long long)ens->pairq_cap)
    return set_error(FC_E_LIMIT, "the candidate-pair queue overflowed (%llu > %lld): this probe times the pair refine only",
                     h[6], (long long)ens->pairq_cap);
  return FC_OK;
}

int fc_bench_prune_rmsd(fc_ensemble *ens, double max_rmsd, double max_dev, int64_t reps,
                        double *ms_simbits_kernel, double *ms_step, uint8_t *mask_out,
                        int64_t *stats) {
  FC_API_LOCK;
  FC_REQUIRE(ens && reps >= 1 && reps <= 4096, "bad arguments");
  FC_TRY(ensure_init());
  FC_TRY(ensemble_shard(ens, 0, 1, default_row_block()));
  // `reps` prunes of the same resident ensemble through prune_pipeline.  FC_BENCH_LANES=1:
  // strictly one after another; default: odd prunes use a second workspace (bit rows, queues,
  // counters, ladder words) over the same coordinates, so that the small kernels of one prune
  // can run beside the screen of the next.
  static const bool two_lanes = [] {
    const char *v = getenv("FC_BENCH_LANES");
    return !(v && atoi(v) == 1);
  }();
  const bool lanes = two_lanes && reps > 1;
  const int64_t stride = ens->W + 16;
  fc_ensemble *ws[kPruneLanes] = {ens, ens, ens};
  if (lanes) {
    const bool fresh = !ens->twin || !ens->twin->twin;
    for (int l = 1; l < kPruneLanes; ++l) FC_TRY(ensemble_twin(ws[l - 1], &ws[l]));  // a chain of twins over the same coordinates
    // one whole prune per workspace on the home stream: every grow-only buffer reaches its
    // size here, so no block changes hands while several streams are in flight
    if (fresh) FC_TRY(prune_pipeline(ws, kPruneLanes, max_rmsd, max_dev, 20, false, stride, nullptr, nullptr));
  }
  std::vector<fc_ensemble *> work((size_t)reps);
  for (int64_t r = 0; r < reps; ++r) work[(size_t)r] = ws[lanes ? r % kPruneLanes : 0];
  double t_kernel = 0.0, total = 0.0;
  FC_TRY(prune_pipeline(work.data(), reps, max_rmsd, max_dev, 20, lanes, stride, &t_kernel, &total));
  int64_t levels = 0, survivors = 0;
  unsigned long long cnt[8] = {0};
  bool redo = false;
  for (int64_t r = 0; r < reps; ++r)
    if (!ladder_collect(ens, r, mask_out, &levels, &survivors, cnt, stride)) redo = true;
  Context &c = ctx();
  if (redo) {  // dense similarity: the pair ladder declined; one synchronous prune through the bit matrix
    ens->lean = false;
    FC_HIP_TRY(hipMemsetAsync(ens->counters.p, 0, kCounters * sizeof(uint64_t), c.stream));
    FC_TRY(launch_simbits_screen(ens, max_rmsd * max_rmsd + kScreenMargin));
    FC_TRY(launch_simbits_refine(ens, max_rmsd, max_dev, nullptr, 0.0));
    FC_TRY(ladder_single(ens, ens->bits.as<uint64_t>(), 20, mask_out, &levels, &survivors, cnt,
                         ens->simq.as<uint64_t>(), false, true));
  }
  if (getenv("FC_DEBUG")) {
    const uint64_t *cnt_host = static_cast<const uint64_t *>(c.pinned) + (size_t)(reps - 1) * (size_t)stride + ens->W;
    fprintf(stderr, "[fc] bench prune: candidates %llu, similar %llu, screen units the subset stage queued: %llu\n",
            (unsigned long long)cnt_host[1], (unsigned long long)cnt_host[2], (unsigned long long)cnt_host[13]);
  }
  if (ms_simbits_kernel) *ms_simbits_kernel = t_kernel;  // mean over the timed prunes
  if (ms_step) *ms_step = total / (double)reps;
  if (stats) {
    stats[0] = ens->N * (ens->N - 1) / 2;
    stats[1] = (int64_t)cnt[1];
    stats[2] = (int64_t)cnt[2];
    stats[3] = (int64_t)cnt[3];
    stats[4] = levels;
    stats[5] = survivors;
    // the subset stage of the lean fp32 screen in the last prune: units it queued for the full test
    // (0 with the single-stage kernels), and whether its sample found similarity dense
    const uint64_t *cnt_last = static_cast<const uint64_t *>(c.pinned) + (size_t)(reps - 1) * (size_t)stride + ens->W;
    stats[6] = redo ? 0 : (int64_t)cnt_last[13];
    stats[7] = redo ? 0 : (int64_t)cnt_last[15];
  }
  return FC_OK;
}

}  // extern "C"
