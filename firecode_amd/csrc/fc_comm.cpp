// fc_comm.cpp -- the one exchange the sharded path has: an all-gather over
// the GPUs of one node, RCCL over xGMI, bound at RUN time.
//
// libfc_hip.so has no link-time dependency on RCCL: librccl.so is dlopen()ed by
// fc_comm_unique_id / fc_comm_init, so a single-GPU user (or a host without RCCL) never
// touches it.  No PyTorch anywhere: ranks are ordinary processes, one per GPU, started by
// any launcher before their first GPU call; rank 0 makes the ncclUniqueId and hands it to
// the others out of band (a file or an environment variable: firecode_amd/dist.py).
//
// Every collective is enqueued on Context::s_comm (highest stream priority: the all-pairs
// screen fills every workgroup slot of the chip) between two events, so the caller's
// stream waits for it without a host round trip.  One communicator, one stream, calls in
// program order: the order is the same on every rank by construction.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only; every symbol is resolved with dlsym

#include <algorithm>
#include <memory>

#include "fc_common.h"

namespace fc {
namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  Rccl api;
  ncclComm_t comm = nullptr;
  int64_t rank = 0, world = 1;
  bool ready = false;
  uint64_t epoch = 0;
  DevBuf stage_send, stage_recv;  // device staging of the host-buffer collectives
};

Comm &comm() {
  static Comm *c = new Comm;  // never destroyed: its DevBufs must not outlive the pool otherwise
  return *c;
}

int load_rccl(Rccl &r) {
  if (r.lib) return FC_OK;
  const char *names[] = {getenv("FC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  std::string tried;
  for (const char *n : names) {
    if (!n || !*n) continue;
    r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
    const char *err = dlerror();  // ONE call: glibc clears the message when it is read
    tried += std::string(n) + ": " + (err ? err : "?") + "; ";
  }
  if (!r.lib) return set_error(FC_E_NODEVICE, "RCCL is not available (%s)", tried.c_str());
#define FC_SYM(field, name)                                                                     \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));                            \
  if (!r.field) return set_error(FC_E_NODEVICE, "librccl lacks the symbol %s", name)
  FC_SYM(GetUniqueId, "ncclGetUniqueId");
  FC_SYM(CommInitRank, "ncclCommInitRank");
  FC_SYM(CommDestroy, "ncclCommDestroy");
  FC_SYM(AllGather, "ncclAllGather");
  FC_SYM(GetErrorString, "ncclGetErrorString");
#undef FC_SYM
  return FC_OK;
}

#define FC_NCCL_TRY(expr)                                                                        \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess)                                                                       \
      return set_error(FC_E_HIP, "%s failed: %s", #expr, comm().api.GetErrorString(_r));         \
  } while (0)

}  // namespace

// test hook (fc_debug_comm_loopback): act as `rank` of `world` without a communicator -- the
// all-gather writes only this rank's slot of the receive buffer, the other slots keep what earlier
// calls (the other logical ranks, run one after the other on the same buffers) left there
static int64_t g_loop_rank = -1, g_loop_world = 0;

bool comm_ready() { return comm().ready && comm().epoch == ctx().epoch; }
int comm_rank() { return g_loop_world > 0 ? (int)g_loop_rank : comm_ready() ? (int)comm().rank : 0; }
int comm_world() { return g_loop_world > 0 ? (int)g_loop_world : comm_ready() ? (int)comm().world : 1; }

// All-gather of `bytes` bytes per rank between device buffers, ordered behind everything the
// current stream (ctx().stream) holds and in front of everything enqueued on it afterwards;
// lane selects the event pair (two collectives may be in flight, one per lane of the sharded
// steps).  world == 1 without a communicator: a device copy.
int comm_allgather_dev(const void *send_dev, void *recv_dev, size_t bytes, int lane) {
  Comm &k = comm();
  Context &c = ctx();
  if (g_loop_world > 0) {
    FC_HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv_dev) + (size_t)g_loop_rank * bytes, send_dev, bytes,
                              hipMemcpyDeviceToDevice, c.stream));
    return FC_OK;
  }
  if (!comm_ready()) {
    if (k.ready) return set_error(FC_E_INVALID, "the communicator belongs to a context that was shut down");
    FC_HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, bytes, hipMemcpyDeviceToDevice, c.stream));
    return FC_OK;
  }
  FC_TRY(side_streams());
  hipEvent_t const ev = c.ev_comm[lane & 1];
  FC_HIP_TRY(hipEventRecord(ev, c.stream));
  FC_HIP_TRY(hipStreamWaitEvent(c.s_comm, ev, 0));
  FC_NCCL_TRY(k.api.AllGather(send_dev, recv_dev, bytes, ncclUint8, k.comm, c.s_comm));
  FC_HIP_TRY(hipEventRecord(ev, c.s_comm));
  FC_HIP_TRY(hipStreamWaitEvent(c.stream, ev, 0));
  return FC_OK;
}

void comm_teardown() {
  Comm &k = comm();
  if (k.ready && k.comm && k.api.CommDestroy) (void)k.api.CommDestroy(k.comm);
  k.comm = nullptr;
  k.ready = false;
  k.rank = 0;
  k.world = 1;
  k.stage_send.release();
  k.stage_recv.release();
}

}  // namespace fc

using namespace fc;

extern "C" {

int fc_comm_unique_id(uint8_t *id_out) {
  FC_API_LOCK;
  FC_REQUIRE(id_out != nullptr, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == FC_COMM_ID_BYTES, "FC_COMM_ID_BYTES must match ncclUniqueId");
  FC_TRY(load_rccl(comm().api));  // first: the loader's own failure modes are testable without a device
  FC_TRY(ensure_init());
  ncclUniqueId id;
  FC_NCCL_TRY(comm().api.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return FC_OK;
}

int fc_comm_init(int64_t rank, int64_t world, const uint8_t *id) {
  FC_API_LOCK;
  FC_REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad rank/world %lld/%lld", (long long)rank,
             (long long)world);
  FC_REQUIRE(id != nullptr, "id is NULL");
  FC_TRY(ensure_init());
  Comm &k = comm();
  if (k.ready) comm_teardown();
  FC_TRY(load_rccl(k.api));
  FC_TRY(side_streams());
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  FC_NCCL_TRY(k.api.CommInitRank(&k.comm, (int)world, uid, (int)rank));
  k.rank = rank;
  k.world = world;
  k.epoch = ctx().epoch;
  k.ready = true;
  return FC_OK;
}

int fc_comm_destroy(void) {
  FC_API_LOCK;
  if (ctx().ready) {
    (void)hipStreamSynchronize(ctx().stream);
    if (ctx().s_comm) (void)hipStreamSynchronize(ctx().s_comm);
  }
  comm_teardown();
  return FC_OK;
}

int fc_comm_info(int64_t *rank, int64_t *world) {
  FC_API_LOCK;
  if (rank) *rank = comm_rank();
  if (world) *world = comm_world();
  return FC_OK;
}

int fc_debug_comm_loopback(int64_t rank, int64_t world) {
  FC_API_LOCK;
  FC_REQUIRE(world == 0 || (world >= 1 && world <= 64 && rank >= 0 && rank < world), "bad rank/world");
  g_loop_rank = world > 0 ? rank : -1;
  g_loop_world = world > 0 ? world : 0;
  return FC_OK;
}

int fc_allgather_u8_dev(const uint8_t *send_dev, uint8_t *recv_dev, int64_t bytes_per_rank) {
  FC_API_LOCK;
  FC_REQUIRE(bytes_per_rank >= 0, "negative size");
  if (bytes_per_rank == 0) return FC_OK;
  FC_REQUIRE(send_dev && recv_dev, "NULL pointer argument");
  FC_TRY(ensure_init());
  return comm_allgather_dev(send_dev, recv_dev, (size_t)bytes_per_rank, 0);
}

// SURVEY 8b's fc_allgather_mask: host buffers, blocking
int fc_allgather_mask(const uint8_t *mask_local, int64_t n_local, uint8_t *mask_global) {
  FC_API_LOCK;
  FC_REQUIRE(n_local >= 0, "negative size");
  if (n_local == 0) return FC_OK;
  FC_REQUIRE(mask_local && mask_global, "NULL pointer argument");
  FC_TRY(ensure_init());
  Comm &k = comm();
  const int64_t world = comm_world();
  FC_TRY(k.stage_send.reserve((size_t)n_local));
  FC_TRY(k.stage_recv.reserve((size_t)n_local * (size_t)world));
  if (g_loop_world > 0) FC_HIP_TRY(hipMemsetAsync(k.stage_recv.p, 1, (size_t)n_local * (size_t)world, ctx().stream));
  FC_TRY(h2d(k.stage_send.p, mask_local, (size_t)n_local));
  FC_TRY(comm_allgather_dev(k.stage_send.p, k.stage_recv.p, (size_t)n_local, 0));
  FC_TRY(d2h(mask_global, k.stage_recv.p, (size_t)n_local * (size_t)world));
  return sync();
}

int fc_comm_barrier(void) {
  FC_API_LOCK;
  uint8_t mine = 1, all[64] = {0};
  return fc_allgather_mask(&mine, 1, all);
}

}  // extern "C"
