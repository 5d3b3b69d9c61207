// tests/stubs/rccl_stub.cpp -- a STAND-IN for librccl, test infrastructure only (FC_RCCL_LIB points libfc_hip.so's
// run-time loader at it; tests/test_gpu_comm_stub.py builds it).  RCCL refuses a communicator whose ranks share a device,
// and the pool gives a test ONE GPU: this library accepts the duplicate device and moves the all-gather's bytes between
// the rank PROCESSES through a shared-memory file, so that fc_comm_init -> fc_prune_rmsd_sharded (both lanes, the s_comm
// event ordering, the staging buffers, the per-level fallback) run end to end with world > 1.  It is not a collective
// library and measures nothing: every call blocks the host until all ranks have contributed.
//
// Exports exactly what fc_comm.cpp resolves: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather,
// ncclGetErrorString (types from <rccl/rccl.h>).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

namespace {
constexpr size_t kHalf = (size_t)96 << 20;          // bytes per buffer half (all ranks' pieces of one call)
constexpr size_t kHeader = 4096;
struct Header {
  std::atomic<uint64_t> arrived[64];                 // calls completed by each rank (its piece of call k is written)
  std::atomic<uint64_t> joined;                      // ranks that have mapped the file (ncclCommInitRank is collective)
};
struct StubComm {
  int rank = 0, world = 1;
  char *base = nullptr;
  uint64_t seq = 0;
  char name[80] = {0};
};
size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  std::random_device rd;
  std::snprintf(id->internal, sizeof id->internal, "/fc_rccl_stub_%08x%08x", (unsigned)rd(), (unsigned)rd());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  auto *c = new StubComm;
  c->rank = rank;
  c->world = nranks;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  const size_t bytes = kHeader + 2 * kHalf;
  if (ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;  // (fresh pages are zero: every counter starts at 0)
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->base = static_cast<char *>(p);
  // collective, like the real one: return once every rank has joined.  (The callers rely on that -- rank 0 of
  // firecode_amd.dist.comm_init_from_env removes the id file behind fc_comm_init; a stand-in that returned at once let
  // rank 0 remove it before a slow rank had read it: one time-out in ~20 runs of tests/test_gpu_comm_stub.py.)
  auto *h = reinterpret_cast<Header *>(c->base);
  h->joined.fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  while (h->joined.load(std::memory_order_acquire) < (uint64_t)nranks) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  auto *c = reinterpret_cast<StubComm *>(comm);
  if (!c) return ncclSuccess;
  if (c->base) munmap(c->base, kHeader + 2 * kHalf);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}

// blocking: waits for everything `stream` holds, exchanges through the file, returns with recv filled
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm,
                           hipStream_t stream) {
  auto *c = reinterpret_cast<StubComm *>(comm);
  const size_t bytes = count * type_size(type);
  if (bytes * (size_t)c->world > kHalf) return ncclInvalidArgument;
  auto *h = reinterpret_cast<Header *>(c->base);
  char *half = c->base + kHeader + (c->seq & 1) * kHalf;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(half + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  h->arrived[c->rank].store(c->seq + 1, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < c->world; ++r)
    while (h->arrived[r].load(std::memory_order_acquire) < c->seq + 1) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;  // a rank died
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  if (hipMemcpy(recv, half, bytes * (size_t)c->world, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  ++c->seq;  // (this half is written again two calls from now: by then every rank has read it -- it has passed call seq + 1)
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "invalid argument (stub)";
    case ncclSystemError: return "system error (stub: shared memory, or a rank did not arrive)";
    default: return "error (stub)";
  }
}

}  // extern "C"
