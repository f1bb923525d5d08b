// RCCL communicator of the partitioned path: see comm.hpp.  RCCL is resolved at run time
// (dlopen): in a PyTorch process that is the librccl PyTorch itself has loaded (same SONAME),
// so there is ONE RCCL per process; without PyTorch it is ROCm's.  No link-time dependency:
// a single-GPU user never loads it.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace gf {
namespace {

struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                           hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
};

const Api& api() {
  static Api a;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      err = std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(h, n);
      if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.AllToAll = reinterpret_cast<decltype(a.AllToAll)>(sym("ncclAllToAll"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
  });
  if (!err.empty()) throw Error(GF_ERR_INVALID_ARGUMENT, err);
  return a;
}

#define GF_RCCL(expr)                                                                  \
  do {                                                                                 \
    ncclResult_t _r = (expr);                                                          \
    if (_r != ncclSuccess)                                                             \
      throw ::gf::Error(GF_ERR_HIP, std::string(#expr) + ": " + api().GetErrorString(_r)); \
  } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == RcclComm::kIdBytes, "ncclUniqueId is 128 bytes");

void RcclComm::unique_id(uint8_t out[kIdBytes]) {
  ncclUniqueId id;
  GF_RCCL(api().GetUniqueId(&id));
  std::memcpy(out, &id, kIdBytes);
}

RcclComm::RcclComm(const uint8_t idb[kIdBytes], int world, int rank, int device)
    : world_(world), rank_(rank), device_(device) {
  GF_REQUIRE(idb != nullptr && world >= 1 && rank >= 0 && rank < world, "comm: bad rank / world");
  DeviceGuard dg(device);
  ncclUniqueId id;
  std::memcpy(&id, idb, kIdBytes);
  ncclComm_t c = nullptr;
  GF_RCCL(api().CommInitRank(&c, world, id, rank));   // collective: every rank calls it
  comm_ = c;
  GF_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
  GF_HIP(hipEventCreateWithFlags(&fork_, hipEventDisableTiming));
  GF_HIP(hipEventCreateWithFlags(&done_, hipEventDisableTiming));
}

RcclComm::~RcclComm() {
  if (side_) {
    (void)hipStreamSynchronize(side_);
    (void)hipStreamDestroy(side_);
  }
  if (fork_) (void)hipEventDestroy(fork_);
  if (done_) (void)hipEventDestroy(done_);
  if (comm_) (void)api().CommDestroy(static_cast<ncclComm_t>(comm_));
}

void RcclComm::all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                          hipStream_t stream) {
  GF_REQUIRE(send && recv, "all_to_all: null buffer");
  if (bytes_per_peer == 0) return;
  DeviceGuard dg(device_);
  // 8-byte words when the slots allow it (they do: requests are 16 B rows, replies 24 B words)
  if (bytes_per_peer % 8 == 0)
    GF_RCCL(api().AllToAll(send, recv, bytes_per_peer / 8, ncclInt64,
                           static_cast<ncclComm_t>(comm_), stream));
  else
    GF_RCCL(api().AllToAll(send, recv, bytes_per_peer, ncclInt8, static_cast<ncclComm_t>(comm_),
                           stream));
}

void RcclComm::all_to_all_forked(const void* send, void* recv, size_t bytes_per_peer,
                                 hipStream_t after) {
  DeviceGuard dg(device_);
  GF_HIP(hipEventRecord(fork_, after));
  GF_HIP(hipStreamWaitEvent(side_, fork_, 0));
  all_to_all(send, recv, bytes_per_peer, side_);
  GF_HIP(hipEventRecord(done_, side_));
}

void RcclComm::join(hipStream_t stream) {
  DeviceGuard dg(device_);
  GF_HIP(hipStreamWaitEvent(stream, done_, 0));
}

void RcclComm::all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off,
                            void* recv, const size_t* recv_bytes, const size_t* recv_off,
                            hipStream_t stream) {
  GF_REQUIRE(send_bytes && send_off && recv_bytes && recv_off, "all_to_all_v: null split arrays");
  DeviceGuard dg(device_);
  const Api& a = api();
  ncclComm_t c = static_cast<ncclComm_t>(comm_);
  GF_RCCL(a.GroupStart());
  try {
    for (int p = 0; p < world_; ++p) {
      if (send_bytes[p])
        GF_RCCL(a.Send(static_cast<const char*>(send) + send_off[p], send_bytes[p], ncclInt8, p, c,
                       stream));
      if (recv_bytes[p])
        GF_RCCL(a.Recv(static_cast<char*>(recv) + recv_off[p], recv_bytes[p], ncclInt8, p, c,
                       stream));
    }
  } catch (...) {
    (void)a.GroupEnd();
    throw;
  }
  GF_RCCL(a.GroupEnd());
}

}  // namespace gf
