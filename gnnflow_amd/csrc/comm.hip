// RCCL communicator of the partitioned path: see comm.hpp.  RCCL is resolved at run time
// (dlopen): in a PyTorch process that is the librccl PyTorch itself has loaded (same SONAME),
// so there is ONE RCCL per process; without PyTorch it is ROCm's.  No link-time dependency:
// a single-GPU user never loads it.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
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
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(sym("ncclCommUserRank"));
    a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(sym("ncclCommCuDevice"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
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
  DeviceGuard dg(device_);
  if (side_) {
    if (comm_) (void)hipStreamSynchronize(side_);   // (aborted: its kernels may never finish)
    (void)hipStreamDestroy(side_);
  }
  if (fork_) (void)hipEventDestroy(fork_);
  if (done_) (void)hipEventDestroy(done_);
  if (comm_) (void)api().CommDestroy(static_cast<ncclComm_t>(comm_));
}

// RCCL's own view of the communicator (the driver's question after a first multi-GPU run:
// did RCCL see N ranks, on N devices?)
void RcclComm::info(int out[4]) const {
  out[0] = out[1] = out[2] = -1;
  out[3] = 0;
  if (!comm_) return;
  ncclComm_t c = static_cast<ncclComm_t>(comm_);
  GF_RCCL(api().CommCount(c, &out[0]));
  GF_RCCL(api().CommUserRank(c, &out[1]));
  GF_RCCL(api().CommCuDevice(c, &out[2]));
}

void RcclComm::abort() {
  if (!comm_) return;
  ncclComm_t c = static_cast<ncclComm_t>(comm_);
  comm_ = nullptr;            // the destructor must not wait for peers that will never come
  (void)api().CommAbort(c);
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

// ---- IpcExchange ---------------------------------------------------------------------------
struct IpcExchange::Shared {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> sense;
  uint32_t pad[14];
  // per (parity, sender, receiver): where the receiver's chunk sits in the sender's mailbox
  struct Slot { uint64_t off, bytes; } slot[2][64][64];
};

static_assert(sizeof(hipIpcMemHandle_t) == IpcExchange::kHandleBytes, "hipIpcMemHandle_t is 64 bytes");

IpcExchange::IpcExchange(int world, int rank, int device, size_t box_bytes, const char* shm_name)
    : world_(world), rank_(rank), device_(device), box_bytes_(box_bytes), shm_name_(shm_name) {
  GF_REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "ipc comm: bad rank / world");
  GF_REQUIRE(box_bytes >= 4096 && shm_name && shm_name[0] == '/', "ipc comm: bad mailbox / name");
  DeviceGuard dg(device);
  shm_bytes_ = sizeof(Shared);
  // Rank 0 creates the object afresh (never a leftover of a crashed run: O_EXCL after an
  // unlink; a fresh object is zero-filled, so the barrier counters start at 0); the other ranks
  // only ever open what rank 0 made, and wait for it to have its full size.
  int fd = -1;
  if (rank == 0) {
    shm_unlink(shm_name);
    fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    GF_REQUIRE(fd >= 0, "ipc comm: shm_open failed");
    if (ftruncate(fd, static_cast<off_t>(shm_bytes_)) != 0) {
      close(fd);
      shm_unlink(shm_name);
      throw Error(GF_ERR_IO, "ipc comm: ftruncate failed");
    }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (fd < 0) fd = shm_open(shm_name, O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && fstat(fd, &sb) == 0 && static_cast<size_t>(sb.st_size) >= shm_bytes_) break;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
        if (fd >= 0) close(fd);
        throw Error(GF_ERR_IO, "ipc comm: rank 0's shared-memory object did not appear");
      }
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, shm_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  GF_REQUIRE(p != MAP_FAILED, "ipc comm: mmap failed");
  shm_ = static_cast<Shared*>(p);   // a fresh object is zero-filled: counters start at 0
  GF_HIP(hipMalloc(reinterpret_cast<void**>(&box_), box_bytes_));
  peer_box_.assign(world, nullptr);
  peer_box_[rank] = box_;
}

IpcExchange::~IpcExchange() {
  for (int q = 0; q < world_; ++q)
    if (q != rank_ && peer_box_[q]) (void)hipIpcCloseMemHandle(peer_box_[q]);
  if (box_) (void)hipFree(box_);
  if (shm_) munmap(shm_, shm_bytes_);
  if (rank_ == 0) shm_unlink(shm_name_.c_str());
}

void IpcExchange::handle(uint8_t out[kHandleBytes]) const {
  hipIpcMemHandle_t h;
  GF_HIP(hipIpcGetMemHandle(&h, box_));
  std::memcpy(out, &h, kHandleBytes);
}

void IpcExchange::open_peers(const uint8_t* handles) {
  GF_REQUIRE(handles != nullptr, "ipc comm: null handles");
  DeviceGuard dg(device_);
  for (int q = 0; q < world_; ++q) {
    if (q == rank_) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + static_cast<size_t>(q) * kHandleBytes, kHandleBytes);
    void* p = nullptr;
    GF_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    peer_box_[q] = static_cast<char*>(p);
  }
  barrier();
}

// sense-reversing barrier over the shared counters; a peer that never arrives (it died) ends
// the wait after 30 s with an error instead of a hang
void IpcExchange::barrier() {
  if (world_ == 1) return;
  const uint32_t my = sense_ ^= 1u;
  if (shm_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<uint32_t>(world_)) {
    shm_->arrived.store(0, std::memory_order_relaxed);
    shm_->sense.store(my, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0; shm_->sense.load(std::memory_order_acquire) != my; ++spin) {
    if ((spin & 255) == 255) {
      sched_yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
        throw Error(GF_ERR_HIP, "ipc comm: a rank did not reach the barrier within 30 s");
    }
  }
}

void IpcExchange::all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                             hipStream_t stream) {
  std::vector<size_t> b(world_, bytes_per_peer), o(world_);
  for (int q = 0; q < world_; ++q) o[q] = static_cast<size_t>(q) * bytes_per_peer;
  all_to_all_v(send, b.data(), o.data(), recv, b.data(), o.data(), stream);
}

void IpcExchange::all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off,
                               void* recv, const size_t* recv_bytes, const size_t* recv_off,
                               hipStream_t stream) {
  GF_REQUIRE(send_bytes && send_off && recv_bytes && recv_off, "all_to_all_v: null split arrays");
  DeviceGuard dg(device_);
  // The mailbox has two halves used alternately, so ONE barrier per exchange is enough: a peer
  // can only overwrite the half this rank is still reading two exchanges later, i.e. after this
  // rank has reached the next exchange's barrier.  That argument needs this rank's copy-ins of
  // the PREVIOUS exchange to be complete when it reaches this barrier: they are ordered before
  // the hipStreamSynchronize below when both exchanges use one stream — a change of stream
  // synchronises the old one first.
  if (last_stream_set_ && last_stream_ != stream) GF_HIP(hipStreamSynchronize(last_stream_));
  last_stream_ = stream;
  last_stream_set_ = true;
  const uint32_t par = parity_;
  parity_ ^= 1u;
  const size_t half = box_bytes_ / 2;
  // 1. pack this rank's chunks into its mailbox and say where each peer's chunk is
  size_t at = par * half;
  for (int q = 0; q < world_; ++q) {
    GF_REQUIRE(at + send_bytes[q] <= (par + 1) * half, "ipc comm: message larger than the mailbox");
    shm_->slot[par][rank_][q].off = at;
    shm_->slot[par][rank_][q].bytes = send_bytes[q];
    if (send_bytes[q])
      GF_HIP(hipMemcpyAsync(box_ + at, static_cast<const char*>(send) + send_off[q], send_bytes[q],
                            hipMemcpyDeviceToDevice, stream));
    at += (send_bytes[q] + 255) & ~size_t{255};
  }
  GF_HIP(hipStreamSynchronize(stream));
  barrier();   // every mailbox is complete
  // 2. fetch this rank's chunk from every peer's mailbox
  for (int q = 0; q < world_; ++q) {
    const Shared::Slot s = shm_->slot[par][q][rank_];
    GF_REQUIRE(s.bytes == recv_bytes[q], "ipc comm: send / receive sizes disagree");
    if (s.bytes)
      GF_HIP(hipMemcpyAsync(static_cast<char*>(recv) + recv_off[q], peer_box_[q] + s.off, s.bytes,
                            hipMemcpyDeviceToDevice, stream));
  }
  // (the copies are ordered on `stream`; nobody touches this half again before the next barrier)
}

// ---- LoopbackExchange ----------------------------------------------------------------------
class LoopbackGroup {
 public:
  explicit LoopbackGroup(int world) : world_(world), post_(world) {}
  struct Post {
    const char* send = nullptr;
    std::vector<size_t> bytes, off;
  };
  Post& post(int rank) { return post_[rank]; }
  // a rank whose thread never arrives (it raised) ends the others' wait after 60 s with an
  // error instead of a hang
  void barrier() {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t gen = generation_;
    if (++arrived_ == world_) {
      arrived_ = 0;
      ++generation_;
      cv_.notify_all();
      return;
    }
    if (!cv_.wait_for(lk, std::chrono::seconds(60), [&] { return generation_ != gen; })) {
      --arrived_;
      throw Error(GF_ERR_HIP, "loopback comm: a rank did not reach the exchange within 60 s");
    }
  }

 private:
  const int world_;
  std::vector<Post> post_;
  std::mutex mu_;
  std::condition_variable cv_;
  int arrived_ = 0;
  uint64_t generation_ = 0;
};

std::vector<std::unique_ptr<LoopbackExchange>> LoopbackExchange::create(int world, int device) {
  GF_REQUIRE(world >= 1 && world <= 64, "loopback comm: world size must be 1..64");
  auto group = std::make_shared<LoopbackGroup>(world);
  std::vector<std::unique_ptr<LoopbackExchange>> out;
  for (int r = 0; r < world; ++r)
    out.emplace_back(new LoopbackExchange(group, world, r, device));
  return out;
}

LoopbackExchange::LoopbackExchange(std::shared_ptr<LoopbackGroup> g, int world, int rank,
                                   int device)
    : group_(std::move(g)), world_(world), rank_(rank), device_(device) {}

LoopbackExchange::~LoopbackExchange() = default;

void LoopbackExchange::all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                                  hipStream_t stream) {
  std::vector<size_t> b(world_, bytes_per_peer), o(world_);
  for (int q = 0; q < world_; ++q) o[q] = static_cast<size_t>(q) * bytes_per_peer;
  all_to_all_v(send, b.data(), o.data(), recv, b.data(), o.data(), stream);
}

void LoopbackExchange::all_to_all_v(const void* send, const size_t* send_bytes,
                                    const size_t* send_off, void* recv, const size_t* recv_bytes,
                                    const size_t* recv_off, hipStream_t stream) {
  GF_REQUIRE(send_bytes && send_off && recv_bytes && recv_off, "all_to_all_v: null split arrays");
  DeviceGuard dg(device_);
  LoopbackGroup::Post& mine = group_->post(rank_);
  mine.send = static_cast<const char*>(send);
  mine.bytes.assign(send_bytes, send_bytes + world_);
  mine.off.assign(send_off, send_off + world_);
  GF_HIP(hipStreamSynchronize(stream));   // this rank's send buffer is complete
  group_->barrier();                      // ... and so is everybody's
  for (int q = 0; q < world_; ++q) {
    const LoopbackGroup::Post& p = group_->post(q);
    GF_REQUIRE(p.bytes[rank_] == recv_bytes[q], "loopback comm: send / receive sizes disagree");
    if (recv_bytes[q])
      GF_HIP(hipMemcpyAsync(static_cast<char*>(recv) + recv_off[q], p.send + p.off[rank_],
                            recv_bytes[q], hipMemcpyDeviceToDevice, stream));
  }
  GF_HIP(hipStreamSynchronize(stream));   // this rank has read the peers' buffers
  group_->barrier();                      // ... and so has everybody: they may be reused
}

}  // namespace gf
