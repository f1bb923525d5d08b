// RCCL communicator of the partitioned path: the all-to-alls of the sampling
// exchange and of the feature pull, issued by the library itself — a collective then costs
// what a kernel launch costs on the host (torch.distributed's all_to_all_single was measured
// at ~43 us of host time per call, 4 calls per sample).  The reference's counterpart is the
// torch RPC layer between machines (gnnflow/distributed/dist_sampler.py:188-242,
// kvstore.py:285-339); within one node the GPUs talk over xGMI through RCCL.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace gf {

// What the native chains need from a transport: all-to-alls ordered on a HIP stream.
class Exchange {
 public:
  virtual ~Exchange() = default;
  virtual int world() const = 0;
  virtual int rank() const = 0;
  // equal split: bytes_per_peer bytes to / from every rank (this one included), enqueued on
  // `stream`
  virtual void all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                          hipStream_t stream) = 0;
  // the same on the communicator's own stream, ordered behind what `after` holds now; the
  // caller's stream picks the result up with join()
  virtual void all_to_all_forked(const void* send, void* recv, size_t bytes_per_peer,
                                 hipStream_t after) = 0;
  virtual void join(hipStream_t stream) = 0;
  // variable split, counts / displacements in BYTES (host arrays of world() entries)
  virtual void all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off,
                            void* recv, const size_t* recv_bytes, const size_t* recv_off,
                            hipStream_t stream) = 0;
  // what the TRANSPORT itself says it is: out = {ranks, this rank, device, kind (0 RCCL, 1
  // hipIpc, 2 loopback)} — RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice
  virtual void info(int out[4]) const { out[0] = world(); out[1] = rank(); out[2] = -1; out[3] = -1; }
  // gives up the communicator without waiting for its peers (a rank whose collective never
  // completes); it must not be used afterwards
  virtual void abort() {}
};

class RcclComm : public Exchange {
 public:
  static constexpr size_t kIdBytes = 128;   // sizeof(ncclUniqueId)
  static void unique_id(uint8_t out[kIdBytes]);
  RcclComm(const uint8_t id[kIdBytes], int world, int rank, int device);
  ~RcclComm() override;
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  void all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                  hipStream_t stream) override;
  void all_to_all_forked(const void* send, void* recv, size_t bytes_per_peer,
                         hipStream_t after) override;
  void join(hipStream_t stream) override;
  void all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off, void* recv,
                    const size_t* recv_bytes, const size_t* recv_off, hipStream_t stream) override;
  void info(int out[4]) const override;
  void abort() override;

 private:
  void* comm_ = nullptr;   // ncclComm_t
  int world_, rank_, device_;
  hipStream_t side_ = nullptr;
  hipEvent_t fork_ = nullptr, done_ = nullptr;
};

// The same exchanges between processes that can map each other's device memory (hipIpc*): the
// ranks of one node — including several ranks SHARING one GPU, where RCCL refuses to run — so
// that the native multi-rank chains run for real on a one-GPU box.  Host-synchronising: every
// exchange is copy-out, process barrier (POSIX shared memory), copy-in from the peers'
// mailboxes, barrier.  A test / single-box transport, not a fast path.
class IpcExchange : public Exchange {
 public:
  static constexpr size_t kHandleBytes = 64;   // sizeof(hipIpcMemHandle_t)
  // shm_name: the name of a POSIX shared-memory object all ranks use (created by whoever comes
  // first); box_bytes: capacity of this rank's mailbox = TWICE the largest message it sends
  // (two halves, used alternately)
  IpcExchange(int world, int rank, int device, size_t box_bytes, const char* shm_name);
  ~IpcExchange() override;
  void handle(uint8_t out[kHandleBytes]) const;          // this rank's mailbox, to publish
  void open_peers(const uint8_t* handles);               // [world][kHandleBytes], then a barrier
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  void all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                  hipStream_t stream) override;
  void all_to_all_forked(const void* send, void* recv, size_t bytes_per_peer,
                         hipStream_t after) override { all_to_all(send, recv, bytes_per_peer, after); }
  void join(hipStream_t) override {}
  void all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off, void* recv,
                    const size_t* recv_bytes, const size_t* recv_off, hipStream_t stream) override;

 private:
  struct Shared;
  void barrier();
  int world_, rank_, device_;
  size_t box_bytes_;
  char* box_ = nullptr;               // this rank's mailbox (device)
  std::vector<char*> peer_box_;       // the peers' mailboxes as mapped here
  Shared* shm_ = nullptr;
  size_t shm_bytes_ = 0;
  std::string shm_name_;
  uint32_t sense_ = 0, parity_ = 0;
  hipStream_t last_stream_ = nullptr;   // the stream of the previous exchange
  bool last_stream_set_ = false;
};

// The same exchanges between "ranks" that are objects of ONE process: every virtual rank has its
// own sampler / cache / pull session and its own host thread, and they meet in a LoopbackGroup.
// An exchange = publish the send buffer, synchronise the own stream, thread barrier, copy this
// rank's chunks straight out of the peers' send buffers on the own stream, synchronise, barrier
// (after which the peers may overwrite their send buffers).  Host-synchronising test transport:
// it runs the native multi-rank chains (gf_sampler_sample_partitioned_comm, gf_pull_round) at
// world sizes the one-GPU test box cannot give as processes (8 ranks = 8 GPU processes > the
// box's limit).  Each rank's calls must come from its own thread; the *_async entry points (one
// enqueue thread for all ranks) would deadlock and are refused.
class LoopbackGroup;
class LoopbackExchange : public Exchange {
 public:
  static std::vector<std::unique_ptr<LoopbackExchange>> create(int world, int device);
  ~LoopbackExchange() override;
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  void all_to_all(const void* send, void* recv, size_t bytes_per_peer,
                  hipStream_t stream) override;
  void all_to_all_forked(const void* send, void* recv, size_t bytes_per_peer,
                         hipStream_t after) override { all_to_all(send, recv, bytes_per_peer, after); }
  void join(hipStream_t) override {}
  void all_to_all_v(const void* send, const size_t* send_bytes, const size_t* send_off, void* recv,
                    const size_t* recv_bytes, const size_t* recv_off, hipStream_t stream) override;

 private:
  LoopbackExchange(std::shared_ptr<LoopbackGroup> g, int world, int rank, int device);
  std::shared_ptr<LoopbackGroup> group_;
  int world_, rank_, device_;
};

}  // namespace gf
