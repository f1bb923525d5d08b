// RCCL communicator of the partitioned path: the all-to-alls of the sampling
// exchange and of the feature pull, issued by the library itself — a collective then costs
// what a kernel launch costs on the host (torch.distributed's all_to_all_single was measured
// at ~43 us of host time per call, 4 calls per sample).  The reference's counterpart is the
// torch RPC layer between machines (gnnflow/distributed/dist_sampler.py:188-242,
// kvstore.py:285-339); within one node the GPUs talk over xGMI through RCCL.
#pragma once

#include <cstddef>
#include <cstdint>

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

 private:
  void* comm_ = nullptr;   // ncclComm_t
  int world_, rank_, device_;
  hipStream_t side_ = nullptr;
  hipEvent_t fork_ = nullptr, done_ = nullptr;
};

}  // namespace gf
