// Device-side ordering of an ingest batch by (source, timestamp, input position); see
// ingest_sort.hip.  Used by EdgeStore::add_edges for batches worth a device sort.
#pragma once

#include <cstdint>

#include "common.hpp"
#include "edge_store.hpp"

namespace gf {

class IngestSorter {
 public:
  ~IngestSorter();
  // Uploads the batch, orders it and builds the group table on the device; returns the number
  // of groups (distinct sources).  node_bits = bits needed for the largest source id (<= 32).
  size_t order(const int64_t* h_src, const int64_t* h_dst, const float* h_ts,
               const int64_t* h_eid, size_t n, unsigned node_bits, hipStream_t stream);
  // group g = source h_group_src[g], sorted positions [h_group_start[g], h_group_start[g+1]);
  // h_sorted_ts[n] = the batch's timestamps in sorted order
  void download(uint32_t* h_group_src, uint32_t* h_group_start, float* h_sorted_ts,
                hipStream_t stream);
  // writes sorted edge i of group g to pool element h_group_base[g] + (i - start[g])
  void scatter(const uint64_t* h_group_base, float* ts_pool, EdgePair* nbr_pool,
               const FenceView& fence, hipStream_t stream);

 private:
  void reserve(size_t n, hipStream_t stream);
  // host -> device through two pinned slots filled by several threads: the runtime's own
  // staging of a pageable source ran anywhere between 11 and 47 GB/s on the same box
  void upload(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream);
  PinnedBuffer stage_[2];
  hipEvent_t stage_done_[2] = {nullptr, nullptr};
  int stage_next_ = 0;
  DeviceBuffer buf_;
  size_t cap_ = 0, n_ = 0, groups_ = 0, tmp_bytes_ = 0;
  size_t o_src_ = 0, o_dst_ = 0, o_eid_ = 0, o_ts_ = 0, o_keys0_ = 0, o_keys1_ = 0, o_vals0_ = 0,
         o_vals1_ = 0, o_sorted_ts_ = 0, o_flag_ = 0, o_gid_ = 0, o_gsrc_ = 0, o_gstart_ = 0,
         o_gbase_ = 0, o_tmp_ = 0;
  uint32_t* perm_ = nullptr;
};

}  // namespace gf
