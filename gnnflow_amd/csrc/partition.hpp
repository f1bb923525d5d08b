// Root bucketing of the hash-partitioned sampler (partition.hip): the entry points other
// translation units call.
#pragma once

#include <cstddef>
#include <cstdint>

#include "common.hpp"

namespace gf {

size_t partition_scratch_bytes(size_t R, int world_size);

void partition_plan_dev(const int64_t* d_nodes, const float* d_ts, const uint64_t* d_R,
                        size_t R_bound, int world_size, int rank, int64_t* d_requests,
                        uint32_t* d_pos, uint64_t* d_counts, void* d_scratch,
                        size_t scratch_bytes, int device, hipStream_t stream,
                        uint32_t* d_root_of, uint32_t stride, uint32_t* d_overflow,
                        int overflow_store);

// One job of a fused plan launch (slotted form, layers of <= 32 768 roots).  Several samples
// may share ONE request buffer — and so one exchange: job j of m writes owner q's rows into slot
// q * m + j (slot_mul = m, slot_add = j; a single sample: 1, 0), its own share from row
// `own_base` on.  To the exchange the buffer is P runs of m slots.
struct PlanJob {
  const int64_t* nodes;
  const float* ts;
  const uint64_t* d_R;      // device-resident root count, or null: R_host
  uint64_t R_host;
  int64_t* requests;        // the SHARED request buffer
  uint32_t* pos;            // [roots] row of root i in the shared buffer
  uint64_t* counts;         // [P] roots per owner
  uint32_t* d_overflow;     // the sample's overflow word
  int overflow_store;       // store the word (the sample's first plan) or only raise it
  uint32_t slot_mul, slot_add;
  uint32_t own_base;        // first row of this job's own share
  uint32_t force_overflow;  // 1: flag the sample as overflowed whatever its slots hold (the
                            // caller submitted an empty stand-in for a batch too large for
                            // this chain; every rank then redoes that sample)
  // The first *d_skip (or skip_host) roots are NOT requested: they are the previous layer's own
  // roots with the same timestamps, whose most recent neighbours the previous block already
  // holds (the merge takes them from there); their pos[] is kPosReused.
  const uint64_t* d_skip = nullptr;
  uint64_t skip_host = 0;
};
constexpr uint32_t kPosReused = 0xFFFFFFFEu;
// `n` jobs (1..4) in one launch; R_bound sizes the grid (the largest job's bound).
void partition_plan_jobs(const PlanJob* jobs, int n, size_t R_bound, int world_size, int rank,
                         uint32_t stride, int device, hipStream_t stream);
constexpr size_t kPlanJobsMaxRoots = 32768;
uint64_t part_reused_roots();

}  // namespace gf
