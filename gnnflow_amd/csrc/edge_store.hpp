// Temporal edge store in HBM — the MI355X counterpart of the reference's
// DynamicGraph (gnnflow/csrc/dynamic_graph.{h,cu}, doubly_linked_list.{h,cu},
// temporal_block_allocator.{h,cu}).
//
// Layout (DESIGN.md "Data layout in HBM"):
//   ts_pool  : float   [pool_elems]   edge timestamps
//   nbr_pool : EdgePair[pool_elems]   {dst node, edge id, timestamp}, one 32 B sector per edge
//   table    : NodeEntry[max_node_id+1] {start element, live edge count}
// Each node owns ONE contiguous, chronologically sorted segment of the two pools
// (power-of-two capacity, moved to a twice-larger segment when it fills), so a
// sampler resolves a root's time window with a single search over ts_pool and
// reads the k most recent edges as one coalesced run.  The reference's linked
// list of blocks survives only as host-side metadata (LogicalBlock), replayed
// with the reference's exact allocation policy so that offload_old_blocks,
// avg_linked_list_length and the memory counters keep their observable
// block-granular behaviour.
#pragma once

#include <cstdint>
#include <limits>
#include <memory>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace gf {

class IngestSorter;   // ingest_sort.hpp

struct NodeEntry {   // device node table entry, 16 B
  uint64_t start;    // first live element in the pools
  uint32_t size;     // live edges (chronological, oldest first)
  uint32_t reserved;
};

// One neighbour record, 32 B and 32-byte aligned: everything the emit kernels need about a
// selected edge sits in ONE 32-byte DRAM sector.  (With the timestamp only in ts_pool a
// uniformly sampled edge cost two random sectors; on the 10 M-node / 200 M-edge graph the
// emit kernel ran at the random-access rate of HBM, not at its bandwidth.  288 GB of HBM pays
// for the 12 extra bytes per edge: 36 B/edge -> 7.8 G edges per GPU.)
struct alignas(32) EdgePair {
  int64_t dst;
  int64_t eid;
  float ts;          // copy of ts_pool[i] (ts_pool stays dense for the window search)
  uint32_t pad[3];
};

// What a sampling kernel needs from the graph (all device pointers).
struct GraphView {
  const NodeEntry* table;
  uint64_t table_len;
  const float* ts_pool;
  const EdgePair* nbr_pool;
};

// Host-only restatement of one reference TemporalBlock header
// (gnnflow/csrc/common.h:35-48): sizes and time range, no storage.
struct LogicalBlock {
  uint64_t size, capacity;
  float start_ts, end_ts;
};

struct NodeState {
  uint64_t seg_start = 0, seg_cap = 0;   // physical segment (elements)
  uint64_t live_off = 0, live_size = 0;  // live range inside the segment
  uint64_t num_edges = 0;                // HostDoublyLinkedList::num_edges
  uint64_t num_insertions = 0;           // HostDoublyLinkedList::num_insertions
  float last_ts = -std::numeric_limits<float>::infinity();
  uint32_t first_block = 0;              // blocks[first_block..] are live
  uint32_t saved_blocks = 0;             // offload-to-file counter
  std::vector<LogicalBlock> blocks;      // oldest first
  size_t num_blocks() const { return blocks.size() - first_block; }
};

class EdgeStore {
 public:
  EdgeStore(size_t initial_pool_size, size_t maximum_pool_size, int mem_resource_type,
            size_t minimum_block_size, size_t blocks_to_preallocate,
            int insertion_policy, int device, bool adaptive_block_size);
  ~EdgeStore();

  void add_edges(const int64_t* src, const int64_t* dst, const float* ts,
                 const int64_t* eids, size_t n);
  size_t offload_old_blocks(float timestamp, bool to_file);

  size_t num_nodes() const { return num_nodes_; }
  size_t num_src_nodes() const { return num_src_nodes_; }
  size_t num_edges() const { return num_live_eids_; }
  int64_t max_node_id() const { return static_cast<int64_t>(max_node_id_); }
  void out_degree(const int64_t* nodes, size_t n, size_t* out) const;
  size_t nodes(int64_t* out, size_t cap, bool src_only) const;
  size_t edges(int64_t* out, size_t cap) const;
  size_t get_temporal_neighbors(int64_t node, int64_t* dst, float* ts, int64_t* eids,
                                size_t cap) const;
  float avg_linked_list_length() const;
  float graph_mem_usage() const { return static_cast<float>(logical_bytes_); }
  float metadata_mem_usage() const;
  int device() const { return device_; }

  GraphView view() const;

 private:
  struct Move { uint64_t src, dst, count; };

  void add_nodes(int64_t max_node);
  void bump_eids(const int64_t* eids, size_t n);
  void drop_eid(int64_t eid);
  uint64_t seg_alloc(uint64_t cap);
  void seg_free(uint64_t start, uint64_t cap);
  void ensure_pool(uint64_t elems);
  struct BlockDelta { size_t bytes_added = 0, bytes_removed = 0, blocks_added = 0; };
  void simulate_blocks(NodeState& st, const float* ts, size_t n, BlockDelta* delta);
  LogicalBlock new_block(size_t size, BlockDelta* delta);
  void upload_entries(const std::vector<int64_t>& ids);

  // config
  size_t initial_pool_size_, maximum_pool_size_, minimum_block_size_;
  // every MemoryResourceType places the store in HBM; kept for introspection
  [[maybe_unused]] int mem_resource_type_;
  int insertion_policy_, device_;
  bool adaptive_;

  hipStream_t stream_ = nullptr;

  // device state
  DeviceBuffer ts_pool_, nbr_pool_, table_;
  uint64_t pool_elems_ = 0;   // capacity of the pools, in elements
  uint64_t bump_ = 0;         // high-water mark of the segment allocator
  uint64_t table_cap_ = 0;    // entries allocated in table_
  std::unique_ptr<IngestSorter> sorter_holder_;   // device-side ordering of large batches
  DeviceBuffer staging_;      // ingest staging (device)
  PinnedBuffer pinned_;       // ingest staging (host)
  PinnedBuffer order_pinned_; // device-ordered batch: group table + sorted timestamps (host)

  // host state
  std::vector<NodeState> nodes_;
  std::vector<uint8_t> seen_;  // bit0: in nodes_, bit1: in src_nodes_
  size_t max_node_id_ = 0;
  bool any_node_ = false;
  size_t num_nodes_ = 0, num_src_nodes_ = 0;
  std::vector<uint32_t> eid_dense_;
  std::unordered_map<int64_t, uint64_t> eid_sparse_;
  size_t num_live_eids_ = 0;
  uint64_t eids_inserted_ = 0;
  std::vector<std::vector<uint64_t>> free_lists_;  // by log2(capacity)
  size_t logical_bytes_ = 0;    // TemporalBlockAllocator::allocated_
  size_t logical_blocks_ = 0;   // h2d_mapping_.size()
};

}  // namespace gf
