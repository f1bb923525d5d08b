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

#include <atomic>
#include <cstdint>
#include <algorithm>
#include <new>
#include <cstring>
#include <cstdlib>
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
  // bits of the timestamp of the node's NEWEST edge: a root that looks at the node from later
  // than that (the common case: queries are about "now") has its window's upper end at `size`
  // without touching the timestamps — one random access per root instead of two
  uint32_t last_ts_bits;
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

// Fences over the timestamp pool: level l (1 ..) holds every 16^l-th timestamp,
//   fence_l[g] = ts_pool[(g + 1) * 16^l - 1],
// a pure function of the POOL (global element index, not of any segment), so whoever writes
// ts_pool[d] also writes the fences d is the last element of (fence_store below).  Any fence
// whose position lies inside a node's live segment is a valid pivot of that (sorted) segment,
// and 16 consecutive pivots of a level are one 64-byte line: the window search reads ONE line
// per round instead of 16 strided 4-byte probes = 16 sectors (sampler.hip).  1/15 float per
// edge (0.27 B).  Levels sit in one buffer, level l at element offset off[l - 1].
constexpr uint32_t kFenceMaxLevels = 8;   // 16^8 = 4.3 G elements per top-level block
struct FenceView {
  float* base;
  uint64_t off[kFenceMaxLevels];
  uint32_t levels;   // levels that exist for the pool's capacity (16^levels <= capacity)
};
__host__ __device__ inline void fence_store(const FenceView& f, uint64_t d, float v) {
  uint64_t g = d;
  for (uint32_t l = 0; l < f.levels && (g & 15u) == 15u; ++l) {
    g >>= 4;
    f.base[f.off[l] + g] = v;
  }
}

// What a sampling kernel needs from the graph (all device pointers).
struct GraphView {
  const NodeEntry* table;
  uint64_t table_len;
  const float* ts_pool;
  const EdgePair* nbr_pool;
  FenceView fence;   // fence.levels == 0: no fences (GNNFLOW_SEARCH_FENCES=0)
  int nonneg_ts;     // no negative timestamp was ever ingested: a window that starts at 0 starts
                     // at the node's first edge (no search for the lower end either)
};

// Host-only restatement of one reference TemporalBlock header
// (gnnflow/csrc/common.h:35-48): sizes and time range, no storage.
struct LogicalBlock {
  uint64_t size, capacity;
  float start_ts, end_ts;
};

// The chronological logical blocks of one vertex.  The ingest only ever looks at the NEWEST
// block (fill it, or open another one), so that one sits inline in the vertex record — no heap
// allocation for the many vertices with a single block, no second cache miss per vertex on
// the planning passes; older blocks go to a vector that exists only once there are any.
class BlockList {
 public:
  BlockList() = default;
  BlockList(BlockList&& o) noexcept : tail_(o.tail_), older_(o.older_), n_(o.n_) {
    o.older_ = nullptr;
    o.n_ = 0;
  }
  BlockList& operator=(BlockList&& o) noexcept {
    if (this != &o) {
      delete older_;
      tail_ = o.tail_; older_ = o.older_; n_ = o.n_;
      o.older_ = nullptr; o.n_ = 0;
    }
    return *this;
  }
  BlockList(const BlockList& o) : tail_(o.tail_), n_(o.n_) {
    if (o.older_) older_ = new std::vector<LogicalBlock>(*o.older_);
  }
  BlockList& operator=(const BlockList& o) {
    if (this != &o) { BlockList tmp(o); *this = std::move(tmp); }
    return *this;
  }
  ~BlockList() { delete older_; }

  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  LogicalBlock& back() { return tail_; }
  const LogicalBlock& back() const { return tail_; }
  LogicalBlock& operator[](size_t i) { return i + 1 == n_ ? tail_ : (*older_)[i]; }
  const LogicalBlock& operator[](size_t i) const { return i + 1 == n_ ? tail_ : (*older_)[i]; }
  void push_back(const LogicalBlock& b) {
    if (n_ > 0) {
      if (!older_) older_ = new std::vector<LogicalBlock>();
      older_->push_back(tail_);
    }
    tail_ = b;
    ++n_;
  }
  void clear() {
    delete older_;
    older_ = nullptr;
    n_ = 0;
  }
  size_t heap_bytes() const { return older_ ? older_->capacity() * sizeof(LogicalBlock) : 0; }

 private:
  LogicalBlock tail_{};
  std::vector<LogicalBlock>* older_ = nullptr;
  uint32_t n_ = 0;
};

struct NodeState {
  uint64_t seg_start = 0, seg_cap = 0;   // physical segment (elements)
  uint64_t live_off = 0, live_size = 0;  // live range inside the segment
  uint64_t num_edges = 0;                // HostDoublyLinkedList::num_edges
  uint64_t num_insertions = 0;           // HostDoublyLinkedList::num_insertions
  float last_ts = -std::numeric_limits<float>::infinity();
  uint32_t first_block = 0;              // blocks[first_block..] are live
  uint32_t saved_blocks = 0;             // offload-to-file counter
  BlockList blocks;                      // oldest first
  size_t num_blocks() const { return blocks.size() - first_block; }
};

// Zeroed host memory for the big random-access tables of the ingest (vertex records, edge-id
// counters): 2 MB-aligned and advised MADV_HUGEPAGE — 512 times fewer first-touch faults and
// TLB entries that cover the planning passes' random accesses (THP is `madvise` on the target
// boxes, so nothing gets huge pages unasked).  Anonymous mmap: the pages are zero.
void* huge_zeroed_alloc(size_t bytes);
void huge_free(void* p, size_t bytes);

// Zero-initialised uint32 counters in chunks of 2^22 that never move: calloc hands out
// untouched zero pages, so a batch that extends the edge-id counters by 10^7 entries pays
// neither a serial 40 MB memset nor a copy of the old ones — only for the pages it actually
// increments, on the threads that do.
class ZeroedU32 {
 public:
  static constexpr size_t kShift = 22, kChunk = size_t(1) << kShift, kMask = kChunk - 1;
  ZeroedU32() = default;
  ZeroedU32(const ZeroedU32&) = delete;
  ZeroedU32& operator=(const ZeroedU32&) = delete;
  ~ZeroedU32() { for (uint32_t* c : chunks_) huge_free(c, kChunk * sizeof(uint32_t)); }
  size_t size() const { return size_; }
  uint32_t& operator[](size_t i) { return chunks_[i >> kShift][i & kMask]; }
  const uint32_t& operator[](size_t i) const { return chunks_[i >> kShift][i & kMask]; }
  void resize(size_t n) {   // grows only
    if (n <= size_) return;
    const size_t want = (n + kChunk - 1) >> kShift;
    while (chunks_.size() < want) {
      uint32_t* c = static_cast<uint32_t*>(huge_zeroed_alloc(kChunk * sizeof(uint32_t)));
      chunks_.push_back(c);
    }
    size_ = n;
  }
 private:
  std::vector<uint32_t*> chunks_;
  size_t size_ = 0;
};

// Per-vertex host records, in chunks of 2^16 vertices that never move: growing the id space
// constructs only the new chunks (in parallel — 1 GB of records for 10 M vertices is 190 ms of
// page faults on one thread), and no growth ever relocates the records of existing vertices.
class NodeTable {
 public:
  static constexpr size_t kShift = 16, kChunk = size_t(1) << kShift, kMask = kChunk - 1;
  size_t size() const { return size_; }
  NodeState& operator[](size_t v) { return chunks_[v >> kShift][v & kMask]; }
  const NodeState& operator[](size_t v) const { return chunks_[v >> kShift][v & kMask]; }
  void resize(size_t n);   // grows only (edge_store.hip)
  NodeTable() = default;
  NodeTable(const NodeTable&) = delete;
  NodeTable& operator=(const NodeTable&) = delete;
  ~NodeTable();
 private:
  std::vector<NodeState*> chunks_;   // 2 MB-aligned, MADV_HUGEPAGE (edge_store.hip)
  size_t size_ = 0;
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
  // every node id and every edge id ever inserted fits an unsigned 32-bit word (0xFFFFFFFF is
  // kept free): the partitioned sampler's shared chains may then use 12-byte reply slots
  bool ids_fit_u32() const {
    return max_node_id_ < 0xFFFFFFFFull && eid_min_ >= 0 && eid_max_ < 0xFFFFFFFFll;
  }
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
  void publish_entries(const PinnedBuffer& prepared, size_t k);

  // config
  size_t initial_pool_size_, maximum_pool_size_, minimum_block_size_;
  // every MemoryResourceType places the store in HBM; kept for introspection
  [[maybe_unused]] int mem_resource_type_;
  int insertion_policy_, device_;
  bool adaptive_;

  hipStream_t stream_ = nullptr;

  // device state
  GrowBuffer ts_pool_, nbr_pool_;   // grow in place (HIP virtual memory), never move
  DeviceBuffer fence_;              // every 16^l-th timestamp of ts_pool_ (FenceView)
  DeviceBuffer fence_prev_;         // the previous generation (samples enqueued before a growth)
  FenceView fence_view_{nullptr, {0}, 0};
  bool fences_enabled_ = true;
  std::atomic<bool> negative_ts_{false};   // an edge with a negative timestamp was ingested
  void rebuild_fences(uint64_t cap, uint64_t live);
  bool pools_ready_ = false;
  DeviceBuffer table_;
  uint64_t pool_elems_ = 0;   // capacity of the pools, in elements
  uint64_t bump_ = 0;         // high-water mark of the segment allocator
  uint64_t table_cap_ = 0;    // entries allocated in table_
  std::unique_ptr<IngestSorter> sorter_holder_;   // device-side ordering of large batches
  DeviceBuffer staging_;      // ingest staging (device)
  PinnedBuffer pinned_;       // ingest staging (host)
  PinnedBuffer order_pinned_; // device-ordered batch: group table + sorted timestamps (host)
  std::vector<uint64_t> gbase_, newcap_;   // per-group planning scratch, kept across calls
  PinnedBuffer publish_pinned_;  // node-table entries of the batch, written by the planning pass

  // host state
  NodeTable nodes_;
  std::vector<uint8_t> seen_;  // bit0: in nodes_, bit1: in src_nodes_
  size_t max_node_id_ = 0;
  bool any_node_ = false;
  size_t num_nodes_ = 0, num_src_nodes_ = 0;
  ZeroedU32 eid_dense_;
  std::unordered_map<int64_t, uint64_t> eid_sparse_;
  size_t num_live_eids_ = 0;
  int64_t eid_min_ = 0, eid_max_ = -1;   // over every edge id ever inserted
  uint64_t eids_inserted_ = 0;
  std::vector<std::vector<uint64_t>> free_lists_;  // by log2(capacity)
  size_t logical_bytes_ = 0;    // TemporalBlockAllocator::allocated_
  size_t logical_blocks_ = 0;   // h2d_mapping_.size()
};

}  // namespace gf
