// EdgeStore implementation: host-side ingest planning + the HIP kernels that
// move bytes.  See edge_store.hpp for the layout; reference lines are cited on
// each function that restates reference behaviour.
#include "edge_store.hpp"
#include "ingest_sort.hpp"

#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <fstream>
#include <mutex>
#include <numeric>
#include <thread>

namespace gf {

namespace {

constexpr size_t kBlockSpace = 20;       // common.h:23-24 bytes per edge
// fn(begin, end) over [0, n) on up to 8 host threads; chunks of at least `grain` items, inline
// when one chunk covers everything.  Ingest is host-bound (gathers through a permutation), so
// the threads are what the 10^7-edge chunks of a graph build spend their time in.
template <typename F>
void parallel_for(size_t n, size_t grain, unsigned max_threads, F&& fn) {
  static const unsigned hw = std::max(1u, std::min(12u, std::thread::hardware_concurrency()));
  const size_t parts = std::min<size_t>(std::min(hw, std::max(1u, max_threads)),
                                        (n + grain - 1) / std::max<size_t>(grain, 1));
  if (parts <= 1) { fn(size_t(0), n); return; }
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex mu;
  const size_t step = (n + parts - 1) / parts;
  for (size_t p = 0; p < parts; ++p) {
    const size_t a = p * step, b = std::min(n, a + step);
    if (a >= b) break;
    th.emplace_back([&, a, b] {
      try { fn(a, b); } catch (...) { std::lock_guard<std::mutex> lk(mu); err = std::current_exception(); }
    });
  }
  for (auto& t : th) t.join();
  if (err) std::rethrow_exception(err);
}

template <typename F>
void parallel_for(size_t n, size_t grain, F&& fn) {
  parallel_for(n, grain, ~0u, std::forward<F>(fn));
}

constexpr size_t kIngestChunk = 1 << 23; // edges per staging upload

// The per-group passes of add_edges walk `nodes_` in increasing vertex order but with gaps:
// nearly every group is a cache miss on its NodeState (its newest block sits inline in it).
// Software prefetch: the NodeState of the group 16 ahead, the tail of the block list
// of the group 8 ahead (whose NodeState is in cache by then).
template <typename Groups>
inline void prefetch_groups(const NodeTable& nodes, const Groups& groups, size_t g,
                            size_t g_end) {
  if (g + 16 < g_end) {
    const size_t v = static_cast<size_t>(groups[g + 16].v);
    if (v < nodes.size()) __builtin_prefetch(&nodes[v], 1, 1);
  }
}

inline uint64_t pow2_ceil(uint64_t n) {
  uint64_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
inline int log2_exact(uint64_t p) { return 63 - __builtin_clzll(p); }

// ---- kernels -------------------------------------------------------------------
// Append a sorted batch: edge i goes to pool element dest[i].  Writes are 4 B + 16 B
// per edge; within one source node dest[] is consecutive, so stores coalesce.
__global__ void scatter_edges_kernel(const uint64_t* __restrict__ dest,
                                     const int64_t* __restrict__ dst,
                                     const int64_t* __restrict__ eid,
                                     const float* __restrict__ ts, size_t n,
                                     float* __restrict__ ts_pool,
                                     EdgePair* __restrict__ nbr_pool, FenceView fence) {
  size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    uint64_t d = dest[i];
    ts_pool[d] = ts[i];
    fence_store(fence, d, ts[i]);
    EdgePair p;
    p.dst = dst[i];
    p.eid = eid[i];
    p.ts = ts[i];
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
    nbr_pool[d] = p;
  }
}

struct MoveDesc { uint64_t src, dst, count; };

// Relocate full segments (source and destination never overlap: the destination
// is a fresh segment and frees are deferred to the end of the ingest call).
__global__ void move_segments_kernel(const MoveDesc* __restrict__ moves, size_t nmoves,
                                     float* __restrict__ ts_pool,
                                     EdgePair* __restrict__ nbr_pool, FenceView fence) {
  for (size_t m = blockIdx.x; m < nmoves; m += gridDim.x) {
    MoveDesc mv = moves[m];
    for (uint64_t i = threadIdx.x; i < mv.count; i += blockDim.x) {
      const float t = ts_pool[mv.src + i];
      ts_pool[mv.dst + i] = t;
      fence_store(fence, mv.dst + i, t);
      nbr_pool[mv.dst + i] = nbr_pool[mv.src + i];
    }
  }
}

// fence_l[g] = ts_pool[(g + 1) * 16^l - 1] for every block that ends below `live` (after the
// pools were reallocated: the fence buffer is new)
__global__ void rebuild_fences_kernel(const float* __restrict__ ts_pool, uint64_t live,
                                      FenceView fence) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const uint64_t first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (uint32_t l = 0; l < fence.levels; ++l) {
    const uint32_t shift = 4 * (l + 1);
    const uint64_t blocks = live >> shift;
    for (uint64_t g = first; g < blocks; g += stride)
      fence.base[fence.off[l] + g] = ts_pool[((g + 1) << shift) - 1];
  }
}

__global__ void update_nodes_kernel(const int64_t* __restrict__ ids,
                                    const NodeEntry* __restrict__ entries, size_t n,
                                    NodeEntry* __restrict__ table) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) table[ids[i]] = entries[i];
}

struct RangeDesc { uint64_t start, count, out; };

__global__ void gather_ranges_kernel(const RangeDesc* __restrict__ ranges, size_t nranges,
                                     const float* __restrict__ ts_pool,
                                     const EdgePair* __restrict__ nbr_pool,
                                     int64_t* __restrict__ o_dst, int64_t* __restrict__ o_eid,
                                     float* __restrict__ o_ts) {
  for (size_t r = blockIdx.x; r < nranges; r += gridDim.x) {
    RangeDesc rd = ranges[r];
    for (uint64_t i = threadIdx.x; i < rd.count; i += blockDim.x) {
      EdgePair p = nbr_pool[rd.start + i];
      o_dst[rd.out + i] = p.dst;
      o_eid[rd.out + i] = p.eid;
      o_ts[rd.out + i] = ts_pool[rd.start + i];
    }
  }
}

}  // namespace

// ---- lifetime -------------------------------------------------------------------
EdgeStore::EdgeStore(size_t initial_pool_size, size_t maximum_pool_size,
                     int mem_resource_type, size_t minimum_block_size,
                     size_t /*blocks_to_preallocate*/, int insertion_policy, int device,
                     bool adaptive_block_size)
    : initial_pool_size_(initial_pool_size),
      maximum_pool_size_(maximum_pool_size),
      minimum_block_size_(minimum_block_size),
      mem_resource_type_(mem_resource_type),
      insertion_policy_(insertion_policy),
      device_(device),
      adaptive_(adaptive_block_size),
      free_lists_(64) {
  GF_REQUIRE(mem_resource_type >= GF_MEM_CUDA && mem_resource_type <= GF_MEM_SHARED,
             "invalid memory resource type");
  GF_REQUIRE(insertion_policy == GF_INSERTION_POLICY_INSERT ||
                 insertion_policy == GF_INSERTION_POLICY_REPLACE,
             "invalid insertion policy");
  GF_REQUIRE(maximum_pool_size >= initial_pool_size,
             "maximum_pool_size must be >= initial_pool_size");
  int count = 0;
  GF_HIP(hipGetDeviceCount(&count));
  GF_REQUIRE(device >= 0 && device < count, "invalid device id");
  DeviceGuard dg(device_);
  GF_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

EdgeStore::~EdgeStore() {
  if (stream_) {
    (void)hipStreamSynchronize(stream_);
    (void)hipStreamDestroy(stream_);
  }
}

GraphView EdgeStore::view() const {
  GraphView v;
  v.table = table_.as<NodeEntry>();
  v.table_len = any_node_ ? max_node_id_ + 1 : 0;
  v.ts_pool = ts_pool_.as<float>();
  v.nbr_pool = nbr_pool_.as<EdgePair>();
  v.fence = fence_view_;
  static const bool shortcut = [] {
    const char* e = std::getenv("GNNFLOW_SEARCH_LAST_TS");   // tests / A-B runs: 0 = always search
    return !(e && std::atoi(e) == 0);
  }();
  // -1: the newest-edge shortcut is off altogether
  v.nonneg_ts = !shortcut ? -1 : (negative_ts_.load(std::memory_order_relaxed) ? 0 : 1);
  return v;
}

// (Re)allocates the fence levels for a pool of `cap` elements and fills them from the first
// `live` elements of ts_pool_.  Called when the pools grow: rare, and one strided pass.
void EdgeStore::rebuild_fences(uint64_t cap, uint64_t live) {
  static const bool enabled = [] {
    const char* v = std::getenv("GNNFLOW_SEARCH_FENCES");   // tests / A-B runs: 0 = plain search
    return !(v && std::atoi(v) == 0);
  }();
  fences_enabled_ = enabled;
  FenceView f{nullptr, {0}, 0};
  if (!enabled) { fence_view_ = f; return; }
  uint64_t total = 0;
  for (uint32_t l = 0; l < kFenceMaxLevels; ++l) {
    const uint64_t n = cap >> (4 * (l + 1));
    if (n == 0) break;
    f.off[l] = total;
    total += align_up(n, 64);   // every level starts on a 256-byte boundary
    f.levels = l + 1;
  }
  if (f.levels == 0) { fence_view_ = f; return; }
  total += 64;   // a whole aligned window of 16 fences is readable behind the last level
  // ingest kernels queued earlier hold the old view: they are on stream_, and so is this.
  // A sample() enqueued on ANOTHER stream before this add_edges may still be running with the
  // old view (callers order add_edges against sample(), SURVEY 8(b) "Threading" — the
  // reference: wait_for_all_updates_to_finish — but a sample that was merely ENQUEUED earlier
  // is legitimate): the previous generation of the fences therefore stays allocated until the
  // next growth instead of being freed here.
  DeviceBuffer fresh;
  fresh.reserve(total * sizeof(float), 0, stream_);
  GF_HIP(hipStreamSynchronize(stream_));
  std::swap(fence_, fresh);
  std::swap(fence_prev_, fresh);   // `fresh` now holds the generation before the previous one
  f.base = fence_.as<float>();
  fence_view_ = f;
  if (live >= 16) {
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(((live >> 4) + 255) / 256, 4096));
    rebuild_fences_kernel<<<dim3(std::max(grid, 1u)), dim3(256), 0, stream_>>>(
        ts_pool_.as<float>(), live, f);
    GF_HIP(hipGetLastError());
  }
}

// ---- bookkeeping ----------------------------------------------------------------
// dynamic_graph.cu:140-147 AddNodes
namespace {
constexpr size_t kHugePage = size_t(2) << 20;
inline size_t huge_rounded(size_t bytes) { return (bytes + kHugePage - 1) / kHugePage * kHugePage; }
}  // namespace

void* huge_zeroed_alloc(size_t bytes) {
  const size_t rounded = huge_rounded(bytes);
  char* raw = static_cast<char*>(mmap(nullptr, rounded + kHugePage, PROT_READ | PROT_WRITE,
                                      MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (raw == MAP_FAILED) throw std::bad_alloc();
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(raw) + kHugePage - 1) /
                                    kHugePage * kHugePage);
  if (p > raw) (void)munmap(raw, static_cast<size_t>(p - raw));
  const size_t tail = static_cast<size_t>(raw + rounded + kHugePage - (p + rounded));
  if (tail) (void)munmap(p + rounded, tail);
  (void)madvise(p, rounded, MADV_HUGEPAGE);
  return p;
}

void huge_free(void* p, size_t bytes) {
  if (p) (void)munmap(p, huge_rounded(bytes));
}

void NodeTable::resize(size_t n) {
  if (n <= size_) return;
  const size_t have = chunks_.size(), want = (n + kChunk - 1) >> kShift;
  chunks_.resize(want, nullptr);
  parallel_for(want - have, 1, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; ++c) {
      NodeState* chunk = static_cast<NodeState*>(huge_zeroed_alloc(kChunk * sizeof(NodeState)));
      for (size_t i = 0; i < kChunk; ++i) new (chunk + i) NodeState();
      chunks_[have + c] = chunk;
    }
  });
  size_ = n;
}

NodeTable::~NodeTable() {
  for (NodeState* chunk : chunks_) {
    if (!chunk) continue;
    for (size_t i = 0; i < kChunk; ++i) chunk[i].~NodeState();
    huge_free(chunk, kChunk * sizeof(NodeState));
  }
}

void EdgeStore::add_nodes(int64_t max_node) {
  size_t need = static_cast<size_t>(max_node) + 1;
  if (any_node_ && need <= nodes_.size()) return;
  nodes_.resize(need);
  if (need > seen_.capacity()) seen_.reserve(need + need / 4);
  seen_.resize(need, 0);
  if (need > table_cap_) {
    size_t cap = table_cap_ ? table_cap_ : 1024;
    while (cap < need) cap *= 2;
    table_.reserve(cap * sizeof(NodeEntry), table_cap_ * sizeof(NodeEntry), stream_,
                   /*zero_new=*/true);
    table_cap_ = table_.bytes() / sizeof(NodeEntry);
  }
  max_node_id_ = static_cast<size_t>(max_node);
  any_node_ = true;
}

// `if (--edges_[eid] == 0) edges_.erase(eid)` (dynamic_graph.cu:393-397)
void EdgeStore::drop_eid(int64_t eid) {
  if (eid >= 0 && static_cast<uint64_t>(eid) < eid_dense_.size()) {
    uint32_t& c = eid_dense_[eid];
    if (c > 0 && --c == 0) num_live_eids_--;
    return;
  }
  auto it = eid_sparse_.find(eid);
  if (it != eid_sparse_.end() && --it->second == 0) {
    eid_sparse_.erase(it);
    num_live_eids_--;
  }
}

uint64_t EdgeStore::seg_alloc(uint64_t cap) {
  int cls = log2_exact(cap);
  auto& fl = free_lists_[cls];
  if (!fl.empty()) {
    uint64_t s = fl.back();
    fl.pop_back();
    return s;
  }
  uint64_t s = bump_;
  bump_ += cap;
  return s;
}

void EdgeStore::seg_free(uint64_t start, uint64_t cap) {
  free_lists_[log2_exact(cap)].push_back(start);
}

void EdgeStore::ensure_pool(uint64_t elems) {
  if (elems <= pool_elems_) return;
  uint64_t cap = pool_elems_ ? pool_elems_ : std::max<uint64_t>(initial_pool_size_ / kBlockSpace, 1024);
  if (nbr_pool_.in_place() && ts_pool_.in_place()) {
    // growing in place costs nothing but the mapping: a quarter of headroom instead of doubling
    // (at 1.3 G edges doubling left 160 GB of HBM in use for 47 GB of edges)
    cap = std::max(cap, elems + elems / 4);
  } else {
    while (cap < elems) cap *= 2;
  }
  // the whole used prefix [0, old bump) is preserved across the reallocation
  uint64_t keep = std::min<uint64_t>(pool_elems_, bump_);
  if (!pools_ready_) {
    // address ranges for the most the pools may ever hold: power-of-two segments can take up
    // to ~4x the live edges; capped at 64 G elements
    const uint64_t max_elems = std::min<uint64_t>(
        std::max<uint64_t>(16 * (maximum_pool_size_ / kBlockSpace + 1), cap), uint64_t(1) << 36);
    ts_pool_.init(max_elems * sizeof(float), device_);
    nbr_pool_.init(max_elems * sizeof(EdgePair), device_);
    pools_ready_ = true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  ts_pool_.reserve(cap * sizeof(float), keep * sizeof(float), stream_);
  const auto t1 = std::chrono::steady_clock::now();
  nbr_pool_.reserve(cap * sizeof(EdgePair), keep * sizeof(EdgePair), stream_);
  const auto t2 = std::chrono::steady_clock::now();
  if (std::getenv("GNNFLOW_INGEST_PROFILE"))
    std::fprintf(stderr, "ensure_pool: %llu -> %llu elems (keep %llu): ts %.1f ms, nbr %.1f ms\n",
                 (unsigned long long)pool_elems_, (unsigned long long)cap, (unsigned long long)keep,
                 std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(t2 - t1).count());
  pool_elems_ = cap;
  rebuild_fences(cap, keep);
}

// temporal_block_allocator.cu:83-88,134-149 (AlignUp + AllocateInternal header init)
// (the pool limit was checked for the whole batch before anything was mutated, add_edges 2.;
// the byte / block counters are accumulated per thread in `delta` and applied by the caller)
LogicalBlock EdgeStore::new_block(size_t size, BlockDelta* delta) {
  LogicalBlock b;
  b.size = 0;
  b.capacity = size < minimum_block_size_ ? minimum_block_size_ : size;
  b.start_ts = std::numeric_limits<float>::max();
  b.end_ts = 0;
  delta->bytes_added += b.capacity * kBlockSpace;
  delta->blocks_added++;
  return b;
}

// `edges_[eid]++` for a batch (dynamic_graph.cu:93-95; num_edges() = distinct ids).  Dense
// counters for the usual non-negative, roughly contiguous eids, grown once per batch from the
// batch's largest id; ids they do not cover (negative, or far beyond what has been inserted)
// go through a hash map.
void EdgeStore::bump_eids(const int64_t* eids, size_t n) {
  int64_t mx = -1, mn = 0;
  bool consecutive = n > 0;   // eids[i] == eids[0] + i for the whole batch (a running counter)
  {
    std::mutex mu;
    parallel_for(n, 1 << 16, [&](size_t i0, size_t i1) {
      int64_t m = -1, lo = 0;
      bool run = true;
      const int64_t base = eids[0] - 0;
      for (size_t i = i0; i < i1; ++i) {
        m = std::max(m, eids[i]);
        lo = std::min(lo, eids[i]);
        run &= eids[i] == base + static_cast<int64_t>(i);
      }
      std::lock_guard<std::mutex> lk(mu);
      mx = std::max(mx, m);
      mn = std::min(mn, lo);
      consecutive = consecutive && run;
    });
  }
  eid_max_ = std::max(eid_max_, mx);
  eid_min_ = std::min(eid_min_, mn);
  const uint64_t budget = 64 + 8 * (eids_inserted_ + n);   // dense only while reasonably full
  if (mx >= 0 && static_cast<uint64_t>(mx) >= eid_dense_.size() &&
      static_cast<uint64_t>(mx) < budget) {
    const size_t nsz = static_cast<size_t>(mx) + 1;
    eid_dense_.resize(nsz);
    for (auto it = eid_sparse_.begin(); it != eid_sparse_.end();) {   // keep "eid <
      if (it->first >= 0 && static_cast<uint64_t>(it->first) < nsz) {  // dense.size() =>
        eid_dense_[it->first] += static_cast<uint32_t>(it->second);    // counted densely"
        it = eid_sparse_.erase(it);
      } else {
        ++it;
      }
    }
  }
  const uint64_t dense = eid_dense_.size();
  std::mutex mu;
  if (consecutive && eids[0] >= 0 && static_cast<uint64_t>(eids[0]) + n <= dense) {
    // the batch's ids are one run of the dense counters: every thread owns its part of the
    // run, no locked read-modify-write (10^7 of those are ~50 ms on four threads)
    const size_t first = static_cast<size_t>(eids[0]);
    parallel_for(n, 1 << 16, [&](size_t i0, size_t i1) {
      size_t fresh = 0;
      for (size_t i = i0; i < i1; ++i) fresh += eid_dense_[first + i]++ == 0 ? 1 : 0;
      std::lock_guard<std::mutex> lk(mu);
      num_live_eids_ += fresh;
    });
    eids_inserted_ += n;
    return;
  }
  parallel_for(n, 1 << 16, [&](size_t i0, size_t i1) {
    size_t fresh = 0;
    std::vector<int64_t> sparse;
    for (size_t i = i0; i < i1; ++i) {
      const int64_t e = eids[i];
      if (e >= 0 && static_cast<uint64_t>(e) < dense) {
        if (__atomic_fetch_add(&eid_dense_[e], 1u, __ATOMIC_RELAXED) == 0) fresh++;
      } else {
        sparse.push_back(e);
      }
    }
    std::lock_guard<std::mutex> lk(mu);
    num_live_eids_ += fresh;
    for (int64_t e : sparse)
      if (eid_sparse_[e]++ == 0) num_live_eids_++;
  });
  eids_inserted_ += n;
}

// dynamic_graph.cu:206-287 AddEdgesForOneNode + utils.cu:33-63 CopyEdgesToBlock,
// replayed on block headers only (the bytes live in the node's flat segment).
void EdgeStore::simulate_blocks(NodeState& st, const float* ts, size_t n, BlockDelta* delta) {
  auto copy_to = [&](LogicalBlock& b, size_t start_idx, size_t cnt) {
    b.size += cnt;
    b.start_ts = std::min(b.start_ts, ts[start_idx]);
    b.end_ts = ts[start_idx + cnt - 1];
  };
  size_t num = n, start_idx = 0;
  if (st.num_blocks() == 0) {
    st.blocks.push_back(new_block(num, delta));    // case 1: empty list
  } else {
    LogicalBlock& tail = st.blocks.back();
    if (tail.size + num > tail.capacity) {         // case 2: tail overflows
      if (insertion_policy_ == GF_INSERTION_POLICY_INSERT) {
        size_t fill = tail.capacity - tail.size;
        if (fill > 0) {
          copy_to(tail, 0, fill);
          start_idx = fill;
          num -= fill;
        }
        size_t avg = st.num_insertions == 0 ? num : st.num_edges / st.num_insertions;
        size_t new_size = adaptive_ ? pow2_ceil(std::max(num, avg)) : num;
        st.blocks.push_back(new_block(new_size, delta));
      } else {                                     // replace: Reallocate(size + n)
        delta->bytes_removed += tail.capacity * kBlockSpace;
        size_t want = tail.size + num;
        tail.capacity = want < minimum_block_size_ ? minimum_block_size_ : want;
        delta->bytes_added += tail.capacity * kBlockSpace;
      }
    }                                              // case 3: fits in the tail
  }
  copy_to(st.blocks.back(), start_idx, num);
  st.num_edges += n;
  st.num_insertions++;
}

void EdgeStore::upload_entries(const std::vector<int64_t>& ids) {
  if (ids.empty()) return;
  size_t k = ids.size();
  size_t bytes = k * (sizeof(int64_t) + sizeof(NodeEntry));
  pinned_.reserve(bytes);
  staging_.reserve(bytes, 0, stream_);
  int64_t* h_ids = pinned_.as<int64_t>();
  NodeEntry* h_ent = reinterpret_cast<NodeEntry*>(h_ids + k);
  parallel_for(k, 1 << 15, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      if (i + 16 < i1) __builtin_prefetch(&nodes_[ids[i + 16]], 0, 1);
      const NodeState& st = nodes_[ids[i]];
      h_ids[i] = ids[i];
      h_ent[i].start = st.seg_start + st.live_off;
      h_ent[i].size = static_cast<uint32_t>(st.live_size);
      std::memcpy(&h_ent[i].last_ts_bits, &st.last_ts, 4);
    }
  });
  GF_HIP(hipMemcpyAsync(staging_.data(), pinned_.data(), bytes, hipMemcpyHostToDevice, stream_));
  int64_t* d_ids = staging_.as<int64_t>();
  NodeEntry* d_ent = reinterpret_cast<NodeEntry*>(d_ids + k);
  update_nodes_kernel<<<dim3((k + 255) / 256), dim3(256), 0, stream_>>>(
      d_ids, d_ent, k, table_.as<NodeEntry>());
  GF_HIP(hipGetLastError());
  GF_HIP(hipStreamSynchronize(stream_));
}

// entries prepared by the caller in pinned memory: int64 ids[k] followed by NodeEntry[k]
void EdgeStore::publish_entries(const PinnedBuffer& prepared, size_t k) {
  if (k == 0) return;
  const size_t bytes = k * (sizeof(int64_t) + sizeof(NodeEntry));
  staging_.reserve(bytes, 0, stream_);
  GF_HIP(hipMemcpyAsync(staging_.data(), prepared.data(), bytes, hipMemcpyHostToDevice, stream_));
  int64_t* d_ids = staging_.as<int64_t>();
  NodeEntry* d_ent = reinterpret_cast<NodeEntry*>(d_ids + k);
  update_nodes_kernel<<<dim3((k + 255) / 256), dim3(256), 0, stream_>>>(
      d_ids, d_ent, k, table_.as<NodeEntry>());
  GF_HIP(hipGetLastError());
  GF_HIP(hipStreamSynchronize(stream_));
}

// ---- ingest: DynamicGraph::AddEdges, dynamic_graph.cu:77-138 -------------------
namespace {
// GNNFLOW_INGEST_PROFILE=1: per-phase host time of add_edges on stderr
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t0;
  const char* names[12];
  double us[12];
  int k = 0;
  PhaseTimer() : on(std::getenv("GNNFLOW_INGEST_PROFILE") != nullptr),
                 t0(std::chrono::steady_clock::now()) {}
  void mark(const char* name) {
    if (!on || k >= 12) return;
    auto t1 = std::chrono::steady_clock::now();
    names[k] = name;
    us[k++] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    t0 = t1;
  }
  void report(size_t n) {
    if (!on) return;
    double tot = 0;
    for (int i = 0; i < k; ++i) tot += us[i];
    std::fprintf(stderr, "add_edges n=%zu %.1f ms (%.1f M edges/s):", n, tot / 1e3, n / tot);
    for (int i = 0; i < k; ++i) std::fprintf(stderr, " %s %.1f", names[i], us[i] / 1e3);
    std::fprintf(stderr, "\n");
  }
};
}  // namespace

void EdgeStore::add_edges(const int64_t* src, const int64_t* dst, const float* ts,
                          const int64_t* eids, size_t n) {
  GF_REQUIRE(n > 0, "add_edges: empty batch (reference: CHECK_GT(src_nodes.size(), 0))");
  GF_REQUIRE(src && dst && ts && eids, "add_edges: null array");
  GF_REQUIRE(n < 0xFFFFFFFFull, "add_edges: more than 2^32-1 edges in one call");
  DeviceGuard dg(device_);
  PhaseTimer pt;

  int64_t max_node = 0;
  {
    std::mutex mu;
    bool negative = false;
    parallel_for(n, 1 << 16, [&](size_t i0, size_t i1) {
      int64_t m = 0;
      bool neg = false;
      for (size_t i = i0; i < i1; ++i) {
        neg |= (src[i] < 0) | (dst[i] < 0);
        m = std::max(m, std::max(src[i], dst[i]));
      }
      std::lock_guard<std::mutex> lk(mu);
      max_node = std::max(max_node, m);
      negative |= neg;
    });
    GF_REQUIRE(!negative, "add_edges: negative vertex id");
  }

  pt.mark("scan");
  // 1. order by (source, timestamp, input position): the reference groups by
  //    source in input order and stable-sorts each group by timestamp
  //    (dynamic_graph.cu:105-128, utils.h:16-27).  `groups` are the runs of equal source in
  //    that order and `s_ts` the timestamps in it; everything after this works per GROUP.
  //    Batches worth it are ordered on the device (ingest_sort.hip: radix sort + group table;
  //    the per-edge permutation then never exists on the host), small ones — the online
  //    600-edge ingests, where a sort launch chain would cost more than the whole call — by
  //    the host counting sort.
  struct Group { int64_t v; size_t begin, end; };
  // the runs of equal source: a host vector (host ordering), or a view of the device
  // ordering's group table in pinned memory (no 6.5 M-element vector to build per chunk)
  struct GroupList {
    std::vector<Group> host;
    const uint32_t* src = nullptr;
    const uint32_t* start = nullptr;
    size_t n = 0;
    size_t size() const { return src ? n : host.size(); }
    Group operator[](size_t g) const {
      return src ? Group{static_cast<int64_t>(src[g]), start[g], start[g + 1]} : host[g];
    }
    void push_back(const Group& g) { host.push_back(g); }
  } groups;
  std::vector<uint32_t> perm;        // host path only
  std::vector<float> s_ts_vec;
  const float* s_ts = nullptr;
  static const size_t device_sort_min = [] {
    const char* v = std::getenv("GNNFLOW_INGEST_DEVICE_SORT_MIN");   // tests: 0 = always
    return v ? static_cast<size_t>(std::atoll(v)) : (size_t(1) << 17);
  }();
  const bool on_device = n >= device_sort_min && n < 0x7FFFFFFFull &&
                         static_cast<uint64_t>(max_node) < 0xFFFFFFFFull;
  if (on_device && !sorter_holder_) sorter_holder_.reset(new IngestSorter());
  IngestSorter* const sorter = sorter_holder_.get();   // only used when on_device
  if (on_device) {
    unsigned node_bits = 1;
    while (node_bits < 32 && (static_cast<uint64_t>(max_node) >> node_bits)) ++node_bits;
    const size_t G = sorter->order(src, dst, ts, eids, n, node_bits, stream_);
    pt.mark("dsort");
    // pinned landing buffers: a pageable D2H of the 40 MB of sorted timestamps alone costs
    // more than the sort
    const size_t o_start = align_up(G * 4, 64), o_sts = o_start + align_up((G + 1) * 4, 64);
    order_pinned_.reserve(o_sts + n * 4);
    uint32_t* g_src = order_pinned_.as<uint32_t>();
    uint32_t* g_start = reinterpret_cast<uint32_t*>(order_pinned_.as<char>() + o_start);
    float* h_sts = reinterpret_cast<float*>(order_pinned_.as<char>() + o_sts);
    sorter->download(g_src, g_start, h_sts, stream_);
    g_start[G] = static_cast<uint32_t>(n);
    s_ts = h_sts;
    groups.src = g_src;
    groups.start = g_start;
    groups.n = G;
    pt.mark("groups");
  } else {
    perm.resize(n);
    s_ts_vec.resize(n);
    std::vector<float>& st = s_ts_vec;
    const size_t table_len = static_cast<size_t>(max_node) + 1;
    if (table_len <= 4 * n + (1u << 16)) {
      // counting sort by source (stable), then fix up unsorted groups by time
      std::vector<uint32_t> head(table_len + 1, 0);
      for (size_t i = 0; i < n; ++i) head[src[i] + 1]++;
      for (size_t v = 0; v < table_len; ++v) head[v + 1] += head[v];
      {
        std::vector<uint32_t> cur(head.begin(), head.end() - 1);
        for (size_t i = 0; i < n; ++i) perm[cur[src[i]]++] = static_cast<uint32_t>(i);
      }
      pt.mark("csort");
      parallel_for(table_len, 1 << 14, [&](size_t v0, size_t v1) {
        for (size_t v = v0; v < v1; ++v) {
          const size_t a = head[v], b = head[v + 1];
          bool sorted = true;
          for (size_t k = a; k < b; ++k) {
            st[k] = ts[perm[k]];
            if (k > a && st[k] < st[k - 1]) sorted = false;
          }
          if (!sorted) {
            std::stable_sort(perm.begin() + a, perm.begin() + b,
                             [&](uint32_t x, uint32_t y) { return ts[x] < ts[y]; });
            for (size_t k = a; k < b; ++k) st[k] = ts[perm[k]];
          }
        }
      });
      pt.mark("tsgather");
      for (size_t v = 0; v < table_len; ++v)
        if (head[v + 1] > head[v]) groups.push_back({static_cast<int64_t>(v), head[v], head[v + 1]});
    } else {
      std::iota(perm.begin(), perm.end(), 0u);
      std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) {
        if (src[x] != src[y]) return src[x] < src[y];
        if (ts[x] < ts[y]) return true;
        if (ts[y] < ts[x]) return false;
        return x < y;
      });
      parallel_for(n, 1 << 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) st[k] = ts[perm[k]];
      });
      for (size_t i = 0; i < n;) {
        const int64_t v = src[perm[i]];
        size_t j = i;
        while (j < n && src[perm[j]] == v) ++j;
        groups.push_back({v, i, j});
        i = j;
      }
    }
    s_ts = s_ts_vec.data();
    pt.mark("sort");
  }
  const size_t G = groups.size();

  // 2. validate before mutating anything: a group's oldest new edge must not be
  //    older than the node's newest stored edge (reference: CHECK_LE ->
  //    abort, utils.cu:42-43; the docstring promises ValueError,
  //    gnnflow/dynamic_graph.py:99-101) — and a dry run of everything that can fail later:
  //    the logical block bytes the batch adds (simulate_blocks / new_block, i.e.
  //    dynamic_graph.cu:206-287 against the rmm pool limit) and the pool elements its new
  //    segments need (seg_alloc, free lists included).  A batch that does not fit is rejected
  //    HERE — no vertex, counter, block list or allocator state has been touched yet, so the
  //    graph stays exactly as it was.
  const uint64_t min_phys = pow2_ceil(std::max<size_t>(minimum_block_size_, 1));
  {
    auto cap_of = [&](size_t x) { return x < minimum_block_size_ ? minimum_block_size_ : x; };
    std::mutex mu;
    size_t add_bytes = 0;
    std::vector<uint64_t> demand(free_lists_.size(), 0);   // new segments per size class
    std::string first_error;
    int first_code = GF_OK;
    size_t first_at = G;
    parallel_for(G, 1 << 14, [&](size_t g0, size_t g1) {
      size_t bytes = 0;
      std::vector<uint64_t> dem(free_lists_.size(), 0);
      for (size_t g = g0; g < g1; ++g) {
        prefetch_groups(nodes_, groups, g, g1);
        const Group& gr = groups[g];
        const size_t cnt = gr.end - gr.begin;
        const NodeState* st =
            static_cast<size_t>(gr.v) < nodes_.size() ? &nodes_[gr.v] : nullptr;
        if (st) {
          int code = GF_OK;
          std::string msg;
          if (s_ts[gr.begin] < st->last_ts) {
            code = GF_ERR_TIMESTAMP_ORDER;
            msg = "add_edges: vertex " + std::to_string(gr.v) + " got an edge at t=" +
                  std::to_string(s_ts[gr.begin]) + " older than its newest stored edge t=" +
                  std::to_string(st->last_ts);
          } else if (st->live_size + cnt >= 0xFFFFFFFFull) {
            code = GF_ERR_INVALID_ARGUMENT;
            msg = "add_edges: more than 2^32-1 live edges on one vertex";
          }
          if (code != GF_OK) {
            std::lock_guard<std::mutex> lk(mu);
            if (g < first_at) { first_at = g; first_code = code; first_error = msg; }
            continue;
          }
        }
        if (!st || st->num_blocks() == 0) {
          bytes += cap_of(cnt) * kBlockSpace;
        } else {
          const LogicalBlock& tail = st->blocks.back();
          if (tail.size + cnt > tail.capacity) {
            if (insertion_policy_ == GF_INSERTION_POLICY_INSERT) {
              const size_t num = cnt - (tail.capacity - tail.size);
              const size_t avg = st->num_insertions == 0 ? num : st->num_edges / st->num_insertions;
              bytes += cap_of(adaptive_ ? pow2_ceil(std::max(num, avg)) : num) * kBlockSpace;
            } else {
              bytes += (cap_of(tail.size + cnt) - tail.capacity) * kBlockSpace;
            }
          }
        }
        const uint64_t need = (st ? st->live_size : 0) + cnt;
        if (!st || st->seg_cap == 0 || st->live_off + need > st->seg_cap)
          dem[log2_exact(pow2_ceil(std::max<uint64_t>(need, min_phys)))]++;
      }
      std::lock_guard<std::mutex> lk(mu);
      add_bytes += bytes;
      for (size_t c = 0; c < dem.size(); ++c) demand[c] += dem[c];
    });
    if (first_code != GF_OK) throw Error(first_code, first_error);
    if (logical_bytes_ + add_bytes > maximum_pool_size_) {
      throw Error(GF_ERR_OUT_OF_MEMORY,
                  "maximum_pool_size exceeded: temporal blocks need " +
                      std::to_string(logical_bytes_ + add_bytes) + " bytes > " +
                      std::to_string(maximum_pool_size_) + " (batch rejected, graph unchanged)");
    }
    uint64_t extra = 0;   // what the free lists cannot serve comes off the bump pointer
    for (size_t c = 0; c < demand.size(); ++c)
      if (demand[c] > free_lists_[c].size()) extra += (demand[c] - free_lists_[c].size()) << c;
    ensure_pool(bump_ + extra);   // a failing hipMalloc also surfaces before any mutation
  }

  pt.mark("validate");
  // 3. bookkeeping sets (dynamic_graph.cu:89-103): several threads, atomic test-and-set on the
  //    per-vertex bytes / per-edge-id counters; they share nothing with the planning below
  //    (seen_ / eid counters vs nodes_ and the segment allocator), so big batches update them
  //    on helper threads meanwhile.
  add_nodes(max_node);
  auto update_sets = [&] {
    // A locked read-modify-write on a cold line costs ~10x a load, and after the first
    // batches nearly every vertex has been seen: test with a plain load, set only when a bit
    // is missing.  Sources are marked once per GROUP (distinct source), not once per edge.
    std::mutex mu;
    auto mark = [&](int64_t v, uint8_t bits, size_t* nn, size_t* ns) {
      uint8_t cur = __atomic_load_n(&seen_[v], __ATOMIC_RELAXED);
      if ((cur & bits) == bits) return;
      cur = __atomic_fetch_or(&seen_[v], bits, __ATOMIC_RELAXED);
      *nn += !(cur & 1);
      if (bits & 2) *ns += !(cur & 2);
    };
    // (the sources are marked by pass A below, which visits every group anyway; a handful
    // of threads here: the planning passes run at the same time)
    parallel_for(n, 1 << 16, 4, [&](size_t i0, size_t i1) {
      size_t nn = 0, ns = 0;
      for (size_t i = i0; i < i1; ++i) {
        // random bytes of a table that is 120 MB at MAG scale: fetch ahead
        if (i + 32 < i1) __builtin_prefetch(&seen_[dst[i + 32]], 0, 1);
        mark(dst[i], 1, &nn, &ns);
      }
      std::lock_guard<std::mutex> lk(mu);
      num_nodes_ += nn;
    });
    bump_eids(eids, n);
  };
  struct Joiner {
    std::thread t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } sets_thread;
  if (n >= (size_t(1) << 18)) sets_thread.t = std::thread(update_sets);
  else update_sets();

  pt.mark("sets");
  // 4. plan: logical blocks, physical segments, one base destination per group.
  //    Pass A (parallel over groups — every group is another vertex): replay the block policy
  //    and decide whether the vertex needs a larger segment.  Pass B (only the groups that do):
  //    the segment allocator.  Pass C (parallel): bases, live ranges.
  // per-group scratch kept across calls: a fresh 52 MB std::vector per 10^7-edge chunk is an
  // mmap + 13 k page faults + munmap every time
  if (gbase_.size() < G) { gbase_.resize(G); newcap_.resize(G); }
  uint64_t* gbase = gbase_.data();
  uint64_t* newcap = newcap_.data();   // every entry is written by pass A
  std::vector<Move> moves;
  std::vector<std::pair<uint64_t, uint64_t>> deferred_free;
  size_t planA_nodes = 0, planA_srcs = 0;   // added to the counters once the helper has joined
  {
    std::mutex mu;
    size_t d_bytes_pos = 0, d_bytes_neg = 0, d_blocks = 0, src_nodes_new = 0, src_srcs_new = 0;
    parallel_for(G, 1 << 13, [&](size_t g0, size_t g1) {
      BlockDelta delta;
      size_t new_nodes = 0, new_srcs = 0;
      for (size_t g = g0; g < g1; ++g) {
        prefetch_groups(nodes_, groups, g, g1);
        const Group& gr = groups[g];
        const size_t cnt = gr.end - gr.begin;
        NodeState& st = nodes_[gr.v];
        simulate_blocks(st, s_ts + gr.begin, cnt, &delta);
        const uint64_t need = st.live_size + cnt;
        newcap[g] = (st.seg_cap == 0 || st.live_off + need > st.seg_cap)
                        ? pow2_ceil(std::max<uint64_t>(need, min_phys)) : 0;
        // bookkeeping sets, source side (dynamic_graph.cu:89-103): once per distinct source
        uint8_t cur = __atomic_load_n(&seen_[gr.v], __ATOMIC_RELAXED);
        if ((cur & 3) != 3) {
          cur = __atomic_fetch_or(&seen_[gr.v], 3, __ATOMIC_RELAXED);
          new_nodes += !(cur & 1);
          new_srcs += !(cur & 2);
        }
      }
      std::lock_guard<std::mutex> lk(mu);
      d_bytes_pos += delta.bytes_added;
      d_bytes_neg += delta.bytes_removed;
      d_blocks += delta.blocks_added;
      src_nodes_new += new_nodes;
      src_srcs_new += new_srcs;
    });
    logical_bytes_ += d_bytes_pos;
    logical_bytes_ -= d_bytes_neg;
    logical_blocks_ += d_blocks;
    planA_nodes = src_nodes_new;
    planA_srcs = src_srcs_new;
  }
  pt.mark("planA");
  // Pass B, the segment allocator, in parallel too: the requests of one size class are ranked
  // in group order (per-chunk counts, then a prefix over the chunks); the first ones take the
  // class's free list from its back — what seg_alloc() would have handed out — and the rest
  // share one run off the bump pointer, class after class.  Where a segment lies is not
  // observable, only that every vertex gets one of the right size.
  {
    constexpr size_t kChunk = 1 << 13;
    const size_t nchunks = (G + kChunk - 1) / kChunk;
    const size_t ncls = free_lists_.size();
    std::vector<uint32_t> rank0(nchunks * ncls, 0);   // requests per (chunk, class), then ranks
    parallel_for(nchunks, 1, [&](size_t c0, size_t c1) {
      for (size_t c = c0; c < c1; ++c) {
        uint32_t* cnt = &rank0[c * ncls];
        const size_t g1 = std::min(G, (c + 1) * kChunk);
        for (size_t g = c * kChunk; g < g1; ++g)
          if (newcap[g]) cnt[log2_exact(newcap[g])]++;
      }
    });
    std::vector<uint64_t> total(ncls, 0), from_free(ncls, 0), bump_base(ncls, 0);
    for (size_t c = 0; c < nchunks; ++c)
      for (size_t k = 0; k < ncls; ++k) {
        const uint32_t here = rank0[c * ncls + k];
        rank0[c * ncls + k] = static_cast<uint32_t>(total[k]);
        total[k] += here;
      }
    uint64_t bump = bump_;
    for (size_t k = 0; k < ncls; ++k) {
      from_free[k] = std::min<uint64_t>(total[k], free_lists_[k].size());
      bump_base[k] = bump;
      bump += (total[k] - from_free[k]) << k;
    }
    std::vector<std::vector<Move>> chunk_moves(nchunks);
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> chunk_free(nchunks);
    parallel_for(nchunks, 1, [&](size_t c0, size_t c1) {
      std::vector<uint32_t> next(ncls);
      for (size_t c = c0; c < c1; ++c) {
        for (size_t k = 0; k < ncls; ++k) next[k] = rank0[c * ncls + k];
        const size_t g1 = std::min(G, (c + 1) * kChunk);
        for (size_t g = c * kChunk; g < g1; ++g) {
          if (!newcap[g]) continue;
          const int k = log2_exact(newcap[g]);
          const uint64_t j = next[k]++;
          const auto& fl = free_lists_[k];
          const uint64_t ns = j < from_free[k] ? fl[fl.size() - 1 - j]
                                               : bump_base[k] + ((j - from_free[k]) << k);
          NodeState& st = nodes_[groups[g].v];
          if (st.seg_cap) {
            if (st.live_size) chunk_moves[c].push_back({st.seg_start + st.live_off, ns, st.live_size});
            chunk_free[c].emplace_back(st.seg_start, st.seg_cap);
          }
          st.seg_start = ns;
          st.seg_cap = newcap[g];
          st.live_off = 0;
        }
      }
    });
    for (size_t k = 0; k < ncls; ++k)
      free_lists_[k].resize(free_lists_[k].size() - from_free[k]);
    bump_ = bump;
    for (size_t c = 0; c < nchunks; ++c) {
      moves.insert(moves.end(), chunk_moves[c].begin(), chunk_moves[c].end());
      deferred_free.insert(deferred_free.end(), chunk_free[c].begin(), chunk_free[c].end());
    }
  }
  pt.mark("planB");
  std::vector<uint64_t> dest;          // host path: per-edge destinations for the staging copy
  if (!on_device) dest.resize(n);
  // the node-table entries to publish are written here, while the vertex record is in cache
  // (a separate pass over the records just to read them back cost 5 ms per 10^7 edges)
  publish_pinned_.reserve(G * (sizeof(int64_t) + sizeof(NodeEntry)));
  int64_t* pub_ids = publish_pinned_.as<int64_t>();
  NodeEntry* pub_ent = reinterpret_cast<NodeEntry*>(pub_ids + G);
  parallel_for(G, 1 << 13, [&](size_t g0, size_t g1) {
    for (size_t g = g0; g < g1; ++g) {
      if (g + 16 < g1) __builtin_prefetch(&nodes_[groups[g + 16].v], 1, 1);
      const Group& gr = groups[g];
      const size_t cnt = gr.end - gr.begin;
      NodeState& st = nodes_[gr.v];
      const uint64_t base = st.seg_start + st.live_off + st.live_size;
      gbase[g] = base;
      if (!on_device)
        for (size_t k = 0; k < cnt; ++k) dest[gr.begin + k] = base + k;
      st.live_size += cnt;
      st.last_ts = s_ts[gr.end - 1];
      if (s_ts[gr.begin] < 0.0f) negative_ts_.store(true, std::memory_order_relaxed);
      pub_ids[g] = gr.v;
      pub_ent[g].start = st.seg_start + st.live_off;
      pub_ent[g].size = static_cast<uint32_t>(st.live_size);
      std::memcpy(&pub_ent[g].last_ts_bits, &st.last_ts, 4);
    }
  });
  GF_REQUIRE(bump_ <= pool_elems_, "add_edges: internal error: pool smaller than planned");
  pt.mark("planC");
  // 5. device: relocate grown segments, then scatter the batch, then publish entries
  if (!moves.empty()) {
    size_t bytes = moves.size() * sizeof(MoveDesc);
    pinned_.reserve(bytes);
    staging_.reserve(bytes, 0, stream_);
    std::memcpy(pinned_.data(), moves.data(), bytes);
    GF_HIP(hipMemcpyAsync(staging_.data(), pinned_.data(), bytes, hipMemcpyHostToDevice, stream_));
    unsigned grid = static_cast<unsigned>(std::min<size_t>(moves.size(), 4096));
    move_segments_kernel<<<dim3(grid), dim3(256), 0, stream_>>>(
        staging_.as<MoveDesc>(), moves.size(), ts_pool_.as<float>(), nbr_pool_.as<EdgePair>(),
        fence_view_);
    GF_HIP(hipGetLastError());
    GF_HIP(hipStreamSynchronize(stream_));  // staging is reused below
  }
  if (on_device) {
    sorter->scatter(gbase, ts_pool_.as<float>(), nbr_pool_.as<EdgePair>(), fence_view_, stream_);
    GF_HIP(hipStreamSynchronize(stream_));   // gbase is a host vector
  } else {
    for (size_t off = 0; off < n; off += kIngestChunk) {
      size_t m = std::min(kIngestChunk, n - off);
      size_t o_dst = m * sizeof(uint64_t), o_eid = o_dst + m * sizeof(int64_t),
             o_ts = o_eid + m * sizeof(int64_t), bytes = o_ts + m * sizeof(float);
      pinned_.reserve(bytes);
      staging_.reserve(bytes, 0, stream_);
      char* h = pinned_.as<char>();
      uint64_t* h_dest = reinterpret_cast<uint64_t*>(h);
      int64_t* h_dst = reinterpret_cast<int64_t*>(h + o_dst);
      int64_t* h_eid = reinterpret_cast<int64_t*>(h + o_eid);
      float* h_ts = reinterpret_cast<float*>(h + o_ts);
      parallel_for(m, 1 << 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
          const size_t p = perm[off + k];
          h_dest[k] = dest[off + k];
          h_dst[k] = dst[p];
          h_eid[k] = eids[p];
          h_ts[k] = s_ts[off + k];
        }
      });
      GF_HIP(hipMemcpyAsync(staging_.data(), h, bytes, hipMemcpyHostToDevice, stream_));
      char* d = staging_.as<char>();
      unsigned grid = static_cast<unsigned>(std::min<size_t>((m + 255) / 256, 8192));
      scatter_edges_kernel<<<dim3(grid), dim3(256), 0, stream_>>>(
          reinterpret_cast<uint64_t*>(d), reinterpret_cast<int64_t*>(d + o_dst),
          reinterpret_cast<int64_t*>(d + o_eid), reinterpret_cast<float*>(d + o_ts), m,
          ts_pool_.as<float>(), nbr_pool_.as<EdgePair>(), fence_view_);
      GF_HIP(hipGetLastError());
      GF_HIP(hipStreamSynchronize(stream_));  // the pinned chunk is refilled next
    }
  }
  pt.mark("device");
  for (auto& f : deferred_free) seg_free(f.first, f.second);
  publish_entries(publish_pinned_, G);  // ends with the stream sync of dynamic_graph.cu:135-137
  pt.mark("publish");
  // the bookkeeping helper shares nothing with the device phase either: joined last
  if (sets_thread.t.joinable()) sets_thread.t.join();
  num_nodes_ += planA_nodes;
  num_src_nodes_ += planA_srcs;
  pt.mark("join");
  pt.report(n);
}

// ---- DynamicGraph::OffloadOldBlocks, dynamic_graph.cu:382-411 --------------------
size_t EdgeStore::offload_old_blocks(float timestamp, bool to_file) {
  DeviceGuard dg(device_);
  struct Dropped { int64_t node; uint64_t out; std::vector<LogicalBlock> blocks; };
  std::vector<RangeDesc> ranges;
  std::vector<Dropped> dropped;
  std::vector<int64_t> touched;
  uint64_t total = 0;
  size_t num_blocks = 0;
  for (size_t v = 0; v < nodes_.size(); ++v) {
    if (!(seen_[v] & 1)) continue;
    NodeState& st = nodes_[v];
    // blocks are chronological, so the blocks with end_ts < t form a prefix
    uint64_t k = 0;
    Dropped d{static_cast<int64_t>(v), total, {}};
    while (st.num_blocks() > 0 && st.blocks[st.first_block].end_ts < timestamp) {
      const LogicalBlock& b = st.blocks[st.first_block];
      k += b.size;
      logical_bytes_ -= b.capacity * kBlockSpace;
      logical_blocks_--;
      d.blocks.push_back(b);
      st.first_block++;
      num_blocks++;
    }
    if (d.blocks.empty()) continue;
    if (k > 0) ranges.push_back({st.seg_start + st.live_off, k, total});
    total += k;
    st.live_off += k;
    st.live_size -= k;
    if (st.num_blocks() == 0) {
      st.blocks.clear();
      st.first_block = 0;
      if (st.live_size == 0 && st.seg_cap) {
        seg_free(st.seg_start, st.seg_cap);
        st.seg_start = st.seg_cap = st.live_off = 0;
      }
    }
    dropped.push_back(std::move(d));
    touched.push_back(static_cast<int64_t>(v));
  }
  if (total > 0) {
    size_t rbytes = align_up(ranges.size() * sizeof(RangeDesc), 16);
    size_t o_dst = rbytes, o_eid = o_dst + total * 8, o_ts = o_eid + total * 8,
           bytes = o_ts + total * 4;
    pinned_.reserve(bytes);
    staging_.reserve(bytes, 0, stream_);
    std::memcpy(pinned_.data(), ranges.data(), ranges.size() * sizeof(RangeDesc));
    GF_HIP(hipMemcpyAsync(staging_.data(), pinned_.data(), rbytes, hipMemcpyHostToDevice, stream_));
    char* d = staging_.as<char>();
    unsigned grid = static_cast<unsigned>(std::min<size_t>(ranges.size(), 4096));
    gather_ranges_kernel<<<dim3(grid), dim3(256), 0, stream_>>>(
        reinterpret_cast<RangeDesc*>(d), ranges.size(), ts_pool_.as<float>(),
        nbr_pool_.as<EdgePair>(), reinterpret_cast<int64_t*>(d + o_dst),
        reinterpret_cast<int64_t*>(d + o_eid), reinterpret_cast<float*>(d + o_ts));
    GF_HIP(hipGetLastError());
    GF_HIP(hipMemcpyAsync(pinned_.as<char>() + o_dst, d + o_dst, bytes - o_dst,
                          hipMemcpyDeviceToHost, stream_));
    GF_HIP(hipStreamSynchronize(stream_));
    const int64_t* h_dst = reinterpret_cast<const int64_t*>(pinned_.as<char>() + o_dst);
    const int64_t* h_eid = reinterpret_cast<const int64_t*>(pinned_.as<char>() + o_eid);
    const float* h_ts = reinterpret_cast<const float*>(pinned_.as<char>() + o_ts);
    for (uint64_t i = 0; i < total; ++i) drop_eid(h_eid[i]);
    if (to_file) {
      // temporal_block_allocator.cu:182-221 SaveToFile record layout
      for (const Dropped& dr : dropped) {
        uint64_t off = dr.out;
        NodeState& st = nodes_[dr.node];
        for (const LogicalBlock& b : dr.blocks) {
          std::string name = "temporal_block_" + std::to_string(dr.node) + "-" +
                             std::to_string(st.saved_blocks++) + ".bin";
          std::ofstream f(name, std::ios::out | std::ios::binary);
          if (!f) throw Error(GF_ERR_IO, "cannot open " + name);
          uint64_t size = b.size, cap = b.capacity, null_ptr = 0;
          f.write(reinterpret_cast<const char*>(&size), 8);
          f.write(reinterpret_cast<const char*>(&cap), 8);
          f.write(reinterpret_cast<const char*>(&b.start_ts), 4);
          f.write(reinterpret_cast<const char*>(&b.end_ts), 4);
          f.write(reinterpret_cast<const char*>(h_dst + off), 8 * b.size);
          f.write(reinterpret_cast<const char*>(h_ts + off), 4 * b.size);
          f.write(reinterpret_cast<const char*>(h_eid + off), 8 * b.size);
          f.write(reinterpret_cast<const char*>(&null_ptr), 8);  // prev
          f.write(reinterpret_cast<const char*>(&null_ptr), 8);  // next
          off += b.size;
        }
      }
    }
  }
  upload_entries(touched);
  return num_blocks;
}

// ---- accessors: dynamic_graph.cu:289-380 -----------------------------------------
void EdgeStore::out_degree(const int64_t* nodes, size_t n, size_t* out) const {
  for (size_t i = 0; i < n; ++i) {
    GF_REQUIRE(nodes[i] >= 0 && static_cast<size_t>(nodes[i]) < nodes_.size(),
               "out_degree: vertex id out of range");
    out[i] = nodes_[nodes[i]].num_edges;
  }
}

size_t EdgeStore::nodes(int64_t* out, size_t cap, bool src_only) const {
  size_t k = 0;
  const uint8_t bit = src_only ? 2 : 1;
  for (size_t v = 0; v < seen_.size(); ++v) {
    if (!(seen_[v] & bit)) continue;
    if (out && k < cap) out[k] = static_cast<int64_t>(v);
    k++;
  }
  return k;
}

size_t EdgeStore::edges(int64_t* out, size_t cap) const {
  size_t k = 0;
  for (size_t e = 0; e < eid_dense_.size(); ++e) {
    if (!eid_dense_[e]) continue;
    if (out && k < cap) out[k] = static_cast<int64_t>(e);
    k++;
  }
  for (const auto& kv : eid_sparse_) {
    if (out && k < cap) out[k] = kv.first;
    k++;
  }
  return k;
}

size_t EdgeStore::get_temporal_neighbors(int64_t node, int64_t* dst, float* ts,
                                         int64_t* eids, size_t cap) const {
  if (node < 0 || static_cast<size_t>(node) >= nodes_.size()) return 0;
  const NodeState& st = nodes_[node];
  size_t n = st.live_size;
  if (!dst || n == 0) return n;
  GF_REQUIRE(cap >= n, "get_temporal_neighbors: output arrays too small");
  DeviceGuard dg(device_);
  std::vector<float> h_ts(n);
  std::vector<EdgePair> h_nb(n);
  uint64_t s = st.seg_start + st.live_off;
  GF_HIP(hipMemcpy(h_ts.data(), ts_pool_.as<float>() + s, n * sizeof(float), hipMemcpyDeviceToHost));
  GF_HIP(hipMemcpy(h_nb.data(), nbr_pool_.as<EdgePair>() + s, n * sizeof(EdgePair), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {  // newest first
    dst[i] = h_nb[n - 1 - i].dst;
    eids[i] = h_nb[n - 1 - i].eid;
    ts[i] = h_ts[n - 1 - i];
  }
  return n;
}

float EdgeStore::avg_linked_list_length() const {
  float sum = 0;
  for (size_t v = 0; v < nodes_.size(); ++v)
    if (seen_[v] & 1) sum += static_cast<float>(nodes_[v].num_blocks());
  return sum / static_cast<float>(num_nodes_);
}

float EdgeStore::metadata_mem_usage() const {
  // sizeof(TemporalBlock) * #blocks + sizeof(DoublyLinkedList) * table size
  // (dynamic_graph.cu:370-380), with this layout's 16-byte table entries
  return static_cast<float>(64 * logical_blocks_ +
                            sizeof(NodeEntry) * (any_node_ ? max_node_id_ + 1 : 0));
}

}  // namespace gf
