// Device-side ordering of an ingest batch (reference: the per-source
// grouping + per-group stable sort by timestamp of DynamicGraph::AddEdges,
// dynamic_graph.cu:105-128, utils.h:16-27).
//
// A batch is ordered by (source vertex, timestamp, input position): one stable LSD radix sort
// of 64-bit keys (source << 32 | order-preserving timestamp bits) carrying the input position
// — rocPRIM's device radix sort, the one library primitive on this path; the kernels around it
// are written here.  Head flags + an inclusive scan give every sorted edge its group (= source
// vertex) index, so the host only sees per-GROUP data (source, first position) and the sorted
// timestamps; it replays the reference's block policy per group, hands back one base
// destination per group, and `scatter_sorted_kernel` writes the edges of the whole batch into
// their segments from the staged input arrays.  No per-edge host work is left except the
// upload itself.
#include "ingest_sort.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <thread>
#include <vector>

namespace gf {
namespace {

__device__ inline uint32_t orderable(float f) {
  const uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float unorderable(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ void make_keys_kernel(const int64_t* __restrict__ src, const float* __restrict__ ts,
                                 uint32_t n, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ vals) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = (static_cast<uint64_t>(src[i]) << 32) | orderable(ts[i]);
    vals[i] = i;
  }
}

// sorted keys -> sorted timestamps + head flags (1 where a new source vertex starts)
__global__ void split_keys_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                  float* __restrict__ sorted_ts, uint32_t* __restrict__ flag) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t k = keys[i];
    sorted_ts[i] = unorderable(static_cast<uint32_t>(k));
    flag[i] = (i == 0 || (keys[i - 1] >> 32) != (k >> 32)) ? 1u : 0u;
  }
}

// gid_incl = inclusive scan of the head flags: group of edge i is gid_incl[i] - 1
__global__ void group_table_kernel(const uint64_t* __restrict__ keys,
                                   const uint32_t* __restrict__ flag,
                                   const uint32_t* __restrict__ gid_incl, uint32_t n,
                                   uint32_t* __restrict__ group_src,
                                   uint32_t* __restrict__ group_start) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (flag[i]) {
      const uint32_t g = gid_incl[i] - 1;
      group_src[g] = static_cast<uint32_t>(keys[i] >> 32);
      group_start[g] = i;
    }
  }
}

__global__ void scatter_sorted_kernel(const uint32_t* __restrict__ perm,
                                      const uint32_t* __restrict__ gid_incl,
                                      const uint32_t* __restrict__ group_start,
                                      const uint64_t* __restrict__ group_base,
                                      const float* __restrict__ sorted_ts,
                                      const int64_t* __restrict__ dst,
                                      const int64_t* __restrict__ eid, uint32_t n,
                                      float* __restrict__ ts_pool,
                                      EdgePair* __restrict__ nbr_pool, FenceView fence) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t g = gid_incl[i] - 1;
    const uint64_t d = group_base[g] + (i - group_start[g]);
    const uint32_t p = perm[i];
    const float t = sorted_ts[i];
    ts_pool[d] = t;
    fence_store(fence, d, t);
    EdgePair rec;
    rec.dst = dst[p];
    rec.eid = eid[p];
    rec.ts = t;
    rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
    nbr_pool[d] = rec;
  }
}

inline unsigned grid_for(size_t n) {
  return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192)));
}

}  // namespace

IngestSorter::~IngestSorter() {
  for (hipEvent_t e : stage_done_)
    if (e) (void)hipEventDestroy(e);
}

void IngestSorter::upload(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream) {
  constexpr size_t kSlot = size_t(32) << 20;   // bytes per pinned slot
  constexpr unsigned kThreads = 8;
  const char* src = static_cast<const char*>(h_src);
  char* dst = static_cast<char*>(d_dst);
  for (size_t off = 0; off < bytes; off += kSlot) {
    const size_t m = std::min(kSlot, bytes - off);
    const int s = stage_next_;
    stage_next_ ^= 1;
    stage_[s].reserve(kSlot);
    if (!stage_done_[s]) GF_HIP(hipEventCreateWithFlags(&stage_done_[s], hipEventDisableTiming));
    else GF_HIP(hipEventSynchronize(stage_done_[s]));   // the slot's previous copy has left it
    char* slot = stage_[s].as<char>();
    const size_t per = (m + kThreads - 1) / kThreads;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < kThreads; ++t) {
      const size_t a = t * per, b = std::min(m, a + per);
      if (a < b) th.emplace_back([=] { std::memcpy(slot + a, src + off + a, b - a); });
    }
    std::memcpy(slot, src + off, std::min(per, m));
    for (auto& t : th) t.join();
    GF_HIP(hipMemcpyAsync(dst + off, slot, m, hipMemcpyHostToDevice, stream));
    GF_HIP(hipEventRecord(stage_done_[s], stream));
  }
}

void IngestSorter::reserve(size_t n, hipStream_t stream) {
  if (n <= cap_) return;
  size_t cap = std::max<size_t>(cap_ ? cap_ : (1 << 16), 1);
  while (cap < n) cap *= 2;
  // raw batch | keys x2 | vals x2 | sorted ts | flags | group ids | group src / start / base
  size_t bytes = 0;
  auto take = [&](size_t b) { size_t o = bytes; bytes += align_up(b, 256); return o; };
  o_src_ = take(cap * 8); o_dst_ = take(cap * 8); o_eid_ = take(cap * 8); o_ts_ = take(cap * 4);
  o_keys0_ = take(cap * 8); o_keys1_ = take(cap * 8);
  o_vals0_ = take(cap * 4); o_vals1_ = take(cap * 4);
  o_sorted_ts_ = take(cap * 4); o_flag_ = take(cap * 4); o_gid_ = take(cap * 4);
  o_gsrc_ = take(cap * 4); o_gstart_ = take(cap * 4); o_gbase_ = take(cap * 8);
  size_t sort_tmp = 0, scan_tmp = 0;
  GF_HIP(rocprim::radix_sort_pairs(nullptr, sort_tmp, static_cast<uint64_t*>(nullptr),
                                   static_cast<uint64_t*>(nullptr),
                                   static_cast<uint32_t*>(nullptr),
                                   static_cast<uint32_t*>(nullptr), cap, 0, 64, stream));
  GF_HIP(rocprim::inclusive_scan(nullptr, scan_tmp, static_cast<uint32_t*>(nullptr),
                                 static_cast<uint32_t*>(nullptr), cap, rocprim::plus<uint32_t>(),
                                 stream));
  tmp_bytes_ = std::max(sort_tmp, scan_tmp);
  o_tmp_ = take(tmp_bytes_);
  buf_.reserve(bytes, 0, stream);
  cap_ = cap;
}

size_t IngestSorter::order(const int64_t* h_src, const int64_t* h_dst, const float* h_ts,
                           const int64_t* h_eid, size_t n, unsigned node_bits,
                           hipStream_t stream) {
  GF_REQUIRE(n > 0 && n < 0x7FFFFFFFull, "ingest sort: batch size out of range");
  reserve(n, stream);
  char* b = buf_.as<char>();
  upload(b + o_src_, h_src, n * 8, stream);
  upload(b + o_ts_, h_ts, n * 4, stream);
  upload(b + o_dst_, h_dst, n * 8, stream);
  upload(b + o_eid_, h_eid, n * 8, stream);
  const uint32_t n32 = static_cast<uint32_t>(n);
  uint64_t* keys0 = reinterpret_cast<uint64_t*>(b + o_keys0_);
  uint64_t* keys1 = reinterpret_cast<uint64_t*>(b + o_keys1_);
  uint32_t* vals0 = reinterpret_cast<uint32_t*>(b + o_vals0_);
  uint32_t* vals1 = reinterpret_cast<uint32_t*>(b + o_vals1_);
  make_keys_kernel<<<dim3(grid_for(n)), dim3(256), 0, stream>>>(
      reinterpret_cast<const int64_t*>(b + o_src_), reinterpret_cast<const float*>(b + o_ts_), n32,
      keys0, vals0);
  size_t tmp = tmp_bytes_;
  GF_HIP(rocprim::radix_sort_pairs(b + o_tmp_, tmp, keys0, keys1, vals0, vals1, n, 0,
                                   std::min(64u, 32u + node_bits), stream));
  uint32_t* flag = reinterpret_cast<uint32_t*>(b + o_flag_);
  uint32_t* gid = reinterpret_cast<uint32_t*>(b + o_gid_);
  split_keys_kernel<<<dim3(grid_for(n)), dim3(256), 0, stream>>>(
      keys1, n32, reinterpret_cast<float*>(b + o_sorted_ts_), flag);
  tmp = tmp_bytes_;
  GF_HIP(rocprim::inclusive_scan(b + o_tmp_, tmp, flag, gid, n, rocprim::plus<uint32_t>(), stream));
  group_table_kernel<<<dim3(grid_for(n)), dim3(256), 0, stream>>>(
      keys1, flag, gid, n32, reinterpret_cast<uint32_t*>(b + o_gsrc_),
      reinterpret_cast<uint32_t*>(b + o_gstart_));
  GF_HIP(hipGetLastError());
  uint32_t groups = 0;
  GF_HIP(hipMemcpyAsync(&groups, gid + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  GF_HIP(hipStreamSynchronize(stream));
  n_ = n;
  groups_ = groups;
  perm_ = vals1;
  return groups;
}

void IngestSorter::download(uint32_t* h_group_src, uint32_t* h_group_start, float* h_sorted_ts,
                            hipStream_t stream) {
  char* b = buf_.as<char>();
  GF_HIP(hipMemcpyAsync(h_group_src, b + o_gsrc_, groups_ * 4, hipMemcpyDeviceToHost, stream));
  GF_HIP(hipMemcpyAsync(h_group_start, b + o_gstart_, groups_ * 4, hipMemcpyDeviceToHost, stream));
  GF_HIP(hipMemcpyAsync(h_sorted_ts, b + o_sorted_ts_, n_ * 4, hipMemcpyDeviceToHost, stream));
  GF_HIP(hipStreamSynchronize(stream));
}

void IngestSorter::scatter(const uint64_t* h_group_base, float* ts_pool, EdgePair* nbr_pool,
                           const FenceView& fence, hipStream_t stream) {
  char* b = buf_.as<char>();
  GF_HIP(hipMemcpyAsync(b + o_gbase_, h_group_base, groups_ * 8, hipMemcpyHostToDevice, stream));
  scatter_sorted_kernel<<<dim3(grid_for(n_)), dim3(256), 0, stream>>>(
      perm_, reinterpret_cast<const uint32_t*>(b + o_gid_),
      reinterpret_cast<const uint32_t*>(b + o_gstart_),
      reinterpret_cast<const uint64_t*>(b + o_gbase_),
      reinterpret_cast<const float*>(b + o_sorted_ts_),
      reinterpret_cast<const int64_t*>(b + o_dst_), reinterpret_cast<const int64_t*>(b + o_eid_),
      static_cast<uint32_t>(n_), ts_pool, nbr_pool, fence);
  GF_HIP(hipGetLastError());
}

}  // namespace gf
