// Root bucketing for hash-partitioned sampling (the reference buckets by
// partition table on the host, gnnflow/distributed/dist_sampler.py:174-186).
//
// owner(v) = splitmix64(v) mod P.  One call turns a layer's roots into
//   * requests[R][2] = (root id, root-ts bits) ordered  [owners other than this rank, in
//     ascending order | this rank's own roots], each owner's roots in their original order —
//     so the rows that travel are one contiguous prefix (the all-to-all-v send buffer) and the
//     rank's own share is the contiguous suffix it samples locally while the exchange runs;
//   * pos[i] = row of root i in that order (the merge kernel reads replies through it);
//   * counts[P] = roots per owner.
// Three short launches: per-tile owner histograms (wave ballots), one-workgroup scan of the
// tile table, stable scatter.  Integer / byte work, HBM-bound, no atomics.
#include "common.hpp"

#include <cstdint>
#include <cstdlib>

namespace gf {
namespace {

constexpr int kTileThreads = 256;      // one root per thread per tile
constexpr int kMaxParts = 64;

__device__ inline uint32_t owner_of(int64_t v, uint32_t P) {
  uint64_t z = static_cast<uint64_t>(v) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return static_cast<uint32_t>(z % P);
}

// tile_counts[tile][o] = roots of owner o in the tile
__global__ __launch_bounds__(kTileThreads) void partition_count_kernel(
    const int64_t* __restrict__ nodes, const uint64_t* __restrict__ d_R, uint64_t R_host,
    uint32_t P, uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_cnt[kTileThreads / 64][kMaxParts];
  const uint64_t R = d_R ? *d_R : R_host;   // device-resident count: a chained layer
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kTileThreads + threadIdx.x;
  const uint32_t o = i < R ? owner_of(nodes[i], P) : P;   // P = "no root"
  for (uint32_t q = 0; q < P; ++q) {
    const unsigned long long m = __ballot(o == q);
    if (lane == 0) wave_cnt[wave][q] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < P) {
    uint32_t c = 0;
    for (int w = 0; w < kTileThreads / 64; ++w) c += wave_cnt[w][threadIdx.x];
    tile_counts[static_cast<uint64_t>(blockIdx.x) * P + threadIdx.x] = c;
  }
}

// One workgroup: totals per owner, the start of every owner's run in the output order
// (other owners ascending, then `rank`), and tile_base[tile][o] = first output row of the
// tile's roots of owner o.
__global__ __launch_bounds__(1024) void partition_scan_kernel(
    const uint32_t* __restrict__ tile_counts, uint64_t tiles, uint32_t P, uint32_t rank,
    uint32_t* __restrict__ tile_base, uint64_t* __restrict__ counts) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t total_s[kMaxParts];
  __shared__ uint32_t start_s[kMaxParts];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pass 1: per-owner exclusive scan over the tiles (relative to the owner's run)
  for (uint32_t o = 0; o < P; ++o) {
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t t0 = 0; t0 < tiles; t0 += 1024) {
      const uint64_t t = t0 + tid;
      const uint32_t v = t < tiles ? tile_counts[t * P + o] : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      uint32_t wbase = 0;
      for (int w = 0; w < wave; ++w) wbase += wsum[w];
      if (t < tiles) tile_base[t * P + o] = carry_s + wbase + incl - v;
      __syncthreads();
      if (tid == 1023) carry_s += wbase + incl;
      __syncthreads();
    }
    if (tid == 0) total_s[o] = carry_s;
    __syncthreads();
  }
  // pass 2: where each owner's run starts
  if (tid == 0) {
    uint32_t at = 0;
    for (uint32_t o = 0; o < P; ++o)
      if (o != rank) { start_s[o] = at; at += total_s[o]; }
    start_s[rank] = at;
    for (uint32_t o = 0; o < P; ++o) counts[o] = total_s[o];
  }
  __syncthreads();
  for (uint64_t k = tid; k < tiles * P; k += 1024) tile_base[k] += start_s[k % P];
}

__global__ __launch_bounds__(kTileThreads) void partition_scatter_kernel(
    const int64_t* __restrict__ nodes, const float* __restrict__ ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t P,
    const uint32_t* __restrict__ tile_base, int64_t* __restrict__ requests,
    uint32_t* __restrict__ pos) {
  __shared__ uint32_t wave_cnt[kTileThreads / 64][kMaxParts];
  const uint64_t R = d_R ? *d_R : R_host;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kTileThreads + threadIdx.x;
  const int64_t v = i < R ? nodes[i] : 0;
  const uint32_t o = i < R ? owner_of(v, P) : P;
  uint32_t before_in_wave = 0;
  for (uint32_t q = 0; q < P; ++q) {
    const unsigned long long m = __ballot(o == q);
    if (lane == 0) wave_cnt[wave][q] = __popcll(m);
    if (o == q) before_in_wave = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (i >= R) return;
  uint32_t p = tile_base[static_cast<uint64_t>(blockIdx.x) * P + o] + before_in_wave;
  for (int w = 0; w < wave; ++w) p += wave_cnt[w][o];
  requests[2 * static_cast<uint64_t>(p)] = v;
  requests[2 * static_cast<uint64_t>(p) + 1] = static_cast<int64_t>(__float_as_uint(ts[i]));
  pos[i] = p;
}

// Small layers (<= 30 720 roots, <= 16 owners): the whole plan in ONE launch of one
// workgroup — three launches and two kernel boundaries less than the tiled form, which at
// batch-600 sizes is most of its time.  One CU has to touch every root, so every global access
// is coalesced and everything per-root lives in LDS (one uint16 per root):
//   1. thread t hashes roots t, t + 1024, ... (coalesced loads, all issued up front) and
//      stores the owners in LDS;
//   2. thread t counts the owners of the CONTIGUOUS roots [t*c, (t+1)*c) — contiguous chunks
//      keep every owner's roots in their original order — in 16-bit fields packed four to a
//      64-bit word; one workgroup scan per word gives its base inside every owner's run;
//   3. it walks its chunk again and replaces each owner in LDS by the root's output row;
//   4. coalesced again: root i's id / timestamp go to requests[row_i], pos[i] = row_i.
constexpr int kSmallPlanThreads = 1024;
constexpr uint32_t kSmallPlanRoots = 30 * 1024;
constexpr uint32_t kSmallPlanParts = 16;
constexpr uint32_t kSmallPlanLaunchRoots = 4096;   // largest layer that takes this path

__global__ __launch_bounds__(kSmallPlanThreads) void partition_plan_small_kernel(
    const int64_t* __restrict__ nodes, const float* __restrict__ ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t P, uint32_t rank,
    int64_t* __restrict__ requests, uint32_t* __restrict__ pos, uint64_t* __restrict__ counts) {
  __shared__ uint16_t per_root[kSmallPlanRoots];   // owner, then output row
  __shared__ unsigned long long wave_tot[kSmallPlanParts / 4][kSmallPlanThreads / 64];
  __shared__ uint32_t start_s[kSmallPlanParts];
  const uint32_t R = static_cast<uint32_t>(d_R ? *d_R : R_host);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr uint32_t kRounds = kSmallPlanRoots / kSmallPlanThreads;
  // 1. owners, coalesced
  {
    int64_t v[kRounds];
#pragma unroll
    for (uint32_t k = 0; k < kRounds; ++k) {
      const uint32_t i = k * kSmallPlanThreads + tid;
      v[k] = i < R ? nodes[i] : 0;
    }
#pragma unroll
    for (uint32_t k = 0; k < kRounds; ++k) {
      const uint32_t i = k * kSmallPlanThreads + tid;
      if (i < R) per_root[i] = static_cast<uint16_t>(owner_of(v[k], P));
    }
  }
  __syncthreads();
  // 2. per-thread counts over its contiguous chunk, packed; workgroup scan
  const uint32_t c = (R + kSmallPlanThreads - 1) / kSmallPlanThreads;
  const uint32_t i0 = min(R, tid * c), i1 = min(R, i0 + c);
  const uint32_t words = (P + 3) / 4;
  unsigned long long w[kSmallPlanParts / 4] = {0, 0, 0, 0};
  for (uint32_t i = i0; i < i1; ++i) {
    const uint32_t o = per_root[i];
    const unsigned long long inc = 1ull << (16 * (o & 3));
#pragma unroll
    for (uint32_t j = 0; j < kSmallPlanParts / 4; ++j) w[j] += (o >> 2) == j ? inc : 0ull;
  }
  unsigned long long excl[kSmallPlanParts / 4], total[kSmallPlanParts / 4];
  for (uint32_t j = 0; j < words; ++j) {
    unsigned long long incl = w[j];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_tot[j][wave] = incl;
    excl[j] = incl - w[j];
  }
  __syncthreads();
  for (uint32_t j = 0; j < words; ++j) {
    unsigned long long base = 0, tot = 0;
    for (int x = 0; x < kSmallPlanThreads / 64; ++x) {
      const unsigned long long v = wave_tot[j][x];
      if (x < wave) base += v;
      tot += v;
    }
    excl[j] += base;
    total[j] = tot;
  }
  if (tid == 0) {
    uint32_t at = 0;
    for (uint32_t o = 0; o < P; ++o) {
      const uint32_t n = static_cast<uint32_t>((total[o >> 2] >> (16 * (o & 3))) & 0xFFFFu);
      counts[o] = n;
      if (o != rank) { start_s[o] = at; at += n; }
    }
    start_s[rank] = at;
  }
  __syncthreads();
  // 3. output row of every root of the chunk (in place of its owner)
  unsigned long long seen[kSmallPlanParts / 4] = {0, 0, 0, 0};
  for (uint32_t i = i0; i < i1; ++i) {
    const uint32_t o = per_root[i];
    const uint32_t sh = 16 * (o & 3);
    unsigned long long e = 0, sn = 0;    // selects, not dynamic indexing: registers, no scratch
#pragma unroll
    for (uint32_t j = 0; j < kSmallPlanParts / 4; ++j) {
      const bool mine = (o >> 2) == j;
      e = mine ? excl[j] : e;
      sn = mine ? seen[j] : sn;
      seen[j] += mine ? (1ull << sh) : 0ull;
    }
    per_root[i] = static_cast<uint16_t>(start_s[o] + static_cast<uint32_t>((e >> sh) & 0xFFFFu) +
                                        static_cast<uint32_t>((sn >> sh) & 0xFFFFu));
  }
  __syncthreads();
  // 4. write-out, coalesced reads
#pragma unroll
  for (uint32_t k = 0; k < kRounds; ++k) {
    const uint32_t i = k * kSmallPlanThreads + tid;
    if (i < R) {
      const uint32_t p = per_root[i];
      requests[2 * static_cast<uint64_t>(p)] = nodes[i];
      requests[2 * static_cast<uint64_t>(p) + 1] = static_cast<int64_t>(__float_as_uint(ts[i]));
      pos[i] = p;
    }
  }
}

}  // namespace

size_t partition_scratch_bytes(size_t R, int world_size) {
  const size_t tiles = (R + kTileThreads - 1) / kTileThreads;
  return 2 * align_up(std::max<size_t>(tiles, 1) * world_size * sizeof(uint32_t), 16);
}

// R_bound sizes the grids and the scratch; the kernels take the real count from *d_R when it
// is given (a chained layer: the previous layer's R + S never left the device).
void partition_plan_dev(const int64_t* d_nodes, const float* d_ts, const uint64_t* d_R,
                        size_t R_bound, int world_size, int rank, int64_t* d_requests,
                        uint32_t* d_pos, uint64_t* d_counts, void* d_scratch,
                        size_t scratch_bytes, int device, hipStream_t stream) {
  GF_REQUIRE(world_size >= 1 && world_size <= kMaxParts, "partition: world size must be 1..64");
  GF_REQUIRE(rank >= 0 && rank < world_size, "partition: rank out of range");
  GF_REQUIRE(d_counts != nullptr, "partition: null counts");
  GF_REQUIRE(R_bound < 0xFFFFFFFFull, "partition: more than 2^32-1 roots");
  DeviceGuard dg(device);
  if (R_bound == 0) {
    GF_HIP(hipMemsetAsync(d_counts, 0, world_size * sizeof(uint64_t), stream));
    return;
  }
  GF_REQUIRE(d_nodes && d_ts && d_requests && d_pos && d_scratch, "partition: null pointer");
  GF_REQUIRE(scratch_bytes >= partition_scratch_bytes(R_bound, world_size),
             "partition: scratch buffer too small");
  const uint32_t P = static_cast<uint32_t>(world_size);
  static const bool small_plan = [] {
    const char* v = std::getenv("GNNFLOW_PARTITION_SMALL_PLAN");   // tests: 0 = tiled form only
    return !(v && std::atoi(v) == 0);
  }();
  // one CU has to touch every root: measured 6.5 us at 1 800 roots (the tiled form: ~12) but
  // 20-40 us at 19 800, so only the first layers of a sample take this path
  if (small_plan && R_bound <= kSmallPlanLaunchRoots && P <= kSmallPlanParts) {
    partition_plan_small_kernel<<<dim3(1), dim3(kSmallPlanThreads), 0, stream>>>(
        d_nodes, d_ts, d_R, R_bound, P, static_cast<uint32_t>(rank), d_requests, d_pos, d_counts);
    GF_HIP(hipGetLastError());
    return;
  }
  const size_t tiles = (R_bound + kTileThreads - 1) / kTileThreads;
  uint32_t* tile_counts = static_cast<uint32_t*>(d_scratch);
  uint32_t* tile_base = reinterpret_cast<uint32_t*>(
      static_cast<char*>(d_scratch) + align_up(tiles * world_size * sizeof(uint32_t), 16));
  partition_count_kernel<<<dim3(static_cast<unsigned>(tiles)), dim3(kTileThreads), 0, stream>>>(
      d_nodes, d_R, R_bound, P, tile_counts);
  partition_scan_kernel<<<dim3(1), dim3(1024), 0, stream>>>(tile_counts, tiles, P,
                                                            static_cast<uint32_t>(rank), tile_base,
                                                            d_counts);
  partition_scatter_kernel<<<dim3(static_cast<unsigned>(tiles)), dim3(kTileThreads), 0, stream>>>(
      d_nodes, d_ts, d_R, R_bound, P, tile_base, d_requests, d_pos);
  GF_HIP(hipGetLastError());
}

void partition_plan(const int64_t* d_nodes, const float* d_ts, size_t R, int world_size, int rank,
                    int64_t* d_requests, uint32_t* d_pos, uint64_t* d_counts, void* d_scratch,
                    size_t scratch_bytes, int device, hipStream_t stream) {
  partition_plan_dev(d_nodes, d_ts, nullptr, R, world_size, rank, d_requests, d_pos, d_counts,
                     d_scratch, scratch_bytes, device, stream);
}

}  // namespace gf
