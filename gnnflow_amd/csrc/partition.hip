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
  const size_t tiles = (R_bound + kTileThreads - 1) / kTileThreads;
  uint32_t* tile_counts = static_cast<uint32_t*>(d_scratch);
  uint32_t* tile_base = reinterpret_cast<uint32_t*>(
      static_cast<char*>(d_scratch) + align_up(tiles * world_size * sizeof(uint32_t), 16));
  const uint32_t P = static_cast<uint32_t>(world_size);
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
