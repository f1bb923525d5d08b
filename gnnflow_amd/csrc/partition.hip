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
// Layers of up to 32 768 roots: one launch (partition_plan_fused_kernel).  Larger ones: three
// short launches — per-tile owner histograms (wave ballots), one-workgroup scan of the tile
// table, stable scatter.  Integer / byte work, no atomics.
//
// Slotted form (stride != 0; the exchange without a host synchronisation): owner q's run
// starts at the FIXED row q * stride — row q * stride itself is the slot's header
// {rows in the slot, bit 0 = "a slot of the sending rank overflowed"} and up to
// cap = stride - 1 requests follow — and the rank's own share starts at row P * stride, so the
// send buffer is P equal slots whatever the counts are (an equal-split all-to-all needs no
// sizes on the host).  Roots beyond a slot's capacity are not written (their pos is the
// slot's header row, which no reply ever fills): the sample is flagged and sampled again
// through the variable-size exchange.
#include "common.hpp"
#include "owner_hash.hpp"
#include "partition.hpp"

#include <cstdint>
#include <cstdlib>

namespace gf {
namespace {

constexpr int kTileThreads = 256;      // one root per thread per tile
constexpr int kMaxParts = 64;

// tile_counts[tile][o] = roots of owner o in the tile
__global__ __launch_bounds__(kTileThreads) void partition_count_kernel(
    const int64_t* __restrict__ nodes, const uint64_t* __restrict__ d_R, uint64_t R_host,
    OwnerDiv od, uint32_t* __restrict__ tile_counts) {
  const uint32_t P = od.P;
  __shared__ uint32_t wave_cnt[kTileThreads / 64][kMaxParts];
  const uint64_t R = d_R ? *d_R : R_host;   // device-resident count: a chained layer
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kTileThreads + threadIdx.x;
  const uint32_t o = i < R ? owner_of(nodes[i], od) : P;   // P = "no root"
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
    uint32_t* __restrict__ tile_base, uint64_t* __restrict__ counts, uint32_t stride,
    int64_t* __restrict__ requests, uint32_t* __restrict__ d_overflow, int overflow_store) {
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
    if (stride) {
      const uint32_t cap = stride - 1;
      uint32_t ovf = 0;
      for (uint32_t o = 0; o < P; ++o) {
        start_s[o] = o == rank ? P * stride : o * stride + 1;
        if (o != rank && total_s[o] > cap) ovf = 1;
      }
      for (uint32_t o = 0; o < P; ++o) {
        requests[2 * static_cast<uint64_t>(o) * stride] =
            o == rank ? 0 : static_cast<int64_t>(min(total_s[o], cap));
        requests[2 * static_cast<uint64_t>(o) * stride + 1] = ovf;
      }
      if (overflow_store) *d_overflow = ovf;
      else if (ovf) atomicOr(d_overflow, 1u);
    } else {
      uint32_t at = 0;
      for (uint32_t o = 0; o < P; ++o)
        if (o != rank) { start_s[o] = at; at += total_s[o]; }
      start_s[rank] = at;
    }
    for (uint32_t o = 0; o < P; ++o) counts[o] = total_s[o];
  }
  __syncthreads();
  for (uint64_t k = tid; k < tiles * P; k += 1024) tile_base[k] += start_s[k % P];
}

__global__ __launch_bounds__(kTileThreads) void partition_scatter_kernel(
    const int64_t* __restrict__ nodes, const float* __restrict__ ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, OwnerDiv od,
    const uint32_t* __restrict__ tile_base, int64_t* __restrict__ requests,
    uint32_t* __restrict__ pos, uint32_t* __restrict__ root_of, uint32_t stride, uint32_t rank) {
  const uint32_t P = od.P;
  __shared__ uint32_t wave_cnt[kTileThreads / 64][kMaxParts];
  const uint64_t R = d_R ? *d_R : R_host;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kTileThreads + threadIdx.x;
  const int64_t v = i < R ? nodes[i] : 0;
  const uint32_t o = i < R ? owner_of(v, od) : P;
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
  if (stride && o != rank && p - (o * stride + 1) >= stride - 1) {   // beyond the slot
    pos[i] = o * stride;
    return;
  }
  requests[2 * static_cast<uint64_t>(p)] = v;
  requests[2 * static_cast<uint64_t>(p) + 1] = static_cast<int64_t>(__float_as_uint(ts[i]));
  pos[i] = p;
  if (root_of) root_of[p] = static_cast<uint32_t>(i);
}

// Layers of up to 32 768 roots: the whole plan in ONE launch, no scratch, no kernel boundary.
// A workgroup owns a tile of 1024 consecutive roots, and instead of waiting for the other
// tiles' histograms it RECOUNTS them: it hashes every root of the layer itself (coalesced
// 8-byte loads out of L2 — 20 workgroups x 160 KB for a 19 800-root layer — and ~10 integer
// ops per root), keeping per owner the number of roots in the tiles before its own and in the
// whole layer.  That is O(R^2 / 1024) hashes in all, 0.4 M for that layer: less time than one
// kernel boundary, and three launches (count / scan / scatter: 11.4 us on the device, ~14 us
// of host enqueue) become one.  Lane q of every wave keeps owner q's two counters, so up to 64
// owners need no LDS atomics.
constexpr int kFusedThreads = 1024;
constexpr uint32_t kFusedPlanRoots = 32768;
__device__ unsigned long long g_part_reused_roots;   // roots plans did not request (diagnostics)

// slot_mul / slot_add / own_base: several samples sharing one request buffer (PlanJob); a single
// sample: 1, 0, P * stride.
__device__ inline void plan_fused_body(
    const int64_t* __restrict__ nodes, const float* __restrict__ ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, OwnerDiv od, uint32_t rank,
    int64_t* __restrict__ requests, uint32_t* __restrict__ pos, uint64_t* __restrict__ counts,
    uint32_t* __restrict__ root_of, uint32_t stride, uint32_t* __restrict__ d_overflow,
    int overflow_store, uint32_t slot_mul, uint32_t slot_add, uint32_t own_base,
    uint32_t force_overflow = 0, const uint64_t* __restrict__ d_skip = nullptr,
    uint64_t skip_host = 0) {
  const uint32_t P = od.P;
  // roots [0, skip) are not requested (PlanJob::d_skip): they count for nobody
  const uint32_t skip = static_cast<uint32_t>(d_skip ? *d_skip : skip_host);
  if (skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_part_reused_roots, skip);
  __shared__ uint32_t s_before[kFusedThreads / 64][kMaxParts];
  __shared__ uint32_t s_total[kFusedThreads / 64][kMaxParts];
  __shared__ uint32_t s_tile[kFusedThreads / 64][kMaxParts];   // this tile, per wave
  __shared__ uint32_t s_start[kMaxParts];                       // first output row per owner
  const uint32_t R = static_cast<uint32_t>(d_R ? *d_R : R_host);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tile = blockIdx.x;
  if (tile * kFusedThreads >= R && !(tile == 0)) return;   // uniform; tile 0 still reports counts
  // 1. every root of the layer, tile by tile; my own tile's owner stays in a register
  uint32_t acc_before = 0, acc_total = 0;   // lane q: owner q, over the roots this WAVE hashed
  uint32_t my_owner = P;
  int64_t my_node = 0;
  const uint32_t tiles = (R + kFusedThreads - 1) / kFusedThreads;
  constexpr uint32_t kBatch = 8;   // tiles whose loads are in flight together
  for (uint32_t k0 = 0; k0 < tiles; k0 += kBatch) {
    int64_t v[kBatch];
#pragma unroll
    for (uint32_t b = 0; b < kBatch; ++b) {
      const uint32_t i = (k0 + b) * kFusedThreads + tid;
      v[b] = i < R ? nodes[i] : 0;
    }
#pragma unroll
    for (uint32_t b = 0; b < kBatch; ++b) {
      const uint32_t k = k0 + b;
      const uint32_t i = k * kFusedThreads + tid;
      const uint32_t o = (i < R && i >= skip) ? owner_of(v[b], od) : P;   // P = "no root"
      if (k == tile) { my_owner = o; my_node = v[b]; }
      uint32_t mine = 0;
      for (uint32_t q = 0; q < P; ++q) {
        const uint32_t c = __popcll(__ballot(o == q));
        mine = lane == static_cast<int>(q) ? c : mine;
      }
      acc_total += mine;
      if (k < tile) acc_before += mine;
    }
  }
  if (lane < static_cast<int>(P)) {
    s_before[wave][lane] = acc_before;
    s_total[wave][lane] = acc_total;
  }
  // 2. my tile: roots of the same owner before me
  uint32_t before_in_wave = 0;
  for (uint32_t q = 0; q < P; ++q) {
    const unsigned long long m = __ballot(my_owner == q);
    if (lane == 0) s_tile[wave][q] = __popcll(m);
    if (my_owner == q) before_in_wave = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  // 3. owner q's run starts behind the runs of the owners that precede it in the output order
  //    (other owners ascending, then `rank`); thread q < P adds up the waves' counters
  if (tid < static_cast<int>(P)) {
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kFusedThreads / 64; ++w) {
      before += s_before[w][tid];
      total += s_total[w][tid];
    }
    s_before[0][tid] = before;
    s_total[0][tid] = total;
    if (tile == 0) counts[tid] = total;
  }
  __syncthreads();
  if (tid == 0) {
    if (stride) {
      const uint32_t cap = stride - 1;
      uint32_t ovf = force_overflow ? 1u : 0u;
      for (uint32_t o = 0; o < P; ++o) {
        s_start[o] = o == rank ? own_base : (o * slot_mul + slot_add) * stride + 1;
        if (o != rank && s_total[0][o] > cap) ovf = 1;
      }
      if (tile == 0) {   // the slots' headers; the sample-wide flag
        for (uint32_t o = 0; o < P; ++o) {
          const uint64_t h = static_cast<uint64_t>(o * slot_mul + slot_add) * stride;
          requests[2 * h] = o == rank ? 0 : static_cast<int64_t>(min(s_total[0][o], cap));
          requests[2 * h + 1] = ovf;
        }
        if (overflow_store) *d_overflow = ovf;
        else if (ovf) atomicOr(d_overflow, 1u);
      }
    } else {
      uint32_t at = 0;
      for (uint32_t o = 0; o < P; ++o)
        if (o != rank) { s_start[o] = at; at += s_total[0][o]; }
      s_start[rank] = at;
    }
  }
  __syncthreads();
  const uint32_t i = tile * kFusedThreads + tid;
  if (i >= R) return;
  if (i < skip) {
    pos[i] = kPosReused;
    return;
  }
  uint32_t p = s_start[my_owner] + s_before[0][my_owner] + before_in_wave;
  for (int w = 0; w < wave; ++w) p += s_tile[w][my_owner];
  if (stride && my_owner != rank && p - s_start[my_owner] >= stride - 1) {   // beyond the slot
    pos[i] = (my_owner * slot_mul + slot_add) * stride;
    return;
  }
  requests[2 * static_cast<uint64_t>(p)] = my_node;
  requests[2 * static_cast<uint64_t>(p) + 1] = static_cast<int64_t>(__float_as_uint(ts[i]));
  pos[i] = p;
  if (root_of) root_of[p] = i;   // the inverse: which root a request / reply row belongs to
}

__global__ __launch_bounds__(kFusedThreads) void partition_plan_fused_kernel(
    const int64_t* __restrict__ nodes, const float* __restrict__ ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, OwnerDiv od, uint32_t rank,
    int64_t* __restrict__ requests, uint32_t* __restrict__ pos, uint64_t* __restrict__ counts,
    uint32_t* __restrict__ root_of, uint32_t stride, uint32_t* __restrict__ d_overflow,
    int overflow_store) {
  plan_fused_body(nodes, ts, d_R, R_host, od, rank, requests, pos, counts, root_of, stride,
                  d_overflow, overflow_store, 1u, 0u, od.P * stride);
}

// Up to four samples' plans in ONE launch (blockIdx.y picks the job): the chain of a
// partitioned sample is bound by the host thread that issues its launches, so samples that
// share their launches and exchanges divide that cost (sampler.hip sample_partitioned_group).
struct PlanJobs { PlanJob j[4]; };
__global__ __launch_bounds__(kFusedThreads) void partition_plan_jobs_kernel(
    PlanJobs jobs, OwnerDiv od, uint32_t rank, uint32_t stride) {
  const PlanJob& j = jobs.j[blockIdx.y];
  plan_fused_body(j.nodes, j.ts, j.d_R, j.R_host, od, rank, j.requests, j.pos, j.counts, nullptr,
                  stride, j.d_overflow, j.overflow_store, j.slot_mul, j.slot_add, j.own_base,
                  j.force_overflow, j.d_skip, j.skip_host);
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
                        size_t scratch_bytes, int device, hipStream_t stream,
                        uint32_t* d_root_of, uint32_t stride, uint32_t* d_overflow,
                        int overflow_store) {
  GF_REQUIRE(world_size >= 1 && world_size <= kMaxParts, "partition: world size must be 1..64");
  GF_REQUIRE(stride == 0 || (stride >= 2 && d_overflow != nullptr),
             "partition: the slotted form needs a stride >= 2 and an overflow word");
  GF_REQUIRE(static_cast<uint64_t>(stride) * world_size + R_bound < 0xFFFFFFFFull,
             "partition: more than 2^32-1 request rows");
  GF_REQUIRE(rank >= 0 && rank < world_size, "partition: rank out of range");
  GF_REQUIRE(d_counts != nullptr, "partition: null counts");
  GF_REQUIRE(R_bound < 0xFFFFFFFFull, "partition: more than 2^32-1 roots");
  DeviceGuard dg(device);
  const bool no_roots = R_bound == 0;
  if (no_roots && !stride) {
    GF_HIP(hipMemsetAsync(d_counts, 0, world_size * sizeof(uint64_t), stream));
    return;
  }
  // slotted: a rank without roots still writes its (empty) slot headers and the flag — one
  // workgroup of the fused kernel with R = 0
  if (no_roots) { R_bound = 1; d_R = nullptr; }
  GF_REQUIRE(no_roots || (d_nodes && d_ts), "partition: null pointer");
  GF_REQUIRE(d_requests && d_pos && d_scratch, "partition: null pointer");
  GF_REQUIRE(scratch_bytes >= partition_scratch_bytes(R_bound, world_size),
             "partition: scratch buffer too small");
  const uint32_t P = static_cast<uint32_t>(world_size);
  const OwnerDiv od = owner_div(P);
  static const bool fused_plan = [] {
    const char* v = std::getenv("GNNFLOW_PARTITION_SMALL_PLAN");   // tests: 0 = tiled form only
    return !(v && std::atoi(v) == 0);
  }();
  if ((fused_plan || no_roots) && R_bound <= kFusedPlanRoots) {
    const unsigned grid = static_cast<unsigned>((R_bound + kFusedThreads - 1) / kFusedThreads);
    partition_plan_fused_kernel<<<dim3(grid), dim3(kFusedThreads), 0, stream>>>(
        d_nodes, d_ts, d_R, no_roots ? 0 : R_bound, od, static_cast<uint32_t>(rank), d_requests,
        d_pos, d_counts, d_root_of, stride, d_overflow, overflow_store);
    GF_HIP(hipGetLastError());
    return;
  }
  const size_t tiles = (R_bound + kTileThreads - 1) / kTileThreads;
  uint32_t* tile_counts = static_cast<uint32_t*>(d_scratch);
  uint32_t* tile_base = reinterpret_cast<uint32_t*>(
      static_cast<char*>(d_scratch) + align_up(tiles * world_size * sizeof(uint32_t), 16));
  partition_count_kernel<<<dim3(static_cast<unsigned>(tiles)), dim3(kTileThreads), 0, stream>>>(
      d_nodes, d_R, R_bound, od, tile_counts);
  partition_scan_kernel<<<dim3(1), dim3(1024), 0, stream>>>(
      tile_counts, tiles, P, static_cast<uint32_t>(rank), tile_base, d_counts, stride, d_requests,
      d_overflow, overflow_store);
  partition_scatter_kernel<<<dim3(static_cast<unsigned>(tiles)), dim3(kTileThreads), 0, stream>>>(
      d_nodes, d_ts, d_R, R_bound, od, tile_base, d_requests, d_pos, d_root_of, stride,
      static_cast<uint32_t>(rank));
  GF_HIP(hipGetLastError());
}

void partition_plan_jobs(const PlanJob* jobs, int n, size_t R_bound, int world_size, int rank,
                         uint32_t stride, int device, hipStream_t stream) {
  GF_REQUIRE(jobs != nullptr && n >= 1 && n <= 4, "partition: 1..4 plan jobs");
  GF_REQUIRE(world_size >= 1 && world_size <= kMaxParts, "partition: world size must be 1..64");
  GF_REQUIRE(stride >= 2, "partition: plan jobs need the slotted form");
  GF_REQUIRE(rank >= 0 && rank < world_size, "partition: rank out of range");
  GF_REQUIRE(R_bound <= kFusedPlanRoots, "partition: plan jobs are for layers of <= 32 768 roots");
  static_assert(kPlanJobsMaxRoots == kFusedPlanRoots, "one limit");
  for (int k = 0; k < n; ++k)
    GF_REQUIRE(jobs[k].requests && jobs[k].pos && jobs[k].counts && jobs[k].d_overflow,
               "partition: null pointer in a plan job");
  DeviceGuard dg(device);
  const OwnerDiv od = owner_div(static_cast<uint32_t>(world_size));
  const unsigned grid = static_cast<unsigned>(
      (std::max<size_t>(R_bound, 1) + kFusedThreads - 1) / kFusedThreads);
  PlanJobs all;
  for (int k = 0; k < 4; ++k) all.j[k] = jobs[std::min(k, n - 1)];
  partition_plan_jobs_kernel<<<dim3(grid, static_cast<unsigned>(n)), dim3(kFusedThreads), 0,
                               stream>>>(all, od, static_cast<uint32_t>(rank), stride);
  GF_HIP(hipGetLastError());
}

// Roots that plans did NOT bucket / request because the previous layer's block already holds
// their edges (PlanJob::d_skip), since the library was loaded, on the current device.
uint64_t part_reused_roots() {
  unsigned long long v = 0;
  GF_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_part_reused_roots), sizeof(v)));
  return v;
}

void partition_plan(const int64_t* d_nodes, const float* d_ts, size_t R, int world_size, int rank,
                    int64_t* d_requests, uint32_t* d_pos, uint64_t* d_counts, void* d_scratch,
                    size_t scratch_bytes, int device, hipStream_t stream) {
  partition_plan_dev(d_nodes, d_ts, nullptr, R, world_size, rank, d_requests, d_pos, d_counts,
                     d_scratch, scratch_bytes, device, stream, nullptr, 0, nullptr, 0);
}

}  // namespace gf
