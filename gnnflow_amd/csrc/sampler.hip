// Temporal neighbour sampling on MI355X (gfx950).
//
// Replaces gnnflow/csrc/sampling_kernels.cu (SampleLayerRecentKernel :11-107,
// SampleLayerUniformKernel :109-273), the thrust::remove_if compaction
// (temporal_sampler.cu:191-204) and the host-side result assembly
// (temporal_sampler.cu:236-274) with three launches per (layer, snapshot), all
// device resident:
//
//   1. search : one 16-lane group per root (4 lanes in large layers).  Resolves the root's time
//               window ONCE (the reference repeats the block walk and both binary
//               searches in each of the F slot threads) with a group-wide k-ary
//               search over the node's flat timestamp segment: every round the
//               group's lanes probe GROUP pivots in parallel and a ballot/popcount
//               picks the sub-range, so a window over 4096 edges is found in 2
//               dependent memory round trips at GROUP=64 instead of 12.
//   2. scan   : exclusive prefix sum of the per-root valid-slot counts (wave
//               shuffles + LDS), which gives every root its base in the compacted
//               output and the layer's edge count S (and R' = R + S, the next
//               layer's root count, which never leaves HBM).
//   3. emit   : one thread per (root, slot): reads the selected edge (one 32 B
//               {dst, eid, ts} record = one DRAM sector) and writes the final MFG
//               arrays (all_nodes, all_timestamps, delta_timestamps, eids, row, col)
//               directly at base[root] + slot, i.e. already compacted, root-major,
//               newest first — the order thrust's stable remove_if leaves.
//
// Equivalence with the reference's per-block case analysis
// (sampling_kernels.cu:55-86): for chronologically ingested edges the union over
// blocks of [LowerBound(start), LowerBound(end)) equals
// [lower_bound(start), lower_bound(end)) on the node's concatenated sequence, and
// "j-th most recent, spilling to the previous block" (:88-92) is index
// end-1-j on that sequence.  tests/ proves it against the block-walking oracle.
#include "sampler.hpp"
#include "partition.hpp"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/gnnflow_rng.h"

namespace gf {

namespace {

constexpr int kSearchThreads = 256;
constexpr int kEmitThreads = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 4;  // per thread per tile
constexpr size_t kSmallRoots = 32768;  // layers up to this many roots skip the scan launch
constexpr uint32_t kGranuleSpins = 1u << 12;   // ~ a few ms of polling before a tile is recounted
static size_t kLaneSearchRoots = [] {   // layers from this many roots: lane-per-root pass
  const char* v = std::getenv("GNNFLOW_LANE_SEARCH_MIN_ROOTS");   // tuning
  return v ? static_cast<size_t>(std::atoll(v)) : (size_t{1} << 20);
}();
constexpr uint32_t kMaxHubSegs = 2048;         // = the lane pass's largest grid

// sampling_kernels.cu:28-40
__device__ inline void time_window(float root_ts, uint32_t snapshot_idx,
                                   uint32_t num_snapshots, float window, float* start,
                                   float* end) {
  if (num_snapshots == 1) {
    *start = (fabs(static_cast<double>(window)) < 1e-6) ? 0.0f : root_ts - window;
    *end = root_ts;
  } else {
    float k = static_cast<float>(num_snapshots - snapshot_idx - 1);
    *end = fmaf(-k, window, root_ts);  // nvcc contracts `t - k*w` (see oracle)
    *start = *end - window;
  }
}

template <int GROUP>
__device__ inline uint32_t group_count(bool pred, int group_in_wave) {
  unsigned long long m = __ballot(pred);
  if (GROUP == 64) return __popcll(m);
  return __popcll((m >> (group_in_wave * GROUP)) & ((1ull << GROUP) - 1ull));
}

// First index in [0, n) with ts[idx] >= x (utils.cu:96-109 LowerBound), evaluated
// cooperatively by a GROUP-lane group; every lane returns the result.
template <int GROUP>
__device__ inline uint32_t lower_bound_group(const float* __restrict__ ts, uint32_t n,
                                             float x, int lane, int group_in_wave) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > GROUP) {
    uint32_t span = hi - lo;
    uint32_t stride = (span + GROUP - 1) / GROUP;
    uint32_t p = lo + lane * stride;
    bool less = (p < hi) && (ts[p] < x);
    uint32_t c = group_count<GROUP>(less, group_in_wave);
    if (c == 0) {
      hi = lo;
    } else {
      uint32_t nlo = lo + (c - 1) * stride + 1;
      uint32_t nhi = lo + c * stride;
      hi = nhi < hi ? nhi : hi;
      lo = nlo;
    }
  }
  uint32_t p = lo + lane;
  bool less = (p < hi) && (ts[p] < x);
  return lo + group_count<GROUP>(less, group_in_wave);
}

// The same lower bound over the segment [s, s + n) of the timestamp pool, through the fences
// (edge_store.hpp: fence_l[g] = ts_pool[(g + 1) * 16^l - 1], global positions): from the
// coarsest level whose blocks are smaller than the segment down to level 1, every round takes
// the (<= 17) fences whose positions lie inside the current range as pivots — they are
// CONSECUTIVE entries of the level, i.e. one or two 64-byte lines, read by the group as
// contiguous 16-byte / 4-byte loads — and narrows the range to the gap between two of them;
// the last <= 16-element gap is resolved on the timestamps themselves.  ceil(log16 n) rounds
// of one line each, where the strided k-ary search reads GROUP sectors per round
// (sample_search_kernel<4> on the 10 M-node graph: 2.4-4.6x the algorithmic bytes).
template <int GROUP>
__device__ inline uint32_t lower_bound_fenced(const GraphView& g, uint64_t s, uint32_t n, float x,
                                              int lane, int group_in_wave) {
  constexpr int V = 16 / GROUP;   // consecutive values per lane: the group covers 16 per round
  uint64_t lo = s, hi = s + n;    // the answer lies in [lo, hi]
  if (g.fence.levels == 0)   // small layers (view_for), or fences switched off
    return lower_bound_group<GROUP>(g.ts_pool + s, n, x, lane, group_in_wave);
  if (n > 16) {
    int top = (31 - __clz(n - 1)) >> 2;   // coarsest level with 16^top < n
    if (top > static_cast<int>(g.fence.levels)) top = g.fence.levels;
    for (int l = top; l >= 1; --l) {
      const int shift = 4 * l;
      const float* __restrict__ F = g.fence.base + g.fence.off[l - 1];
      // fences whose position ((b + 1) << shift) - 1 lies in [lo, hi)
      uint64_t b_first = ((lo + (1ull << shift)) >> shift) - 1;
      const uint64_t b_end = hi >> shift;   // one past the last
      while (b_first < b_end) {              // at most two rounds per level
        // aligned window of 16 fences (the levels are padded: the whole window is readable)
        const uint64_t w0 = b_first & ~3ull;
        const float* __restrict__ src = F + w0 + static_cast<uint64_t>(lane) * V;
        float val[V];
        if (V == 4) {
          const float4 f = *reinterpret_cast<const float4*>(src);
          val[0] = f.x; val[1 % V] = f.y; val[2 % V] = f.z; val[3 % V] = f.w;
        } else if (V == 2) {
          const float2 f = *reinterpret_cast<const float2*>(src);
          val[0] = f.x; val[1 % V] = f.y;
        } else {
#pragma unroll
          for (int v = 0; v < V; ++v) val[v] = src[v];
        }
        uint32_t mine = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const uint64_t b = w0 + static_cast<uint64_t>(lane) * V + v;
          mine += (b >= b_first && b < b_end && val[v] < x) ? 1u : 0u;
        }
        // sum over the group's lanes
        uint32_t less = mine;
#pragma unroll
        for (int d = 1; d < GROUP; d <<= 1) less += __shfl_xor(less, d, 64);
        const uint64_t seen = min(b_end, w0 + 16) - b_first;   // pivots looked at
        if (less < seen) {   // the (less)-th pivot is the first one >= x
          hi = ((b_first + less + 1) << shift) - 1;
          if (less) lo = (b_first + less) << shift;
          break;
        }
        lo = (b_first + seen) << shift;   // all of them < x
        b_first += seen;
      }
    }
  }
  // the remaining gap (<= 16 elements below a level-1 fence; a whole small segment): 16
  // consecutive timestamps per round
  const float* __restrict__ ts = g.ts_pool;
  for (;;) {
    uint32_t mine = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const uint64_t p = lo + static_cast<uint64_t>(lane) * V + v;
      mine += (p < hi && ts[p] < x) ? 1u : 0u;
    }
    uint32_t less = mine;
#pragma unroll
    for (int d = 1; d < GROUP; d <<= 1) less += __shfl_xor(less, d, 64);
    const uint64_t m = min<uint64_t>(hi - lo, 16);
    if (less < m || lo + 16 >= hi) return static_cast<uint32_t>(lo + less - s);
    lo += 16;
  }
}

// The window [start, end) of one root on its node's segment, with the two shortcuts the node
// entry allows (edge_store.hpp): `end` later than the node's newest edge -> hi = size; a window
// that starts at 0 on a graph without negative timestamps -> lo = 0.  Otherwise the searches.
template <int GROUP>
__device__ inline void window_bounds(const GraphView& g, const NodeEntry& e, float start, float end,
                                     int lane, int group_in_wave, uint32_t* lo_out,
                                     uint32_t* hi_out) {
  uint32_t hi;
  if (g.nonneg_ts >= 0 && end > __uint_as_float(e.last_ts_bits)) hi = e.size;
  else hi = lower_bound_fenced<GROUP>(g, e.start, e.size, end, lane, group_in_wave);
  uint32_t lo = 0;
  if (!(g.nonneg_ts > 0 && start <= 0.0f) && hi > 0) {
    const float first = g.ts_pool[e.start];
    if (start > first) lo = lower_bound_fenced<GROUP>(g, e.start, hi, start, lane, group_in_wave);
  }
  *lo_out = lo;
  *hi_out = hi;
}

__device__ inline uint32_t valid_slots(uint32_t n_cand, uint32_t fanout, int uniform) {
  // recent: slot j valid iff j < #candidates (sampling_kernels.cu:88-104);
  // uniform: every slot valid iff there is a candidate (:202, with replacement)
  if (uniform) return n_cand ? fanout : 0u;
  return n_cand < fanout ? n_cand : fanout;
}

// Size read-back without a memcpy + event wait: the last kernel of a sample() copies the
// per-block {R, S} words into pinned host memory and then stores the call's sequence
// number; the host spins on that word (hipEventSynchronize wakes up 10-20 us late).
struct Publish {
  const uint64_t* d_counts;   // device counts array (all blocks of this sample)
  uint64_t* h_counts;         // pinned host mirror (device-mapped)
  uint64_t* h_flag;           // pinned host sequence word
  uint64_t seq;
  uint32_t num_words;         // 0 = nothing to publish
  const uint32_t* d_extra = nullptr;   // one more word behind the counts (slot overflow), or null
};

// Stream-ordered after the last emit kernel, so every output of the sample is complete
// (and released by the kernel boundary) before the host can observe the sequence word.
__global__ void sample_publish_kernel(Publish p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (uint32_t i = 0; i < p.num_words; ++i) p.h_counts[i] = p.d_counts[i];
  p.h_counts[p.num_words] = p.d_extra ? *p.d_extra : 0;
  __threadfence_system();
  *reinterpret_cast<volatile uint64_t*>(p.h_flag) = p.seq;
}

struct PublishGroup { Publish p[4]; };
__global__ void sample_publish_group_kernel(PublishGroup g) {
  if (threadIdx.x != 0) return;
  const Publish& p = g.p[blockIdx.x];
  for (uint32_t i = 0; i < p.num_words; ++i) p.h_counts[i] = p.d_counts[i];
  p.h_counts[p.num_words] = p.d_extra ? *p.d_extra : 0;
  __threadfence_system();
  *reinterpret_cast<volatile uint64_t*>(p.h_flag) = p.seq;
}

// ---- 1. search --------------------------------------------------------------------
template <int GROUP>
__global__ __launch_bounds__(kSearchThreads) void sample_search_kernel(
    GraphView g, const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t snapshot_idx,
    uint32_t num_snapshots, float window, uint64_t* __restrict__ rec_end,
    uint32_t* __restrict__ rec_cnt, uint32_t fanout, int uniform,
    uint32_t* __restrict__ wg_sum, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ seg_count, uint32_t num_segs, uint32_t seg_cap) {
  __shared__ uint32_t s_sum;
  __shared__ uint32_t seg_prefix[kMaxHubSegs + 1];
  if (wg_sum && threadIdx.x == 0) s_sum = 0;
  if (wg_sum) __syncthreads();
  // list != null: only the hubs the lane-per-root pass of a large layer left over — segment
  // s of the worklist holds seg_count[s] root indices at list[s * seg_cap ...]; every
  // workgroup builds the exclusive prefix of the counts (<= kMaxHubSegs words) in LDS and
  // finds the segment of its i-th hub by binary search, so the hubs are spread evenly over
  // the groups wherever they sat in the batch.
  uint64_t R = d_R ? *d_R : R_host;
  if (list) {
    __shared__ uint32_t wtot[kSearchThreads / 64];
    uint32_t carry = 0;
    for (uint32_t s0 = 0; s0 < num_segs; s0 += kSearchThreads) {   // uniform trip count
      const uint32_t sidx = s0 + threadIdx.x;
      const uint32_t v = sidx < num_segs ? seg_count[sidx] : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((threadIdx.x & 63) >= d) incl += up;
      }
      __syncthreads();
      if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
      __syncthreads();
      uint32_t wbase = 0, tot = 0;
      for (int w = 0; w < kSearchThreads / 64; ++w) {
        if (w < (threadIdx.x >> 6)) wbase += wtot[w];
        tot += wtot[w];
      }
      if (sidx < num_segs) seg_prefix[sidx] = carry + wbase + incl - v;
      carry += tot;
    }
    if (threadIdx.x == 0) seg_prefix[num_segs] = carry;
    __syncthreads();
    R = seg_prefix[num_segs];
  }
  constexpr int kGroupsPerBlock = kSearchThreads / GROUP;
  const int lane = threadIdx.x % GROUP;
  const int group_in_wave = (threadIdx.x % 64) / GROUP;
  const uint64_t group = static_cast<uint64_t>(blockIdx.x) * kGroupsPerBlock + threadIdx.x / GROUP;
  const uint64_t num_groups = static_cast<uint64_t>(gridDim.x) * kGroupsPerBlock;
  for (uint64_t i = group; i < R; i += num_groups) {
    uint64_t r = i;
    if (list) {   // largest segment s with seg_prefix[s] <= i
      uint32_t lo_s = 0, hi_s = num_segs;
      while (hi_s - lo_s > 1) {
        const uint32_t mid = (lo_s + hi_s) >> 1;
        if (seg_prefix[mid] <= i) lo_s = mid; else hi_s = mid;
      }
      r = list[static_cast<uint64_t>(lo_s) * seg_cap + (i - seg_prefix[lo_s])];
    }
    const int64_t nid = roots[r];
    const float t = root_ts[r];
    float start, end;
    time_window(t, snapshot_idx, num_snapshots, window, &start, &end);
    uint64_t end_off = 0;
    uint32_t n_cand = 0;
    if (nid >= 0 && static_cast<uint64_t>(nid) < g.table_len) {
      const NodeEntry e = g.table[nid];
      if (e.size > 0) {
        uint32_t lo, hi;
        window_bounds<GROUP>(g, e, start, end, lane, group_in_wave, &lo, &hi);
        n_cand = hi > lo ? hi - lo : 0;
        end_off = e.start + hi;
      }
    }
    if (lane == 0) {
      rec_end[r] = end_off;
      rec_cnt[r] = n_cand;
      if (wg_sum) atomicAdd(&s_sum, valid_slots(n_cand, fanout, uniform));
    }
  }
  // small-batch path: the grid covers every root exactly once (no striding), so
  // workgroup b owns roots [b*kGroupsPerBlock, (b+1)*kGroupsPerBlock) and publishes
  // their valid-slot total for the emit kernel's prefix
  if (wg_sum) {
    __syncthreads();
    if (threadIdx.x == 0) wg_sum[blockIdx.x] = s_sum;
  }
}

// ---- 1b. search for large layers: lane per root, then groups for the hubs ----------------
// A 16-lane group per root keeps only 4 roots per wave in flight, and a root is a chain of
// 2-7 dependent random reads (table entry -> pivots ...): at 10^5-10^7 roots per layer the
// kernel is bound by that latency, not by HBM (measured on the 10 M-node / 200 M-edge graph:
// 32 G random reads/s against > 100 G/s in the emit kernel).  On a power-law graph > 90 % of
// the roots have at most one 64-byte line of timestamps, so a first pass gives every LANE a
// root (64 table entries in flight per wave) and resolves it on the spot if its segment has
// <= kLaneDeg timestamps (all loads independent: one more round trip); the roots with longer
// segments are appended to the workgroup's own segment of a worklist (an LDS counter: one
// global atomic per wave on a shared counter would serialise at ~88 per microsecond) that a
// second launch of the cooperative k-ary search works off, evenly spread over its groups
// whatever their position in the batch.
constexpr uint32_t kLaneDeg = 16;

__global__ __launch_bounds__(kSearchThreads) void sample_search_lanes_kernel(
    GraphView g, const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t snapshot_idx,
    uint32_t num_snapshots, float window, uint64_t* __restrict__ rec_end,
    uint32_t* __restrict__ rec_cnt, uint32_t* __restrict__ hub_list,
    uint32_t* __restrict__ seg_count, uint32_t seg_cap) {
  __shared__ uint32_t s_seg_n;
  if (threadIdx.x == 0) s_seg_n = 0;
  __syncthreads();
  uint32_t* seg = hub_list + static_cast<uint64_t>(blockIdx.x) * seg_cap;
  const uint64_t R = d_R ? *d_R : R_host;
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * kSearchThreads + threadIdx.x) >> 6;
  const uint64_t num_waves = (static_cast<uint64_t>(gridDim.x) * kSearchThreads) >> 6;
  const uint64_t chunks = (R + 63) / 64;
  for (uint64_t chunk = wave; chunk < chunks; chunk += num_waves) {   // wave-uniform trip count
    const uint64_t r = chunk * 64 + lane;
    const bool in = r < R;
    int64_t nid = -1;
    float start = 0.f, end = 0.f;
    if (in) {
      nid = roots[r];
      time_window(root_ts[r], snapshot_idx, num_snapshots, window, &start, &end);
    }
    NodeEntry e;
    e.start = 0;
    e.size = 0;
    if (in && nid >= 0 && static_cast<uint64_t>(nid) < g.table_len) e = g.table[nid];
    // newest edge older than the window's end and the window open at 0: nothing to read
    const bool whole = g.nonneg_ts > 0 && start <= 0.0f && e.size > 0 &&
                       end > __uint_as_float(e.last_ts_bits);
    const bool big = e.size > kLaneDeg && !whole;
    if (in && whole) {
      rec_end[r] = e.start + e.size;
      rec_cnt[r] = e.size;
    } else if (in && !big) {
      uint32_t hi = 0, lo = 0;
      if (e.size > 0) {
        const float* ts = g.ts_pool + e.start;
        float v[kLaneDeg];
        if ((e.start & 3u) == 0) {
          // 16-byte loads (segments start 64-byte aligned unless a prefix was offloaded):
          // a quarter of the L2 requests of the scalar form, which bound this pass
          const float4* t4 = reinterpret_cast<const float4*>(ts);
#pragma unroll
          for (uint32_t q = 0; q < kLaneDeg / 4; ++q) {
            const float4 x = 4 * q < e.size ? t4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
          }
        } else {
#pragma unroll
          for (uint32_t i = 0; i < kLaneDeg; ++i) v[i] = i < e.size ? ts[i] : 0.f;   // independent
        }
#pragma unroll
        for (uint32_t i = 0; i < kLaneDeg; ++i) {
          hi += (i < e.size && v[i] < end) ? 1u : 0u;
          lo += (i < e.size && v[i] < start) ? 1u : 0u;
        }
      }
      rec_end[r] = e.start + hi;
      rec_cnt[r] = hi > lo ? hi - lo : 0;
    }
    const unsigned long long hubs = __ballot(big);
    if (hubs) {   // append to this workgroup's segment: LDS counter, no global atomic
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&s_seg_n, static_cast<uint32_t>(__popcll(hubs)));
      at = __shfl(at, 0, 64);
      if (big) seg[at + __popcll(hubs & ((1ull << lane) - 1ull))] = static_cast<uint32_t>(r);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) seg_count[blockIdx.x] = s_seg_n;
}

// ---- 2. scan -----------------------------------------------------------------------
__global__ __launch_bounds__(kScanThreads) void sample_scan_kernel(
    const uint32_t* __restrict__ rec_cnt, uint32_t* __restrict__ base,
    const uint64_t* d_R, uint64_t R_host, uint32_t fanout, int uniform, uint64_t* out_R,
    uint64_t* out_S, uint64_t* next_R) {
  __shared__ uint32_t wave_sums[kScanThreads / 64];
  __shared__ uint32_t carry_s;
  const uint64_t R = d_R ? *d_R : R_host;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr uint64_t kTile = static_cast<uint64_t>(kScanThreads) * kScanItems;
  for (uint64_t tile = 0; tile < R; tile += kTile) {
    uint32_t v[kScanItems];
    uint32_t local = 0;
    const uint64_t i0 = tile + static_cast<uint64_t>(tid) * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = (i0 + k < R) ? valid_slots(rec_cnt[i0 + k], fanout, uniform) : 0u;
      local += v[k];
    }
    // inclusive scan of `local` across the wave
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_sums[w];
    uint32_t run = carry_s + wave_base + incl - local;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (i0 + k < R) base[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == kScanThreads - 1) carry_s = run;  // last thread's run = tile total + carry
    __syncthreads();
  }
  if (tid == 0) {
    const uint64_t S = carry_s;
    *out_R = R;
    *out_S = S;
    if (next_R) *next_R = R + S;
  }
}

// ---- 2b. parallel scan for large layers --------------------------------------------
// tile = kScanTile roots.  (a) per-tile sums, (b) one workgroup scans the tile sums and
// publishes S / R', (c) every tile scans itself and adds its base.
constexpr int kScanTile = 4096;

__global__ __launch_bounds__(kScanThreads) void sample_tile_sum_kernel(
    const uint32_t* __restrict__ rec_cnt, const uint64_t* d_R, uint64_t R_host,
    uint32_t fanout, int uniform, uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t red[kScanThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t tiles = (R + kScanTile - 1) / kScanTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    uint32_t local = 0;
    const uint64_t i0 = tile * kScanTile + static_cast<uint64_t>(tid) * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (i0 + k < R) local += valid_slots(rec_cnt[i0 + k], fanout, uniform);
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
      for (int w = 0; w < kScanThreads / 64; ++w) t += red[w];
      tile_sum[tile] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kScanThreads) void sample_tile_scan_kernel(
    const uint32_t* __restrict__ tile_sum, uint32_t* __restrict__ tile_base,
    const uint64_t* d_R, uint64_t R_host, uint64_t* out_R, uint64_t* out_S, uint64_t* next_R) {
  __shared__ uint32_t wave_sums[kScanThreads / 64];
  __shared__ uint32_t carry_s;
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t tiles = (R + kScanTile - 1) / kScanTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t t0 = 0; t0 < tiles; t0 += kScanThreads) {
    const uint64_t i = t0 + tid;
    const uint32_t v = i < tiles ? tile_sum[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_sums[w];
    const uint32_t excl = carry_s + wave_base + incl - v;
    if (i < tiles) tile_base[i] = excl;
    __syncthreads();
    if (tid == kScanThreads - 1) carry_s = excl + v;
    __syncthreads();
  }
  if (tid == 0) {
    const uint64_t S = carry_s;
    *out_R = R;
    *out_S = S;
    if (next_R) *next_R = R + S;
  }
}

__global__ __launch_bounds__(kScanThreads) void sample_tile_apply_kernel(
    const uint32_t* __restrict__ rec_cnt, const uint32_t* __restrict__ tile_base,
    const uint64_t* d_R, uint64_t R_host, uint32_t fanout, int uniform,
    uint32_t* __restrict__ base) {
  __shared__ uint32_t wave_sums[kScanThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t tiles = (R + kScanTile - 1) / kScanTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    uint32_t v[kScanItems], local = 0;
    const uint64_t i0 = tile * kScanTile + static_cast<uint64_t>(tid) * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = (i0 + k < R) ? valid_slots(rec_cnt[i0 + k], fanout, uniform) : 0u;
      local += v[k];
    }
    uint32_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_sums[w];
    uint32_t run = tile_base[tile] + wave_base + incl - local;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (i0 + k < R) base[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
  }
}

// ---- 3. emit -----------------------------------------------------------------------
__global__ __launch_bounds__(kEmitThreads) void sample_emit_kernel(
    GraphView g, const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t fanout, int uniform,
    int prop_time, uint64_t seed, uint64_t call, const uint64_t* __restrict__ rec_end,
    const uint32_t* __restrict__ rec_cnt, const uint32_t* __restrict__ base,
    int64_t* __restrict__ all_nodes, float* __restrict__ all_ts, float* __restrict__ dt,
    int64_t* __restrict__ eids, int64_t* __restrict__ row, int64_t* __restrict__ col,
    Publish pub, int unroll) {
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t total = R * fanout;
  if (pub.num_words && blockIdx.x == 0 && threadIdx.x == 0) {   // sizes: final before this launch
    for (uint32_t i = 0; i < pub.num_words; ++i) pub.h_counts[i] = pub.d_counts[i];
    pub.h_counts[pub.num_words] = 0;
  }
  // Four slots per thread and trip, every load of a stage issued before the first use: a
  // sampled edge is ONE random 32-byte record, and what bounds this kernel at large batches is
  // how many of those reads are in flight (HBM's random-access rate), not bytes.  One slot per
  // trip left each wave with a single record read outstanding between two dependent hops
  // (count / end -> record -> stores); GNNFLOW_EMIT_UNROLL=1 is that form.
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const uint64_t first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  constexpr int K = 4;
  for (uint64_t t0 = first; t0 < total; t0 += (unroll ? K : 1) * stride) {
    uint64_t t[K], r[K], end[K];
    uint32_t j[K], n[K], bs[K];
    float rts[K];
    bool in[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      t[k] = t0 + static_cast<uint64_t>(k) * stride;
      in[k] = (k == 0 || unroll) && t[k] < total;
      r[k] = in[k] ? t[k] / fanout : 0;
      j[k] = static_cast<uint32_t>(t[k] - r[k] * fanout);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      n[k] = in[k] ? rec_cnt[r[k]] : 0u;
      end[k] = in[k] ? rec_end[r[k]] : 0;
      bs[k] = in[k] ? base[r[k]] : 0u;
      rts[k] = in[k] ? root_ts[r[k]] : 0.f;
      if (in[k] && t[k] < R) {  // dst nodes come first in all_nodes / all_timestamps
        all_nodes[t[k]] = roots[t[k]];
        all_ts[t[k]] = root_ts[t[k]];
      }
    }
    EdgePair nb[K];
    bool ok[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ok[k] = in[k] && j[k] < valid_slots(n[k], fanout, uniform);
      if (ok[k]) {
        const uint32_t pick = uniform ? gf_philox4x32_10_first(seed, t[k], call) % n[k] : j[k];
        nb[k] = g.nbr_pool[end[k] - 1 - pick];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (!ok[k]) continue;
      const float ets = nb[k].ts;
      const uint64_t o = static_cast<uint64_t>(bs[k]) + j[k];
      all_nodes[R + o] = nb[k].dst;
      all_ts[R + o] = prop_time ? rts[k] : ets;
      dt[o] = rts[k] - ets;
      eids[o] = nb[k].eid;
      row[o] = static_cast<int64_t>(r[k]);
      col[o] = static_cast<int64_t>(R + o);
    }
  }
}

// ---- 2+3 fused (small batches): emit with an in-kernel prefix -----------------------
// For layers with at most kSmallRoots roots the separate scan launch is dropped: every
// emit workgroup derives the compacted base of its first root from the search kernel's
// per-workgroup sums (a few hundred to a few thousand L2-resident words), scans its own
// <= 256 roots in LDS, and the workgroup owning the last slot publishes S and R' = R + S.
__global__ __launch_bounds__(kEmitThreads) void sample_emit_prefix_kernel(
    GraphView g, const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* d_R, uint64_t R_host, uint32_t fanout, int uniform, int prop_time,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rec_end,
    const uint32_t* __restrict__ rec_cnt, const uint32_t* __restrict__ wg_sum,
    uint32_t roots_per_search_wg, int64_t* __restrict__ all_nodes, float* __restrict__ all_ts,
    float* __restrict__ dt, int64_t* __restrict__ eids, int64_t* __restrict__ row,
    int64_t* __restrict__ col, uint64_t* out_R, uint64_t* out_S, uint64_t* next_R, Publish pub) {
  __shared__ uint32_t red[kEmitThreads / 64];
  __shared__ uint32_t lbase[kEmitThreads];
  __shared__ uint32_t wave_tot[kEmitThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t total = R * fanout;
  const uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * kEmitThreads;
  if (t0 >= total) return;   // uniform for the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t t_last = min(t0 + kEmitThreads - 1, total - 1);
  const uint32_t r_first = static_cast<uint32_t>(t0 / fanout);
  const uint32_t r_last = static_cast<uint32_t>(t_last / fanout);
  const uint32_t nroots = r_last - r_first + 1;   // <= kEmitThreads
  // 1. base of r_first = search-workgroup sums before it + the remainder inside its group
  const uint32_t b_first = r_first / roots_per_search_wg;
  uint32_t part = 0;
  for (uint32_t b = tid; b < b_first; b += kEmitThreads) part += wg_sum[b];
  for (uint32_t r = b_first * roots_per_search_wg + tid; r < r_first; r += kEmitThreads)
    part += valid_slots(rec_cnt[r], fanout, uniform);
  for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
  if (lane == 0) red[wave] = part;
  // 2. exclusive scan of this workgroup's own roots
  const uint32_t mine = tid < static_cast<int>(nroots)
                            ? valid_slots(rec_cnt[r_first + tid], fanout, uniform) : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < kEmitThreads / 64; ++w) base += red[w];
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
  lbase[tid] = base + wbase + incl - mine;
  __syncthreads();
  // 3. emit
  const uint64_t t = t0 + tid;
  if (t < total) {
    if (t < R) {
      all_nodes[t] = roots[t];
      all_ts[t] = root_ts[t];
    }
    const uint32_t r = static_cast<uint32_t>(t / fanout);
    const uint32_t j = static_cast<uint32_t>(t - static_cast<uint64_t>(r) * fanout);
    const uint32_t n = rec_cnt[r];
    if (j < valid_slots(n, fanout, uniform)) {
      const uint32_t pick = uniform ? gf_philox4x32_10_first(seed, t, call) % n : j;
      const uint64_t e = rec_end[r] - 1 - pick;
      const EdgePair nb = g.nbr_pool[e];
      const float ets = nb.ts;
      const float rts = root_ts[r];
      const uint64_t o = static_cast<uint64_t>(lbase[r - r_first]) + j;
      all_nodes[R + o] = nb.dst;
      all_ts[R + o] = prop_time ? rts : ets;
      dt[o] = rts - ets;
      eids[o] = nb.eid;
      row[o] = static_cast<int64_t>(r);
      col[o] = static_cast<int64_t>(R + o);
    }
  }
  // 4. the workgroup that owns the last slot knows the layer's edge count
  if (t_last == total - 1 && tid == static_cast<int>(nroots) - 1) {
    const uint64_t S = static_cast<uint64_t>(lbase[tid]) + mine;
    *out_R = R;
    *out_S = S;
    if (next_R) *next_R = R + S;
    // the LAST kernel of a sample also copies every block's sizes into pinned host memory (the
    // earlier blocks' are final: their kernels are complete; this block's were just written by
    // this thread); the host learns of the sample's completion from the stream's event
    for (uint32_t i = 0; i < pub.num_words; ++i) pub.h_counts[i] = pub.d_counts[i];
    if (pub.num_words) pub.h_counts[pub.num_words] = 0;
  }
}

// Decoupled look-back of the one-launch kernels: the sum of the counts that the `n_before`
// workgroups before this one published as granules {tag | count} (count in the bits of `mask`).
// Every thread polls up to kLookBatch granules PER ROUND TRIP — all loads of a batch are issued
// before the first is looked at (polled one after the other, a thread's 3-5 granules cost 3-5
// dependent agent-scope loads) — and a granule that has not
// shown the tag after kGranuleSpins rounds is recomputed by `recount(b)` (termination does not
// depend on dispatch order).  Returns this THREAD's partial sum; `recounts` counts fallbacks.
constexpr int kLookBatch = 8;
template <int kBlock, typename Recount>
__device__ inline uint32_t lookback_partial(const uint64_t* granules, uint32_t n_before,
                                            uint64_t tag, uint64_t mask, unsigned int* recounts,
                                            Recount recount) {
  uint32_t part = 0;
  for (uint32_t base = 0; base < n_before; base += kLookBatch * kBlock) {   // uniform trip count
    uint64_t gr[kLookBatch];
    bool need[kLookBatch];
    bool any = false;
#pragma unroll
    for (int k = 0; k < kLookBatch; ++k) {
      need[k] = base + k * kBlock + threadIdx.x < n_before;
      any |= need[k];
    }
    for (uint32_t spins = 0; any && spins < kGranuleSpins; ++spins) {
#pragma unroll
      for (int k = 0; k < kLookBatch; ++k)
        gr[k] = need[k] ? __hip_atomic_load(&granules[base + k * kBlock + threadIdx.x],
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : 0ull;
      any = false;
#pragma unroll
      for (int k = 0; k < kLookBatch; ++k) {
        if (!need[k]) continue;
        if ((gr[k] & ~mask) == tag) {
          part += static_cast<uint32_t>(gr[k] & mask);
          need[k] = false;
        } else {
          any = true;
        }
      }
      if (any) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < kLookBatch; ++k) {
      if (need[k]) {
        part += recount(base + k * kBlock + threadIdx.x);
        atomicAdd(recounts, 1u);
      }
    }
  }
  return part;
}

// ---- partitioned sampling: fixed-slot replies and their merge ---------
__device__ inline int64_t pack_f32_pair(float lo, float hi) {
  return static_cast<int64_t>(static_cast<uint64_t>(__float_as_uint(lo)) |
                              (static_cast<uint64_t>(__float_as_uint(hi)) << 32));
}

// Search + select in ONE launch: with `fanout` fixed slots per root there is no compaction,
// hence no prefix sum between the two.  Same window / candidate / selection rules as
// sample_search_kernel + sample_emit_kernel; the Philox counter is the slot index.
struct PaddedCommon {
  uint32_t snapshot_idx, num_snapshots;
  float window;
  uint32_t fanout;
  int uniform, prop_time;
  uint64_t seed;
  // reply slots of 12 B {dst, eid, edge time as u32 / u32 / f32 bits; dst 0xFFFFFFFF = empty}
  // instead of 24 B {dst, eid, (out time, dt)}: the shared chains of graphs whose node and edge
  // ids fit 32 bits (half the bytes on the wire; dt and the out time are recomputed from the
  // root's time by the merge)
  int narrow = 0;
};
struct PaddedJob {
  const int64_t* req;
  uint64_t n;
  uint64_t call;             // Philox call counter of this job
  int64_t* out;
  const uint64_t* d_own;
  const uint64_t* d_total;
  uint64_t total_host;
  const uint32_t* root_of;
  uint32_t* rec_cnt;
  uint32_t stride, world;
  uint32_t* d_overflow;
  // several samples sharing one exchange (sample_partitioned_group): an own share starts at row
  // own_skip (0: world * stride); the inbox holds `world` = P x m slots, slot v belongs to
  // sample v % m and raises THAT sample's word d_overflow_of[v % m] (m = 0: d_overflow)
  uint64_t own_skip = 0;
  uint32_t m = 0;
  uint32_t* d_overflow_of[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* row_cnt = nullptr;   // inbox job, compact replies: valid slots of every served row
};

// d_own != null: "this rank's own share" of a chained partitioned layer — the last *d_own
// of the layer's R request rows (R = *d_total, or total_host), counts still on the device.
// root_of / rec_cnt (own share only): the number of valid slots of every row goes straight
// to its root's counter, so the merge does not have to read the rows back to count them.
// stride != 0: the slotted layout (partition.hip).  Own share: it starts at row
// world * stride.  Otherwise `req` is the INBOX of an equal-split exchange — `world` slots of
// `stride` rows, row 0 of a slot its header {rows that follow, flags} — and only the rows
// a header announces are served (reply row = request row); a sender's overflow flag is
// folded into this rank's word, so every rank learns of it in the same exchange.
template <int GROUP>
__device__ inline void padded_job(const GraphView& g, const PaddedCommon& c, const PaddedJob& j) {
  const int64_t* __restrict__ req = j.req;
  int64_t* __restrict__ out = j.out;
  const uint32_t* __restrict__ root_of = j.root_of;
  uint64_t n = j.n;
  const uint32_t fanout = c.fanout, stride = j.stride;
  uint32_t* __restrict__ out32 = reinterpret_cast<uint32_t*>(j.out);
  if (j.d_own) {
    n = *j.d_own;
    const uint64_t skip = stride ? (j.own_skip ? j.own_skip : static_cast<uint64_t>(j.world) * stride)
                                 : (j.d_total ? *j.d_total : j.total_host) - n;
    req += 2 * skip;
    if (c.narrow) out32 += skip * fanout * 3;   // rows of fanout x 12 B
    else out += skip * fanout * 3;              // rows of fanout x 24 B
    if (root_of) root_of += skip;
  }
  constexpr int kGroupsPerBlock = kSearchThreads / GROUP;
  const int lane = threadIdx.x % GROUP;
  const int group_in_wave = (threadIdx.x % 64) / GROUP;
  const uint64_t group = static_cast<uint64_t>(blockIdx.x) * kGroupsPerBlock + threadIdx.x / GROUP;
  const uint64_t num_groups = static_cast<uint64_t>(gridDim.x) * kGroupsPerBlock;
  const bool inbox = stride && !j.d_own;
  for (uint64_t r = group; r < n; r += num_groups) {
    if (inbox) {
      const uint64_t q = r / stride, jj = r - q * stride;
      if (jj == 0) {
        if (lane == 0 && (req[2 * r + 1] & 1))
          atomicOr(j.m ? j.d_overflow_of[q % j.m] : j.d_overflow, 1u);
        continue;
      }
      const uint64_t rows = static_cast<uint64_t>(req[2 * q * stride]);
      if (jj - 1 >= min(rows, static_cast<uint64_t>(stride - 1))) continue;
    }
    const int64_t nid = req[2 * r];
    const float t = __uint_as_float(static_cast<uint32_t>(static_cast<uint64_t>(req[2 * r + 1])));
    float start, end;
    time_window(t, c.snapshot_idx, c.num_snapshots, c.window, &start, &end);
    uint64_t end_off = 0;
    uint32_t n_cand = 0;
    if (nid >= 0 && static_cast<uint64_t>(nid) < g.table_len) {
      const NodeEntry e = g.table[nid];
      if (e.size > 0) {
        uint32_t lo, hi;
        window_bounds<GROUP>(g, e, start, end, lane, group_in_wave, &lo, &hi);
        n_cand = hi > lo ? hi - lo : 0;
        end_off = e.start + hi;
      }
    }
    const uint32_t valid = valid_slots(n_cand, fanout, c.uniform);
    if (j.rec_cnt && lane == 0) j.rec_cnt[root_of[r]] = valid;
    if (j.row_cnt && lane == 0) j.row_cnt[r] = valid;
    for (uint32_t k = lane; k < fanout; k += GROUP) {
      const uint64_t slot = r * fanout + k;
      if (c.narrow) {
        uint32_t* o = out32 + slot * 3;
        if (k < valid) {
          const uint32_t pick = c.uniform ? gf_philox4x32_10_first(c.seed, slot, j.call) % n_cand : k;
          const EdgePair nb = g.nbr_pool[end_off - 1 - pick];
          o[0] = static_cast<uint32_t>(nb.dst);
          o[1] = static_cast<uint32_t>(nb.eid);
          o[2] = __float_as_uint(nb.ts);
        } else {
          o[0] = 0xFFFFFFFFu;
          o[1] = 0xFFFFFFFFu;
          o[2] = 0xFFFFFFFFu;
        }
        continue;
      }
      int64_t* o = out + slot * 3;
      if (k < valid) {
        const uint32_t pick = c.uniform ? gf_philox4x32_10_first(c.seed, slot, j.call) % n_cand : k;
        const uint64_t e = end_off - 1 - pick;
        const EdgePair nb = g.nbr_pool[e];
        const float ets = nb.ts;
        o[0] = nb.dst;
        o[1] = nb.eid;
        o[2] = pack_f32_pair(c.prop_time ? t : ets, t - ets);
      } else {
        o[0] = -1;
        o[1] = -1;
        o[2] = -1;
      }
    }
  }
}

template <int GROUP>
__global__ __launch_bounds__(kSearchThreads) void sample_padded_kernel(
    GraphView g, const int64_t* __restrict__ req, uint64_t n, uint32_t snapshot_idx,
    uint32_t num_snapshots, float window, uint32_t fanout, int uniform, int prop_time,
    uint64_t seed, uint64_t call, int64_t* __restrict__ out,
    const uint64_t* __restrict__ d_own, const uint64_t* __restrict__ d_total,
    uint64_t total_host, const uint32_t* __restrict__ root_of, uint32_t* __restrict__ rec_cnt,
    uint32_t stride, uint32_t world, uint32_t* __restrict__ d_overflow) {
  const PaddedCommon c{snapshot_idx, num_snapshots, window, fanout, uniform, prop_time, seed};
  padded_job<GROUP>(g, c, PaddedJob{req, n, call, out, d_own, d_total, total_host, root_of, rec_cnt,
                                    stride, world, d_overflow});
}

// Two jobs in one launch (blockIdx.y): the requests this rank received AND its own share —
// one launch and one kernel boundary less per layer when the exchange runs in the sampling
// stream (nothing to overlap the own share with).
template <int GROUP>
__global__ __launch_bounds__(kSearchThreads) void sample_padded_pair_kernel(
    GraphView g, PaddedCommon c, PaddedJob a, PaddedJob b) {
  if (blockIdx.y == 0) padded_job<GROUP>(g, c, a);
  else padded_job<GROUP>(g, c, b);
}
// ... and up to five: the shared inbox of m <= 4 samples and their own shares
struct PaddedJobs { PaddedJob j[5]; };
template <int GROUP>
__global__ __launch_bounds__(kSearchThreads) void sample_padded_group_kernel(
    GraphView g, PaddedCommon c, PaddedJobs jobs) {
  padded_job<GROUP>(g, c, jobs.j[blockIdx.y]);
}

// valid slots of root i's reply row (a prefix of the row for both policies)
__global__ void merge_count_kernel(const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos,
                                   const uint64_t* __restrict__ d_R, uint64_t R_host,
                                   uint32_t fanout, uint32_t* __restrict__ rec_cnt,
                                   uint32_t stride, uint32_t world) {
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const uint32_t p = pos[i];
  uint32_t c = 0;
  // slotted layout: a header row stands for a root that did not fit its owner's slot
  if (!(stride && p < world * stride && p % stride == 0)) {
    const int64_t* s = rep + static_cast<uint64_t>(p) * fanout * 3;
    for (uint32_t j = 0; j < fanout; ++j) c += s[3 * j] >= 0 ? 1u : 0u;
  }
  rec_cnt[i] = c;
}

__global__ __launch_bounds__(kEmitThreads) void merge_emit_kernel(
    const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host,
    uint32_t fanout, const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos,
    const uint32_t* __restrict__ rec_cnt, const uint32_t* __restrict__ base,
    int64_t* __restrict__ all_nodes, float* __restrict__ all_ts, float* __restrict__ dt,
    int64_t* __restrict__ eids, int64_t* __restrict__ row, int64_t* __restrict__ col) {
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t total = R * fanout;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    if (t < R) {
      all_nodes[t] = roots[t];
      all_ts[t] = root_ts[t];
    }
    const uint64_t r = t / fanout;
    const uint32_t j = static_cast<uint32_t>(t - r * fanout);
    if (j >= rec_cnt[r]) continue;
    const int64_t* s = rep + (static_cast<uint64_t>(pos[r]) * fanout + j) * 3;
    const uint64_t packed = static_cast<uint64_t>(s[2]);
    const uint64_t o = static_cast<uint64_t>(base[r]) + j;
    all_nodes[R + o] = s[0];
    all_ts[R + o] = __uint_as_float(static_cast<uint32_t>(packed));
    dt[o] = __uint_as_float(static_cast<uint32_t>(packed >> 32));
    eids[o] = s[1];
    row[o] = static_cast<int64_t>(r);
    col[o] = static_cast<int64_t>(R + o);
  }
}

// Small layers: count + per-workgroup sums in one launch, then an emit that derives its own
// prefix from them (as sample_emit_prefix_kernel does) — two launches instead of count / scan /
// emit.  The layer's root count may be device resident and may be 0 (a rank without roots).
__global__ __launch_bounds__(kEmitThreads) void merge_count_sums_kernel(
    const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t fanout,
    uint32_t* __restrict__ rec_cnt, uint32_t* __restrict__ wg_sum) {
  __shared__ uint32_t red[kEmitThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kEmitThreads + threadIdx.x;
  uint32_t c = 0;
  if (i < R) {
    const int64_t* s = rep + static_cast<uint64_t>(pos[i]) * fanout * 3;
    for (uint32_t j = 0; j < fanout; ++j) c += s[3 * j] >= 0 ? 1u : 0u;
    rec_cnt[i] = c;
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kEmitThreads / 64; ++w) t += red[w];
    wg_sum[blockIdx.x] = t;
  }
}

// Chained form: the own share's counts came from the sampling kernel itself; only the rows
// that arrived from other ranks — the first R - counts[rank] of the reply buffer — are read back
__global__ void merge_count_remote_kernel(const int64_t* __restrict__ rep,
                                          const uint32_t* __restrict__ root_of,
                                          const uint64_t* __restrict__ d_R, uint64_t R_host,
                                          const uint64_t* __restrict__ d_own, uint32_t fanout,
                                          uint32_t* __restrict__ rec_cnt) {
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t n_net = R - min(R, *d_own);
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n_net;
       row += stride) {
    const int64_t* s = rep + row * fanout * 3;
    uint32_t c = 0;
    for (uint32_t j = 0; j < fanout; ++j) c += s[3 * j] >= 0 ? 1u : 0u;
    rec_cnt[root_of[row]] = c;
  }
}

// Slotted layout: the rows that arrived from other ranks sit in `world` slots of `stride`
// rows (row 0 of a slot: the header row, never a reply); slot q holds min(counts[q], cap) rows.
// A root that did not fit its owner's slot (pos = the slot's header row) has no reply: its
// count is set to 0 here, so that the block's sizes stay within the layer's bounds while the
// overflowed sample runs to its end (it is then sampled again, dist.py).
__global__ void merge_count_slots_kernel(const int64_t* __restrict__ rep,
                                         const uint32_t* __restrict__ root_of,
                                         const uint32_t* __restrict__ pos,
                                         const uint64_t* __restrict__ d_R, uint64_t R_host,
                                         const uint64_t* __restrict__ counts, uint32_t stride,
                                         uint32_t world, uint32_t rank, uint32_t fanout,
                                         uint32_t* __restrict__ rec_cnt) {
  const uint64_t rows = static_cast<uint64_t>(world) * stride;
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const uint64_t first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (uint64_t row = first; row < rows; row += step) {
    const uint64_t q = row / stride, j = row - q * stride;
    if (j == 0 || q == rank || j - 1 >= min(counts[q], static_cast<uint64_t>(stride - 1))) continue;
    const int64_t* s = rep + row * fanout * 3;
    uint32_t c = 0;
    for (uint32_t k = 0; k < fanout; ++k) c += s[3 * k] >= 0 ? 1u : 0u;
    rec_cnt[root_of[row]] = c;
  }
  const uint64_t R = d_R ? *d_R : R_host;
  for (uint64_t i = first; i < R; i += step) {
    const uint32_t p = pos[i];
    if (p < rows && p % stride == 0) rec_cnt[i] = 0;
  }
}

__global__ __launch_bounds__(kEmitThreads) void merge_emit_prefix_kernel(
    const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t fanout,
    const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos,
    const uint32_t* __restrict__ rec_cnt, const uint32_t* __restrict__ wg_sum,
    int64_t* __restrict__ all_nodes, float* __restrict__ all_ts, float* __restrict__ dt,
    int64_t* __restrict__ eids, int64_t* __restrict__ row, int64_t* __restrict__ col,
    uint64_t* out_R, uint64_t* out_S, uint64_t* next_R) {
  __shared__ uint32_t red[kEmitThreads / 64];
  __shared__ uint32_t lbase[kEmitThreads];
  __shared__ uint32_t wave_tot[kEmitThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t total = R * fanout;
  if (total == 0) {   // nobody owns "the last slot": workgroup 0 reports the empty block
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *out_R = 0;
      *out_S = 0;
      if (next_R) *next_R = 0;
    }
    return;
  }
  const uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * kEmitThreads;
  if (t0 >= total) return;   // uniform for the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t t_last = min(t0 + kEmitThreads - 1, total - 1);
  const uint32_t r_first = static_cast<uint32_t>(t0 / fanout);
  const uint32_t r_last = static_cast<uint32_t>(t_last / fanout);
  const uint32_t nroots = r_last - r_first + 1;   // <= kEmitThreads
  const uint32_t b_first = r_first / kEmitThreads;   // count workgroups of kEmitThreads roots
  uint32_t part = 0;
  if (wg_sum) {
    for (uint32_t b = tid; b < b_first; b += kEmitThreads) part += wg_sum[b];
    for (uint32_t r = b_first * kEmitThreads + tid; r < r_first; r += kEmitThreads) part += rec_cnt[r];
  } else {
    // no per-workgroup sums: add up the counts of all the roots before this workgroup's
    // (coalesced, <= 128 KB out of L2: cheaper than the launch that would have summed them)
    for (uint32_t r = tid; r < r_first; r += kEmitThreads) part += rec_cnt[r];
  }
  for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
  if (lane == 0) red[wave] = part;
  const uint32_t mine = tid < static_cast<int>(nroots) ? rec_cnt[r_first + tid] : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < kEmitThreads / 64; ++w) base += red[w];
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
  lbase[tid] = base + wbase + incl - mine;
  __syncthreads();
  const uint64_t t = t0 + tid;
  if (t < total) {
    if (t < R) {
      all_nodes[t] = roots[t];
      all_ts[t] = root_ts[t];
    }
    const uint32_t r = static_cast<uint32_t>(t / fanout);
    const uint32_t j = static_cast<uint32_t>(t - static_cast<uint64_t>(r) * fanout);
    if (j < rec_cnt[r]) {
      const int64_t* s = rep + (static_cast<uint64_t>(pos[r]) * fanout + j) * 3;
      const uint64_t packed = static_cast<uint64_t>(s[2]);
      const uint64_t o = static_cast<uint64_t>(lbase[r - r_first]) + j;
      all_nodes[R + o] = s[0];
      all_ts[R + o] = __uint_as_float(static_cast<uint32_t>(packed));
      dt[o] = __uint_as_float(static_cast<uint32_t>(packed >> 32));
      eids[o] = s[1];
      row[o] = static_cast<int64_t>(r);
      col[o] = static_cast<int64_t>(R + o);
    }
  }
  if (t_last == total - 1 && tid == static_cast<int>(nroots) - 1) {
    const uint64_t S = static_cast<uint64_t>(lbase[tid]) + mine;
    *out_R = R;
    *out_S = S;
    if (next_R) *next_R = R + S;
  }
}

// Slotted layout, small layers: the whole merge in ONE launch.  The slots of the layer, in
// (root, slot) order, are compacted: thread t owns slot (r, j) = (t / fanout, t % fanout), which
// is valid iff root r has a reply row (pos[r] is not a slot's header row = the root fitted its
// owner's slot) and that row's slot j holds an edge; its place in the output is the number of
// valid slots before it.  The prefix over the workgroups' tiles travels through 8-byte granules
// {launch tag, tile count}: every workgroup publishes its tile's count with ONE relaxed
// agent-scope store before it looks at anybody else's, then adds up the granules of the tiles
// before its own (decoupled look-back; a granule is one naturally aligned sc1 store / sc1 load,
// so no fence is needed: /opt/skills/guides MI355X_MICROARCH "granule").  Tiles are dispatched in
// index order, so normally the lowest unfinished tile never waits for an undispatched one; a
// poll that does not see its granule within kGranuleSpins tries stops waiting and recounts that
// tile itself (see the look-back loop: termination does not depend on dispatch order).
// Replaces merge_count_slots_kernel + merge_emit_prefix_kernel: the chain of a sample is bound
// by the host thread that issues its launches, so one launch less per layer is ~3 us per sample.
__device__ unsigned int g_merge_recounts;          // tiles a look-back had to count itself
constexpr uint64_t kGranuleCountMask = 0x3FF;      // a tile has kEmitThreads = 256 slots
struct MergeJob {
  const int64_t* roots;
  const float* root_ts;
  const uint64_t* d_R;
  uint64_t R_host;
  const int64_t* rep;        // the (shared) reply buffer
  const uint32_t* pos;
  uint32_t slot_rows;        // rows of the buffer that belong to slots (P x m x stride)
  uint64_t* granules;
  uint64_t tag;
  uint32_t* d_overflow;
  int64_t* all_nodes; float* all_ts; float* dt; int64_t* eids; int64_t* row; int64_t* col;
  uint64_t* out_R; uint64_t* out_S; uint64_t* next_R;
  // compact replies (null: the slots' rows are fixed-fanout rows of `rep` like the own share's):
  // the received slots, cslot bytes each — u32 [0] edges of the slot, [r] edges before row r,
  // [stride] the sender's overflow word, then the edges, edge_cap at most
  const char* crep = nullptr;
  uint32_t cslot = 0, edge_cap = 0, m = 1, jidx = 0, off_bytes = 4;
  // reuse of the previous layer (roots whose pos[] is kPosReused): root r < R_prev of this
  // layer IS root r of the previous one, same timestamp, and its edges are entries
  // [first_prev[r], first_prev[r + 1]) of the previous block; first_out[r] = this block's first
  // edge of root r (R + 1 entries), for the next layer
  const uint32_t* first_prev = nullptr;
  const uint64_t* d_R_prev = nullptr;
  uint64_t R_prev_host = 0;
  const int64_t* nodes_prev = nullptr; const float* ts_prev = nullptr;
  const float* dt_prev = nullptr; const int64_t* eids_prev = nullptr;
  uint32_t* first_out = nullptr;
};
struct MergeReuse {
  const uint32_t* first_prev; uint64_t R_prev;
  const int64_t* nodes_prev; const float* ts_prev; const float* dt_prev; const int64_t* eids_prev;
  uint32_t* first_out;
};

__device__ inline void merge_slots_fused_body(
    const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t fanout,
    const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos, uint32_t stride,
    uint32_t slot_rows, uint64_t* granules, uint64_t tag, uint32_t* d_overflow,
    int64_t* __restrict__ all_nodes, float* __restrict__ all_ts, float* __restrict__ dt,
    int64_t* __restrict__ eids, int64_t* __restrict__ row, int64_t* __restrict__ col,
    uint64_t* out_R, uint64_t* out_S, uint64_t* next_R, int narrow = 0,
    const char* __restrict__ crep = nullptr, uint32_t cslot = 0, uint32_t edge_cap = 0,
    uint32_t gm = 1, uint32_t gj = 0, uint32_t off_bytes = 4,
    MergeReuse reuse = MergeReuse{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr}) {
  // narrow: 0 = 24 B reply slots; 1 = 12 B slots {dst, eid, edge time}, the out time is the
  // edge's; 2 = 12 B slots, the out time is the root's (prop_time)
  const uint32_t* __restrict__ rep32 = reinterpret_cast<const uint32_t*>(rep);
  const uint32_t cedges = (off_bytes * (stride + 1) + 15) & ~15u;   // a compact slot's edges
  auto offset_at = [&](const char* base, uint32_t i) -> uint32_t {
    return off_bytes == 2 ? reinterpret_cast<const uint16_t*>(base)[i]
                          : reinterpret_cast<const uint32_t*>(base)[i];
  };
  // where slot j of the reply row p is: null = no such edge.  Rows of the peers' slots come
  // in the compact form when `crep` is set, everything else as fixed-fanout rows of `rep`.
  auto record = [&](uint32_t p, uint32_t j) -> const void* {
    if (crep && p < slot_rows) {
      const uint32_t sl = p / stride, rw = p - sl * stride;
      const char* base = crep + static_cast<uint64_t>(sl) * cslot;
      const uint32_t lo = min(offset_at(base, rw), edge_cap);
      const uint32_t hi = min(offset_at(base, rw + 1 < stride ? rw + 1 : 0), edge_cap);
      if (j >= hi - lo) return nullptr;
      return base + cedges + static_cast<uint64_t>(lo + j) * (narrow ? 12 : 24);
    }
    if (narrow) {
      const uint32_t* q = rep32 + (static_cast<uint64_t>(p) * fanout + j) * 3;
      return q[0] != 0xFFFFFFFFu ? q : nullptr;
    }
    const int64_t* q = rep + (static_cast<uint64_t>(p) * fanout + j) * 3;
    return q[0] >= 0 ? q : nullptr;
  };
  // a sender whose compact slot overflowed says so in every slot it sends: all ranks redo
  if (crep && blockIdx.x == 0 && threadIdx.x * gm + gj < slot_rows / stride) {
    const char* base = crep + static_cast<uint64_t>(threadIdx.x * gm + gj) * cslot;
    if (offset_at(base, stride)) atomicOr(d_overflow, 1u);
  }
  __shared__ uint32_t wave_cnt[kEmitThreads / 64];
  __shared__ uint32_t red[kEmitThreads / 64];
  const uint64_t R = d_R ? *d_R : R_host;
  const uint64_t total = R * fanout;
  if (total == 0) {   // nobody owns "the last slot": workgroup 0 reports the empty block
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *out_R = 0;
      *out_S = 0;
      if (next_R) *next_R = 0;
      if (reuse.first_out) reuse.first_out[0] = 0;
    }
    return;
  }
  const uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * kEmitThreads;
  if (t0 >= total) return;   // uniform for the workgroup; no tile behind it exists either
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t t = t0 + tid;
  bool valid = false;
  uint32_t r = 0;
  int64_t s0 = 0, s1 = 0;
  uint64_t packed = 0;
  if (t < total) {
    r = static_cast<uint32_t>(t / fanout);
    const uint32_t j = static_cast<uint32_t>(t - static_cast<uint64_t>(r) * fanout);
    const uint32_t p = pos[r];
    if (p == kPosReused) {
      // the previous block holds this root's edges (same root, same time, same fanout)
      const uint32_t lo = reuse.first_prev[r], hi = reuse.first_prev[r + 1];
      valid = j < hi - lo;
      if (valid) {
        const uint32_t e = lo + j;
        s0 = reuse.nodes_prev[reuse.R_prev + e];
        s1 = reuse.eids_prev[e];
        packed = static_cast<uint64_t>(
            pack_f32_pair(reuse.ts_prev[reuse.R_prev + e], reuse.dt_prev[e]));
      }
    } else if (!(p < slot_rows && p % stride == 0)) {
      const void* rec = record(p, j);
      valid = rec != nullptr;
      if (valid && narrow) {
        const uint32_t* s = static_cast<const uint32_t*>(rec);
        s0 = static_cast<int64_t>(s[0]);
        s1 = static_cast<int64_t>(s[1]);
        const float t = root_ts[r], ets = __uint_as_float(s[2]);
        packed = static_cast<uint64_t>(pack_f32_pair(narrow == 2 ? t : ets, t - ets));
      } else if (valid) {
        const int64_t* s = static_cast<const int64_t*>(rec);
        s0 = s[0];
        s1 = s[1];
        packed = static_cast<uint64_t>(s[2]);
      }
    }
  }
  const uint64_t ballot = __ballot(valid);
  const uint32_t before = static_cast<uint32_t>(__popcll(ballot & ((1ull << lane) - 1ull)));
  if (lane == 0) wave_cnt[wave] = static_cast<uint32_t>(__popcll(ballot));
  __syncthreads();
  uint32_t tile_cnt = 0, wbase = 0;
#pragma unroll
  for (int w = 0; w < kEmitThreads / 64; ++w) {
    if (w < wave) wbase += wave_cnt[w];
    tile_cnt += wave_cnt[w];
  }
  if (tid == 0)
    __hip_atomic_store(&granules[blockIdx.x], tag | tile_cnt, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // look-back: the tiles before this one.  A granule that has not arrived after kGranuleSpins
  // polls is NOT waited for any longer: the thread counts that tile's valid slots itself (256
  // slots, two loads each: slow, but it depends on nobody).  Termination therefore does not
  // rest on the order in which workgroups are dispatched — with the GPU oversubscribed (several
  // such kernels of different streams or processes in flight, workgroups dealt to the XCDs
  // independently) a tile could otherwise wait for one that cannot be dispatched because its
  // XCD is full of waiters: observed with 4 rank processes sharing one GPU.
  uint32_t part = lookback_partial<kEmitThreads>(
      granules, blockIdx.x, tag, kGranuleCountMask, &g_merge_recounts, [&](uint32_t b) {
        uint32_t cnt = 0;
        const uint64_t lo = static_cast<uint64_t>(b) * kEmitThreads;
        const uint64_t hi = min(lo + kEmitThreads, total);
        for (uint64_t u = lo; u < hi; ++u) {
          const uint32_t ru = static_cast<uint32_t>(u / fanout);
          const uint32_t ju = static_cast<uint32_t>(u - static_cast<uint64_t>(ru) * fanout);
          const uint32_t pu = pos[ru];
          if (pu == kPosReused)
            cnt += ju < reuse.first_prev[ru + 1] - reuse.first_prev[ru] ? 1u : 0u;
          else if (!(pu < slot_rows && pu % stride == 0)) cnt += record(pu, ju) != nullptr ? 1u : 0u;
        }
        return cnt;
      });
  for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < kEmitThreads / 64; ++w) base += red[w];
  if (t < total) {
    if (t < R) {
      all_nodes[t] = roots[t];
      all_ts[t] = root_ts[t];
    }
    if (reuse.first_out && t % fanout == 0)   // slot 0 of root r: the edges before root r
      reuse.first_out[r] = base + wbase + before;
    if (valid) {
      const uint64_t o = static_cast<uint64_t>(base) + wbase + before;
      all_nodes[R + o] = s0;
      all_ts[R + o] = __uint_as_float(static_cast<uint32_t>(packed));
      dt[o] = __uint_as_float(static_cast<uint32_t>(packed >> 32));
      eids[o] = s1;
      row[o] = static_cast<int64_t>(r);
      col[o] = static_cast<int64_t>(R + o);
    }
  }
  if (t0 + kEmitThreads >= total && tid == 0) {   // the tile with the last slot
    const uint64_t S = static_cast<uint64_t>(base) + tile_cnt;
    *out_R = R;
    *out_S = S;
    if (next_R) *next_R = R + S;
    if (reuse.first_out) reuse.first_out[R] = static_cast<uint32_t>(S);
  }
}

__global__ __launch_bounds__(kEmitThreads) void merge_slots_fused_kernel(
    const int64_t* __restrict__ roots, const float* __restrict__ root_ts,
    const uint64_t* __restrict__ d_R, uint64_t R_host, uint32_t fanout,
    const int64_t* __restrict__ rep, const uint32_t* __restrict__ pos, uint32_t stride,
    uint32_t world, uint64_t* granules, uint64_t tag, uint32_t* d_overflow,
    int64_t* __restrict__ all_nodes, float* __restrict__ all_ts, float* __restrict__ dt,
    int64_t* __restrict__ eids, int64_t* __restrict__ row, int64_t* __restrict__ col,
    uint64_t* out_R, uint64_t* out_S, uint64_t* next_R) {
  merge_slots_fused_body(roots, root_ts, d_R, R_host, fanout, rep, pos, stride, world * stride,
                         granules, tag, d_overflow, all_nodes, all_ts, dt, eids, row, col, out_R,
                         out_S, next_R);
}

// m <= 4 samples that shared their exchange (blockIdx.y picks the job; each has its own granules)
struct MergeJobs { MergeJob j[4]; };
__global__ __launch_bounds__(kEmitThreads) void merge_slots_fused_group_kernel(
    MergeJobs jobs, uint32_t fanout, uint32_t stride, int narrow) {
  const MergeJob& j = jobs.j[blockIdx.y];
  merge_slots_fused_body(j.roots, j.root_ts, j.d_R, j.R_host, fanout, j.rep, j.pos, stride,
                         j.slot_rows, j.granules, j.tag, j.d_overflow, j.all_nodes, j.all_ts, j.dt,
                         j.eids, j.row, j.col, j.out_R, j.out_S, j.next_R, narrow, j.crep, j.cslot,
                         j.edge_cap, j.m, j.jidx, j.off_bytes,
                         MergeReuse{j.first_prev, j.d_R_prev ? *j.d_R_prev : j.R_prev_host,
                                    j.nodes_prev, j.ts_prev, j.dt_prev, j.eids_prev, j.first_out});
}

// Compact replies of a shared chain: one workgroup per received request slot turns the slot's
// served rows (fixed `fanout` records each, of which a few hold an edge) into what travels back:
// u32 [0] = edges of the slot, [r] = edges of the rows before row r (1 <= r < stride),
// [stride] = "a slot of this sender overflowed its edge capacity" (written for ALL slots of the
// sample by whichever workgroup finishes last: one atomic carries the ticket and the flag), then
// the edges packed in row order.  Rows the request header does not announce hold nothing.
// (The reference ships back exactly the sampled edges of a partition,
// gnnflow/distributed/common.py:4-19, dist_sampler.py:244-314.)
struct CompactArgs {
  const int64_t* inbox;
  const void* served;
  const uint32_t* row_cnt;
  char* cserved;
  // [m] {launch tag, overflow count << 16 | slots done}: a word that carries another launch's tag
  // (a chain abandoned half-way, whatever the reason) starts over — nothing relies on a reset
  unsigned long long* ticket;
  uint32_t stride, fanout, m, world, edge_cap, cslot, narrow, off_bytes, tag;
};
constexpr int kCompactThreads = 1024;
__global__ __launch_bounds__(kCompactThreads) void reply_compact_kernel(CompactArgs a) {
  const uint32_t sl = blockIdx.x, tid = threadIdx.x, stride = a.stride, F = a.fanout;
  const uint64_t announced = static_cast<uint64_t>(a.inbox[2 * static_cast<uint64_t>(sl) * stride]);
  const uint32_t rows = static_cast<uint32_t>(min(announced, static_cast<uint64_t>(stride - 1)));
  char* base = a.cserved + static_cast<uint64_t>(sl) * a.cslot;
  auto put = [&](char* slot, uint32_t i, uint32_t v) {
    if (a.off_bytes == 2) reinterpret_cast<uint16_t*>(slot)[i] = static_cast<uint16_t>(min(v, 65535u));
    else reinterpret_cast<uint32_t*>(slot)[i] = v;
  };
  char* edges = base + ((a.off_bytes * (stride + 1) + 15) & ~15u);
  const uint32_t rb = a.narrow ? 12u : 24u;
  __shared__ uint32_t wsum[kCompactThreads / 64];
  __shared__ uint32_t carry;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t b0 = 1; b0 < stride; b0 += kCompactThreads) {
    const uint32_t r = b0 + tid;                         // row of the slot (row 0 is its header)
    const uint64_t grow = static_cast<uint64_t>(sl) * stride + r;
    const uint32_t cnt = (r < stride && r - 1 < rows) ? a.row_cnt[grow] : 0u;
    uint32_t incl = cnt;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = carry, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; ++w) {
      const uint32_t x = wsum[w];
      if (w < wave) before += x;
      total += x;
    }
    const uint32_t at = before + incl - cnt;
    if (r < stride) put(base, r, at);
    for (uint32_t k = 0; k < cnt; ++k) {
      if (at + k >= a.edge_cap) break;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(
          static_cast<const char*>(a.served) + (grow * F + k) * rb);
      uint32_t* dst = reinterpret_cast<uint32_t*>(edges + static_cast<uint64_t>(at + k) * rb);
      for (uint32_t w = 0; w < rb / 4; ++w) dst[w] = src[w];
    }
    __syncthreads();
    if (tid == 0) carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t total = carry;
    put(base, 0, total);
    const uint32_t j = sl % a.m, ovf = total > a.edge_cap ? 1u : 0u;
    const unsigned long long fresh = static_cast<unsigned long long>(a.tag) << 32;
    unsigned long long old = __hip_atomic_load(&a.ticket[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prev;
    for (;;) {
      const unsigned long long cur = static_cast<uint32_t>(old >> 32) == a.tag ? old : fresh;
      const unsigned long long seen = atomicCAS(&a.ticket[j], old, cur + 1u + (ovf << 16));
      if (seen == old) { prev = static_cast<uint32_t>(cur); break; }
      old = seen;
    }
    if ((prev & 0xFFFFu) == a.world - 1) {       // the last of this sample's `world` slots
      const uint32_t any = ((prev >> 16) + ovf) ? 1u : 0u;
      for (uint32_t q = 0; q < a.world; ++q)
        put(a.cserved + static_cast<uint64_t>(q * a.m + j) * a.cslot, stride, any);
    }
  }
}

inline unsigned capped_grid(uint64_t work_items, unsigned per_block, unsigned cap) {
  uint64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return static_cast<unsigned>(std::min<uint64_t>(g, cap));
}

// Lanes cooperating on one root in the search kernels.  A small layer (<= 32 768 roots,
// every root in flight at once) is a pure latency chain, so it wants FEW rounds: 16 lanes,
// log16(deg) + 1 dependent reads.  A large layer is bound by how many roots the resident
// waves keep in flight, so it wants NARROW groups: 4 lanes put 16 roots in flight per wave
// and issue 40 probes per 10^7-edge segment instead of 96 (measured on the 10 M-node /
// 200 M-edge graph, profiles/).  Read when a sampler is created so tests can compare widths.
int group_width_from_env(const char* name, int fallback) {
  const char* v = std::getenv(name);
  const int g = v ? std::atoi(v) : fallback;
  return (g == 2 || g == 4 || g == 8 || g == 16) ? g : fallback;
}

template <typename... Args>
void launch_search(int width, unsigned grid, hipStream_t stream, Args... args) {
  switch (width) {
    case 2: sample_search_kernel<2><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    case 4: sample_search_kernel<4><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    case 8: sample_search_kernel<8><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    default: sample_search_kernel<16><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
  }
}

template <typename... Args>
void launch_padded(int width, unsigned grid, hipStream_t stream, Args... args) {
  switch (width) {
    case 2: sample_padded_kernel<2><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    case 4: sample_padded_kernel<4><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    case 8: sample_padded_kernel<8><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
    default: sample_padded_kernel<16><<<dim3(grid), dim3(kSearchThreads), 0, stream>>>(args...); break;
  }
}

void launch_padded_pair(int width, unsigned grid, hipStream_t stream, const GraphView& g,
                        const PaddedCommon& c, const PaddedJob& a, const PaddedJob& b) {
  const dim3 gr(grid, 2), bl(kSearchThreads);
  switch (width) {
    case 2: sample_padded_pair_kernel<2><<<gr, bl, 0, stream>>>(g, c, a, b); break;
    case 4: sample_padded_pair_kernel<4><<<gr, bl, 0, stream>>>(g, c, a, b); break;
    case 8: sample_padded_pair_kernel<8><<<gr, bl, 0, stream>>>(g, c, a, b); break;
    default: sample_padded_pair_kernel<16><<<gr, bl, 0, stream>>>(g, c, a, b); break;
  }
}

void launch_padded_group(int width, unsigned grid, int jobs_n, hipStream_t stream,
                         const GraphView& g, const PaddedCommon& c, const PaddedJobs& jobs) {
  const dim3 gr(grid, static_cast<unsigned>(jobs_n)), bl(kSearchThreads);
  switch (width) {
    case 2: sample_padded_group_kernel<2><<<gr, bl, 0, stream>>>(g, c, jobs); break;
    case 4: sample_padded_group_kernel<4><<<gr, bl, 0, stream>>>(g, c, jobs); break;
    case 8: sample_padded_group_kernel<8><<<gr, bl, 0, stream>>>(g, c, jobs); break;
    default: sample_padded_group_kernel<16><<<gr, bl, 0, stream>>>(g, c, jobs); break;
  }
}

}  // namespace

// The fences pay where a layer is bound by memory traffic (large layers: one line per round
// instead of GROUP sectors; config 3, batch 300 k: search 538 -> 516 us).  A small layer is a
// pure latency chain with the same number of rounds either way, and the fenced search's extra
// address arithmetic made it slower (REDDIT-shaped batch 600: 6.0 -> 7.7 us per launch): small
// layers search the timestamps directly.
inline GraphView view_for(const EdgeStore* g, size_t roots) {
  GraphView v = g->view();
  if (roots <= kSmallRoots) v.fence.levels = 0;
  return v;
}

// ---- host driver -------------------------------------------------------------------
Sampler::Sampler(EdgeStore* graph, const uint32_t* fanouts, size_t num_layers, int policy,
                 uint32_t num_snapshots, float window, bool prop_time, uint64_t seed)
    : graph_(graph),
      fanouts_(fanouts, fanouts + num_layers),
      policy_(policy),
      num_snapshots_(num_snapshots),
      window_(window),
      prop_time_(prop_time),
      seed_(seed) {
  GF_REQUIRE(graph != nullptr, "sampler: null graph");
  GF_REQUIRE(num_layers > 0, "sampler: fanouts must not be empty");
  for (uint32_t f : fanouts_) GF_REQUIRE(f > 0, "sampler: fanout must be positive");
  GF_REQUIRE(policy == GF_SAMPLING_POLICY_RECENT || policy == GF_SAMPLING_POLICY_UNIFORM,
             "sampler: invalid sampling policy");
  GF_REQUIRE(num_snapshots >= 1, "sampler: num_snapshots must be >= 1");
  search_group_ = group_width_from_env("GNNFLOW_SEARCH_GROUP", 16);
  large_group_ = group_width_from_env("GNNFLOW_SEARCH_GROUP_LARGE", 4);
  {
    const char* v = std::getenv("GNNFLOW_SAMPLER_FUSED_SCAN");
    fused_scan_ = !(v && std::atoi(v) == 0);
    v = std::getenv("GNNFLOW_SAMPLER_HYBRID_SEARCH");   // tests / A-B runs
    hybrid_search_ = !(v && std::atoi(v) == 0);
  }
  DeviceGuard dg(graph_->device());
  for (InFlight& f : ring_) GF_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
  rec_words_ = 2 + 2 * num_layers * num_snapshots;   // flag, {R, S} per block, overflow
  h_counts_.reserve(kMaxInFlight * rec_words_ * sizeof(uint64_t));
  std::memset(h_counts_.data(), 0, kMaxInFlight * rec_words_ * sizeof(uint64_t));
  h_layer_counts_.reserve(2 * sizeof(uint64_t));
  // (own_stream_ — the stream of the host-array entry points — is created on first use: every
  // stream a process creates shifts the hardware-queue placement of the ones created after it,
  // and a pipelined loop clones its sampler per lane)
}

Sampler::~Sampler() {
  for (InFlight& f : ring_) {
    if (f.done) {
      (void)hipEventSynchronize(f.done);
      (void)hipEventDestroy(f.done);
    }
  }
  if (own_stream_) {
    (void)hipStreamSynchronize(own_stream_);
    (void)hipStreamDestroy(own_stream_);
  }
}

size_t Sampler::root_bound(size_t R, size_t layer) const {
  size_t b = R;
  for (size_t l = 0; l < layer; ++l) b += b * fanouts_[l];
  return b;
}

// Per-block carve of the output buffer; every array 16-byte aligned.
size_t Sampler::layer_output_bytes(size_t Rb, size_t layer) const {
  const size_t F = fanouts_[layer], Sb = Rb * F;
  return align_up((Rb + Sb) * 8, 16) + align_up((Rb + Sb) * 4, 16) + align_up(Sb * 4, 16) +
         3 * align_up(Sb * 8, 16);
}

Sampler::BlockPtrs Sampler::carve(char* p, size_t Rb, uint32_t F) const {
  const size_t Sb = Rb * F;
  BlockPtrs b;
  b.all_nodes = reinterpret_cast<int64_t*>(p); p += align_up((Rb + Sb) * 8, 16);
  b.eids = reinterpret_cast<int64_t*>(p);      p += align_up(Sb * 8, 16);
  b.row = reinterpret_cast<int64_t*>(p);       p += align_up(Sb * 8, 16);
  b.col = reinterpret_cast<int64_t*>(p);       p += align_up(Sb * 8, 16);
  b.all_ts = reinterpret_cast<float*>(p);      p += align_up((Rb + Sb) * 4, 16);
  b.dt = reinterpret_cast<float*>(p);
  return b;
}

size_t Sampler::output_bytes(size_t R) const {
  size_t total = 0;
  for (size_t l = 0; l < fanouts_.size(); ++l)
    total += num_snapshots_ * layer_output_bytes(root_bound(R, l), l);
  return total;
}

void Sampler::reserve_workspace(size_t Rb, size_t num_blocks, hipStream_t stream) {
  if (Rb <= ws_roots_ && num_blocks <= ws_blocks_) return;
  ws_roots_ = std::max(ws_roots_, Rb);
  ws_blocks_ = std::max(ws_blocks_, num_blocks);
  size_t bytes = align_up(ws_roots_ * 8, 16) + 3 * align_up(ws_roots_ * 4, 16) +
                 ws_blocks_ * 2 * sizeof(uint64_t) + 64;
  // kernels of earlier samples may still be queued on `stream` with the old workspace: it
  // is retired behind an event and freed later (stream-ordered swap, no stall)
  retired_.collect();
  DeviceBuffer fresh;
  fresh.reserve(bytes, 0, stream);
  // the first array doubles as the fused merge's granules: no stale tag in fresh memory
  GF_HIP(hipMemsetAsync(fresh.data(), 0, align_up(ws_roots_ * 8, 16), stream));
  std::swap(ws_, fresh);
  retired_.retire(std::move(fresh), stream);
}

void Sampler::enqueue_layer(const int64_t* d_roots, const float* d_ts, size_t Rb,
                            const uint64_t* d_R, uint64_t R_host, uint32_t layer,
                            uint32_t snapshot, const BlockPtrs& out, uint64_t* d_counts_slot,
                            uint64_t* next_R, hipStream_t stream, const void* publish) {
  Publish pub{};
  if (publish) pub = *static_cast<const Publish*>(publish);
  const uint32_t F = fanouts_[layer];
  const int uniform = policy_ == GF_SAMPLING_POLICY_UNIFORM;
  GF_REQUIRE(static_cast<uint64_t>(Rb) * F < 0xFFFFFFFFull,
             "sampler: more than 2^32-1 slots in one layer");
  char* w = ws_.as<char>();
  uint64_t* rec_end = reinterpret_cast<uint64_t*>(w); w += align_up(ws_roots_ * 8, 16);
  uint32_t* rec_cnt = reinterpret_cast<uint32_t*>(w); w += align_up(ws_roots_ * 4, 16);
  uint32_t* base = reinterpret_cast<uint32_t*>(w);    w += align_up(ws_roots_ * 4, 16);
  uint32_t* wg_sum = reinterpret_cast<uint32_t*>(w);
  const GraphView gv = view_for(graph_, Rb);
  const uint64_t call = calls_++;
  // small layers: search publishes per-workgroup sums and emit does its own prefix.  (Search +
  // prefix + emit in ONE launch through look-back granules was built and measured: 17 us per
  // layer against 6.5 + 6 us + a 1.5 us boundary — across XCDs a count reaches its readers
  // through memory, which a kernel boundary does for free; profiles/README, round 5.)
  const bool small = fused_scan_ && Rb <= kSmallRoots;
  const unsigned roots_per_wg = kSearchThreads / search_group_;
  {
    ProfileScope ps(kProfSearch, stream);
    const unsigned grid = small ? static_cast<unsigned>((Rb + roots_per_wg - 1) / roots_per_wg)
                                : capped_grid(Rb, roots_per_wg, 256 * 8);
    if (hybrid_search_ && Rb >= kLaneSearchRoots) {
      // worklist: segment w (of the lane pass's workgroup w) can hold every root that
      // workgroup looks at; seg_count lives in the tile scratch (wg_sum), unused until the scan.
      const unsigned lgrid = capped_grid(Rb, kSearchThreads, kMaxHubSegs);
      const uint64_t chunks_per_wg = ((Rb + 63) / 64 + (lgrid * 4ull) - 1) / (lgrid * 4ull);
      const uint32_t seg_cap = static_cast<uint32_t>(chunks_per_wg * 4 * 64);
      const size_t hub_bytes = static_cast<size_t>(seg_cap) * lgrid * sizeof(uint32_t);
      if (hub_bytes > hub_buf_.bytes()) {   // stream-ordered swap, as for the workspace
        DeviceBuffer fresh;
        fresh.reserve(hub_bytes, 0, stream);
        std::swap(hub_buf_, fresh);
        retired_.retire(std::move(fresh), stream);
      }
      uint32_t* hub_list_ = hub_buf_.as<uint32_t>();
      uint32_t* seg_count = wg_sum;
      sample_search_lanes_kernel<<<dim3(lgrid), dim3(kSearchThreads), 0, stream>>>(
          gv, d_roots, d_ts, d_R, R_host, snapshot, num_snapshots_, window_, rec_end, rec_cnt,
          hub_list_, seg_count, seg_cap);
      launch_search(large_group_, capped_grid(Rb / 4, kSearchThreads / large_group_, 256 * 8),
                    stream, gv, d_roots, d_ts, nullptr, 0, snapshot, num_snapshots_, window_,
                    rec_end, rec_cnt, F, uniform, nullptr, hub_list_, seg_count, lgrid, seg_cap);
    } else if (small) {
      launch_search(search_group_, grid, stream, gv, d_roots, d_ts, d_R, R_host, snapshot,
                    num_snapshots_, window_, rec_end, rec_cnt, F, uniform, wg_sum, nullptr,
                    nullptr, 0, 0);
    } else {
      launch_search(large_group_, capped_grid(Rb, kSearchThreads / large_group_, 256 * 8), stream,
                    gv, d_roots, d_ts, d_R, R_host, snapshot, num_snapshots_, window_, rec_end,
                    rec_cnt, F, uniform, nullptr, nullptr, nullptr, 0, 0);
    }
    GF_HIP(hipGetLastError());
  }
  if (small) {
    ProfileScope ps(kProfEmit, stream);
    const unsigned grid = static_cast<unsigned>(
        (static_cast<uint64_t>(Rb) * F + kEmitThreads - 1) / kEmitThreads);
    sample_emit_prefix_kernel<<<dim3(grid), dim3(kEmitThreads), 0, stream>>>(
        gv, d_roots, d_ts, d_R, R_host, F, uniform, prop_time_ ? 1 : 0, seed_, call, rec_end,
        rec_cnt, wg_sum, roots_per_wg, out.all_nodes, out.all_ts, out.dt, out.eids, out.row,
        out.col, d_counts_slot, d_counts_slot + 1, next_R, pub);
    GF_HIP(hipGetLastError());
    return;
  }
  {
    ProfileScope ps(kProfScan, stream);
    if (Rb <= 65536) {
      sample_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
          rec_cnt, base, d_R, R_host, F, uniform, d_counts_slot, d_counts_slot + 1, next_R);
    } else {
      // wg_sum doubles as the tile-sum / tile-base scratch (2 * tiles <= ws_roots_ words)
      const size_t tiles = (Rb + kScanTile - 1) / kScanTile;
      uint32_t* tile_sum = wg_sum;
      uint32_t* tile_base = wg_sum + tiles;
      const unsigned grid = static_cast<unsigned>(std::min<size_t>(tiles, 2048));
      sample_tile_sum_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(
          rec_cnt, d_R, R_host, F, uniform, tile_sum);
      sample_tile_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
          tile_sum, tile_base, d_R, R_host, d_counts_slot, d_counts_slot + 1, next_R);
      sample_tile_apply_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(
          rec_cnt, tile_base, d_R, R_host, F, uniform, base);
    }
    GF_HIP(hipGetLastError());
  }
  {
    ProfileScope ps(kProfEmit, stream);
    unsigned grid = capped_grid(static_cast<uint64_t>(Rb) * F, kEmitThreads, 256 * 16);
    static const int unroll = [] {
      const char* v = std::getenv("GNNFLOW_EMIT_UNROLL");   // A/B runs: 1 = one slot per trip
      return (v && std::atoi(v) == 1) ? 0 : 1;
    }();
    sample_emit_kernel<<<dim3(grid), dim3(kEmitThreads), 0, stream>>>(
        gv, d_roots, d_ts, d_R, R_host, F, uniform, prop_time_ ? 1 : 0, seed_, call, rec_end,
        rec_cnt, base, out.all_nodes, out.all_ts, out.dt, out.eids, out.row, out.col, pub,
        unroll);
    GF_HIP(hipGetLastError());
  }
}

// TemporalSampler::Sample, temporal_sampler.cu:279-305 — split in two so a caller can
// overlap the sampling of batch i+1 (on its own stream) with other work on batch i:
// begin() enqueues every kernel plus the size read-back and returns; end() waits on the
// completion event and reports the block sizes.
void Sampler::sample_begin(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                           size_t out_bytes, hipStream_t stream) {
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  InFlight* slot;
  {
    std::lock_guard<std::mutex> lk(ring_mu_);
    GF_REQUIRE(ring_count_ < kMaxInFlight, "sample_begin: too many samples in flight on this sampler");
    if (ring_count_ > 0) {
      const InFlight& newest = ring_[(ring_head_ + ring_count_ - 1) % kMaxInFlight];
      GF_REQUIRE(newest.roots == 0 || R == 0 || newest.stream == stream,
                 "sample_begin: samples in flight on one sampler must share a stream");
    }
    slot = &ring_[(ring_head_ + ring_count_) % kMaxInFlight];
  }
  // (only this thread begins samples: the slot stays free until ring_count_ is bumped below)
  slot->ptrs.assign(L * NS, BlockPtrs{});
  slot->roots = R;
  slot->by_event = false;
  slot->stream = stream;
  if (R == 0) {  // temporal_sampler.cu:107-114
    calls_ += L * NS;
    std::lock_guard<std::mutex> lk(ring_mu_);
    ++ring_count_;
    return;
  }
  GF_REQUIRE(d_roots && d_ts && d_out, "sample: null device pointer");
  GF_REQUIRE(out_bytes >= output_bytes(R), "sample: output buffer too small");
  DeviceGuard dg(graph_->device());
  reserve_workspace(root_bound(R, L - 1), L * NS, stream);
  uint64_t* d_counts = reinterpret_cast<uint64_t*>(
      ws_.as<char>() + align_up(ws_roots_ * 8, 16) + 3 * align_up(ws_roots_ * 4, 16));

  std::vector<BlockPtrs>& ptrs = slot->ptrs;
  char* p = static_cast<char*>(d_out);
  for (size_t l = 0; l < L; ++l) {
    const size_t Rb = root_bound(R, l);
    for (size_t s = 0; s < NS; ++s) {
      ptrs[l * NS + s] = carve(p, Rb, fanouts_[l]);
      p += layer_output_bytes(Rb, l);
    }
  }
  slot->seq = ++publish_seq_;
  uint64_t* rec = h_counts_.as<uint64_t>() + (slot->seq % kMaxInFlight) * rec_words_;
  *reinterpret_cast<volatile uint64_t*>(rec) = 0;   // this record's publish is pending
  Publish pub;
  pub.d_counts = d_counts;
  pub.h_counts = rec + 1;   // word 0 is the sequence flag
  pub.h_flag = rec;
  pub.seq = slot->seq;
  pub.num_words = static_cast<uint32_t>(L * NS * 2);
  // No publish kernel: the sample's LAST kernel copies the sizes to pinned memory and the host
  // polls the stream's event for completion — one launch less per sample on the sampling
  // stream (round 4, seven same-box pairs: 32.6-32.8 us per step against 33.0-33.7, and none of
  // the occasional 36-37 us runs; GNNFLOW_PUBLISH_EVENT=0: the publish kernel and its flag)
  static const bool by_event = [] {
    const char* v = std::getenv("GNNFLOW_PUBLISH_EVENT");
    return !(v && std::atoi(v) == 0);
  }();
  slot->by_event = by_event;
  for (size_t l = 0; l < L; ++l) {
    const size_t Rb = root_bound(R, l);
    for (size_t s = 0; s < NS; ++s) {
      const size_t b = l * NS + s;
      uint64_t* cslot = d_counts + 2 * b;
      // the next layer of the same snapshot reads its root count R + S from next_R
      uint64_t* next_R = (l + 1 < L) ? cslot + 2 * NS : nullptr;
      const void* last = (by_event && b + 1 == L * NS) ? &pub : nullptr;
      if (l == 0) {
        enqueue_layer(d_roots, d_ts, Rb, nullptr, R, l, s, ptrs[b], cslot, next_R, stream, last);
      } else {
        const BlockPtrs& prev = ptrs[(l - 1) * NS + s];
        enqueue_layer(prev.all_nodes, prev.all_ts, Rb, cslot, 0, l, s, ptrs[b], cslot, next_R,
                      stream, last);
      }
    }
  }
  if (!by_event) {
    sample_publish_kernel<<<dim3(1), dim3(64), 0, stream>>>(pub);
    GF_HIP(hipGetLastError());
  }
  GF_HIP(hipEventRecord(slot->done, stream));
  std::lock_guard<std::mutex> lk(ring_mu_);
  ++ring_count_;
}

size_t Sampler::in_flight() const {
  std::lock_guard<std::mutex> lk(ring_mu_);
  return ring_count_;
}

void Sampler::sample_end(gf_block* blocks) {
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  GF_REQUIRE(blocks != nullptr, "sample: null blocks array");
  InFlight* slot;
  {
    std::lock_guard<std::mutex> lk(ring_mu_);
    GF_REQUIRE(ring_count_ > 0, "sample_end: no sample in flight");
    slot = &ring_[ring_head_];
  }
  auto pop = [&] {
    std::lock_guard<std::mutex> lk(ring_mu_);
    ring_head_ = (ring_head_ + 1) % kMaxInFlight;
    --ring_count_;
  };
  if (slot->roots == 0) {
    for (size_t b = 0; b < L * NS; ++b) std::memset(&blocks[b], 0, sizeof(gf_block));
    last_overflow_ = false;
    pop();
    return;
  }
  DeviceGuard dg(graph_->device());
  // spin on the pinned sequence word (sub-microsecond reaction); fall back to the event
  // if the kernel has not published after a generous number of polls
  const uint64_t* rec = h_counts_.as<uint64_t>() + (slot->seq % kMaxInFlight) * rec_words_;
  volatile const uint64_t* flag = rec;
  bool seen = false;
  if (slot->by_event) {   // completion = the event behind the sample's last kernel
    for (uint64_t spin = 0;; ++spin) {
      const hipError_t q = hipEventQuery(slot->done);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) { pop(); GF_HIP(q); }
      if (spin > 4096 && (spin & 63) == 0) sched_yield();
    }
    seen = true;
  }
  for (uint64_t spin = 0; !seen && spin < (1ull << 26); ++spin) {
    if (*flag == slot->seq) { seen = true; break; }
    __builtin_ia32_pause();
    // a short pure spin covers the usual few microseconds; beyond that give the core away
    // (8 ranks per node each have a spinner and an enqueue thread)
    if (spin > 4096 && (spin & 63) == 0) sched_yield();
  }
  if (!seen) {
    const hipError_t e = hipEventSynchronize(slot->done);
    if (e != hipSuccess) { pop(); GF_HIP(e); }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const uint64_t* hc = rec + 1;
  for (size_t b = 0; b < L * NS; ++b) {
    gf_block& o = blocks[b];
    o.all_nodes = slot->ptrs[b].all_nodes;
    o.all_timestamps = slot->ptrs[b].all_ts;
    o.delta_timestamps = slot->ptrs[b].dt;
    o.eids = slot->ptrs[b].eids;
    o.row = slot->ptrs[b].row;
    o.col = slot->ptrs[b].col;
    o.num_dst_nodes = hc[2 * b];
    o.num_edges = hc[2 * b + 1];
    o.num_src_nodes = o.num_dst_nodes + o.num_edges;
  }
  last_overflow_ = hc[2 * L * NS] != 0;
  pop();
}

void Sampler::sample(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                     size_t out_bytes, gf_block* blocks, hipStream_t stream) {
  GF_REQUIRE(blocks != nullptr, "sample: null blocks array");
  sample_begin(d_roots, d_ts, R, d_out, out_bytes, stream);
  sample_end(blocks);
}

// TemporalSampler::SampleLayer, temporal_sampler.cu:97-277
void Sampler::sample_layer(const int64_t* d_roots, const float* d_ts, size_t R, uint32_t layer,
                           uint32_t snapshot, void* d_out, size_t out_bytes, gf_block* block,
                           hipStream_t stream) {
  GF_REQUIRE(layer < fanouts_.size(), "sample_layer: layer out of range");
  GF_REQUIRE(snapshot < num_snapshots_, "sample_layer: snapshot out of range");
  GF_REQUIRE(block != nullptr, "sample_layer: null block");
  if (R == 0) {
    std::memset(block, 0, sizeof(gf_block));
    calls_++;
    return;
  }
  GF_REQUIRE(d_roots && d_ts && d_out, "sample_layer: null device pointer");
  GF_REQUIRE(out_bytes >= layer_output_bytes(R, layer), "sample_layer: output buffer too small");
  DeviceGuard dg(graph_->device());
  reserve_workspace(R, 2, stream);
  uint64_t* d_counts = reinterpret_cast<uint64_t*>(
      ws_.as<char>() + align_up(ws_roots_ * 8, 16) + 3 * align_up(ws_roots_ * 4, 16));
  BlockPtrs ptrs = carve(static_cast<char*>(d_out), R, fanouts_[layer]);
  enqueue_layer(d_roots, d_ts, R, nullptr, R, layer, snapshot, ptrs, d_counts, nullptr, stream,
                nullptr);
  GF_HIP(hipMemcpyAsync(h_layer_counts_.data(), d_counts, 2 * sizeof(uint64_t),
                        hipMemcpyDeviceToHost, stream));
  GF_HIP(hipStreamSynchronize(stream));
  const uint64_t* hc = h_layer_counts_.as<uint64_t>();
  block->all_nodes = ptrs.all_nodes;
  block->all_timestamps = ptrs.all_ts;
  block->delta_timestamps = ptrs.dt;
  block->eids = ptrs.eids;
  block->row = ptrs.row;
  block->col = ptrs.col;
  block->num_dst_nodes = hc[0];
  block->num_edges = hc[1];
  block->num_src_nodes = hc[0] + hc[1];
}

// ---- partitioned sampling ------------------------------------------------------------
void Sampler::sample_layer_padded(const int64_t* d_requests, size_t n, uint32_t layer,
                                  uint32_t snapshot, int64_t* d_out, hipStream_t stream) {
  GF_REQUIRE(layer < fanouts_.size(), "sample_layer_padded: layer out of range");
  GF_REQUIRE(snapshot < num_snapshots_, "sample_layer_padded: snapshot out of range");
  const uint64_t call = calls_++;
  if (n == 0) return;
  GF_REQUIRE(d_requests && d_out, "sample_layer_padded: null device pointer");
  const uint32_t F = fanouts_[layer];
  GF_REQUIRE(static_cast<uint64_t>(n) * F < 0xFFFFFFFFull,
             "sampler: more than 2^32-1 slots in one layer");
  DeviceGuard dg(graph_->device());
  const GraphView gv = view_for(graph_, n);
  const int uniform = policy_ == GF_SAMPLING_POLICY_UNIFORM;
  const int width = n > kSmallRoots ? large_group_ : search_group_;
  const unsigned grid = capped_grid(n, kSearchThreads / width, 256 * 8);
  ProfileScope ps(kProfSearch, stream);
  launch_padded(width, grid, stream, gv, d_requests, static_cast<uint64_t>(n), snapshot,
                num_snapshots_, window_, F, uniform, prop_time_ ? 1 : 0, seed_, call, d_out,
                static_cast<const uint64_t*>(nullptr), static_cast<const uint64_t*>(nullptr),
                static_cast<uint64_t>(0), static_cast<const uint32_t*>(nullptr),
                static_cast<uint32_t*>(nullptr), 0u, 0u, static_cast<uint32_t*>(nullptr));
  GF_HIP(hipGetLastError());
}

void Sampler::merge_padded(const int64_t* d_roots, const float* d_ts, size_t R, uint32_t layer,
                           const int64_t* d_replies, const uint32_t* d_pos, void* d_out,
                           size_t out_bytes, gf_block* block, hipStream_t stream) {
  GF_REQUIRE(layer < fanouts_.size(), "merge_padded: layer out of range");
  GF_REQUIRE(block != nullptr, "merge_padded: null block");
  if (R == 0) {
    std::memset(block, 0, sizeof(gf_block));
    return;
  }
  GF_REQUIRE(d_roots && d_ts && d_replies && d_pos && d_out, "merge_padded: null device pointer");
  GF_REQUIRE(out_bytes >= layer_output_bytes(R, layer), "merge_padded: output buffer too small");
  const uint32_t F = fanouts_[layer];
  GF_REQUIRE(static_cast<uint64_t>(R) * F < 0xFFFFFFFFull,
             "sampler: more than 2^32-1 slots in one layer");
  DeviceGuard dg(graph_->device());
  reserve_workspace(R, 2, stream);
  char* w = ws_.as<char>();
  w += align_up(ws_roots_ * 8, 16);                                   // rec_end: unused here
  uint32_t* rec_cnt = reinterpret_cast<uint32_t*>(w); w += align_up(ws_roots_ * 4, 16);
  uint32_t* base = reinterpret_cast<uint32_t*>(w);    w += align_up(ws_roots_ * 4, 16);
  uint32_t* tile_scratch = reinterpret_cast<uint32_t*>(w); w += align_up(ws_roots_ * 4, 16);
  uint64_t* d_counts = reinterpret_cast<uint64_t*>(w);
  BlockPtrs out = carve(static_cast<char*>(d_out), R, F);
  merge_count_kernel<<<dim3(static_cast<unsigned>((R + 255) / 256)), dim3(256), 0, stream>>>(
      d_replies, d_pos, nullptr, R, F, rec_cnt, 0u, 0u);
  if (R <= 65536) {
    sample_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
        rec_cnt, base, nullptr, R, F, 0, d_counts, d_counts + 1, nullptr);
  } else {
    const size_t tiles = (R + kScanTile - 1) / kScanTile;
    uint32_t* tile_sum = tile_scratch;
    uint32_t* tile_base = tile_scratch + tiles;
    const unsigned grid = static_cast<unsigned>(std::min<size_t>(tiles, 2048));
    sample_tile_sum_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(rec_cnt, nullptr, R, F,
                                                                         0, tile_sum);
    sample_tile_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
        tile_sum, tile_base, nullptr, R, d_counts, d_counts + 1, nullptr);
    sample_tile_apply_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(
        rec_cnt, tile_base, nullptr, R, F, 0, base);
  }
  merge_emit_kernel<<<dim3(capped_grid(static_cast<uint64_t>(R) * F, kEmitThreads, 256 * 16)),
                      dim3(kEmitThreads), 0, stream>>>(
      d_roots, d_ts, nullptr, R, F, d_replies, d_pos, rec_cnt, base, out.all_nodes, out.all_ts,
      out.dt, out.eids, out.row, out.col);
  GF_HIP(hipGetLastError());
  GF_HIP(hipMemcpyAsync(h_layer_counts_.data(), d_counts, 2 * sizeof(uint64_t),
                        hipMemcpyDeviceToHost, stream));
  GF_HIP(hipStreamSynchronize(stream));
  const uint64_t* hc = h_layer_counts_.as<uint64_t>();
  block->all_nodes = out.all_nodes;
  block->all_timestamps = out.all_ts;
  block->delta_timestamps = out.dt;
  block->eids = out.eids;
  block->row = out.row;
  block->col = out.col;
  block->num_dst_nodes = hc[0];
  block->num_edges = hc[1];
  block->num_src_nodes = hc[0] + hc[1];
}

std::atomic<uint64_t> g_part_host_ns[8];

// Tag of the look-back granules of one fused-merge launch: unique in the PROCESS, not per
// sampler — a sampler's workspace may be memory another sampler's launches wrote granules into
// (freed and allocated again), and a stale granule must never carry a tag a later launch uses.
std::atomic<uint64_t> g_merge_epoch{0};
inline uint64_t next_merge_tag() { return (g_merge_epoch.fetch_add(1) + 1) << 10; }

// ---- partitioned sampling, chained on the device ------------------------------------------
// part_begin -> for every (layer, snapshot): part_plan_own, [the caller's exchange: request
// all-to-all-v, sample_layer_padded for what it received, reply all-to-all-v], part_merge ->
// part_commit.  Every kernel takes the layer's root count from device memory (it is the
// previous layer's R + S), so nothing is read back between the layers; part_commit publishes
// the block sizes like sample_begin does and sample_end() returns them.  With one rank there
// is no exchange and sample_partitioned() issues the whole chain in one call.
void Sampler::part_layout(size_t R0, uint32_t layer, int world_size, double slack,
                          size_t slot_roots, gf_part_layout* out, bool skip_prev) const {
  GF_REQUIRE(layer < fanouts_.size(), "part_layout: layer out of range");
  GF_REQUIRE(out != nullptr, "part_layout: null output");
  GF_REQUIRE(slack >= 0.0, "part_layout: negative slack");
  const size_t Rb = root_bound(R0, layer), F = fanouts_[layer];
  // slotted form (slack > 0): per-peer capacity = the even share of the worst-case root count
  // times `slack`, never more than the layer can have; rows = P slots of (header + cap) + the
  // own share's region.  The capacity follows from `slot_roots` — the batch size every rank
  // agreed on — not from this rank's own R0: the slots must have the same size on all ranks.
  size_t stride = 0, rows = Rb;
  if (slack > 0.0) {
    // (skip_prev: the layer's first roots — the previous layer's — are not requested again, so
    // the slots hold at most the roots that are new in this layer)
    const size_t Sb = root_bound(std::max<size_t>(slot_roots, 1), layer) -
                      ((skip_prev && layer > 0) ? root_bound(std::max<size_t>(slot_roots, 1), layer - 1) : 0);
    const double share = std::ceil(static_cast<double>(Sb) * slack / world_size);
    const size_t cap = std::max<size_t>(1, std::min<size_t>(Sb, static_cast<size_t>(share)));
    stride = cap + 1;
    rows = static_cast<size_t>(world_size) * stride + Rb;
    GF_REQUIRE(rows < 0xFFFFFFFFull, "part_layout: more than 2^32-1 request rows");
  }
  out->root_bound = Rb;
  out->requests = 0;
  out->replies = rows * 16;
  out->counts = out->replies + rows * F * 24;
  out->pos = out->counts + align_up(static_cast<size_t>(world_size) * 8, 16);
  out->scratch = align_up(out->pos + Rb * 4, 16);
  out->scratch_bytes = partition_scratch_bytes(Rb, world_size);
  size_t end = align_up(out->scratch + out->scratch_bytes, 256);
  out->slot_stride = stride;
  out->inbox = out->served = 0;
  if (stride) {
    const size_t slots = static_cast<size_t>(world_size) * stride;
    out->inbox = end;
    out->served = align_up(out->inbox + slots * 16, 256);
    end = align_up(out->served + slots * F * 24, 256);
  }
  out->total = end;
}

void Sampler::part_begin(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                         size_t out_bytes, int world_size, int rank, double slack,
                         size_t slot_roots, hipStream_t stream) {
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  GF_REQUIRE(!part_.active, "part_begin: a partitioned sample is already being built");
  GF_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "part_begin: bad rank / world");
  GF_REQUIRE(R == 0 || (d_roots && d_ts), "part_begin: null device pointer");
  // layers chain through worst-case-sized blocks even when this rank has no root of its own
  // (R = 0): it still plans, serves the other ranks' requests and merges empty blocks
  const size_t Rs = std::max<size_t>(R, 1);
  GF_REQUIRE(d_out && out_bytes >= output_bytes(Rs), "part_begin: output buffer too small");
  InFlight* slot;
  {
    std::lock_guard<std::mutex> lk(ring_mu_);
    GF_REQUIRE(ring_count_ < kMaxInFlight, "part_begin: too many samples in flight on this sampler");
    slot = &ring_[(ring_head_ + ring_count_) % kMaxInFlight];
  }
  DeviceGuard dg(graph_->device());
  // root_of[] is indexed by request ROW: the slotted form has more rows than roots
  gf_part_layout last;
  part_layout(Rs, static_cast<uint32_t>(L - 1), world_size, slack, slot_roots, &last);
  reserve_workspace(std::max(root_bound(Rs, L - 1), last.replies / 16), L * NS, stream);
  slot->ptrs.assign(L * NS, BlockPtrs{});
  slot->roots = Rs;          // never the "R = 0 short-circuit" record: sizes come from the device
  slot->by_event = false;    // a partitioned sample publishes through its publish kernel
  slot->stream = stream;
  char* p = static_cast<char*>(d_out);
  for (size_t l = 0; l < L; ++l) {
    const size_t Rb = root_bound(Rs, l);
    for (size_t s = 0; s < NS; ++s) {
      slot->ptrs[l * NS + s] = carve(p, Rb, fanouts_[l]);
      p += layer_output_bytes(Rb, l);
    }
  }
  part_ = PartState{};
  part_.active = true;
  part_.slot = slot;
  part_.d_roots = d_roots;
  part_.d_ts = d_ts;
  part_.R = R;
  part_.Rs = Rs;
  part_.world = world_size;
  part_.rank = rank;
  part_.slack = slack;
  part_.slot_roots = slot_roots;
  part_.stream = stream;
}

// the sample-wide "a slot overflowed somewhere" word: behind the block counters
uint32_t* Sampler::part_overflow() const {
  return reinterpret_cast<uint32_t*>(part_counts() + 2 * ws_blocks_);
}

// Scratch of the chained merge in the sampler's workspace (ordered by the stream like the rest
// of it): rec_cnt[i] = valid slots of root i, root_of[row] = root of a request / reply row
// (the workspace's `base` array, which the fused merge does not need).
uint32_t* Sampler::part_rec_cnt() const {
  return reinterpret_cast<uint32_t*>(ws_.as<char>() + align_up(ws_roots_ * 8, 16));
}
uint32_t* Sampler::part_root_of() const {
  return reinterpret_cast<uint32_t*>(ws_.as<char>() + align_up(ws_roots_ * 8, 16) +
                                     align_up(ws_roots_ * 4, 16));
}
// Layers that take the fused merge get their own share's counts from the sampling kernel.
bool Sampler::part_own_counts(size_t root_bound) const {
  return fused_scan_ && root_bound <= kSmallRoots && root_bound > 0;
}

// Slotted form, layers of <= kSmallRoots roots: the merge is ONE launch (merge_slots_fused_kernel);
// neither the per-root counts nor root_of[] are needed then.  GNNFLOW_PART_FUSED_MERGE=0: the
// count + emit pair.
bool Sampler::part_fused_merge(size_t root_bound, uint32_t fanout) const {
  static const bool on = [] {
    const char* v = std::getenv("GNNFLOW_PART_FUSED_MERGE");
    return !(v && std::atoi(v) == 0);
  }();
  return on && part_.slack > 0.0 && part_own_counts(root_bound) && fanout <= kEmitThreads &&
         (root_bound * fanout + kEmitThreads - 1) / kEmitThreads <= ws_roots_;
}

uint64_t* Sampler::part_counts() const {
  return reinterpret_cast<uint64_t*>(ws_.as<char>() + align_up(ws_roots_ * 8, 16) +
                                     3 * align_up(ws_roots_ * 4, 16));
}

// roots of (layer, snapshot): the caller's for layer 0, else the previous layer's block
void Sampler::part_roots(uint32_t layer, uint32_t snapshot, const int64_t** roots,
                         const float** ts, const uint64_t** d_R, uint64_t* R_host) const {
  const size_t NS = num_snapshots_;
  if (layer == 0) {
    *roots = part_.d_roots;
    *ts = part_.d_ts;
    *d_R = nullptr;
    *R_host = part_.R;
  } else {
    const BlockPtrs& prev = part_.slot->ptrs[(layer - 1) * NS + snapshot];
    *roots = prev.all_nodes;
    *ts = prev.all_ts;
    *d_R = part_counts() + 2 * (layer * NS + snapshot);   // written by the previous merge
    *R_host = 0;
  }
}

void Sampler::part_plan_own(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes,
                            int phases) {
  GF_REQUIRE(part_.active, "part_plan_own: no partitioned sample is being built");
  GF_REQUIRE(layer < fanouts_.size() && snapshot < num_snapshots_, "part_plan_own: out of range");
  gf_part_layout lay;
  part_layout(part_.Rs, layer, part_.world, part_.slack, part_.slot_roots, &lay);
  GF_REQUIRE(d_ws && ws_bytes >= lay.total, "part_plan_own: workspace too small");
  DeviceGuard dg(graph_->device());
  char* w = static_cast<char*>(d_ws);
  const int64_t* roots; const float* ts; const uint64_t* d_R; uint64_t R_host;
  part_roots(layer, snapshot, &roots, &ts, &d_R, &R_host);
  hipStream_t stream = part_.stream;
  const size_t Rb = lay.root_bound;
  const uint32_t stride = static_cast<uint32_t>(lay.slot_stride);
  uint64_t* d_counts = reinterpret_cast<uint64_t*>(w + lay.counts);
  if (!(phases & 1)) {
    // planned by an earlier call
  } else if (layer == 0 && part_.R == 0 && !stride) {
    GF_HIP(hipMemsetAsync(d_counts, 0, part_.world * sizeof(uint64_t), stream));
  } else {
    // slotted: the sample's first plan STORES the overflow word (the workspace is shared by
    // the samples in flight on this stream), the later ones only raise it
    partition_plan_dev(roots, ts, d_R, layer == 0 ? part_.R : Rb, part_.world, part_.rank,
                       reinterpret_cast<int64_t*>(w + lay.requests),
                       reinterpret_cast<uint32_t*>(w + lay.pos), d_counts, w + lay.scratch,
                       lay.scratch_bytes, graph_->device(), stream,
                       (part_own_counts(layer == 0 ? part_.R : Rb) &&
                        !part_fused_merge(layer == 0 ? part_.R : Rb, fanouts_[layer]))
                           ? part_root_of() : nullptr,
                       stride, stride ? part_overflow() : nullptr,
                       layer == 0 && snapshot == 0 ? 1 : 0);
  }
  if (!(phases & 2)) return;
  // this rank's own share: the last counts[rank] request rows (slotted: the rows from
  // world * stride on); the kernel takes the count from the device, so with one rank nothing
  // is read back, and with several the caller issues it right after starting the request
  // exchange, which it then overlaps
  const uint64_t call = calls_++;
  const uint32_t F = fanouts_[layer];
  const size_t n_bound = layer == 0 ? part_.R : Rb;
  if (n_bound) {
    const int width = n_bound > kSmallRoots ? large_group_ : search_group_;
    const unsigned grid = capped_grid(n_bound, kSearchThreads / width, 256 * 8);
    ProfileScope ps(kProfSearch, stream);
    launch_padded(width, grid, stream, view_for(graph_, n_bound),
                  reinterpret_cast<const int64_t*>(w + lay.requests), static_cast<uint64_t>(0),
                  snapshot, num_snapshots_, window_, F, policy_ == GF_SAMPLING_POLICY_UNIFORM ? 1 : 0,
                  prop_time_ ? 1 : 0, seed_, call, reinterpret_cast<int64_t*>(w + lay.replies),
                  static_cast<const uint64_t*>(d_counts + part_.rank), d_R,
                  static_cast<uint64_t>(R_host), static_cast<const uint32_t*>(part_root_of()),
                  (part_own_counts(n_bound) && !part_fused_merge(n_bound, F))
                      ? part_rec_cnt() : static_cast<uint32_t*>(nullptr),
                  stride, static_cast<uint32_t>(part_.world), static_cast<uint32_t*>(nullptr));
    GF_HIP(hipGetLastError());
  }
}

// Slotted form: serves the request inbox (what the equal-split exchange delivered: one slot per
// rank) from this rank's shard into `served`, reply row = request row; the caller sends
// `served` back slot for slot into the prefix of the reply buffer.
void Sampler::part_serve(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes,
                         bool with_own) {
  GF_REQUIRE(part_.active, "part_serve: no partitioned sample is being built");
  GF_REQUIRE(layer < fanouts_.size() && snapshot < num_snapshots_, "part_serve: out of range");
  gf_part_layout lay;
  part_layout(part_.Rs, layer, part_.world, part_.slack, part_.slot_roots, &lay);
  GF_REQUIRE(lay.slot_stride, "part_serve: the sample was not begun in the slotted form");
  GF_REQUIRE(d_ws && ws_bytes >= lay.total, "part_serve: workspace too small");
  DeviceGuard dg(graph_->device());
  char* w = static_cast<char*>(d_ws);
  hipStream_t stream = part_.stream;
  const uint64_t call = calls_++;
  const uint32_t F = fanouts_[layer];
  const uint32_t stride = static_cast<uint32_t>(lay.slot_stride);
  const uint64_t n = static_cast<uint64_t>(part_.world) * stride;
  GF_REQUIRE(n * F < 0xFFFFFFFFull, "sampler: more than 2^32-1 slots in one layer");
  if (with_own) {
    // the received requests and this rank's own share in ONE launch (part_plan_own phase 2 is
    // then not called for this layer)
    const int64_t* roots; const float* ts; const uint64_t* d_R; uint64_t R_host;
    part_roots(layer, snapshot, &roots, &ts, &d_R, &R_host);
    const size_t n_bound = layer == 0 ? part_.R : lay.root_bound;
    const uint64_t call_own = calls_++;
    const size_t n_max = std::max<size_t>(n, n_bound);
    // width by the layer's roots, not by the (mostly empty) slot rows
    const size_t n_real = std::max<size_t>(lay.root_bound, n_bound);
    const int width = n_real > kSmallRoots ? large_group_ : search_group_;
    const unsigned grid = capped_grid(n_max, kSearchThreads / width, 256 * 8);
    uint64_t* d_counts = reinterpret_cast<uint64_t*>(w + lay.counts);
    const PaddedCommon pc{snapshot, num_snapshots_, window_, F,
                          policy_ == GF_SAMPLING_POLICY_UNIFORM ? 1 : 0, prop_time_ ? 1 : 0, seed_};
    const PaddedJob serve{reinterpret_cast<const int64_t*>(w + lay.inbox), n, call,
                          reinterpret_cast<int64_t*>(w + lay.served), nullptr, nullptr, 0, nullptr,
                          nullptr, stride, static_cast<uint32_t>(part_.world), part_overflow()};
    const PaddedJob own{reinterpret_cast<const int64_t*>(w + lay.requests), 0, call_own,
                        reinterpret_cast<int64_t*>(w + lay.replies), d_counts + part_.rank, d_R,
                        R_host, part_root_of(),
                        (part_own_counts(n_bound) && !part_fused_merge(n_bound, F))
                            ? part_rec_cnt() : nullptr, stride,
                        static_cast<uint32_t>(part_.world), nullptr};
    ProfileScope ps(kProfSearch, stream);
    launch_padded_pair(width, grid, stream, view_for(graph_, n_real), pc, serve, own);
    GF_HIP(hipGetLastError());
    return;
  }
  const int width = n > kSmallRoots ? large_group_ : search_group_;
  const unsigned grid = capped_grid(n, kSearchThreads / width, 256 * 8);
  ProfileScope ps(kProfSearch, stream);
  launch_padded(width, grid, stream, view_for(graph_, n),
                reinterpret_cast<const int64_t*>(w + lay.inbox), n, snapshot, num_snapshots_,
                window_, F, policy_ == GF_SAMPLING_POLICY_UNIFORM ? 1 : 0, prop_time_ ? 1 : 0,
                seed_, call, reinterpret_cast<int64_t*>(w + lay.served),
                static_cast<const uint64_t*>(nullptr), static_cast<const uint64_t*>(nullptr),
                static_cast<uint64_t>(0), static_cast<const uint32_t*>(nullptr),
                static_cast<uint32_t*>(nullptr), stride, static_cast<uint32_t>(part_.world),
                part_overflow());
  GF_HIP(hipGetLastError());
}

void Sampler::part_merge(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes) {
  GF_REQUIRE(part_.active, "part_merge: no partitioned sample is being built");
  GF_REQUIRE(layer < fanouts_.size() && snapshot < num_snapshots_, "part_merge: out of range");
  gf_part_layout lay;
  part_layout(part_.Rs, layer, part_.world, part_.slack, part_.slot_roots, &lay);
  GF_REQUIRE(d_ws && ws_bytes >= lay.total, "part_merge: workspace too small");
  DeviceGuard dg(graph_->device());
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  char* w = static_cast<char*>(d_ws);
  const int64_t* roots; const float* ts; const uint64_t* d_R; uint64_t R_host;
  part_roots(layer, snapshot, &roots, &ts, &d_R, &R_host);
  hipStream_t stream = part_.stream;
  const size_t Rb = layer == 0 ? part_.R : lay.root_bound;
  const uint32_t F = fanouts_[layer];
  const size_t b = layer * NS + snapshot;
  uint64_t* slot = part_counts() + 2 * b;
  uint64_t* next_R = (layer + 1 < L) ? slot + 2 * NS : nullptr;
  const BlockPtrs& out = part_.slot->ptrs[b];
  char* sw = ws_.as<char>();
  sw += align_up(ws_roots_ * 8, 16);                                   // rec_end: unused here
  uint32_t* rec_cnt = reinterpret_cast<uint32_t*>(sw); sw += align_up(ws_roots_ * 4, 16);
  uint32_t* base = reinterpret_cast<uint32_t*>(sw);    sw += align_up(ws_roots_ * 4, 16);
  uint32_t* tile_scratch = reinterpret_cast<uint32_t*>(sw);
  const int64_t* rep = reinterpret_cast<const int64_t*>(w + lay.replies);
  const uint32_t* pos = reinterpret_cast<const uint32_t*>(w + lay.pos);
  if (Rb == 0) {   // layer 0 of a rank without roots: an empty block, R = S = 0
    GF_HIP(hipMemsetAsync(slot, 0, 2 * sizeof(uint64_t), stream));
    if (next_R) GF_HIP(hipMemsetAsync(next_R, 0, sizeof(uint64_t), stream));
    return;
  }
  ProfileScope ps(kProfEmit, stream);
  if (lay.slot_stride && part_fused_merge(Rb, F)) {
    // granules: the workspace's rec_end array (8 B per root, unused by the partitioned path)
    const unsigned egrid = static_cast<unsigned>(
        (static_cast<uint64_t>(Rb) * F + kEmitThreads - 1) / kEmitThreads);
    const uint64_t tag = next_merge_tag();
    merge_slots_fused_kernel<<<dim3(egrid), dim3(kEmitThreads), 0, stream>>>(
        roots, ts, d_R, R_host, F, rep, pos, static_cast<uint32_t>(lay.slot_stride),
        static_cast<uint32_t>(part_.world), reinterpret_cast<uint64_t*>(ws_.as<char>()), tag,
        part_overflow(), out.all_nodes, out.all_ts, out.dt, out.eids, out.row, out.col, slot,
        slot + 1, next_R);
    GF_HIP(hipGetLastError());
    return;
  }
  if (part_own_counts(Rb)) {
    // rec_cnt lives in the sampler workspace; the own share's counts are already there
    // (part_plan_own phase 2), the rows received from other ranks are counted here; the emit
    // derives its own prefix from the counts (no scan launch, no per-workgroup sums)
    (void)base; (void)tile_scratch;
    if (lay.slot_stride) {
      const uint64_t* d_counts = reinterpret_cast<const uint64_t*>(w + lay.counts);
      merge_count_slots_kernel<<<dim3(capped_grid(part_.world * lay.slot_stride + Rb, 256, 1024)),
                                 dim3(256), 0, stream>>>(
          rep, part_root_of(), pos, d_R, R_host, d_counts, static_cast<uint32_t>(lay.slot_stride),
          static_cast<uint32_t>(part_.world), static_cast<uint32_t>(part_.rank), F, rec_cnt);
    } else if (part_.world > 1) {
      uint64_t* d_counts = reinterpret_cast<uint64_t*>(w + lay.counts);
      merge_count_remote_kernel<<<dim3(capped_grid(Rb, 256, 1024)), dim3(256), 0, stream>>>(
          rep, part_root_of(), d_R, R_host, d_counts + part_.rank, F, rec_cnt);
    }
    const unsigned egrid = static_cast<unsigned>(
        (static_cast<uint64_t>(Rb) * F + kEmitThreads - 1) / kEmitThreads);
    merge_emit_prefix_kernel<<<dim3(egrid), dim3(kEmitThreads), 0, stream>>>(
        roots, ts, d_R, R_host, F, rep, pos, rec_cnt, static_cast<const uint32_t*>(nullptr),
        out.all_nodes, out.all_ts, out.dt, out.eids, out.row, out.col, slot, slot + 1, next_R);
    GF_HIP(hipGetLastError());
    return;
  }
  merge_count_kernel<<<dim3(static_cast<unsigned>((Rb + 255) / 256)), dim3(256), 0, stream>>>(
      rep, pos, d_R, R_host, F, rec_cnt, static_cast<uint32_t>(lay.slot_stride),
      static_cast<uint32_t>(part_.world));
  if (Rb <= 65536) {
    sample_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
        rec_cnt, base, d_R, R_host, F, 0, slot, slot + 1, next_R);
  } else {
    const size_t tiles = (Rb + kScanTile - 1) / kScanTile;
    uint32_t* tile_sum = tile_scratch;
    uint32_t* tile_base = tile_scratch + tiles;
    const unsigned grid = static_cast<unsigned>(std::min<size_t>(tiles, 2048));
    sample_tile_sum_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(rec_cnt, d_R, R_host, F,
                                                                         0, tile_sum);
    sample_tile_scan_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
        tile_sum, tile_base, d_R, R_host, slot, slot + 1, next_R);
    sample_tile_apply_kernel<<<dim3(grid), dim3(kScanThreads), 0, stream>>>(
        rec_cnt, tile_base, d_R, R_host, F, 0, base);
  }
  merge_emit_kernel<<<dim3(capped_grid(static_cast<uint64_t>(Rb) * F, kEmitThreads, 256 * 16)),
                      dim3(kEmitThreads), 0, stream>>>(
      roots, ts, d_R, R_host, F, rep, pos, rec_cnt, base, out.all_nodes, out.all_ts, out.dt,
      out.eids, out.row, out.col);
  GF_HIP(hipGetLastError());
}

// the publish record of the sample being built (its pinned words are reset here)
void Sampler::part_commit_prepare(void* publish_out) {
  GF_REQUIRE(part_.active, "part_commit: no partitioned sample is being built");
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  InFlight* slot = part_.slot;
  part_.active = false;
  slot->seq = ++publish_seq_;
  uint64_t* rec = h_counts_.as<uint64_t>() + (slot->seq % kMaxInFlight) * rec_words_;
  *reinterpret_cast<volatile uint64_t*>(rec) = 0;
  Publish& pub = *static_cast<Publish*>(publish_out);
  pub.d_counts = part_counts();
  pub.h_counts = rec + 1;
  pub.h_flag = rec;
  pub.seq = slot->seq;
  pub.num_words = static_cast<uint32_t>(L * NS * 2);
  pub.d_extra = part_.slack > 0.0 ? part_overflow() : nullptr;
}

void Sampler::part_commit_finish() {
  GF_HIP(hipEventRecord(part_.slot->done, part_.stream));
  std::lock_guard<std::mutex> lk(ring_mu_);
  ++ring_count_;
}

void Sampler::part_commit() {
  DeviceGuard dg(graph_->device());
  Publish pub;
  part_commit_prepare(&pub);
  sample_publish_kernel<<<dim3(1), dim3(64), 0, part_.stream>>>(pub);
  GF_HIP(hipGetLastError());
  part_commit_finish();
}

void Sampler::part_abort() { part_.active = false; }

// one rank: no exchange — the whole chain in one call (begin form: sample_end() completes it)
void Sampler::sample_partitioned(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                                 size_t out_bytes, void* d_ws, size_t ws_bytes,
                                 hipStream_t stream) {
  DeviceGuard dg(graph_->device());   // once for the chain: the steps' own guards then find it set
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  part_begin(d_roots, d_ts, R, d_out, out_bytes, 1, 0, 0.0, 0, stream);
  try {
    char* w = static_cast<char*>(d_ws);
    size_t off = 0;
    for (size_t l = 0; l < L; ++l) {
      gf_part_layout lay;
      part_layout(part_.Rs, static_cast<uint32_t>(l), 1, 0.0, 0, &lay);
      for (size_t s = 0; s < NS; ++s) {
        GF_REQUIRE(off + lay.total <= ws_bytes, "sample_partitioned: workspace too small");
        part_plan_own(static_cast<uint32_t>(l), static_cast<uint32_t>(s), w + off, lay.total, 3);
        part_merge(static_cast<uint32_t>(l), static_cast<uint32_t>(s), w + off, lay.total);
        off += lay.total;
      }
    }
    part_commit();
  } catch (...) {
    part_abort();
    throw;
  }
}

// several ranks: plan -> request slots out (equal split) -> own share + serve -> reply slots back
// -> merge, per (layer, snapshot), all enqueued by this one call (dist.py issues the same chain
// step by step when the exchange has to go through torch.distributed)
void Sampler::sample_partitioned_slotted(const int64_t* d_roots, const float* d_ts, size_t R,
                                         void* d_out, size_t out_bytes, void* d_ws,
                                         size_t ws_bytes, double slack, size_t slot_roots,
                                         Exchange& ex, bool overlap, hipStream_t stream) {
  const size_t L = fanouts_.size(), NS = num_snapshots_;
  GF_REQUIRE(slack > 0.0, "sample_partitioned_slotted: slack must be positive");
  DeviceGuard dg(graph_->device());   // once for the chain: the steps' own guards then find it set
  // host time of the issuing thread per stage (gf_debug_part_host_us): the chain is ~13 stream
  // operations issued by ONE thread, which is what bounds its throughput at batch 600
  using clk = std::chrono::steady_clock;
  auto t_prev = clk::now();
  auto lap = [&](int stage) {
    const auto t = clk::now();
    g_part_host_ns[stage].fetch_add(
        std::chrono::duration_cast<std::chrono::nanoseconds>(t - t_prev).count(),
        std::memory_order_relaxed);
    t_prev = t;
  };
  part_begin(d_roots, d_ts, R, d_out, out_bytes, ex.world(), ex.rank(), slack, slot_roots, stream);
  lap(0);
  try {
    char* w = static_cast<char*>(d_ws);
    size_t off = 0;
    for (size_t l = 0; l < L; ++l) {
      gf_part_layout lay;
      part_layout(part_.Rs, static_cast<uint32_t>(l), part_.world, slack, slot_roots, &lay);
      const size_t slot_rows = lay.slot_stride;
      const size_t F = fanouts_[l];
      for (size_t s = 0; s < NS; ++s) {
        GF_REQUIRE(off + lay.total <= ws_bytes, "sample_partitioned_slotted: workspace too small");
        char* b = w + off;
        const uint32_t li = static_cast<uint32_t>(l), si = static_cast<uint32_t>(s);
        part_plan_own(li, si, b, lay.total, 1);
        lap(1);
        if (overlap) {
          ex.all_to_all_forked(b + lay.requests, b + lay.inbox, slot_rows * 16, stream);
          part_plan_own(li, si, b, lay.total, 2);
          ex.join(stream);
        } else {
          ex.all_to_all(b + lay.requests, b + lay.inbox, slot_rows * 16, stream);
        }
        lap(2);
        part_serve(li, si, b, lay.total, /*with_own=*/!overlap);
        lap(3);
        if (overlap) {   // one communicator, one stream: the reply exchange goes there too
          ex.all_to_all_forked(b + lay.served, b + lay.replies, slot_rows * F * 24, stream);
          ex.join(stream);
        } else {
          ex.all_to_all(b + lay.served, b + lay.replies, slot_rows * F * 24, stream);
        }
        lap(4);
        part_merge(li, si, b, lay.total);
        lap(5);
        off += lay.total;
      }
    }
    part_commit();
    lap(6);
    g_part_host_ns[7].fetch_add(1, std::memory_order_relaxed);   // samples
  } catch (...) {
    part_abort();
    throw;
  }
}

// gf_philox4x32_10_first evaluated ON THE DEVICE for n (seed, slot, call) triples: the uniform
// sampler's draws share include/gnnflow_rng.h with the CPU oracle, so a device-side miscompile
// of the Philox rounds would not show in HIP-vs-oracle parity; the Random123 known-answer
// vectors evaluated here would (tests/test_gpu_sampler_parity.py).
__global__ void philox_debug_kernel(const uint64_t* __restrict__ in, size_t n,
                                    uint32_t* __restrict__ out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gf_philox4x32_10_first(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
}
void philox_on_device(const uint64_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream) {
  if (n == 0) return;
  philox_debug_kernel<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream>>>(
      d_in, n, d_out);
  GF_HIP(hipGetLastError());
}

// Tiles whose look-back granule did not arrive in time and were recounted by the waiting thread
// (fused merge), since the library was loaded, on the current device.
uint64_t merge_recounts() {
  unsigned int v = 0;
  GF_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_merge_recounts), sizeof(v)));
  return v;
}

// Host time the issuing thread spent per stage of the slotted chain since the last reset:
// out[0..6] = begin, plan, request exchange, serve, reply exchange, merge, commit (us, summed
// over all samples), out[7] = samples.
void part_host_us(double out[8], bool reset) {
  for (int i = 0; i < 8; ++i) {
    const uint64_t v = reset ? g_part_host_ns[i].exchange(0) : g_part_host_ns[i].load();
    out[i] = i < 7 ? v / 1e3 : static_cast<double>(v);
  }
}

// ---- two samples in ONE chain ---------------------------------------------------------------
// The slotted chain of a sample is ~11 stream operations, and at batch 600 its throughput is
// bound by the host thread that issues them (measured: 3-6 us each), not by the GPU.  Up to
// kMaxGroup = 4 consecutive batches therefore SHARE their launches and exchanges: sample j of
// m runs through its own sampler (clones on the same graph: own output, counters, publish
// record), all write their requests into one buffer — owner q's rows of sample j into slot
// m q + j, so the buffer is P runs of m slots and ONE equal-split all-to-all moves them all —,
// the received slots (m P of them, served alike) and the m own shares are sampled by one
// launch, one exchange brings the replies back, one launch merges all, one publishes all: 11
// operations per m samples.  Conditions (else the caller issues single chains): one snapshot,
// every layer of every sample within the fused plan / fused merge limits (<= 32 768 roots,
// fanout <= 256).  Layout of the shared workspace of layer l, rows of 16 B (requests) and
// fanout x 24 B (replies):  [m P slots of `stride` rows | own share 0 | ... | own share m-1].
size_t Sampler::group_ws_bytes(const Sampler& a, const size_t* R, int m, int world, double slack,
                               size_t slot_roots, bool narrow, double edge_fill, bool reuse_roots) {
  size_t total = 0;
  for (size_t l = 0; l < a.fanouts_.size(); ++l) {
    GroupLayout lay;
    a.group_layout(R, m, static_cast<uint32_t>(l), world, slack, slot_roots, narrow, edge_fill, &lay,
                   reuse_roots);
    total += lay.total;
  }
  return total;
}

bool Sampler::group_ok(const size_t* R, int m) const {
  if (num_snapshots_ != 1 || m < 1 || m > kMaxGroup || !fused_scan_) return false;
  for (size_t l = 0; l < fanouts_.size(); ++l) {
    if (fanouts_[l] > kEmitThreads) return false;
    for (int j = 0; j < m; ++j) {
      const size_t bound = root_bound(std::max<size_t>(R[j], 1), l);
      if (bound > kSmallRoots || bound > kPlanJobsMaxRoots) return false;
    }
  }
  return true;
}

void Sampler::group_layout(const size_t* R, int m, uint32_t layer, int world, double slack,
                           size_t slot_roots, bool narrow, double edge_fill,
                           GroupLayout* out, bool reuse_roots) const {
  GF_REQUIRE(m >= 1 && m <= kMaxGroup, "group layout: 1..4 samples");
  gf_part_layout one;
  part_layout(std::max<size_t>(R[0], 1), layer, world, slack, slot_roots, &one,
              layer_reuses_roots(reuse_roots, layer));   // slot stride
  const size_t F = fanouts_[layer];
  const size_t slot_rows = static_cast<size_t>(m) * world * one.slot_stride;
  size_t bound[kMaxGroup], rows = slot_rows;
  for (int j = 0; j < m; ++j) {
    bound[j] = root_bound(std::max<size_t>(R[j], 1), layer);
    out->own[j] = rows;
    rows += bound[j];
  }
  GF_REQUIRE(rows < 0xFFFFFFFFull, "group layout: more than 2^32-1 request rows");
  out->stride = one.slot_stride;
  out->slot_rows = slot_rows;
  size_t at = 0;
  const size_t rb = narrow ? 12 : 24;   // bytes per reply slot
  out->requests = at; at = align_up(at + rows * 16, 256);
  out->replies = at;  at = align_up(at + rows * F * rb, 256);
  out->inbox = at;    at = align_up(at + slot_rows * 16, 256);
  out->served = at;   at = align_up(at + slot_rows * F * rb, 256);
  for (int j = 0; j < m; ++j) { out->counts[j] = at; at = align_up(at + static_cast<size_t>(world) * 8, 256); }
  for (int j = 0; j < m; ++j) { out->pos[j] = at; at = align_up(at + bound[j] * 4, 256); }
  // first edge of every root in the merged block (+ the total): what the NEXT layer needs to
  // take the edges of the roots it does not request again from this block
  for (int j = 0; j < m; ++j) { out->first[j] = at; at = align_up(at + (bound[j] + 1) * 4, 256); }
  out->edge_cap = out->cslot = out->row_cnt = out->cserved = out->creplies = out->off_bytes = 0;
  if (edge_fill > 0.0) {
    // compact reply slot: offsets [0] = its edges, [r] = edges of the rows before row r
    // (1 <= r < stride), [stride] = "a slot of this sender overflowed"; then the edges.  The
    // first layer's roots are the batch itself — most of them have edges — while deeper layers
    // thin out: layer l gets the share edge_fill^(l / (L - 1)) of its fixed records (1 for the
    // first layer, edge_fill for the last).  16-bit offsets while the capacity allows.
    const size_t L = fanouts_.size();
    const double f = L > 1 ? std::pow(edge_fill, static_cast<double>(layer) / (L - 1)) : 1.0;
    const size_t cap = static_cast<size_t>(
        std::ceil(f * static_cast<double>((one.slot_stride - 1) * F)));
    out->edge_cap = std::max<size_t>(cap, F);
    out->off_bytes = out->edge_cap < 65535 ? 2 : 4;
    out->cslot = align_up(out->off_bytes * (one.slot_stride + 1), 16) + align_up(out->edge_cap * rb, 16);
    const size_t slots = static_cast<size_t>(m) * world;
    out->row_cnt = at;  at = align_up(at + slot_rows * 4, 256);
    out->cserved = at;  at = align_up(at + slots * out->cslot, 256);
    out->creplies = at; at = align_up(at + slots * out->cslot, 256);
  }
  out->total = at;
}

void Sampler::sample_partitioned_group(const GroupSample* gs, int m, void* d_ws, size_t ws_bytes,
                                       double slack, size_t slot_roots, Exchange* ex,
                                       hipStream_t stream, unsigned force_overflow, bool narrow,
                                       double edge_fill, bool reuse_roots) {
  GF_REQUIRE(gs != nullptr && m >= 1 && m <= kMaxGroup, "sample_partitioned_group: 1..4 samples");
  Sampler& a = *gs[0].s;
  size_t Rin[kMaxGroup];
  for (int j = 0; j < m; ++j) {
    GF_REQUIRE(gs[j].s != nullptr, "sample_partitioned_group: null sampler");
    for (int k = 0; k < j; ++k)
      GF_REQUIRE(gs[j].s != gs[k].s, "sample_partitioned_group: the samples need a sampler each");
    const Sampler& b = *gs[j].s;
    GF_REQUIRE(a.graph_ == b.graph_ && a.fanouts_ == b.fanouts_ && a.policy_ == b.policy_ &&
                   a.num_snapshots_ == b.num_snapshots_ && a.window_ == b.window_ &&
                   a.prop_time_ == b.prop_time_ && a.seed_ == b.seed_,
               "sample_partitioned_group: the samplers differ");
    Rin[j] = gs[j].R;
  }
  GF_REQUIRE(slack > 0.0, "sample_partitioned_group: slack must be positive");
  GF_REQUIRE(a.group_ok(Rin, m), "sample_partitioned_group: these samples cannot share a chain");
  DeviceGuard dg(a.graph_->device());
  const size_t L = a.fanouts_.size();
  // ex == null: ONE rank and nothing to exchange (every root is its own): the same chain
  // without its two all-to-alls and without the inbox job
  const int P = ex ? ex->world() : 1, me = ex ? ex->rank() : 0;
  using clk = std::chrono::steady_clock;
  auto t_prev = clk::now();
  auto lap = [&](int stage) {
    const auto t = clk::now();
    g_part_host_ns[stage].fetch_add(
        std::chrono::duration_cast<std::chrono::nanoseconds>(t - t_prev).count(),
        std::memory_order_relaxed);
    t_prev = t;
  };
  int begun = 0;
  auto abort_all = [&]() { for (int j = 0; j < begun; ++j) gs[j].s->part_abort(); };
  try {
    for (; begun < m; ++begun) {
      const GroupSample& g = gs[begun];
      g.s->part_begin(g.d_roots, g.d_ts, g.R, g.d_out, g.out_bytes, P, me, slack, slot_roots,
                      stream);
    }
  } catch (...) {
    abort_all();
    throw;
  }
  lap(0);
  try {
    char* w = static_cast<char*>(d_ws);
    size_t off = 0;
    size_t Rs[kMaxGroup];
    for (int j = 0; j < m; ++j) Rs[j] = gs[j].s->part_.Rs;
    const uint32_t* first_prev[kMaxGroup] = {nullptr, nullptr, nullptr, nullptr};
    for (size_t l = 0; l < L; ++l) {
      // Layer l's first roots ARE layer l - 1's roots, with the same timestamps (all_nodes =
      // roots ++ neighbours): with most-recent sampling and the same fanout their k most recent
      // neighbours are what the previous block already holds, so they are neither bucketed nor
      // requested nor sampled again — the merge copies their edges out of the previous block
      // (the reference requests every root of every layer, dist_sampler.py:174-186).
      const bool reuse = a.layer_reuses_roots(reuse_roots, l);
      GroupLayout lay;
      a.group_layout(Rs, m, static_cast<uint32_t>(l), P, slack, slot_roots, narrow, edge_fill, &lay,
                     reuse_roots);
      GF_REQUIRE(off + lay.total <= ws_bytes, "sample_partitioned_group: workspace too small");
      const size_t rb = narrow ? 12 : 24;
      char* base = w + off;
      const uint32_t F = a.fanouts_[l], stride = static_cast<uint32_t>(lay.stride);
      int64_t* requests = reinterpret_cast<int64_t*>(base + lay.requests);
      int64_t* replies = reinterpret_cast<int64_t*>(base + lay.replies);
      const int64_t* roots[kMaxGroup]; const float* ts[kMaxGroup]; const uint64_t* d_R[kMaxGroup];
      uint64_t R_host[kMaxGroup];
      size_t bound = 0;
      for (int j = 0; j < m; ++j) {
        Sampler& s = *gs[j].s;
        s.part_roots(static_cast<uint32_t>(l), 0, &roots[j], &ts[j], &d_R[j], &R_host[j]);
        bound = std::max(bound, l == 0 ? s.part_.R : s.root_bound(s.part_.Rs, l));
      }
      // 1. all plans
      PlanJob pj[kMaxGroup];
      for (int j = 0; j < m; ++j) {
        pj[j] = PlanJob{roots[j], ts[j], d_R[j], R_host[j], requests,
                        reinterpret_cast<uint32_t*>(base + lay.pos[j]),
                        reinterpret_cast<uint64_t*>(base + lay.counts[j]),
                        gs[j].s->part_overflow(), l == 0 ? 1 : 0, static_cast<uint32_t>(m),
                        static_cast<uint32_t>(j), static_cast<uint32_t>(lay.own[j]),
                        (force_overflow >> j) & 1u};
        if (reuse) {
          const int64_t* r_; const float* t_;
          gs[j].s->part_roots(static_cast<uint32_t>(l - 1), 0, &r_, &t_, &pj[j].d_skip,
                              &pj[j].skip_host);
        }
      }
      partition_plan_jobs(pj, m, bound, P, me, stride, a.graph_->device(), stream);
      lap(1);
      // 2. every sample's request slots out
      if (ex) ex->all_to_all(requests, base + lay.inbox, static_cast<size_t>(m) * stride * 16, stream);
      lap(2);
      // 3. the received slots (of all samples, served alike) and the own shares
      const uint64_t n_inbox = ex ? lay.slot_rows : 0;
      const size_t n_max = std::max<size_t>(n_inbox, bound);
      // group width by the roots there really are (<= the layer's bound per sample), not by the
      // slot rows, most of which are empty: a latency chain wants the 16-lane search
      // (layer 1 of the batch-600 pair: 9.3 -> see profiles/README.md round 4)
      // ... and by ALL the roots of the launch: m samples' layers together no longer fit the
      // GPU with 16 lanes per root, and the launch shares the GPU with the other lanes' chains
      // and the fetch kernels, so roots in flight per wave count for more than search rounds:
      // 2 lanes per root from 4 096 roots on (batch 600, 4 samples per chain, one rank over
      // RCCL: 56.7 us per step with 16 lanes, 50.7 with 4, 46.8 with 2, 43.8 with 2 also for
      // the 7 200-root first layer; profiles/README.md round 4)
      static const int chain_width = group_width_from_env("GNNFLOW_PART_CHAIN_WIDTH", 2);
      static const size_t chain_small = [] {
        const char* v = std::getenv("GNNFLOW_PART_CHAIN_SMALL");
        return v ? static_cast<size_t>(std::atol(v)) : size_t{4096};
      }();
      const int width = static_cast<size_t>(m) * bound > chain_small ? chain_width : a.search_group_;
      const unsigned grid = capped_grid(n_max, kSearchThreads / width, 256 * 8);
      const PaddedCommon pc{0, 1, a.window_, F, a.policy_ == GF_SAMPLING_POLICY_UNIFORM ? 1 : 0,
                            a.prop_time_ ? 1 : 0, a.seed_, narrow ? 1 : 0};
      PaddedJobs jobs;
      jobs.j[0] = PaddedJob{reinterpret_cast<const int64_t*>(base + lay.inbox), n_inbox, a.calls_++,
                            reinterpret_cast<int64_t*>(base + lay.served), nullptr, nullptr, 0,
                            nullptr, nullptr, stride, static_cast<uint32_t>(m * P),
                            a.part_overflow()};
      jobs.j[0].m = static_cast<uint32_t>(m);
      for (int j = 0; j < m; ++j) jobs.j[0].d_overflow_of[j] = gs[j].s->part_overflow();
      const bool compact = ex != nullptr && lay.edge_cap > 0;
      if (compact) jobs.j[0].row_cnt = reinterpret_cast<uint32_t*>(base + lay.row_cnt);
      for (int j = 0; j < m; ++j) {
        PaddedJob& own = jobs.j[1 + j];
        own = PaddedJob{requests, 0, gs[j].s->calls_++, replies,
                        reinterpret_cast<const uint64_t*>(base + lay.counts[j]) + me, d_R[j],
                        R_host[j], nullptr, nullptr, stride, static_cast<uint32_t>(m * P), nullptr};
        own.own_skip = lay.own[j];
      }
      {
        ProfileScope ps(kProfSearch, stream);
        launch_padded_group(width, grid, 1 + m, stream, view_for(a.graph_, bound), pc, jobs);
        GF_HIP(hipGetLastError());
      }
      lap(3);
      // 4. the replies back: the sampled edges packed per slot (compact), or the fixed slots
      if (compact) {
        if (!a.part_ticket_.data()) {
          a.part_ticket_.reserve(256);
          GF_HIP(hipMemsetAsync(a.part_ticket_.data(), 0, 256, stream));
        }
        if (++a.part_tag_ == 0) ++a.part_tag_;   // (a zeroed ticket must never look current)
        reply_compact_kernel<<<dim3(static_cast<unsigned>(m * P)), dim3(kCompactThreads), 0, stream>>>(
            CompactArgs{reinterpret_cast<const int64_t*>(base + lay.inbox), base + lay.served,
                        reinterpret_cast<const uint32_t*>(base + lay.row_cnt), base + lay.cserved,
                        a.part_ticket_.as<unsigned long long>(), stride, F,
                        static_cast<uint32_t>(m), static_cast<uint32_t>(P),
                        static_cast<uint32_t>(lay.edge_cap), static_cast<uint32_t>(lay.cslot),
                        narrow ? 1u : 0u, static_cast<uint32_t>(lay.off_bytes), a.part_tag_});
        GF_HIP(hipGetLastError());
        ex->all_to_all(base + lay.cserved, base + lay.creplies, static_cast<size_t>(m) * lay.cslot, stream);
      } else if (ex) {
        ex->all_to_all(base + lay.served, replies, static_cast<size_t>(m) * stride * F * rb, stream);
      }
      lap(4);
      // 5. all merges
      MergeJobs mj;
      for (int j = 0; j < m; ++j) {
        Sampler& s = *gs[j].s;
        uint64_t* cslot = s.part_counts() + 2 * l;
        const BlockPtrs& out = s.part_.slot->ptrs[l];
        mj.j[j] = MergeJob{roots[j], ts[j], d_R[j], R_host[j], replies,
                           reinterpret_cast<const uint32_t*>(base + lay.pos[j]),
                           static_cast<uint32_t>(lay.slot_rows),
                           reinterpret_cast<uint64_t*>(s.ws_.as<char>()), next_merge_tag(),
                           s.part_overflow(), out.all_nodes, out.all_ts, out.dt, out.eids, out.row,
                           out.col, cslot, cslot + 1, (l + 1 < L) ? cslot + 2 : nullptr};
        if (compact) {
          mj.j[j].crep = base + lay.creplies;
          mj.j[j].cslot = static_cast<uint32_t>(lay.cslot);
          mj.j[j].edge_cap = static_cast<uint32_t>(lay.edge_cap);
          mj.j[j].m = static_cast<uint32_t>(m);
          mj.j[j].jidx = static_cast<uint32_t>(j);
          mj.j[j].off_bytes = static_cast<uint32_t>(lay.off_bytes);
        }
        mj.j[j].first_out = reinterpret_cast<uint32_t*>(base + lay.first[j]);
        if (reuse) {
          const int64_t* r_; const float* t_;
          s.part_roots(static_cast<uint32_t>(l - 1), 0, &r_, &t_, &mj.j[j].d_R_prev,
                       &mj.j[j].R_prev_host);
          const BlockPtrs& pb = s.part_.slot->ptrs[l - 1];
          mj.j[j].first_prev = first_prev[j];
          mj.j[j].nodes_prev = pb.all_nodes;
          mj.j[j].ts_prev = pb.all_ts;
          mj.j[j].dt_prev = pb.dt;
          mj.j[j].eids_prev = pb.eids;
        }
        first_prev[j] = reinterpret_cast<const uint32_t*>(base + lay.first[j]);
      }
      {
        ProfileScope ps(kProfEmit, stream);
        const unsigned egrid = static_cast<unsigned>(
            (static_cast<uint64_t>(std::max<size_t>(bound, 1)) * F + kEmitThreads - 1) /
            kEmitThreads);
        merge_slots_fused_group_kernel<<<dim3(egrid, static_cast<unsigned>(m)), dim3(kEmitThreads),
                                         0, stream>>>(mj, F, stride,
                                                      narrow ? (a.prop_time_ ? 2 : 1) : 0);
        GF_HIP(hipGetLastError());
      }
      lap(5);
      off += lay.total;
    }
    PublishGroup pg;
    for (int j = 0; j < m; ++j) gs[j].s->part_commit_prepare(&pg.p[j]);
    sample_publish_group_kernel<<<dim3(static_cast<unsigned>(m)), dim3(64), 0, stream>>>(pg);
    GF_HIP(hipGetLastError());
    for (int j = 0; j < m; ++j) gs[j].s->part_commit_finish();
    lap(6);
    g_part_host_ns[7].fetch_add(static_cast<uint64_t>(m), std::memory_order_relaxed);   // samples
  } catch (...) {
    abort_all();
    throw;
  }
}

// Copies device-resident blocks into freshly malloc'ed host arrays
// (api.cc:17-24 vec2npy copies likewise).
void Sampler::to_host_blocks(const gf_block* dev, gf_block* host, size_t n, hipStream_t stream) {
  auto dup = [&](const void* d, size_t bytes) -> void* {
    void* h = std::malloc(bytes ? bytes : 1);
    if (!h) throw std::bad_alloc();
    if (bytes) GF_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream));
    return h;
  };
  for (size_t b = 0; b < n; ++b) {
    const gf_block& d = dev[b];
    gf_block& h = host[b];
    h.num_dst_nodes = d.num_dst_nodes;
    h.num_src_nodes = d.num_src_nodes;
    h.num_edges = d.num_edges;
    h.all_nodes = static_cast<int64_t*>(dup(d.all_nodes, d.num_src_nodes * 8));
    h.all_timestamps = static_cast<float*>(dup(d.all_timestamps, d.num_src_nodes * 4));
    h.delta_timestamps = static_cast<float*>(dup(d.delta_timestamps, d.num_edges * 4));
    h.eids = static_cast<int64_t*>(dup(d.eids, d.num_edges * 8));
    h.row = static_cast<int64_t*>(dup(d.row, d.num_edges * 8));
    h.col = static_cast<int64_t*>(dup(d.col, d.num_edges * 8));
  }
  GF_HIP(hipStreamSynchronize(stream));
}

void Sampler::sample_host(const int64_t* nodes, const float* ts, size_t R, gf_block* blocks) {
  const size_t nb = fanouts_.size() * num_snapshots_;
  if (R == 0) {
    // R = 0 short-circuit (temporal_sampler.cu:107-114): empty arrays, 0 nodes
    calls_ += nb;
    for (size_t b = 0; b < nb; ++b) {
      std::memset(&blocks[b], 0, sizeof(gf_block));
      blocks[b].all_nodes = static_cast<int64_t*>(std::malloc(1));
      blocks[b].all_timestamps = static_cast<float*>(std::malloc(1));
      blocks[b].delta_timestamps = static_cast<float*>(std::malloc(1));
      blocks[b].eids = static_cast<int64_t*>(std::malloc(1));
      blocks[b].row = static_cast<int64_t*>(std::malloc(1));
      blocks[b].col = static_cast<int64_t*>(std::malloc(1));
    }
    return;
  }
  GF_REQUIRE(nodes && ts, "sample: null input array");
  DeviceGuard dg(graph_->device());
  const size_t in_bytes = align_up(R * 8, 16) + align_up(R * 4, 16);
  const size_t out_bytes = output_bytes(R);
  if (!own_stream_) GF_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  host_io_.reserve(in_bytes + out_bytes, 0, own_stream_);
  char* d = host_io_.as<char>();
  int64_t* d_nodes = reinterpret_cast<int64_t*>(d);
  float* d_ts = reinterpret_cast<float*>(d + align_up(R * 8, 16));
  GF_HIP(hipMemcpyAsync(d_nodes, nodes, R * 8, hipMemcpyHostToDevice, own_stream_));
  GF_HIP(hipMemcpyAsync(d_ts, ts, R * 4, hipMemcpyHostToDevice, own_stream_));
  std::vector<gf_block> dev(nb);
  sample(d_nodes, d_ts, R, d + in_bytes, out_bytes, dev.data(), own_stream_);
  to_host_blocks(dev.data(), blocks, nb, own_stream_);
}

void Sampler::sample_layer_host(const int64_t* nodes, const float* ts, size_t R, uint32_t layer,
                                uint32_t snapshot, gf_block* block) {
  GF_REQUIRE(layer < fanouts_.size(), "sample_layer: layer out of range");
  GF_REQUIRE(snapshot < num_snapshots_, "sample_layer: snapshot out of range");
  if (R == 0) {
    calls_++;
    std::memset(block, 0, sizeof(gf_block));
    block->all_nodes = static_cast<int64_t*>(std::malloc(1));
    block->all_timestamps = static_cast<float*>(std::malloc(1));
    block->delta_timestamps = static_cast<float*>(std::malloc(1));
    block->eids = static_cast<int64_t*>(std::malloc(1));
    block->row = static_cast<int64_t*>(std::malloc(1));
    block->col = static_cast<int64_t*>(std::malloc(1));
    return;
  }
  GF_REQUIRE(nodes && ts, "sample_layer: null input array");
  DeviceGuard dg(graph_->device());
  const size_t in_bytes = align_up(R * 8, 16) + align_up(R * 4, 16);
  const size_t out_bytes = layer_output_bytes(R, layer);
  if (!own_stream_) GF_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  host_io_.reserve(in_bytes + out_bytes, 0, own_stream_);
  char* d = host_io_.as<char>();
  int64_t* d_nodes = reinterpret_cast<int64_t*>(d);
  float* d_ts = reinterpret_cast<float*>(d + align_up(R * 8, 16));
  GF_HIP(hipMemcpyAsync(d_nodes, nodes, R * 8, hipMemcpyHostToDevice, own_stream_));
  GF_HIP(hipMemcpyAsync(d_ts, ts, R * 4, hipMemcpyHostToDevice, own_stream_));
  gf_block dev;
  sample_layer(d_nodes, d_ts, R, layer, snapshot, d + in_bytes, out_bytes, &dev, own_stream_);
  to_host_blocks(&dev, block, 1, own_stream_);
}

}  // namespace gf
