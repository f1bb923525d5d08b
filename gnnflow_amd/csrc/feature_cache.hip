// Fused feature gather + LRU replacement on MI355X.
//
// Reference behaviour restated (gnnflow/cache/cache.py:255-400, lru_cache.py:121-201),
// per block of ids:
//   out[i,:] = cache_buffer[map[id_i]] if id_i is cached else feats[id_i]
//   hit ratio = #cached / n
//   if update and any miss: count -= 1 for every slot; hit slots -> 0; the
//   k = min(#unique missed ids, capacity) slots with the smallest count are
//   evicted and refilled with the missed ids' rows.
// The reference spends ~10 ATen launches, a host round trip for the missed rows
// (unique -> CPU index_select -> pinned -> H2D) and a topk over the whole capacity
// on this.  Here:
//   * ONE gather kernel reads ids, probes the id->slot map, picks the source row
//     (cache slot in HBM, or the feature table — HBM or device-mapped pinned host
//     memory) and streams it to the output with 16-byte loads/stores; a wave owns
//     64 consecutive output rows so the stores are one contiguous 64*dim*4-byte run
//     and every lane keeps 4 independent 16 B loads in flight.  It also records each
//     row's slot (for the LRU pass) and the hit count.  This kernel moves ~all the
//     bytes (2 * dim * 4 per row) and is the one priced against the HBM roofline.
//   * LRU bookkeeping runs entirely on the device with no host synchronisation:
//     `count` is kept as an epoch stamp per slot (count == stamp - epoch), victims
//     are selected with a two-level 2048-bin histogram select over slot ages
//     (no sort, no topk), ties resolved towards the lowest slot index so the result
//     is deterministic, and the missed rows are installed from the just-written
//     output rows (already in HBM) instead of being fetched a second time.
//     Every bookkeeping kernel exits immediately when the block had no miss
//     (the reference skips update_*_cache in that case too).
#include "feature_cache.hpp"

#include <algorithm>
#include <climits>
#include <cstring>
#include <vector>

namespace gf {

namespace {

constexpr int32_t kAbsent = INT32_MIN;  // map[] value of an uncached id
constexpr int kThreads = 256;
constexpr int kBins = 2048;             // 11 bits per histogram level
constexpr uint32_t kAgeMax = (1u << 22) - 1;
constexpr int kTile = 1024;             // slots per tie-count tile
constexpr int kScanThreads = 1024;

struct Counters {
  uint32_t hits;        // rows served from the cache
  uint32_t n_miss;      // rows served from the feature table
  uint32_t n_unique;    // distinct missed ids
  uint32_t victim_ctr;  // install tickets handed out
  uint32_t reserved[4];
};

struct Workspace {
  Counters* ctr;
  uint32_t* hist_hi;    // [kBins]
  uint32_t* hist_lo;    // [kBins]
  int32_t* slot_of_row; // [n]  >=0 slot (hit), -1 miss, -2 invalid id
  uint32_t* rep_flag;   // [n]  1 = first row of a distinct missed id
  uint32_t* rep_rank;   // [n]  exclusive scan of rep_flag
  uint32_t* rep_row;    // [n]  rank -> row
  uint32_t* tile_tie;   // [tiles] slots at the threshold age
  uint32_t* tile_old;   // [tiles] slots older than the threshold
  uint32_t* tie_base;   // [tiles] exclusive scans of the two
  uint32_t* old_base;   // [tiles]
  uint2* pairs;         // [n]  ticket -> {slot, row}
};

template <typename VecT> __device__ inline VecT vec_zero();
template <> __device__ inline float vec_zero<float>() { return 0.0f; }
template <> __device__ inline float4 vec_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- the gather kernel -------------------------------------------------------------
// VecT = float4 (dim % 4 == 0, 16 B aligned rows) or float.
template <typename VecT>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(
    const int64_t* __restrict__ ids, uint32_t n, const int32_t* __restrict__ map,
    const VecT* __restrict__ cache_buf, const VecT* __restrict__ feats, uint64_t num_ids,
    uint32_t dimv, VecT* __restrict__ out, int32_t* __restrict__ slot_of_row,
    Counters* __restrict__ ctr, uint32_t* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  const uint32_t num_waves = (gridDim.x * kThreads) >> 6;
  const uint32_t tiles = (n + 63) / 64;
  for (uint32_t tile = wave; tile < tiles; tile += num_waves) {
    const uint32_t row0 = tile * 64;
    const uint32_t rows = min(64u, n - row0);
    const VecT* src = nullptr;
    int32_t slot = -2;
    if (lane < static_cast<int>(rows)) {
      const int64_t id = ids[row0 + lane];
      if (id >= 0 && static_cast<uint64_t>(id) < num_ids) {
        slot = map ? map[id] : -1;
        if (slot >= 0) {
          src = cache_buf + static_cast<uint64_t>(slot) * dimv;
        } else {
          slot = -1;
          src = feats + static_cast<uint64_t>(id) * dimv;
        }
      }
      if (slot_of_row) slot_of_row[row0 + lane] = slot;
    }
    if (ctr) {
      const uint32_t hits = __popcll(__ballot(slot >= 0));
      const uint32_t miss = __popcll(__ballot(slot == -1));
      if (lane == 0) {
        if (hits) atomicAdd(&ctr->hits, hits);
        if (miss) atomicAdd(&ctr->n_miss, miss);
        if (stats && hits) atomicAdd(&stats[0], hits);
      }
    }
    const uint64_t src_bits = reinterpret_cast<uint64_t>(src);
    const uint32_t total = rows * dimv;
    VecT* o = out + static_cast<uint64_t>(row0) * dimv;
    // The loop trip count is wave-uniform and every lane executes the cross-lane read:
    // ds_bpermute returns 0 for a source lane that EXEC has switched off, so the
    // row-base broadcast must never sit under a per-lane condition.
    auto load = [&](uint32_t fu, bool* valid) -> VecT {
      *valid = fu < total;
      const uint32_t r = *valid ? fu / dimv : 0u;
      const uint32_t c = fu - r * dimv;
      const VecT* s = reinterpret_cast<const VecT*>(__shfl(src_bits, r, 64));
      return (*valid && s) ? s[c] : vec_zero<VecT>();
    };
    // 4 independent 16-byte loads in flight per lane
    for (uint32_t base = 0; base < total; base += 256) {
      const uint32_t f = base + lane;
      bool p0, p1, p2, p3;
      const VecT v0 = load(f, &p0), v1 = load(f + 64, &p1), v2 = load(f + 128, &p2),
                 v3 = load(f + 192, &p3);
      if (p0) o[f] = v0;
      if (p1) o[f + 64] = v1;
      if (p2) o[f + 128] = v2;
      if (p3) o[f + 192] = v3;
    }
  }
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[1], n);
}

// ---- LRU bookkeeping kernels -------------------------------------------------------
// claim: the lowest row of every distinct missed id wins map[id] = -(row+1)
__global__ void lru_claim_kernel(const int64_t* __restrict__ ids, uint32_t n,
                                 const int32_t* __restrict__ slot_of_row, int32_t* map,
                                 const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || slot_of_row[i] != -1) return;
  atomicMax(&map[ids[i]], -static_cast<int32_t>(i + 1));
}

// mark: representatives of the distinct missed ids; hit slots get the new epoch
// (`self.cache_*_count[cached_index] = 0`, lru_cache.py:138-139)
__global__ void lru_mark_kernel(const int64_t* __restrict__ ids, uint32_t n,
                                const int32_t* __restrict__ slot_of_row,
                                const int32_t* __restrict__ map, uint32_t* __restrict__ stamp,
                                const uint32_t* __restrict__ epoch,
                                uint32_t* __restrict__ rep_flag,
                                const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t s = slot_of_row[i];
  uint32_t rep = 0;
  if (s >= 0) {
    stamp[s] = *epoch + 1;
  } else if (s == -1) {
    rep = map[ids[i]] == -static_cast<int32_t>(i + 1);
  }
  rep_flag[i] = rep;
}

// single-workgroup chained exclusive scan (n up to a few million)
__global__ __launch_bounds__(kScanThreads) void scan_u32_kernel(
    const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
    uint32_t* __restrict__ total_out, const uint32_t* __restrict__ in2,
    uint32_t* __restrict__ out2, const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  __shared__ uint32_t wave_sums[kScanThreads / 64];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr uint32_t kItems = 4;
  for (int pass = 0; pass < (in2 ? 2 : 1); ++pass) {
  if (pass == 1) { in = in2; out = out2; total_out = nullptr; }
  __syncthreads();
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t tile = 0; tile < n; tile += kScanThreads * kItems) {
    uint32_t v[kItems], local = 0;
    const uint32_t i0 = tile + tid * kItems;
#pragma unroll
    for (uint32_t k = 0; k < kItems; ++k) {
      v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
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
    uint32_t run = carry_s + wave_base + incl - local;
#pragma unroll
    for (uint32_t k = 0; k < kItems; ++k) {
      if (i0 + k < n) out[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == kScanThreads - 1) carry_s = run;
    __syncthreads();
  }
  if (tid == 0 && total_out) *total_out = carry_s;
  }
}

__device__ inline uint32_t slot_age(uint32_t epoch_new, uint32_t stamp) {
  const uint32_t a = epoch_new - stamp;
  return a < kAgeMax ? a : kAgeMax;
}

// Part A (rows): rank -> row table of the representatives that will be installed;
// representatives beyond the capacity give their claim back
// ("we only cache the first self.capacity", lru_cache.py:127-133).
// Part B (slots): level-1 histogram of slot ages (bits 21..11).
__global__ __launch_bounds__(kThreads) void lru_rank_hist_kernel(
    const int64_t* __restrict__ ids, uint32_t n, const uint32_t* __restrict__ rep_flag,
    const uint32_t* __restrict__ rep_rank, uint32_t* __restrict__ rep_row, int32_t* map,
    const uint32_t* __restrict__ stamp, uint32_t capacity, const uint32_t* __restrict__ epoch,
    uint32_t* __restrict__ hist_hi, const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  __shared__ uint32_t h[kBins];
  for (int b = threadIdx.x; b < kBins; b += kThreads) h[b] = 0;
  __syncthreads();
  const uint32_t k = min(ctr->n_unique, capacity);
  const uint32_t epoch_new = *epoch + 1;
  const uint32_t stride = gridDim.x * kThreads;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    if (!rep_flag[i]) continue;
    const uint32_t rank = rep_rank[i];
    if (rank < k) rep_row[rank] = i;
    else map[ids[i]] = kAbsent;
  }
  for (uint32_t s = blockIdx.x * kThreads + threadIdx.x; s < capacity; s += stride)
    atomicAdd(&h[slot_age(epoch_new, stamp[s]) >> 11], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kThreads)
    if (h[b]) atomicAdd(&hist_hi[b], h[b]);
}

// Finds the bin B (scanning from the oldest = highest bin) where the cumulative count
// reaches k; returns B and k_rem = k - (count in bins > B).  Called by EVERY thread of
// the workgroup (barriers inside); the first kThreads threads do the work and the
// result is broadcast through LDS.
__device__ inline void find_bin_from_top(const uint32_t* __restrict__ hist, uint32_t k,
                                         uint32_t* bin, uint32_t* k_rem) {
  __shared__ uint32_t part[kThreads];
  __shared__ uint32_t res[2];
  constexpr int kPer = kBins / kThreads;  // 8 bins per thread
  const int t = threadIdx.x;
  const bool worker = t < kThreads;
  // thread t owns bins [hi_first - kPer + 1, hi_first], hi_first descending with t
  const int hi_first = kBins - 1 - t * kPer;
  uint32_t mine[kPer], sum = 0;
  if (worker) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      mine[j] = hist[hi_first - j];
      sum += mine[j];
    }
    part[t] = sum;
  }
  if (t == 0) { res[0] = 0; res[1] = k; }
  __syncthreads();
  if (worker && k > 0) {
    uint32_t before = 0;  // exclusive prefix over threads (older bins first)
    for (int u = 0; u < t; ++u) before += part[u];
    if (before < k && before + sum >= k) {
      uint32_t acc = before;
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        if (acc + mine[j] >= k) {
          res[0] = hi_first - j;
          res[1] = k - acc;
          break;
        }
        acc += mine[j];
      }
    }
  }
  __syncthreads();
  *bin = res[0];
  *k_rem = res[1];
  __syncthreads();
}

// level-2 histogram (bits 10..0) of the slots whose level-1 bin is the boundary bin
__global__ __launch_bounds__(kThreads) void lru_hist_lo_kernel(
    const uint32_t* __restrict__ stamp, uint32_t capacity, const uint32_t* __restrict__ epoch,
    const uint32_t* __restrict__ hist_hi, uint32_t* __restrict__ hist_lo,
    const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  __shared__ uint32_t h[kBins];
  for (int b = threadIdx.x; b < kBins; b += kThreads) h[b] = 0;
  const uint32_t k = min(ctr->n_unique, capacity);
  uint32_t b_hi, k_rem;
  find_bin_from_top(hist_hi, k, &b_hi, &k_rem);
  const uint32_t epoch_new = *epoch + 1;
  const uint32_t stride = gridDim.x * kThreads;
  for (uint32_t s = blockIdx.x * kThreads + threadIdx.x; s < capacity; s += stride) {
    const uint32_t a = slot_age(epoch_new, stamp[s]);
    if ((a >> 11) == b_hi) atomicAdd(&h[a & (kBins - 1)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kThreads)
    if (h[b]) atomicAdd(&hist_lo[b], h[b]);
}

struct Threshold { uint32_t age; uint32_t k_tie; };

__device__ inline Threshold find_threshold(const uint32_t* hist_hi, const uint32_t* hist_lo,
                                           uint32_t k) {
  uint32_t b_hi, k_rem, b_lo, k_tie;
  find_bin_from_top(hist_hi, k, &b_hi, &k_rem);
  find_bin_from_top(hist_lo, k_rem, &b_lo, &k_tie);
  Threshold t;
  t.age = (b_hi << 11) | b_lo;
  t.k_tie = k_tie;
  return t;
}

// per tile of kTile slots: how many sit exactly at the threshold age, and how many are
// older than it (all of those are evicted)
__global__ __launch_bounds__(kThreads) void lru_tie_count_kernel(
    const uint32_t* __restrict__ stamp, uint32_t capacity, const uint32_t* __restrict__ epoch,
    const uint32_t* __restrict__ hist_hi, const uint32_t* __restrict__ hist_lo,
    uint32_t* __restrict__ tile_tie, uint32_t* __restrict__ tile_old,
    const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  __shared__ uint32_t cnt[2];
  const uint32_t k = min(ctr->n_unique, capacity);
  const Threshold th = find_threshold(hist_hi, hist_lo, k);
  const uint32_t epoch_new = *epoch + 1;
  const uint32_t tiles = (capacity + kTile - 1) / kTile;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c = 0, o = 0;
    for (uint32_t s = tile * kTile + threadIdx.x; s < min(capacity, (tile + 1) * kTile);
         s += kThreads) {
      const uint32_t a = slot_age(epoch_new, stamp[s]);
      c += a == th.age;
      o += a > th.age;
    }
    for (int d = 32; d > 0; d >>= 1) {
      c += __shfl_down(c, d, 64);
      o += __shfl_down(o, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (c) atomicAdd(&cnt[0], c);
      if (o) atomicAdd(&cnt[1], o);
    }
    __syncthreads();
    if (threadIdx.x == 0) { tile_tie[tile] = cnt[0]; tile_old[tile] = cnt[1]; }
    __syncthreads();
  }
}

// evict + install: every slot older than the threshold plus the first k_tie slots (in
// slot order) exactly at it (lru_cache.py:141-160 with a deterministic tie rule).  The
// i-th evicted slot in slot order receives the i-th distinct missed id in block order,
// so the whole update is deterministic (no atomics).
__global__ __launch_bounds__(kTile) void lru_install_kernel(
    const int64_t* __restrict__ ids, const uint32_t* __restrict__ rep_row, int32_t* map,
    int64_t* __restrict__ slot_id, uint32_t* __restrict__ stamp, uint32_t capacity,
    const uint32_t* __restrict__ epoch, const uint32_t* __restrict__ hist_hi,
    const uint32_t* __restrict__ hist_lo, const uint32_t* __restrict__ tie_base,
    const uint32_t* __restrict__ old_base, uint2* __restrict__ pairs, Counters* ctr) {
  if (ctr->n_miss == 0) return;
  __shared__ uint32_t wave_tie[kTile / 64];
  __shared__ uint32_t wave_old[kTile / 64];
  const uint32_t k = min(ctr->n_unique, capacity);
  const Threshold th = find_threshold(hist_hi, hist_lo, k);
  const uint32_t epoch_new = *epoch + 1;
  const uint32_t tiles = (capacity + kTile - 1) / kTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const uint32_t s = tile * kTile + threadIdx.x;
    const bool in = s < capacity;
    const uint32_t a = in ? slot_age(epoch_new, stamp[s]) : 0u;
    const bool tie = in && a == th.age;
    const bool older = in && a > th.age;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long mt = __ballot(tie), mo = __ballot(older);
    if (lane == 0) { wave_tie[wave] = __popcll(mt); wave_old[wave] = __popcll(mo); }
    __syncthreads();
    uint32_t ties_before = tie_base[tile] + __popcll(mt & below);
    uint32_t old_before = old_base[tile] + __popcll(mo & below);
    for (int w = 0; w < wave; ++w) { ties_before += wave_tie[w]; old_before += wave_old[w]; }
    const bool evict = k > 0 && (older || (tie && ties_before < th.k_tie));
    if (evict) {
      const uint32_t v = old_before + min(ties_before, th.k_tie);  // rank in slot order
      if (v < k) {
        const uint32_t row = rep_row[v];
        const int64_t nid = ids[row];
        const int64_t old = slot_id[s];
        if (old >= 0) map[old] = kAbsent;
        slot_id[s] = nid;
        map[nid] = static_cast<int32_t>(s);
        stamp[s] = epoch_new;
        pairs[v] = make_uint2(s, row);
      }
    }
    __syncthreads();
  }
}

// copy the installed rows out[row,:] -> cache_buffer[slot,:]; publish the new epoch
template <typename VecT>
__global__ __launch_bounds__(kThreads) void lru_copy_rows_kernel(
    const uint2* __restrict__ pairs, const VecT* __restrict__ out, VecT* __restrict__ cache_buf,
    uint32_t dimv, uint32_t capacity, uint32_t* epoch, const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  const uint32_t k = min(ctr->n_unique, capacity);
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  const uint32_t num_waves = (gridDim.x * kThreads) >> 6;
  for (uint32_t v = wave; v < k; v += num_waves) {
    const uint2 p = pairs[v];
    const VecT* src = out + static_cast<uint64_t>(p.y) * dimv;
    VecT* dst = cache_buf + static_cast<uint64_t>(p.x) * dimv;
    for (uint32_t c = lane; c < dimv; c += 64) dst[c] = src[c];
  }
}

__global__ void lru_bump_epoch_kernel(uint32_t* epoch, const Counters* __restrict__ ctr) {
  if (ctr->n_miss == 0) return;
  *epoch += 1;
}

__global__ void cache_fill_kernel(int32_t* map, uint64_t num_ids, int64_t* slot_id,
                                  uint32_t* stamp, uint64_t capacity, int identity) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_ids;
       i += stride)
    map[i] = (identity && i < capacity) ? static_cast<int32_t>(i) : kAbsent;
  for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < capacity;
       s += stride) {
    slot_id[s] = identity ? static_cast<int64_t>(s) : -1;
    stamp[s] = 0;
  }
}

inline bool vec4_ok(size_t dim, const void* a, const void* b, const void* c) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return dim % 4 == 0 && al(a) && al(b) && al(c);
}

inline unsigned gather_grid(size_t n) {
  // one wave per 64 rows, 4 waves per workgroup; enough workgroups to fill 256 CUs
  size_t waves = (n + 63) / 64;
  size_t blocks = (waves + 3) / 4;
  return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(blocks, 256 * 16)));
}

void launch_gather(const int64_t* ids, size_t n, const int32_t* map, const float* cache_buf,
                   const float* feats, size_t num_ids, size_t dim, float* out,
                   int32_t* slot_of_row, Counters* ctr, uint32_t* stats, hipStream_t stream) {
  GF_REQUIRE(n < 0x7FFFFFFFull, "gather: more than 2^31-1 rows in one block");
  ProfileScope ps(kProfGather, stream);
  const unsigned grid = gather_grid(n);
  if (vec4_ok(dim, cache_buf, feats, out)) {
    gather_rows_kernel<float4><<<dim3(grid), dim3(kThreads), 0, stream>>>(
        ids, static_cast<uint32_t>(n), map, reinterpret_cast<const float4*>(cache_buf),
        reinterpret_cast<const float4*>(feats), num_ids, static_cast<uint32_t>(dim / 4),
        reinterpret_cast<float4*>(out), slot_of_row, ctr, stats);
  } else {
    gather_rows_kernel<float><<<dim3(grid), dim3(kThreads), 0, stream>>>(
        ids, static_cast<uint32_t>(n), map, cache_buf, feats, num_ids,
        static_cast<uint32_t>(dim), out, slot_of_row, ctr, stats);
  }
  GF_HIP(hipGetLastError());
}

}  // namespace

void gather_rows(const float* d_feats, size_t num_rows, size_t dim, const int64_t* d_ids,
                 size_t n, float* d_out, int device, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_feats && d_ids && d_out, "gather_rows: null pointer");
  GF_REQUIRE(dim > 0, "gather_rows: dim must be positive");
  DeviceGuard dg(device);
  launch_gather(d_ids, n, nullptr, nullptr, d_feats, num_rows, dim, d_out, nullptr, nullptr,
                nullptr, stream);
}

FeatureCache::FeatureCache(size_t num_ids, size_t capacity, size_t dim, const float* d_feats,
                           int device)
    : num_ids_(num_ids), capacity_(capacity), dim_(dim), feats_(d_feats), device_(device) {
  GF_REQUIRE(dim > 0, "cache: dim must be positive");
  GF_REQUIRE(d_feats != nullptr, "cache: null feature table");
  GF_REQUIRE(capacity <= num_ids, "cache: capacity larger than the id space");
  GF_REQUIRE(capacity < 0x7FFFFFFFull, "cache: capacity must be < 2^31");
  DeviceGuard dg(device_);
  buffer_.reserve(std::max<size_t>(capacity * dim * sizeof(float), 16), 0, nullptr, true);
  map_.reserve(std::max<size_t>(num_ids * sizeof(int32_t), 16));
  slot_id_.reserve(std::max<size_t>(capacity * sizeof(int64_t), 16));
  stamp_.reserve(std::max<size_t>(capacity * sizeof(uint32_t), 16));
  state_.reserve(16, 0, nullptr, true);
  cache_fill_kernel<<<dim3(1024), dim3(256), 0, nullptr>>>(
      map_.as<int32_t>(), num_ids_, slot_id_.as<int64_t>(), stamp_.as<uint32_t>(), capacity_, 0);
  GF_HIP(hipGetLastError());
  GF_HIP(hipMemsetAsync(state_.data(), 0, 16, nullptr));
  GF_HIP(hipMemsetAsync(buffer_.data(), 0, buffer_.bytes(), nullptr));
  GF_HIP(hipStreamSynchronize(nullptr));
}

// Cache.init_cache (cache.py:175-195) / LRUCache.reset (lru_cache.py:91-105)
void FeatureCache::init(hipStream_t stream) {
  DeviceGuard dg(device_);
  cache_fill_kernel<<<dim3(1024), dim3(256), 0, stream>>>(
      map_.as<int32_t>(), num_ids_, slot_id_.as<int64_t>(), stamp_.as<uint32_t>(), capacity_, 1);
  GF_HIP(hipGetLastError());
  GF_HIP(hipMemsetAsync(state_.data(), 0, 16, stream));
  if (capacity_)
    GF_HIP(hipMemcpyAsync(buffer_.data(), feats_, capacity_ * dim_ * sizeof(float),
                          hipMemcpyDefault, stream));
}

// Cache.resize (cache.py:197-221): grow the id space / capacity, keep the contents
void FeatureCache::resize(size_t new_num_ids, size_t new_capacity, const float* d_feats,
                          hipStream_t stream) {
  GF_REQUIRE(new_num_ids >= num_ids_ && new_capacity >= capacity_,
             "cache: resize can only grow");
  GF_REQUIRE(new_capacity <= new_num_ids && new_capacity < 0x7FFFFFFFull,
             "cache: invalid capacity");
  DeviceGuard dg(device_);
  if (d_feats) feats_ = d_feats;
  if (new_num_ids > num_ids_) {
    DeviceBuffer nmap;
    nmap.reserve(new_num_ids * sizeof(int32_t));
    // new ids start uncached
    std::vector<int32_t> tail(new_num_ids - num_ids_, kAbsent);
    GF_HIP(hipMemcpyAsync(nmap.data(), map_.data(), num_ids_ * sizeof(int32_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipMemcpyAsync(nmap.as<int32_t>() + num_ids_, tail.data(),
                          tail.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    GF_HIP(hipStreamSynchronize(stream));
    std::swap(map_, nmap);
  }
  if (new_capacity > capacity_) {
    buffer_.reserve(new_capacity * dim_ * sizeof(float), capacity_ * dim_ * sizeof(float),
                    stream, true);
    DeviceBuffer nid, nst;
    nid.reserve(new_capacity * sizeof(int64_t));
    nst.reserve(new_capacity * sizeof(uint32_t));
    std::vector<int64_t> empty_ids(new_capacity - capacity_, -1);
    GF_HIP(hipMemcpyAsync(nid.data(), slot_id_.data(), capacity_ * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipMemcpyAsync(nid.as<int64_t>() + capacity_, empty_ids.data(),
                          empty_ids.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    GF_HIP(hipMemsetAsync(nst.data(), 0, new_capacity * sizeof(uint32_t), stream));
    GF_HIP(hipMemcpyAsync(nst.data(), stamp_.data(), capacity_ * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipStreamSynchronize(stream));
    std::swap(slot_id_, nid);
    std::swap(stamp_, nst);
  }
  num_ids_ = new_num_ids;
  capacity_ = new_capacity;
}

void FeatureCache::reserve_workspace(size_t n) {
  if (n <= ws_rows_ && ws_.data()) return;
  ws_rows_ = std::max(ws_rows_, n);
  const size_t tiles = (capacity_ + kTile - 1) / kTile + 1;
  size_t bytes = align_up(sizeof(Counters), 16) + 2 * kBins * sizeof(uint32_t) +
                 4 * align_up(ws_rows_ * 4, 16) + align_up(ws_rows_ * 8, 16) +
                 4 * align_up(tiles * 4, 16) + 64;
  ws_.reserve(bytes, 0, nullptr);
}

// One block of Cache.fetch_feature (cache.py:269-323 / :326-400)
void FeatureCache::fetch(const int64_t* d_ids, size_t n, float* d_out, bool update,
                         uint32_t* d_stats, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_ids && d_out, "cache fetch: null pointer");
  DeviceGuard dg(device_);
  reserve_workspace(n);
  const size_t tiles = (capacity_ + kTile - 1) / kTile;
  Workspace w;
  char* p = ws_.as<char>();
  w.ctr = reinterpret_cast<Counters*>(p);           p += align_up(sizeof(Counters), 16);
  w.hist_hi = reinterpret_cast<uint32_t*>(p);       p += kBins * sizeof(uint32_t);
  w.hist_lo = reinterpret_cast<uint32_t*>(p);       p += kBins * sizeof(uint32_t);
  const size_t zero_bytes = p - ws_.as<char>();
  w.slot_of_row = reinterpret_cast<int32_t*>(p);    p += align_up(ws_rows_ * 4, 16);
  w.rep_flag = reinterpret_cast<uint32_t*>(p);      p += align_up(ws_rows_ * 4, 16);
  w.rep_rank = reinterpret_cast<uint32_t*>(p);      p += align_up(ws_rows_ * 4, 16);
  w.rep_row = reinterpret_cast<uint32_t*>(p);       p += align_up(ws_rows_ * 4, 16);
  w.pairs = reinterpret_cast<uint2*>(p);            p += align_up(ws_rows_ * 8, 16);
  w.tile_tie = reinterpret_cast<uint32_t*>(p);      p += align_up((tiles + 1) * 4, 16);
  w.tile_old = reinterpret_cast<uint32_t*>(p);      p += align_up((tiles + 1) * 4, 16);
  w.tie_base = reinterpret_cast<uint32_t*>(p);      p += align_up((tiles + 1) * 4, 16);
  w.old_base = reinterpret_cast<uint32_t*>(p);

  GF_HIP(hipMemsetAsync(ws_.data(), 0, zero_bytes, stream));
  launch_gather(d_ids, n, capacity_ ? map_.as<int32_t>() : nullptr, buffer_.as<float>(), feats_,
                num_ids_, dim_, d_out, w.slot_of_row, w.ctr, d_stats, stream);
  if (!update || capacity_ == 0) return;

  ProfileScope ps(kProfLru, stream);
  const uint32_t n32 = static_cast<uint32_t>(n), cap32 = static_cast<uint32_t>(capacity_);
  uint32_t* epoch = state_.as<uint32_t>();
  const unsigned row_grid = static_cast<unsigned>((n + kThreads - 1) / kThreads);
  const unsigned slot_grid = static_cast<unsigned>(
      std::max<size_t>(1, std::min<size_t>((std::max(n, capacity_) + kThreads - 1) / kThreads, 2048)));
  const unsigned tile_grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(tiles, 2048)));
  lru_claim_kernel<<<dim3(row_grid), dim3(kThreads), 0, stream>>>(d_ids, n32, w.slot_of_row,
                                                                  map_.as<int32_t>(), w.ctr);
  lru_mark_kernel<<<dim3(row_grid), dim3(kThreads), 0, stream>>>(
      d_ids, n32, w.slot_of_row, map_.as<int32_t>(), stamp_.as<uint32_t>(), epoch, w.rep_flag,
      w.ctr);
  scan_u32_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
      w.rep_flag, w.rep_rank, n32, &w.ctr->n_unique, nullptr, nullptr, w.ctr);
  lru_rank_hist_kernel<<<dim3(slot_grid), dim3(kThreads), 0, stream>>>(
      d_ids, n32, w.rep_flag, w.rep_rank, w.rep_row, map_.as<int32_t>(), stamp_.as<uint32_t>(),
      cap32, epoch, w.hist_hi, w.ctr);
  lru_hist_lo_kernel<<<dim3(slot_grid), dim3(kThreads), 0, stream>>>(
      stamp_.as<uint32_t>(), cap32, epoch, w.hist_hi, w.hist_lo, w.ctr);
  lru_tie_count_kernel<<<dim3(tile_grid), dim3(kThreads), 0, stream>>>(
      stamp_.as<uint32_t>(), cap32, epoch, w.hist_hi, w.hist_lo, w.tile_tie, w.tile_old, w.ctr);
  scan_u32_kernel<<<dim3(1), dim3(kScanThreads), 0, stream>>>(
      w.tile_tie, w.tie_base, static_cast<uint32_t>(tiles), nullptr, w.tile_old, w.old_base,
      w.ctr);
  lru_install_kernel<<<dim3(tile_grid), dim3(kTile), 0, stream>>>(
      d_ids, w.rep_row, map_.as<int32_t>(), slot_id_.as<int64_t>(), stamp_.as<uint32_t>(), cap32,
      epoch, w.hist_hi, w.hist_lo, w.tie_base, w.old_base, w.pairs, w.ctr);
  const unsigned copy_grid = static_cast<unsigned>(
      std::max<size_t>(1, std::min<size_t>((std::min(n, capacity_) + 3) / 4, 4096)));
  if (vec4_ok(dim_, buffer_.data(), d_out, d_out)) {
    lru_copy_rows_kernel<float4><<<dim3(copy_grid), dim3(kThreads), 0, stream>>>(
        w.pairs, reinterpret_cast<const float4*>(d_out), buffer_.as<float4>(),
        static_cast<uint32_t>(dim_ / 4), cap32, epoch, w.ctr);
  } else {
    lru_copy_rows_kernel<float><<<dim3(copy_grid), dim3(kThreads), 0, stream>>>(
        w.pairs, d_out, buffer_.as<float>(), static_cast<uint32_t>(dim_), cap32, epoch, w.ctr);
  }
  lru_bump_epoch_kernel<<<dim3(1), dim3(1), 0, stream>>>(epoch, w.ctr);
  GF_HIP(hipGetLastError());
}

void FeatureCache::slot_ids(int64_t* out, size_t capacity) const {
  GF_REQUIRE(out != nullptr && capacity >= capacity_, "slot_ids: output too small");
  if (!capacity_) return;
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  GF_HIP(hipMemcpy(out, slot_id_.data(), capacity_ * sizeof(int64_t), hipMemcpyDeviceToHost));
}

size_t FeatureCache::mem_bytes() const {
  return capacity_ * dim_ * sizeof(float) + num_ids_ * sizeof(int32_t) +
         capacity_ * (sizeof(int64_t) + sizeof(uint32_t));
}

}  // namespace gf
