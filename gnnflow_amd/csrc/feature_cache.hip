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
// on every block.  Here a *round* of up to three independent blocks (one node-cache
// block, one edge-cache block, one cache-free gather) is 1 + 4 launches, none of which
// waits for the host; each kernel takes the round's contexts by value and blockIdx.y
// selects the context, so the node and edge caches advance in the same launches:
//
//   gather   : reads ids, probes the id->slot map, picks the source row (cache slot in
//              HBM, or the feature table — HBM or device-mapped pinned host memory) and
//              streams it to the output with 16-byte loads/stores.  A wave owns
//              `tile_rows` consecutive output rows, flattened, so its stores form one
//              contiguous run and every lane keeps 4 independent 16 B loads in flight;
//              tile_rows shrinks for small blocks so that a 10k-row block still spreads
//              over >2000 waves.  It also records each row's slot, marks hit slots as
//              touched in this epoch, counts hits/misses (once per workgroup, sharded) and
//              lets every missed row claim its id with atomicMax(map[id], -(row+1)) — the
//              lowest row of each distinct missed id wins (this replaces torch.unique).
//              This kernel moves ~all the bytes (2*dim*4 per row) and is the one priced
//              against the HBM roofline.
//   scan+h1  : workgroup 0 prefix-sums the representative flags (rank in first-seen order,
//              #unique); the other workgroups histogram the slot ages.
//   rank+h2  : rank -> row table of the ids to install (the rest give their claim back);
//              second-level histogram only when the eviction threshold is older than 2047
//              epochs.
//   count    : per 1024-slot tile, slots older than / exactly at the threshold age.
//   install  : evicts every older slot plus the first k_tie threshold-age slots in slot
//              order, gives the i-th evicted slot (slot order) the i-th distinct missed id
//              (block order), copies the freshly gathered rows from the output (already in
//              HBM) into the cache and turns this epoch's touch marks into stamps.
//              No atomics: fully deterministic.
//
// The scan / rank / count / install chain above serves LFU (and, without the histogram
// half, FIFO): `stamp` holds the use count and the k smallest are found with a histogram
// select (no sort, no topk); ties go to the lowest slot index.
//
// LRU — the policy on the hot path — needs no selection at all: the reference's `count`
// only ever changes to "newest" (hit or install: count = 0 while all others sink by one,
// lru_cache.py:134-160), so the eviction order is a LIST, least recently refreshed slot
// first, that every update permutes in the same simple way: the slots hit by the block move
// behind the others, the first k = #distinct misses entries are the victims and go, refilled,
// to the very back.  `queue` holds that list (a permutation of the slots, double-buffered);
// no stamps, no histogram, no threshold, no atomics, and ties are resolved STABLY — slots of
// equal `count` keep their relative order, what a stable sort by `count` yields (the
// reference leaves it to torch.topk's unspecified tie-breaking).  Two launches per round:
//   list scan   : row-tile workgroups rank the representatives of the distinct missed ids;
//                 list-tile workgroups count the hit slots per tile of the list; one more
//                 workgroup reads the victims off the front of the list.
//   list install: one thread per block row installs the m-th missed id in the m-th victim's
//                 slot (map / slot_id / row copy from the freshly gathered output — spread
//                 over as many workgroups as the block has rows); list-tile workgroups write
//                 the permuted list into the other buffer.
// All bookkeeping kernels return at once for a context whose block had no miss (the
// reference skips update_*_cache then too, cache.py:318); a hit only changes replacement
// state if the block also had a miss, exactly as in the reference.
#include "feature_cache.hpp"
#include "owner_hash.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <chrono>
#include <climits>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace gf {

namespace {

constexpr int32_t kAbsent = INT32_MIN;  // map[] value of an uncached id
// gather workgroup (same-box A/B of the headline fetch, us per launch: 64 threads 23.5, 128 16.1,
// 256 13.3, 512 13.9, 1024 15.3 — profiles/r06_gather_hop_trace.txt)
constexpr int kThreads = 256;
constexpr int kWide = 1024;             // slot kernels, scans
constexpr int kFine = 2048;             // ages 0..2047: one bin each
constexpr int kBins1 = 4096;            // + 2048 coarse bins of 2048 ages each
constexpr int kBins2 = 2048;            // second level inside one coarse bin
constexpr uint32_t kAgeMax = kFine + 2048u * 2048u - 1u;
constexpr int kTile = 1024;             // slots per tile (= install workgroup)
constexpr int kRing = 32;               // per-fetch counter records
constexpr int kMaxCtx = 4;              // contexts per round
constexpr uint32_t kRowTile = 4096;     // rows per scan workgroup (kWide threads x 4)
constexpr uint32_t kMaxRowTiles = 1024; // more row tiles than this: chained single-workgroup scan
constexpr uint32_t kQGroup = 64;        // LRU: list tiles per group sum
constexpr uint32_t kInstRows = 256;     // LRU: block rows per install workgroup
// LRU: block rows per scan workgroup.  One row per thread: the row role is a handful of
// scattered loads per row, which a CU retires at ~one 64-line instruction per 64 cycles, so a
// 20 k-row block wants 20 CUs on it, not 5 (kRowTile rows per workgroup).
constexpr uint32_t kLruRows = 1024;
constexpr uint32_t kMaxStageTiles = 1024;   // LRU list form: list tiles that stage their victims
constexpr uint32_t kBitTile = 4096;         // LRU queue form: words of the hit bitmap per tile (kWide x 4)
constexpr uint32_t kMaxBitGroups = 1024;    // ... entries of the install kernel's LDS prefix over the tiles

// One record per fetch.  hits / misses are accumulated once per workgroup into one of 8
// shards that sit on separate 128-byte lines: same-address atomics retire at only
// ~88/us on MI355X, so one counter word per wave would dominate the gather itself.
constexpr int kShards = 8;
struct Shard {
  uint32_t hits;      // rows served from the cache
  uint32_t n_miss;    // rows served from the feature table
  uint32_t pad[30];
};
struct Counters {
  Shard shard[kShards];
  uint32_t n_unique;  // distinct missed ids
  uint32_t th_age;    // eviction threshold (written by the tile-count kernel)
  uint32_t th_k_tie;
  uint32_t fifo_start;  // FIFO: first slot of this block's refill arc
  uint32_t ticket;      // workgroups of the rank kernel that finished their level-2 histogram
  uint32_t q_parity;    // LRU list: buffer that is current during this update
  uint32_t q_found;     // LRU list: not-hit victims found by the list scan
  uint32_t q_head;      // LRU queue: head / tail of the queue during this update
  uint32_t q_tail;
  uint32_t pad[23];
};
constexpr uint32_t kCounterWords = sizeof(Counters) / 4;

// Everything one block fetch needs on the device.  `update` == 0: gather only.
// {parity, flip_tag} form ONE aligned 64-bit word: the fused list update flips the parity with
// a single store of {new parity, its launch tag}, so a workgroup of the same launch that starts
// late and reads the word knows from the tag that it already sees the NEW parity.
struct QueueState { uint32_t parity, flip_tag, head, tail, lone_walks, pad; };

struct Ctx {
  const int64_t* ids;
  uint32_t n;
  int vec4;                 // rows are float4-addressable
  uint32_t dimv;            // row length in float4s (vec4 / odd4) or floats
  // rows of dim % 4 != 0 floats (GDELT: 413 / 186) or misaligned bases: dimv = ceil(dim / 4)
  // 16-byte vectors at 4-byte alignment per row, the last one ending with the row (it overlaps
  // its neighbour); rows `dim` floats apart
  int odd4;
  uint32_t dim, tail;
  // the feature table is in HBM: the install kernel copies a missed row into the cache from
  // the table (read by the gather a moment ago: in L2) rather than from the streamed output
  int inst_from_table;
  uint32_t tile_rows;       // rows per wave in the gather
  uint32_t inflight;        // 16-byte loads a lane keeps in flight while copying a tile
  float* out;
  const float* feats;
  // sharded feature tables (Cache(distributed=True)): a missed row i is read from row
  // miss_index[i] of miss_rows — the rows the caller pulled from their owners — not from feats
  const float* miss_rows;
  const uint32_t* miss_index;
  // ... or, when the pull was planned natively (gf_pull_*): from row req_pos[rep] of miss_rows,
  // rep = the row whose claim on map[id] the plan settled (cache-free context: the row itself)
  const uint32_t* req_pos;
  // serving a shard: the table row of id is remap[id] (global id -> local row, < 0: not owned
  // -> *flag is raised and row 0 is served)
  const int32_t* remap;
  uint32_t* flag;
  // host-resident table with a staging ring ("staging ring" below): a missed id whose pmap entry
  // {generation, row} lies in [st_lo, st_lo + st_span] is read from that row of the generation's
  // region of the ring — an HBM copy of its table row pulled ahead of this launch
  const unsigned long long* pmap;
  const float* ring;
  uint32_t st_lo, st_span, st_mask, st_cap;
  uint32_t* progress;       // pinned host word: this launch stores progress_val = the number of
  uint32_t progress_val;    // ring-reading launches enqueued before it (all finished by now)
  unsigned long long* st_fallback;   // rows this cache's gathers read from the HOST table
  // diagnostics (gf_debug_lru_trace): per workgroup of the one-launch list update, 8 stamps of the
  // 100 MHz wall clock; [0 .. 3] of the buffer: count / row / write workgroups, launch tag
  unsigned long long* trace;
  uint64_t num_ids;
  int32_t* map;             // null: no cache (plain gather)
  float* cache_buf;
  int64_t* slot_id;
  uint32_t* stamp;          // LFU: use count (FIFO: install epoch; LRU: unused)
  uint32_t* touched;        // epoch of the last hit (pending until the block misses) — LRU list
                            // form: indexed by the entry's LIST POSITION (qpos[slot]), so the
                            // two list passes read it densely, next to the list itself;
                            // otherwise (queue form, LFU) by slot
  uint32_t* queue[2];       // LRU: the slots, least recently refreshed first (double buffer)
  QueueState* qstate;       // LRU: which buffer is current, device resident
  uint32_t tiles_per_wg;    // LRU: row tiles per scan workgroup (1 unless > 1M rows)
  uint32_t inst_rows;       // LRU: block rows per install workgroup (kInstRows or kWide)
  // LRU of a LARGE cache (queue form, see "LRU as a queue" below); qmode == 0: list form
  int qmode;                // this update appends to the queue instead of rewriting the list
  uint32_t* qpos;           // [capacity] position of the slot's live queue entry
  uint32_t* qbits;          // one bit per queue position: entry of a slot hit by this block
                            // (set by the gather; all zero between updates)
  uint2* wsnap;             // per word of qbits: {the word, hit entries before it in its tile}
  uint32_t q_group;         // bitmap tiles per entry of the install kernel's LDS prefix
  // list form: the first stage_tiles list tiles leave their not-hit entries (the victims, in
  // list order) packed per tile in v_slot and — if that is the whole list — their hit entries
  // in v_pos (the next victims when a block needs more slots than its hits leave over);
  // 0: one workgroup walks the list instead (more than kMaxStageTiles tiles needed)
  uint32_t stage_tiles;
  uint32_t stage_min;       // ... for blocks that missed more rows than this
  int stage_hits;
  uint32_t v_chunks;        // victim walk: chunks of kRowTile queue entries behind the head
  uint32_t* v_slot;         // [(v_chunks * kRowTile) + n] candidates per chunk (+ the lone walk's)
  uint32_t* v_pos;          // their queue positions
  uint32_t* v_count;        // [v_chunks + 1]
  // LRU list form, ONE launch (lru_list_fused_kernel): granules {launch tag, count} per list
  // tile / per row workgroup, and the front tiles' entries staged with the id they hold
  int fused;
  uint32_t fuse_tag;        // unique per launch and cache (never reset), > 0
  uint32_t fuse_rows;       // block rows per row workgroup (kInstRows or kWide)
  unsigned long long* g_cnt;   // [kFuseMaxTiles]
  unsigned long long* g_row;   // [kFuseMaxRowWgs]
  long long* v_old;         // id held by v_slot's entry
  long long* v_hold;        // ... by v_pos's (the hit entries)
  uint32_t capacity;
  uint32_t epoch_new;
  int update;
  int policy;               // GF_CACHE_LRU / _LFU / _FIFO
  uint32_t* fifo_ptr;       // FIFO: last refilled slot (fifo_cache.py:66-69), device resident
  int32_t* slot_of_row;
  uint32_t* rep_flag;
  uint32_t* rep_rank;
  uint32_t* rep_row;        // rank -> row of the representative
  int64_t* rep_id;          // rank -> id (saves the install kernel a dependent load)
  uint32_t* row_tile_sum;   // [ceil(n / kRowTile)] representatives per row tile
                            // (LRU: per scan workgroup)
  uint32_t* hist1;
  uint32_t* hist2;
  uint32_t* tile_tie;
  uint32_t* tile_old;
  Counters* ctr;            // this fetch's record (zeroed by the previous fetch)
  Counters* ctr_next;       // record of the next fetch on this cache: zeroed here
  uint32_t* stats;          // caller's 16-word hit statistics, may be null
};
struct Round {
  Ctx c[kMaxCtx];
  int count;
};

__device__ inline uint32_t total_miss(const Counters* c) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < kShards; ++i) m += c->shard[i].n_miss;
  return m;
}

// LRU list form: a block that missed more rows than this finds its victims through the list
// tiles' staged entries; fewer (one or two trips) are cheaper for the one-workgroup walk
// (headline workload, ~5 k missed rows per block: 39.6-40.0 us per step with the walk,
// 40.4-41.0 staged; 30 k-row blocks with 15-30 k misses: 37.7 us per fetch with the walk, 29.4
// staged).  The same counter is read by both kernels, so they agree.
constexpr uint32_t kStageMinWant = 8192;
__device__ inline bool use_staged_victims(uint32_t stage_tiles, uint32_t missed_rows,
                                          uint32_t min_want) {
  return stage_tiles != 0 && missed_rows > min_want;
}

// a float4 that is only 4-byte aligned: global memory takes unaligned 16-byte accesses, a
// wave's 1 KB run then touches 9 lines instead of 8
typedef float uf4 __attribute__((ext_vector_type(4), aligned(4)));

typedef float nf4 __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float v, float* p) { __builtin_nontemporal_store(v, p); }
__device__ inline void nt_store(const float4& v, float4* p) {
  nf4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(p));
}

// loads through a pointer known to be global memory (global_load_*, not flat_load_*)
template <typename VecT> __device__ inline VecT global_load(const void* p);
template <> __device__ inline float global_load<float>(const void* p) {
  return *(const __attribute__((address_space(1))) float*)p;
}
template <> __device__ inline uf4 global_load<uf4>(const void* p) {
  return *(const __attribute__((address_space(1))) uf4*)p;
}
template <> __device__ inline float4 global_load<float4>(const void* p) {
  const nf4 t = *(const __attribute__((address_space(1))) nf4*)p;
  return make_float4(t.x, t.y, t.z, t.w);
}

template <typename VecT> __device__ inline VecT vec_zero();
template <> __device__ inline uf4 vec_zero<uf4>() { return uf4{0.f, 0.f, 0.f, 0.f}; }
// (streaming, like the float4 rows: GDELT-shaped step 257 -> 233 us of gather per step; writing
// a tile's contiguous output as ALIGNED float4s instead changed nothing on top of that — the
// rest of the gap to 16-byte-aligned row widths, 212 us, is on the load side)
__device__ inline void nt_store(const uf4& v, uf4* p) { __builtin_nontemporal_store(v, p); }
template <> __device__ inline float vec_zero<float>() { return 0.0f; }
template <> __device__ inline float4 vec_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// rep_flag[row]: representative of a distinct missed id (rank among them in the low bits) /
// queue form: THE row that stands for a hit slot (the old queue position of its entry)
constexpr uint32_t kRepMiss = 1u << 31, kRepRank = kRepMiss - 1u;
constexpr uint32_t kRepHit = 1u << 30, kRepPos = kRepHit - 1u;

// gf_debug_lru_trace buffer: [0 .. 3] header, 8 stamps per workgroup of the one-launch update
// (at most 2 * 2048 list tiles + 1024 row workgroups), then 8 per workgroup of the gather launch
// before it (the traced cache's context)
constexpr uint32_t kGatherTraceBase = 4u + 8u * (2u * 2048u + 1024u);
constexpr uint32_t kGatherTraceWgs = 1024u;

// ---- the gather kernel -------------------------------------------------------------
// kLean: the instantiation for rounds of float4 rows on list-form / cache-free contexts with
// the default 12 loads in flight (no queue-form hit path, one copy loop)
template <typename VecT, bool kOdd = false, bool kLean = false, bool kStaged = !kLean,
          bool kDirect = false>
__device__ inline void gather_body(const Ctx& kc, uint32_t bx, uint32_t grid_x) {
  // The context lives in the kernel-argument segment and the compiler loads a field where it is
  // first used: seven dependent rounds of scalar loads (each a trip to memory for a CU's first
  // wave) stood before the first id was read.  Pinning the hot fields into SGPRs HERE makes them
  // one round.
  const Ctx& c = kc;
  // (input operands: the values must be in SGPRs here — their loads are issued together before
  // this point — and stay the kernel arguments they are, pointers into GLOBAL memory; as in/out
  // operands they came back as generic pointers and every access through them was a flat one)
  if (kLean)   // (the general kernel has no scalar registers to spare, and its launches are long)
    asm volatile("" :: "s"(c.ids), "s"(c.n), "s"(c.num_ids), "s"(c.map), "s"(c.feats), "s"(c.out),
                 "s"(c.dimv), "s"(c.tile_rows), "s"(c.update), "s"(c.policy), "s"(c.touched),
                 "s"(c.qpos), "s"(c.epoch_new), "s"(c.slot_of_row), "s"(grid_x), "s"(c.dim),
                 "s"(c.ctr), "s"(c.ctr_next), "s"(c.stats), "s"(c.tile_old), "s"(c.hist1),
                 "s"(c.capacity));
  if (kLean && !kDirect)
    asm volatile("" :: "s"(c.cache_buf), "s"(c.miss_rows), "s"(c.remap), "s"(c.pmap), "s"(c.ring),
                 "s"(c.st_lo), "s"(c.st_span), "s"(c.st_mask), "s"(c.st_cap));
  if (c.n == 0) return;
  // diagnostics (scripts/gather_hop_trace.py): wave 0 of every workgroup stamps the wall clock at
  // the stages of its first tile
  unsigned long long* tr = nullptr;
  if (kLean && kDirect && c.trace && threadIdx.x == 0 && bx < kGatherTraceWgs)
    tr = c.trace + kGatherTraceBase + bx * 8u;
  if (tr) {
    tr[0] = wall_clock64();
    // where it runs: HW_ID (cu 8-11, sh 12, se 13-15 on gfx9) and XCC_ID
    tr[5] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(4 | (31 << 11))) |
            (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (31 << 11))) << 32);
  }
  const int lane = threadIdx.x & 63;
  const uint32_t gtid = bx * kThreads + threadIdx.x;
  const uint32_t nthreads = grid_x * kThreads;
  // housekeeping for later launches: this fetch's histograms and the NEXT fetch's counter
  // record are cleared here (neither is in use by anyone else at this point)
  if (c.update && c.policy == GF_CACHE_LRU) {   // per-group hit counts of the list scan
    const uint32_t groups = ((c.capacity + kRowTile - 1) / kRowTile + kQGroup - 1) / kQGroup;
    for (uint32_t i = gtid; i < groups; i += nthreads) c.tile_old[i] = 0;
  } else if (c.update) {
    for (uint32_t i = gtid; i < kBins1 + kBins2; i += nthreads) c.hist1[i] = 0;  // hist2 follows
  }
  if (c.ctr_next) {
    uint32_t* nxt = reinterpret_cast<uint32_t*>(c.ctr_next);
    for (uint32_t i = gtid; i < kCounterWords; i += nthreads) nxt[i] = 0;
  }
  // row stride in units of VecT — or, for odd rows, in floats (rowu) with VecT at any float
  using Unit = std::conditional_t<kOdd, float, VecT>;
  constexpr uint32_t kVF = kOdd ? 4u : 1u;   // Units per VecT
  const Unit* feats = reinterpret_cast<const Unit*>(c.feats);
  const Unit* cache_buf = reinterpret_cast<const Unit*>(c.cache_buf);
  Unit* out = reinterpret_cast<Unit*>(c.out);
  const uint32_t dimv = c.dimv, tile_rows = c.tile_rows, n = c.n;
  const uint32_t rowu = kOdd ? c.dim : c.dimv;
  const uint32_t wave = gtid >> 6;
  const uint32_t num_waves = nthreads >> 6;
  const uint32_t tiles = (n + tile_rows - 1) / tile_rows;
  uint32_t acc_hits = 0, acc_miss = 0;   // wave-uniform
  uint32_t acc_host = 0;                 // rows read from the host table (staged contexts)
  // direct: every row comes from feats[id] whatever the probe says (table in HBM, no row mirror,
  // no pulled rows, no staging ring) — the probe then only feeds the counters and the marks
  // (a template parameter: the two orders in one instantiation cost 180 instead of 104 VGPRs)
  constexpr bool direct = kDirect;
  for (uint32_t tile = wave; tile < tiles; tile += num_waves) {
    const uint32_t row0 = tile * tile_rows;
    const uint32_t rows = min(tile_rows, n - row0);
    const Unit* src = nullptr;
    int32_t slot = -2;
    uint32_t hit_code = 0;
    bool from_host = false;   // staged context: the row is read from the host table after all
    int64_t id = -1;
    bool known = false;
    if (lane < static_cast<int>(rows)) {
      id = c.ids[row0 + lane];
      known = id >= 0 && static_cast<uint64_t>(id) < c.num_ids;
      if (known) {
        slot = c.map ? c.map[id] : -1;
        if (direct) src = feats + static_cast<uint64_t>(id) * rowu;
      }
    }
    if (kLean && kDirect && c.trace && tile == wave) {   // (wave-uniform; traced launches only)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tr) tr[1] = wall_clock64();   // ids (and the map values) in
    }
    uint64_t src_bits = reinterpret_cast<uint64_t>(src);
    const uint32_t total = rows * dimv;
    Unit* o = out + static_cast<uint64_t>(row0) * rowu;
    // The loop trip count is wave-uniform and every lane executes the cross-lane read:
    // ds_bpermute returns 0 for a source lane that EXEC has switched off, so the
    // row-base broadcast must never sit under a per-lane condition.
    auto load = [&](uint32_t fu, bool* valid, uint32_t* at) -> VecT {
      *valid = fu < total;
      const uint32_t r = *valid ? fu / dimv : 0u;
      const uint32_t cc = fu - r * dimv;
      // odd rows: the last vector of a row ends with the row (it overlaps its neighbour by
      // 4 - dim % 4 floats, which are simply written twice) — no scalar tail pass
      const uint32_t off = kOdd ? min(cc * kVF, rowu - kVF) : cc * kVF;
      // where it goes, in Units from the tile's first row (even rows: r * dimv + cc = fu)
      *at = kOdd ? r * rowu + off : fu;
      const Unit* s = reinterpret_cast<const Unit*>(__shfl(src_bits, r, 64));
      // An unconditional GLOBAL load (a lane with nothing to read reads the tile's first output
      // row): a load under a per-lane branch, or a flat one — the pointer went through a
      // cross-lane read and lost its address space — makes the compiler wait for ALL loads in
      // flight (s_waitcnt vmcnt(0)) wherever it needs one of them.
      const bool take = *valid && s != nullptr;
      const VecT x = global_load<VecT>(take ? s + off : o);
      return take ? x : vec_zero<VecT>();
    };
    // direct context: the first trip's row loads are issued here, right behind the map load and
    // before anything looks at its result — the chain is launch -> ids -> rows -> stores, the
    // probe (map -> marks / claims, which only the update reads) hangs off its side
    constexpr int K0 = kDirect ? 13 : 12;   // (13: a tile of 19 172-d rows still is one trip)
    VecT v0[K0];
    bool p0[K0];
    uint32_t at0[K0];
    constexpr bool early = direct;   // (a direct body runs with 12 loads in flight: the callers)
    // K independent 16-byte loads in flight per lane, then the stores.  (12 covers a whole
    // 16-row tile of 172-d rows in one trip; measured 14.8-14.9 us per launch against 15.5-15.7
    // with 4 on the same box.)
    auto copy = [&](auto kk, uint32_t first) {
      constexpr int K = decltype(kk)::value;
      for (uint32_t base = first; base < total; base += 64 * K) {
        VecT v[K];
        bool p[K];
        uint32_t at[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = load(base + lane + 64 * k, &p[k], &at[k]);
        // streaming stores: the 21 MB of output rows of a launch would otherwise sit dirty in
        // the L2s until the kernel's end-of-kernel write-back (14.1 -> 13.2 us per launch; the
        // install kernel, which reads the missed rows back, pays 0.5-1 us of that again;
        // storing only the hit rows this way was slower than either)
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (p[k]) nt_store(v[k], reinterpret_cast<VecT*>(o + at[k]));
      }
    };
    if (early) {
      // ... and stored as they arrive; the probe's result is looked at behind the copy
#pragma unroll
      for (int k = 0; k < K0; ++k) v0[k] = load(lane + 64 * k, &p0[k], &at0[k]);
#pragma unroll
      for (int k = 0; k < K0; ++k)
        if (p0[k]) nt_store(v0[k], reinterpret_cast<VecT*>(o + at0[k]));
      // (rows wider than one trip covers are rare and this loop's registers count for the whole
      // kernel: four in flight keeps it at 4 waves per SIMD)
      copy(std::integral_constant<int, 4>{}, 64u * K0);
      if (kLean && c.trace && tile == wave) {
        if (tr) tr[2] = wall_clock64();   // every row of the tile in, its stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tr) tr[3] = wall_clock64();   // stores acknowledged
      }
    }
    if (known) {
      const int32_t claim = slot;   // a missed id of a planned pull: -(representative row + 1)
      if (slot >= 0) {
        // (no row mirror — table in HBM, gf_cache_set_row_mirror: a hit is the table's row too)
        if (!direct)
          src = c.cache_buf ? cache_buf + static_cast<uint64_t>(slot) * rowu
                            : feats + static_cast<uint64_t>(id) * rowu;
        // a hit is recorded (LRU: refreshes the slot, LFU: counts a use) but takes effect
        // only if the block also misses; FIFO ignores hits (fifo_cache.py:77-161).
        if (!kLean && c.qmode) {
          // queue form: the mark is the bit of the entry's queue position; the row whose
          // atomic set it stands for the slot (it will append the slot's new entry)
          const uint32_t pos = c.qpos[slot], bit = 1u << (pos & 31u);
          const uint32_t was = atomicOr(&c.qbits[pos >> 5], bit);
          hit_code = (was & bit) ? 0u : (kRepHit | pos);
        } else if (c.update && c.policy != GF_CACHE_FIFO) {
          c.touched[c.policy == GF_CACHE_LRU ? c.qpos[slot] : slot] = c.epoch_new;
        }
      } else {
        slot = -1;
        if (direct) {
        } else if (c.miss_rows) {
          const uint32_t at = c.req_pos
              ? c.req_pos[c.map ? static_cast<uint32_t>(-(claim + 1)) : row0 + lane]
              : c.miss_index[row0 + lane];
          src = reinterpret_cast<const Unit*>(c.miss_rows) + static_cast<uint64_t>(at) * rowu;
        } else if (c.remap) {
          int32_t local = c.remap[id];
          if (local < 0) { *c.flag = 1u; local = 0; }
          src = feats + static_cast<uint64_t>(local) * rowu;
        } else {
          src = feats + static_cast<uint64_t>(id) * rowu;
          if (kStaged && c.pmap) {
            // {newest entry, the one before it}: an id that a generation running beside this
            // launch stages AGAIN is still readable where it was (the GDELT-shaped node block,
            // every id a dozen times per block, sent 7 k rows per step to the host otherwise)
            const ulonglong2 pq = reinterpret_cast<const ulonglong2*>(c.pmap)[id];
            const bool newest = static_cast<uint32_t>(pq.x >> 32) - c.st_lo <= c.st_span;
            const unsigned long long p = newest ? pq.x : pq.y;
            const uint32_t g = static_cast<uint32_t>(p >> 32);
            if (g - c.st_lo <= c.st_span)
              src = reinterpret_cast<const Unit*>(c.ring) +
                    (static_cast<uint64_t>(g & c.st_mask) * c.st_cap + static_cast<uint32_t>(p)) * rowu;
            else
              from_host = true;
          }
        }
        if (c.update) atomicMax(&c.map[id], -static_cast<int32_t>(row0 + lane + 1));
      }
    }
    if (lane < static_cast<int>(rows) && c.slot_of_row) c.slot_of_row[row0 + lane] = slot;
    acc_hits += __popcll(__ballot(slot >= 0));
    acc_miss += __popcll(__ballot(slot == -1));
    if (kStaged && c.pmap) acc_host += __popcll(__ballot(from_host));
    if (!direct) src_bits = reinterpret_cast<uint64_t>(src);
    if (early) {
    } else if (kLean || c.inflight >= 12) {
      copy(std::integral_constant<int, 12>{}, 0u);
    } else if (c.inflight >= 8) {
      copy(std::integral_constant<int, 8>{}, 0u);
    } else {
      copy(std::integral_constant<int, 4>{}, 0u);
    }
    // (behind the copy: the atomic's return value has long arrived)
    if (!kLean && c.qmode && lane < static_cast<int>(rows)) c.rep_flag[row0 + lane] = hit_code;
  }
  if (c.ctr) {
    __shared__ uint32_t wg_hits, wg_miss;
    if (threadIdx.x == 0) { wg_hits = 0; wg_miss = 0; }
    __syncthreads();
    if (lane == 0) {
      if (acc_hits) atomicAdd(&wg_hits, acc_hits);
      if (acc_miss) atomicAdd(&wg_miss, acc_miss);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int sh = bx & (kShards - 1);
      if (wg_hits) atomicAdd(&c.ctr->shard[sh].hits, wg_hits);
      if (wg_miss) atomicAdd(&c.ctr->shard[sh].n_miss, wg_miss);
      if (c.stats && wg_hits) atomicAdd(&c.stats[2 * sh], wg_hits);
    }
  }
  if (tr) tr[4] = wall_clock64();       // marks, claims and counters issued
  if (c.stats && gtid == 0) atomicAdd(&c.stats[1], n);
  if (kStaged && c.pmap && acc_host && lane == 0)
    atomicAdd(c.st_fallback, static_cast<unsigned long long>(acc_host));
  if (kStaged && c.progress && gtid == 0)
    __hip_atomic_store(c.progress, c.progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every kind of row and context: float4 rows, odd widths (16-byte vectors at 4-byte alignment),
// scalar rows; list-form, queue-form and cache-free contexts; 4 / 8 / 12 loads in flight.
__device__ inline bool ctx_direct(const Ctx& c) {
  return !c.cache_buf && !c.miss_rows && !c.remap && !c.pmap;
}
__global__ __launch_bounds__(kThreads) void gather_rows_any_kernel(Round r) {
  const Ctx& c = r.c[blockIdx.y];
  if (c.n == 0) return;
  const uint32_t bx = blockIdx.x, gx = gridDim.x;
  if (ctx_direct(c) && c.inflight >= 12) {
    if (c.vec4) gather_body<float4, false, false, false, true>(c, bx, gx);
    else if (c.odd4) gather_body<uf4, true, false, false, true>(c, bx, gx);
    else gather_body<float, false, false, false, true>(c, bx, gx);
    return;
  }
  if (c.vec4) gather_body<float4>(c, bx, gx);
  else if (c.odd4) gather_body<uf4, true>(c, bx, gx);
  else gather_body<float>(c, bx, gx);
}

// The same for rounds whose contexts ALL take the float4 / list-form or cache-free / 12-in-flight
// path (launch_round picks): a third of the code and 104-128 instead of 138 VGPRs (4 waves per
// SIMD) — same-box A/B in profiles/README, round 5.  gather_rows_kernel: every context direct
// (tables in HBM, no row mirror: the headline replay); _mirror_: rows come from wherever the probe
// says (row mirror, pulled rows, remapped local rows).
// The lean kernels' grid is ONE row of workgroups, the contexts' workgroups back to back
// (first[k] = first workgroup of context k + 1, first.w = all): the dispatcher hands workgroups to
// the 256 CUs round robin and a CU moves its workgroups' rows at ~44 GB/s however many it holds —
// the launch ends with the fullest CU (profiles/r06_gather_hop_trace.txt), and in a (x, context)
// grid the contexts' unused workgroups shift the round robin so that some CUs get one more.
// (the lean kernels' one-row grid: which context a workgroup belongs to, its index there and
// that context's workgroup count)
__device__ inline uint32_t packed_ctx(const uint4& first, uint32_t* bx, uint32_t* gx) {
  const uint32_t b = blockIdx.x;
  const uint32_t y = (b >= first.x ? 1u : 0u) + (b >= first.y ? 1u : 0u) + (b >= first.z ? 1u : 0u);
  const uint32_t lo = y == 0 ? 0u : y == 1 ? first.x : y == 2 ? first.y : first.z;
  const uint32_t hi = y == 0 ? first.x : y == 1 ? first.y : y == 2 ? first.z : first.w;
  *bx = b - lo;
  *gx = hi - lo;
  return y;
}
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(uint4 first, Round r) {
  uint32_t bx, gx;
  const uint32_t y = packed_ctx(first, &bx, &gx);
  gather_body<float4, false, true, false, true>(r.c[y], bx, gx);
}
__global__ __launch_bounds__(kThreads) void gather_rows_mirror_kernel(uint4 first, Round r) {
  uint32_t bx, gx;
  const uint32_t y = packed_ctx(first, &bx, &gx);
  gather_body<float4, false, true>(r.c[y], bx, gx);
}

// ... and the lean kernel for rounds over a host-resident table with a staging ring
__global__ __launch_bounds__(kThreads) void gather_rows_staged_kernel(uint4 first, Round r) {
  uint32_t bx, gx;
  const uint32_t y = packed_ctx(first, &bx, &gx);
  gather_body<float4, false, true, true>(r.c[y], bx, gx);
}

// ---- staging ring: rows of a HOST-resident table pulled into HBM ahead of the gather ----------
// Reference: the tables live in host memory and every miss travels host -> pinned -> device inside
// fetch_feature (cache.py:288-313,381-388, utils.py:284-297).  Here the ids of batch i+1 exist
// while batch i is fetched (ReplayPipeline), so a kernel on a side stream pulls the table rows of
// the ids that are not cached into a ring in HBM — over PCIe, beside the fetch chain — and the
// gather then takes a missed row from the ring.  The ring is G regions of C rows, one region per
// prefetch GENERATION; pmap[id] = {generation, row in its region} of the newest staging of id and
// of the one before it.  A prefetch stages an id only
// if it is neither cached (nor claimed by the fetch in flight) nor staged in a generation that is
// still readable, so a row pulled for one batch (the batch's own target edges, above all: the next
// batches sample exactly those) serves the misses of the next G - D - 1 batches too.  It is a
// HINT: the cache's state (map, slots, hit counts) never depends on it, and an id the speculation
// missed — evicted by the update in between, or a region that was full — is read from the host
// table by the gather as before.  Rows are feats[ids] bit for bit either way.
struct StageCtx {
  const int64_t* ids;
  uint32_t n;
  const int32_t* map;          // null: cache-free context (target rows)
  uint64_t num_ids;
  unsigned long long* pmap;
  uint32_t* region_rows;       // [G] rows taken in each region
  long long* region_ids;       // [C] id staged in each row of THIS generation's region
  uint32_t gen, lo, mask, cap;
  // LRU: a CACHED id whose entry is among the first `risk` of the eviction order may be gone
  // when the fetch this prefetch works for runs (up to kStageAhead updates lie in between, each
  // taking at most its block's rows from the front) — it is staged as well.  A small cache that
  // a block turns over (the headline's node cache: 2 196 slots, ~800 installs per step) would
  // otherwise send a few hundred rows per step to the host table from inside the gather.
  const uint32_t* qpos;        // null: no such rule (LFU / FIFO, cache-free context)
  const QueueState* qstate;    // queue form: the head the positions count from
  uint32_t risk;
};
struct StageRound {
  StageCtx c[kMaxCtx];
  int count;
};

// claim: one thread per block row.  A row whose id is neither cached nor staged in a readable
// generation takes the next row of this generation's region (one atomic per wave) and settles the
// id's pmap entry with a compare-and-swap — of several rows with the same id one wins, the others'
// region rows stay unused.
__global__ __launch_bounds__(256) void stage_claim_kernel(StageRound r) {
  const StageCtx& c = r.c[blockIdx.y];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t region = c.gen & c.mask, span = c.gen - c.lo;
  for (uint32_t base = blockIdx.x * 256u; base < c.n; base += gridDim.x * 256u) {
    const uint32_t i = base + threadIdx.x;
    long long id = -1;
    unsigned long long p = 0;
    bool want = false;
    if (i < c.n) {
      id = c.ids[i];
      if (id >= 0 && static_cast<uint64_t>(id) < c.num_ids) {
        // (a negative map value other than kAbsent is the claim of a fetch in flight; its update
        // installs the id — unless the block misses more ids than the cache has slots, a small
        // cache's every step — so it counts as absent: an id the fetch in flight misses is in
        // the ring already and costs nothing here)
        const int32_t slot = c.map ? c.map[id] : kAbsent;
        bool maybe = slot < 0;
        if (slot >= 0 && c.qpos) {
          uint32_t at = c.qpos[slot];
          if (c.qstate) at -= c.qstate->head;
          maybe = at < c.risk;
        }
        if (maybe) {
          p = c.pmap[2 * id];
          want = static_cast<uint32_t>(p >> 32) - c.lo > span;
        }
      }
    }
    const unsigned long long wm = __ballot(want);
    if (!wm) continue;
    const int leader = __ffsll(static_cast<long long>(wm)) - 1;
    uint32_t wbase = 0;
    if (static_cast<int>(lane) == leader)
      wbase = atomicAdd(&c.region_rows[region], static_cast<uint32_t>(__popcll(wm)));
    wbase = __shfl(wbase, leader, 64);
    const uint32_t pos = wbase + static_cast<uint32_t>(__popcll(wm & ((1ull << lane) - 1ull)));
    if (!want || pos >= c.cap) continue;
    const unsigned long long mine = (static_cast<unsigned long long>(c.gen) << 32) | pos;
    long long staged = -1;   // a row of the region that nobody reads is not pulled either
    for (;;) {
      // (the entry it replaces stays behind it: launches already in flight read the id there)
      c.pmap[2 * id + 1] = p;
      const unsigned long long old = atomicCAS(&c.pmap[2 * id], p, mine);
      if (old == p) { staged = id; break; }
      if (static_cast<uint32_t>(old >> 32) - c.lo <= span) break;   // another row of this id was first
      p = old;
    }
    c.region_ids[pos] = staged;
  }
}

// pull: the rows the claim kernel settled, host table -> this generation's region of the ring.
// A wave owns 8 consecutive ring rows — one contiguous run of stores — and keeps 2 16-byte loads
// per lane in flight over the host link (PCIe round trips are ~2 us: what counts is the number of
// reads in flight, which the number of waves provides — 6 per lane was 1.3 us per step slower in
// every grid shape, profiles/r06_pinned_pull_arrangements.txt — and every wave of the grid has
// its own rows; a first version that copied the winners of a 256-row tile inside the claim
// workgroup took 82 us for the 600 target rows of three workgroups).
struct PullJob {
  const long long* ids;        // [cap] (-1: unused row)
  const float* feats;
  float* dst;                  // the region's first row
  uint32_t* region_rows;       // rows taken in the region (may exceed cap: the excess was dropped)
  uint32_t* next_rows;         // the next generation's counter, cleared here
  unsigned long long* pulled;  // rows pulled so far (diagnostics)
  uint32_t cap, dim, vec4;
};
struct PullJobs {
  PullJob j[2];
  int count;
};

// kOdd: rows whose width is not a multiple of 4 floats move as 16-byte vectors at 4-byte alignment,
// the last one ending with the row (as the gather's odd path: GDELT's 186-d / 413-d rows went as
// single floats at first — 256 B per load on the link)
template <typename VecT, bool kOdd, uint32_t K>
__device__ inline void stage_pull_body(const PullJob& j, uint32_t n) {
  constexpr uint32_t kRows = 8;
  using Unit = std::conditional_t<kOdd, float, VecT>;
  constexpr uint32_t kVF = kOdd ? 4u : 1u;   // Units per VecT
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
  const uint32_t dimv = kOdd ? (j.dim + 3u) / 4u : j.dim / (sizeof(VecT) / sizeof(float));
  const uint32_t rowu = kOdd ? j.dim : dimv;
  const Unit* feats = reinterpret_cast<const Unit*>(j.feats);
  Unit* dst = reinterpret_cast<Unit*>(j.dst);
  for (uint32_t row0 = wave * kRows; row0 < n; row0 += nwaves * kRows) {
    const uint32_t rows = min(kRows, n - row0);
    const long long id = lane < rows ? j.ids[row0 + lane] : -1;
    const uint32_t valid = static_cast<uint32_t>(__popcll(__ballot(id >= 0)));
    if (lane == 0 && valid) atomicAdd(j.pulled, static_cast<unsigned long long>(valid));
    const uint32_t total = rows * dimv;
    Unit* o = dst + static_cast<uint64_t>(row0) * rowu;
    for (uint32_t base = 0; base < total; base += 64u * K) {
      VecT v[K];
      uint32_t at[K], ok = 0;   // (a bit per load: an array of flags went to scratch)
#pragma unroll
      for (uint32_t k = 0; k < K; ++k) {
        const uint32_t f = base + lane + 64u * k;
        const uint32_t rr = f < total ? f / dimv : 0u;
        const uint32_t cc = f - rr * dimv;
        const uint32_t off = kOdd ? min(cc * kVF, rowu - kVF) : cc;
        at[k] = rr * rowu + off;
        const long long src = __shfl(id, rr, 64);     // (every lane executes the cross-lane read)
        if (f < total && src >= 0) {
          v[k] = *reinterpret_cast<const VecT*>(feats + static_cast<uint64_t>(src) * rowu + off);
          ok |= 1u << k;
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < K; ++k)
        if (ok & (1u << k)) *reinterpret_cast<VecT*>(o + at[k]) = v[k];
    }
  }
}

__global__ __launch_bounds__(256) void stage_pull_kernel(PullJobs jobs) {
  // (selected, not indexed: a dynamic index into the by-value argument sent it to scratch)
  const PullJob j = blockIdx.y == 0 ? jobs.j[0] : jobs.j[1];
  const uint32_t n = min(*j.region_rows, j.cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *j.next_rows = 0u;   // (prefetch generations run in order on one stream)
  }
  // (nf4 / uf4, the clang vector types: an array of HIP's float4 went to scratch here)
  if (j.vec4) stage_pull_body<nf4, false, 2>(j, n);
  else if (j.dim >= 4u) stage_pull_body<uf4, true, 2>(j, n);
  else stage_pull_body<float, false, 6>(j, n);
}

// ---- planning a pull from sharded feature tables -------------------------------------------
// Cache(distributed=True) (reference: cache.py:288-313,351-388 probe the cache, `unique` the
// missed ids and pull their rows from the owning machine's KVStore, kvstore.py:285-339).  Here
// the owners are the GPUs of the node.  Per fetch round, for up to kMaxCtx contexts (a node
// block, an edge block, cache-free target rows) in the same launches:
//   claim    every missed row claims its id (atomicMax(map[id], -(row + 1)): the lowest row
//            wins — the same claim the gather makes, which it will find settled);
//   count    the rows that TRAVEL — the winners, and every row of a cache-free context — per
//            owner(key) = splitmix64(key) mod P (key: the node id; for edge rows the edge's
//            source node);
//   (the caller exchanges the counts, reads them back — the round's one host synchronisation —
//    and derives the owner-major offsets)
//   scatter  the travelling ids into the compact owner-major send buffer; req_pos[row] = the
//            position of the row's id = the index of its row in the pulled rows, which arrive
//            in the same order.
struct PullCtx {
  const int64_t* ids;
  uint32_t n;
  const int64_t* key_base;    // owner key of row i: key_base[key_index[i]] | key_base[i] | ids[i]
  const int64_t* key_index;
  int32_t* map;               // null: cache-free (every row travels)
  uint64_t num_ids;
  uint32_t* counts;           // rows per owner q at counts[q * cstride] (count: written;
  uint32_t cstride;           // scatter: read — the owner-major offsets are their prefix)
  uint32_t* cursor;           // [world] zeroed
  int64_t* send_ids;
  uint32_t* req_pos;          // [n]
};
struct PullRound {
  PullCtx c[kMaxCtx];
  int count;
  OwnerDiv od;
};

__device__ inline int64_t pull_key(const PullCtx& c, uint32_t i) {
  if (!c.key_base) return c.ids[i];
  return c.key_base[c.key_index ? c.key_index[i] : static_cast<int64_t>(i)];
}

__global__ __launch_bounds__(256) void pull_claim_kernel(PullRound r) {
  const PullCtx& c = r.c[blockIdx.y];
  if (!c.map) return;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += stride) {
    const int64_t id = c.ids[i];
    if (id < 0 || static_cast<uint64_t>(id) >= c.num_ids) continue;
    if (c.map[id] < 0) atomicMax(&c.map[id], -static_cast<int32_t>(i + 1));
  }
}

// does row i travel, and to whom (P = it does not)
__device__ inline uint32_t pull_owner(const PullCtx& c, uint32_t i, OwnerDiv od) {
  if (i >= c.n) return od.P;
  const int64_t id = c.ids[i];
  if (id < 0 || static_cast<uint64_t>(id) >= c.num_ids) return od.P;
  if (c.map && c.map[id] != -static_cast<int32_t>(i + 1)) return od.P;   // hit, or not the winner
  return owner_of(pull_key(c, i), od);
}

template <bool kScatter>
__global__ __launch_bounds__(256) void pull_bucket_kernel(PullRound r) {
  const PullCtx& c = r.c[blockIdx.y];
  const uint32_t P = r.od.P;
  const int lane = threadIdx.x & 63;
  __shared__ uint32_t s_off[64];
  if (kScatter) {   // first send position per owner: the exclusive prefix of the counts
    if (threadIdx.x == 0) {
      uint32_t at = 0;
      for (uint32_t q = 0; q < P; ++q) { s_off[q] = at; at += c.counts[q * c.cstride]; }
    }
    __syncthreads();
  }
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (c.n + stride - 1) / stride;   // uniform trip count (ballots inside)
  for (uint32_t k = 0; k < rounds; ++k) {
    const uint32_t i = k * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t o = pull_owner(c, i, r.od);
    for (uint32_t q = 0; q < P; ++q) {
      const unsigned long long m = __ballot(o == q);
      if (!m) continue;                                   // wave-uniform
      if (!kScatter) {
        if (lane == 0) atomicAdd(&c.counts[q * c.cstride], static_cast<uint32_t>(__popcll(m)));
      } else {
        uint32_t base = 0;
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        if (lane == leader) base = atomicAdd(&c.cursor[q], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, leader, 64);
        if (o == q) {
          const uint32_t at = s_off[q] + base + __popcll(m & ((1ull << lane) - 1ull));
          c.send_ids[at] = c.ids[i];
          c.req_pos[i] = at;
        }
      }
    }
  }
}

// ---- LRU bookkeeping ---------------------------------------------------------------
// Eviction priority of a slot under LFU: larger goes first, ties to the lowest slot.
// `stamp` holds the use count; priority = kAgeMax - count, with this block's hit already
// counted (`count[cached_index] += 1` before topk, lfu_cache.py:159-163).  (FIFO takes its
// victims from the rotation pointer and LRU from its queue; neither gets here.)
__device__ inline uint32_t slot_age_of(const Ctx& c, uint32_t touched, uint32_t stamp) {
  const uint32_t cnt = stamp + (touched == c.epoch_new ? 1u : 0u);
  return kAgeMax - (cnt < kAgeMax ? cnt : kAgeMax);
}
__device__ inline uint32_t slot_age(const Ctx& c, uint32_t s) {
  return slot_age_of(c, c.touched[s], c.stamp[s]);
}
__device__ inline uint32_t age_bin1(uint32_t a) {
  return a < kFine ? a : kFine + ((a - kFine) >> 11);
}
// One launch, two kinds of workgroups (per context):
//  * scan workgroups: each owns one tile of kRowTile rows, finds the representatives (first
//    row of every distinct missed id) in it, ranks them inside the tile and publishes the
//    tile's count; the next kernel adds the counts of the preceding tiles.  (Blocks of more
//    than kMaxRowTiles tiles fall back to one workgroup chaining over all tiles.)
//  * histogram workgroups: level-1 histogram of the slot ages.
__global__ __launch_bounds__(kWide) void lru_scan_hist_kernel(Round r, uint32_t scan_blocks) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy == GF_CACHE_LRU) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Both kinds of workgroup issue their first batch of loads BEFORE they look at the miss
  // count of the fetch record: one memory round trip instead of two on the critical path.
  if (blockIdx.x < scan_blocks) {
    __shared__ uint32_t wave_sums[kWide / 64];
    __shared__ uint32_t carry_s;
    constexpr uint32_t kItems = kRowTile / kWide;
    const uint32_t row_tiles = (c.n + kRowTile - 1) / kRowTile;
    const bool chained = row_tiles > kMaxRowTiles;   // one workgroup walks every tile
    if (chained ? blockIdx.x != 0 : blockIdx.x >= row_tiles) return;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const uint32_t t_begin = chained ? 0 : blockIdx.x, t_end = chained ? row_tiles : blockIdx.x + 1;
    for (uint32_t t = t_begin; t < t_end; ++t) {
      uint32_t v[kItems], local = 0;
      int32_t sr[kItems];
      int64_t idv[kItems];
      const uint32_t i0 = t * kRowTile + tid * kItems;
#pragma unroll
      for (uint32_t k = 0; k < kItems; ++k) {
        const bool ok = i0 + k < c.n;
        sr[k] = ok ? c.slot_of_row[i0 + k] : 0;
        idv[k] = ok ? c.ids[i0 + k] : 0;
      }
      if (t == t_begin && total_miss(c.ctr) == 0) return;   // uniform: nothing to update
#pragma unroll
      for (uint32_t k = 0; k < kItems; ++k) {
        // first row of a distinct missed id: its claim survived the gather's atomicMax
        v[k] = (i0 + k < c.n && sr[k] == -1 &&
                c.map[idv[k]] == -static_cast<int32_t>(i0 + k + 1)) ? 1u : 0u;
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
        if (i0 + k < c.n) {
          c.rep_flag[i0 + k] = v[k];
          c.rep_rank[i0 + k] = run;
        }
        run += v[k];
      }
      __syncthreads();
      if (tid == kWide - 1) {
        if (chained) carry_s = run;               // ranks are global already
        else c.row_tile_sum[t] = run;             // tile-local ranks + the tile's count
      }
      __syncthreads();
    }
    if (chained && tid == 0) {
      c.row_tile_sum[0] = carry_s;                // the whole block as "one tile"
    }
    return;
  }
  if (c.policy == GF_CACHE_FIFO) return;   // victims come from the rotation pointer
  __shared__ uint32_t h[kBins1];
  for (int b = tid; b < kBins1; b += kWide) h[b] = 0;
  const uint32_t hist_blocks = gridDim.x - scan_blocks;
  const uint32_t stride = hist_blocks * kWide;
  constexpr int kBatch = 4;   // slots per thread whose loads are in flight together
  bool first = true;
  const uint32_t s_first = (blockIdx.x - scan_blocks) * kWide + tid;
  for (uint32_t base = 0; base < c.capacity; base += kBatch * stride) {   // uniform trip count
    uint32_t tv[kBatch], sv[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const uint32_t s = base + s_first + j * stride;
      tv[j] = s < c.capacity ? c.touched[s] : 0u;
      sv[j] = s < c.capacity ? c.stamp[s] : 0u;
    }
    if (first) {
      first = false;
      if (total_miss(c.ctr) == 0) return;   // uniform across the launch
      __syncthreads();                      // h[] is zero
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j)
      if (base + s_first + j * stride < c.capacity)
        atomicAdd(&h[age_bin1(slot_age_of(c, tv[j], sv[j]))], 1u);
  }
  __syncthreads();
  for (int b = tid; b < kBins1; b += kWide)
    if (h[b]) atomicAdd(&c.hist1[b], h[b]);
}

// Exclusive prefix of the row-tile counts into LDS (every thread of the kWide-wide workgroup
// calls it: one tile per thread, workgroup scan); returns the number of distinct missed ids.
// `v` is the thread's own tile count, loaded by the caller with row_tile_count().
__device__ inline uint32_t row_tile_count(const Ctx& c) {
  const uint32_t row_tiles = (c.n + kRowTile - 1) / kRowTile;
  const uint32_t m = row_tiles > kMaxRowTiles ? 1u : row_tiles;   // chained scan: one entry
  return threadIdx.x < m ? c.row_tile_sum[threadIdx.x] : 0u;
}
__device__ inline uint32_t row_tile_prefix(const Ctx& c, uint32_t v,
                                           uint32_t* prefix /*[kMaxRowTiles]*/) {
  __shared__ uint32_t wsum[kWide / 64];
  __shared__ uint32_t total_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t row_tiles = (c.n + kRowTile - 1) / kRowTile;
  const bool chained = row_tiles > kMaxRowTiles;   // one entry holding the total, base 0
  const uint32_t m = chained ? 1u : row_tiles;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wsum[w];
  if (static_cast<uint32_t>(tid) < m) prefix[tid] = chained ? 0u : wbase + incl - v;
  if (tid == kWide - 1) total_s = wbase + incl;
  __syncthreads();
  return total_s;
}

// Finds the bin B (scanning from the oldest = highest bin) where the cumulative count
// reaches k; returns B and k_rem = k - (count in bins > B).  bin_load() only issues the
// loads (the first 256 threads own NBINS / 256 bins each, oldest bins first) so that a caller
// can overlap them with its other loads; bin_resolve() is called by EVERY thread of the
// workgroup (barriers inside).
template <int NBINS>
struct BinLoad {
  uint32_t mine[NBINS / 256];
  uint32_t sum;
};
template <int NBINS>
__device__ inline void bin_load(const uint32_t* __restrict__ hist, BinLoad<NBINS>& l) {
  constexpr int kPer = NBINS / 256;
  const int t = threadIdx.x;
  l.sum = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    l.mine[j] = t < 256 ? hist[NBINS - 1 - t * kPer - j] : 0u;
    l.sum += l.mine[j];
  }
}
template <int NBINS>
__device__ inline void bin_resolve(const BinLoad<NBINS>& l, uint32_t k, uint32_t* bin,
                                   uint32_t* k_rem) {
  constexpr int kPer = NBINS / 256;
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t res[2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool worker = t < 256;
  const int hi_first = NBINS - 1 - t * kPer;
  uint32_t incl = l.sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (worker && lane == 63) wsum[wave] = incl;
  if (t == 0) { res[0] = 0; res[1] = k; }
  __syncthreads();
  if (worker && k > 0) {
    uint32_t before = incl - l.sum;  // exclusive prefix over threads (older bins first)
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (before < k && before + l.sum >= k) {
      uint32_t acc = before;
#pragma unroll
      for (int j = 0; j < kPer; ++j) {
        if (acc + l.mine[j] >= k) {
          res[0] = hi_first - j;
          res[1] = k - acc;
          break;
        }
        acc += l.mine[j];
      }
    }
  }
  __syncthreads();
  *bin = res[0];
  *k_rem = res[1];
  __syncthreads();
}

// eviction threshold: every slot older than `age` goes, plus the first k_tie slots
// (slot order) of exactly that age
struct Threshold { uint32_t age; uint32_t k_tie; };

// per tile of kTile slots: how many sit exactly at the threshold age, and how many are
// older than it (all of those are evicted).  Tiles first, first + step, ... of the context;
// the caller may hand over the first tile's slot state (tv0 / sv0) if it loaded it already.
__device__ inline void count_tiles(const Ctx& c, Threshold th, uint32_t first, uint32_t step,
                                   bool preloaded, uint32_t tv0, uint32_t sv0) {
  __shared__ uint32_t cnt[2];
  const uint32_t tiles = (c.capacity + kTile - 1) / kTile;
  for (uint32_t tile = first; tile < tiles; tile += step) {
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t s = tile * kTile + threadIdx.x;
    const bool in = s < c.capacity;
    uint32_t a = 0;
    if (in) a = (preloaded && tile == first) ? slot_age_of(c, tv0, sv0) : slot_age(c, s);
    const uint32_t t = __popcll(__ballot(in && a == th.age));
    const uint32_t o = __popcll(__ballot(in && a > th.age));
    if ((threadIdx.x & 63) == 0) {
      if (t) atomicAdd(&cnt[0], t);
      if (o) atomicAdd(&cnt[1], o);
    }
    __syncthreads();
    if (threadIdx.x == 0) { c.tile_tie[tile] = cnt[0]; c.tile_old[tile] = cnt[1]; }
    __syncthreads();
  }
}

// rows: rank -> row table of the representatives that will be installed; the ones beyond
// the capacity give their claim back ("we only cache the first self.capacity",
// lru_cache.py:127-133).  slots: the eviction threshold and the per-tile counts the install
// kernel turns into ranks.  The threshold comes straight from the level-1 histogram unless
// it lies in a coarse bin (a slot untouched for more than 2047 updates): then every
// workgroup adds its share of the level-2 histogram and the LAST one to finish — told by a
// ticket — resolves the threshold and counts all tiles alone (rare, so not parallel).
// All first-pass loads (row flags, histogram, slot state, tile counts, fetch record) are
// issued together before the first dependent use.
__global__ __launch_bounds__(kWide) void lru_rank_tile_kernel(Round r) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy == GF_CACHE_LRU) return;
  const bool fifo = c.policy == GF_CACHE_FIFO;
  const uint32_t tiles = (c.capacity + kTile - 1) / kTile;
  const uint32_t i_first = blockIdx.x * kWide + threadIdx.x;
  uint32_t f0 = 0, rk0 = 0;
  int64_t id0 = 0;
  if (i_first < c.n) { f0 = c.rep_flag[i_first]; rk0 = c.rep_rank[i_first]; id0 = c.ids[i_first]; }
  const uint32_t my_tile_count = row_tile_count(c);
  BinLoad<kBins1> bl;
  bl.sum = 0;
  uint32_t tv0 = 0, sv0 = 0;
  const uint32_t s_first = blockIdx.x * kTile + threadIdx.x;
  if (!fifo) {
    bin_load<kBins1>(c.hist1, bl);
    if (blockIdx.x < tiles && s_first < c.capacity) { tv0 = c.touched[s_first]; sv0 = c.stamp[s_first]; }
  }
  if (total_miss(c.ctr) == 0) return;   // block without a miss: nothing to update

  __shared__ uint32_t tile_prefix[kMaxRowTiles];
  const uint32_t n_unique = row_tile_prefix(c, my_tile_count, tile_prefix);
  if (blockIdx.x == 0 && threadIdx.x == 0) c.ctr->n_unique = n_unique;   // for the install
  const uint32_t k = min(n_unique, c.capacity);
  if (fifo && blockIdx.x == 0 && threadIdx.x == 0) {
    // fifo_cache.py:96-105: the k slots after the pointer (wrapping) are refilled and the
    // pointer moves to the last of them; k == capacity leaves it where it was
    const uint32_t p = *c.fifo_ptr;
    c.ctr->fifo_start = p + 1 == c.capacity ? 0u : p + 1;
    *c.fifo_ptr = p + k >= c.capacity ? p + k - c.capacity : p + k;
  }
  const bool chained = (c.n + kRowTile - 1) / kRowTile > kMaxRowTiles;
  const uint32_t stride = gridDim.x * kWide;
  for (uint32_t i = i_first; i < c.n; i += stride) {
    uint32_t f = f0, rk = rk0;
    int64_t id = id0;
    if (i != i_first) {
      f = c.rep_flag[i];
      if (!f) continue;
      rk = c.rep_rank[i];
      id = c.ids[i];
    }
    if (!f) continue;
    const uint32_t rank = rk + (chained ? 0u : tile_prefix[i / kRowTile]);
    if (rank < k) {
      c.rep_row[rank] = i;
      c.rep_id[rank] = id;
    } else {
      c.map[id] = kAbsent;
    }
  }
  if (fifo) return;   // victims come from the rotation pointer
  uint32_t b1, k_rem;
  bin_resolve<kBins1>(bl, k, &b1, &k_rem);   // uniform across the workgroups
  Threshold th;
  if (b1 < kFine) {
    th.age = b1;
    th.k_tie = k_rem;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.ctr->th_age = th.age; c.ctr->th_k_tie = th.k_tie; }
    count_tiles(c, th, blockIdx.x, gridDim.x, true, tv0, sv0);
    return;
  }
  __shared__ uint32_t h[kBins2];
  __shared__ uint32_t last_s;
  for (int b = threadIdx.x; b < kBins2; b += kWide) h[b] = 0;
  __syncthreads();
  for (uint32_t s = blockIdx.x * kWide + threadIdx.x; s < c.capacity; s += stride) {
    const uint32_t a = slot_age(c, s);
    if (age_bin1(a) == b1) atomicAdd(&h[(a - kFine) & (kBins2 - 1)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins2; b += kWide)
    if (h[b]) atomicAdd(&c.hist2[b], h[b]);
  // every add above is a device-scope atomic that has completed (vmcnt(0) at the barrier)
  // before this workgroup takes its ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last_s = atomicAdd(&c.ctr->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last_s) return;
  for (int b = threadIdx.x; b < kBins2; b += kWide) h[b] = atomicAdd(&c.hist2[b], 0u);
  __syncthreads();
  BinLoad<kBins2> b2l;
  bin_load<kBins2>(h, b2l);
  uint32_t b2, k_tie;
  bin_resolve<kBins2>(b2l, k_rem, &b2, &k_tie);
  th.age = kFine + ((b1 - kFine) << 11) + b2;
  th.k_tie = k_tie;
  if (threadIdx.x == 0) { c.ctr->th_age = th.age; c.ctr->th_k_tie = th.k_tie; }
  count_tiles(c, th, 0, 1, false, 0, 0);
}

// evict + install + copy (lru_cache.py:141-160 with a deterministic tie rule): every slot
// older than the threshold plus the first k_tie slots (in slot order) exactly at it; the
// i-th evicted slot in slot order receives the i-th distinct missed id in block order,
// and its row is copied from the output rows gathered a moment ago.  Slots hit in this
// block get their stamp here.
template <typename VecT>
__device__ inline void install_body(const Ctx& c) {
  __shared__ uint32_t wave_tie[kTile / 64];
  __shared__ uint32_t wave_old[kTile / 64];
  __shared__ uint32_t red[2][kTile / 64];
  __shared__ uint2 inst[kTile];   // {slot, row} of this tile's installs
  __shared__ uint32_t n_inst;
  const VecT* out = reinterpret_cast<const VecT*>(c.out);
  VecT* cache_buf = reinterpret_cast<VecT*>(c.cache_buf);
  const uint32_t tiles = (c.capacity + kTile - 1) / kTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool fifo = c.policy == GF_CACHE_FIFO;
  const bool lfu = c.policy == GF_CACHE_LFU;
  uint32_t k = 0, start = 0, head = 0;
  Threshold th{0, 0};
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    // Every load that does not depend on another one is issued up front — this tile's slot
    // state, the preceding tiles' counts and (first pass) the fetch record — so the kernel
    // pays one memory round trip for all of them instead of one each.
    const uint32_t s = tile * kTile + threadIdx.x;
    const bool in = s < c.capacity;
    const uint32_t tv = in ? c.touched[s] : 0u;
    const uint32_t sv = in ? c.stamp[s] : 0u;
    uint32_t pt = 0, po = 0;
    if (!fifo)
      for (uint32_t t = threadIdx.x; t < tile; t += kTile) { pt += c.tile_tie[t]; po += c.tile_old[t]; }
    if (tile == blockIdx.x) {
      const uint32_t miss = total_miss(c.ctr);
      const uint32_t n_unique = c.ctr->n_unique;
      th.age = c.ctr->th_age;
      th.k_tie = c.ctr->th_k_tie;
      start = c.ctr->fifo_start;
      if (!c.update || miss == 0) return;   // block without a miss: nothing changes
      k = min(n_unique, c.capacity);
      // FIFO: the victims are the arc [start, start + k) of the slot ring; in slot order the
      // wrapped head [0, head) comes first, then [start, capacity) (fifo_cache.py:100-103)
      head = fifo && start + k > c.capacity ? start + k - c.capacity : 0u;
    }
    // bases = counts of all preceding tiles (summed by the whole workgroup)
    for (int d = 32; d > 0; d >>= 1) { pt += __shfl_down(pt, d, 64); po += __shfl_down(po, d, 64); }
    if (lane == 0) { red[0][wave] = pt; red[1][wave] = po; }
    if (threadIdx.x == 0) n_inst = 0;
    __syncthreads();
    uint32_t tie_base = 0, old_base = 0;
    for (int w = 0; w < kTile / 64; ++w) { tie_base += red[0][w]; old_base += red[1][w]; }

    const bool hit = in && tv == c.epoch_new;
    const uint32_t a = in ? slot_age_of(c, tv, sv) : 0u;
    const bool tie = in && a == th.age;
    const bool older = in && a > th.age;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long mt = __ballot(tie), mo = __ballot(older);
    if (lane == 0) { wave_tie[wave] = __popcll(mt); wave_old[wave] = __popcll(mo); }
    __syncthreads();
    uint32_t ties_before = tie_base + __popcll(mt & below);
    uint32_t old_before = old_base + __popcll(mo & below);
    for (int w = 0; w < wave; ++w) { ties_before += wave_tie[w]; old_before += wave_old[w]; }
    bool evict = k > 0 && (older || (tie && ties_before < th.k_tie));
    uint32_t v = old_before + min(ties_before, th.k_tie);   // rank in slot order
    if (fifo) {
      evict = in && (s < head || (s >= start && s - start < k));
      v = s < head ? s : head + (s - start);
    }
    bool stamped = false;
    if (evict && v < k) {
      // three independent loads, only in the (few) evicting lanes
      const uint32_t row = c.rep_row[v];
      const int64_t nid = c.rep_id[v];
      const int64_t old = c.slot_id[s];
      if (old >= 0) c.map[old] = kAbsent;
      c.slot_id[s] = nid;
      c.map[nid] = static_cast<int32_t>(s);
      c.stamp[s] = lfu ? 1u : c.epoch_new;   // lfu: count = 1
      stamped = true;
      inst[atomicAdd(&n_inst, 1u)] = make_uint2(s, row);
    }
    if (hit && !stamped)   // lru: count[cached_index] = 0; lfu: count[cached_index] += 1
      c.stamp[s] = lfu ? sv + 1u : c.epoch_new;
    __syncthreads();
    // copy the installed rows out of the block's output; the workgroup sweeps the m rows as
    // one flat array so that all of them are in flight together
    const uint32_t total = cache_buf ? n_inst * c.dimv : 0u;   // (no row mirror: ids only)
    for (uint32_t f = threadIdx.x; f < total; f += kTile) {
      const uint32_t i = f / c.dimv, cc = f - i * c.dimv;
      const uint2 p = inst[i];
      cache_buf[static_cast<uint64_t>(p.x) * c.dimv + cc] = out[static_cast<uint64_t>(p.y) * c.dimv + cc];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kTile) void lru_install_kernel(Round r) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy == GF_CACHE_LRU) return;
  if (c.vec4) install_body<float4>(c);
  else install_body<float>(c);
}

// ---- LRU as a list ------------------------------------------------------------------
// (see the file header).  c.touched[slot] = epoch of the slot's last hit (plain stores by the
// gather); c.queue[0 / 1] are the two list buffers, qstate->parity says which one is current.

// Exclusive scan of one value per thread over a kWide-wide workgroup; *total gets the sum.
// Every thread calls it (barriers inside); `ws` is kWide / 64 words of LDS.
__device__ inline uint32_t wide_excl_scan(uint32_t v, uint32_t* ws, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  __syncthreads();            // ws may still be read from a previous call
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kWide / 64; ++w) {
    const uint32_t x = ws[w];
    if (w < wave) base += x;
    sum += x;
  }
  *total = sum;
  return base + incl - v;
}

// sum of one value per thread over the workgroup (every thread calls it and gets the sum)
__device__ inline uint32_t wide_sum(uint32_t v, uint32_t* ws) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t sum = 0;
#pragma unroll
  for (int w = 0; w < kWide / 64; ++w) sum += ws[w];
  return sum;
}

// Queue form: four consecutive queue entries from p0 (16-byte aligned); bit j of the result:
// entry p0 + j is live (qpos points at it) and its slot was not hit by this block (the
// gather marked the hit entries' positions in qbits: read densely here).
// Chunks of the queue behind the head the victim walk covers for a block that missed `want`
// rows: the host sizes everything for 2 x block rows + 2 tiles (it does not know the misses);
// the device needs that much only if every row missed.  Scan, walk and install agree on it.
__device__ inline uint32_t victim_chunks_used(const Ctx& c, uint32_t want) {
  return min(c.v_chunks, (2u * want + kRowTile - 1) / kRowTile + 2u);
}

__device__ inline uint32_t victim_walk4(const Ctx& c, const uint32_t* list, uint32_t head,
                                        uint32_t tail, uint32_t p0, uint32_t* sl) {
  // the buffers are allocated 16 entries past queue_cap: a whole vector is readable
  const uint4 v = p0 < tail ? *reinterpret_cast<const uint4*>(list + p0)
                            : make_uint4(0u, 0u, 0u, 0u);
  // the four positions share one word of the hit bitmap (p0 is a multiple of 4)
  const uint32_t hitw = p0 < tail ? c.qbits[p0 >> 5] >> (p0 & 31u) : 0u;
  uint32_t qp[4], mask = 0;
  sl[0] = v.x; sl[1] = v.y; sl[2] = v.z; sl[3] = v.w;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    const bool in = p0 + j >= head && p0 + j < tail;
    if (!in) sl[j] = 0u;   // beyond the tail: not initialised
    qp[j] = c.qpos[sl[j]];
  }
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) {
    const bool in = p0 + j >= head && p0 + j < tail;
    if (in && qp[j] == p0 + j && !((hitw >> j) & 1u)) mask |= 1u << j;
  }
  return mask;
}

// One launch, three kinds of workgroups (per context), all reading what the gather left:
//  * row workgroups   [0, row_blocks): each owns `tiles_per_wg` consecutive tiles of kRowTile
//    rows, finds the representatives of the distinct missed ids in them (the row whose claim
//    on map[id] survived the gather's atomicMax), ranks them in row order inside its span and
//    publishes the span's count;
//  * list workgroups  [row_blocks, row_blocks + list_blocks): count, per tile of kRowTile list
//    entries, the slots hit by this block (they will move behind the others);
//  * the victim workgroup (last) walks the list from its front and writes down the first
//    not-hit entries — as many as the block has missed ROWS (an upper bound of the distinct
//    missed ids, which only the next kernel knows) — and the hit entries it passes on the way
//    (the next victims if a block needs more slots than its own hits leave over).
// (8 waves per SIMD = 64 VGPRs, no spill: TWO workgroups per CU — a GDELT-shaped round launches
// 730 of them, 4 us each: 19.9 -> 15.6 us per launch)
__global__ __launch_bounds__(kWide, 8) void lru_list_scan_kernel(Round r, uint32_t row_blocks,
                                                              uint32_t list_blocks,
                                                              uint32_t victim_blocks) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy != GF_CACHE_LRU || c.fused) return;
  const int tid = threadIdx.x;
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t parity = c.qstate->parity;
  const uint32_t* list = c.queue[parity & 1u];
  constexpr uint32_t kItems = kRowTile / kWide;
  if (blockIdx.x < row_blocks) {
    __shared__ uint32_t carry;
    const uint32_t row_tiles = (c.n + kLruRows - 1) / kLruRows;
    constexpr uint32_t kItems = kLruRows / kWide;   // shadows the list role's
    const uint32_t t_begin = blockIdx.x * c.tiles_per_wg;
    if (t_begin >= row_tiles) return;
    const uint32_t t_end = min(t_begin + c.tiles_per_wg, row_tiles);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t t = t_begin; t < t_end; ++t) {
      const uint32_t i0 = t * kLruRows + tid * kItems;
      int32_t sr[kItems];
      int64_t idv[kItems];
#pragma unroll
      for (uint32_t k = 0; k < kItems; ++k) {
        const bool ok = i0 + k < c.n;
        sr[k] = ok ? c.slot_of_row[i0 + k] : -2;
        idv[k] = ok ? c.ids[i0 + k] : 0;
      }
      if (t == t_begin && total_miss(c.ctr) == 0) return;   // uniform: nothing to update
      uint32_t fm[kItems], lm = 0;
#pragma unroll
      for (uint32_t k = 0; k < kItems; ++k) {
        fm[k] = (sr[k] == -1 && c.map[idv[k]] == -static_cast<int32_t>(i0 + k + 1)) ? 1u : 0u;
        lm += fm[k];
      }
      uint32_t tm;
      uint32_t run = carry + wide_excl_scan(lm, ws, &tm);
#pragma unroll
      for (uint32_t k = 0; k < kItems; ++k) {
        // (queue form: the gather left kRepHit | position for the rows that stand for a hit
        // slot and 0 for the others)
        if (i0 + k < c.n && (fm[k] || !c.qmode)) c.rep_flag[i0 + k] = fm[k] ? (kRepMiss | run) : 0u;
        run += fm[k];
      }
      __syncthreads();
      if (tid == 0) carry += tm;
      __syncthreads();
    }
    if (tid == 0) c.row_tile_sum[blockIdx.x] = carry;
    return;
  }
  const uint32_t cap = c.capacity;
  if (blockIdx.x < row_blocks + list_blocks) {
    if (c.qmode) {
      // queue form: the hit bitmap (set by the gather: 1/32 of the queue positions [head, tail))
      // per tile of kBitTile words — hit entries per tile, and per word a snapshot {word, hit
      // entries before it in its tile}: the rank of a hit entry among all of them = tile
      // prefix (the install kernel's LDS) + that + the bits below its own.  The words
      // themselves are cleared by the rows that set them, once the snapshot is all anyone reads.
      const uint32_t head = c.qstate->head, tail = c.qstate->tail;
      const uint32_t w_lo = head >> 5, w_hi = (tail + 31u) >> 5;
      const uint32_t t0 = w_lo / kBitTile;
      const uint32_t btiles = (w_hi + kBitTile - 1) / kBitTile - t0;
      const bool none = total_miss(c.ctr) == 0;
      for (uint32_t t = blockIdx.x - row_blocks; t < btiles; t += list_blocks) {
        const size_t wi = static_cast<size_t>(t0 + t) * kBitTile + tid * 4;   // four words per thread
        const uint4 wd = *reinterpret_cast<const uint4*>(c.qbits + wi);
        if (none) {
          // a block without a miss leaves the cache as it is (lru_cache.py: update() is only
          // called with missed ids): its hit marks are dropped
          if (wd.x | wd.y | wd.z | wd.w) *reinterpret_cast<uint4*>(c.qbits + wi) = make_uint4(0u, 0u, 0u, 0u);
          continue;
        }
        const uint32_t p0 = __popc(wd.x), p1 = __popc(wd.y), p2 = __popc(wd.z), p3 = __popc(wd.w);
        uint32_t total;
        const uint32_t b = wide_excl_scan(p0 + p1 + p2 + p3, ws, &total);
        // (only words with a bit set are ever looked up: 5-10 % of them on the GDELT-shaped step)
        uint4* sn = reinterpret_cast<uint4*>(c.wsnap + wi);
        if (wd.x | wd.y) sn[0] = make_uint4(wd.x, b, wd.y, b + p0);
        if (wd.z | wd.w) sn[1] = make_uint4(wd.z, b + p0 + p1, wd.w, b + p0 + p1 + p2);
        if (tid == 0) c.tile_tie[t] = total;
      }
      return;
    }
    const uint32_t list_tiles = (cap + kRowTile - 1) / kRowTile;
    if (blockIdx.x == row_blocks && tid == 0) c.ctr->q_parity = parity;
    bool first = true, staged = false;
    for (uint32_t t = blockIdx.x - row_blocks; t < list_tiles; t += list_blocks) {
      const uint32_t p0 = t * kRowTile + tid * kItems;
      uint32_t sl[kItems], hit[kItems], tc[kItems], local = 0;
      // the hit marks are indexed by list position: their loads do not wait for the list's
#pragma unroll
      for (uint32_t j = 0; j < kItems; ++j) tc[j] = p0 + j < cap ? c.touched[p0 + j] : 0u;
      if (first) {
        // the first tile is read from BOTH buffers while the parity word is still on its
        // way (one dependent hop less on the kernel's critical chain)
        uint32_t alt[kItems];
#pragma unroll
        for (uint32_t j = 0; j < kItems; ++j) {
          sl[j] = p0 + j < cap ? c.queue[0][p0 + j] : 0u;
          alt[j] = p0 + j < cap ? c.queue[1][p0 + j] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kItems; ++j) sl[j] = (parity & 1u) ? alt[j] : sl[j];
        first = false;
        const uint32_t missed = total_miss(c.ctr);
        if (missed == 0) return;   // uniform across the launch
        staged = use_staged_victims(c.stage_tiles, missed, c.stage_min);
      } else {
#pragma unroll
        for (uint32_t j = 0; j < kItems; ++j) sl[j] = p0 + j < cap ? list[p0 + j] : 0u;
      }
#pragma unroll
      for (uint32_t j = 0; j < kItems; ++j) {
        hit[j] = (p0 + j < cap && tc[j] == c.epoch_new) ? 1u : 0u;
        local += hit[j];
      }
      uint32_t total;
      if (staged && t < c.stage_tiles) {
        // the victims come off the FRONT of the list: the first tiles leave their not-hit
        // entries packed, in list order (thread order = list order); the install kernel
        // finds the m-th one through the per-tile hit counts
        uint32_t before_hits = wide_excl_scan(local, ws, &total);
        uint32_t at_keep = t * kRowTile + tid * kItems - before_hits;   // not-hit before me
        uint32_t at_hit = t * kRowTile + before_hits;
#pragma unroll
        for (uint32_t j = 0; j < kItems; ++j) {
          if (p0 + j < cap) {
            if (!hit[j]) c.v_slot[at_keep++] = sl[j];
            else if (c.stage_hits) c.v_pos[at_hit++] = sl[j];
          }
        }
      } else {
        total = wide_sum(local, ws);
      }
      if (tid == 0) {
        c.tile_tie[t] = total;
        if (total) atomicAdd(&c.tile_old[t / kQGroup], total);   // zeroed by the gather
      }
    }
    return;
  }
  const uint32_t want = min(total_miss(c.ctr), cap);
  if (c.qmode) {
    // queue form: the victims are the first LIVE entries from the head that the block did
    // not hit (the host only chooses this form for blocks of <= capacity / 4 rows, so there
    // are always enough).  The walk is spread over the victim workgroups — one CU alone is
    // bound by its 64-line-per-instruction address rate on the two scattered loads per entry
    // (measured 27-37 us for 20 k victims) — in chunks of kRowTile entries: every chunk leaves
    // its candidates and their count (the install kernel finds the m-th of them through the
    // counts).  The chunks cover 2 * rows + 2 tiles from the head; should that not yield `want`
    // candidates (many dead entries right behind the head), lru_queue_walk_kernel walks on.
    // (No "last workgroup" ticket here: the __threadfence() it needs writes the XCD's whole
    // L2 back on this part — measured +15 us.)
    const uint32_t vb = blockIdx.x - row_blocks - list_blocks;
    const uint32_t head = c.qstate->head, tail = c.qstate->tail;
    if (vb == 0 && tid == 0) { c.ctr->q_parity = parity; c.ctr->q_head = head; c.ctr->q_tail = tail; }
    if (want == 0) return;
    const uint32_t hbase = head & ~3u, chunks = victim_chunks_used(c, want);
    for (uint32_t ch = vb; ch < chunks; ch += victim_blocks) {
      const uint32_t p0 = hbase + ch * kRowTile + tid * 4;
      uint32_t sl[4];
      const uint32_t mask = victim_walk4(c, list, head, tail, p0, sl);
      uint32_t total;
      uint32_t at = ch * kRowTile + wide_excl_scan(__popc(mask), ws, &total);
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (mask & (1u << j)) { c.v_slot[at] = sl[j]; c.v_pos[at] = p0 + j; ++at; }
      if (tid == 0) c.v_count[ch] = total;
    }
    return;
  }
  if (blockIdx.x != row_blocks + list_blocks ||
      use_staged_victims(c.stage_tiles, total_miss(c.ctr), c.stage_min)) return;
  if (want == 0) return;
  uint32_t* kept = c.rep_row;     // victims: not-hit entries from the front of the list
  uint32_t* moved = c.rep_rank;   // hit entries passed on the way
  uint32_t found = 0, found_hit = 0;
  for (uint32_t base = 0; base < cap && found < want; base += kRowTile) {
    const uint32_t p0 = base + tid * kItems;
    uint32_t sl[kItems], hit[kItems], lk = 0, lh = 0;
#pragma unroll
    for (uint32_t j = 0; j < kItems; ++j) sl[j] = p0 + j < cap ? list[p0 + j] : 0u;
#pragma unroll
    for (uint32_t j = 0; j < kItems; ++j) {
      const bool in = p0 + j < cap;
      hit[j] = in ? (c.touched[p0 + j] == c.epoch_new ? 1u : 0u) : 2u;
      lk += hit[j] == 0u;
      lh += hit[j] == 1u;
    }
    uint32_t tk, th;
    uint32_t ik = found + wide_excl_scan(lk, ws, &tk);
    uint32_t ih = found_hit + wide_excl_scan(lh, ws, &th);
#pragma unroll
    for (uint32_t j = 0; j < kItems; ++j) {
      if (hit[j] == 0u) { if (ik < want) kept[ik] = sl[j]; ++ik; }
      else if (hit[j] == 1u) { if (ih < want) moved[ih] = sl[j]; ++ih; }
    }
    found += tk;
    found_hit += th;
  }
  if (tid == 0) c.ctr->q_found = min(found, want);
}

// Queue form, between the two kernels, ONE workgroup per context: did the chunks yield enough
// victim candidates?  If not (many dead entries right behind the head) it walks on alone, tile
// by tile, and leaves what it finds as one more chunk (index v_chunks).  Leaves q_found.
__global__ __launch_bounds__(kWide) void lru_queue_walk_kernel(Round r) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy != GF_CACHE_LRU || !c.qmode) return;
  const int tid = threadIdx.x;
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t want = min(total_miss(c.ctr), c.capacity);
  if (want == 0) return;
  const uint32_t head = c.ctr->q_head, tail = c.ctr->q_tail;
  const uint32_t chunks = victim_chunks_used(c, want);
  uint32_t sum = 0;
  for (uint32_t u = tid; u < chunks; u += kWide) sum += c.v_count[u];
  const uint32_t found0 = wide_sum(sum, ws);
  uint32_t found = found0;
  const uint32_t* list = c.queue[c.ctr->q_parity & 1u];
  const uint32_t limit = want > found0 ? want - found0 : 0u;   // <= block rows: fits behind the chunks
  bool walked = false;
  for (uint32_t base = (head & ~3u) + chunks * kRowTile; base < tail && found < want;
       base += kRowTile) {
    walked = true;
    const uint32_t p0 = base + tid * 4;
    uint32_t sl[4];
    const uint32_t mask = victim_walk4(c, list, head, tail, p0, sl);
    uint32_t total;
    uint32_t at = found - found0 + wide_excl_scan(__popc(mask), ws, &total);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      if (mask & (1u << j)) {
        if (at < limit) {
          c.v_slot[chunks * kRowTile + at] = sl[j];
          c.v_pos[chunks * kRowTile + at] = p0 + j;
        }
        ++at;
      }
    }
    found += total;
  }
  if (tid == 0) {
    c.v_count[chunks] = min(found - found0, limit);
    c.ctr->q_found = min(found, want);
    if (walked) c.qstate->lone_walks += 1u;
  }
}

// workgroup-wide helpers for kBlock threads (the queue form's install kernel runs many small
// workgroups per CU; the wide_* ones above are for kWide)
template <int kBlock>
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* ws, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  __syncthreads();            // ws may still be read from a previous call
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const uint32_t x = ws[w];
    if (w < wave) base += x;
    sum += x;
  }
  *total = sum;
  return base + incl - v;
}

// Copies the rows a workgroup installed — inst[j] = {slot, row} — from the block's output into
// the cache, as one flat array of 16-byte vectors, kBlock threads, K loads in flight per thread
// (rows of `rowf` floats; VecT float4 for 16-byte-aligned rows, uf4 otherwise).
template <typename VecT, int K, uint32_t kBlock = kWide>
__device__ inline void copy_installed(const Ctx& c, const uint2* inst, const int64_t* inst_id,
                                      uint32_t n_inst, uint32_t rowf, int tid) {
  if (!c.cache_buf) return;   // no row mirror: the slots hold ids only
  const uint32_t total = n_inst * c.dimv;
  const bool table = c.inst_from_table != 0;
#pragma unroll 1
  for (uint32_t f0 = tid; f0 < total; f0 += K * kBlock) {
    float4 v[K];   // (an array of the under-aligned uf4 would live in scratch)
    uint32_t dj[K], dc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t f = f0 + k * kBlock;
      const bool ok = f < total;
      const uint32_t j = ok ? f / c.dimv : 0u, cc = ok ? f - j * c.dimv : 0u;
      const uint2 pr = inst[j];
      dj[k] = ok ? pr.x : ~0u;
      dc[k] = min(cc * 4, rowf - 4);   // odd rows: the last vector ends with the row
      const float* srow = table ? c.feats + static_cast<uint64_t>(inst_id[j]) * rowf
                                : c.out + static_cast<uint64_t>(pr.y) * rowf;
      const VecT t = *reinterpret_cast<const VecT*>(srow + dc[k]);
      v[k] = make_float4(t.x, t.y, t.z, t.w);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (dj[k] != ~0u) {
        VecT t;
        t.x = v[k].x; t.y = v[k].y; t.z = v[k].z; t.w = v[k].w;
        *reinterpret_cast<VecT*>(c.cache_buf + static_cast<uint64_t>(dj[k]) * rowf + dc[k]) = t;
      }
    }
  }
}

// Queue form: applies the update, one thread per block row, kQInst rows per workgroup (many
// small workgroups per CU: every step is a chain of scattered word accesses, which only
// independent workgroups overlap):
//  * the m-th distinct missed id (m < k = min(#distinct misses, capacity, victims found))
//    takes the m-th victim candidate's slot — chunk through the chunk counts' prefix (LDS,
//    binary search), then map / slot_id / row copy as in the list form — and appends the
//    slot's new entry at tail + #hit entries + m; the one with m = k - 1 moves the head behind
//    its victim;
//  * the row that stands for a hit slot (the gather's kRepHit | old position) appends the
//    slot's new entry at tail + (hit entries before the old one) and clears its bitmap word;
//  * old entries die because qpos[] moves on.
constexpr int kQInst = 256;
constexpr uint32_t kMaxVChunks = 2048;   // victim chunks (+ the walk's) the LDS prefix holds
__global__ __launch_bounds__(kQInst) void lru_queue_install_kernel(Round r) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy != GF_CACHE_LRU || !c.qmode) return;
  const int tid = threadIdx.x;
  const uint32_t row_chunks = (c.n + kQInst - 1) / kQInst;
  if (blockIdx.x >= row_chunks || total_miss(c.ctr) == 0) return;   // uniform
  __shared__ uint32_t ws[kQInst / 64];
  __shared__ uint32_t s_tpre[kMaxBitGroups];   // hit entries before a group of bitmap tiles
  __shared__ uint32_t s_cpre[kMaxVChunks];     // victim candidates before a chunk
  __shared__ uint2 inst[kQInst];               // {slot, row} installed by this workgroup
  __shared__ int64_t inst_id[kQInst];
  __shared__ uint32_t n_inst;
  const uint32_t cap = c.capacity;
  const uint32_t q_found = c.ctr->q_found, head = c.ctr->q_head, tail = c.ctr->q_tail;
  uint32_t* q = c.queue[c.ctr->q_parity & 1u];
  const uint32_t w_lo = head >> 5, w_hi = (tail + 31u) >> 5;
  const uint32_t t0 = w_lo / kBitTile;
  const uint32_t btiles = (w_hi + kBitTile - 1) / kBitTile - t0;
  const uint32_t G = c.q_group, ngroups = (btiles + G - 1) / G;   // <= kMaxBitGroups
  const uint32_t nchunks = victim_chunks_used(c, min(total_miss(c.ctr), cap)) + 1;   // <= kMaxVChunks
  // every independent load first: the counts of the bitmap tiles (a run of consecutive groups
  // per thread), of the victim chunks (likewise) and of the scan workgroups
  constexpr uint32_t kPerT = kMaxBitGroups / kQInst, kPerC = kMaxVChunks / kQInst;
  uint32_t tv[kPerT], cv[kPerC], tm_part = 0;
  const uint32_t per_t = (ngroups + kQInst - 1) / kQInst, per_c = (nchunks + kQInst - 1) / kQInst;
#pragma unroll
  for (uint32_t j = 0; j < kPerT; ++j) {
    const uint32_t g = tid * per_t + j;
    tv[j] = 0;
    if (j < per_t && g < ngroups)
      for (uint32_t u = g * G; u < min((g + 1) * G, btiles); ++u) tv[j] += c.tile_tie[u];
  }
#pragma unroll
  for (uint32_t j = 0; j < kPerC; ++j) {
    const uint32_t ch = tid * per_c + j;
    cv[j] = (j < per_c && ch < nchunks) ? c.v_count[ch] : 0u;
  }
  const uint32_t row_tiles = (c.n + kLruRows - 1) / kLruRows;
  const uint32_t spans = (row_tiles + c.tiles_per_wg - 1) / c.tiles_per_wg;
  const uint32_t span_rows = c.tiles_per_wg * kLruRows;
  for (uint32_t t = tid; t < spans; t += kQInst) tm_part += c.row_tile_sum[t];
  // exclusive prefixes into LDS: one workgroup scan of the runs' sums each
  uint32_t th, tm, unused;
  {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPerT; ++j) sum += tv[j];
    uint32_t run = block_excl_scan<kQInst>(sum, ws, &th);
#pragma unroll
    for (uint32_t j = 0; j < kPerT; ++j) {
      const uint32_t g = tid * per_t + j;
      if (j < per_t && g < ngroups) s_tpre[g] = run;
      run += tv[j];
    }
  }
  {
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPerC; ++j) sum += cv[j];
    uint32_t run = block_excl_scan<kQInst>(sum, ws, &unused);
#pragma unroll
    for (uint32_t j = 0; j < kPerC; ++j) {
      const uint32_t ch = tid * per_c + j;
      if (j < per_c && ch < nchunks) s_cpre[ch] = run;
      run += cv[j];
    }
  }
  block_excl_scan<kQInst>(tm_part, ws, &tm);
  const uint32_t k = min(min(tm, cap), q_found);
  if (blockIdx.x == 0 && tid == 0) c.qstate->tail = tail + th + k;   // (head: by the last victim's row)
  for (uint32_t chunk = blockIdx.x; chunk < row_chunks; chunk += gridDim.x) {
    const uint32_t i = chunk * kQInst + tid;
    const bool in = i < c.n;
    const uint32_t code = in ? c.rep_flag[i] : 0u;
    const int64_t id = in ? c.ids[i] : 0;
    const uint32_t w = (chunk * kQInst) / span_rows;   // scan workgroup of these rows
    uint32_t pm_part = 0;
    for (uint32_t t = tid; t < w; t += kQInst) pm_part += c.row_tile_sum[t];
    uint32_t pm;
    block_excl_scan<kQInst>(pm_part, ws, &pm);
    if (tid == 0) n_inst = 0;
    __syncthreads();
    if (code & kRepMiss) {
      const uint32_t m = pm + (code & kRepRank);
      if (m < k) {
        uint32_t lo = 0, hi = nchunks;   // largest chunk with s_cpre[chunk] <= m
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_cpre[mid] <= m) lo = mid; else hi = mid;
        }
        const uint32_t at = lo * kRowTile + (m - s_cpre[lo]);
        const uint32_t slot = c.v_slot[at];
        const int64_t old = c.slot_id[slot];
        if (m == k - 1) c.qstate->head = c.v_pos[at] + 1u;
        if (old >= 0) c.map[old] = kAbsent;
        c.slot_id[slot] = id;
        c.map[id] = static_cast<int32_t>(slot);
        const uint32_t qa = tail + th + m;   // behind the hit entries, in victim order
        q[qa] = slot;
        c.qpos[slot] = qa;
        const uint32_t j = atomicAdd(&n_inst, 1u);
        inst[j] = make_uint2(slot, i);
        inst_id[j] = id;
      } else {
        c.map[id] = kAbsent;   // "we only cache the first self.capacity", lru_cache.py:127-133
      }
    } else if (code & kRepHit) {
      const uint32_t pos = code & kRepPos, wd = pos >> 5;
      const uint2 sn = c.wsnap[wd];
      const uint32_t slot = static_cast<uint32_t>(c.slot_of_row[i]);
      const uint32_t t = wd / kBitTile - t0, g = t / G;
      uint32_t rank = s_tpre[g] + sn.y + __popc(sn.x & ((1u << (pos & 31u)) - 1u));
      for (uint32_t u = g * G; u < t; ++u) rank += c.tile_tie[u];
      q[tail + rank] = slot;
      c.qpos[slot] = tail + rank;
      c.qbits[wd] = 0u;   // all zero again for the next update (rows sharing a word all store 0)
    }
    __syncthreads();
    // copy the installed rows out of the block's output, as one flat array
    if (c.vec4) {
      copy_installed<float4, 8, kQInst>(c, inst, inst_id, n_inst, c.dimv * 4, tid);
    } else if (c.odd4) {
      copy_installed<uf4, 8, kQInst>(c, inst, inst_id, n_inst, c.dim, tid);
    } else {
      const uint32_t total = c.cache_buf ? n_inst * c.dimv : 0u;
      for (uint32_t f = tid; f < total; f += kQInst) {
        const uint32_t j = f / c.dimv, cc = f - j * c.dimv;
        const uint2 pr = inst[j];
        c.cache_buf[static_cast<uint64_t>(pr.x) * c.dimv + cc] =
            c.inst_from_table ? c.feats[static_cast<uint64_t>(inst_id[j]) * c.dimv + cc]
                              : c.out[static_cast<uint64_t>(pr.y) * c.dimv + cc];
      }
    }
    __syncthreads();
  }
}

// Applies the update; two kinds of workgroups:
//  * row workgroups [0, row_blocks), one thread per block row: the m-th distinct missed id
//    (m < k = min(#distinct misses, capacity)) takes the m-th victim's slot — map / slot_id /
//    row copy from the freshly gathered output; the others give their claim on map[id] back;
//  * list workgroups rewrite the list into the other buffer: with L = not-hit entries ++ hit
//    entries (both in list order), the first k of L are the victims and go, in that order, to
//    the back; everything else moves up by k.  The last one flips the parity.
__global__ __launch_bounds__(kWide) void lru_list_install_kernel(Round r, uint32_t row_blocks,
                                                                 uint32_t list_blocks) {
  const Ctx& c = r.c[blockIdx.y];
  if (!c.update || c.policy != GF_CACHE_LRU || c.fused || c.qmode) return;
  const int tid = threadIdx.x;
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t row_tiles = (c.n + kLruRows - 1) / kLruRows;
  const uint32_t spans = (row_tiles + c.tiles_per_wg - 1) / c.tiles_per_wg;
  const uint32_t cap = c.capacity;
  if (blockIdx.x < row_blocks) {
    // inst_rows block rows per workgroup, one thread each (256 for the usual blocks: ALL kWide
    // threads then copy the installed rows, so the copy of a block's ~thousands of missed rows
    // is spread over n / 256 workgroups; 1024 from 65 536 rows on, where a workgroup's fixed
    // ~10 us of dependent loads — one workgroup fits a CU — would otherwise come n / 256 / 256
    // times in a row)
    __shared__ uint2 inst[kWide];   // {slot, row} installed by this workgroup
    __shared__ int64_t inst_id[kWide];   // ... and the id (the row's place in the table)
    __shared__ uint32_t n_inst;
    __shared__ uint32_t s_keep[kMaxStageTiles], s_hitp[kMaxStageTiles], s_nonhit;
    const uint32_t span_rows = c.tiles_per_wg * kLruRows;
    const uint32_t inst_rows = c.inst_rows;
    const uint32_t chunks = (c.n + inst_rows - 1) / inst_rows;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += row_blocks) {
      const uint32_t i = chunk * inst_rows + tid;
      const bool in = tid < static_cast<int>(inst_rows) && i < c.n;
      // every independent load first: the row's code and id, the span counts, the record
      const uint32_t code = in ? c.rep_flag[i] : 0u;
      const int64_t id = in ? c.ids[i] : 0;
      const uint32_t w = (chunk * inst_rows) / span_rows;   // scan workgroup of these rows
      uint32_t pm = 0, tm = 0;
      for (uint32_t t = tid; t < spans; t += kWide) {
        const uint32_t m = c.row_tile_sum[t];
        tm += m;
        if (t < w) pm += m;
      }
      const uint32_t q_found = c.ctr->q_found;
      const bool staged = use_staged_victims(c.stage_tiles, total_miss(c.ctr), c.stage_min);
      uint32_t stage_hit = 0, stage_len = 0, th_part = 0;
      if (staged && chunk == blockIdx.x) {
        // list form: hit counts of the tiles that staged their entries, and of the whole list
        if (tid < static_cast<int>(c.stage_tiles)) {
          stage_hit = c.tile_tie[tid];
          stage_len = min(kRowTile, cap - tid * kRowTile);
        }
        const uint32_t groups = ((cap + kRowTile - 1) / kRowTile + kQGroup - 1) / kQGroup;
        for (uint32_t g = tid; g < groups; g += kWide) th_part += c.tile_old[g];
      }
      if (chunk == blockIdx.x && total_miss(c.ctr) == 0) return;   // uniform
      pm = wide_sum(pm, ws);
      tm = wide_sum(tm, ws);
      if (staged && chunk == blockIdx.x) {
        uint32_t unused;
        const uint32_t th = wide_sum(th_part, ws);
        const uint32_t keep_before = wide_excl_scan(stage_len - stage_hit, ws, &unused);
        const uint32_t hit_before = wide_excl_scan(stage_hit, ws, &unused);
        if (tid < static_cast<int>(kMaxStageTiles)) {
          s_keep[tid] = keep_before;
          s_hitp[tid] = hit_before;
        }
        if (tid == 0) s_nonhit = cap - th;
      }
      if (tid == 0) n_inst = 0;
      __syncthreads();
      const uint32_t k = min(tm, cap);
      if (code & kRepMiss) {
        const uint32_t m = pm + (code & kRepRank);
        if (m < k) {
          uint32_t slot;
          int64_t old;
          if (staged) {
            // the m-th entry of (not-hit entries ++ hit entries), both in list order
            const bool keep = m < s_nonhit;
            const uint32_t x = keep ? m : m - s_nonhit;
            const uint32_t* pref = keep ? s_keep : s_hitp;
            uint32_t lo = 0, hi = c.stage_tiles;   // largest tile with pref[tile] <= x
            while (hi - lo > 1) {
              const uint32_t mid = (lo + hi) >> 1;
              if (pref[mid] <= x) lo = mid; else hi = mid;
            }
            const uint32_t at = lo * kRowTile + (x - pref[lo]);
            slot = (keep ? c.v_slot : c.v_pos)[at];
            old = c.slot_id[slot];
          } else {
            slot = m < q_found ? c.rep_row[m] : c.rep_rank[m - q_found];
            old = c.slot_id[slot];
          }
          if (old >= 0) c.map[old] = kAbsent;
          c.slot_id[slot] = id;
          c.map[id] = static_cast<int32_t>(slot);
          const uint32_t at = atomicAdd(&n_inst, 1u);
          inst[at] = make_uint2(slot, i);
          inst_id[at] = id;
        } else {
          c.map[id] = kAbsent;   // "we only cache the first self.capacity", lru_cache.py:127-133
        }
      }
      __syncthreads();
      // copy the installed rows out of the block's output, as one flat array
      const uint32_t total = n_inst * c.dimv;
      if (c.vec4) {
        if (c.inst_rows > kInstRows) copy_installed<float4, 6>(c, inst, inst_id, n_inst, c.dimv * 4, tid);
        else copy_installed<float4, 2>(c, inst, inst_id, n_inst, c.dimv * 4, tid);
      } else if (c.odd4) {
        if (c.inst_rows > kInstRows) copy_installed<uf4, 6>(c, inst, inst_id, n_inst, c.dim, tid);
        else copy_installed<uf4, 2>(c, inst, inst_id, n_inst, c.dim, tid);
      } else {
        for (uint32_t f = tid; c.cache_buf && f < total; f += kWide) {
          const uint32_t j = f / c.dimv, cc = f - j * c.dimv;
          const uint2 pr = inst[j];
          c.cache_buf[static_cast<uint64_t>(pr.x) * c.dimv + cc] =
              c.inst_from_table ? c.feats[static_cast<uint64_t>(inst_id[j]) * c.dimv + cc]
                                : c.out[static_cast<uint64_t>(pr.y) * c.dimv + cc];
        }
      }
      __syncthreads();
    }
    return;
  }
  if (blockIdx.x >= row_blocks + list_blocks) return;
  const uint32_t parity = c.ctr->q_parity;
  const uint32_t* list = c.queue[parity & 1u];
  uint32_t* next = c.queue[(parity & 1u) ^ 1u];
  // A workgroup rewrites SUB-tiles of kWide entries, one per thread (the scan kernel counted
  // the hits per tile of kRowTile = 4 sub-tiles): the rewrite's 2 x capacity scattered stores
  // — next[] nearly dense, qpos[] anywhere — are bound by the address rate of the CUs that
  // issue them, so they are spread over 4 x as many (install 11.7 -> see profiles/ with 33
  // workgroups of 4096 entries on the 134 k-slot cache).
  constexpr uint32_t kSubs = kRowTile / kWide;
  const uint32_t list_tiles = (cap + kRowTile - 1) / kRowTile;
  const uint32_t sub_tiles = (cap + kWide - 1) / kWide;
  const uint32_t groups = (list_tiles + kQGroup - 1) / kQGroup;
  // The first sub-tile's entries are read from BOTH buffers right away, together with the
  // parity word and the counts, and its hit marks (indexed by position: no need to wait for
  // the entries): two dependent hops less on the kernel's critical chain.
  const uint32_t st_first = blockIdx.x - row_blocks;
  uint32_t sl0, tc0;
  {
    const uint32_t p = st_first * kWide + tid;
    const uint32_t a0 = p < cap ? c.queue[0][p] : 0u;
    const uint32_t a1 = p < cap ? c.queue[1][p] : 0u;
    tc0 = p < cap ? c.touched[p] : 0u;
    sl0 = (parity & 1u) ? a1 : a0;
  }
  // #distinct misses and #hit slots of the whole block
  uint32_t tm = 0, th = 0;
  for (uint32_t t = tid; t < spans; t += kWide) tm += c.row_tile_sum[t];
  for (uint32_t g = tid; g < groups; g += kWide) th += c.tile_old[g];
  if (total_miss(c.ctr) == 0) return;   // block without a miss: the list stays as it is
  tm = wide_sum(tm, ws);
  th = wide_sum(th, ws);
  const uint32_t k = min(tm, cap), n_kept = cap - th;
  for (uint32_t st = st_first; st < sub_tiles; st += list_blocks) {
    const uint32_t t = st / kSubs, q = st - t * kSubs;
    const uint32_t p = st * kWide + tid;
    uint32_t sl, tc;
    if (st == st_first) {
      sl = sl0; tc = tc0;
    } else {
      sl = p < cap ? list[p] : 0u;
      tc = p < cap ? c.touched[p] : 0u;
    }
    // hit entries before this sub-tile: whole groups, the tiles of this tile's group, and
    // the sub-tiles of this tile before it (their marks, read densely)
    uint32_t before = 0;
    const uint32_t g0 = t / kQGroup;
    for (uint32_t g = tid; g < g0; g += kWide) before += c.tile_old[g];
    for (uint32_t u = g0 * kQGroup + tid; u < t; u += kWide) before += c.tile_tie[u];
    for (uint32_t j = 0; j < q; ++j) {
      const uint32_t pj = t * kRowTile + j * kWide + tid;   // < p <= cap
      before += (pj < cap && c.touched[pj] == c.epoch_new) ? 1u : 0u;
    }
    const uint32_t hit = (p < cap && tc == c.epoch_new) ? 1u : 0u;
    before = wide_sum(before, ws);
    uint32_t total;
    const uint32_t hb = before + wide_excl_scan(hit, ws, &total);   // hit entries before p
    if (p < cap) {
      const uint32_t l = hit ? n_kept + hb : p - hb;   // index in L
      const uint32_t at = l < k ? cap - k + l : l - k;
      next[at] = sl;
      c.qpos[sl] = at;   // where the next block's hits of this slot leave their mark
    }
  }
  if (blockIdx.x == row_blocks && tid == 0) c.qstate->parity = parity ^ 1u;
}

// ---- LRU list form in ONE launch ----------------------------------------------------------
// lru_list_scan_kernel + lru_list_install_kernel as one launch: what the second launch read
// from the first — counts per tile, the victims at the front of the list — travels between
// workgroups of the SAME launch: counts as 8-byte granules {launch tag, count} (one relaxed
// agent-scope store; the mechanism of merge_slots_fused_kernel, sampler.hip), the staged
// victims as write-through (sc1) stores that are drained (s_waitcnt vmcnt(0), workgroup
// barrier) before the tile's granule is published, and read with sc1 loads only
// (MI355X_MICROARCH, inter-workgroup visibility, "valid forms": row 1 of the table).
//
// Three kinds of workgroups, in this order of blockIdx.x — every wait is for a workgroup with
// a LOWER index, which was dispatched earlier:
//  * count  [0, cb)            a tile of kFuseTile list entries: marks read densely (they are
//                              indexed by list position), hits counted; the tiles that can
//                              hold one of the block's victims (those below `want` + hit rows)
//                              stage their not-hit entries packed in list order, each with
//                              the id it holds (the row role then needs no hop through
//                              slot_id[]); publishes {tag, #hits}.  Waits for nobody.
//  * row    [cb, cb + rb)      fuse_rows block rows: representatives of the distinct missed
//                              ids ranked in the span; publishes {tag, #representatives},
//                              looks back over the row workgroups before it (global rank m),
//                              reads every count granule (the m-th entry of not-hit ++ hit
//                              entries = the victim: tile by binary search in LDS, entry from
//                              the staging arrays), installs — map / slot_id / row copy.
//  * write  [cb + rb, …)       a tile of kFuseTile list entries: needs #distinct misses (all
//                              row granules) and the hits before it (count granules), writes
//                              the permuted list into the other buffer and qpos[]; the first
//                              one flips the parity.
// A poll that has not seen its granule after g_fuse_spins (4 096) tries stops waiting and computes the
// value itself from the kernel's immutable inputs (marks, list, claims), so termination does
// not depend on dispatch order (several such launches of different processes sharing the
// GPU can fill an XCD with waiters: DESIGN 6.1).  The one input that is NOT immutable is the
// claim map[id] == -(row + 1) of a representative, which the row role overwrites when it
// installs: a representative therefore first marks slot_of_row[row] = kRepMark (write-through,
// drained) and a recount reads the claim first, the mark second.
constexpr uint32_t kFuseTile = kWide;         // list entries per count / write tile
constexpr uint32_t kFuseMaxTiles = 2048;      // list tiles (LDS prefix arrays): <= 2 M slots
constexpr uint32_t kFuseMaxRowWgs = 1024;     // row workgroups: <= 1 M block rows
constexpr uint32_t kFuseSpinsDefault = 1u << 12;
// (a device word so that a test can force every wait into its recount path:
// GNNFLOW_LRU_FUSE_SPINS, read when the library loads its first cache)
__device__ uint32_t g_fuse_spins = kFuseSpinsDefault;
constexpr int32_t kRepMark = -3;              // slot_of_row[]: representative of a missed id
__device__ unsigned int g_lru_recounts;       // granules a waiter had to recompute itself

#define GF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Polls up to three granules with all loads in flight per round (a look-back granule and two
// count granules cost one round trip, not three); out[k] = the count, or ~0u for a granule
// that never showed the tag (null pointer: not wanted, 0).
__device__ inline void fuse_poll3(const unsigned long long* g0, const unsigned long long* g1,
                                  const unsigned long long* g2, uint32_t tag, uint32_t* out) {
  const unsigned long long* g[3] = {g0, g1, g2};
  bool need[3], any = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    need[k] = g[k] != nullptr;
    out[k] = need[k] ? ~0u : 0u;
    any |= need[k];
  }
  const uint32_t budget = g_fuse_spins;
  for (uint32_t spins = 0; any && spins < budget; ++spins) {
    unsigned long long x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = need[k] ? __hip_atomic_load(g[k], GF_RLX_AGENT) : 0ull;
    any = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (need[k]) {
        if (static_cast<uint32_t>(x[k] >> 32) == tag) {
          out[k] = static_cast<uint32_t>(x[k]);
          need[k] = false;
        } else {
          any = true;
        }
      }
    }
    if (any) __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (need[k]) atomicAdd(&g_lru_recounts, 1u);
}

__device__ inline void fuse_publish(unsigned long long* g, uint32_t tag, uint32_t value) {
  __hip_atomic_store(g, (static_cast<unsigned long long>(tag) << 32) | value, GF_RLX_AGENT);
}

__device__ inline uint32_t total_hits(const Counters* c) {
  uint32_t h = 0;
#pragma unroll
  for (int i = 0; i < kShards; ++i) h += c->shard[i].hits;
  return h;
}

// parity of the list buffer that is current DURING the launch tagged `tag`
__device__ inline uint32_t fuse_parity(const Ctx& c) {
  const unsigned long long w = *reinterpret_cast<const unsigned long long*>(c.qstate);
  const uint32_t parity = static_cast<uint32_t>(w), flip = static_cast<uint32_t>(w >> 32);
  return (flip == c.fuse_tag ? parity ^ 1u : parity) & 1u;
}

// hits in list tile t, from the marks (what a count workgroup publishes)
__device__ inline uint32_t fuse_recount_tile(const Ctx& c, uint32_t t) {
  uint32_t h = 0;
  const uint32_t lo = t * kFuseTile, hi = min(lo + kFuseTile, c.capacity);
  for (uint32_t p = lo; p < hi; ++p) h += c.touched[p] == c.epoch_new ? 1u : 0u;
  return h;
}

// representatives among the rows of row workgroup b (what it publishes)
__device__ inline uint32_t fuse_recount_rows(const Ctx& c, uint32_t b) {
  uint32_t m = 0;
  const uint32_t lo = b * c.fuse_rows, hi = min(lo + c.fuse_rows, c.n);
  for (uint32_t i = lo; i < hi; ++i) {
    const int64_t id = c.ids[i];
    if (id < 0 || static_cast<uint64_t>(id) >= c.num_ids) continue;
    // the claim first, the mark second (see above)
    const int32_t claim = __hip_atomic_load(&c.map[id], GF_RLX_AGENT);
    const int32_t sr = __hip_atomic_load(&c.slot_of_row[i], GF_RLX_AGENT);
    if (sr == kRepMark || (sr == -1 && claim == -static_cast<int32_t>(i + 1))) ++m;
  }
  return m;
}

// the x-th hit (want_hit) / not-hit entry of list tile t, walked serially (fallback of a row
// thread whose tile never published its staged entries)
__device__ inline uint32_t fuse_walk_tile(const Ctx& c, const uint32_t* list, uint32_t t,
                                          uint32_t x, bool want_hit) {
  const uint32_t lo = t * kFuseTile, hi = min(lo + kFuseTile, c.capacity);
  uint32_t seen = 0;
  for (uint32_t p = lo; p < hi; ++p) {
    const bool hit = c.touched[p] == c.epoch_new;
    if (hit == want_hit) {
      if (seen == x) return list[p];
      ++seen;
    }
  }
  return list[lo];   // unreachable: the prefix said the tile has more than x such entries
}

#define GF_STAMP(k) \
  do { if (c.trace && threadIdx.x == 0) c.trace[4 + vx * 8 + (k)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(kWide) void lru_list_fused_kernel(Round r, uint32_t count_blocks,
                                                               uint32_t row_blocks,
                                                               uint32_t write_blocks) {
  // (the roles in THIS order of blockIdx.x — count, row, write — because every wait is for a
  // workgroup dispatched earlier; dispatching the row role, whose chain is the longest, first
  // saved 0.5 us of an isolated launch and cost 7 us per step in the pipelined loop, where the
  // row workgroups then spin for count workgroups that other kernels keep from starting:
  // profiles/README.md, round 6)
  // (the hot fields pinned into SGPRs here: one round of scalar loads instead of one per field
  // where it is first used — see gather_body)
  const Ctx& c = r.c[blockIdx.y];
  asm volatile("" :: "s"(c.update), "s"(c.policy), "s"(c.fused), "s"(c.trace), "s"(c.n),
               "s"(c.capacity), "s"(c.fuse_tag), "s"(c.fuse_rows), "s"(c.touched), "s"(c.queue[0]),
               "s"(c.queue[1]), "s"(c.qstate), "s"(c.ctr), "s"(c.epoch_new), "s"(c.slot_id),
               "s"(c.ids), "s"(c.map), "s"(c.slot_of_row), "s"(c.g_cnt), "s"(c.g_row),
               "s"(c.v_slot), "s"(c.v_old), "s"(c.v_pos), "s"(c.v_hold), "s"(c.qpos),
               "s"(c.num_ids), "s"(c.cache_buf));
  const uint32_t vx = blockIdx.x;
  if (!c.update || c.policy != GF_CACHE_LRU || !c.fused) return;
  const int tid = threadIdx.x;
  if (c.trace && vx == 0 && tid == 0) {
    c.trace[0] = count_blocks; c.trace[1] = row_blocks; c.trace[2] = write_blocks; c.trace[3] = c.fuse_tag;
  }
  GF_STAMP(0);
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t cap = c.capacity, tag = c.fuse_tag;
  const uint32_t tiles = (cap + kFuseTile - 1) / kFuseTile;
  const uint32_t row_wgs = (c.n + c.fuse_rows - 1) / c.fuse_rows;

  if (vx < count_blocks) {
    // ---- count role ----
    bool first = true;
    uint32_t par = 0, bound = 0;
    bool stage_hits = false;
    for (uint32_t t = vx; t < tiles; t += count_blocks) {
      const uint32_t p = t * kFuseTile + tid;
      const bool in = p < cap;
      const uint32_t tc = in ? c.touched[p] : 0u;
      uint32_t sl;
      if (first) {
        // both buffers while the parity word is on its way (one dependent hop less)
        const uint32_t a0 = in ? c.queue[0][p] : 0u;
        const uint32_t a1 = in ? c.queue[1][p] : 0u;
        par = fuse_parity(c);
        const uint32_t missed = total_miss(c.ctr);
        if (missed == 0) return;   // uniform across the launch
        const uint32_t hit_rows = total_hits(c.ctr);
        const uint32_t want = min(missed, cap);
        // the m-th not-hit entry (m < want) lies below list position want + #hit entries
        bound = min(cap, want + hit_rows);
        // victims beyond the not-hit entries: only if misses + hits exceed the capacity
        stage_hits = static_cast<uint64_t>(want) + hit_rows > cap;
        sl = par ? a1 : a0;
        first = false;
      } else {
        sl = in ? c.queue[par][p] : 0u;
      }
      const bool hit = in && tc == c.epoch_new;
      const bool stage = t * kFuseTile < bound;
      if (t == vx) GF_STAMP(1);   // marks, list entries, parity and counters are in
      long long old = -1;
      if (stage && in && (!hit || stage_hits)) old = c.slot_id[sl];
      uint32_t total;
      const uint32_t hb = wide_excl_scan(hit ? 1u : 0u, ws, &total);
      if (stage && in) {
        if (!hit) {
          const uint32_t at = t * kFuseTile + (tid - hb);
          __hip_atomic_store(&c.v_slot[at], sl, GF_RLX_AGENT);
          __hip_atomic_store(&c.v_old[at], old, GF_RLX_AGENT);
        } else if (stage_hits) {
          const uint32_t at = t * kFuseTile + hb;
          __hip_atomic_store(&c.v_pos[at], sl, GF_RLX_AGENT);
          __hip_atomic_store(&c.v_hold[at], old, GF_RLX_AGENT);
        }
      }
      // every storing wave drains its write-through stores, then the barrier, then ONE lane
      // publishes
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) fuse_publish(&c.g_cnt[t], tag, total);
      if (t == vx) GF_STAMP(2);   // staged entries drained, count published
    }
    return;
  }

  __shared__ uint32_t s_keep[kFuseMaxTiles + 1], s_hitp[kFuseMaxTiles + 1];
  __shared__ uint32_t s_direct;

  if (vx < count_blocks + row_blocks) {
    // ---- row role ----
    const uint32_t b = vx - count_blocks;
    if (b >= row_wgs) return;
    __shared__ uint2 inst[kWide];        // {slot, row} installed by this workgroup
    __shared__ int64_t inst_id[kWide];
    __shared__ uint32_t n_inst;
    const uint32_t i = b * c.fuse_rows + tid;
    const bool in = tid < static_cast<int>(c.fuse_rows) && i < c.n;
    const int32_t sr = in ? c.slot_of_row[i] : -2;
    const int64_t id = in ? c.ids[i] : 0;
    const uint32_t par = fuse_parity(c);
    if (total_miss(c.ctr) == 0) return;   // uniform
    const bool fm = sr == -1 && c.map[id] == -static_cast<int32_t>(i + 1);
    if (fm) {
      // write-through and DRAINED before this workgroup stores anything else: a recount by
      // another workgroup reads the claim first, the mark second, and must find the mark once
      // the install below has overwritten the claim
      __hip_atomic_store(&c.slot_of_row[i], kRepMark, GF_RLX_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t cnt;
    const uint32_t rank = wide_excl_scan(fm ? 1u : 0u, ws, &cnt);
    if (tid == 0) {
      fuse_publish(&c.g_row[b], tag, cnt);
      n_inst = 0;
      s_direct = 0;
    }
    GF_STAMP(1);   // rows read (slot_of_row, ids, claims), representatives ranked and published
    // look-back over the row workgroups before this one (at most kFuseMaxRowWgs = kWide: one
    // per thread) and the hits per list tile (two per thread), all in flight together
    constexpr uint32_t kPer = kFuseMaxTiles / kWide;
    static_assert(kPer == 2 && kFuseMaxRowWgs <= kWide, "fuse_poll3: one row + two count granules");
    uint32_t pv[3];
    fuse_poll3(static_cast<uint32_t>(tid) < b ? &c.g_row[tid] : nullptr,
               tid * kPer < tiles ? &c.g_cnt[tid * kPer] : nullptr,
               tid * kPer + 1 < tiles ? &c.g_cnt[tid * kPer + 1] : nullptr, tag, pv);
    if (pv[0] == ~0u) pv[0] = fuse_recount_rows(c, tid);
    GF_STAMP(5);   // thread 0's own granules are in (diagnostics)
    const uint32_t pm = wide_sum(pv[0], ws);
    GF_STAMP(6);   // every thread's are (the sum is a barrier)
    // hits per list tile -> prefix of not-hit / hit entries per tile
    uint32_t hv[kPer], run_h = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      hv[k] = pv[1 + k];
      if (hv[k] == ~0u) {
        hv[k] = fuse_recount_tile(c, tid * kPer + k);
        s_direct = 1u;   // its staged entries may never arrive: walk the tiles instead
      }
      run_h += hv[k];
    }
    uint32_t th;
    uint32_t hb = wide_excl_scan(run_h, ws, &th);
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      const uint32_t t = tid * kPer + k;
      if (t <= tiles) {
        s_hitp[t] = hb;
        s_keep[t] = min(t * kFuseTile, cap) - hb;
      }
      hb += hv[k];
    }
    __syncthreads();
    GF_STAMP(2);   // every granule before this workgroup is in, prefixes in LDS
    const uint32_t n_kept = cap - th;
    if (fm) {
      const uint32_t m = pm + rank;
      if (m < cap) {   // "we only cache the first self.capacity", lru_cache.py:127-133
        const bool keep = m < n_kept;
        const uint32_t x = keep ? m : m - n_kept;
        const uint32_t* pref = keep ? s_keep : s_hitp;
        uint32_t lo = 0, hi = tiles;   // largest tile with pref[tile] <= x
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (pref[mid] <= x) lo = mid; else hi = mid;
        }
        uint32_t slot;
        long long old;
        if (s_direct) {
          slot = fuse_walk_tile(c, c.queue[par], lo, x - pref[lo], !keep);
          old = c.slot_id[slot];
        } else {
          const uint32_t at = lo * kFuseTile + (x - pref[lo]);
          slot = __hip_atomic_load(keep ? &c.v_slot[at] : &c.v_pos[at], GF_RLX_AGENT);
          old = __hip_atomic_load(keep ? &c.v_old[at] : &c.v_hold[at], GF_RLX_AGENT);
        }
        // (the mark store above has long been drained by the waits in between)
        if (old >= 0) c.map[old] = kAbsent;
        c.slot_id[slot] = id;
        c.map[id] = static_cast<int32_t>(slot);
        const uint32_t at = atomicAdd(&n_inst, 1u);
        inst[at] = make_uint2(slot, i);
        inst_id[at] = id;
      } else {
        c.map[id] = kAbsent;
      }
    }
    __syncthreads();
    GF_STAMP(3);   // victims read, map / slot_id written
    const uint32_t total = n_inst * c.dimv;
    if (c.vec4) {
      if (c.fuse_rows > kInstRows) copy_installed<float4, 6>(c, inst, inst_id, n_inst, c.dimv * 4, tid);
      else copy_installed<float4, 2>(c, inst, inst_id, n_inst, c.dimv * 4, tid);
    } else if (c.odd4) {
      if (c.fuse_rows > kInstRows) copy_installed<uf4, 6>(c, inst, inst_id, n_inst, c.dim, tid);
      else copy_installed<uf4, 2>(c, inst, inst_id, n_inst, c.dim, tid);
    } else {
      for (uint32_t f = tid; c.cache_buf && f < total; f += kWide) {
        const uint32_t j = f / c.dimv, cc = f - j * c.dimv;
        const uint2 pr = inst[j];
        c.cache_buf[static_cast<uint64_t>(pr.x) * c.dimv + cc] =
            c.inst_from_table ? c.feats[static_cast<uint64_t>(inst_id[j]) * c.dimv + cc]
                              : c.out[static_cast<uint64_t>(pr.y) * c.dimv + cc];
      }
    }
    GF_STAMP(4);   // installed rows copied
    return;
  }

  // ---- write role ----
  const uint32_t wb = vx - count_blocks - row_blocks;
  if (wb >= write_blocks || wb >= tiles) return;
  uint32_t sl0, tc0, par;
  {
    const uint32_t p = wb * kFuseTile + tid;
    const uint32_t a0 = p < cap ? c.queue[0][p] : 0u;
    const uint32_t a1 = p < cap ? c.queue[1][p] : 0u;
    tc0 = p < cap ? c.touched[p] : 0u;
    par = fuse_parity(c);
    sl0 = par ? a1 : a0;
  }
  if (total_miss(c.ctr) == 0) return;   // the list stays as it is
  // Stay off the granules' lines for ~2 us: nothing this role waits for is there before, and every
  // poll of a line slows the hand-over of the granules in it down — the row role's look-back, which
  // is the launch's critical path, completes 1.1 us earlier when the 132 write workgroups of the
  // headline's update do not poll beside it (profiles/r06_lru_hop_trace.txt; a longer nap makes
  // the late-dispatched write workgroups the tail instead: 3 / 4 / 5 us: +0.5 / +1.3 / +2.1 us)
  __builtin_amdgcn_s_sleep(32);
  __builtin_amdgcn_s_sleep(32);
  // #distinct misses of the whole block (every row granule) and the hits per tile, all in
  // flight together
  constexpr uint32_t kPer = kFuseMaxTiles / kWide;
  uint32_t pv[3];
  fuse_poll3(static_cast<uint32_t>(tid) < row_wgs ? &c.g_row[tid] : nullptr,
             tid * kPer < tiles ? &c.g_cnt[tid * kPer] : nullptr,
             tid * kPer + 1 < tiles ? &c.g_cnt[tid * kPer + 1] : nullptr, tag, pv);
  if (pv[0] == ~0u) pv[0] = fuse_recount_rows(c, tid);
  const uint32_t tm = wide_sum(pv[0], ws);
  // hits per tile -> hits before every tile
  uint32_t hv[kPer], run_h = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    hv[k] = pv[1 + k];
    if (hv[k] == ~0u) hv[k] = fuse_recount_tile(c, tid * kPer + k);
    run_h += hv[k];
  }
  uint32_t th;
  uint32_t hbt = wide_excl_scan(run_h, ws, &th);
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t t = tid * kPer + k;
    if (t <= tiles) s_hitp[t] = hbt;
    hbt += hv[k];
  }
  __syncthreads();
  GF_STAMP(1);   // every row and count granule is in
  const uint32_t k = min(tm, cap), n_kept = cap - th;
  const uint32_t* list = c.queue[par];
  uint32_t* next = c.queue[par ^ 1u];
  for (uint32_t t = wb; t < tiles; t += write_blocks) {
    const uint32_t p = t * kFuseTile + tid;
    uint32_t sl, tc;
    if (t == wb) {
      sl = sl0; tc = tc0;
    } else {
      sl = p < cap ? list[p] : 0u;
      tc = p < cap ? c.touched[p] : 0u;
    }
    const uint32_t hit = (p < cap && tc == c.epoch_new) ? 1u : 0u;
    uint32_t total;
    const uint32_t hb = s_hitp[t] + wide_excl_scan(hit, ws, &total);   // hit entries before p
    if (p < cap) {
      const uint32_t l = hit ? n_kept + hb : p - hb;   // index in (not-hit ++ hit entries)
      const uint32_t at = l < k ? cap - k + l : l - k;
      next[at] = sl;
      c.qpos[sl] = at;   // where the next block's hits of this slot leave their mark
    }
  }
  GF_STAMP(2);   // list tile(s) rewritten
  if (wb == 0 && tid == 0) {
    // {new parity, this launch's tag} in ONE store: fuse_parity() of a late workgroup of this
    // launch still resolves to `par`
    *reinterpret_cast<unsigned long long*>(c.qstate) =
        (static_cast<unsigned long long>(tag) << 32) | (par ^ 1u);
  }
}

// list of a freshly initialised cache: slot order; `prefix` new slots [first, first + prefix)
// go in front of the `old_n` entries of `old` (Cache.resize)
__global__ void list_fill_kernel(uint32_t* list, uint32_t first, uint32_t prefix,
                                 const uint32_t* old, uint32_t old_n) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < prefix + old_n; j += stride)
    list[j] = j < prefix ? first + j : old[j - prefix];
}

// ---- LRU as a queue (large caches) --------------------------------------------------------
// The list passes above cost O(capacity) per update: 352 us for a 30 k-row block on a 40 M-slot
// cache (GDELT scale) against 17 us for the gather itself.  From queue_min_capacity() slots on
// (0.5 M), the SAME list is therefore kept as a queue with dead entries: `queue` holds entries
// [head, tail) (capacity * 3/2 allocated), qpos[slot] is the position of the slot's one LIVE
// entry, and an update only appends — the distinct hit slots in the order of their old entries,
// then the k victims, which are the first k live, not-hit entries from the head.  Old entries
// die because qpos[] moves on.  Reading the live entries from head to tail gives exactly the
// list of the list form, so both forms — and the oracle — make the same decisions.
//   gather       : a hit sets the bit of the slot's queue position in `qbits` (atomicOr); the
//                  row whose atomic set it stands for the slot (rep_flag = kRepHit | position).
//                  The hit slots thus come out deduplicated AND in queue order without a sort
//                  (a 7-launch device radix sort cost 35 us here), and nothing else in the
//                  update touches a per-slot hit mark
//   list scan    : row role — representatives of the distinct missed ids, as in the list form;
//                  victim role — chunks of the queue behind the head, one workgroup each, keep
//                  their live, not-hit entries (one scattered load per entry: qpos; the hit
//                  bits are read densely); bitmap role — per tile of kBitTile words (the
//                  bitmap is 1/32 of the queue: 7.5 MB at 40 M slots) the hit entries, per word
//                  a snapshot {word, hits before it in the tile}
//   queue walk   : ONE workgroup: did the chunks yield enough candidates?  If not it walks on
//                  and leaves what it finds as one more chunk
//   queue install: 256-thread workgroups, one thread per block row.  The m-th distinct missed
//                  id takes the m-th victim candidate's slot (chunk through the counts' prefix
//                  in LDS) and appends its entry at tail + hits + m; the row that stands for a
//                  hit slot appends at tail + (hit entries before its old one: tile prefix
//                  from LDS + the snapshot) and clears its bitmap word; head / tail move.
// Every step is O(block rows) and row-parallel: on the GDELT-shaped step (38 M slots, 198 k-
// row blocks) the update costs 108 us per step against 232 with round 4's position-parallel
// append (the non-empty bitmap tiles expanded serially per workgroup); profiles/README.
// When the queue's tail would pass its allocation it is compacted into the other buffer (two
// launches, O(capacity), once per ~capacity / (2 * block rows) updates); a block of more than
// capacity / 4 rows is handled by the list form on the compacted queue (its passes are no
// longer the larger term then) and qpos[] is rebuilt behind it.
struct CompactState { uint32_t parity, tail, pad[2]; };

__global__ __launch_bounds__(kWide) void lru_queue_compact_count_kernel(
    const uint32_t* q0, const uint32_t* q1, const QueueState* qs, const uint32_t* qpos,
    unsigned long long* live_bits, uint32_t* tile_cnt, uint32_t* group_sum, CompactState* st) {
  const int tid = threadIdx.x;
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t parity = qs->parity, tail = qs->tail;
  const uint32_t* q = (parity & 1u) ? q1 : q0;
  if (blockIdx.x == 0 && tid == 0) { st->parity = parity; st->tail = tail; }
  constexpr uint32_t kItems = kRowTile / kWide;
  const uint32_t tiles = (tail + kRowTile - 1) / kRowTile;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    uint32_t local = 0;
#pragma unroll
    for (uint32_t j = 0; j < kItems; ++j) {
      const uint32_t p = t * kRowTile + j * kWide + tid;
      bool live = false;
      if (p < tail) live = qpos[q[p]] == p;
      const unsigned long long b = __ballot(live);
      if ((tid & 63) == 0) live_bits[p >> 6] = b;
      local += live ? 1u : 0u;
    }
    const uint32_t total = wide_sum(local, ws);
    if (tid == 0) {
      tile_cnt[t] = total;
      if (total) atomicAdd(&group_sum[t / kQGroup], total);
    }
  }
}

__global__ __launch_bounds__(kWide) void lru_queue_compact_write_kernel(
    uint32_t* q0, uint32_t* q1, QueueState* qs, uint32_t* qpos,
    const unsigned long long* live_bits, const uint32_t* tile_cnt, const uint32_t* group_sum,
    const CompactState* st, uint32_t capacity) {
  const int tid = threadIdx.x;
  __shared__ uint32_t ws[kWide / 64];
  const uint32_t parity = st->parity, tail = st->tail;
  const uint32_t* q = (parity & 1u) ? q1 : q0;
  uint32_t* next = (parity & 1u) ? q0 : q1;
  constexpr uint32_t kItems = kRowTile / kWide;
  const uint32_t tiles = (tail + kRowTile - 1) / kRowTile;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    uint32_t before = 0;
    const uint32_t g0 = t / kQGroup;
    for (uint32_t g = tid; g < g0; g += kWide) before += group_sum[g];
    for (uint32_t u = g0 * kQGroup + tid; u < t; u += kWide) before += tile_cnt[u];
    uint32_t run = wide_sum(before, ws);
#pragma unroll
    for (uint32_t j = 0; j < kItems; ++j) {
      const uint32_t p = t * kRowTile + j * kWide + tid;
      const uint32_t live = static_cast<uint32_t>((live_bits[p >> 6] >> (tid & 63)) & 1ull);
      uint32_t total;
      const uint32_t at = run + wide_excl_scan(live, ws, &total);
      if (live) {
        const uint32_t s = q[p];
        next[at] = s;
        qpos[s] = at;
      }
      run += total;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    qs->parity = parity ^ 1u;
    qs->head = 0;
    qs->tail = capacity;   // every slot has exactly one live entry
  }
}

// qpos of a dense list (after init, resize, or a list-form update of a queue-capable cache)
__global__ void lru_queue_index_kernel(const uint32_t* q0, const uint32_t* q1,
                                       const QueueState* qs, uint32_t* qpos, uint32_t capacity) {
  const uint32_t* list = (qs->parity & 1u) ? q1 : q0;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < capacity; p += stride)
    qpos[list[p]] = p;
}

__global__ void cache_fill_kernel(int32_t* map, uint64_t num_ids, int64_t* slot_id,
                                  uint32_t* stamp, uint32_t* touched, uint64_t capacity,
                                  int identity, uint32_t stamp0) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_ids;
       i += stride)
    map[i] = (identity && i < capacity) ? static_cast<int32_t>(i) : kAbsent;
  for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < capacity;
       s += stride) {
    slot_id[s] = identity ? static_cast<int64_t>(s) : -1;
    stamp[s] = stamp0;
    touched[s] = 0;
  }
}

// slot of every id (>= 0: cached there, -1: not cached, -2: out of range): what a caller
// that pulls missed rows from their owners needs to know before the fetch
__global__ void cache_probe_kernel(const int64_t* __restrict__ ids, uint64_t n,
                                   const int32_t* __restrict__ map, uint64_t num_ids,
                                   int32_t* __restrict__ slot) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t id = ids[i];
    int32_t v = -2;
    if (id >= 0 && static_cast<uint64_t>(id) < num_ids) {
      v = map ? map[id] : kAbsent;
      if (v < 0) v = -1;
    }
    slot[i] = v;
  }
}

inline bool pointer_on_device(const void* p) {
  hipPointerAttribute_t attr;
  std::memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

// rows that are not float4-addressable (dim % 4 != 0, or a misaligned base) still move as
// 16-byte vectors at 4-byte alignment; `allowed`: the kernels that will see
// this context know the mode (the LFU / FIFO install does not)
inline void set_odd4(Ctx& c, size_t dim, bool allowed) {
  c.dim = static_cast<uint32_t>(dim);
  c.odd4 = 0;
  c.tail = 0;
  if (c.vec4 || !allowed || dim < 8) return;
  static const bool enabled = [] {
    const char* v = std::getenv("GNNFLOW_GATHER_ODD_VEC4");   // tuning / tests; 0 = scalar rows
    return !(v && std::atoi(v) == 0);
  }();
  if (!enabled) return;
  c.odd4 = 1;
  c.dimv = static_cast<uint32_t>((dim + 3) / 4);   // the last vector overlaps its neighbour
  c.tail = static_cast<uint32_t>(dim % 4);
}

inline bool vec4_ok(size_t dim, const void* a, const void* b, const void* c) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return dim % 4 == 0 && al(a) && al(b) && al(c);
}

// LRU caches of at least this many slots are kept as a queue (O(block rows) updates): a
// 30 k-row fetch with update costs 30 / 42 / 77 / 152 / 352 us in the list form at 0.13 / 1 / 4
// / 16 / 40 M slots and 45 / 45 / 49 us in the queue form at 4 / 16 / 40 M
// (profiles/r02_lru_capacity_sweep.jsonl)
// (round 5, 30 k-row blocks, one-launch list update: 23.9 / 36.3 / 48.5 / 68.8 us per fetch at
// 0.13 / 0.5 / 1 / 2 M slots; row-parallel queue form 41.6 / 36.8 / 35.0 / 34.9 / 34.5 / 37.7 at
// 0.13 / 0.5 / 1 / 2 / 16 / 40 M: they cross at ~0.5 M slots; profiles/r05_lru_capacity_sweep.txt)
inline size_t queue_min_capacity() {
  const char* v = std::getenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY");   // tuning / tests
  return v ? static_cast<size_t>(std::atoll(v)) : (size_t{1} << 19);
}

// chunks of kRowTile queue entries the victim walk covers behind the head: twice the rows
// of the block (at most that many victims are needed) + 2
inline size_t victim_chunks(size_t n) { return (2 * n + kRowTile - 1) / kRowTile + 2; }

// list form: blocks that need more list tiles than this for their victims fall back to the
// one-workgroup walk (kMaxStageTiles: the install kernel keeps the tile prefix in LDS)
inline size_t max_stage_tiles() {
  static const size_t v = [] {
    const char* e = std::getenv("GNNFLOW_LRU_STAGE_TILES_MAX");   // tuning / tests; 0: always walk
    const size_t x = e ? static_cast<size_t>(std::atoll(e)) : kMaxStageTiles;
    return std::min<size_t>(x, kMaxStageTiles);
  }();
  return v;
}

// entries of the victim staging arrays: list form — the tiles at the front of the list (at most
// the whole list, at most kMaxStageTiles); queue form — the chunks behind the head
inline size_t stage_entries(size_t n, size_t capacity) {
  const size_t list_tiles = (capacity + kRowTile - 1) / kRowTile;
  const size_t list_form = std::min<size_t>(std::min(list_tiles, victim_chunks(n)), kMaxStageTiles);
  return std::max(list_form, victim_chunks(n)) * kRowTile + n;
}

// fused list update: entries of the staged front tiles (the ids they hold; the slots share
// the v_slot / v_pos arrays): at most min(capacity, rows) list positions, in whole tiles
inline size_t fuse_stage_entries(size_t n, size_t capacity) {
  return (std::min(n, capacity) / kFuseTile + 2) * kFuseTile;
}

inline bool lru_fused_enabled() {
  const char* e = std::getenv("GNNFLOW_LRU_FUSED");   // 0: list scan + list install (A/B, tests)
  return e ? std::atoi(e) != 0 : true;
}

// bitmap over the queue positions, in whole tiles of kRowTile words (+ one tile)
inline size_t qbits_bytes(size_t queue_cap) {
  const size_t words = (queue_cap + 64 + 31) / 32;
  return ((words + kRowTile - 1) / kRowTile + 1) * kRowTile * sizeof(uint32_t);
}

// rows per wave: 64 for big blocks; fewer for small ones so the block still spreads
// over >= 1024 waves (4 per CU)
inline uint32_t pick_tile_rows(size_t n) {
  static const int forced = [] {
    const char* v = std::getenv("GNNFLOW_GATHER_TILE_ROWS");   // tuning / tests
    return v ? std::atoi(v) : 0;
  }();
  if (forced >= 1 && forced <= 64) return static_cast<uint32_t>(forced);
  // measured on the batch-600 blocks (10k-30k rows): 16 rows per wave beats both 4 (more,
  // shorter waves: 17.8 us/launch) and 32 (16.7 us) at 13.8 us; aim for >= 1024 waves, but
  // never below 16 rows — a 16-row tile of 172-d rows is one trip of 11 loads per lane, and
  // the replay's mid-size blocks (5-16 k rows) ran at 8 rows per wave before: whole replay
  // 14.1-14.3 -> 13.6 us per launch (round 4, same box; 8 rows everywhere: 20.2 us)
  uint32_t t = 64;
  while (t > 16 && (n + t - 1) / t < 1024) t >>= 1;
  return t;
}

inline uint32_t pick_inflight() {
  static const int v = [] {
    const char* e = std::getenv("GNNFLOW_GATHER_INFLIGHT");   // tuning / tests
    return e ? std::atoi(e) : 12;   // a 16-row tile of 172-d rows = 11 loads per lane: one trip
  }();
  return static_cast<uint32_t>(v);
}

inline unsigned gather_grid_for(size_t n, uint32_t tile_rows) {
  const size_t waves = (n + tile_rows - 1) / tile_rows;
  return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((waves + 3) / 4, 1024)));
}

// Side stream of the calling host thread for rounds that update caches of two forms (created on
// first use only: an extra stream shifts the process's hardware-queue mapping).
struct RoundFork {
  hipStream_t side = nullptr;
  hipEvent_t begin = nullptr, end = nullptr;
  int device = -1;
  ~RoundFork() {
    if (side) (void)hipStreamDestroy(side);
    if (begin) (void)hipEventDestroy(begin);
    if (end) (void)hipEventDestroy(end);
  }
};
inline RoundFork& round_fork() {
  static thread_local RoundFork f;
  return f;
}
inline bool fork_round(hipStream_t stream, hipStream_t* side) {
  static const bool enabled = [] {
    const char* v = std::getenv("GNNFLOW_LRU_FORK");   // tuning / tests; 0: one stream
    return !(v && std::atoi(v) == 0);
  }();
  if (!enabled) return false;
  RoundFork& f = round_fork();
  int dev = 0;
  GF_HIP(hipGetDevice(&dev));
  if (f.side && f.device != dev) return false;   // one device per host thread in practice
  if (!f.side) {
    GF_HIP(hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking));
    GF_HIP(hipEventCreateWithFlags(&f.begin, hipEventDisableTiming));
    GF_HIP(hipEventCreateWithFlags(&f.end, hipEventDisableTiming));
    f.device = dev;
  }
  GF_HIP(hipEventRecord(f.begin, stream));
  GF_HIP(hipStreamWaitEvent(f.side, f.begin, 0));
  *side = f.side;
  return true;
}
inline void fork_done() { GF_HIP(hipEventRecord(round_fork().end, round_fork().side)); }
inline void join_round(hipStream_t stream) { GF_HIP(hipStreamWaitEvent(stream, round_fork().end, 0)); }

// Issues one round: the gather for every context, then (if any context updates its cache)
// the four bookkeeping launches.
void launch_round(Round& r, hipStream_t stream) {
  if (r.count == 0) return;
  unsigned ggrid = 1;
  size_t max_n = 0, max_cap = 0, max_tiles = 0;
  bool any_update = false;
  for (int i = 0; i < r.count; ++i) {
    const Ctx& c = r.c[i];
    GF_REQUIRE(c.n < 0x7FFFFFFFull, "gather: more than 2^31-1 rows in one block");
    ggrid = std::max(ggrid, gather_grid_for(c.n, c.tile_rows));
    if (c.update) {
      any_update = true;
      max_n = std::max<size_t>(max_n, c.n);
      max_cap = std::max<size_t>(max_cap, c.capacity);
      max_tiles = std::max<size_t>(max_tiles, (c.capacity + kTile - 1) / kTile);
    }
  }
  {
    bool lean = true, staged = false, direct = true;
    for (int i = 0; i < r.count; ++i) {
      const Ctx& c = r.c[i];
      lean = lean && c.vec4 && !c.qmode && c.inflight >= 12;
      staged = staged || c.pmap != nullptr;
      direct = direct && !c.cache_buf && !c.miss_rows && !c.remap && !c.pmap;
    }
    if (lean && direct) {
      // Two workgroups per CU, not three: the dispatcher hands workgroups to the 256 CUs round
      // robin, and the launch ends with the CUs that received a third one (~2.3 us per further
      // workgroup: profiles/r06_gather_hop_trace.txt).  If slightly larger tiles — still one trip
      // of loads — bring the round down to 512 workgroups, take them.
      static const bool fit = [] {
        const char* v = std::getenv("GNNFLOW_GATHER_FIT_CUS");   // A/B
        return !(v && std::atoi(v) == 0);
      }();
      auto wgs = [&](uint32_t t) {
        size_t total = 0;
        for (int i = 0; i < r.count; ++i) total += ((r.c[i].n + t - 1) / t + 3) / 4;
        return total;
      };
      uint32_t t0 = 0, dimv = 1;
      bool same = true;
      for (int i = 0; i < r.count; ++i) {
        if (r.c[i].n == 0) continue;
        if (t0 == 0) t0 = r.c[i].tile_rows;
        same = same && (r.c[i].tile_rows == t0 || r.c[i].n <= 4u * r.c[i].tile_rows);
        dimv = std::max(dimv, r.c[i].dimv);
      }
      const uint32_t t_max = std::min<uint32_t>(64u, 13u * 64u / dimv);
      if (fit && same && t0 == 16 && wgs(t0) > 512 && t_max > t0) {
        uint32_t t = t0 + 1;
        while (t < t_max && wgs(t) > 512) ++t;
        if (wgs(t) <= 512) {
          ggrid = 1;
          for (int i = 0; i < r.count; ++i) {
            if (r.c[i].n > 4u * r.c[i].tile_rows) r.c[i].tile_rows = t;
            ggrid = std::max(ggrid, gather_grid_for(r.c[i].n, r.c[i].tile_rows));
          }
        }
      }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (lean) {
      auto* kernel = staged ? gather_rows_staged_kernel
                   : direct ? gather_rows_kernel : gather_rows_mirror_kernel;
      uint32_t first[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < kMaxCtx; ++i)
        first[i + 1] = first[i] + (i < r.count && r.c[i].n ? gather_grid_for(r.c[i].n, r.c[i].tile_rows) : 0u);
      const uint4 f = make_uint4(first[1], first[2], first[3], first[4]);
      const unsigned total = std::max(1u, first[4]);
      if (profile_begin(kProfGather, &e0, &e1)) {
        // the events ride on the dispatch itself: its begin / end timestamps
        hipExtLaunchKernelGGL(kernel, dim3(total), dim3(kThreads), 0, stream, e0, e1, 0, f, r);
        profile_end(kProfGather, e0, e1);
      } else {
        kernel<<<dim3(total), dim3(kThreads), 0, stream>>>(f, r);
      }
    } else {
      auto* kernel = gather_rows_any_kernel;
      if (profile_begin(kProfGather, &e0, &e1)) {
        hipExtLaunchKernelGGL(kernel, dim3(ggrid, r.count), dim3(kThreads), 0, stream, e0, e1, 0, r);
        profile_end(kProfGather, e0, e1);
      } else {
        kernel<<<dim3(ggrid, r.count), dim3(kThreads), 0, stream>>>(r);
      }
    }
    GF_HIP(hipGetLastError());
  }
  if (!any_update) return;
  // the LRU slot of the profile: ONE fused launch carries its own dispatch events (the clock
  // rocprofv3 reads: begin / end of the dispatch); several launches sit between two stream events
  bool only_fused = true;
  for (int i = 0; i < r.count; ++i)
    only_fused = only_fused && (!r.c[i].update || (r.c[i].policy == GF_CACHE_LRU && r.c[i].fused));
  ProfileScope ps(only_fused ? -1 : kProfLru, stream);
  size_t q_scan_blocks = 0, q_rows = 0, q_cap = 0, q_bit_tiles = 0, q_victim_blocks = 1;
  size_t q_inst_blocks = 0, qq_rows = 0;
  size_t h_n = 0, h_cap = 0, h_tiles = 0;
  size_t f_tiles = 0, f_rows = 0;
  bool forked = false;
  hipStream_t side = nullptr;
  for (int i = 0; i < r.count; ++i) {
    const Ctx& c = r.c[i];
    if (!c.update) continue;
    if (c.policy == GF_CACHE_LRU && c.fused) {
      f_tiles = std::max<size_t>(f_tiles, (c.capacity + kFuseTile - 1) / kFuseTile);
      f_rows = std::max<size_t>(f_rows, (c.n + c.fuse_rows - 1) / c.fuse_rows);
    } else if (c.policy == GF_CACHE_LRU) {
      const size_t row_tiles = (c.n + kLruRows - 1) / kLruRows;
      q_scan_blocks = std::max(q_scan_blocks, (row_tiles + c.tiles_per_wg - 1) / c.tiles_per_wg);
      q_rows = std::max<size_t>(q_rows, c.n);
      if (c.qmode) {
        // queue form: bitmap tiles of kBitTile words (32 queue positions per word; the tail
        // is below 1.5 * capacity + 64)
        const size_t bit_tiles = ((size_t{c.capacity} * 3 / 2 + 128) / 32 + kBitTile - 1) / kBitTile + 1;
        q_bit_tiles = std::max(q_bit_tiles, bit_tiles);
        q_victim_blocks = std::max<size_t>(q_victim_blocks, std::min<size_t>(c.v_chunks, 1024));
        qq_rows = std::max<size_t>(qq_rows, c.n);
      } else {
        q_cap = std::max<size_t>(q_cap, c.capacity);
        q_inst_blocks = std::max<size_t>(q_inst_blocks, (c.n + c.inst_rows - 1) / c.inst_rows);
      }
    } else {
      h_n = std::max<size_t>(h_n, c.n);
      h_cap = std::max<size_t>(h_cap, c.capacity);
      h_tiles = std::max<size_t>(h_tiles, (c.capacity + kTile - 1) / kTile);
    }
  }
  if (f_tiles) {   // LRU list form, one launch
    const unsigned cb = static_cast<unsigned>(std::min<size_t>(f_tiles, 1024));
    const unsigned rb = static_cast<unsigned>(std::max<size_t>(f_rows, 1));
    const unsigned wb = static_cast<unsigned>(std::min<size_t>(f_tiles, kFuseMaxTiles));
    // a round that also carries a queue-form (or two-launch) update — a small node cache beside a
    // GDELT-scale edge cache — runs this launch on a side stream, beside those launches: the
    // contexts are different caches, and both chains are bound by dependent accesses, not by CUs
    forked = q_rows != 0 && fork_round(stream, &side);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (only_fused && profile_begin(kProfLru, &e0, &e1)) {
      hipExtLaunchKernelGGL(lru_list_fused_kernel, dim3(cb + rb + wb, r.count), dim3(kWide), 0, stream,
                            e0, e1, 0, r, cb, rb, wb);
      profile_end(kProfLru, e0, e1);
    } else {
      lru_list_fused_kernel<<<dim3(cb + rb + wb, r.count), dim3(kWide), 0, forked ? side : stream>>>(
          r, cb, rb, wb);
    }
    GF_HIP(hipGetLastError());
    if (forked) fork_done();
  }
  if (q_rows) {   // LRU: list scan + list install
    const unsigned rb = static_cast<unsigned>(q_scan_blocks);
    // list workgroups: per kRowTile list entries of the list-form contexts (none: queue form
    // only); the install kernel's also append for the queue-form contexts
    const unsigned lb_list = static_cast<unsigned>(
        std::min<size_t>(std::max((q_cap + kRowTile - 1) / kRowTile, q_bit_tiles), 1024));
    // install: sub-tiles of kWide list entries per workgroup for the list-form contexts
    const unsigned lb_sub = static_cast<unsigned>(std::min<size_t>((q_cap + kWide - 1) / kWide, 1024));
    const unsigned lb = std::max<unsigned>(1, lb_sub);
    const unsigned vb = static_cast<unsigned>(q_victim_blocks);
    lru_list_scan_kernel<<<dim3(rb + lb_list + vb, r.count), dim3(kWide), 0, stream>>>(
        r, rb, lb_list, vb);
    if (q_bit_tiles) {
      lru_queue_walk_kernel<<<dim3(1, r.count), dim3(kWide), 0, stream>>>(r);
      const unsigned qb = static_cast<unsigned>(
          std::max<size_t>(1, std::min<size_t>((qq_rows + kQInst - 1) / kQInst, 16384)));
      lru_queue_install_kernel<<<dim3(qb, r.count), dim3(kQInst), 0, stream>>>(r);
    }
    if (q_inst_blocks) {
      const unsigned ib = static_cast<unsigned>(std::min<size_t>(q_inst_blocks, 4096));
      lru_list_install_kernel<<<dim3(ib + lb, r.count), dim3(kWide), 0, stream>>>(r, ib, lb);
    }
    GF_HIP(hipGetLastError());
    if (forked) join_round(stream);
  }
  if (!h_cap) return;
  max_n = h_n; max_cap = h_cap; max_tiles = h_tiles;
  const unsigned slot_grid = static_cast<unsigned>(
      std::max<size_t>(1, std::min<size_t>((max_cap + 4 * kWide - 1) / (4 * kWide), 1024)));
  const unsigned both_grid = static_cast<unsigned>(std::max<size_t>(
      1, std::min<size_t>((std::max(max_n, max_cap) + 4 * kWide - 1) / (4 * kWide), 1024)));
  const unsigned tile_grid =
      static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(max_tiles, 2048)));
  const unsigned scan_blocks = static_cast<unsigned>(
      std::max<size_t>(1, std::min<size_t>((max_n + kRowTile - 1) / kRowTile, kMaxRowTiles)));
  lru_scan_hist_kernel<<<dim3(scan_blocks + slot_grid, r.count), dim3(kWide), 0, stream>>>(
      r, scan_blocks);
  lru_rank_tile_kernel<<<dim3(std::min(std::max(both_grid, tile_grid), 1024u), r.count),
                         dim3(kWide), 0, stream>>>(r);
  lru_install_kernel<<<dim3(tile_grid, r.count), dim3(kTile), 0, stream>>>(r);
  GF_HIP(hipGetLastError());
}

Ctx plain_ctx(const float* feats, size_t num_rows, size_t dim, const int64_t* ids, size_t n,
              float* out) {
  Ctx c;
  std::memset(&c, 0, sizeof(c));
  c.ids = ids;
  c.n = static_cast<uint32_t>(n);
  c.vec4 = vec4_ok(dim, feats, out, out) ? 1 : 0;
  c.dimv = static_cast<uint32_t>(c.vec4 ? dim / 4 : dim);
  set_odd4(c, dim, true);
  c.tile_rows = pick_tile_rows(n);
  c.inflight = pick_inflight();
  c.out = out;
  c.feats = feats;
  c.num_ids = num_rows;
  return c;
}

}  // namespace

void gather_rows(const float* d_feats, size_t num_rows, size_t dim, const int64_t* d_ids,
                 size_t n, float* d_out, int device, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_feats && d_ids && d_out, "gather_rows: null pointer");
  GF_REQUIRE(dim > 0, "gather_rows: dim must be positive");
  DeviceGuard dg(device);
  Round r;
  r.count = 1;
  r.c[0] = plain_ctx(d_feats, num_rows, dim, d_ids, n, d_out);
  launch_round(r, stream);
}

// Several cache-free gathers that share one id list (TGN memory: four tables), one launch.
void gather_rows_multi(const float* const* tables, const size_t* dims, float* const* outs,
                       size_t num_tables, size_t num_rows, const int64_t* d_ids, size_t n,
                       int device, hipStream_t stream) {
  if (n == 0 || num_tables == 0) return;
  GF_REQUIRE(num_tables <= static_cast<size_t>(kMaxCtx), "gather_rows_multi: too many tables");
  GF_REQUIRE(tables && dims && outs && d_ids, "gather_rows_multi: null pointer");
  DeviceGuard dg(device);
  Round r;
  r.count = static_cast<int>(num_tables);
  for (size_t t = 0; t < num_tables; ++t) {
    GF_REQUIRE(tables[t] && outs[t] && dims[t] > 0, "gather_rows_multi: bad table");
    r.c[t] = plain_ctx(tables[t], num_rows, dims[t], d_ids, n, outs[t]);
  }
  launch_round(r, stream);
}

FeatureCache::FeatureCache(size_t num_ids, size_t capacity, size_t dim, const float* d_feats,
                           int device)
    : num_ids_(num_ids), capacity_(capacity), dim_(dim), feats_(d_feats), device_(device) {
  GF_REQUIRE(dim > 0, "cache: dim must be positive");
  GF_REQUIRE(d_feats != nullptr, "cache: null feature table");
  GF_REQUIRE(capacity <= num_ids, "cache: capacity larger than the id space");
  GF_REQUIRE(capacity < 0x7FFFFFFFull, "cache: capacity must be < 2^31");
  DeviceGuard dg(device_);
  {
    // GNNFLOW_LRU_FUSE_SPINS (tests): the polls' budget before a waiter recomputes the value
    // itself — 0 sends EVERY look-back of the one-launch LRU update through its fallback
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), device_) == done.end()) {
      done.push_back(device_);
      if (const char* v = std::getenv("GNNFLOW_LRU_FUSE_SPINS")) {
        const uint32_t spins = static_cast<uint32_t>(std::atoll(v));
        GF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_spins), &spins, sizeof(spins)));
      }
    }
  }
  table_on_device_ = pointer_on_device(d_feats);
  buffer_.reserve(std::max<size_t>(capacity * dim * sizeof(float), 16), 0, nullptr, true);
  map_.reserve(std::max<size_t>(num_ids * sizeof(int32_t), 16));
  slot_id_.reserve(std::max<size_t>(capacity * sizeof(int64_t), 16));
  stamp_.reserve(std::max<size_t>(capacity * sizeof(uint32_t), 16));
  touched_.reserve(std::max<size_t>(capacity * sizeof(uint32_t), 16));
  state_.reserve(kRing * sizeof(Counters), 0, nullptr, true);
  fifo_ptr_.reserve(16);
  qstate_.reserve(sizeof(QueueState));
  rewind_fifo(nullptr);
  cache_fill_kernel<<<dim3(1024), dim3(256), 0, nullptr>>>(
      map_.as<int32_t>(), num_ids_, slot_id_.as<int64_t>(), stamp_.as<uint32_t>(),
      touched_.as<uint32_t>(), capacity_, 0, 0u);
  GF_HIP(hipGetLastError());
  init_queue(nullptr);
  GF_HIP(hipMemsetAsync(buffer_.data(), 0, buffer_.bytes(), nullptr));
  GF_HIP(hipMemsetAsync(state_.data(), 0, state_.bytes(), nullptr));
  GF_HIP(hipStreamSynchronize(nullptr));
}

FeatureCache::~FeatureCache() {
  for (hipEvent_t e : stage_events_)
    if (e) (void)hipEventDestroy(e);
}

// ---- staging ring, host side ---------------------------------------------------------------
static std::atomic<uint64_t> g_stage_stream_waits{0};

void FeatureCache::set_staging(size_t generations, size_t rows_per_generation) {
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());   // (a configuration call: nothing of this cache is in flight after it)
  if (generations == 0 || rows_per_generation == 0) {
    stage_gens_ = stage_cap_ = 0;
    ring_.release();
    pmap_.release();
    region_rows_.release();
    region_ids_.release();
    synced_gen_ = 0;
    return;
  }
  GF_REQUIRE(!table_on_device_, "staging ring: the feature table is already in device memory");
  GF_REQUIRE(generations >= 2 * kStageAhead && generations <= 64 &&
                 (generations & (generations - 1)) == 0,
             "staging ring: generations must be a power of two in 8..64");
  GF_REQUIRE(rows_per_generation < (size_t{1} << 31), "staging ring: too many rows per generation");
  stage_gens_ = static_cast<uint32_t>(generations);
  stage_cap_ = static_cast<uint32_t>(rows_per_generation);
  ring_.release();
  ring_.reserve(generations * rows_per_generation * dim_ * sizeof(float) + 16);
  pmap_.reserve(std::max<size_t>(2 * num_ids_ * sizeof(unsigned long long), 16));
  region_rows_.reserve(64 * sizeof(uint32_t) + 64);   // + rows pulled, + rows read from the host, + ticket
  region_ids_.release();
  region_ids_.reserve(rows_per_generation * sizeof(long long) + 16);
  progress_.reserve(64);
  *progress_.as<volatile uint32_t>() = 0;
  for (hipEvent_t& e : stage_events_)
    if (!e) GF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  GF_HIP(hipMemset(pmap_.data(), 0, pmap_.bytes()));
  GF_HIP(hipMemset(region_rows_.data(), 0, region_rows_.bytes()));
  gen_issued_ = 0;
  stage_reads_ = 0;
  stage_read_pending_ = false;
  synced_gen_ = 0;
  std::memset(gen_event_, 0, sizeof(gen_event_));
  std::memset(reads_at_gen_, 0, sizeof(reads_at_gen_));
}

void FeatureCache::invalidate_staging() {
  if (!staging()) return;
  // every entry staged so far falls out of every window a later launch accepts; the regions
  // the skipped generations would have used are simply never read
  gen_issued_ += stage_gens_ + 1;
  for (uint32_t& v : reads_at_gen_) v = stage_reads_;
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  GF_HIP(hipMemset(region_rows_.data(), 0, 64 * sizeof(uint32_t)));
  *progress_.as<volatile uint32_t>() = stage_reads_;   // (device idle: every launch has finished)
  synced_gen_ = gen_issued_;
  std::memset(gen_event_, 0, sizeof(gen_event_));
}

// diagnostics: stamps of the one-launch list update (gf_debug_lru_trace)
void FeatureCache::lru_trace_enable(bool on) {
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  if (!on) { trace_.release(); return; }
  trace_.reserve((kGatherTraceBase + 8 * kGatherTraceWgs) * sizeof(unsigned long long));
  GF_HIP(hipMemset(trace_.data(), 0, trace_.bytes()));
}
size_t FeatureCache::lru_trace_read(uint64_t* out, size_t capacity_words) {
  if (!trace_.data()) return 0;
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  const size_t words = std::min(capacity_words, trace_.bytes() / sizeof(unsigned long long));
  GF_HIP(hipMemcpy(out, trace_.data(), words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return words;
}

void FeatureCache::staging_state(uint64_t out[9]) {
  out[7] = static_cast<uint64_t>(stage_spin_us_);
  out[8] = g_stage_stream_waits.load(std::memory_order_relaxed);
  out[0] = stage_gens_;
  out[1] = stage_cap_;
  out[2] = gen_issued_;
  out[3] = stage_drops_;
  out[4] = 0;
  out[5] = staging() ? ring_.bytes() + pmap_.bytes() : 0;
  out[6] = 0;
  if (staging()) {
    DeviceGuard dg(device_);
    GF_HIP(hipDeviceSynchronize());
    unsigned long long v[2] = {0, 0};
    GF_HIP(hipMemcpy(v, region_rows_.as<uint32_t>() + 64, sizeof(v), hipMemcpyDeviceToHost));
    out[4] = v[0];
    out[6] = v[1];
  }
}

static inline bool stage_risk_enabled() {
  static const bool on = [] {
    const char* v = std::getenv("GNNFLOW_STAGE_AT_RISK");   // tuning / tests; 0: absent ids only
    return !(v && std::atoi(v) == 0);
  }();
  return on;
}

// The window of generations a launch may read when `issued` is the newest one: region g & mask
// is rewritten by generation g + G, and up to kStageAhead newer generations may be pulled while
// the launch runs.
static inline uint32_t stage_window_lo(uint32_t issued, uint32_t gens, uint32_t ahead) {
  const uint32_t keep = gens - ahead;   // generations issued, issued - 1, ..., issued - keep + 1
  return issued >= keep ? issued - keep + 1u : 1u;
}

// The next generation X rewrites the region of generation X - G.  Launches that may read that
// region were enqueued before generation X - kStageAhead was issued; they are known to have
// finished once a LATER ring-reading launch has started (it stores the number of such launches
// before it in `progress`).  The issuing thread waits for that — it is what keeps the host from
// running arbitrarily far ahead of the fetch stream, where a prefetch would see a cache state
// many updates old — and gives the generation up after GNNFLOW_STAGE_SPIN_US (a hint may be
// dropped; waiting for ever may not: nothing guarantees that the caller fetches again).
bool FeatureCache::stage_advance() {
  const uint32_t next = gen_issued_ + 1u;
  if (next > kStageAhead) {
    const uint32_t need = reads_at_gen_[(next - kStageAhead) & 63u];
    volatile uint32_t* progress = progress_.as<volatile uint32_t>();
    if (static_cast<int32_t>(*progress - need) < 0) {
      static const long spin_us = [] {
        const char* v = std::getenv("GNNFLOW_STAGE_SPIN_US");
        return v ? std::atol(v) : 20000L;
      }();
      bool ok = false;
      const auto t_spin = std::chrono::steady_clock::now();
      if (stage_reads_ != need && spin_us > 0) {   // (== : no later launch exists that could report)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0;; ++i) {
          if (static_cast<int32_t>(*progress - need) >= 0) { ok = true; break; }
          __builtin_ia32_pause();
          if ((i & 255u) == 255u &&
              std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
      }
      stage_spin_us_ += std::chrono::duration<double, std::micro>(
                            std::chrono::steady_clock::now() - t_spin).count();
      if (!ok) { ++stage_drops_; return false; }
    }
  }
  gen_issued_ = next;
  reads_at_gen_[next & 63u] = stage_reads_;
  return true;
}

// the fetch stream waits for the prefetches issued so far (one wait per distinct event)
// The newest generation a fetch issued now depends on: all but the `stage_lag_` newest.
uint32_t FeatureCache::stage_hi() const {
  return gen_issued_ > stage_lag_ ? gen_issued_ - stage_lag_ : 0u;
}

// The fetch stream waits for the pulls this fetch depends on — the generations up to stage_hi(),
// i.e. for the newest of them (the pull stream runs them in order).  An event that has completed
// by now — the usual case when the loop announces a batch two steps ahead of its fetch
// (gf_cache_set_staging_lag) — costs one query; one that has not is handed to the stream, whose
// wait for another queue's signal takes 12-20 us to resolve.
void FeatureCache::stage_sync(hipStream_t stream, hipEvent_t* seen, int* num_seen) {
  const uint32_t hi = stage_hi();
  if (!staging() || hi == 0 || hi <= synced_gen_) return;
  synced_gen_ = hi;
  hipEvent_t ev = gen_event_[hi % kStageEvents];
  if (!ev) return;
  for (int i = 0; i < *num_seen; ++i)
    if (seen[i] == ev) return;
  seen[(*num_seen)++] = ev;
  const hipError_t q = hipEventQuery(ev);
  if (q == hipSuccess) return;
  if (q != hipErrorNotReady) GF_HIP(q);
  (void)hipGetLastError();
  g_stage_stream_waits.fetch_add(1, std::memory_order_relaxed);
  GF_HIP(hipStreamWaitEvent(stream, ev, 0));
}

// Context of one block for the generation just taken (stage_advance).
bool FeatureCache::stage_begin(void* stage_ctx_out, const int64_t* d_ids, size_t n, bool cached) {
  StageCtx& c = *static_cast<StageCtx*>(stage_ctx_out);
  std::memset(&c, 0, sizeof(c));
  c.ids = d_ids;
  c.n = static_cast<uint32_t>(n);
  c.map = (cached && capacity_) ? map_.as<int32_t>() : nullptr;
  c.num_ids = num_ids_;
  c.pmap = pmap_.as<unsigned long long>();
  c.region_rows = region_rows_.as<uint32_t>();
  c.region_ids = region_ids_.as<long long>();
  c.gen = gen_issued_;
  // (an id staged in one of the two oldest readable generations is staged again: its fetch is
  // issued one to three generations from now, when those have left the window)
  c.lo = std::min(gen_issued_, stage_window_lo(gen_issued_, stage_gens_, kStageAhead) + kStageAhead - 1u);
  c.mask = stage_gens_ - 1u;
  c.cap = stage_cap_;
  // (only a cache that kStageAhead blocks of this size can turn over: for a larger one the
  // entries at the front of the order are rarely among a block's hits, and the rule pulled 370
  // rows per step for the headline's edge cache — 134 k slots, 9.5 k-row blocks — to save 4)
  if (c.map && policy_ == GF_CACHE_LRU && stage_risk_enabled() &&
      capacity_ <= size_t{kStageAhead} * n) {
    c.qpos = qpos_.as<uint32_t>();
    c.qstate = queue_form_ ? qstate_.as<QueueState>() : nullptr;
    c.risk = static_cast<uint32_t>(capacity_);
  }
  return true;
}

// ... and the pull of what its blocks claimed
void FeatureCache::stage_pull(void* pull_job_out) {
  PullJob& j = *static_cast<PullJob*>(pull_job_out);
  const uint32_t region = gen_issued_ & (stage_gens_ - 1u);
  j.ids = region_ids_.as<long long>();
  j.feats = feats_;
  j.dst = ring_.as<float>() + static_cast<uint64_t>(region) * stage_cap_ * dim_;
  j.region_rows = region_rows_.as<uint32_t>() + region;
  j.next_rows = region_rows_.as<uint32_t>() + ((gen_issued_ + 1u) & (stage_gens_ - 1u));
  j.pulled = reinterpret_cast<unsigned long long*>(region_rows_.as<uint32_t>() + 64);
  j.cap = stage_cap_;
  j.dim = static_cast<uint32_t>(dim_);
  j.vec4 = vec4_ok(dim_, feats_, ring_.data(), ring_.data()) ? 1u : 0u;
}

void FeatureCache::stage_fill(void* ctx_out) {
  const uint32_t hi = stage_hi();
  if (!staging() || hi == 0) return;
  Ctx& c = *static_cast<Ctx*>(ctx_out);
  if (c.miss_rows || c.remap) return;
  c.pmap = pmap_.as<unsigned long long>();
  c.ring = ring_.as<float>();
  // (the window's lower end follows the newest generation ISSUED: that one's successors are the
  // ones that may overwrite regions while this launch runs)
  c.st_lo = stage_window_lo(gen_issued_, stage_gens_, kStageAhead);
  if (hi < c.st_lo) return;
  c.st_span = hi - c.st_lo;
  c.st_mask = stage_gens_ - 1u;
  c.st_cap = stage_cap_;
  c.progress = progress_.as<uint32_t>();
  c.progress_val = stage_reads_;
  c.st_fallback = reinterpret_cast<unsigned long long*>(region_rows_.as<uint32_t>() + 66);
  stage_read_pending_ = true;
}

void FeatureCache::stage_round_done() {
  if (stage_read_pending_) {
    ++stage_reads_;
    stage_read_pending_ = false;
  }
}

// Cache.init_cache (cache.py:175-195) / LRUCache.reset (lru_cache.py:91-105)
void FeatureCache::init(hipStream_t stream) {
  DeviceGuard dg(device_);
  cache_fill_kernel<<<dim3(1024), dim3(256), 0, stream>>>(
      map_.as<int32_t>(), num_ids_, slot_id_.as<int64_t>(), stamp_.as<uint32_t>(),
      touched_.as<uint32_t>(), capacity_, 1,
      policy_ == GF_CACHE_LFU ? 1u : 0u);   // LFUCache.init_cache: count += 1 (lfu_cache.py:80-84)
  GF_HIP(hipGetLastError());
  epoch_ = 0;
  rewind_fifo(stream);
  init_queue(stream);
  if (capacity_ && mirror_)
    GF_HIP(hipMemcpyAsync(buffer_.data(), feats_, capacity_ * dim_ * sizeof(float),
                          hipMemcpyDefault, stream));
}

// ---- LRU list, host side -----------------------------------------------------------------
// slot order: the order of a freshly initialised cache (every `count` equal)
void FeatureCache::init_queue(hipStream_t stream) {
  if (policy_ != GF_CACHE_LRU) return;
  queue_form_ = capacity_ >= queue_min_capacity();
  queue_cap_ = queue_form_ ? capacity_ + capacity_ / 2 + 64 : capacity_;
  const size_t bytes = (queue_cap_ + 16) * sizeof(uint32_t);   // + one 16-byte vector past the end
  queue_.reserve(bytes, 0, stream);
  queue_alt_.reserve(bytes, 0, stream);
  if (capacity_) {
    list_fill_kernel<<<dim3(1024), dim3(256), 0, stream>>>(
        queue_.as<uint32_t>(), 0u, static_cast<uint32_t>(capacity_), nullptr, 0u);
    GF_HIP(hipGetLastError());
  }
  const QueueState qs{0u, 0u, 0u, static_cast<uint32_t>(capacity_), 0u, 0u};
  GF_HIP(hipMemcpyAsync(qstate_.data(), &qs, sizeof(qs), hipMemcpyHostToDevice, stream));
  GF_HIP(hipStreamSynchronize(stream));   // qs is a stack variable
  tail_bound_ = capacity_;
  qpos_.reserve(std::max<size_t>(capacity_, 4) * sizeof(uint32_t), 0, stream);
  if (queue_form_) {
    GF_REQUIRE(queue_cap_ < (size_t{1} << 30), "LRU queue form: more than 2^30 queue positions");
    wsnap_.reserve(2 * qbits_bytes(queue_cap_), 0, stream);
    qbits_.reserve(qbits_bytes(queue_cap_), 0, stream);
    GF_HIP(hipMemsetAsync(qbits_.data(), 0, qbits_bytes(queue_cap_), stream));
    const size_t tiles = (queue_cap_ + kRowTile - 1) / kRowTile + 1;
    const size_t groups = (tiles + kQGroup - 1) / kQGroup + 1;
    compact_.reserve(align_up(tiles * (kRowTile / 64) * 8, 256) + align_up(tiles * 4, 256) +
                     align_up(groups * 4, 256) + 256, 0, stream);
  } else {
    wsnap_.release();
    qbits_.release();
    compact_.release();
  }
  index_queue(stream);
}

// qpos[] of a dense list
void FeatureCache::index_queue(hipStream_t stream) {
  if (!capacity_) return;
  lru_queue_index_kernel<<<dim3(2048), dim3(256), 0, stream>>>(
      queue_.as<uint32_t>(), queue_alt_.as<uint32_t>(), qstate_.as<QueueState>(),
      qpos_.as<uint32_t>(), static_cast<uint32_t>(capacity_));
  GF_HIP(hipGetLastError());
}

// Queue form: drops the dead entries (dense list in the other buffer, head = 0, tail = capacity)
void FeatureCache::compact_queue(hipStream_t stream) {
  if (!queue_form_ || tail_bound_ == capacity_) return;
  const size_t tiles = (tail_bound_ + kRowTile - 1) / kRowTile;
  const size_t groups = (tiles + kQGroup - 1) / kQGroup;
  char* p = compact_.as<char>();
  auto* live_bits = reinterpret_cast<unsigned long long*>(p);
  p += align_up(((queue_cap_ + kRowTile - 1) / kRowTile + 1) * (kRowTile / 64) * 8, 256);
  auto* tile_cnt = reinterpret_cast<uint32_t*>(p);
  p += align_up(((queue_cap_ + kRowTile - 1) / kRowTile + 1) * 4, 256);
  auto* group_sum = reinterpret_cast<uint32_t*>(p);
  p += align_up((((queue_cap_ + kRowTile - 1) / kRowTile + 1 + kQGroup - 1) / kQGroup + 1) * 4, 256);
  auto* st = reinterpret_cast<CompactState*>(p);
  GF_HIP(hipMemsetAsync(group_sum, 0, groups * 4, stream));
  const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(tiles, 2048)));
  lru_queue_compact_count_kernel<<<dim3(grid), dim3(kWide), 0, stream>>>(
      queue_.as<uint32_t>(), queue_alt_.as<uint32_t>(), qstate_.as<QueueState>(),
      qpos_.as<uint32_t>(), live_bits, tile_cnt, group_sum, st);
  lru_queue_compact_write_kernel<<<dim3(grid), dim3(kWide), 0, stream>>>(
      queue_.as<uint32_t>(), queue_alt_.as<uint32_t>(), qstate_.as<QueueState>(),
      qpos_.as<uint32_t>(), live_bits, tile_cnt, group_sum, st,
      static_cast<uint32_t>(capacity_));
  GF_HIP(hipGetLastError());
  tail_bound_ = capacity_;
  ++compactions_;
}

// FIFOCache.reset (fifo_cache.py:70-75) rewinds the rotation pointer and keeps the cached
// ids: with install-epoch stamps that is "all slots equally old" -> refill from slot 0.
// LFUCache.reset (lfu_cache.py:86-118) re-initialises and then zeroes the use counts.
void FeatureCache::reset_order(hipStream_t stream) {
  DeviceGuard dg(device_);
  if (capacity_) {
    GF_HIP(hipMemsetAsync(stamp_.data(), 0, capacity_ * sizeof(uint32_t), stream));
    GF_HIP(hipMemsetAsync(touched_.data(), 0, capacity_ * sizeof(uint32_t), stream));
  }
  epoch_ = 0;
  rewind_fifo(stream);
}

// cache_*_pointer = capacity - 1 (fifo_cache.py:63-69): the next refill starts at slot 0
void FeatureCache::rewind_fifo(hipStream_t stream) {
  GF_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(fifo_ptr_.data()),
                           capacity_ ? static_cast<int>(capacity_ - 1) : 0, 1, stream));
}

void FeatureCache::set_policy(int policy) {
  GF_REQUIRE(policy == GF_CACHE_LRU || policy == GF_CACHE_LFU || policy == GF_CACHE_FIFO,
             "cache: unknown replacement policy");
  if (policy == policy_) return;
  policy_ = policy;
  // the per-slot words mean different things per policy (LRU: queue position / hit claim)
  DeviceGuard dg(device_);
  if (capacity_) {
    GF_HIP(hipMemsetAsync(stamp_.data(), 0, capacity_ * sizeof(uint32_t), nullptr));
    GF_HIP(hipMemsetAsync(touched_.data(), 0, capacity_ * sizeof(uint32_t), nullptr));
  }
  init_queue(nullptr);
  GF_HIP(hipStreamSynchronize(nullptr));
  if (policy_ != GF_CACHE_LRU) {   // only LRU keeps a queue
    queue_.release();
    queue_alt_.release();
    qpos_.release();
    wsnap_.release();
    qbits_.release();
    compact_.release();
    queue_form_ = false;
  }
}

namespace {
__global__ void cache_install_ids_kernel(const int64_t* __restrict__ ids, uint64_t n,
                                         uint64_t num_ids, uint32_t dim,
                                         const float* __restrict__ feats,
                                         const float* __restrict__ rows, int32_t* map,
                                         int64_t* slot_id, float* buffer) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t s = wave; s < n; s += nwaves) {
    const int64_t id = ids[s];
    if (id < 0 || static_cast<uint64_t>(id) >= num_ids) continue;
    if (lane == 0) { map[id] = static_cast<int32_t>(s); slot_id[s] = id; }
    for (uint32_t c = lane; buffer && c < dim; c += 64)
      buffer[s * dim + c] = rows ? rows[s * dim + c] : feats[static_cast<uint64_t>(id) * dim + c];
  }
}
}  // namespace

// GNNLabStaticCache.init_cache (gnnlab_static_cache.py:87-168): slot i holds ids[i]
void FeatureCache::init_ids(const int64_t* d_ids, size_t n, hipStream_t stream,
                            const float* d_rows) {
  GF_REQUIRE(n <= capacity_, "cache: more ids than slots");
  GF_REQUIRE(d_ids != nullptr || n == 0, "cache: null id list");
  DeviceGuard dg(device_);
  cache_fill_kernel<<<dim3(1024), dim3(256), 0, stream>>>(
      map_.as<int32_t>(), num_ids_, slot_id_.as<int64_t>(), stamp_.as<uint32_t>(),
      touched_.as<uint32_t>(), capacity_, 0, 0u);
  epoch_ = 0;
  rewind_fifo(stream);
  init_queue(stream);
  if (n) {
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((n + 3) / 4, 4096));
    cache_install_ids_kernel<<<dim3(grid), dim3(256), 0, stream>>>(
        d_ids, n, num_ids_, static_cast<uint32_t>(dim_), feats_, d_rows, map_.as<int32_t>(),
        slot_id_.as<int64_t>(), mirror_ ? buffer_.as<float>() : nullptr);
  }
  GF_HIP(hipGetLastError());
}

// Cache.resize (cache.py:197-221): grow the id space / capacity, keep the contents
void FeatureCache::resize(size_t new_num_ids, size_t new_capacity, const float* d_feats,
                          hipStream_t stream) {
  GF_REQUIRE(new_num_ids >= num_ids_ && new_capacity >= capacity_,
             "cache: resize can only grow");
  GF_REQUIRE(new_capacity <= new_num_ids && new_capacity < 0x7FFFFFFFull,
             "cache: invalid capacity");
  DeviceGuard dg(device_);
  if (d_feats) {
    feats_ = d_feats;
    table_on_device_ = pointer_on_device(d_feats);
    GF_REQUIRE(mirror_ || table_on_device_,
               "cache: a table in host memory needs the row mirror (gf_cache_set_row_mirror)");
  }
  if (policy_ == GF_CACHE_LRU && new_capacity > capacity_) compact_queue(stream);   // dense list
  if (new_num_ids > num_ids_) {
    DeviceBuffer nmap;
    nmap.reserve(new_num_ids * sizeof(int32_t));
    std::vector<int32_t> tail(new_num_ids - num_ids_, kAbsent);  // new ids start uncached
    GF_HIP(hipMemcpyAsync(nmap.data(), map_.data(), num_ids_ * sizeof(int32_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipMemcpyAsync(nmap.as<int32_t>() + num_ids_, tail.data(),
                          tail.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    GF_HIP(hipStreamSynchronize(stream));
    std::swap(map_, nmap);
  }
  if (new_capacity > capacity_) {
    if (mirror_)
      buffer_.reserve(new_capacity * dim_ * sizeof(float), capacity_ * dim_ * sizeof(float),
                      stream, true);
    DeviceBuffer nid, nst, ntc;
    nid.reserve(new_capacity * sizeof(int64_t));
    nst.reserve(new_capacity * sizeof(uint32_t));
    ntc.reserve(new_capacity * sizeof(uint32_t));
    std::vector<int64_t> empty_ids(new_capacity - capacity_, -1);
    GF_HIP(hipMemcpyAsync(nid.data(), slot_id_.data(), capacity_ * sizeof(int64_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipMemcpyAsync(nid.as<int64_t>() + capacity_, empty_ids.data(),
                          empty_ids.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    GF_HIP(hipMemsetAsync(nst.data(), 0, new_capacity * sizeof(uint32_t), stream));
    GF_HIP(hipMemcpyAsync(nst.data(), stamp_.data(), capacity_ * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipMemsetAsync(ntc.data(), 0, new_capacity * sizeof(uint32_t), stream));
    GF_HIP(hipMemcpyAsync(ntc.data(), touched_.data(), capacity_ * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, stream));
    GF_HIP(hipStreamSynchronize(stream));
    std::swap(slot_id_, nid);
    std::swap(stamp_, nst);
    std::swap(touched_, ntc);
  }
  const size_t old_capacity = capacity_;
  num_ids_ = new_num_ids;
  capacity_ = new_capacity;
  ws_rows_ = 0;   // tile arrays depend on the capacity
  if (staging()) {   // new table, more ids: the ring starts over
    const size_t gens = stage_gens_, rows = stage_cap_;
    if (table_on_device_) set_staging(0, 0);
    else set_staging(gens, rows);
  }
  if (policy_ == GF_CACHE_LRU && new_capacity > old_capacity) {
    // the new (empty) slots are the first to be refilled, in slot order: they go to the front
    // of the list, the old entries follow in their order
    QueueState qs;
    GF_HIP(hipMemcpyAsync(&qs, qstate_.data(), sizeof(qs), hipMemcpyDeviceToHost, stream));
    GF_HIP(hipStreamSynchronize(stream));
    DeviceBuffer& cur = (qs.parity & 1u) ? queue_alt_ : queue_;
    queue_form_ = new_capacity >= queue_min_capacity();
    queue_cap_ = queue_form_ ? new_capacity + new_capacity / 2 + 64 : new_capacity;
    DeviceBuffer na, nb;
    na.reserve((queue_cap_ + 16) * sizeof(uint32_t));
    nb.reserve((queue_cap_ + 16) * sizeof(uint32_t));
    list_fill_kernel<<<dim3(1024), dim3(256), 0, stream>>>(
        na.as<uint32_t>(), static_cast<uint32_t>(old_capacity),
        static_cast<uint32_t>(new_capacity - old_capacity), cur.as<uint32_t>(),
        static_cast<uint32_t>(old_capacity));
    GF_HIP(hipGetLastError());
    const QueueState fresh{0u, 0u, 0u, static_cast<uint32_t>(new_capacity), 0u, 0u};
    GF_HIP(hipMemcpyAsync(qstate_.data(), &fresh, sizeof(fresh), hipMemcpyHostToDevice, stream));
    GF_HIP(hipStreamSynchronize(stream));
    std::swap(queue_, na);
    std::swap(queue_alt_, nb);
    tail_bound_ = new_capacity;
    {
      DeviceBuffer np;
      np.reserve(new_capacity * sizeof(uint32_t));
      std::swap(qpos_, np);
    }
    if (queue_form_) {
      GF_REQUIRE(queue_cap_ < (size_t{1} << 30), "LRU queue form: more than 2^30 queue positions");
      DeviceBuffer nh, nc, nbits;
      nh.reserve(2 * qbits_bytes(queue_cap_));
      nbits.reserve(qbits_bytes(queue_cap_));
      GF_HIP(hipMemsetAsync(nbits.data(), 0, qbits_bytes(queue_cap_), stream));
      std::swap(qbits_, nbits);
      const size_t tiles = (queue_cap_ + kRowTile - 1) / kRowTile + 1;
      const size_t groups = (tiles + kQGroup - 1) / kQGroup + 1;
      nc.reserve(align_up(tiles * (kRowTile / 64) * 8, 256) + align_up(tiles * 4, 256) +
                 align_up(groups * 4, 256) + 256);
      std::swap(wsnap_, nh);
      std::swap(compact_, nc);
    }
    index_queue(stream);
    GF_HIP(hipStreamSynchronize(stream));
  }
}

void FeatureCache::reserve_workspace(size_t n, hipStream_t stream) {
  retired_.collect();
  if (n <= ws_rows_ && ws_.data()) return;
  ws_rows_ = std::max(ws_rows_, n);
  const size_t tiles = (capacity_ + kTile - 1) / kTile + 1;
  size_t bytes = (kBins1 + kBins2) * sizeof(uint32_t) + 4 * align_up(ws_rows_ * 4, 16) +
                 align_up(ws_rows_ * 8, 16) + 2 * align_up(tiles * 4, 16) +
                 align_up((kMaxRowTiles + 1) * 4, 16) + 64;
  if (policy_ == GF_CACHE_LRU)   // staged victims per list tile / queue chunk, chunk counts
    bytes += 2 * align_up(stage_entries(ws_rows_, capacity_) * 4, 256) +
             align_up((victim_chunks(ws_rows_) + 2) * 4, 256) + 256 +
             2 * align_up(fuse_stage_entries(ws_rows_, capacity_) * 8, 256);
  // Kernels already queued on `stream` may still use the old scratch: it is retired behind
  // an event on that stream and freed once the event has completed — a stream-ordered swap,
  // no device-wide stall when a larger block arrives mid-run.
  DeviceBuffer fresh;
  fresh.reserve(bytes, 0, stream);
  std::swap(ws_, fresh);
  retired_.retire(std::move(fresh), stream);
}

// Fills the device context of one block fetch and advances this cache's host-side state
// (epoch, counter ring).  The caller launches the round.
void FeatureCache::prepare(const int64_t* d_ids, size_t n, float* d_out, bool update,
                           uint32_t* d_stats, void* ctx_out, hipStream_t stream) {
  GF_REQUIRE(d_ids && d_out, "cache fetch: null pointer");
  reserve_workspace(n, stream);
  const size_t tiles = (capacity_ + kTile - 1) / kTile;
  Ctx& c = *static_cast<Ctx*>(ctx_out);
  std::memset(&c, 0, sizeof(c));
  char* p = ws_.as<char>();
  c.hist1 = reinterpret_cast<uint32_t*>(p);         p += kBins1 * sizeof(uint32_t);
  c.hist2 = reinterpret_cast<uint32_t*>(p);         p += kBins2 * sizeof(uint32_t);
  c.slot_of_row = reinterpret_cast<int32_t*>(p);    p += align_up(ws_rows_ * 4, 16);
  c.rep_flag = reinterpret_cast<uint32_t*>(p);      p += align_up(ws_rows_ * 4, 16);
  c.rep_rank = reinterpret_cast<uint32_t*>(p);      p += align_up(ws_rows_ * 4, 16);
  c.rep_row = reinterpret_cast<uint32_t*>(p);       p += align_up(ws_rows_ * 4, 16);
  c.rep_id = reinterpret_cast<int64_t*>(p);         p += align_up(ws_rows_ * 8, 16);
  c.tile_tie = reinterpret_cast<uint32_t*>(p);      p += align_up((tiles + 1) * 4, 16);
  c.tile_old = reinterpret_cast<uint32_t*>(p);      p += align_up((tiles + 1) * 4, 16);
  c.row_tile_sum = reinterpret_cast<uint32_t*>(p);  p += align_up((kMaxRowTiles + 1) * 4, 16);
  char* qscratch = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(p), 256));
  c.ids = d_ids;
  c.n = static_cast<uint32_t>(n);
  c.vec4 = vec4_ok(dim_, mirror_ ? buffer_.data() : feats_, feats_, d_out) ? 1 : 0;
  c.dimv = static_cast<uint32_t>(c.vec4 ? dim_ / 4 : dim_);
  c.inst_from_table = table_on_device_ ? 1 : 0;
  set_odd4(c, dim_, policy_ == GF_CACHE_LRU || !update || capacity_ == 0);
  c.tile_rows = pick_tile_rows(n);
  c.inflight = pick_inflight();
  c.out = d_out;
  c.feats = feats_;
  c.num_ids = num_ids_;
  c.map = capacity_ ? map_.as<int32_t>() : nullptr;
  c.cache_buf = mirror_ ? buffer_.as<float>() : nullptr;
  c.slot_id = slot_id_.as<int64_t>();
  c.stamp = stamp_.as<uint32_t>();
  c.touched = touched_.as<uint32_t>();
  c.capacity = static_cast<uint32_t>(capacity_);
  c.update = (update && capacity_ > 0) ? 1 : 0;
  c.policy = policy_;
  c.fifo_ptr = fifo_ptr_.as<uint32_t>();
  c.epoch_new = c.update ? ++epoch_ : epoch_;
  c.ctr = state_.as<Counters>() + (ring_pos_ % kRing);
  c.ctr_next = state_.as<Counters>() + ((ring_pos_ + 1) % kRing);
  ring_pos_++;
  c.stats = d_stats;
  if (c.update && policy_ == GF_CACHE_LRU) {
    const size_t row_tiles = (n + kLruRows - 1) / kLruRows;
    c.tiles_per_wg = static_cast<uint32_t>((row_tiles + kMaxRowTiles - 1) / kMaxRowTiles);
    c.inst_rows = n >= 65536 ? kWide : kInstRows;
    c.queue[0] = queue_.as<uint32_t>();
    c.queue[1] = queue_alt_.as<uint32_t>();
    c.qstate = qstate_.as<QueueState>();
    {
      const size_t stage = align_up(stage_entries(ws_rows_, capacity_) * 4, 256);
      c.v_slot = reinterpret_cast<uint32_t*>(qscratch);
      c.v_pos = reinterpret_cast<uint32_t*>(qscratch + stage);
      c.v_count = reinterpret_cast<uint32_t*>(qscratch + 2 * stage);
      const size_t vc = align_up((victim_chunks(ws_rows_) + 2) * 4, 256) + 256;
      const size_t fst = align_up(fuse_stage_entries(ws_rows_, capacity_) * 8, 256);
      c.v_old = reinterpret_cast<long long*>(qscratch + 2 * stage + vc);
      c.v_hold = reinterpret_cast<long long*>(qscratch + 2 * stage + vc + fst);
      // list form: the tiles at the front of the list that stage their entries
      const size_t list_tiles = (capacity_ + kRowTile - 1) / kRowTile;
      const size_t st = std::min(list_tiles, (2 * n + kRowTile - 1) / kRowTile + 2);
      c.stage_tiles = st <= max_stage_tiles() ? static_cast<uint32_t>(st) : 0u;
      static const uint32_t stage_min = [] {
        const char* e = std::getenv("GNNFLOW_LRU_STAGE_MIN_WANT");   // tuning / tests
        return e ? static_cast<uint32_t>(std::atoll(e)) : kStageMinWant;
      }();
      c.stage_min = stage_min;
      c.stage_hits = st == list_tiles ? 1 : 0;
    }
    c.qpos = qpos_.as<uint32_t>();
    if (queue_form_) {
      if (n <= capacity_ / 4 && victim_chunks(n) + 1 <= kMaxVChunks) {
        // appends at most n entries (#distinct hit slots + #victims <= rows)
        if (tail_bound_ + n > queue_cap_) compact_queue(stream);
        tail_bound_ += n;
        c.qmode = 1;
        c.qbits = qbits_.as<uint32_t>();
        c.wsnap = wsnap_.as<uint2>();
        const size_t bit_tiles = ((queue_cap_ + 64) / 32 + kBitTile - 1) / kBitTile + 1;
        c.q_group = static_cast<uint32_t>((bit_tiles + kMaxBitGroups - 1) / kMaxBitGroups);
        static const uint32_t forced_group = [] {
          const char* e = std::getenv("GNNFLOW_LRU_QUEUE_GROUP");   // tests: the > 89 M-slot path
          return e ? static_cast<uint32_t>(std::atoi(e)) : 0u;
        }();
        c.q_group = std::max(c.q_group, forced_group);
        c.v_chunks = static_cast<uint32_t>(victim_chunks(n));
        c.stage_tiles = 0;
      } else {
        // a block this large is cheaper in the list form, on the dense list (whose install
        // leaves qpos[] = the new list positions, which is what the queue form expects)
        compact_queue(stream);
        ++list_form_updates_;
      }
    }
    // list form in one launch (lru_list_fused_kernel) where its LDS tables and granule arrays fit
    // 256 block rows per row workgroup while that gives <= kFuseMaxRowWgs of them: the rows a
    // small cache installs belong to the FIRST representatives, i.e. to the first few row
    // workgroups, which then copy all of its rows (GDELT-scale node cache, 3 336 slots, 218 k-row
    // blocks: 57 us per update with 1 024 rows per workgroup — four workgroups copied 5.5 MB)
    // Without a row mirror nothing is copied, and 1 024 rows per workgroup keep the launch small
    // enough for every count and row workgroup to be resident from the start (1 024-thread
    // workgroups: one per CU; with 256 rows the headline's row workgroups entered 2.6 us into the
    // launch, behind the count workgroups — profiles/r06_lru_hop_trace.txt).
    static const bool wide_rows = [] {
      const char* v = std::getenv("GNNFLOW_LRU_FUSE_WIDE_ROWS");   // A/B
      return !(v && std::atoi(v) == 0);
    }();
    c.fuse_rows = (n > size_t{kInstRows} * kFuseMaxRowWgs || (wide_rows && !mirror_)) ? kWide : kInstRows;
    if (!c.qmode && lru_fused_enabled() &&
        (capacity_ + kFuseTile - 1) / kFuseTile <= kFuseMaxTiles &&
        (n + c.fuse_rows - 1) / c.fuse_rows <= kFuseMaxRowWgs) {
      if (!granules_.data()) {
        granules_.reserve((kFuseMaxTiles + kFuseMaxRowWgs) * sizeof(unsigned long long), 0, stream);
        GF_HIP(hipMemsetAsync(granules_.data(), 0, granules_.bytes(), stream));
      }
      if (fuse_tag_ == 0xFFFFFFFFu) {   // the tags wrap: no granule may carry one from last time round
        GF_HIP(hipMemsetAsync(granules_.data(), 0, granules_.bytes(), stream));
        fuse_tag_ = 0;
      }
      c.fused = 1;
      c.trace = trace_.data() ? trace_.as<unsigned long long>() : nullptr;
      c.fuse_tag = ++fuse_tag_;
      c.g_cnt = granules_.as<unsigned long long>();
      c.g_row = c.g_cnt + kFuseMaxTiles;
    }
  }
}

// One block of Cache.fetch_feature (cache.py:269-323 / :326-400)
void FeatureCache::fetch(const int64_t* d_ids, size_t n, float* d_out, bool update,
                         uint32_t* d_stats, hipStream_t stream) {
  if (n == 0) return;
  DeviceGuard dg(device_);
  Round r;
  r.count = 1;
  hipEvent_t seen[2];
  int num_seen = 0;
  stage_sync(stream, seen, &num_seen);
  prepare(d_ids, n, d_out, update, d_stats, &r.c[0], stream);
  stage_fill(&r.c[0]);
  launch_round(r, stream);
  stage_round_done();
}

// Cache(distributed=True): the slots of a block's ids, so that the caller can pull the
// missed rows from their owners ...
void FeatureCache::probe(const int64_t* d_ids, size_t n, int32_t* d_slot, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_ids && d_slot, "cache probe: null pointer");
  DeviceGuard dg(device_);
  const unsigned grid = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 4096));
  cache_probe_kernel<<<dim3(grid), dim3(256), 0, stream>>>(
      d_ids, n, capacity_ ? map_.as<int32_t>() : nullptr, num_ids_, d_slot);
  GF_HIP(hipGetLastError());
}

// ... and the block's fetch with those rows standing in for the local feature table
void FeatureCache::fetch_pulled(const int64_t* d_ids, size_t n, float* d_out, bool update,
                                uint32_t* d_stats, const float* d_miss_rows,
                                const uint32_t* d_miss_index, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_miss_rows && d_miss_index, "cache fetch: null pulled rows");
  GF_REQUIRE(mirror_, "cache fetch: pulled rows need the row mirror (gf_cache_set_row_mirror)");
  DeviceGuard dg(device_);
  Round r;
  r.count = 1;
  prepare(d_ids, n, d_out, update, d_stats, &r.c[0], stream);
  r.c[0].miss_rows = d_miss_rows;
  r.c[0].miss_index = d_miss_index;
  r.c[0].inst_from_table = 0;   // the missed rows are the pulled ones, not the local table's
  if (r.c[0].vec4 && (reinterpret_cast<uintptr_t>(d_miss_rows) & 15u)) {
    r.c[0].vec4 = 0;
    r.c[0].dimv = static_cast<uint32_t>(dim_);
    set_odd4(r.c[0], dim_, policy_ == GF_CACHE_LRU || !update || capacity_ == 0);
  }
  launch_round(r, stream);
}

void FeatureCache::gather_plain(const int64_t* d_ids, size_t n, float* d_out,
                                hipStream_t stream) {
  gather_rows(feats_, num_ids_, dim_, d_ids, n, d_out, device_, stream);
}

// All feature fetches of one fetch_feature() call (cache.py:255-413).  The node cache and
// the edge cache are independent, so round i carries the i-th node block AND the i-th edge
// block (the edge blocks must stay ordered: each sees the LRU state the previous one left);
// cache-free gathers ride in the first round.  Every round is 1 + 4 launches whatever the
// number of contexts in it.
void fetch_blocks(FeatureCache* node, FeatureCache* edge, const gf_fetch_desc* descs, size_t n,
                  hipStream_t stream) {
  GF_REQUIRE(descs != nullptr || n == 0, "fetch_blocks: null descriptors");
  std::vector<const gf_fetch_desc*> nodes, edges, plain;
  for (size_t i = 0; i < n; ++i) {
    const gf_fetch_desc& d = descs[i];
    GF_REQUIRE(d.kind >= 0 && d.kind <= 2, "fetch_blocks: bad kind");
    if (d.n == 0) continue;
    if (d.kind == 0) {
      GF_REQUIRE(node != nullptr, "fetch_blocks: node block without a node cache");
      nodes.push_back(&d);
    } else {
      GF_REQUIRE(edge != nullptr, "fetch_blocks: edge block without an edge cache");
      (d.kind == 1 ? edges : plain).push_back(&d);
    }
  }
  const int device = node ? node->device() : (edge ? edge->device() : 0);
  DeviceGuard dg(device);
  // size each cache's scratch for its largest block up front: no reallocation (and device
  // synchronisation) between the rounds of one fetch
  size_t max_node_rows = 0, max_edge_rows = 0;
  for (const gf_fetch_desc* d : nodes) max_node_rows = std::max(max_node_rows, d->n);
  for (const gf_fetch_desc* d : edges) max_edge_rows = std::max(max_edge_rows, d->n);
  if (node && max_node_rows) node->reserve_workspace(max_node_rows, stream);
  if (edge && max_edge_rows) edge->reserve_workspace(max_edge_rows, stream);
  {   // host-resident tables: rows pulled ahead of this fetch must have landed
    hipEvent_t seen[2];
    int num_seen = 0;
    if (node) node->stage_sync(stream, seen, &num_seen);
    if (edge) edge->stage_sync(stream, seen, &num_seen);
  }
  auto done = [&] {
    if (node) node->stage_round_done();
    if (edge) edge->stage_round_done();
  };
  size_t pi = 0;
  const size_t rounds = std::max(nodes.size(), edges.size());
  for (size_t i = 0; i < rounds; ++i) {
    Round r;
    r.count = 0;
    if (i < nodes.size()) {
      const gf_fetch_desc& d = *nodes[i];
      node->prepare(d.d_ids, d.n, d.d_out, d.update != 0, d.d_stats, &r.c[r.count], stream);
      node->stage_fill(&r.c[r.count++]);
    }
    if (i < edges.size()) {
      const gf_fetch_desc& d = *edges[i];
      edge->prepare(d.d_ids, d.n, d.d_out, d.update != 0, d.d_stats, &r.c[r.count], stream);
      edge->stage_fill(&r.c[r.count++]);
    }
    while (pi < plain.size() && r.count < kMaxCtx) {
      const gf_fetch_desc& d = *plain[pi++];
      r.c[r.count] = plain_ctx(edge->feats_, edge->num_ids_, edge->dim_, d.d_ids, d.n, d.d_out);
      edge->stage_fill(&r.c[r.count++]);
    }
    launch_round(r, stream);
    done();
  }
  while (pi < plain.size()) {   // cache-free gathers that did not fit into a round
    Round r;
    r.count = 0;
    while (pi < plain.size() && r.count < kMaxCtx) {
      const gf_fetch_desc& d = *plain[pi++];
      r.c[r.count] = plain_ctx(edge->feats_, edge->num_ids_, edge->dim_, d.d_ids, d.n, d.d_out);
      edge->stage_fill(&r.c[r.count++]);
    }
    launch_round(r, stream);
    done();
  }
}

// Cache.prefetch_feature: one staging generation per cache for the blocks a coming
// fetch_blocks(descs) will gather (feature_cache.hpp)
bool prefetch_blocks(FeatureCache* node, FeatureCache* edge, const gf_fetch_desc* descs, size_t n,
                     hipStream_t stream) {
  GF_REQUIRE(descs != nullptr || n == 0, "prefetch_blocks: null descriptors");
  const bool node_on = node && node->staging(), edge_on = edge && edge->staging();
  if (!node_on && !edge_on) return false;
  const int device = node ? node->device() : edge->device();
  DeviceGuard dg(device);
  bool node_use = false, edge_use = false;
  for (size_t i = 0; i < n; ++i) {
    const gf_fetch_desc& d = descs[i];
    GF_REQUIRE(d.kind >= 0 && d.kind <= 2, "prefetch_blocks: bad kind");
    if (d.n == 0) continue;
    GF_REQUIRE(d.d_ids != nullptr, "prefetch_blocks: null ids");
    GF_REQUIRE(d.n < 0x7FFFFFFFull, "prefetch_blocks: more than 2^31-1 rows in one block");
    if (d.kind == 0) node_use = node_use || node_on;
    else edge_use = edge_use || edge_on;
  }
  if (node_use) node_use = node->stage_advance();
  if (edge_use) edge_use = edge->stage_advance();
  if (!node_use && !edge_use) return false;
  size_t max_n = 0, node_rows = 0, edge_rows = 0;
  StageRound r;
  r.count = 0;
  auto flush = [&] {
    if (r.count == 0) return;
    const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((max_n + 255) / 256, 512)));
    stage_claim_kernel<<<dim3(grid, r.count), dim3(256), 0, stream>>>(r);
    GF_HIP(hipGetLastError());
    r.count = 0;
    max_n = 0;
  };
  for (size_t i = 0; i < n; ++i) {
    const gf_fetch_desc& d = descs[i];
    if (d.n == 0) continue;
    FeatureCache* c = d.kind == 0 ? (node_use ? node : nullptr) : (edge_use ? edge : nullptr);
    if (!c) continue;
    c->stage_begin(&r.c[r.count++], d.d_ids, d.n, d.kind != 2);
    max_n = std::max(max_n, d.n);
    (d.kind == 0 ? node_rows : edge_rows) += d.n;
    if (r.count == kMaxCtx) flush();
  }
  flush();
  {
    PullJobs jobs;
    jobs.count = 0;
    size_t most = 0;
    if (node_use) {
      node->stage_pull(&jobs.j[jobs.count++]);
      most = std::max(most, std::min<size_t>(node_rows, node->stage_cap_));
    }
    if (edge_use) {
      edge->stage_pull(&jobs.j[jobs.count++]);
      most = std::max(most, std::min<size_t>(edge_rows, edge->stage_cap_));
    }
    // 8 rows per wave, 4 waves per workgroup; the kernel reads the rows really claimed
    // (grid-stride: 64 workgroups = 256 waves; 192 stretched the GDELT-scale gathers beside the pull from 254 to 578 us)
    static const size_t max_wgs = [] {
      const char* v = std::getenv("GNNFLOW_STAGE_PULL_WGS");   // tuning
      return v ? static_cast<size_t>(std::max(1, std::atoi(v))) : size_t{64};
    }();
    const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((most + 31) / 32, max_wgs)));
    stage_pull_kernel<<<dim3(grid, jobs.count), dim3(256), 0, stream>>>(jobs);
    GF_HIP(hipGetLastError());
  }
  FeatureCache* lead = edge_use ? edge : node;
  hipEvent_t ev = lead->stage_events_[lead->gen_issued_ % FeatureCache::kStageEvents];
  GF_HIP(hipEventRecord(ev, stream));
  if (node_use) node->gen_event_[node->gen_issued_ % FeatureCache::kStageEvents] = ev;
  if (edge_use) edge->gen_event_[edge->gen_issued_ % FeatureCache::kStageEvents] = ev;
  return true;
}

// ---- sharded feature tables: plan, serve, fetch (kernels above: "planning a pull") ---------
namespace {
PullRound make_pull_round(const gf_pull_desc* descs, size_t n, int world, FeatureCache* const* caches,
                          uint32_t* d_counts, uint32_t* d_cursor, size_t* max_rows,
                          uint32_t cstride = 1) {
  GF_REQUIRE(descs != nullptr && n >= 1 && n <= static_cast<size_t>(kMaxCtx),
             "pull: 1..4 contexts per round");
  GF_REQUIRE(world >= 1 && world <= 64, "pull: world size must be 1..64");
  PullRound r;
  std::memset(&r, 0, sizeof(r));
  r.count = static_cast<int>(n);
  r.od = owner_div(static_cast<uint32_t>(world));
  *max_rows = 0;
  for (size_t i = 0; i < n; ++i) {
    const gf_pull_desc& d = descs[i];
    GF_REQUIRE(d.n == 0 || d.d_ids != nullptr, "pull: null ids");
    GF_REQUIRE(d.n < 0x7FFFFFFFull, "pull: more than 2^31-1 rows in one block");
    PullCtx& c = r.c[i];
    c.ids = d.d_ids;
    c.n = static_cast<uint32_t>(d.n);
    c.key_base = d.d_key_base;
    c.key_index = d.d_key_index;
    c.map = caches[i] ? caches[i]->pull_map() : nullptr;
    c.num_ids = caches[i] ? caches[i]->num_ids() : d.num_ids;
    c.counts = cstride == 1 ? d_counts + i * world : d_counts + i;   // [ctx][owner] | [owner][ctx]
    c.cstride = cstride;
    c.cursor = d_cursor ? d_cursor + i * world : nullptr;
    c.send_ids = d.d_send_ids;
    c.req_pos = d.d_req_pos;
    *max_rows = std::max(*max_rows, d.n);
  }
  return r;
}
inline unsigned pull_grid(size_t rows) {
  return static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>((rows + 255) / 256, 2048)));
}
}  // namespace

void pull_count(const gf_pull_desc* descs, size_t n, int world, FeatureCache* const* caches,
                uint32_t* d_counts, int device, hipStream_t stream) {
  GF_REQUIRE(d_counts != nullptr, "pull_count: null counts");
  DeviceGuard dg(device);
  size_t rows;
  PullRound r = make_pull_round(descs, n, world, caches, d_counts, nullptr, &rows);
  GF_HIP(hipMemsetAsync(d_counts, 0, n * world * sizeof(uint32_t), stream));
  if (rows == 0) return;
  const dim3 grid(pull_grid(rows), static_cast<unsigned>(n));
  pull_claim_kernel<<<grid, dim3(256), 0, stream>>>(r);
  pull_bucket_kernel<false><<<grid, dim3(256), 0, stream>>>(r);
  GF_HIP(hipGetLastError());
}

void pull_scatter(const gf_pull_desc* descs, size_t n, int world, FeatureCache* const* caches,
                  uint32_t* d_counts, uint32_t* d_cursor, int device, hipStream_t stream) {
  GF_REQUIRE(d_counts && d_cursor, "pull_scatter: null counts / cursor");
  DeviceGuard dg(device);
  size_t rows;
  PullRound r = make_pull_round(descs, n, world, caches, d_counts, d_cursor, &rows);
  for (size_t i = 0; i < n; ++i)
    GF_REQUIRE(descs[i].n == 0 || (descs[i].d_send_ids && descs[i].d_req_pos),
               "pull_scatter: null send / position buffer");
  GF_HIP(hipMemsetAsync(d_cursor, 0, n * world * sizeof(uint32_t), stream));
  if (rows == 0) return;
  const dim3 grid(pull_grid(rows), static_cast<unsigned>(n));
  pull_bucket_kernel<true><<<grid, dim3(256), 0, stream>>>(r);
  GF_HIP(hipGetLastError());
}

// the owner's side: out[i,:] = rows[index[ids[i]],:]
void gather_rows_indexed(const float* d_rows, size_t num_local_rows, size_t dim,
                         const int32_t* d_index, size_t num_ids, const int64_t* d_ids, size_t n,
                         float* d_out, uint32_t* d_flag, int device, hipStream_t stream) {
  if (n == 0) return;
  GF_REQUIRE(d_rows && d_index && d_ids && d_out && d_flag, "gather_rows_indexed: null pointer");
  GF_REQUIRE(dim > 0 && num_local_rows > 0, "gather_rows_indexed: empty shard");
  DeviceGuard dg(device);
  Round r;
  r.count = 1;
  r.c[0] = plain_ctx(d_rows, num_ids, dim, d_ids, n, d_out);
  r.c[0].remap = d_index;
  r.c[0].flag = d_flag;
  launch_round(r, stream);
}

// All fetches of one fetch_feature() call over sharded tables, rounds as in fetch_blocks: the
// pulled rows stand in for the local table, a missed row finds its own through the claim the
// plan settled (Ctx::req_pos).
void fetch_blocks_pulled(FeatureCache* node, FeatureCache* edge, const gf_fetch_pulled_desc* descs,
                         size_t n, hipStream_t stream) {
  GF_REQUIRE(descs != nullptr || n == 0, "fetch_blocks_pulled: null descriptors");
  std::vector<const gf_fetch_pulled_desc*> nodes, edges;
  for (size_t i = 0; i < n; ++i) {
    const gf_fetch_pulled_desc& d = descs[i];
    GF_REQUIRE(d.kind == 0 || d.kind == 1, "fetch_blocks_pulled: kind must be 0 (node) or 1 (edge)");
    GF_REQUIRE((d.kind == 0 ? node : edge) == nullptr || (d.kind == 0 ? node : edge)->mirror_,
               "fetch_blocks_pulled: pulled rows need the row mirror");
    if (d.n == 0) continue;
    GF_REQUIRE(d.d_pulled_rows && d.d_req_pos, "fetch_blocks_pulled: null pulled rows");
    GF_REQUIRE((d.kind == 0 ? node : edge) != nullptr, "fetch_blocks_pulled: block without its cache");
    (d.kind == 0 ? nodes : edges).push_back(&d);
  }
  const int device = node ? node->device() : (edge ? edge->device() : 0);
  DeviceGuard dg(device);
  size_t max_node_rows = 0, max_edge_rows = 0;
  for (const auto* d : nodes) max_node_rows = std::max(max_node_rows, d->n);
  for (const auto* d : edges) max_edge_rows = std::max(max_edge_rows, d->n);
  if (node && max_node_rows) node->reserve_workspace(max_node_rows, stream);
  if (edge && max_edge_rows) edge->reserve_workspace(max_edge_rows, stream);
  const size_t rounds = std::max(nodes.size(), edges.size());
  for (size_t i = 0; i < rounds; ++i) {
    Round r;
    r.count = 0;
    auto add = [&](FeatureCache* fc, const gf_fetch_pulled_desc& d) {
      Ctx& c = r.c[r.count++];
      fc->prepare(d.d_ids, d.n, d.d_out, d.update != 0, d.d_stats, &c, stream);
      c.miss_rows = d.d_pulled_rows;
      c.req_pos = d.d_req_pos;
      c.inst_from_table = 0;   // the missed rows are the pulled ones, not a local table's
      if (c.vec4 && (reinterpret_cast<uintptr_t>(d.d_pulled_rows) & 15u)) {
        c.vec4 = 0;
        c.dimv = static_cast<uint32_t>(fc->dim_);
        set_odd4(c, fc->dim_, fc->policy_ == GF_CACHE_LRU || !d.update || fc->capacity_ == 0);
      }
    };
    if (i < nodes.size()) add(node, *nodes[i]);
    if (i < edges.size()) add(edge, *edges[i]);
    launch_round(r, stream);
  }
}

// out[i,:] = rows[pos[i],:] — the cache-free context of a pull round (every row travelled;
// req_pos is its place among the pulled rows)
namespace {
__global__ void rows_by_pos_kernel(const float* __restrict__ rows, const uint32_t* __restrict__ pos,
                                   uint32_t n, uint32_t dim, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t i = wave; i < n; i += nwaves) {
    const float* s = rows + static_cast<uint64_t>(pos[i]) * dim;
    float* o = out + static_cast<uint64_t>(i) * dim;
    for (uint32_t c = lane; c < dim; c += 64) o[c] = s[c];
  }
}
}  // namespace

// One fetch round over sharded tables as ONE native call: plan -> count exchange -> the round's
// host synchronisation -> scatter -> ids out -> serve -> rows back -> fetch (the stages
// Cache._pull_round issues one by one when the exchange has to go through torch.distributed).
PullSession::PullSession(Exchange* ex, int device) : ex_(ex), device_(device) {}

void PullSession::round(FeatureCache* node, FeatureCache* edge, const gf_pull_ctx* ctxs, size_t n,
                        int flag, int* any_flag, uint64_t* rows_pulled, uint64_t* bytes_sent,
                        uint32_t* d_error_flag, hipStream_t st) {
  GF_REQUIRE(ctxs != nullptr && n >= 1 && n <= static_cast<size_t>(kMaxCtx),
             "pull round: 1..4 contexts");
  GF_REQUIRE(d_error_flag != nullptr, "pull round: null error flag");
  DeviceGuard dg(device_);
  const int P = ex_ ? ex_->world() : 1, me = ex_ ? ex_->rank() : 0;
  const size_t W = n + 1;   // words per owner: the contexts' counts + this rank's flag
  // buffers
  gf_pull_desc descs[kMaxCtx];
  FeatureCache* caches[kMaxCtx] = {nullptr, nullptr, nullptr, nullptr};
  for (size_t k = 0; k < n; ++k) {
    const gf_pull_ctx& c = ctxs[k];
    GF_REQUIRE(c.kind >= 0 && c.kind <= 2, "pull round: bad kind");
    GF_REQUIRE(c.dim > 0 && c.d_shard_rows && c.d_shard_index, "pull round: bad shard");
    caches[k] = c.kind == 0 ? node : (c.kind == 1 ? edge : nullptr);
    GF_REQUIRE(c.kind == 2 || caches[k] != nullptr, "pull round: block without its cache");
    send_ids_[k].reserve(std::max<size_t>(c.pull.n, 1) * 8, 0, st);
    req_pos_[k].reserve(std::max<size_t>(c.pull.n, 1) * 4, 0, st);
    // sized BEFORE the round's first exchange from what is known now (at most pull.n rows leave;
    // about as many arrive when the ids spread evenly), doubling: in the steady state no
    // hipMalloc / hipFree — a device-wide synchronisation — sits between two collectives
    pulled_[k].reserve(std::max<size_t>(c.pull.n, 1) * c.dim * 4, 0, st);
    got_[k].reserve(std::max<size_t>(c.pull.n, 1) * 8, 0, st);
    served_[k].reserve(std::max<size_t>(c.pull.n, 1) * c.dim * 4, 0, st);
    descs[k] = c.pull;
    descs[k].cache = nullptr;   // caches[] carries it
    descs[k].d_send_ids = send_ids_[k].as<int64_t>();
    descs[k].d_req_pos = req_pos_[k].as<uint32_t>();
  }
  counts_.reserve((2 * P * W + n * P) * 4, 0, st);
  h_counts_.reserve(2 * P * W * 4);
  uint32_t* d_counts = counts_.as<uint32_t>();        // [P][W] own, then [P][W] received
  uint32_t* d_recv = d_counts + P * W;
  uint32_t* d_cursor = d_recv + P * W;
  // 1. claims + per-owner counts, owner-major so that row q goes to rank q as it is
  size_t rows;
  PullRound r = make_pull_round(descs, n, P, caches, d_counts, nullptr, &rows,
                                static_cast<uint32_t>(W));
  GF_HIP(hipMemsetAsync(d_counts, 0, P * W * 4, st));
  if (rows) {
    const dim3 grid(pull_grid(rows), static_cast<unsigned>(n));
    pull_claim_kernel<<<grid, dim3(256), 0, st>>>(r);
    pull_bucket_kernel<false><<<grid, dim3(256), 0, st>>>(r);
    GF_HIP(hipGetLastError());
  }
  if (flag)   // this rank's flag rides in word n of every owner's row (any non-zero value)
    GF_HIP(hipMemset2DAsync(d_counts + n, W * 4, 1, 4, P, st));
  if (ex_) ex_->all_to_all(d_counts, d_recv, W * 4, st);
  else GF_HIP(hipMemcpyAsync(d_recv, d_counts, P * W * 4, hipMemcpyDeviceToDevice, st));
  // 2. the round's one host synchronisation
  uint32_t* h = h_counts_.as<uint32_t>();
  GF_HIP(hipMemcpyAsync(h, d_counts, 2 * P * W * 4, hipMemcpyDeviceToHost, st));
  GF_HIP(hipStreamSynchronize(st));
  const uint32_t* hs = h;             // hs[q * W + k]: rows of context k this rank sends to q
  const uint32_t* hr = h + P * W;     // hr[q * W + k]: rows rank q asks this rank for
  int any = flag ? 1 : 0;
  for (int q = 0; q < P; ++q) any |= hr[q * W + n] ? 1 : 0;
  if (any_flag) *any_flag = any;
  // 3. ids into the compact owner-major send buffers
  r = make_pull_round(descs, n, P, caches, d_counts, d_cursor, &rows, static_cast<uint32_t>(W));
  GF_HIP(hipMemsetAsync(d_cursor, 0, n * P * 4, st));
  if (rows) {
    pull_bucket_kernel<true><<<dim3(pull_grid(rows), static_cast<unsigned>(n)), dim3(256), 0, st>>>(r);
    GF_HIP(hipGetLastError());
  }
  std::vector<size_t> sb(P), so(P), rb(P), ro(P);
  gf_fetch_pulled_desc fd[kMaxCtx];
  size_t nf = 0;
  // a skewed round (more rows asked of this rank than it asks for itself): grow for every
  // context now, before the first of the id / row exchanges
  for (size_t k = 0; k < n; ++k) {
    size_t n_recv = 0;
    for (int q = 0; q < P; ++q) n_recv += hr[q * W + k];
    got_[k].reserve(std::max<size_t>(n_recv, 1) * 8, 0, st);
    served_[k].reserve(std::max<size_t>(n_recv, 1) * ctxs[k].dim * 4, 0, st);
  }
  for (size_t k = 0; k < n; ++k) {
    const gf_pull_ctx& c = ctxs[k];
    size_t n_send = 0, n_recv = 0;
    for (int q = 0; q < P; ++q) { n_send += hs[q * W + k]; n_recv += hr[q * W + k]; }
    GF_REQUIRE(n_send <= std::max<size_t>(c.pull.n, 1), "pull round: more rows claimed than asked");
    auto exchange = [&](const void* send, void* recv, size_t row_bytes, bool back) {
      // forward: this rank's ids to their owners; back: the owners' rows to the requesters
      size_t a = 0, b = 0;
      for (int q = 0; q < P; ++q) {
        const size_t s_rows = back ? hr[q * W + k] : hs[q * W + k];
        const size_t r_rows = back ? hs[q * W + k] : hr[q * W + k];
        sb[q] = s_rows * row_bytes; so[q] = a; a += sb[q];
        rb[q] = r_rows * row_bytes; ro[q] = b; b += rb[q];
      }
      // every rank makes every call, whatever its own sizes are: a transport may synchronise
      // the ranks inside it
      if (ex_) ex_->all_to_all_v(send, sb.data(), so.data(), recv, rb.data(), ro.data(), st);
      else if (a) GF_HIP(hipMemcpyAsync(recv, send, a, hipMemcpyDeviceToDevice, st));
    };
    // 4. ids out, served by their owners, rows back
    exchange(send_ids_[k].data(), got_[k].data(), 8, false);
    if (n_recv)
      gather_rows_indexed(c.d_shard_rows, c.shard_rows, c.dim, c.d_shard_index, c.pull.num_ids,
                          got_[k].as<int64_t>(), n_recv, served_[k].as<float>(), d_error_flag,
                          device_, st);
    exchange(served_[k].data(), pulled_[k].data(), c.dim * 4, true);
    if (rows_pulled) rows_pulled[k] = n_send - hs[me * W + k];
    if (bytes_sent)
      bytes_sent[k] = 8 * (n_send - hs[me * W + k]) + 4 * c.dim * (n_recv - hr[me * W + k]);
    // 5. the fetch itself
    if (c.pull.n == 0) continue;
    GF_REQUIRE(c.d_out != nullptr, "pull round: null output");
    if (c.kind == 2) {
      const unsigned grid = static_cast<unsigned>(std::min<size_t>((c.pull.n + 3) / 4, 2048));
      rows_by_pos_kernel<<<dim3(grid), dim3(256), 0, st>>>(
          pulled_[k].as<float>(), req_pos_[k].as<uint32_t>(), static_cast<uint32_t>(c.pull.n),
          static_cast<uint32_t>(c.dim), c.d_out);
      GF_HIP(hipGetLastError());
      continue;
    }
    gf_fetch_pulled_desc& f = fd[nf++];
    f.kind = c.kind;
    f.update = c.update;
    f.d_ids = c.pull.d_ids;
    f.n = c.pull.n;
    f.d_out = c.d_out;
    f.d_stats = c.d_stats;
    f.d_pulled_rows = pulled_[k].as<float>();
    f.d_req_pos = req_pos_[k].as<uint32_t>();
  }
  if (nf) fetch_blocks_pulled(node, edge, fd, nf, st);
}

void FeatureCache::slot_ids(int64_t* out, size_t capacity) const {
  GF_REQUIRE(out != nullptr && capacity >= capacity_, "slot_ids: output too small");
  if (!capacity_) return;
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  GF_HIP(hipMemcpy(out, slot_id_.data(), capacity_ * sizeof(int64_t), hipMemcpyDeviceToHost));
}

void FeatureCache::lru_state(uint64_t out[7]) const {
  for (int i = 0; i < 7; ++i) out[i] = 0;
  if (policy_ != GF_CACHE_LRU || !capacity_) return;
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  QueueState qs;
  GF_HIP(hipMemcpy(&qs, qstate_.data(), sizeof(qs), hipMemcpyDeviceToHost));
  out[0] = queue_form_ ? 1 : 0;
  out[1] = queue_cap_;
  out[2] = qs.head;
  out[3] = qs.tail;
  out[4] = compactions_;
  out[5] = list_form_updates_;
  out[6] = qs.lone_walks;
}

// Granules of the fused LRU list update that did not arrive within the polling budget and were
// recomputed by the waiting thread (since the library was loaded, current device).
uint64_t lru_recounts() {
  unsigned int v = 0;
  GF_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lru_recounts), sizeof(v)));
  return v;
}

size_t FeatureCache::mem_bytes() const {
  return (mirror_ ? capacity_ * dim_ * sizeof(float) : 0) + num_ids_ * sizeof(int32_t) +
         capacity_ * (sizeof(int64_t) + 2 * sizeof(uint32_t));
}

// The cached rows' copy in HBM (`buffer`).  With the feature table itself in HBM a hit and a miss
// are the same bytes at the same distance: without the mirror the slots hold ids only, every
// row is read from the table, an install moves nothing — the replacement state (what the
// reference's protocol lets a caller observe: hit ratios, which ids are cached) is unchanged.
void FeatureCache::set_row_mirror(bool on) {
  DeviceGuard dg(device_);
  GF_HIP(hipDeviceSynchronize());
  GF_REQUIRE(on || table_on_device_, "cache: only a table in device memory can do without the row mirror");
  if (on == mirror_) return;
  mirror_ = on;
  if (!on) { buffer_.release(); return; }
  buffer_.reserve(std::max<size_t>(capacity_ * dim_ * sizeof(float), 16), 0, nullptr, true);
  // the rows of the ids cached right now
  if (capacity_) {
    std::vector<int64_t> ids(capacity_);
    GF_HIP(hipMemcpy(ids.data(), slot_id_.data(), capacity_ * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (size_t s2 = 0; s2 < capacity_; ++s2)
      if (ids[s2] >= 0)
        GF_HIP(hipMemcpy(buffer_.as<float>() + s2 * dim_, feats_ + static_cast<size_t>(ids[s2]) * dim_,
                         dim_ * sizeof(float), hipMemcpyDefault));
  }
}

}  // namespace gf
