// Fused feature gather + LRU feature cache in HBM — MI355X counterpart of
// gnnflow/cache/cache.py (Cache.fetch_feature) and gnnflow/cache/lru_cache.py.
#pragma once

#include <cstdint>

#include "comm.hpp"
#include "common.hpp"

namespace gf {

// out[i,:] = feats[ids[i],:]  (gnnflow/utils.py:465-474; cache.py:411)
void gather_rows(const float* d_feats, size_t num_rows, size_t dim, const int64_t* d_ids,
                 size_t n, float* d_out, int device, hipStream_t stream);

void gather_rows_multi(const float* const* tables, const size_t* dims, float* const* outs,
                       size_t num_tables, size_t num_rows, const int64_t* d_ids, size_t n,
                       int device, hipStream_t stream);

// TGN memory / mailbox update, memory_ops.hip
void memory_update(float* node_memory, float* node_memory_ts, float* mailbox, float* mailbox_ts,
                   size_t num_nodes, size_t dim_memory, size_t dim_edge, const int64_t* nid,
                   const float* memory, const float* ts, const float* edge_feats, size_t n,
                   int neg_sample_ratio, unsigned long long* win_mail,
                   unsigned long long* win_mem, unsigned long long epoch, int device,
                   hipStream_t stream);

class FeatureCache;
void fetch_blocks(FeatureCache* node, FeatureCache* edge, const gf_fetch_desc* descs, size_t n,
                  hipStream_t stream);
// Pulls the table rows of the ids a coming fetch_blocks(descs) will miss into the caches' staging
// rings, on `stream` (a side stream); d_out / d_stats / update of the descriptors are ignored.
// Returns false when nothing was issued (no cache has a staging ring, or the generation had to be
// dropped because fetches that may still read the region it would overwrite are in flight).
bool prefetch_blocks(FeatureCache* node, FeatureCache* edge, const gf_fetch_desc* descs, size_t n,
                     hipStream_t stream);
// sharded feature tables (Cache(distributed=True)): plan the pull of a round's contexts,
// serve received ids from a shard, fetch with the pulled rows (feature_cache.hip)
void pull_count(const gf_pull_desc* descs, size_t n, int world, FeatureCache* const* caches,
                uint32_t* d_counts, int device, hipStream_t stream);
void pull_scatter(const gf_pull_desc* descs, size_t n, int world, FeatureCache* const* caches,
                  uint32_t* d_counts, uint32_t* d_cursor, int device, hipStream_t stream);
void gather_rows_indexed(const float* d_rows, size_t num_local_rows, size_t dim,
                         const int32_t* d_index, size_t num_ids, const int64_t* d_ids, size_t n,
                         float* d_out, uint32_t* d_flag, int device, hipStream_t stream);
void fetch_blocks_pulled(FeatureCache* node, FeatureCache* edge, const gf_fetch_pulled_desc* descs,
                         size_t n, hipStream_t stream);

// One fetch round over sharded feature tables as one native call (gf_pull_round): the buffers
// of the round (grow-only) and the transport (null: one rank, nothing travels).
class PullSession {
 public:
  PullSession(Exchange* ex, int device);
  void round(FeatureCache* node, FeatureCache* edge, const gf_pull_ctx* ctxs, size_t n, int flag,
             int* any_flag, uint64_t* rows_pulled, uint64_t* bytes_sent, uint32_t* d_error_flag,
             hipStream_t stream);

 private:
  Exchange* ex_;
  int device_;
  DeviceBuffer send_ids_[4], req_pos_[4], got_[4], served_[4], pulled_[4], counts_;
  PinnedBuffer h_counts_;
};

class FeatureCache {
 public:
  FeatureCache(size_t num_ids, size_t capacity, size_t dim, const float* d_feats, int device);
  ~FeatureCache();

  void init(hipStream_t stream);
  // slot i caches d_ids[i]; its row comes from d_rows[i] if given, else from the feature table
  void init_ids(const int64_t* d_ids, size_t n, hipStream_t stream,
                const float* d_rows = nullptr);
  void probe(const int64_t* d_ids, size_t n, int32_t* d_slot, hipStream_t stream);
  void fetch_pulled(const int64_t* d_ids, size_t n, float* d_out, bool update,
                    uint32_t* d_stats, const float* d_miss_rows, const uint32_t* d_miss_index,
                    hipStream_t stream);
  void set_policy(int policy);
  // off: no copy of the cached rows (the table is in HBM: hits read it too); see feature_cache.hip
  void set_row_mirror(bool on);
  bool row_mirror() const { return mirror_; }
  void reset_order(hipStream_t stream);
  void rewind_fifo(hipStream_t stream);
  void resize(size_t new_num_ids, size_t new_capacity, const float* d_feats, hipStream_t stream);
  void fetch(const int64_t* d_ids, size_t n, float* d_out, bool update, uint32_t* d_stats,
             hipStream_t stream);
  void gather_plain(const int64_t* d_ids, size_t n, float* d_out, hipStream_t stream);
  void slot_ids(int64_t* out, size_t capacity) const;
  void lru_state(uint64_t out[7]) const;   // gf_cache_lru_state
  size_t mem_bytes() const;
  int device() const { return device_; }
  size_t num_ids() const { return num_ids_; }
  int32_t* pull_map() { return capacity_ ? map_.as<int32_t>() : nullptr; }
  // Staging ring for a HOST-resident table (feature_cache.hip, "staging ring"): `generations`
  // (a power of two, >= 4) regions of `rows_per_generation` rows; 0 generations: off.
  void set_staging(size_t generations, size_t rows_per_generation);
  bool staging() const { return stage_gens_ != 0; }
  // forget every staged row (the table's contents changed)
  void invalidate_staging();
  void set_staging_lag(size_t lag) { stage_lag_ = static_cast<uint32_t>(std::min<size_t>(lag, kStageAhead - 1)); }
  // out[0..6]: generations, rows per generation, generations issued, generations dropped,
  // rows pulled over the host link so far (reads a device counter: synchronises), ring bytes,
  // rows the gathers read from the host table after all (not staged, or staged too long ago)
  // out[7]: microseconds the issuing thread waited in all for fetches to leave a region
  // out[8]: fetches (process-wide) that had to make their stream wait for a pull's event
  void staging_state(uint64_t out[9]);
  // diagnostics: 100 MHz wall-clock stamps of every workgroup of the last one-launch list update
  void lru_trace_enable(bool on);
  size_t lru_trace_read(uint64_t* out, size_t capacity_words);

 private:
  friend void fetch_blocks(FeatureCache*, FeatureCache*, const gf_fetch_desc*, size_t,
                           hipStream_t);
  friend bool prefetch_blocks(FeatureCache*, FeatureCache*, const gf_fetch_desc*, size_t,
                              hipStream_t);
  friend void fetch_blocks_pulled(FeatureCache*, FeatureCache*, const gf_fetch_pulled_desc*, size_t,
                                  hipStream_t);
  void reserve_workspace(size_t n, hipStream_t stream);
  // fills a device context (feature_cache.hip: struct Ctx) for one block fetch and advances
  // the host-side epoch / counter ring (LRU: schedules a queue compaction when due)
  void prepare(const int64_t* d_ids, size_t n, float* d_out, bool update, uint32_t* d_stats,
               void* ctx_out, hipStream_t stream);
  void init_queue(hipStream_t stream);   // LRU list (feature_cache.hip header)

  size_t num_ids_, capacity_, dim_;
  const float* feats_;
  int device_;
  bool table_on_device_ = false;   // the feature table is HBM-resident (not pinned host memory)
  bool mirror_ = true;             // `buffer_` holds a copy of every cached row

  DeviceBuffer buffer_;    // float[capacity * dim]        cache rows
  DeviceBuffer map_;       // int32[num_ids]               id -> slot (kAbsent if none)
  DeviceBuffer slot_id_;   // int64[capacity]              slot -> id (-1 empty)
  DeviceBuffer stamp_;     // uint32[capacity]  LFU: use count; FIFO: install epoch
  DeviceBuffer touched_;   // uint32[capacity]  epoch of the last hit (pending); LRU list form:
                           // indexed by list position, else by slot
  DeviceBuffer queue_, queue_alt_;   // uint32[capacity]  LRU: slots, least recently refreshed
                                     // first; two buffers, the device knows which is current
  DeviceBuffer qstate_;    // LRU: parity of the current list buffer (+ head / tail of the
                           // queue form), device resident
  // LRU of a large cache is kept as a queue with dead entries (feature_cache.hip, "LRU as a
  // queue"): updates cost O(block rows), not O(capacity)
  DeviceBuffer qpos_;      // uint32[capacity]  position of the slot's (live) list / queue entry
  DeviceBuffer wsnap_;     // uint2 per word of qbits_: {word, hit entries before it in its tile}
  DeviceBuffer qbits_;     // one bit per queue position: entries hit by the current block
  DeviceBuffer compact_;   // scratch of the (rare) queue compaction
  bool queue_form_ = false;
  size_t queue_cap_ = 0;   // entries allocated per queue buffer (capacity if list form)
  size_t tail_bound_ = 0;  // host-side upper bound of the device-resident queue tail
  uint64_t compactions_ = 0, list_form_updates_ = 0;
  void compact_queue(hipStream_t stream);
  void index_queue(hipStream_t stream);
  RetiredBuffers retired_; // scratch replaced while kernels may still use it
  DeviceBuffer state_;     // ring of per-fetch counter records
  DeviceBuffer fifo_ptr_;  // uint32: FIFO rotation pointer
  DeviceBuffer ws_;        // per-fetch scratch
  size_t ws_rows_ = 0;
  DeviceBuffer granules_;  // LRU list form in one launch: {launch tag, count} per list tile / row
                           // workgroup (feature_cache.hip, lru_list_fused_kernel)
  uint32_t fuse_tag_ = 0;  // tag of the last such launch: unique per cache, never reset
  DeviceBuffer trace_;     // gf_debug_lru_trace
  uint32_t epoch_ = 0;     // fetches with update so far (host side; kernel argument)
  // staging ring (set_staging)
  static constexpr uint32_t kStageEvents = 8;
  // generations that may be pulled while a gather that reads the ring is still running
  static constexpr uint32_t kStageAhead = 4;
  DeviceBuffer ring_;         // float[generations * rows * dim]
  DeviceBuffer pmap_;         // uint64[num_ids][2]  {generation, row in its region} newest, previous; 0: never staged
  DeviceBuffer region_rows_;  // uint32[generations] rows taken per region, + uint64 rows pulled
  DeviceBuffer region_ids_;   // int64[rows] ids claimed for the generation being pulled
  PinnedBuffer progress_;     // uint32: ring-reading launches known to have finished
  uint32_t stage_gens_ = 0, stage_cap_ = 0;
  uint32_t gen_issued_ = 0;   // last generation handed to a prefetch
  uint32_t stage_reads_ = 0;  // ring-reading launches enqueued so far
  bool stage_read_pending_ = false;       // ... the round being built reads the ring
  uint32_t reads_at_gen_[64] = {};        // stage_reads_ when generation g was issued
  uint64_t stage_drops_ = 0;
  double stage_spin_us_ = 0;
  hipEvent_t stage_events_[kStageEvents] = {};
  hipEvent_t gen_event_[kStageEvents] = {};   // event behind generation g's pull: [g % kStageEvents]
  uint32_t synced_gen_ = 0;               // newest generation a fetch stream has waited for
  uint32_t stage_lag_ = 0;                // newest generations a fetch does not depend on
  bool stage_advance();                   // takes the next generation; false: dropped
  bool stage_begin(void* stage_ctx_out, const int64_t* d_ids, size_t n, bool cached);
  void stage_pull(void* pull_job_out);
  void stage_sync(hipStream_t stream, hipEvent_t* seen, int* num_seen);
  uint32_t stage_hi() const;
  void stage_fill(void* ctx_out);         // ring fields of a gather context
  void stage_round_done();
  uint64_t ring_pos_ = 0;
  int policy_ = GF_CACHE_LRU;
};

}  // namespace gf
