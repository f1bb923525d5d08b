/*
 * gnnflow_hip.h — C ABI of the MI355X-native GNNFlow hot path
 * (temporal edge store -> TemporalSampler.sample() -> feature gather / LRU cache).
 *
 * This is the drop-in boundary: every entry point below replaces one method of
 * the reference's pybind11 module `libgnnflow` (gnnflow/csrc/api.cc) or one step
 * of the Python feature cache (gnnflow/cache/cache.py, lru_cache.py).  Signatures
 * use only plain pointers and sizes — no torch / pybind / STL types — so the same
 * library binds from ctypes (gnnflow_amd/_capi.py), from pybind11
 * (gnnflow_amd/csrc/pybind_libgnnflow.cpp builds a module named `libgnnflow` with
 * the reference's class/method names) or from any other FFI.
 *
 * Conventions
 *   - every function returns GF_OK (0) or a GF_ERR_* code; gf_last_error() gives
 *     the message for the calling thread (the reference CHECK()/abort()s instead,
 *     gnnflow/csrc/logging.h:9-46 — an error code is strictly friendlier).
 *   - "host" pointers are ordinary CPU memory; "device" pointers are HBM
 *     addresses on the graph's device.  `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream).
 *   - node / edge ids are int64, timestamps float32 (gnnflow/csrc/common.h:13-15).
 *   - not thread-safe per handle (one in-flight call per graph / sampler / cache,
 *     as in the reference: gnnflow/csrc/temporal_sampler.h:73-80); different
 *     handles may be used from different threads.  Calls do not hold the GIL.
 */
#ifndef GNNFLOW_HIP_H_
#define GNNFLOW_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_API __attribute__((visibility("default")))

/* ---- status ----------------------------------------------------------------- */
enum {
  GF_OK = 0,
  GF_ERR_INVALID_ARGUMENT = 1,
  GF_ERR_TIMESTAMP_ORDER = 2, /* add_edges: edges older than the node's newest edge
                                 (reference: CHECK_LE at gnnflow/csrc/utils.cu:42-43) */
  GF_ERR_OUT_OF_MEMORY = 3,   /* maximum_pool_size exceeded / hipMalloc failed */
  GF_ERR_HIP = 4,             /* a HIP runtime call failed */
  GF_ERR_IO = 5
};

/* gnnflow/csrc/api.cc:27-39 enum values, same order */
enum { GF_INSERTION_POLICY_INSERT = 0, GF_INSERTION_POLICY_REPLACE = 1 };
enum { GF_SAMPLING_POLICY_RECENT = 0, GF_SAMPLING_POLICY_UNIFORM = 1 };
enum { GF_MEM_CUDA = 0, GF_MEM_UNIFIED = 1, GF_MEM_PINNED = 2, GF_MEM_SHARED = 3 };

GF_API const char* gf_last_error(void);
GF_API const char* gf_version(void);

/* ---- DynamicGraph: gnnflow/csrc/api.cc:41-85, dynamic_graph.h:26-170 --------- */
typedef struct gf_graph gf_graph;

/* _DynamicGraph.__init__ (api.cc:42-47; `minium_block_size` spelling is the
 * reference's).  All four memory resource types place the edge store in HBM on
 * `device` (288 GB per MI355X makes the managed / pinned spill modes moot). */
GF_API int gf_graph_create(gf_graph** out, size_t initial_pool_size,
                           size_t maximum_pool_size, int mem_resource_type,
                           size_t minium_block_size, size_t blocks_to_preallocate,
                           int insertion_policy, int device, int adaptive_block_size);
GF_API int gf_graph_destroy(gf_graph* g);

/* _DynamicGraph.add_edges (api.cc:48-50 -> DynamicGraph::AddEdges,
 * dynamic_graph.cu:77-138).  Four host arrays of length n, any order inside the
 * batch; grouped by source and stable-sorted by timestamp on ingest. */
GF_API int gf_graph_add_edges(gf_graph* g, const int64_t* src, const int64_t* dst,
                              const float* ts, const int64_t* eids, size_t n);

/* _DynamicGraph.offload_old_blocks (api.cc:51-52 -> dynamic_graph.cu:382-411):
 * drops every whole logical block whose end timestamp < `timestamp`;
 * to_file != 0 also writes temporal_block_<node>-<k>.bin
 * (temporal_block_allocator.cu:182-221 format). */
GF_API int gf_graph_offload_old_blocks(gf_graph* g, float timestamp, int to_file,
                                       size_t* num_blocks);

/* api.cc:53-55,70-71 */
GF_API int gf_graph_num_vertices(const gf_graph* g, size_t* out);
GF_API int gf_graph_num_source_vertices(const gf_graph* g, size_t* out);
GF_API int gf_graph_num_edges(const gf_graph* g, size_t* out);
GF_API int gf_graph_max_vertex_id(const gf_graph* g, int64_t* out);
/* *out = 1 when every node id and every edge id inserted so far is in [0, 2^32 - 2]: the shared
 * chains of the partitioned sampler may then carry 12-byte reply slots (narrow_ids below).  No
 * reference counterpart. */
GF_API int gf_graph_ids_fit_u32(const gf_graph* g, int* out);
/* api.cc:56-59 out_degree(list) */
GF_API int gf_graph_out_degree(const gf_graph* g, const int64_t* nodes, size_t n,
                               size_t* out);
/* api.cc:60-69 nodes() / src_nodes() / edges(): pass out = NULL to query *count */
GF_API int gf_graph_nodes(const gf_graph* g, int64_t* out, size_t capacity, size_t* count);
GF_API int gf_graph_src_nodes(const gf_graph* g, int64_t* out, size_t capacity, size_t* count);
GF_API int gf_graph_edges(const gf_graph* g, int64_t* out, size_t capacity, size_t* count);
/* api.cc:72-78 get_temporal_neighbors(node) -> newest first; out arrays may be
 * NULL to query *count */
GF_API int gf_graph_get_temporal_neighbors(const gf_graph* g, int64_t node, int64_t* dst,
                                           float* ts, int64_t* eids, size_t capacity,
                                           size_t* count);
/* api.cc:79-85 */
GF_API int gf_graph_avg_linked_list_length(const gf_graph* g, float* out);
GF_API int gf_graph_memory_usage(const gf_graph* g, float* out);
GF_API int gf_graph_metadata_memory_usage(const gf_graph* g, float* out);
GF_API int gf_graph_device(const gf_graph* g, int* out);

/* ---- TemporalSampler: api.cc:111-120, temporal_sampler.h:19-38 -------------- */
typedef struct gf_sampler gf_sampler;

/* _TemporalSampler.__init__ (api.cc:112-116).  The sampler keeps a pointer to
 * the graph: destroy the sampler first (the reference holds `const DynamicGraph&`,
 * temporal_sampler.h:62). */
GF_API int gf_sampler_create(gf_sampler** out, gf_graph* g, const uint32_t* fanouts,
                             size_t num_layers, int sampling_policy,
                             uint32_t num_snapshots, float snapshot_time_window,
                             int prop_time, uint64_t seed);
GF_API int gf_sampler_destroy(gf_sampler* s);

/* One (layer, snapshot) SamplingResult (gnnflow/csrc/common.h:51-60).  Pointers
 * address the caller's output buffer; counts are valid after the call returns. */
typedef struct gf_block {
  int64_t* all_nodes;       /* [num_src_nodes] roots ++ sampled neighbours */
  float* all_timestamps;    /* [num_src_nodes] */
  float* delta_timestamps;  /* [num_edges] root ts - edge ts */
  int64_t* eids;            /* [num_edges] */
  int64_t* row;             /* [num_edges] root index of each edge (non-decreasing) */
  int64_t* col;             /* [num_edges] num_dst_nodes .. num_src_nodes-1 */
  uint64_t num_dst_nodes;   /* R */
  uint64_t num_src_nodes;   /* R + S */
  uint64_t num_edges;       /* S */
} gf_block;

/* Worst-case byte size of the output buffer of gf_sampler_sample for
 * `num_roots` roots (all layers x snapshots, every slot valid). */
GF_API int gf_sampler_output_bytes(const gf_sampler* s, size_t num_roots, size_t* bytes);

/* _TemporalSampler.sample (api.cc:117-118 -> TemporalSampler::Sample,
 * temporal_sampler.cu:279-305), device resident: roots / timestamps and the
 * output buffer are device pointers; all layers and snapshots are sampled without
 * leaving HBM (layer l+1 roots = layer l all_nodes / all_timestamps).
 * blocks: host array [num_layers * num_snapshots], index layer*num_snapshots+snap
 * (the reference's [layer][snapshot] order, before Python reverses the layers).
 * Synchronises `stream` once at the end to read the per-block sizes. */
GF_API int gf_sampler_sample(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                             size_t num_roots, void* d_out, size_t out_bytes,
                             gf_block* blocks, void* stream);

/* The same call split in two, for software pipelining (the reference overlaps the next
 * batch's sample() with training on a Python thread, scripts/offline_edge_prediction.py:
 * 343-346): begin enqueues everything on `stream` and returns without waiting; end waits
 * for that work and fills `blocks`.  One sample may be in flight per sampler. */
GF_API int gf_sampler_sample_begin(gf_sampler* s, const int64_t* d_roots,
                                   const float* d_root_ts, size_t num_roots, void* d_out,
                                   size_t out_bytes, void* stream);
GF_API int gf_sampler_sample_end(gf_sampler* s, gf_block* blocks);
/* Uniform sampling draws philox4x32-10(seed; slot, call) (include/gnnflow_rng.h), `call` = the
 * number of sample_layer invocations of the sampler so far — one per layer and snapshot of a
 * sample() (the reference draws from per-thread curand states, sampling_kernels.cu:109-273).
 * gf_sampler_call_counter reads it; gf_sampler_set_call_counter sets the number the NEXT begun
 * sample starts from (through_enqueue_thread != 0: ordered with gf_sampler_sample_begin_async).
 * Clones of one sampler over the same graph can then take turns — sampling lanes — and still
 * reproduce the one sampler's stream of draws: sample j starts at j x layers x snapshots. */
GF_API int gf_sampler_call_counter(const gf_sampler* s, uint64_t* out);
GF_API int gf_sampler_set_call_counter(gf_sampler* s, uint64_t value, int through_enqueue_thread);
/* As gf_sampler_sample_begin, but the launches are issued by the library's enqueue thread
 * (the one gf_cache_fetch_blocks_async uses, in submission order); gf_sampler_sample_end
 * waits for it.  The inputs must stay alive until gf_sampler_sample_end returns. */
GF_API int gf_sampler_sample_begin_async(gf_sampler* s, const int64_t* d_roots,
                                         const float* d_root_ts, size_t num_roots,
                                         void* d_out, size_t out_bytes, void* stream);
/* Which of the library's two sampling enqueue threads issues this sampler's
 * gf_sampler_sample_begin_async launches (1: the default, 2: the second one).  A sampling-only
 * pipeline whose lanes are a sampler and its clone puts the clone on 2: the four launches + event
 * of a sample cost ~19 us of issuing time, and one thread serving both lanes spends that per
 * sample — the loop then runs at the issuer's pace, not the GPU's.  No sample may be in flight.
 * (The reference samples on the caller's thread: gnnflow/temporal_sampler.py:149-165.) */
GF_API int gf_sampler_set_enqueue_lane(gf_sampler* s, int lane);

/* _TemporalSampler.sample_layer (api.cc:119-120 -> TemporalSampler::SampleLayer,
 * temporal_sampler.cu:97-277), device resident; *bytes variant sizes the buffer. */
GF_API int gf_sampler_layer_output_bytes(const gf_sampler* s, size_t num_roots,
                                         uint32_t layer, size_t* bytes);
GF_API int gf_sampler_sample_layer(gf_sampler* s, const int64_t* d_roots,
                                   const float* d_root_ts, size_t num_roots,
                                   uint32_t layer, uint32_t snapshot, void* d_out,
                                   size_t out_bytes, gf_block* block, void* stream);

/* Host-vector forms with the reference's exact calling convention (host arrays
 * in, freshly allocated host arrays out; api.cc:17-24 copies into numpy the same
 * way).  Results are released with gf_host_blocks_free. */
GF_API int gf_sampler_sample_host(gf_sampler* s, const int64_t* nodes, const float* ts,
                                  size_t num_roots, gf_block* blocks);
GF_API int gf_sampler_sample_layer_host(gf_sampler* s, const int64_t* nodes,
                                        const float* ts, size_t num_roots, uint32_t layer,
                                        uint32_t snapshot, gf_block* block);
GF_API void gf_host_blocks_free(gf_block* blocks, size_t n);

/* ---- feature gather + LRU cache: gnnflow/cache/cache.py, lru_cache.py -------- */
typedef struct gf_cache gf_cache;
typedef struct gf_comm gf_comm;   /* RCCL communicator of the partitioned path, below */

/* One cache "kind" (node or edge) of Cache.__init__ (cache.py:82-134) +
 * LRUCache.__init__ (lru_cache.py:56-61): `capacity` rows of `dim` float32 in
 * HBM, id -> slot map over `num_ids`, slot -> id table, LRU stamps.
 * d_feats: the full feature table [num_ids, dim] float32 — a device pointer, or
 * device-accessible pinned host memory (misses are then read over PCIe). */
GF_API int gf_cache_create(gf_cache** out, size_t num_ids, size_t capacity, size_t dim,
                           const float* d_feats, int device);
GF_API int gf_cache_destroy(gf_cache* c);
/* Replacement policy: LRU (lru_cache.py, the default), LFU (lfu_cache.py:
 * a hit adds 1 to the slot's use count, the k smallest counts are evicted, a new entry
 * starts at 1) or FIFO (fifo_cache.py: slots are refilled in rotation, hits change nothing).
 * Set before the first fetch. */
enum { GF_CACHE_LRU = 0, GF_CACHE_LFU = 1, GF_CACHE_FIFO = 2 };
GF_API int gf_cache_set_policy(gf_cache* c, int policy);
/* FIFOCache.reset (fifo_cache.py:70-75): forget the replacement order (next refill starts
 * at slot 0 again) but keep the cached rows. */
GF_API int gf_cache_reset_order(gf_cache* c, void* stream);
/* GNNLabStaticCache.init_cache (gnnlab_static_cache.py:87-168): slot i caches d_ids[i]
 * (n <= capacity, device array); used with update = 0 fetches. */
GF_API int gf_cache_init_ids(gf_cache* c, const int64_t* d_ids, size_t n, void* stream);
/* Cache.init_cache / LRUCache.reset (cache.py:157-195, lru_cache.py:74-105):
 * slots 0..capacity-1 hold ids 0..capacity-1, LRU state zeroed. */
GF_API int gf_cache_init(gf_cache* c, void* stream);
/* Cache.resize (cache.py:197-221) */
GF_API int gf_cache_resize(gf_cache* c, size_t new_num_ids, size_t new_capacity,
                           const float* d_feats, void* stream);

/* One block of Cache.fetch_feature (cache.py:255-400): out[i,:] = feats[ids[i],:]
 * served from the cache on a hit and from d_feats on a miss, then (update != 0
 * and at least one miss) the LRU replacement of lru_cache.py:121-201.
 * d_ids [n] int64 device, d_out [n, dim] float32 device.  d_stats (device,
 * 16 x uint32, zeroed by the caller, may be NULL) accumulates the block's hit count in
 * the even words (8 shards; hits = sum of d_stats[0,2,..,14]) and n in d_stats[1]; no
 * host synchronisation. */
GF_API int gf_cache_fetch(gf_cache* c, const int64_t* d_ids, size_t n, float* d_out,
                          int update, uint32_t* d_stats, void* stream);

/* Sharded feature tables (Cache(distributed=True): cache.py:293-312,351-388 pull the missed
 * rows from the owning machine's KVStore, gnnflow/distributed/kvstore.py:70-126; here the
 * owners are the GPUs of the node and the pull is the caller's all-to-all-v):
 *   gf_cache_probe        d_slot[i] = slot of d_ids[i] (>= 0), -1 not cached, -2 out of range
 *   gf_cache_fetch_pulled as gf_cache_fetch, but the row of a missed id i is
 *                         d_miss_rows[d_miss_index[i]] (the pulled rows), not d_feats[id]
 *   gf_cache_init_rows    slot i caches d_ids[i] with row d_rows[i] (init_cache when the first
 *                         `capacity` rows live on other GPUs, cache.py:175-195) */
GF_API int gf_cache_probe(gf_cache* c, const int64_t* d_ids, size_t n, int32_t* d_slot,
                          void* stream);
GF_API int gf_cache_fetch_pulled(gf_cache* c, const int64_t* d_ids, size_t n, float* d_out,
                                 int update, uint32_t* d_stats, const float* d_miss_rows,
                                 const uint32_t* d_miss_index, void* stream);
GF_API int gf_cache_init_rows(gf_cache* c, const int64_t* d_ids, size_t n, const float* d_rows,
                              void* stream);

/* The same pull planned natively, for all contexts of a fetch round at once (a node block, an
 * edge block, cache-free target rows — up to 4), without torch ops and with ONE host
 * synchronisation per round (reference: cache.py:288-313,351-388 probe / unique / pull;
 * kvstore.py:285-339).  owner(key) = splitmix64(key) mod world_size; the key of row i is
 * d_key_base[d_key_index[i]] (edge rows: the edge's source node = the block's root of that
 * edge), d_key_base[i], or the id itself when d_key_base is NULL.
 *   gf_pull_count    settles the claims of the missed ids (lowest row wins, the claim the
 *                    fetch's own gather makes) and counts the rows that travel — the winners,
 *                    and every row of a cache-free context — per owner:
 *                    d_counts[ctx * world_size + owner] (uint32, zeroed by the call);
 *   — the caller exchanges the counts and reads own and received counts back: the round's one
 *     host synchronisation (they are the split sizes of the two exchanges below) —
 *   gf_pull_scatter  writes the travelling ids compact and owner-major (offsets = the exclusive
 *                    prefix of gf_pull_count's d_counts) into d_send_ids and d_req_pos[row] =
 *                    the position of the row's id = the index of its row among the pulled rows,
 *                    which arrive in the same order (d_cursor: [n * world_size] scratch);
 *   — ids out (all-to-all-v), gf_gather_rows_indexed on the owner, rows back —
 *   gf_cache_fetch_blocks_pulled   the fetch round with the pulled rows standing in for the
 *                    local table; a missed row finds its row through the settled claim. */
typedef struct gf_pull_desc {
  gf_cache* cache;             /* NULL: cache-free rows (every row travels) */
  const int64_t* d_ids;
  size_t n;
  const int64_t* d_key_base;   /* NULL: the id is the key */
  const int64_t* d_key_index;  /* NULL: key = d_key_base[i] */
  size_t num_ids;              /* id space of a cache-free context (a cache knows its own) */
  int64_t* d_send_ids;         /* [n] gf_pull_scatter */
  uint32_t* d_req_pos;         /* [n] gf_pull_scatter */
} gf_pull_desc;
GF_API int gf_pull_count(const gf_pull_desc* descs, size_t n, int world_size, uint32_t* d_counts,
                         int device, void* stream);
GF_API int gf_pull_scatter(const gf_pull_desc* descs, size_t n, int world_size,
                           uint32_t* d_counts, uint32_t* d_cursor, int device, void* stream);
/* the owner's side of a pull: d_out[i,:] = d_rows[d_index[d_ids[i]],:] (d_index: global id ->
 * local row of this shard, int32, -1 = not owned: *d_flag is set to 1 and row 0 is served) */
GF_API int gf_gather_rows_indexed(const float* d_rows, size_t num_local_rows, size_t dim,
                                  const int32_t* d_index, size_t num_ids, const int64_t* d_ids,
                                  size_t n, float* d_out, uint32_t* d_flag, int device,
                                  void* stream);
typedef struct gf_fetch_pulled_desc {
  int kind;                    /* 0: node block, 1: edge block (executed in array order) */
  int update;
  const int64_t* d_ids;
  size_t n;
  float* d_out;
  uint32_t* d_stats;
  const float* d_pulled_rows;  /* the rows that came back for this context */
  const uint32_t* d_req_pos;   /* gf_pull_scatter's */
} gf_fetch_pulled_desc;
GF_API int gf_cache_fetch_blocks_pulled(gf_cache* node_cache, gf_cache* edge_cache,
                                        const gf_fetch_pulled_desc* descs, size_t n,
                                        void* stream);

/* The whole round in ONE call (the stages above + the exchanges through `comm`, which may be
 * NULL with one rank): per context the pull description, this rank's shard of the context's
 * table (rows + global id -> local row index) and where the fetched rows go.  kind 2 = cache-free
 * (target edge features): d_out[i,:] = the pulled row of d_ids[i].  `flag` travels to every rank
 * in the count exchange and *any_flag = OR over all ranks (Cache uses it to agree on the prefix
 * alias).  rows_pulled / bytes_sent [n]: per context, rows fetched from OTHER ranks and bytes
 * this rank put on the wire (ids out + rows served).  One host synchronisation. */
typedef struct gf_pull_ctx {
  gf_pull_desc pull;           /* cache, d_send_ids, d_req_pos are ignored (the session's own) */
  const float* d_shard_rows;   /* [shard_rows, dim] */
  size_t shard_rows;
  const int32_t* d_shard_index;/* [pull.num_ids] */
  size_t dim;
  int kind;                    /* 0 node block, 1 edge block, 2 cache-free rows */
  int update;
  float* d_out;                /* [pull.n, dim] */
  uint32_t* d_stats;
} gf_pull_ctx;
typedef struct gf_pull_session gf_pull_session;
GF_API int gf_pull_session_create(gf_pull_session** out, gf_comm* comm, int device);
GF_API int gf_pull_session_destroy(gf_pull_session* s);
GF_API int gf_pull_round(gf_pull_session* s, gf_cache* node_cache, gf_cache* edge_cache,
                         const gf_pull_ctx* ctxs, size_t n, int flag, int* any_flag,
                         uint64_t* rows_pulled, uint64_t* bytes_sent, uint32_t* d_error_flag,
                         void* stream);

/* All fetches of one Cache.fetch_feature() call (cache.py:255-413) in one call.
 * kind 0: block of node ids through node_cache (srcdata['h'], cache.py:269-323);
 * kind 1: block of edge ids through edge_cache (edata['f'], cache.py:326-400), executed
 *         in array order; kind 2: cache-free gather from the edge feature table
 *         (target_edge_features, cache.py:411).
 * The i-th node block and the i-th edge block share one round of launches (the kernels
 * take several contexts and blockIdx.y selects one), so the two caches advance together
 * on `stream`; no host synchronisation. */
typedef struct gf_fetch_desc {
  int kind;
  int update;
  const int64_t* d_ids;
  size_t n;
  float* d_out;
  uint32_t* d_stats;
} gf_fetch_desc;
GF_API int gf_cache_fetch_blocks(gf_cache* node_cache, gf_cache* edge_cache,
                                 const gf_fetch_desc* descs, size_t n, void* stream);

/* Asynchronous submission of the same work: the descriptors are copied and handed to an
 * enqueue thread inside the library, the call returns at once with a ticket, and
 * gf_cache_fetch_wait(ticket) blocks until that submission has been ENQUEUED on `stream`
 * (and returns its status); ordering of the results is then the stream's.  Submissions
 * are enqueued in ticket order.  The caller keeps ids / outputs alive until the wait. */
GF_API int gf_cache_fetch_blocks_async(gf_cache* node_cache, gf_cache* edge_cache,
                                       const gf_fetch_desc* descs, size_t n, void* stream,
                                       uint64_t* ticket);
GF_API int gf_cache_fetch_wait(uint64_t ticket);

/* ---- host-resident feature tables: the staging ring ---------------------------------------
 * Reference: the tables live in host memory and every miss is host -> pinned -> device inside
 * fetch_feature (gnnflow/cache/cache.py:288-313,381-388, gnnflow/utils.py:284-297).  With
 * gf_cache_set_staging(generations, rows) a cache over a HOST table (pinned, device-mapped) owns
 * a ring of `generations` x `rows` table rows in HBM; gf_cache_prefetch_blocks(descs) — same
 * descriptors as the coming gf_cache_fetch_blocks, d_out / d_stats / update ignored — pulls, on
 * `stream` (a side stream), the rows of the ids that are neither cached nor already in the ring,
 * one generation per call; the next gf_cache_fetch_blocks makes its stream wait for the pulls
 * issued so far and reads a missed row from the ring when it is there, from the host table
 * otherwise.  A hint: cache state, hit counts and the fetched rows never depend on it.
 * *issued = 0 when the call issued nothing (no ring, or the generation was dropped because
 * fetches that may still read the region it would overwrite had not finished).
 * generations: a power of two in 8..64, or 0 = ring off; the ring's index takes 16 bytes per
 * table row (where an id was staged last and the time before).  gf_cache_invalidate_staging: the
 * table's contents changed.  gf_cache_staging_state out[9]: generations, rows per generation,
 * generations issued, generations dropped, rows pulled over the host link (synchronises), bytes
 * of HBM the ring and its index take, rows the gathers still read from the host table,
 * microseconds the issuing thread waited for fetches to leave a region before reusing it,
 * fetches (of the process) whose stream had to wait for a pull's event (it had not landed). */
/* The cached rows' copy in HBM.  on (default): slot s holds a copy of its id's row, a hit reads
 * it (the reference's cache buffer, gnnflow/cache/cache.py:84-99).  off — only for a feature
 * table that is itself in device memory: the slots hold ids only, every row is read from the
 * table (a hit and a miss are the same bytes at the same distance there), an install moves
 * nothing; hit ratios and the set of cached ids — the state the reference's protocol lets a
 * caller observe — are unchanged.  gnnflow_amd.cache.Cache turns it off for
 * feature_placement="device"; pinned / sharded tables keep it. */
GF_API int gf_cache_set_row_mirror(gf_cache* c, int on);
/* Diagnostics of the one-launch LRU list update (lru_list_fused_kernel): with tracing on, every
 * workgroup of an update stamps the 100 MHz wall clock at its role's stages; gf_debug_lru_trace
 * copies the last update's stamps out — words [0..3] = count / row / write workgroups and the
 * launch tag, then 8 words per workgroup (0: entry; count role 1: inputs in, 2: published; row
 * role 1: ranked + published, 2: look-back complete, 3: installed, 4: rows copied; write role 1:
 * granules in, 2: list written).  scripts/lru_hop_trace.py prints the account. */
GF_API int gf_debug_lru_trace_enable(gf_cache* c, int on);
GF_API int gf_debug_lru_trace(gf_cache* c, uint64_t* out, size_t capacity_words, size_t* words);
GF_API int gf_cache_set_staging(gf_cache* c, size_t generations, size_t rows_per_generation);
GF_API int gf_cache_invalidate_staging(gf_cache* c);
GF_API int gf_cache_staging_state(gf_cache* c, uint64_t* out);
GF_API int gf_cache_prefetch_blocks(gf_cache* node_cache, gf_cache* edge_cache,
                                    const gf_fetch_desc* descs, size_t n, void* stream,
                                    int* issued);
/* ... through the enqueue thread that issues the asynchronous fetches (same order of issue);
 * the ticket is waited for with gf_cache_fetch_wait. */
GF_API int gf_cache_prefetch_blocks_async(gf_cache* node_cache, gf_cache* edge_cache,
                                          const gf_fetch_desc* descs, size_t n, void* stream,
                                          uint64_t* ticket);
/* One submission for a pipelined step: gf_cache_fetch_blocks_async(descs) of batch i followed by
 * the prefetch of a later batch (one hand-over to the enqueue thread, one ticket).  The later
 * batch is given by descriptors (next_descs) and / or by the block array a sample() of it filled
 * (next_blocks: [next_layers x next_snapshots] as gf_sampler_sample_end wrote it, copied before
 * the call returns): the node ids of the last layer's blocks and the edge ids of every block are
 * announced — the ring's index drops the ids that blocks share. */
GF_API int gf_cache_fetch_announce_async(gf_cache* node_cache, gf_cache* edge_cache,
                                         const gf_fetch_desc* descs, size_t n, void* stream,
                                         const gf_fetch_desc* next_descs, size_t next_n,
                                         const gf_block* next_blocks, size_t next_layers,
                                         size_t next_snapshots, void* prefetch_stream,
                                         uint64_t* ticket);
/* A pipelined loop announces batches `lag` + 1 steps before it fetches them, so that a pull has
 * landed when its fetch is issued: a fetch then depends on — waits for, and reads rows of — the
 * generations up to the newest one but `lag` (default 0: every generation issued so far). */
GF_API int gf_cache_set_staging_lag(gf_cache* c, size_t lag);
/* Diagnostics: cumulative time the enqueue thread spent issuing work, and jobs done. */
GF_API int gf_worker_stats(double* busy_us, uint64_t* jobs);

/* Cache-free gather, gnnflow/utils.py:465-474 prepare_input and
 * cache.py:411 `edge_feats[eid]`: out[i,:] = feats[ids[i],:]. */
GF_API int gf_gather_rows(const float* d_feats, size_t num_rows, size_t dim,
                          const int64_t* d_ids, size_t n, float* d_out, int device,
                          void* stream);

/* ---- TGN memory / mailbox: gnnflow/models/modules/memory.py -------- */
/* Memory.prepare_input (memory.py:156-190): mem = node_memory[ids], mem_ts =
 * node_memory_ts[ids], mail_ts = mailbox_ts[ids], mem_input = mailbox[ids] — the reference's
 * CPU unique + inverse scatter is only an optimisation of exactly that.  One launch.
 * dim_mail = 2 * dim_memory + dim_edge. */
GF_API int gf_memory_prepare_input(const float* d_node_memory, const float* d_node_memory_ts,
                                   const float* d_mailbox, const float* d_mailbox_ts,
                                   size_t num_nodes, size_t dim_memory, size_t dim_mail,
                                   const int64_t* d_ids, size_t n, float* d_mem, float* d_mem_ts,
                                   float* d_mail_ts, float* d_mem_input, int device,
                                   void* stream);
/* Memory.update_mem_mail (memory.py:192-269): d_nid / d_memory / d_ts are the
 * last_updated_{nid,memory,ts} of the n = (2 + neg_sample_ratio) * B roots; d_edge_feats
 * [B, dim_edge] or NULL (zeros).  Per distinct node the LAST occurrence wins (mailbox: in the
 * interleaved src0,dst0,src1,... order; memory: in src.. ++ dst.. order).  d_win_mail /
 * d_win_mem: caller-owned uint64[num_nodes] scratch, zero-initialised once; epoch: strictly
 * increasing, starting at 1. */
GF_API int gf_memory_update(float* d_node_memory, float* d_node_memory_ts, float* d_mailbox,
                            float* d_mailbox_ts, size_t num_nodes, size_t dim_memory,
                            size_t dim_edge, const int64_t* d_nid, const float* d_memory,
                            const float* d_ts, const float* d_edge_feats, size_t n,
                            int neg_sample_ratio, uint64_t* d_win_mail, uint64_t* d_win_mem,
                            uint64_t epoch, int device, void* stream);

/* Introspection for tests / get_mem_size (cache.py:136-155): copies the slot ->
 * id table (int64[capacity], -1 = empty) to a host array. */
GF_API int gf_cache_slot_ids(const gf_cache* c, int64_t* out, size_t capacity);
GF_API int gf_cache_mem_bytes(const gf_cache* c, size_t* out);
/* LRU bookkeeping state (no reference counterpart; diagnostics and tests).  out[0] = 1 if the
 * replacement order is kept as a queue with dead entries (caches of >= 0.5 M slots),
 * 0 if as a list; out[1] = entries allocated per queue buffer; out[2] = head, out[3] = tail of
 * the queue; out[4] = compactions so far; out[5] = updates of a queue-form cache that went
 * through the list form (block > capacity / 4 rows); out[6] = updates whose victim search had
 * to walk past the chunks behind the head.  Synchronises the device. */
GF_API int gf_cache_lru_state(const gf_cache* c, uint64_t out[7]);

/* ---- hash-partitioned sampling across the GPUs of a node ---------- */
/* The reference buckets a layer's roots by partition on the host and samples remote ones over
 * RPC (gnnflow/distributed/dist_sampler.py:159-314).  Here each rank runs, per layer:
 *   gf_partition_plan            bucket the roots by owner(v) = splitmix64(v) mod world_size
 *   (all-to-all-v of the request prefix — the caller's torch.distributed / RCCL call)
 *   gf_sampler_sample_layer_padded   on its own share (overlapping the exchange) and on the
 *                                    requests it received: fixed `fanout` slots per root
 *   (all-to-all-v of the replies)
 *   gf_sampler_merge_padded      replies -> the layer's block in the original root order,
 *                                bit-identical to single-GPU sampling (most-recent policy).
 * d_requests [n][2] int64 = (root id, root-ts bits in the low word), ordered
 * [other owners ascending | this rank]; d_pos[i] = row of root i; d_counts[world_size] device
 * words.  Replies [n][fanout][3] int64 = (dst, eid, ts | dt << 32), unused slots -1. */
GF_API int gf_partition_scratch_bytes(size_t num_roots, int world_size, size_t* out);
GF_API int gf_partition_plan(const int64_t* d_nodes, const float* d_ts, size_t num_roots,
                             int world_size, int rank, int64_t* d_requests, uint32_t* d_pos,
                             uint64_t* d_counts, void* d_scratch, size_t scratch_bytes,
                             int device, void* stream);
GF_API int gf_sampler_sample_layer_padded(gf_sampler* s, const int64_t* d_requests, size_t n,
                                          uint32_t layer, uint32_t snapshot, int64_t* d_out,
                                          void* stream);
/* d_out / out_bytes / block as gf_sampler_sample_layer (gf_sampler_layer_output_bytes);
 * synchronises `stream` once to read the edge count. */
GF_API int gf_sampler_merge_padded(gf_sampler* s, const int64_t* d_roots, const float* d_ts,
                                   size_t n, uint32_t layer, const int64_t* d_replies,
                                   const uint32_t* d_pos, void* d_out, size_t out_bytes,
                                   gf_block* block, void* stream);

/* Chained form of the same exchange — no read-back between the layers (reference loop:
 * gnnflow/distributed/dist_sampler.py:129-157 sample(), one RPC round per layer):
 *   gf_sampler_part_begin
 *   for every (layer, snapshot):
 *     gf_sampler_part_plan_own    bucket the layer's roots (their count is device resident:
 *                                 the previous layer's R + S) and sample this rank's own share
 *     — the caller exchanges: counts, request rows [0, R - counts[rank]) of the workspace out,
 *       gf_sampler_sample_layer_padded on what it received, reply rows back into the workspace —
 *     gf_sampler_part_merge       replies -> the layer's block, sizes stay on the device
 *   gf_sampler_part_commit        publishes the sizes; gf_sampler_sample_end returns the blocks
 * A rank with no roots (R = 0) still takes part in every step.  With one rank
 * gf_sampler_sample_partitioned issues the whole chain (`d_ws`: the layouts' totals, layer
 * after layer, snapshot after snapshot).  Offsets of one (layer, snapshot) workspace: */
typedef struct gf_part_layout {
  size_t root_bound;     /* worst-case roots of the layer when sample() starts from R0 roots */
  size_t requests;       /* [root_bound][2] int64, ordered [other owners ascending | own] */
  size_t replies;        /* [root_bound][fanout][3] int64, same row order */
  size_t counts;         /* [world_size] uint64: roots per owner */
  size_t pos;            /* [root_bound] uint32: row of root i */
  size_t scratch;
  size_t scratch_bytes;
  size_t total;          /* bytes of this workspace */
  /* slotted form only (0 otherwise): */
  size_t slot_stride;    /* request / reply rows per peer slot = 1 header row + capacity */
  size_t inbox;          /* [world_size][slot_stride][2] int64: the requests this rank received */
  size_t served;         /* [world_size][slot_stride][fanout][3] int64: its replies to them */
} gf_part_layout;
GF_API int gf_sampler_part_layout(const gf_sampler* s, size_t num_roots, uint32_t layer,
                                  int world_size, gf_part_layout* out);
GF_API int gf_sampler_part_begin(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                                 size_t num_roots, void* d_out, size_t out_bytes,
                                 int world_size, int rank, void* stream);
/* phases: 1 = bucket the roots, 2 = sample this rank's own share (call it right after starting
 * the request all-to-all-v: the two overlap), 3 = both */
GF_API int gf_sampler_part_plan_own(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                                    size_t ws_bytes, int phases);
GF_API int gf_sampler_part_merge(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                                 size_t ws_bytes);
GF_API int gf_sampler_part_commit(gf_sampler* s);
GF_API int gf_sampler_part_abort(gf_sampler* s);

/* Slotted form of the chained exchange: NO host synchronisation inside a sample.  The
 * variable-size exchange above needs each layer's per-owner counts on the host (they are the
 * split sizes of the all-to-all-v).  Here every peer has a slot of FIXED capacity — slack x
 * the even share of the layer's worst-case root count for a batch of `slot_roots` roots, a
 * number all ranks agree on once (their own batches may be smaller or larger: a larger one
 * merely overflows sooner) — in the request and reply buffers:
 *   requests  [world_size slots][slot_stride rows][2] int64, row 0 of a slot = its header
 *             {rows that follow, bit 0: a slot of the sender overflowed}, then this rank's own
 *             share from row world_size * slot_stride on; replies: the same rows x fanout x 3
 * so the exchanges are equal-split all-to-alls whose sizes the host knows in advance:
 *   gf_sampler_part_begin_slotted
 *   per (layer, snapshot):
 *     gf_sampler_part_plan_own(phase 1)   bucket into the slots, write the headers
 *     — all-to-all, equal split: request rows [0, world_size * slot_stride) -> `inbox` —
 *     gf_sampler_part_plan_own(phase 2)   own share, overlapping the exchange
 *     gf_sampler_part_serve               inbox -> `served` (reply row = request row)
 *     — all-to-all, equal split: `served` -> reply rows [0, world_size * slot_stride) —
 *     gf_sampler_part_merge
 *   gf_sampler_part_commit; gf_sampler_sample_end; gf_sampler_part_overflowed
 * A slot that would overflow drops the excess roots and raises a flag that travels in the
 * headers, so EVERY rank sees it in the same exchange; gf_sampler_part_overflowed reports it
 * for the sample gf_sampler_sample_end returned last, and all ranks then sample that batch
 * again through the variable-size exchange (no reference counterpart: the reference's RPC
 * futures are variable-size by construction, gnnflow/distributed/dist_sampler.py:188-242). */
GF_API int gf_sampler_part_layout_slotted(const gf_sampler* s, size_t num_roots, uint32_t layer,
                                          int world_size, double slack, size_t slot_roots,
                                          gf_part_layout* out);
/* What one slot of a shared chain's exchanges takes on the wire for `layer` (the layout the
 * native chains use, sampler.hip group_layout): out[0] = rows per request slot (16 B each, header
 * row included), out[1] = bytes of one reply slot (compact: offsets + packed edges; edge_fill 0:
 * the fixed records), out[2] = edges a compact reply slot holds at most, out[3] = bytes of its
 * offsets (2 / 4).  narrow: bit 0 = 12-byte reply records, bit 1 = layer l + 1 does not request
 * layer l's roots again (its slots are sized for the roots that are new in it).
 * gnnflow_amd.dist.DevicePartitionedSampler.wire_bytes_per_sample reports from this, so the
 * figures in bench.py's line are the native layout's, not a restatement. */
GF_API int gf_sampler_part_group_slot(const gf_sampler* s, size_t num_roots, uint32_t layer,
                                      int world_size, double slack, size_t slot_roots, int narrow,
                                      double edge_fill, uint64_t* out);
GF_API int gf_sampler_part_begin_slotted(gf_sampler* s, const int64_t* d_roots,
                                         const float* d_root_ts, size_t num_roots, void* d_out,
                                         size_t out_bytes, int world_size, int rank, double slack,
                                         size_t slot_roots, void* stream);
GF_API int gf_sampler_part_serve(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                                 size_t ws_bytes);
GF_API int gf_sampler_part_overflowed(const gf_sampler* s, int* out);
GF_API int gf_sampler_sample_partitioned(gf_sampler* s, const int64_t* d_roots,
                                         const float* d_root_ts, size_t num_roots, void* d_out,
                                         size_t out_bytes, void* d_ws, size_t ws_bytes,
                                         void* stream);
/* ... the same, issued by the library's enqueue thread (see gf_sampler_sample_begin_async) */
GF_API int gf_sampler_sample_partitioned_async(gf_sampler* s, const int64_t* d_roots,
                                               const float* d_root_ts, size_t num_roots,
                                               void* d_out, size_t out_bytes, void* d_ws,
                                               size_t ws_bytes, void* stream);

/* ---- RCCL communicator of the partitioned path ------------------------------------------- */
/* The exchanges above issued by the library itself over RCCL / xGMI (the reference's transport
 * is torch RPC between machines, gnnflow/distributed/dist_sampler.py:188-242).  RCCL is resolved
 * at run time (the librccl.so.1 already in the process, else ROCm's), so a single-GPU user never
 * loads it.  Bootstrap: rank 0 calls gf_comm_unique_id and hands the 128 bytes to every rank by
 * whatever channel it has (torch.distributed broadcast, a file, MPI); then EVERY rank calls
 * gf_comm_create (collective).  gf_comm_all_to_all: bytes_per_peer bytes to / from every rank
 * (equal split, this rank included), ordered on `stream`; gf_comm_all_to_all_v: host arrays of
 * world_size byte counts / byte offsets. */
GF_API int gf_comm_unique_id(uint8_t out[128]);
GF_API int gf_comm_create(gf_comm** out, const uint8_t id[128], int world_size, int rank,
                          int device);
GF_API int gf_comm_destroy(gf_comm* c);
GF_API int gf_comm_all_to_all(gf_comm* c, const void* d_send, void* d_recv, size_t bytes_per_peer,
                              void* stream);
GF_API int gf_comm_all_to_all_v(gf_comm* c, const void* d_send, const size_t* send_bytes,
                                const size_t* send_offsets, void* d_recv, const size_t* recv_bytes,
                                const size_t* recv_offsets, void* stream);
/* What the transport itself reports: out = {ranks, this rank, device, kind (0 RCCL, 1 hipIpc, 2
 * loopback)}; for RCCL these are ncclCommCount / ncclCommUserRank / ncclCommCuDevice — bench.py
 * puts them into the N > 1 line ("did RCCL see N ranks on N devices?").  The reference all-gathers
 * its per-rank sampling timers the same way (gnnflow/distributed/dist_sampler.py:108-127). */
GF_API int gf_comm_info(gf_comm* c, int32_t out[4]);
/* Gives the communicator up without waiting for its peers (ncclCommAbort): for a rank whose
 * collective never completes.  Only gf_comm_destroy may follow. */
GF_API int gf_comm_abort(gf_comm* c);
/* `iters` back-to-back equal-split all-to-alls of `bytes_per_peer` on scratch buffers: device time
 * (events around the batch) and the issuing thread's host time, per exchange, in microseconds —
 * the x and c of DESIGN 6.2's projection.  Collective. */
GF_API int gf_comm_time_all_to_all(gf_comm* c, size_t bytes_per_peer, int iters, void* stream,
                                   double* device_us, double* host_us);
/* PCI bus id of a HIP device ("0000:c1:00.0"), len >= 16. */
GF_API int gf_device_pci_bus_id(int device, char* out, size_t len);
/* Do two streams of `device` run on the same hardware queue?  HIP spreads its streams over four
 * hardware queues and kernels of streams that share one run one after the other: a side stream
 * meant to work BESIDE another (a sampling lane, the staging ring's pull stream) must not share
 * its queue.  The probe holds stream `a` with a kernel that spins for `spin_us` microseconds of
 * wall clock and looks whether a one-thread kernel on stream `b` finishes meanwhile.
 * *shared = 1: it did not (same queue, or the device would not run the two side by side).
 * Synchronises both streams.  gnnflow_amd.pipeline.side_stream picks its streams with it. */
GF_API int gf_streams_share_queue(int device, void* a, void* b, unsigned spin_us, int* shared);
/* The same exchanges between processes that map each other's device memory (hipIpc*) — the ranks
 * of one node, several ranks SHARING one GPU included (where RCCL refuses to run): a
 * host-synchronising test / single-box transport (copy-out, process barrier over POSIX shared
 * memory, copy-in from the peers' mailboxes, barrier).  Every rank: gf_ipc_comm_create (same
 * `shm_name`, "/..."; `mailbox_bytes` = its largest message), gf_ipc_comm_handle -> 64 bytes to
 * publish to all ranks, gf_ipc_comm_open(all handles, rank-major).  The handle then works with
 * every gf_comm_* / *_comm entry point. */
GF_API int gf_ipc_comm_create(gf_comm** out, int world_size, int rank, int device,
                              size_t mailbox_bytes, const char* shm_name);
GF_API int gf_ipc_comm_handle(gf_comm* c, uint8_t out[64]);
GF_API int gf_ipc_comm_open(gf_comm* c, const uint8_t* handles);

/* The same exchanges between "ranks" that live in ONE process (world_size handles, rank r =
 * out[r], all on `device`): each rank is driven by its own host thread, an exchange is a thread
 * barrier + device copies straight out of the peers' send buffers.  Host-synchronising test
 * transport: it runs the native multi-rank chains at world sizes a one-GPU box cannot give as
 * processes (8 ranks).  Works with every gf_comm_* / *_comm entry point except the *_async
 * ones (one enqueue thread cannot serve ranks that wait for each other).  No reference
 * counterpart (the reference has no distributed test). */
GF_API int gf_loopback_comm_create(gf_comm** out, int world_size, int device);

/* The slotted chain of one sample() over `c` in ONE call (and through the enqueue thread):
 * gf_sampler_part_begin_slotted, then per (layer, snapshot) plan -> request slots out -> own
 * share + serve -> reply slots back -> merge, then commit; nothing is read back.
 * d_ws: the slotted layouts' totals, layer after layer, snapshot after snapshot.
 * overlap != 0: the exchanges run on the communicator's own stream, the request exchange while
 * this rank's own share is sampled on `stream`.  gf_sampler_sample_end + gf_sampler_part_overflowed
 * complete it. */
GF_API int gf_sampler_sample_partitioned_comm(gf_sampler* s, gf_comm* c, const int64_t* d_roots,
                                              const float* d_root_ts, size_t num_roots,
                                              void* d_out, size_t out_bytes, void* d_ws,
                                              size_t ws_bytes, double slack, size_t slot_roots,
                                              int overlap, void* stream);
GF_API int gf_sampler_sample_partitioned_comm_async(gf_sampler* s, gf_comm* c,
                                                    const int64_t* d_roots, const float* d_root_ts,
                                                    size_t num_roots, void* d_out, size_t out_bytes,
                                                    void* d_ws, size_t ws_bytes, double slack,
                                                    size_t slot_roots, int overlap, void* stream);

/* Up to GF_PART_GROUP_MAX samples in ONE chain: at batch 600 the chain's throughput is bound by
 * the host thread that issues its ~11 stream operations, so m consecutive batches share their
 * launches and their exchanges — sample j through its own sampler `samples[j].sampler` (m
 * samplers over the same graph with the same arguments: each keeps its own output, block
 * counters and publish record; gf_sampler_sample_end / gf_sampler_part_overflowed per sampler as
 * usual).  Owner q's requests of sample j travel in slot m q + j of ONE buffer, so one
 * equal-split all-to-all moves every sample's slots; 11 operations per m samples.
 * gf_sampler_part_group_ws_bytes: size of the shared exchange workspace for samples of
 * roots[0..m) roots, 0 if they cannot share a chain (several snapshots, a layer beyond 32 768
 * roots, fanout > 256): the caller then issues single chains.  force_overflow (bit j: sample
 * j): flag that sample as overflowed whatever its slots hold — for a batch that is too large for
 * this chain although the batch size the ranks agreed on is not, the caller submits an EMPTY
 * stand-in with this bit set, every rank sees the flag in the same exchange and the real batch
 * is sampled in the redo.  c == NULL: ONE rank with nothing to exchange (world size 1, every root
 * its own) — the same chain without the two all-to-alls.  narrow_ids != 0: reply slots of 12
 * bytes {destination, edge id, edge time} instead of 24 — half the bytes of the reply exchange;
 * the SAME value on every rank (it is part of the wire format), and only while every rank's
 * shard satisfies gf_graph_ids_fit_u32 (a rank whose shard stops fitting sets every bit of
 * force_overflow: all ranks then redo those samples through the wide variable-size form).
 * No reference counterpart (its RPC
 * futures are per partition and per call, gnnflow/distributed/dist_sampler.py:188-220). */
#define GF_PART_GROUP_MAX 4
typedef struct gf_group_sample {
  gf_sampler* sampler;
  const int64_t* d_roots;
  const float* d_root_ts;
  size_t num_roots;
  void* d_out;          /* gf_sampler_output_bytes(num_roots) bytes, as for gf_sampler_sample */
  size_t out_bytes;
} gf_group_sample;
/* `narrow_ids` of the three entry points below is a flags word: bit 0 = 12-byte reply records;
 * bit 1 = a layer does not request the previous layer's roots again (its first roots ARE those,
 * with the same timestamps: with most-recent sampling and equal fanouts the merge takes their
 * edges from the previous layer's block; the reference requests every root of every layer,
 * gnnflow/distributed/dist_sampler.py:174-186);
 * bits 8..23 = COMPACT replies, the edge fill in 1/1000: a reply slot that travels back then holds,
 * per request row, the offset of its edges and, packed behind the offsets, the edges themselves —
 * at most fill x (slot rows x fanout) of them — instead of `fanout` fixed records per row, most of
 * which are empty (the reference ships back exactly a partition's sampled edges,
 * gnnflow/distributed/common.py:4-19, dist_sampler.py:244-314).  A slot with more edges flags the
 * sample as overflowed on every rank (it is redone through the variable-size exchange).  Part of
 * the wire format: the same on every rank; 0 = the fixed records travel. */
GF_API int gf_sampler_part_group_ws_bytes(const gf_sampler* s, const size_t* roots, int m,
                                          int world_size, double slack, size_t slot_roots,
                                          int narrow_ids, size_t* bytes);
GF_API int gf_sampler_sample_partitioned_comm_group(gf_comm* c, const gf_group_sample* samples,
                                                    int m, void* d_ws, size_t ws_bytes,
                                                    double slack, size_t slot_roots,
                                                    int force_overflow, int narrow_ids,
                                                    void* stream);
/* ... issued by the library's enqueue thread (like gf_sampler_sample_partitioned_comm_async) */
GF_API int gf_sampler_sample_partitioned_comm_group_async(gf_comm* c,
                                                          const gf_group_sample* samples, int m,
                                                          void* d_ws, size_t ws_bytes,
                                                          double slack, size_t slot_roots,
                                                          int force_overflow, int narrow_ids,
                                                          void* stream);

/* ---- message passing on a sampled block ---------------------- */
/* The DGL calls of the reference's layers on an MFG (gnnflow/models/modules/layers.py:153-159,
 * models/graphsage.py:27-31, models/gat.py:28-46), as segment operations: a block's edges are
 * grouped by destination (`row` non-decreasing, as the sampler emits them).  fp32, device
 * pointers, ordered on `stream`.
 * offsets[d] = first edge with row >= d, d = 0..num_dst (so offsets[num_dst] = num_edges). */
GF_API int gf_block_segment_offsets(const int64_t* d_row, size_t num_edges, size_t num_dst,
                                    int64_t* d_offsets, int device, void* stream);
/* dgl.ops.edge_softmax(block, logits[num_edges, heads]): softmax over each destination's edges */
GF_API int gf_block_edge_softmax(const int64_t* d_offsets, size_t num_dst, size_t num_edges,
                                 size_t heads, const float* d_logits, float* d_out, int device,
                                 void* stream);
GF_API int gf_block_edge_softmax_backward(const int64_t* d_offsets, size_t num_dst,
                                          size_t num_edges, size_t heads, const float* d_out,
                                          const float* d_grad_out, float* d_grad_logits,
                                          int device, void* stream);
/* update_all(copy_src | u_mul_e, sum | mean): out[d, :] = reduce over edges k of d of
 * edge_weight[k, head] * src[col[k], :]  (d_edge_weight NULL: copy_src; else [num_edges, heads]
 * with dim % heads == 0, each weight covering dim / heads consecutive columns).
 * d_col NULL = the sampler's own layout col[k] = num_dst + k (then num_src must be
 * num_dst + num_edges): contiguous reads, and a backward pass without atomics. */
GF_API int gf_block_reduce(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                           const float* d_src, size_t dim, const float* d_edge_weight,
                           size_t heads, int mean, float* d_out, int device, void* stream);
/* d_grad_src [num_src, dim] is zeroed then accumulated (NULL: skipped); d_grad_edge_weight
 * [num_edges, heads] or NULL. */
GF_API int gf_block_reduce_backward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                                    const float* d_src, size_t dim, const float* d_edge_weight,
                                    size_t heads, int mean, const float* d_grad_out,
                                    float* d_grad_src, size_t num_src,
                                    float* d_grad_edge_weight, int device, void* stream);

/* update_all(copy_src, max) — dgl.nn.SAGEConv's 'pool' aggregator: out[d, c] = max over the
 * in-edges of src[col[k], c] (0 without in-edges); d_arg [num_dst, dim] receives the winning
 * edge per element (-1: none), which the backward pass routes the gradient through. */
GF_API int gf_block_reduce_max(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                               const float* d_src, size_t dim, float* d_out, int64_t* d_arg,
                               int device, void* stream);
GF_API int gf_block_reduce_max_backward(size_t num_dst, const int64_t* d_col, size_t dim,
                                        const float* d_grad_out, const int64_t* d_arg,
                                        float* d_grad_src, size_t num_src, int device,
                                        void* stream);

/* ---- measurement support (bench.py) ---------------------------------------- */
/* Accumulated device time of a kernel family since the last reset, measured with
 * HIP events recorded around each launch on the launching stream.
 * which: 0 = sampler search, 1 = sampler emit, 2 = feature gather, 3 = sampler scan,
 * 4 = LRU update (all bookkeeping kernels of one fetch as one interval).
 * gf_profile_enable takes a bit mask of the families to time (0 = off). */
GF_API int gf_profile_enable(int mask);
GF_API int gf_profile_reset(void);
GF_API int gf_profile_get(int which, double* total_ms, uint64_t* launches);
/* Time only every stride-th interval of a family (default 1 = all): an event pair costs
 * host time and a few microseconds of stream time per launch, which a latency-bound
 * workload feels; gf_profile_get then reports the sampled intervals. */
GF_API int gf_profile_set_stride(unsigned stride);
/* Host time the issuing thread spent per stage of the slotted partitioned chain since the last
 * reset: out[0..6] = begin, plan, request exchange, serve, reply exchange, merge, commit
 * (microseconds, summed over the samples), out[7] = samples.  Diagnostics: the chain is ~13
 * stream operations issued by one thread, which bounds its throughput at small batches. */
GF_API int gf_debug_part_host_us(double out[8], int reset);
/* Tiles of the fused partitioned merge whose look-back granule did not arrive within the polling
 * budget and were recounted by the waiting thread instead (current device, since the library
 * was loaded).  0 in normal operation; non-zero when the GPU is heavily oversubscribed (several
 * rank processes sharing one card).  Synchronises nothing but the copy itself. */
GF_API int gf_debug_merge_recounts(uint64_t* out);
/* Roots the plans of the shared partitioned chains did NOT request because the previous layer's
 * block already holds their edges (flags bit 1 of the group entry points), summed since the
 * library was loaded, current device.  Diagnostics / tests. */
GF_API int gf_debug_part_reused_roots(uint64_t* out);
/* d_out[i] = gf_philox4x32_10_first(seed, slot, call) of d_in[i] = {seed, slot, call}, evaluated by
 * a kernel: the Random123 known-answer vectors checked ON THE DEVICE (include/gnnflow_rng.h is
 * shared with the CPU oracle, so parity alone would not catch a device-side miscompile). */
GF_API int gf_debug_philox(const uint64_t* d_in, size_t n, uint32_t* d_out, void* stream);
/* The same for the one-launch LRU list update (lru_list_fused_kernel): granules {launch tag,
 * count} a waiting workgroup did not see within its polling budget and recomputed from the
 * launch's inputs.  0 in normal operation. */
GF_API int gf_debug_lru_recounts(uint64_t* out);
/* All launches of a family seen since the last reset while it was enabled, timed or not. */
GF_API int gf_profile_launches(int which, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GNNFLOW_HIP_H_ */
