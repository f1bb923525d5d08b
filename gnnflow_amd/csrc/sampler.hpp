// TemporalSampler host driver — MI355X counterpart of
// gnnflow/csrc/temporal_sampler.{h,cu} + sampling_kernels.cu.
#pragma once

#include <cstdint>
#include <mutex>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "edge_store.hpp"

namespace gf {

class Sampler {
 public:
  Sampler(EdgeStore* graph, const uint32_t* fanouts, size_t num_layers, int policy,
          uint32_t num_snapshots, float window, bool prop_time, uint64_t seed);
  ~Sampler();

  size_t num_layers() const { return fanouts_.size(); }
  uint32_t fanout(size_t layer) const { return fanouts_[layer]; }
  uint32_t num_snapshots() const { return num_snapshots_; }
  int device() const { return graph_->device(); }

  // worst-case number of roots entering `layer` when sample() starts from R roots
  size_t root_bound(size_t R, size_t layer) const;
  size_t layer_output_bytes(size_t num_roots, size_t layer) const;  // one snapshot
  size_t output_bytes(size_t num_roots) const;                      // all layers x snapshots

  void sample(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
              size_t out_bytes, gf_block* blocks, hipStream_t stream);
  // Split form: begin() enqueues every kernel plus the size read-back and returns; end()
  // waits for the OLDEST begun sample and reports its block sizes.  Up to kMaxInFlight
  // samples may be begun before the first end() — they must all use the same stream (the
  // kernels share one workspace, which stream order keeps consistent).
  static constexpr size_t kMaxInFlight = 4;
  void sample_begin(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                    size_t out_bytes, hipStream_t stream);
  void sample_end(gf_block* blocks);
  size_t in_flight() const;
  void sample_layer(const int64_t* d_roots, const float* d_ts, size_t R, uint32_t layer,
                    uint32_t snapshot, void* d_out, size_t out_bytes, gf_block* block,
                    hipStream_t stream);
  // ---- partitioned sampling: the owner's and the requester's halves -------
  // Owner: samples n packed requests (root id, root-ts bits) on this shard into a FIXED
  // `fanout` slots per root: d_out[n][fanout][3] int64 = (dst, eid, ts | dt << 32), unused
  // slots -1.  The output size follows from n, so nothing is read back: no host sync.
  void sample_layer_padded(const int64_t* d_requests, size_t n, uint32_t layer,
                           uint32_t snapshot, int64_t* d_out, hipStream_t stream);
  // Requester: turns the replies (rows in owner-sorted order, d_pos[i] = row of root i) into
  // the layer's block in the ORIGINAL root order, laid out like sample_layer's output.
  void merge_padded(const int64_t* d_roots, const float* d_ts, size_t R, uint32_t layer,
                    const int64_t* d_replies, const uint32_t* d_pos, void* d_out,
                    size_t out_bytes, gf_block* block, hipStream_t stream);
  // ---- the same, chained on the device (no read-back between layers; sampler.hip) --------
  // slack > 0: the slotted form (fixed-capacity slots per peer, equal-split exchange, no count
  // read-back; partition.hip), capacity = slack x the even share of the layer's worst case
  // of a sample that starts from `slot_roots` roots — the same number on every rank
  // skip_prev: the slots are sized without the previous layer's roots (see group_layout)
  void part_layout(size_t R0, uint32_t layer, int world_size, double slack, size_t slot_roots,
                   gf_part_layout* out, bool skip_prev = false) const;
  void part_begin(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                  size_t out_bytes, int world_size, int rank, double slack, size_t slot_roots,
                  hipStream_t stream);
  // with_own: also sample this rank's own share in the same launch (instead of part_plan_own
  // phase 2)
  void part_serve(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes,
                  bool with_own = false);
  // slotted form: did a slot overflow anywhere in the sample sample_end() returned last
  bool last_overflow() const { return last_overflow_; }
  // phases: 1 = bucket the roots, 2 = sample this rank's own share, 3 = both
  void part_plan_own(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes, int phases);
  void part_merge(uint32_t layer, uint32_t snapshot, void* d_ws, size_t ws_bytes);
  void part_commit();
  void part_abort();
  // Up to kMaxGroup samples in ONE chain (sampler.hip "SHARE their launches and exchanges"):
  // sample j runs through its own sampler — all over the same graph with the same arguments;
  // shared launches, exchanges and exchange workspace, separate outputs / counters / publish
  // records.
  static constexpr int kMaxGroup = 4;
  struct GroupSample {
    Sampler* s;
    const int64_t* d_roots;
    const float* d_ts;
    size_t R;
    void* d_out;
    size_t out_bytes;
  };
  struct GroupLayout {
    size_t stride, slot_rows, own[kMaxGroup];                                   // rows
    size_t requests, replies, inbox, served, counts[kMaxGroup], pos[kMaxGroup], total;  // bytes
    size_t first[kMaxGroup];   // u32 [roots + 1]: first edge of every root in the merged block
    // compact replies (part_edge_fill_ > 0): what travels back is, per slot, the rows' edge
    // offsets + the edges packed behind them — cslot bytes per slot, edge_cap edges at most
    size_t edge_cap, off_bytes, cslot, row_cnt, cserved, creplies;
  };
  // edge_fill: compact reply slots — a slot carries at most this share of its stride x fanout
  // fixed-fanout records (0: the fixed records themselves travel).  Part of the wire format.
  bool group_ok(const size_t* R, int m) const;
  // narrow: 12-byte reply slots (ids that fit 32 bits; sampler.hip PaddedCommon)
  // reuse_roots: layer l + 1 does not request layer l's roots again (most-recent, equal
  // fanouts) — its slots are sized for the roots it still requests.  Part of the wire format.
  bool layer_reuses_roots(bool reuse_roots, size_t layer) const {
    return reuse_roots && layer > 0 && policy_ != GF_SAMPLING_POLICY_UNIFORM &&
           fanouts_[layer] == fanouts_[layer - 1];
  }
  void group_layout(const size_t* R, int m, uint32_t layer, int world, double slack,
                    size_t slot_roots, bool narrow, double edge_fill, GroupLayout* out,
                    bool reuse_roots = false) const;
  static size_t group_ws_bytes(const Sampler& a, const size_t* R, int m, int world, double slack,
                               size_t slot_roots, bool narrow, double edge_fill = 0.0,
                               bool reuse_roots = false);
  static void sample_partitioned_group(const GroupSample* gs, int m, void* d_ws, size_t ws_bytes,
                                       double slack, size_t slot_roots, Exchange* ex,
                                       hipStream_t stream, unsigned force_overflow = 0,
                                       bool narrow = false, double edge_fill = 0.0,
                                       bool reuse_roots = false);
  void sample_partitioned(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                          size_t out_bytes, void* d_ws, size_t ws_bytes, hipStream_t stream);
  // several ranks, slotted form, the exchanges issued through `ex` (RCCL): the whole chain in
  // one call, nothing read back.  overlap: the request exchange runs on the communicator's
  // stream while this rank's own share is sampled on `stream`.
  void sample_partitioned_slotted(const int64_t* d_roots, const float* d_ts, size_t R, void* d_out,
                                  size_t out_bytes, void* d_ws, size_t ws_bytes, double slack,
                                  size_t slot_roots, Exchange& ex, bool overlap,
                                  hipStream_t stream);
  // host-vector forms (reference calling convention)
  void sample_host(const int64_t* nodes, const float* ts, size_t R, gf_block* blocks);
  void sample_layer_host(const int64_t* nodes, const float* ts, size_t R, uint32_t layer,
                         uint32_t snapshot, gf_block* block);

 private:
  struct BlockPtrs {
    int64_t* all_nodes; float* all_ts; float* dt; int64_t* eids; int64_t* row; int64_t* col;
  };
  BlockPtrs carve(char* base, size_t Rb, uint32_t fanout) const;
  // Enqueues one (layer, snapshot); R comes from d_R (device) when non-null.
  void enqueue_layer(const int64_t* d_roots, const float* d_ts, size_t R_bound,
                     const uint64_t* d_R, uint64_t R_host, uint32_t layer, uint32_t snapshot,
                     const BlockPtrs& out, uint64_t* d_counts_slot, uint64_t* next_R,
                     hipStream_t stream, const void* publish);
  void reserve_workspace(size_t R_bound_max, size_t num_blocks, hipStream_t stream);
  void to_host_blocks(const gf_block* dev, gf_block* host, size_t n, hipStream_t stream);

  EdgeStore* graph_;
  std::vector<uint32_t> fanouts_;
  int policy_;
  uint32_t num_snapshots_;
  float window_;
  bool prop_time_;
  uint64_t seed_;
  DeviceBuffer part_ticket_;   // reply_compact_kernel: "last slot of the sample" tickets
  uint32_t part_tag_ = 0;      // ... tagged per launch
  uint64_t calls_ = 0;  // sample_layer invocations so far (uniform RNG counter)
 public:
  uint64_t call_counter() const { return calls_; }
  void set_call_counter(uint64_t v) { calls_ = v; }
 private:
  int search_group_ = 16;   // lanes per root, layers of <= 32 768 roots
  int large_group_ = 4;     // lanes per root, larger layers (sampler.hip: group_width_from_env)
  bool fused_scan_ = true;
  bool hybrid_search_ = true;   // large layers: lane-per-root search, groups for the hubs
  // begun, not yet ended samples (FIFO).  begin() may run on the library's enqueue thread
  // while end() runs on the caller's: the ring bookkeeping is under ring_mu_.
  struct InFlight {
    size_t roots = 0;
    uint64_t seq = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool by_event = false;   // sizes published by the sample's last kernel, completion by `done`
    std::vector<BlockPtrs> ptrs;
  };
  struct PartState {
    bool active = false;
    InFlight* slot = nullptr;
    const int64_t* d_roots = nullptr;
    const float* d_ts = nullptr;
    size_t R = 0, Rs = 1;
    int world = 1, rank = 0;
    double slack = 0.0;
    size_t slot_roots = 0;
    hipStream_t stream = nullptr;
  };
  uint64_t* part_counts() const;
  uint32_t* part_overflow() const;
  bool last_overflow_ = false;
  uint32_t* part_rec_cnt() const;
  uint32_t* part_root_of() const;
  bool part_own_counts(size_t root_bound) const;
  // slotted form, small layers: count + prefix + emit of the merge in ONE launch (granules)
  bool part_fused_merge(size_t root_bound, uint32_t fanout) const;
  void part_commit_prepare(void* publish_out);   // Publish record of the sample being built
  void part_commit_finish();
  void part_roots(uint32_t layer, uint32_t snapshot, const int64_t** roots, const float** ts,
                  const uint64_t** d_R, uint64_t* R_host) const;
  InFlight ring_[kMaxInFlight];
  size_t ring_head_ = 0, ring_count_ = 0;
  mutable std::mutex ring_mu_;
  PartState part_;
  uint64_t publish_seq_ = 0;
  size_t rec_words_ = 0;   // uint64 words per pinned publish record: flag + 2 per block

  DeviceBuffer ws_;        // per-root search records + scan scratch + counters
  DeviceBuffer hub_buf_;   // large layers: worklist of the roots with long segments
  size_t ws_roots_ = 0, ws_blocks_ = 0;
  PinnedBuffer h_counts_;        // kMaxInFlight publish records (flag + block sizes)
  PinnedBuffer h_layer_counts_;  // blocking single-layer calls
  RetiredBuffers retired_;       // workspace replaced while kernels may still use it
  DeviceBuffer host_io_;   // device buffers behind the *_host entry points
  hipStream_t own_stream_ = nullptr;
};

}  // namespace gf
