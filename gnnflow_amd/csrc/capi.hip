// extern "C" entry points of include/gnnflow_hip.h: thin, exception-free shims
// over EdgeStore / Sampler / FeatureCache.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <unistd.h>

#include "comm.hpp"
#include "common.hpp"
#include "edge_store.hpp"
#include "feature_cache.hpp"
#include "partition.hpp"
#include "sampler.hpp"

struct gf_graph { gf::EdgeStore impl; template <typename... A> explicit gf_graph(A&&... a) : impl(std::forward<A>(a)...) {} };
struct gf_sampler { gf::Sampler impl; std::deque<uint64_t> begin_tickets; /* 0 = begun synchronously */ int plain_lane = 1; /* enqueue thread of sample_begin_async (gf_sampler_set_enqueue_lane) */ template <typename... A> explicit gf_sampler(A&&... a) : impl(std::forward<A>(a)...) {} };
struct gf_cache { gf::FeatureCache impl; template <typename... A> explicit gf_cache(A&&... a) : impl(std::forward<A>(a)...) {} };
struct gf_comm {
  std::unique_ptr<gf::Exchange> owned;
  gf::Exchange& impl;
  gf::IpcExchange* ipc = nullptr;
  gf_comm(const uint8_t* id, int world, int rank, int device)
      : owned(new gf::RcclComm(id, world, rank, device)), impl(*owned) {}
  gf_comm(gf::IpcExchange* x) : owned(x), impl(*owned), ipc(x) {}
  gf_comm(gf::LoopbackExchange* x) : owned(x), impl(*owned), loopback(true) {}
  bool loopback = false;   // ranks are threads of this process: no shared enqueue thread
};
struct gf_pull_session {
  gf::PullSession impl;
  // over RCCL the round is issued by the enqueue thread that also issues the partitioned
  // sampler's chains: ONE global order of collectives over all communicators, on every rank
  bool ordered = false;
  gf_pull_session(gf::Exchange* ex, int device) : impl(ex, device) {}
};

namespace gf {

namespace {
thread_local std::string g_last_error;

// A process forked from one that holds handles (multiprocessing's fork start method: a Manager
// server, a DataLoader worker) inherits the Python objects and may finalise them — its garbage
// collector runs their __del__.  The GPU state behind a handle belongs to the process that
// loaded the library: in any other process a destroy call is a no-op (the child's copy of the
// host memory goes with the process), it must never free the parent's device memory.
const pid_t g_load_pid = getpid();
inline bool foreign_process() { return getpid() != g_load_pid; }

struct ProfileRecord { int slot; hipEvent_t start, stop; };
std::mutex g_prof_mu;
// read by the launching threads (caller + enqueue thread) without the mutex
std::atomic<unsigned> g_prof_mask{0};
std::atomic<unsigned> g_prof_stride{1};   // time every n-th interval of a slot
std::atomic<uint64_t> g_prof_seq[kProfSlots];
std::vector<hipEvent_t> g_prof_free;      // recycled events (creating one costs microseconds)
hipEvent_t take_event() {
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
std::vector<ProfileRecord> g_prof_pending;
double g_prof_ms[kProfSlots] = {0};
uint64_t g_prof_launches[kProfSlots] = {0};

void drain_profile_locked() {
  for (ProfileRecord& r : g_prof_pending) {
    float ms = 0;
    if (hipEventSynchronize(r.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      g_prof_ms[r.slot] += ms;
      g_prof_launches[r.slot]++;
    }
    g_prof_free.push_back(r.start);
    g_prof_free.push_back(r.stop);
  }
  g_prof_pending.clear();
}
}  // namespace

void set_last_error(const std::string& msg) { g_last_error = msg; }
bool profile_enabled() { return g_prof_mask.load(std::memory_order_relaxed) != 0; }

ProfileScope::ProfileScope(int slot_, hipStream_t stream_) : slot(slot_), stream(stream_) {
  if (slot < 0) return;   // (the caller times the launch itself)
  if (!(g_prof_mask.load(std::memory_order_relaxed) & (1u << slot))) return;
  if (g_prof_seq[slot].fetch_add(1, std::memory_order_relaxed) %
          g_prof_stride.load(std::memory_order_relaxed) != 0) return;
  start = take_event();
  if (start) (void)hipEventRecord(start, stream);
}

ProfileScope::~ProfileScope() {
  if (!start) return;
  hipEvent_t stop = take_event();
  if (!stop) { (void)hipEventDestroy(start); return; }
  (void)hipEventRecord(stop, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_pending.push_back({slot, start, stop});
}

bool profile_begin(int slot, hipEvent_t* start, hipEvent_t* stop) {
  if (!(g_prof_mask.load(std::memory_order_relaxed) & (1u << slot))) return false;
  if (g_prof_seq[slot].fetch_add(1, std::memory_order_relaxed) %
          g_prof_stride.load(std::memory_order_relaxed) != 0) return false;
  *start = take_event();
  *stop = take_event();
  if (*start && *stop) return true;
  if (*start) (void)hipEventDestroy(*start);
  if (*stop) (void)hipEventDestroy(*stop);
  return false;
}

void profile_end(int slot, hipEvent_t start, hipEvent_t stop) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_pending.push_back({slot, start, stop});
}

}  // namespace gf

namespace gf {
namespace {

// Enqueue worker: issuing the ~25 launches of one step costs more host time (~3 us per
// launch) than the kernels take on the GPU at batch 600, so asynchronous submissions hand
// the work to this single thread (one issuer: no runtime-lock convoy between threads) and
// return; the caller overlaps its own host work (Python, building the next batch) and
// later waits for the *enqueue* to have happened (stream order covers the execution).
class EnqueueWorker {
 public:
  using Job = std::function<void()>;
  // lane 0: feature fetches, lane 1: sampling.  Two issuers by default: the sampling launches
  // (side stream) and the fetch launches (caller's stream) of a pipelined step go to different
  // HIP queues, and one thread issuing all 8 is the step's bottleneck whenever the host is
  // busy; GNNFLOW_ENQUEUE_LANES=1 puts both on one thread.
  static EnqueueWorker& get(int lane = 0) {
    static const bool two = [] {
      const char* v = std::getenv("GNNFLOW_ENQUEUE_LANES");
      return !(v && std::atoi(v) == 1);
    }();
    static EnqueueWorker* w0 = new EnqueueWorker();   // intentionally leaked: no exit-order issues
    if (lane == 0 || !two) return *w0;
    static EnqueueWorker* w1 = new EnqueueWorker();
    if (lane != 2) return *w1;
    // lane 2: a second sampling issuer (gf_sampler_set_enqueue_lane) — a sample's four launches +
    // event cost 19 us of issuing time, which ONE thread serving both lanes of a sampling-only
    // loop spends per step: that loop runs at the issuer's pace, not at the GPU's
    static EnqueueWorker* w2 = new EnqueueWorker();
    return *w2;
  }
  uint64_t submit(Job&& job) {
    bool wake;
    uint64_t ticket;
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back(std::move(job));
      ticket = ++submitted_;
      wake = sleeping_;
    }
    pending_.fetch_add(1, std::memory_order_release);
    if (wake) cv_job_.notify_one();   // a futex wake costs microseconds: only when needed
    return ticket;
  }
  // status of the submission `ticket` — its own, not an earlier job's — once it has been
  // enqueued
  int wait(uint64_t ticket, std::string* err) {
    // the enqueue usually finishes within microseconds: poll before sleeping on the condvar
    for (int i = 0; i < 20000 && done_.load(std::memory_order_acquire) < ticket; ++i)
      __builtin_ia32_pause();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return completed_ >= ticket; });
    auto it = failed_.find(ticket);
    if (it == failed_.end()) return GF_OK;
    const int rc = it->second.first;
    *err = std::move(it->second.second);
    failed_.erase(it);
    return rc;
  }

 private:
  EnqueueWorker() { std::thread(&EnqueueWorker::run, this).detach(); }
  void run() {
    for (;;) {
      Job job;
      // In a running pipeline the next job arrives within tens of microseconds: poll for it
      // (bounded, ~100 us) before sleeping, so that the submitter does not pay a futex wake
      // and this thread does not pay the wake-up latency.
      // (GNNFLOW_ENQUEUE_SPIN_US=0 turns the polling off: one busy thread less per lane when
      // many ranks share few cores.)
      static const long spin_us = [] {
        const char* v = std::getenv("GNNFLOW_ENQUEUE_SPIN_US");
        return v ? std::atol(v) : 100L;
      }();
      if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; pending_.load(std::memory_order_acquire) == 0; ++i) {
          __builtin_ia32_pause();
          if ((i & 255) == 255 &&
              std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
      }
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (q_.empty()) {
          sleeping_ = true;
          cv_job_.wait(lk, [&] { return !q_.empty(); });
          sleeping_ = false;
        }
        job = std::move(q_.front());
        q_.pop_front();
      }
      pending_.fetch_sub(1, std::memory_order_relaxed);
      int rc = GF_OK;
      std::string msg;
      const auto t0 = std::chrono::steady_clock::now();
      try {
        job();
      } catch (const Error& e) {
        rc = e.code; msg = e.what();
      } catch (const std::exception& e) {
        rc = GF_ERR_INVALID_ARGUMENT; msg = e.what();
      }
      const auto t1 = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(mu_);
      busy_us_ += std::chrono::duration<double, std::micro>(t1 - t0).count();
      ++completed_;
      done_.store(completed_, std::memory_order_release);
      if (rc != GF_OK) {
        failed_[completed_] = std::make_pair(rc, msg);   // jobs complete in ticket order
        while (failed_.size() > 64) failed_.erase(failed_.begin());   // never waited for
      }
      cv_done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  std::deque<Job> q_;
  uint64_t submitted_ = 0, completed_ = 0;
  bool sleeping_ = false;                 // worker is (about to be) blocked on cv_job_
  std::atomic<uint64_t> pending_{0};      // jobs queued and not yet taken
  std::atomic<uint64_t> done_{0};         // == completed_, readable without the mutex
  std::map<uint64_t, std::pair<int, std::string>> failed_;   // ticket -> status of that job

 public:
  double busy_us_ = 0;   // time spent issuing work (diagnostics)
  void stats(double* busy_us, uint64_t* jobs) {
    std::unique_lock<std::mutex> lk(mu_);
    *busy_us = busy_us_;
    *jobs = completed_;
  }
};

}  // namespace

// The enqueue thread that issues EVERYTHING with a collective in it — the partitioned sampler's
// chains and the pull rounds of sharded features: one thread, one order of collectives over all
// communicators, the same on every rank.  0 = the fetch lane's thread (default: two issuing
// threads slow each other down), 1 = the sampling lane's (GNNFLOW_PART_OWN_THREAD=1).
int collective_lane() {
  static const int lane = [] {
    const char* v = std::getenv("GNNFLOW_PART_OWN_THREAD");
    return (v && std::atoi(v) != 0) ? 1 : 0;
  }();
  return lane;
}
}  // namespace gf

using gf::guarded;

namespace gf {
// sampler.hip
void part_host_us(double out[8], bool reset);
uint64_t merge_recounts();
void philox_on_device(const uint64_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream);
// feature_cache.hip
uint64_t lru_recounts();
// partition.hip
size_t partition_scratch_bytes(size_t R, int world_size);
void partition_plan(const int64_t* d_nodes, const float* d_ts, size_t R, int world_size, int rank,
                    int64_t* d_requests, uint32_t* d_pos, uint64_t* d_counts, void* d_scratch,
                    size_t scratch_bytes, int device, hipStream_t stream);
// block_ops.hip
void segment_offsets(const int64_t* d_row, size_t num_edges, size_t num_dst, int64_t* d_offsets,
                     int device, hipStream_t stream);
void edge_softmax(const int64_t* d_offsets, size_t num_dst, size_t num_edges, size_t heads,
                  const float* d_y_or_x, const float* d_grad_y, float* d_out, int device,
                  hipStream_t stream);
void segment_reduce_forward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                            const float* d_src, size_t dim, const float* d_w, size_t heads,
                            bool mean, float* d_out, int device, hipStream_t stream);
void segment_reduce_backward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                             const float* d_src, size_t dim, const float* d_w, size_t heads,
                             bool mean, const float* d_grad_out, float* d_grad_src,
                             size_t num_src, float* d_grad_w, int device, hipStream_t stream);
void segment_max_forward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                         const float* d_src, size_t dim, float* d_out, int64_t* d_arg, int device,
                         hipStream_t stream);
void segment_max_backward(size_t num_dst, const int64_t* d_col, size_t dim,
                          const float* d_grad_out, const int64_t* d_arg, float* d_grad_src,
                          size_t num_src, int device, hipStream_t stream);
}  // namespace gf

extern "C" {

const char* gf_last_error(void) { return gf::g_last_error.c_str(); }
const char* gf_version(void) { return "gnnflow_amd 0.1 (gfx950)"; }

// ---- graph -------------------------------------------------------------------------
int gf_graph_create(gf_graph** out, size_t initial_pool_size, size_t maximum_pool_size,
                    int mem_resource_type, size_t minium_block_size,
                    size_t blocks_to_preallocate, int insertion_policy, int device,
                    int adaptive_block_size) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_graph_create: null out");
    *out = new gf_graph(initial_pool_size, maximum_pool_size, mem_resource_type,
                        minium_block_size, blocks_to_preallocate, insertion_policy, device,
                        adaptive_block_size != 0);
  });
}
int gf_graph_destroy(gf_graph* g) {
  if (gf::foreign_process()) return GF_OK;
  return guarded([&] { delete g; });
}
#define GF_G(g) GF_REQUIRE((g) != nullptr, "null graph handle")

int gf_graph_add_edges(gf_graph* g, const int64_t* src, const int64_t* dst, const float* ts,
                       const int64_t* eids, size_t n) {
  return guarded([&] { GF_G(g); g->impl.add_edges(src, dst, ts, eids, n); });
}
int gf_graph_offload_old_blocks(gf_graph* g, float timestamp, int to_file, size_t* num_blocks) {
  return guarded([&] {
    GF_G(g);
    size_t n = g->impl.offload_old_blocks(timestamp, to_file != 0);
    if (num_blocks) *num_blocks = n;
  });
}
int gf_graph_num_vertices(const gf_graph* g, size_t* out) {
  return guarded([&] { GF_G(g); *out = g->impl.num_nodes(); });
}
int gf_graph_num_source_vertices(const gf_graph* g, size_t* out) {
  return guarded([&] { GF_G(g); *out = g->impl.num_src_nodes(); });
}
int gf_graph_num_edges(const gf_graph* g, size_t* out) {
  return guarded([&] { GF_G(g); *out = g->impl.num_edges(); });
}
int gf_graph_max_vertex_id(const gf_graph* g, int64_t* out) {
  return guarded([&] { GF_G(g); *out = g->impl.max_node_id(); });
}
int gf_graph_ids_fit_u32(const gf_graph* g, int* out) {
  return guarded([&] { GF_G(g); *out = g->impl.ids_fit_u32() ? 1 : 0; });
}
int gf_graph_out_degree(const gf_graph* g, const int64_t* nodes, size_t n, size_t* out) {
  return guarded([&] { GF_G(g); g->impl.out_degree(nodes, n, out); });
}
int gf_graph_nodes(const gf_graph* g, int64_t* out, size_t capacity, size_t* count) {
  return guarded([&] { GF_G(g); *count = g->impl.nodes(out, capacity, false); });
}
int gf_graph_src_nodes(const gf_graph* g, int64_t* out, size_t capacity, size_t* count) {
  return guarded([&] { GF_G(g); *count = g->impl.nodes(out, capacity, true); });
}
int gf_graph_edges(const gf_graph* g, int64_t* out, size_t capacity, size_t* count) {
  return guarded([&] { GF_G(g); *count = g->impl.edges(out, capacity); });
}
int gf_graph_get_temporal_neighbors(const gf_graph* g, int64_t node, int64_t* dst, float* ts,
                                    int64_t* eids, size_t capacity, size_t* count) {
  return guarded([&] {
    GF_G(g);
    *count = g->impl.get_temporal_neighbors(node, dst, ts, eids, capacity);
  });
}
int gf_graph_avg_linked_list_length(const gf_graph* g, float* out) {
  return guarded([&] { GF_G(g); *out = g->impl.avg_linked_list_length(); });
}
int gf_graph_memory_usage(const gf_graph* g, float* out) {
  return guarded([&] { GF_G(g); *out = g->impl.graph_mem_usage(); });
}
int gf_graph_metadata_memory_usage(const gf_graph* g, float* out) {
  return guarded([&] { GF_G(g); *out = g->impl.metadata_mem_usage(); });
}
int gf_graph_device(const gf_graph* g, int* out) {
  return guarded([&] { GF_G(g); *out = g->impl.device(); });
}

// ---- sampler -----------------------------------------------------------------------
#define GF_S(s) GF_REQUIRE((s) != nullptr, "null sampler handle")

int gf_sampler_create(gf_sampler** out, gf_graph* g, const uint32_t* fanouts, size_t num_layers,
                      int sampling_policy, uint32_t num_snapshots, float snapshot_time_window,
                      int prop_time, uint64_t seed) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_sampler_create: null out");
    GF_G(g);
    GF_REQUIRE(fanouts != nullptr, "gf_sampler_create: null fanouts");
    *out = new gf_sampler(&g->impl, fanouts, num_layers, sampling_policy, num_snapshots,
                          snapshot_time_window, prop_time != 0, seed);
  });
}
int gf_sampler_destroy(gf_sampler* s) {
  if (gf::foreign_process()) return GF_OK;
  return guarded([&] { delete s; });
}
int gf_sampler_output_bytes(const gf_sampler* s, size_t num_roots, size_t* bytes) {
  return guarded([&] { GF_S(s); *bytes = s->impl.output_bytes(num_roots); });
}
int gf_sampler_layer_output_bytes(const gf_sampler* s, size_t num_roots, uint32_t layer,
                                  size_t* bytes) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(layer < s->impl.num_layers(), "layer out of range");
    *bytes = s->impl.layer_output_bytes(num_roots, layer);
  });
}
int gf_sampler_sample(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                      size_t num_roots, void* d_out, size_t out_bytes, gf_block* blocks,
                      void* stream) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(s->begin_tickets.empty(), "sample: asynchronous samples are still in flight");
    s->impl.sample(d_roots, d_root_ts, num_roots, d_out, out_bytes, blocks,
                   static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_sample_begin(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                            size_t num_roots, void* d_out, size_t out_bytes, void* stream) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
               "sample_begin: earlier samples were begun through the enqueue thread");
    s->impl.sample_begin(d_roots, d_root_ts, num_roots, d_out, out_bytes,
                         static_cast<hipStream_t>(stream));
    s->begin_tickets.push_back(0);
  });
}
int gf_sampler_sample_begin_async(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                                  size_t num_roots, void* d_out, size_t out_bytes,
                                  void* stream) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(s->begin_tickets.size() < gf::Sampler::kMaxInFlight,
               "sample_begin_async: too many samples in flight on this sampler");
    gf::Sampler* impl = &s->impl;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int lane = s->plain_lane;
    const uint64_t mark = lane == 2 ? (1ull << 62) : 0ull;
    s->begin_tickets.push_back(mark | gf::EnqueueWorker::get(lane).submit(
        [impl, d_roots, d_root_ts, num_roots, d_out, out_bytes, st]() {
          impl->sample_begin(d_roots, d_root_ts, num_roots, d_out, out_bytes, st);
        }));
  });
}
int gf_sampler_set_enqueue_lane(gf_sampler* s, int lane) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(lane == 1 || lane == 2, "gf_sampler_set_enqueue_lane: lane must be 1 or 2");
    GF_REQUIRE(s->begin_tickets.empty(), "gf_sampler_set_enqueue_lane: samples are in flight");
    s->plain_lane = lane;
  });
}
int gf_sampler_call_counter(const gf_sampler* s, uint64_t* out) {
  return guarded([&] {
    GF_S(s);
    GF_REQUIRE(out != nullptr, "gf_sampler_call_counter: null output");
    *out = s->impl.call_counter();
  });
}
int gf_sampler_set_call_counter(gf_sampler* s, uint64_t value, int through_enqueue_thread) {
  return guarded([&] {
    GF_S(s);
    gf::Sampler* impl = &s->impl;
    if (through_enqueue_thread) {
      // (jobs of the sampling lane run in submission order: the begin submitted next sees it)
      gf::EnqueueWorker::get(s->plain_lane).submit([impl, value]() { impl->set_call_counter(value); });
    } else {
      GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
                 "set_call_counter: samples begun through the enqueue thread are in flight");
      impl->set_call_counter(value);
    }
  });
}
int gf_sampler_sample_end(gf_sampler* s, gf_block* blocks) {
  if (s && !s->begin_tickets.empty()) {
    const uint64_t t = s->begin_tickets.front();
    s->begin_tickets.pop_front();
    if (t) {   // begun through the enqueue thread: wait for the enqueue of THIS sample
      std::string err;
      // bit 63: the job went to the fetch lane's thread (the chains of a communicator), bit 62:
      // to the second sampling issuer
      const int lane = (t >> 63) ? 0 : ((t >> 62) & 1) ? 2 : 1;
      const int rc = gf::EnqueueWorker::get(lane).wait(t & ~(3ull << 62), &err);
      if (rc != GF_OK) { gf::set_last_error(err); return rc; }
    }
  }
  return guarded([&] { GF_S(s); s->impl.sample_end(blocks); });
}
int gf_sampler_sample_layer(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                            size_t num_roots, uint32_t layer, uint32_t snapshot, void* d_out,
                            size_t out_bytes, gf_block* block, void* stream) {
  return guarded([&] {
    GF_S(s);
    s->impl.sample_layer(d_roots, d_root_ts, num_roots, layer, snapshot, d_out, out_bytes, block,
                         static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_sample_host(gf_sampler* s, const int64_t* nodes, const float* ts,
                           size_t num_roots, gf_block* blocks) {
  return guarded([&] { GF_S(s); s->impl.sample_host(nodes, ts, num_roots, blocks); });
}
int gf_sampler_sample_layer_host(gf_sampler* s, const int64_t* nodes, const float* ts,
                                 size_t num_roots, uint32_t layer, uint32_t snapshot,
                                 gf_block* block) {
  return guarded([&] {
    GF_S(s);
    s->impl.sample_layer_host(nodes, ts, num_roots, layer, snapshot, block);
  });
}
void gf_host_blocks_free(gf_block* blocks, size_t n) {
  if (!blocks) return;
  for (size_t i = 0; i < n; ++i) {
    free(blocks[i].all_nodes);
    free(blocks[i].all_timestamps);
    free(blocks[i].delta_timestamps);
    free(blocks[i].eids);
    free(blocks[i].row);
    free(blocks[i].col);
    blocks[i] = gf_block{};
  }
}

// ---- feature cache -----------------------------------------------------------------
#define GF_C(c) GF_REQUIRE((c) != nullptr, "null cache handle")

int gf_cache_create(gf_cache** out, size_t num_ids, size_t capacity, size_t dim,
                    const float* d_feats, int device) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_cache_create: null out");
    *out = new gf_cache(num_ids, capacity, dim, d_feats, device);
  });
}
int gf_cache_destroy(gf_cache* c) {
  if (gf::foreign_process()) return GF_OK;
  return guarded([&] { delete c; });
}
int gf_cache_set_policy(gf_cache* c, int policy) {
  return guarded([&] { GF_C(c); c->impl.set_policy(policy); });
}
int gf_cache_reset_order(gf_cache* c, void* stream) {
  return guarded([&] { GF_C(c); c->impl.reset_order(static_cast<hipStream_t>(stream)); });
}
int gf_cache_init_ids(gf_cache* c, const int64_t* d_ids, size_t n, void* stream) {
  return guarded([&] { GF_C(c); c->impl.init_ids(d_ids, n, static_cast<hipStream_t>(stream)); });
}
int gf_cache_init(gf_cache* c, void* stream) {
  return guarded([&] { GF_C(c); c->impl.init(static_cast<hipStream_t>(stream)); });
}
int gf_cache_resize(gf_cache* c, size_t new_num_ids, size_t new_capacity, const float* d_feats,
                    void* stream) {
  return guarded([&] {
    GF_C(c);
    c->impl.resize(new_num_ids, new_capacity, d_feats, static_cast<hipStream_t>(stream));
  });
}
int gf_cache_fetch(gf_cache* c, const int64_t* d_ids, size_t n, float* d_out, int update,
                   uint32_t* d_stats, void* stream) {
  return guarded([&] {
    GF_C(c);
    c->impl.fetch(d_ids, n, d_out, update != 0, d_stats, static_cast<hipStream_t>(stream));
  });
}
int gf_cache_init_rows(gf_cache* c, const int64_t* d_ids, size_t n, const float* d_rows,
                       void* stream) {
  return guarded([&] {
    GF_C(c);
    GF_REQUIRE(d_rows != nullptr || n == 0, "cache: null rows");
    c->impl.init_ids(d_ids, n, static_cast<hipStream_t>(stream), d_rows);
  });
}
int gf_cache_probe(gf_cache* c, const int64_t* d_ids, size_t n, int32_t* d_slot, void* stream) {
  return guarded([&] {
    GF_C(c);
    c->impl.probe(d_ids, n, d_slot, static_cast<hipStream_t>(stream));
  });
}
int gf_pull_count(const gf_pull_desc* descs, size_t n, int world_size, uint32_t* d_counts,
                  int device, void* stream) {
  return guarded([&] {
    GF_REQUIRE(descs != nullptr && n >= 1 && n <= 4, "gf_pull_count: 1..4 contexts");
    gf::FeatureCache* caches[4] = {nullptr, nullptr, nullptr, nullptr};
    for (size_t i = 0; i < n; ++i) caches[i] = descs[i].cache ? &descs[i].cache->impl : nullptr;
    gf::pull_count(descs, n, world_size, caches, d_counts, device, static_cast<hipStream_t>(stream));
  });
}
int gf_pull_scatter(const gf_pull_desc* descs, size_t n, int world_size, uint32_t* d_counts,
                    uint32_t* d_cursor, int device, void* stream) {
  return guarded([&] {
    GF_REQUIRE(descs != nullptr && n >= 1 && n <= 4, "gf_pull_scatter: 1..4 contexts");
    gf::FeatureCache* caches[4] = {nullptr, nullptr, nullptr, nullptr};
    for (size_t i = 0; i < n; ++i) caches[i] = descs[i].cache ? &descs[i].cache->impl : nullptr;
    gf::pull_scatter(descs, n, world_size, caches, d_counts, d_cursor, device,
                     static_cast<hipStream_t>(stream));
  });
}
int gf_gather_rows_indexed(const float* d_rows, size_t num_local_rows, size_t dim,
                           const int32_t* d_index, size_t num_ids, const int64_t* d_ids, size_t n,
                           float* d_out, uint32_t* d_flag, int device, void* stream) {
  return guarded([&] {
    gf::gather_rows_indexed(d_rows, num_local_rows, dim, d_index, num_ids, d_ids, n, d_out, d_flag,
                            device, static_cast<hipStream_t>(stream));
  });
}
int gf_cache_fetch_blocks_pulled(gf_cache* node_cache, gf_cache* edge_cache,
                                 const gf_fetch_pulled_desc* descs, size_t n, void* stream) {
  return guarded([&] {
    gf::fetch_blocks_pulled(node_cache ? &node_cache->impl : nullptr,
                            edge_cache ? &edge_cache->impl : nullptr, descs, n,
                            static_cast<hipStream_t>(stream));
  });
}
int gf_pull_session_create(gf_pull_session** out, gf_comm* comm, int device) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_pull_session_create: null output");
    *out = new gf_pull_session(comm ? &comm->impl : nullptr, device);
    (*out)->ordered = comm != nullptr && !comm->loopback && comm->ipc == nullptr;
  });
}
int gf_pull_session_destroy(gf_pull_session* s) {
  if (gf::foreign_process()) return GF_OK;
  return guarded([&] { delete s; });
}
int gf_pull_round(gf_pull_session* s, gf_cache* node_cache, gf_cache* edge_cache,
                  const gf_pull_ctx* ctxs, size_t n, int flag, int* any_flag, uint64_t* rows_pulled,
                  uint64_t* bytes_sent, uint32_t* d_error_flag, void* stream) {
  if (s != nullptr && s->ordered) {
    // The sampler's chains (collectives on the lanes' communicators) are issued by the fetch
    // lane's enqueue thread, this round's collectives (on the session's communicator) would be
    // issued by the caller's: two threads, no common order across ranks — RCCL kernels of
    // different communicators that share a hardware queue could then wait for each other.  The
    // round therefore takes its place in that thread's queue and the caller waits for it.
    gf::PullSession* impl = &s->impl;
    gf::FeatureCache* nc = node_cache ? &node_cache->impl : nullptr;
    gf::FeatureCache* ec = edge_cache ? &edge_cache->impl : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    gf::EnqueueWorker& w = gf::EnqueueWorker::get(gf::collective_lane());
    const uint64_t t = w.submit([=]() {
      impl->round(nc, ec, ctxs, n, flag, any_flag, rows_pulled, bytes_sent, d_error_flag, st);
    });
    std::string err;
    const int rc = w.wait(t, &err);
    if (rc != GF_OK) gf::set_last_error(err);
    return rc;
  }
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null pull session");
    s->impl.round(node_cache ? &node_cache->impl : nullptr, edge_cache ? &edge_cache->impl : nullptr,
                  ctxs, n, flag, any_flag, rows_pulled, bytes_sent, d_error_flag,
                  static_cast<hipStream_t>(stream));
  });
}
int gf_cache_fetch_pulled(gf_cache* c, const int64_t* d_ids, size_t n, float* d_out, int update,
                          uint32_t* d_stats, const float* d_miss_rows,
                          const uint32_t* d_miss_index, void* stream) {
  return guarded([&] {
    GF_C(c);
    c->impl.fetch_pulled(d_ids, n, d_out, update != 0, d_stats, d_miss_rows, d_miss_index,
                         static_cast<hipStream_t>(stream));
  });
}
int gf_cache_fetch_blocks(gf_cache* node_cache, gf_cache* edge_cache, const gf_fetch_desc* descs,
                          size_t n, void* stream) {
  return guarded([&] {
    gf::fetch_blocks(node_cache ? &node_cache->impl : nullptr,
                     edge_cache ? &edge_cache->impl : nullptr, descs, n,
                     static_cast<hipStream_t>(stream));
  });
}
int gf_cache_fetch_blocks_async(gf_cache* node_cache, gf_cache* edge_cache,
                                const gf_fetch_desc* descs, size_t n, void* stream,
                                uint64_t* ticket) {
  return guarded([&] {
    GF_REQUIRE(ticket != nullptr, "fetch_blocks_async: null ticket");
    GF_REQUIRE(descs != nullptr || n == 0, "fetch_blocks_async: null descriptors");
    gf::FeatureCache* node = node_cache ? &node_cache->impl : nullptr;
    gf::FeatureCache* edge = edge_cache ? &edge_cache->impl : nullptr;
    std::vector<gf_fetch_desc> copy(descs, descs + n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    *ticket = gf::EnqueueWorker::get().submit([node, edge, copy = std::move(copy), st]() {
      gf::fetch_blocks(node, edge, copy.data(), copy.size(), st);
    });
  });
}
int gf_cache_set_row_mirror(gf_cache* c, int on) {
  return guarded([&] { GF_C(c); c->impl.set_row_mirror(on != 0); });
}
int gf_debug_lru_trace_enable(gf_cache* c, int on) {
  return guarded([&] { GF_C(c); c->impl.lru_trace_enable(on != 0); });
}
int gf_debug_lru_trace(gf_cache* c, uint64_t* out, size_t capacity_words, size_t* words) {
  return guarded([&] {
    GF_C(c);
    GF_REQUIRE(out != nullptr && words != nullptr, "gf_debug_lru_trace: null output");
    *words = c->impl.lru_trace_read(out, capacity_words);
  });
}
int gf_cache_set_staging(gf_cache* c, size_t generations, size_t rows_per_generation) {
  return guarded([&] { GF_C(c); c->impl.set_staging(generations, rows_per_generation); });
}
int gf_cache_set_staging_lag(gf_cache* c, size_t lag) {
  return guarded([&] { GF_C(c); c->impl.set_staging_lag(lag); });
}
int gf_cache_invalidate_staging(gf_cache* c) {
  return guarded([&] { GF_C(c); c->impl.invalidate_staging(); });
}
int gf_cache_staging_state(gf_cache* c, uint64_t* out) {
  return guarded([&] {
    GF_C(c);
    GF_REQUIRE(out != nullptr, "gf_cache_staging_state: null output");
    c->impl.staging_state(out);
  });
}
int gf_cache_prefetch_blocks(gf_cache* node_cache, gf_cache* edge_cache,
                             const gf_fetch_desc* descs, size_t n, void* stream, int* issued) {
  return guarded([&] {
    const bool did = gf::prefetch_blocks(node_cache ? &node_cache->impl : nullptr,
                                         edge_cache ? &edge_cache->impl : nullptr, descs, n,
                                         static_cast<hipStream_t>(stream));
    if (issued) *issued = did ? 1 : 0;
  });
}
int gf_cache_prefetch_blocks_async(gf_cache* node_cache, gf_cache* edge_cache,
                                   const gf_fetch_desc* descs, size_t n, void* stream,
                                   uint64_t* ticket) {
  return guarded([&] {
    GF_REQUIRE(ticket != nullptr, "prefetch_blocks_async: null ticket");
    GF_REQUIRE(descs != nullptr || n == 0, "prefetch_blocks_async: null descriptors");
    gf::FeatureCache* node = node_cache ? &node_cache->impl : nullptr;
    gf::FeatureCache* edge = edge_cache ? &edge_cache->impl : nullptr;
    std::vector<gf_fetch_desc> copy(descs, descs + n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    *ticket = gf::EnqueueWorker::get().submit([node, edge, copy = std::move(copy), st]() {
      gf::prefetch_blocks(node, edge, copy.data(), copy.size(), st);
    });
  });
}
int gf_cache_fetch_announce_async(gf_cache* node_cache, gf_cache* edge_cache,
                                  const gf_fetch_desc* descs, size_t n, void* stream,
                                  const gf_fetch_desc* next_descs, size_t next_n,
                                  const gf_block* next_blocks, size_t next_layers,
                                  size_t next_snapshots, void* prefetch_stream, uint64_t* ticket) {
  return guarded([&] {
    GF_REQUIRE(ticket != nullptr, "fetch_announce_async: null ticket");
    GF_REQUIRE((descs != nullptr || n == 0) && (next_descs != nullptr || next_n == 0),
               "fetch_announce_async: null descriptors");
    gf::FeatureCache* node = node_cache ? &node_cache->impl : nullptr;
    gf::FeatureCache* edge = edge_cache ? &edge_cache->impl : nullptr;
    std::vector<gf_fetch_desc> copy(descs, descs + n), next(next_descs, next_descs + next_n);
    if (next_blocks != nullptr) {
      const size_t L = next_layers, NS = next_snapshots;
      GF_REQUIRE(L >= 1 && NS >= 1, "fetch_announce_async: empty block array");
      for (size_t s = 0; node && s < NS; ++s) {   // mfgs[0]: the last sampled layer
        const gf_block& b = next_blocks[(L - 1) * NS + s];
        if (b.num_src_nodes) next.push_back(gf_fetch_desc{0, 1, b.all_nodes, b.num_src_nodes, nullptr, nullptr});
      }
      for (size_t i = 0; edge && i < L * NS; ++i) {
        const gf_block& b = next_blocks[i];
        if (b.num_edges) next.push_back(gf_fetch_desc{1, 1, b.eids, b.num_edges, nullptr, nullptr});
      }
    }
    hipStream_t st = static_cast<hipStream_t>(stream), pst = static_cast<hipStream_t>(prefetch_stream);
    *ticket = gf::EnqueueWorker::get().submit(
        [node, edge, copy = std::move(copy), next = std::move(next), st, pst]() {
          gf::fetch_blocks(node, edge, copy.data(), copy.size(), st);
          gf::prefetch_blocks(node, edge, next.data(), next.size(), pst);
        });
  });
}
int gf_cache_fetch_wait(uint64_t ticket) {
  std::string err;
  const int rc = gf::EnqueueWorker::get().wait(ticket, &err);
  if (rc != GF_OK) gf::set_last_error(err);
  return rc;
}
int gf_memory_prepare_input(const float* d_node_memory, const float* d_node_memory_ts,
                            const float* d_mailbox, const float* d_mailbox_ts, size_t num_nodes,
                            size_t dim_memory, size_t dim_mail, const int64_t* d_ids, size_t n,
                            float* d_mem, float* d_mem_ts, float* d_mail_ts, float* d_mem_input,
                            int device, void* stream) {
  return guarded([&] {
    const float* tables[4] = {d_node_memory, d_mailbox, d_node_memory_ts, d_mailbox_ts};
    const size_t dims[4] = {dim_memory, dim_mail, 1, 1};
    float* outs[4] = {d_mem, d_mem_input, d_mem_ts, d_mail_ts};
    gf::gather_rows_multi(tables, dims, outs, 4, num_nodes, d_ids, n, device,
                          static_cast<hipStream_t>(stream));
  });
}
int gf_memory_update(float* d_node_memory, float* d_node_memory_ts, float* d_mailbox,
                     float* d_mailbox_ts, size_t num_nodes, size_t dim_memory, size_t dim_edge,
                     const int64_t* d_nid, const float* d_memory, const float* d_ts,
                     const float* d_edge_feats, size_t n, int neg_sample_ratio,
                     uint64_t* d_win_mail, uint64_t* d_win_mem, uint64_t epoch, int device,
                     void* stream) {
  return guarded([&] {
    gf::memory_update(d_node_memory, d_node_memory_ts, d_mailbox, d_mailbox_ts, num_nodes,
                      dim_memory, dim_edge, d_nid, d_memory, d_ts, d_edge_feats, n,
                      neg_sample_ratio, reinterpret_cast<unsigned long long*>(d_win_mail),
                      reinterpret_cast<unsigned long long*>(d_win_mem), epoch, device,
                      static_cast<hipStream_t>(stream));
  });
}
int gf_worker_stats(double* busy_us, uint64_t* jobs) {
  return guarded([&] {
    GF_REQUIRE(busy_us && jobs, "gf_worker_stats: null output");
    gf::EnqueueWorker::get(0).stats(busy_us, jobs);
    double b1 = 0;
    uint64_t j1 = 0;
    if (&gf::EnqueueWorker::get(1) != &gf::EnqueueWorker::get(0)) {
      gf::EnqueueWorker::get(1).stats(&b1, &j1);
      double b2 = 0;
      uint64_t j2 = 0;
      gf::EnqueueWorker::get(2).stats(&b2, &j2);
      if (std::getenv("GNNFLOW_WORKER_STATS"))
        std::fprintf(stderr, "[worker] lane0 %.0f us / %llu jobs, lane1 %.0f us / %llu jobs, lane2 %.0f us / %llu jobs\n",
                     *busy_us, (unsigned long long)*jobs, b1, (unsigned long long)j1, b2, (unsigned long long)j2);
      *busy_us += b1 + b2;
      *jobs += j1 + j2;
    }
  });
}
int gf_gather_rows(const float* d_feats, size_t num_rows, size_t dim, const int64_t* d_ids,
                   size_t n, float* d_out, int device, void* stream) {
  return guarded([&] {
    gf::gather_rows(d_feats, num_rows, dim, d_ids, n, d_out, device,
                    static_cast<hipStream_t>(stream));
  });
}
int gf_cache_slot_ids(const gf_cache* c, int64_t* out, size_t capacity) {
  return guarded([&] { GF_C(c); c->impl.slot_ids(out, capacity); });
}
int gf_cache_mem_bytes(const gf_cache* c, size_t* out) {
  return guarded([&] { GF_C(c); *out = c->impl.mem_bytes(); });
}
int gf_cache_lru_state(const gf_cache* c, uint64_t out[7]) {
  return guarded([&] {
    GF_C(c);
    GF_REQUIRE(out != nullptr, "gf_cache_lru_state: null output");
    c->impl.lru_state(out);
  });
}

// ---- profiling ---------------------------------------------------------------------
int gf_partition_scratch_bytes(size_t num_roots, int world_size, size_t* out) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_partition_scratch_bytes: null output");
    GF_REQUIRE(world_size >= 1, "partition: world size must be >= 1");
    *out = gf::partition_scratch_bytes(num_roots, world_size);
  });
}
int gf_partition_plan(const int64_t* d_nodes, const float* d_ts, size_t num_roots, int world_size,
                      int rank, int64_t* d_requests, uint32_t* d_pos, uint64_t* d_counts,
                      void* d_scratch, size_t scratch_bytes, int device, void* stream) {
  return guarded([&] {
    gf::partition_plan(d_nodes, d_ts, num_roots, world_size, rank, d_requests, d_pos, d_counts,
                       d_scratch, scratch_bytes, device, static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_sample_layer_padded(gf_sampler* s, const int64_t* d_requests, size_t n,
                                   uint32_t layer, uint32_t snapshot, int64_t* d_out,
                                   void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.sample_layer_padded(d_requests, n, layer, snapshot, d_out,
                                static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_merge_padded(gf_sampler* s, const int64_t* d_roots, const float* d_ts, size_t n,
                            uint32_t layer, const int64_t* d_replies, const uint32_t* d_pos,
                            void* d_out, size_t out_bytes, gf_block* block, void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.merge_padded(d_roots, d_ts, n, layer, d_replies, d_pos, d_out, out_bytes, block,
                         static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_part_layout(const gf_sampler* s, size_t num_roots, uint32_t layer, int world_size,
                           gf_part_layout* out) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(world_size >= 1 && world_size <= 64, "partition: world size must be 1..64");
    s->impl.part_layout(std::max<size_t>(num_roots, 1), layer, world_size, 0.0, 0, out);
  });
}
int gf_sampler_part_layout_slotted(const gf_sampler* s, size_t num_roots, uint32_t layer,
                                   int world_size, double slack, size_t slot_roots,
                                   gf_part_layout* out) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(world_size >= 1 && world_size <= 64, "partition: world size must be 1..64");
    GF_REQUIRE(slack > 0.0, "part_layout_slotted: slack must be positive");
    s->impl.part_layout(std::max<size_t>(num_roots, 1), layer, world_size, slack, slot_roots, out);
  });
}
int gf_sampler_part_group_slot(const gf_sampler* s, size_t num_roots, uint32_t layer,
                               int world_size, double slack, size_t slot_roots, int narrow,
                               double edge_fill, uint64_t* out) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(out != nullptr, "part_group_slot: null output");
    GF_REQUIRE(world_size >= 1 && world_size <= 64, "partition: world size must be 1..64");
    GF_REQUIRE(slack > 0.0, "part_group_slot: slack must be positive");
    GF_REQUIRE(layer < s->impl.num_layers(), "layer out of range");
    gf::Sampler::GroupLayout lay;
    const size_t R[1] = {std::max<size_t>(num_roots, 1)};
    s->impl.group_layout(R, 1, layer, world_size, slack, slot_roots, (narrow & 1) != 0, edge_fill,
                         &lay, (narrow & 2) != 0);
    const size_t rb = (narrow & 1) ? 12 : 24;
    out[0] = lay.stride;
    out[1] = edge_fill > 0.0 ? lay.cslot : lay.stride * s->impl.fanout(layer) * rb;
    out[2] = lay.edge_cap;
    out[3] = lay.off_bytes;
  });
}
int gf_sampler_part_begin(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                          size_t num_roots, void* d_out, size_t out_bytes, int world_size,
                          int rank, void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
               "part_begin: earlier samples were begun through the enqueue thread");
    s->impl.part_begin(d_roots, d_root_ts, num_roots, d_out, out_bytes, world_size, rank, 0.0, 0,
                       static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_part_begin_slotted(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                                  size_t num_roots, void* d_out, size_t out_bytes, int world_size,
                                  int rank, double slack, size_t slot_roots, void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(slack > 0.0, "part_begin_slotted: slack must be positive");
    GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
               "part_begin: earlier samples were begun through the enqueue thread");
    s->impl.part_begin(d_roots, d_root_ts, num_roots, d_out, out_bytes, world_size, rank, slack,
                       slot_roots, static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_part_serve(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                          size_t ws_bytes) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.part_serve(layer, snapshot, d_ws, ws_bytes);
  });
}
int gf_sampler_part_overflowed(const gf_sampler* s, int* out) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr && out != nullptr, "part_overflowed: null argument");
    *out = s->impl.last_overflow() ? 1 : 0;
  });
}
int gf_sampler_part_plan_own(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                             size_t ws_bytes, int phases) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(phases >= 1 && phases <= 3, "part_plan_own: phases must be 1, 2 or 3");
    s->impl.part_plan_own(layer, snapshot, d_ws, ws_bytes, phases);
  });
}
int gf_sampler_part_merge(gf_sampler* s, uint32_t layer, uint32_t snapshot, void* d_ws,
                          size_t ws_bytes) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.part_merge(layer, snapshot, d_ws, ws_bytes);
  });
}
int gf_sampler_part_commit(gf_sampler* s) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.part_commit();
    s->begin_tickets.push_back(0);
  });
}
int gf_sampler_part_abort(gf_sampler* s) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    s->impl.part_abort();
  });
}
int gf_sampler_sample_partitioned(gf_sampler* s, const int64_t* d_roots, const float* d_root_ts,
                                  size_t num_roots, void* d_out, size_t out_bytes, void* d_ws,
                                  size_t ws_bytes, void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
               "sample_partitioned: earlier samples were begun through the enqueue thread");
    s->impl.sample_partitioned(d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws, ws_bytes,
                               static_cast<hipStream_t>(stream));
    s->begin_tickets.push_back(0);
  });
}
int gf_sampler_sample_partitioned_async(gf_sampler* s, const int64_t* d_roots,
                                        const float* d_root_ts, size_t num_roots, void* d_out,
                                        size_t out_bytes, void* d_ws, size_t ws_bytes,
                                        void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr, "null sampler handle");
    GF_REQUIRE(s->begin_tickets.size() < gf::Sampler::kMaxInFlight,
               "sample_partitioned_async: too many samples in flight on this sampler");
    gf::Sampler* impl = &s->impl;
    hipStream_t st = static_cast<hipStream_t>(stream);
    s->begin_tickets.push_back(gf::EnqueueWorker::get(1).submit(
        [impl, d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws, ws_bytes, st]() {
          impl->sample_partitioned(d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws,
                                   ws_bytes, st);
        }));
  });
}
// ---- RCCL communicator (comm.hip) --------------------------------------------------------
int gf_comm_unique_id(uint8_t* out) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_comm_unique_id: null output");
    gf::RcclComm::unique_id(out);
  });
}
int gf_comm_create(gf_comm** out, const uint8_t* id, int world_size, int rank, int device) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr && id != nullptr, "gf_comm_create: null argument");
    *out = new gf_comm(id, world_size, rank, device);
  });
}
int gf_comm_destroy(gf_comm* c) {
  if (gf::foreign_process()) return GF_OK;
  return guarded([&] { delete c; });
}
int gf_ipc_comm_create(gf_comm** out, int world_size, int rank, int device, size_t mailbox_bytes,
                       const char* shm_name) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_ipc_comm_create: null output");
    *out = new gf_comm(new gf::IpcExchange(world_size, rank, device, mailbox_bytes, shm_name));
  });
}
int gf_ipc_comm_handle(gf_comm* c, uint8_t* out) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr && c->ipc != nullptr && out != nullptr, "not an IPC communicator");
    c->ipc->handle(out);
  });
}
int gf_ipc_comm_open(gf_comm* c, const uint8_t* handles) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr && c->ipc != nullptr, "not an IPC communicator");
    c->ipc->open_peers(handles);
  });
}
int gf_loopback_comm_create(gf_comm** out, int world_size, int device) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_loopback_comm_create: null output");
    auto ranks = gf::LoopbackExchange::create(world_size, device);
    for (int r = 0; r < world_size; ++r) out[r] = new gf_comm(ranks[r].release());
  });
}
int gf_comm_info(gf_comm* c, int32_t* out) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr && out != nullptr, "gf_comm_info: null argument");
    int v[4];
    c->impl.info(v);
    if (v[3] < 0) v[3] = c->ipc ? 1 : 2;
    for (int i = 0; i < 4; ++i) out[i] = v[i];
  });
}
int gf_comm_abort(gf_comm* c) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr, "null communicator");
    c->impl.abort();
  });
}
// `iters` equal-split all-to-alls of bytes_per_peer bytes per peer on scratch buffers, one after
// the other on `stream`: device time per exchange from events around the batch, host time per
// exchange of the issuing thread.  Collective: every rank calls it with the same arguments.
int gf_comm_time_all_to_all(gf_comm* c, size_t bytes_per_peer, int iters, void* stream,
                            double* device_us, double* host_us) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr && device_us != nullptr && host_us != nullptr && iters > 0,
               "gf_comm_time_all_to_all: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t bytes = std::max<size_t>(bytes_per_peer, 8) * static_cast<size_t>(c->impl.world());
    gf::DeviceBuffer send, recv;
    send.reserve(bytes);
    recv.reserve(bytes);
    GF_HIP(hipMemsetAsync(send.data(), 0, bytes, st));
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) c->impl.all_to_all(send.data(), recv.data(), bytes_per_peer, st);
    GF_HIP(hipStreamSynchronize(st));
    GF_HIP(hipEventRecord(e0, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) c->impl.all_to_all(send.data(), recv.data(), bytes_per_peer, st);
    const auto t1 = std::chrono::steady_clock::now();
    GF_HIP(hipEventRecord(e1, st));
    GF_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GF_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *device_us = 1e3 * ms / iters;
    *host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
  });
}
namespace {
__global__ void probe_spin_kernel(unsigned long long ticks) {   // 100 MHz wall clock
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void probe_touch_kernel(unsigned* out) { if (out) *out = 1u; }
}  // namespace
int gf_streams_share_queue(int device, void* a, void* b, unsigned spin_us, int* shared) {
  return guarded([&] {
    GF_REQUIRE(shared != nullptr, "gf_streams_share_queue: null output");
    GF_REQUIRE(spin_us >= 20 && spin_us <= 100000, "gf_streams_share_queue: spin_us out of range");
    gf::DeviceGuard dg(device);
    hipStream_t sa = static_cast<hipStream_t>(a), sb = static_cast<hipStream_t>(b);
    hipEvent_t ea = nullptr, eb = nullptr;
    GF_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    GF_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    GF_HIP(hipStreamSynchronize(sa));
    GF_HIP(hipStreamSynchronize(sb));
    int votes = 0;
    for (int round = 0; round < 3; ++round) {   // (a busy box may delay the small kernel once)
      probe_spin_kernel<<<dim3(1), dim3(64), 0, sa>>>(static_cast<unsigned long long>(spin_us) * 100ull);
      GF_HIP(hipEventRecord(ea, sa));
      probe_touch_kernel<<<dim3(1), dim3(1), 0, sb>>>(nullptr);
      GF_HIP(hipEventRecord(eb, sb));
      // b's kernel done while a's still spins -> the two run side by side
      bool beside = false;
      for (;;) {
        const hipError_t qb = hipEventQuery(eb);
        const hipError_t qa = hipEventQuery(ea);
        if (qb == hipSuccess && qa == hipErrorNotReady) { beside = true; break; }
        if (qa == hipSuccess) break;
        if (qa != hipErrorNotReady) GF_HIP(qa);
        if (qb != hipSuccess && qb != hipErrorNotReady) GF_HIP(qb);
      }
      (void)hipGetLastError();
      GF_HIP(hipStreamSynchronize(sa));
      GF_HIP(hipStreamSynchronize(sb));
      if (beside) ++votes;
    }
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    *shared = votes >= 2 ? 0 : 1;
  });
}
int gf_device_pci_bus_id(int device, char* out, size_t len) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr && len >= 16, "gf_device_pci_bus_id: output too small");
    GF_HIP(hipDeviceGetPCIBusId(out, static_cast<int>(len), device));
  });
}
int gf_comm_all_to_all(gf_comm* c, const void* d_send, void* d_recv, size_t bytes_per_peer,
                       void* stream) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr, "null communicator");
    c->impl.all_to_all(d_send, d_recv, bytes_per_peer, static_cast<hipStream_t>(stream));
  });
}
int gf_comm_all_to_all_v(gf_comm* c, const void* d_send, const size_t* send_bytes,
                         const size_t* send_offsets, void* d_recv, const size_t* recv_bytes,
                         const size_t* recv_offsets, void* stream) {
  return guarded([&] {
    GF_REQUIRE(c != nullptr, "null communicator");
    c->impl.all_to_all_v(d_send, send_bytes, send_offsets, d_recv, recv_bytes, recv_offsets,
                         static_cast<hipStream_t>(stream));
  });
}
int gf_sampler_sample_partitioned_comm(gf_sampler* s, gf_comm* c, const int64_t* d_roots,
                                       const float* d_root_ts, size_t num_roots, void* d_out,
                                       size_t out_bytes, void* d_ws, size_t ws_bytes, double slack,
                                       size_t slot_roots, int overlap, void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr && c != nullptr, "null sampler / communicator handle");
    GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
               "sample_partitioned_comm: earlier samples were begun through the enqueue thread");
    s->impl.sample_partitioned_slotted(d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws,
                                       ws_bytes, slack, slot_roots, c->impl, overlap != 0,
                                       static_cast<hipStream_t>(stream));
    s->begin_tickets.push_back(0);
  });
}
int gf_sampler_sample_partitioned_comm_async(gf_sampler* s, gf_comm* c, const int64_t* d_roots,
                                             const float* d_root_ts, size_t num_roots, void* d_out,
                                             size_t out_bytes, void* d_ws, size_t ws_bytes,
                                             double slack, size_t slot_roots, int overlap,
                                             void* stream) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr && c != nullptr, "null sampler / communicator handle");
    GF_REQUIRE(!c->loopback, "sample_partitioned_comm_async: a loopback communicator's ranks are "
                             "threads; one enqueue thread cannot serve them (use the synchronous call)");
    GF_REQUIRE(s->begin_tickets.size() < gf::Sampler::kMaxInFlight,
               "sample_partitioned_comm_async: too many samples in flight on this sampler");
    gf::Sampler* impl = &s->impl;
    gf::Exchange* comm = &c->impl;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ov = overlap != 0;
    // The chain of a sample over a communicator is ~11 stream operations, collectives among
    // them, and the pipelined loop is bound by the host time of issuing them.  Two issuing
    // threads slow each other down here (measured, one rank over RCCL, 3 lanes: the chain's
    // issue time 40 us with one thread for chains AND fetches, 63-83 us with a thread each;
    // step 53 vs 72-84 us), so the chains share the fetch lane's thread.  It also keeps ONE
    // global order of everything that is enqueued, on every rank.  GNNFLOW_PART_OWN_THREAD=1:
    // the sampling lane's own thread.
    const int lane = gf::collective_lane();
    const uint64_t mark = lane == 0 ? (1ull << 63) : 0;
    s->begin_tickets.push_back(mark | gf::EnqueueWorker::get(lane).submit(
        [impl, comm, d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws, ws_bytes, slack,
         slot_roots, ov, st]() {
          impl->sample_partitioned_slotted(d_roots, d_root_ts, num_roots, d_out, out_bytes, d_ws,
                                           ws_bytes, slack, slot_roots, *comm, ov, st);
        }));
  });
}
// `narrow_ids` of the shared-chain entry points: bit 0 = 12-byte reply records; bits 8..23 = the
// compact reply slots' edge fill in 1/1000 (0: the fixed records travel)
inline bool flag_narrow(int f) { return (f & 1) != 0; }
inline double flag_edge_fill(int f) { return ((f >> 8) & 0xFFFF) / 1000.0; }
// bit 1: layer l + 1 does not request layer l's roots again (most-recent, equal fanouts)
inline bool flag_reuse(int f) { return (f & 2) != 0; }
int gf_sampler_part_group_ws_bytes(const gf_sampler* s, const size_t* roots, int m, int world_size,
                                   double slack, size_t slot_roots, int narrow_ids,
                                   size_t* bytes) {
  return guarded([&] {
    GF_REQUIRE(s != nullptr && roots != nullptr && bytes != nullptr,
               "part_group_ws_bytes: null argument");
    GF_REQUIRE(m >= 1 && m <= GF_PART_GROUP_MAX, "group: 1..4 samples");
    GF_REQUIRE(world_size >= 1 && world_size <= 32, "group: world size must be 1..32");
    size_t R[GF_PART_GROUP_MAX];
    for (int j = 0; j < m; ++j) R[j] = std::max<size_t>(roots[j], 1);
    *bytes = (slack > 0.0 && s->impl.group_ok(R, m))
                 ? gf::Sampler::group_ws_bytes(s->impl, R, m, world_size, slack, slot_roots,
                                               flag_narrow(narrow_ids), flag_edge_fill(narrow_ids),
                                               flag_reuse(narrow_ids))
                 : 0;
  });
}
namespace {
// the group's samples as the sampler takes them (checked)
std::vector<gf::Sampler::GroupSample> group_samples(gf_comm* c, const gf_group_sample* samples,
                                                    int m) {
  (void)c;   // null: one rank, nothing to exchange
  GF_REQUIRE(samples != nullptr, "null samples");
  GF_REQUIRE(m >= 1 && m <= GF_PART_GROUP_MAX, "group: 1..4 samples");
  std::vector<gf::Sampler::GroupSample> gs(m);
  for (int j = 0; j < m; ++j) {
    GF_REQUIRE(samples[j].sampler != nullptr, "null sampler handle");
    gs[j] = gf::Sampler::GroupSample{&samples[j].sampler->impl, samples[j].d_roots,
                                     samples[j].d_root_ts, samples[j].num_roots, samples[j].d_out,
                                     samples[j].out_bytes};
  }
  return gs;
}
}  // namespace
int gf_sampler_sample_partitioned_comm_group(gf_comm* c, const gf_group_sample* samples, int m,
                                             void* d_ws, size_t ws_bytes, double slack,
                                             size_t slot_roots, int force_overflow,
                                             int narrow_ids, void* stream) {
  return guarded([&] {
    const auto gs = group_samples(c, samples, m);
    for (int j = 0; j < m; ++j) {
      gf_sampler* s = samples[j].sampler;
      GF_REQUIRE(s->begin_tickets.empty() || s->begin_tickets.back() == 0,
                 "sample_partitioned_comm_group: earlier samples were begun through the enqueue thread");
    }
    gf::Sampler::sample_partitioned_group(gs.data(), m, d_ws, ws_bytes, slack, slot_roots,
                                          c ? &c->impl : nullptr, static_cast<hipStream_t>(stream),
                                          static_cast<unsigned>(force_overflow), flag_narrow(narrow_ids), flag_edge_fill(narrow_ids), flag_reuse(narrow_ids));
    for (int j = 0; j < m; ++j) samples[j].sampler->begin_tickets.push_back(0);
  });
}
int gf_sampler_sample_partitioned_comm_group_async(gf_comm* c, const gf_group_sample* samples,
                                                   int m, void* d_ws, size_t ws_bytes,
                                                   double slack, size_t slot_roots,
                                                   int force_overflow, int narrow_ids,
                                                   void* stream) {
  return guarded([&] {
    auto gs = group_samples(c, samples, m);
    GF_REQUIRE(!c || !c->loopback, "sample_partitioned_comm_group_async: a loopback "
                                   "communicator's ranks are threads (use the synchronous call)");
    for (int j = 0; j < m; ++j)
      GF_REQUIRE(samples[j].sampler->begin_tickets.size() < gf::Sampler::kMaxInFlight,
                 "sample_partitioned_comm_group_async: too many samples in flight on a sampler");
    gf::Exchange* comm = c ? &c->impl : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // no communicator, no collective: the sampling lane's thread, like the plain sampler's
    const int lane = c ? gf::collective_lane() : 1;
    const uint64_t mark = lane == 0 ? (1ull << 63) : 0;
    // ONE job for all samples of the group: every sampler's ticket is this job's
    const uint64_t t = mark | gf::EnqueueWorker::get(lane).submit(
        [gs = std::move(gs), m, d_ws, ws_bytes, slack, slot_roots, comm, st, force_overflow,
         narrow_ids]() {
          gf::Sampler::sample_partitioned_group(gs.data(), m, d_ws, ws_bytes, slack, slot_roots,
                                                comm, st, static_cast<unsigned>(force_overflow),
                                                flag_narrow(narrow_ids), flag_edge_fill(narrow_ids),
                                                flag_reuse(narrow_ids));
        });
    for (int j = 0; j < m; ++j) samples[j].sampler->begin_tickets.push_back(t);
  });
}
int gf_block_segment_offsets(const int64_t* d_row, size_t num_edges, size_t num_dst,
                             int64_t* d_offsets, int device, void* stream) {
  return guarded([&] {
    gf::segment_offsets(d_row, num_edges, num_dst, d_offsets, device,
                        static_cast<hipStream_t>(stream));
  });
}
int gf_block_edge_softmax(const int64_t* d_offsets, size_t num_dst, size_t num_edges, size_t heads,
                          const float* d_logits, float* d_out, int device, void* stream) {
  return guarded([&] {
    gf::edge_softmax(d_offsets, num_dst, num_edges, heads, d_logits, nullptr, d_out, device,
                     static_cast<hipStream_t>(stream));
  });
}
int gf_block_edge_softmax_backward(const int64_t* d_offsets, size_t num_dst, size_t num_edges,
                                   size_t heads, const float* d_out, const float* d_grad_out,
                                   float* d_grad_logits, int device, void* stream) {
  return guarded([&] {
    GF_REQUIRE(d_grad_out != nullptr || num_edges == 0, "edge_softmax backward: null gradient");
    gf::edge_softmax(d_offsets, num_dst, num_edges, heads, d_out, d_grad_out, d_grad_logits,
                     device, static_cast<hipStream_t>(stream));
  });
}
int gf_block_reduce(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                    const float* d_src, size_t dim, const float* d_edge_weight, size_t heads,
                    int mean, float* d_out, int device, void* stream) {
  return guarded([&] {
    gf::segment_reduce_forward(d_offsets, num_dst, d_col, d_src, dim, d_edge_weight, heads,
                               mean != 0, d_out, device, static_cast<hipStream_t>(stream));
  });
}
int gf_block_reduce_backward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                             const float* d_src, size_t dim, const float* d_edge_weight,
                             size_t heads, int mean, const float* d_grad_out, float* d_grad_src,
                             size_t num_src, float* d_grad_edge_weight, int device,
                             void* stream) {
  return guarded([&] {
    gf::segment_reduce_backward(d_offsets, num_dst, d_col, d_src, dim, d_edge_weight, heads,
                                mean != 0, d_grad_out, d_grad_src, num_src, d_grad_edge_weight,
                                device, static_cast<hipStream_t>(stream));
  });
}

int gf_block_reduce_max(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                        const float* d_src, size_t dim, float* d_out, int64_t* d_arg, int device,
                        void* stream) {
  return guarded([&] {
    gf::segment_max_forward(d_offsets, num_dst, d_col, d_src, dim, d_out, d_arg, device,
                            static_cast<hipStream_t>(stream));
  });
}
int gf_block_reduce_max_backward(size_t num_dst, const int64_t* d_col, size_t dim,
                                 const float* d_grad_out, const int64_t* d_arg, float* d_grad_src,
                                 size_t num_src, int device, void* stream) {
  return guarded([&] {
    gf::segment_max_backward(num_dst, d_col, dim, d_grad_out, d_arg, d_grad_src, num_src, device,
                             static_cast<hipStream_t>(stream));
  });
}

int gf_debug_philox(const uint64_t* d_in, size_t n, uint32_t* d_out, void* stream) {
  return guarded([&] {
    GF_REQUIRE(n == 0 || (d_in != nullptr && d_out != nullptr), "gf_debug_philox: null buffer");
    gf::philox_on_device(d_in, n, d_out, static_cast<hipStream_t>(stream));
  });
}
int gf_debug_part_reused_roots(uint64_t* out) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_debug_part_reused_roots: null output");
    *out = gf::part_reused_roots();
  });
}
int gf_debug_merge_recounts(uint64_t* out) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_debug_merge_recounts: null output");
    *out = gf::merge_recounts();
  });
}
int gf_debug_lru_recounts(uint64_t* out) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_debug_lru_recounts: null output");
    *out = gf::lru_recounts();
  });
}
int gf_debug_part_host_us(double* out, int reset) {
  return guarded([&] {
    GF_REQUIRE(out != nullptr, "gf_debug_part_host_us: null output");
    gf::part_host_us(out, reset != 0);
  });
}
int gf_profile_enable(int mask) {
  std::lock_guard<std::mutex> lk(gf::g_prof_mu);
  gf::g_prof_mask = static_cast<unsigned>(mask);
  return GF_OK;
}
int gf_profile_reset(void) {
  std::lock_guard<std::mutex> lk(gf::g_prof_mu);
  gf::drain_profile_locked();
  for (int i = 0; i < gf::kProfSlots; ++i) {
    gf::g_prof_ms[i] = 0;
    gf::g_prof_launches[i] = 0;
    gf::g_prof_seq[i] = 0;
  }
  return GF_OK;
}
int gf_profile_launches(int which, uint64_t* launches) {
  return guarded([&] {
    GF_REQUIRE(which >= 0 && which < gf::kProfSlots && launches, "gf_profile_launches: bad argument");
    std::lock_guard<std::mutex> lk(gf::g_prof_mu);
    *launches = gf::g_prof_seq[which];
  });
}
int gf_profile_set_stride(unsigned stride) {
  std::lock_guard<std::mutex> lk(gf::g_prof_mu);
  gf::g_prof_stride = stride ? stride : 1;
  return GF_OK;
}
int gf_profile_get(int which, double* total_ms, uint64_t* launches) {
  return guarded([&] {
    GF_REQUIRE(which >= 0 && which < gf::kProfSlots, "gf_profile_get: bad slot");
    std::lock_guard<std::mutex> lk(gf::g_prof_mu);
    gf::drain_profile_locked();
    if (total_ms) *total_ms = gf::g_prof_ms[which];
    if (launches) *launches = gf::g_prof_launches[which];
  });
}

}  // extern "C"
