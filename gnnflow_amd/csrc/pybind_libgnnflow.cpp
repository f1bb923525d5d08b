// pybind11 module named `libgnnflow`: the reference's native module surface
// (gnnflow/csrc/api.cc:26-128 — enums, _DynamicGraph, SamplingResult, _TemporalSampler,
// same method names / keyword arguments / numpy return types) implemented over the C ABI
// of include/gnnflow_hip.h.  With this module on sys.path the reference's own Python
// wrappers (gnnflow/dynamic_graph.py, gnnflow/temporal_sampler.py) import and run
// unmodified.  Compiled with g++ (no HIP code here); links libgnnflow_hip.so.
//
// KVStore (api.cc:122-127, kvstore.h:13-40) is exported as the plain host map it is in the
// reference (key -> row tensor), so that the import chain gnnflow.cache -> gnnflow.distributed
// .kvstore (`from libgnnflow import KVStore`, kvstore.py:12) resolves against this module.  The
// RPC server / client around it stay out of scope; the single-node counterpart of
// the feature store is gf_pull_round.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gnnflow_hip.h"

namespace py = pybind11;

namespace {

void check(int rc) {
  if (rc == GF_OK) return;
  std::string msg = gf_last_error();
  if (rc == GF_ERR_INVALID_ARGUMENT || rc == GF_ERR_TIMESTAMP_ORDER) throw py::value_error(msg);
  if (rc == GF_ERR_OUT_OF_MEMORY) throw std::bad_alloc();
  throw std::runtime_error(msg);
}

template <typename T>
py::array_t<T> to_numpy(const T* data, size_t n) {
  py::array_t<T> a(n);
  if (n) std::memcpy(a.mutable_data(), data, n * sizeof(T));
  return a;
}

enum class InsertionPolicy { INSERT = GF_INSERTION_POLICY_INSERT, REPLACE = GF_INSERTION_POLICY_REPLACE };
enum class SamplingPolicy { RECENT = GF_SAMPLING_POLICY_RECENT, UNIFORM = GF_SAMPLING_POLICY_UNIFORM };
enum class MemoryResourceType { CUDA = GF_MEM_CUDA, UNIFIED = GF_MEM_UNIFIED, PINNED = GF_MEM_PINNED, SHARED = GF_MEM_SHARED };

struct Graph {
  gf_graph* h = nullptr;
  Graph(size_t initial_pool_size, size_t maximum_pool_size, MemoryResourceType mem,
        size_t minium_block_size, size_t blocks_to_preallocate, InsertionPolicy policy, int device,
        bool adaptive_block_size) {
    check(gf_graph_create(&h, initial_pool_size, maximum_pool_size, static_cast<int>(mem),
                          minium_block_size, blocks_to_preallocate, static_cast<int>(policy),
                          device, adaptive_block_size ? 1 : 0));
  }
  ~Graph() { if (h) gf_graph_destroy(h); }
  Graph(const Graph&) = delete;
  Graph& operator=(const Graph&) = delete;
};

// One (layer, snapshot) result, host arrays owned by the C library until freed.
struct Result {
  gf_block b{};
  Result() = default;
  ~Result() { gf_host_blocks_free(&b, 1); }
  Result(const Result&) = delete;
  Result& operator=(const Result&) = delete;
};

struct Sampler {
  gf_sampler* h = nullptr;
  std::shared_ptr<Graph> graph;  // keep the graph alive (temporal_sampler.h:62 holds a ref)
  size_t num_layers, num_snapshots;
  Sampler(std::shared_ptr<Graph> g, const std::vector<uint32_t>& fanouts, SamplingPolicy policy,
          uint32_t num_snapshots_, float window, bool prop_time, uint64_t seed)
      : graph(std::move(g)), num_layers(fanouts.size()), num_snapshots(num_snapshots_) {
    check(gf_sampler_create(&h, graph->h, fanouts.data(), fanouts.size(),
                            static_cast<int>(policy), num_snapshots_, window, prop_time ? 1 : 0,
                            seed));
  }
  ~Sampler() { if (h) gf_sampler_destroy(h); }
  Sampler(const Sampler&) = delete;
  Sampler& operator=(const Sampler&) = delete;
};

// kvstore.h:13-40 / kvstore.cc: key -> row.  The values are torch tensors held as Python
// objects (this module is compiled without the torch headers): `set` keeps values[i] — a view
// of the caller's tensor, like the reference's `store_[keys[i]] = values[i]` — and `get`
// returns the stored rows in key order; a key never set yields None (the reference returns an
// undefined at::Tensor, which its type caster turns into None) and, like `operator[]`, enters
// the map.
struct KVStore {
  using Key = unsigned int;
  std::unordered_map<Key, py::object> store;
  std::mutex mutex;

  void set(const std::vector<Key>& keys, const py::object& values) {
    py::tuple rows = values.attr("unbind")(0);
    if (rows.size() < keys.size())
      throw py::index_error("KVStore.set: fewer rows than keys");
    std::lock_guard<std::mutex> lock(mutex);
    for (size_t i = 0; i < keys.size(); ++i) store[keys[i]] = rows[i];
  }
  py::list get(const std::vector<Key>& keys) {
    py::list out;
    for (Key k : keys) {
      auto it = store.find(k);
      if (it == store.end()) it = store.emplace(k, py::none()).first;
      out.append(it->second);
    }
    return out;
  }
  void fill_zeros() {
    for (auto& kv : store)
      if (!kv.second.is_none()) kv.second.attr("fill_")(0);
  }
  // kvstore.h:28-32: only the map itself is counted; sizeof(at::Tensor) is one pointer
  size_t memory_usage() const { return (sizeof(Key) + sizeof(void*)) * store.size(); }
};

}  // namespace

PYBIND11_MODULE(libgnnflow, m) {
  m.doc() = "MI355X-native drop-in for GNNFlow's libgnnflow (temporal edge store + sampler)";

  py::enum_<InsertionPolicy>(m, "InsertionPolicy")
      .value("INSERT", InsertionPolicy::INSERT)
      .value("REPLACE", InsertionPolicy::REPLACE);
  py::enum_<SamplingPolicy>(m, "SamplingPolicy")
      .value("RECENT", SamplingPolicy::RECENT)
      .value("UNIFORM", SamplingPolicy::UNIFORM);
  py::enum_<MemoryResourceType>(m, "MemoryResourceType")
      .value("CUDA", MemoryResourceType::CUDA)
      .value("UNIFIED", MemoryResourceType::UNIFIED)
      .value("PINNED", MemoryResourceType::PINNED)
      .value("SHARED", MemoryResourceType::SHARED);

  py::class_<Graph, std::shared_ptr<Graph>>(m, "_DynamicGraph")
      .def(py::init<size_t, size_t, MemoryResourceType, size_t, size_t, InsertionPolicy, int, bool>(),
           py::arg("initial_pool_size"), py::arg("maximum_pool_size"),
           py::arg("mem_resource_type"), py::arg("minium_block_size"),
           py::arg("blocks_to_preallocate"), py::arg("insertion_policy"), py::arg("device"),
           py::arg("adaptive_block_size"))
      .def("add_edges",
           [](Graph& g, py::array_t<int64_t, py::array::c_style | py::array::forcecast> src,
              py::array_t<int64_t, py::array::c_style | py::array::forcecast> dst,
              py::array_t<float, py::array::c_style | py::array::forcecast> ts,
              py::array_t<int64_t, py::array::c_style | py::array::forcecast> eids) {
             const size_t n = src.size();
             if (static_cast<size_t>(dst.size()) != n || static_cast<size_t>(ts.size()) != n ||
                 static_cast<size_t>(eids.size()) != n)
               throw py::value_error("add_edges: arrays must have the same length");
             const int64_t* s = src.data(); const int64_t* d = dst.data();
             const float* t = ts.data(); const int64_t* e = eids.data();
             int rc;
             {
               py::gil_scoped_release release;  // api.cc:50
               rc = gf_graph_add_edges(g.h, s, d, t, e, n);
             }
             check(rc);
           },
           py::arg("source_vertices"), py::arg("target_vertices"), py::arg("timestamps"),
           py::arg("eids"))
      .def("offload_old_blocks",
           [](Graph& g, float timestamp, bool to_file) {
             size_t n = 0;
             check(gf_graph_offload_old_blocks(g.h, timestamp, to_file ? 1 : 0, &n));
             return n;
           },
           py::arg("timestamp"), py::arg("to_file") = false)
      .def("num_vertices", [](const Graph& g) { size_t n; check(gf_graph_num_vertices(g.h, &n)); return n; })
      .def("num_source_vertices", [](const Graph& g) { size_t n; check(gf_graph_num_source_vertices(g.h, &n)); return n; })
      .def("num_edges", [](const Graph& g) { size_t n; check(gf_graph_num_edges(g.h, &n)); return n; })
      .def("out_degree",
           [](const Graph& g, std::vector<int64_t> nodes) {
             std::vector<size_t> out(nodes.size());
             check(gf_graph_out_degree(g.h, nodes.data(), nodes.size(), out.data()));
             return to_numpy(out.data(), out.size());
           })
      .def("nodes",
           [](const Graph& g) {
             size_t n = 0;
             check(gf_graph_nodes(g.h, nullptr, 0, &n));
             std::vector<int64_t> v(n);
             check(gf_graph_nodes(g.h, v.data(), n, &n));
             return to_numpy(v.data(), n);
           })
      .def("src_nodes",
           [](const Graph& g) {
             size_t n = 0;
             check(gf_graph_src_nodes(g.h, nullptr, 0, &n));
             std::vector<int64_t> v(n);
             check(gf_graph_src_nodes(g.h, v.data(), n, &n));
             return to_numpy(v.data(), n);
           })
      .def("edges",
           [](const Graph& g) {
             size_t n = 0;
             check(gf_graph_edges(g.h, nullptr, 0, &n));
             std::vector<int64_t> v(n);
             check(gf_graph_edges(g.h, v.data(), n, &n));
             return to_numpy(v.data(), n);
           })
      .def("max_vertex_id", [](const Graph& g) { int64_t v; check(gf_graph_max_vertex_id(g.h, &v)); return v; })
      .def("get_temporal_neighbors",
           [](const Graph& g, int64_t node) {
             size_t n = 0;
             check(gf_graph_get_temporal_neighbors(g.h, node, nullptr, nullptr, nullptr, 0, &n));
             std::vector<int64_t> d(n), e(n);
             std::vector<float> t(n);
             if (n) check(gf_graph_get_temporal_neighbors(g.h, node, d.data(), t.data(), e.data(), n, &n));
             return py::make_tuple(to_numpy(d.data(), n), to_numpy(t.data(), n), to_numpy(e.data(), n));
           })
      .def("avg_linked_list_length", [](const Graph& g) { float v; check(gf_graph_avg_linked_list_length(g.h, &v)); return v; })
      .def("get_graph_memory_usage", [](const Graph& g) { float v; check(gf_graph_memory_usage(g.h, &v)); return v; })
      .def("get_metadata_memory_usage", [](const Graph& g) { float v; check(gf_graph_metadata_memory_usage(g.h, &v)); return v; });

  py::class_<Result, std::shared_ptr<Result>>(m, "SamplingResult")
      .def("row", [](const Result& r) { return to_numpy(r.b.row, r.b.num_edges); })
      .def("col", [](const Result& r) { return to_numpy(r.b.col, r.b.num_edges); })
      .def("all_nodes", [](const Result& r) { return to_numpy(r.b.all_nodes, r.b.num_src_nodes); })
      .def("all_timestamps", [](const Result& r) { return to_numpy(r.b.all_timestamps, r.b.num_src_nodes); })
      .def("delta_timestamps", [](const Result& r) { return to_numpy(r.b.delta_timestamps, r.b.num_edges); })
      .def("eids", [](const Result& r) { return to_numpy(r.b.eids, r.b.num_edges); })
      .def("num_src_nodes", [](const Result& r) { return r.b.num_src_nodes; })
      .def("num_dst_nodes", [](const Result& r) { return r.b.num_dst_nodes; });

  py::class_<Sampler>(m, "_TemporalSampler")
      .def(py::init<std::shared_ptr<Graph>, const std::vector<uint32_t>&, SamplingPolicy, uint32_t,
                    float, bool, uint64_t>(),
           py::arg("dgraph"), py::arg("fanouts"), py::arg("sampling_policy"),
           py::arg("num_snapshots"), py::arg("snapshot_time_window"), py::arg("prop_time"),
           py::arg("seed"))
      .def("sample",
           [](Sampler& s, py::array_t<int64_t, py::array::c_style | py::array::forcecast> nodes,
              py::array_t<float, py::array::c_style | py::array::forcecast> ts) {
             if (nodes.size() != ts.size())
               throw py::value_error("sample: nodes and timestamps differ in length");
             const size_t nb = s.num_layers * s.num_snapshots;
             std::vector<gf_block> blocks(nb);
             const int64_t* n = nodes.data(); const float* t = ts.data();
             const size_t R = nodes.size();
             int rc;
             {
               py::gil_scoped_release release;  // api.cc:118
               rc = gf_sampler_sample_host(s.h, n, t, R, blocks.data());
             }
             check(rc);
             std::vector<std::vector<std::shared_ptr<Result>>> out(s.num_layers);
             for (size_t l = 0; l < s.num_layers; ++l)
               for (size_t k = 0; k < s.num_snapshots; ++k) {
                 auto r = std::make_shared<Result>();
                 r->b = blocks[l * s.num_snapshots + k];
                 out[l].push_back(std::move(r));
               }
             return out;
           })
      .def("sample_layer",
           [](Sampler& s, py::array_t<int64_t, py::array::c_style | py::array::forcecast> nodes,
              py::array_t<float, py::array::c_style | py::array::forcecast> ts, uint32_t layer,
              uint32_t snapshot) {
             if (nodes.size() != ts.size())
               throw py::value_error("sample_layer: nodes and timestamps differ in length");
             auto r = std::make_shared<Result>();
             const int64_t* n = nodes.data(); const float* t = ts.data();
             const size_t R = nodes.size();
             int rc;
             {
               py::gil_scoped_release release;  // api.cc:120
               rc = gf_sampler_sample_layer_host(s.h, n, t, R, layer, snapshot, &r->b);
             }
             check(rc);
             return r;
           });

  py::class_<KVStore>(m, "KVStore")
      .def(py::init<>())
      .def("set", &KVStore::set)
      .def("get", &KVStore::get)
      .def("memory_usage", &KVStore::memory_usage)
      .def("fill_zeros", &KVStore::fill_zeros);
}
