// Message passing on a sampled block: the two DGL primitives the
// reference's attention layer calls on every MFG — dgl.ops.edge_softmax and
// update_all(copy_src, sum) (gnnflow/models/modules/layers.py:153-159) — plus the mean /
// u_mul_e variants dgl.nn.SAGEConv / GATConv are made of (models/graphsage.py:27-31,
// models/gat.py:28-46).  dgl itself (requirements.txt: dgl >= 0.7) is a third-party
// dependency that is not vendored in the reference; what is restated here is its published
// semantics: softmax over the edges that share a destination, and per-destination sums of
// source rows.
//
// A block's edges are grouped by destination (the sampler emits them root-major, `row`
// non-decreasing), so both are SEGMENT operations over `offsets[num_dst + 1]`:
// HBM-bound streaming with no atomics on the forward side.  fp32 throughout.
#include "common.hpp"

#include <cfloat>
#include <cstdint>

namespace gf {
namespace {

constexpr int kThreads = 256;

// offsets[d] = first edge whose destination index is >= d (lower bound on the sorted row[])
__global__ void segment_offsets_kernel(const int64_t* __restrict__ row, uint64_t num_edges,
                                       uint64_t num_dst, int64_t* __restrict__ offsets) {
  const uint64_t d = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (d > num_dst) return;
  uint64_t lo = 0, hi = num_edges;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (static_cast<uint64_t>(row[mid]) < d) lo = mid + 1; else hi = mid;
  }
  offsets[d] = static_cast<int64_t>(lo);
}

// ---- edge softmax --------------------------------------------------------------------
// Short segments (a sampled block has at most `fanout` edges per destination): one thread
// per (destination, head); adjacent threads read adjacent heads / adjacent segments.
__global__ void edge_softmax_fwd_thread(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                        uint32_t heads, const float* __restrict__ x,
                                        float* __restrict__ y) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= num_dst * heads) return;
  const uint64_t d = i / heads;
  const uint32_t h = static_cast<uint32_t>(i - d * heads);
  const int64_t b = offsets[d], e = offsets[d + 1];
  float m = -FLT_MAX;
  for (int64_t k = b; k < e; ++k) m = fmaxf(m, x[k * heads + h]);
  float s = 0.f;
  for (int64_t k = b; k < e; ++k) s += __expf(x[k * heads + h] - m);
  const float inv = 1.f / s;
  for (int64_t k = b; k < e; ++k) y[k * heads + h] = __expf(x[k * heads + h] - m) * inv;
}

// grad_x = y * (grad_y - sum_segment(grad_y * y))
__global__ void edge_softmax_bwd_thread(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                        uint32_t heads, const float* __restrict__ y,
                                        const float* __restrict__ gy, float* __restrict__ gx) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= num_dst * heads) return;
  const uint64_t d = i / heads;
  const uint32_t h = static_cast<uint32_t>(i - d * heads);
  const int64_t b = offsets[d], e = offsets[d + 1];
  float dot = 0.f;
  for (int64_t k = b; k < e; ++k) dot += gy[k * heads + h] * y[k * heads + h];
  for (int64_t k = b; k < e; ++k) gx[k * heads + h] = y[k * heads + h] * (gy[k * heads + h] - dot);
}

__device__ inline float wave_max(float v) {
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ inline float wave_sum(float v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Long segments (blocks that were not sampled with a small fanout): one wave per
// (destination, head), lanes stride over the segment.
__global__ void edge_softmax_fwd_wave(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                      uint32_t heads, const float* __restrict__ x,
                                      float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (w >= num_dst * heads) return;   // wave-uniform
  const uint64_t d = w / heads;
  const uint32_t h = static_cast<uint32_t>(w - d * heads);
  const int64_t b = offsets[d], e = offsets[d + 1];
  float m = -FLT_MAX;
  for (int64_t k = b + lane; k < e; k += 64) m = fmaxf(m, x[k * heads + h]);
  m = wave_max(m);
  float s = 0.f;
  for (int64_t k = b + lane; k < e; k += 64) s += __expf(x[k * heads + h] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int64_t k = b + lane; k < e; k += 64) y[k * heads + h] = __expf(x[k * heads + h] - m) * inv;
}

__global__ void edge_softmax_bwd_wave(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                      uint32_t heads, const float* __restrict__ y,
                                      const float* __restrict__ gy, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (w >= num_dst * heads) return;
  const uint64_t d = w / heads;
  const uint32_t h = static_cast<uint32_t>(w - d * heads);
  const int64_t b = offsets[d], e = offsets[d + 1];
  float dot = 0.f;
  for (int64_t k = b + lane; k < e; k += 64) dot += gy[k * heads + h] * y[k * heads + h];
  dot = wave_sum(dot);
  for (int64_t k = b + lane; k < e; k += 64)
    gx[k * heads + h] = y[k * heads + h] * (gy[k * heads + h] - dot);
}

// ---- per-destination reduction of source rows ------------------------------------------
// out[d, :] = sum (or mean) over the edges k of segment d of  w[k, head(c)] * src[col[k], :]
// (w == nullptr: plain copy_src).  col == nullptr means the sampler's layout, col[k] =
// num_dst + k (roots first, then one new source node per edge): the k-th edge reads row
// num_dst + k, a destination's sources are one contiguous run, and the backward pass needs
// neither atomics nor a full memset.  One wave per destination, lanes over the row; every
// load is a contiguous run of the source row.  `per_head` = dim / heads columns share one
// edge weight (GATConv's u_mul_e with [E, H, 1] weights).
__global__ void segment_reduce_fwd(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                   const int64_t* __restrict__ col,
                                   const float* __restrict__ src, uint32_t dim,
                                   const float* __restrict__ w, uint32_t heads, int mean,
                                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t d = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (d >= num_dst) return;
  const int64_t b = offsets[d], e = offsets[d + 1];
  const uint32_t per_head = w ? dim / heads : dim;
  const float scale = (mean && e > b) ? 1.f / static_cast<float>(e - b) : 1.f;
  for (uint32_t c = lane; c < dim; c += 64) {
    const uint32_t h = w ? c / per_head : 0u;
    float acc = 0.f;
    for (int64_t k = b; k < e; ++k) {
      const uint64_t s = col ? static_cast<uint64_t>(col[k]) : num_dst + static_cast<uint64_t>(k);
      const float v = src[s * dim + c];
      acc += w ? v * w[k * heads + h] : v;
    }
    out[d * dim + c] = acc * scale;
  }
}

// grad_src[col[k], :] += scale * w[k, h] * grad_out[d, :]   (atomic: a source row may feed
// several edges in a general block; in a sampled block every address is added to once)
// grad_w[k, h]        = scale * sum_c grad_out[d, c] * src[col[k], c]   over head h's columns
__global__ void segment_reduce_bwd(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                   const int64_t* __restrict__ col,
                                   const float* __restrict__ src, uint32_t dim,
                                   const float* __restrict__ w, uint32_t heads, int mean,
                                   const float* __restrict__ gout, float* __restrict__ gsrc,
                                   float* __restrict__ gw) {
  const int lane = threadIdx.x & 63;
  const uint64_t d = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (d >= num_dst) return;
  const int64_t b = offsets[d], e = offsets[d + 1];
  const uint32_t per_head = w ? dim / heads : dim;
  const float scale = (mean && e > b) ? 1.f / static_cast<float>(e - b) : 1.f;
  for (int64_t k = b; k < e; ++k) {
    const uint64_t s = col ? static_cast<uint64_t>(col[k]) : num_dst + static_cast<uint64_t>(k);
    if (gsrc) {
      for (uint32_t c = lane; c < dim; c += 64) {
        const float g = gout[d * dim + c] * scale;
        const float v = w ? g * w[k * heads + c / per_head] : g;
        if (col) atomicAdd(&gsrc[s * dim + c], v);
        else gsrc[s * dim + c] = v;          // every source row feeds exactly one edge
      }
    }
    if (gw) {
      for (uint32_t h = 0; h < heads; ++h) {
        float acc = 0.f;
        for (uint32_t c = h * per_head + lane; c < (h + 1) * per_head; c += 64)
          acc += gout[d * dim + c] * src[s * dim + c];
        acc = wave_sum(acc);
        if (lane == 0) gw[k * heads + h] = acc * scale;
      }
    }
  }
}

// out[d, c] = max over the edges k of d of src[col[k], c]  (0 for a destination without
// in-edges, as dgl's max reducer leaves it); arg[d, c] = the edge that won (-1: none), lowest
// edge index on ties — the backward pass routes grad_out[d, c] to that edge's source row.
__global__ void segment_max_fwd(const int64_t* __restrict__ offsets, uint64_t num_dst,
                                const int64_t* __restrict__ col, const float* __restrict__ src,
                                uint32_t dim, float* __restrict__ out, int64_t* __restrict__ arg) {
  const int lane = threadIdx.x & 63;
  const uint64_t d = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (d >= num_dst) return;
  const int64_t b = offsets[d], e = offsets[d + 1];
  for (uint32_t c = lane; c < dim; c += 64) {
    float best = 0.f;
    int64_t who = -1;
    for (int64_t k = b; k < e; ++k) {
      const uint64_t s = col ? static_cast<uint64_t>(col[k]) : num_dst + static_cast<uint64_t>(k);
      const float v = src[s * dim + c];
      if (who < 0 || v > best) { best = v; who = k; }
    }
    out[d * dim + c] = best;
    arg[d * dim + c] = who;
  }
}

__global__ void segment_max_bwd(uint64_t num_dst, const int64_t* __restrict__ col, uint32_t dim,
                                const float* __restrict__ gout, const int64_t* __restrict__ arg,
                                float* __restrict__ gsrc) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= num_dst * dim) return;
  const int64_t k = arg[i];
  if (k < 0) return;
  const uint64_t s = col ? static_cast<uint64_t>(col[k]) : num_dst + static_cast<uint64_t>(k);
  atomicAdd(&gsrc[s * dim + (i % dim)], gout[i]);
}

inline unsigned blocks_for(uint64_t threads) {
  return static_cast<unsigned>((threads + kThreads - 1) / kThreads);
}

}  // namespace

void segment_offsets(const int64_t* d_row, size_t num_edges, size_t num_dst, int64_t* d_offsets,
                     int device, hipStream_t stream) {
  GF_REQUIRE(d_offsets != nullptr, "segment_offsets: null output");
  GF_REQUIRE(d_row != nullptr || num_edges == 0, "segment_offsets: null row array");
  DeviceGuard dg(device);
  segment_offsets_kernel<<<dim3(blocks_for(num_dst + 1)), dim3(kThreads), 0, stream>>>(
      d_row, num_edges, num_dst, d_offsets);
  GF_HIP(hipGetLastError());
}

void edge_softmax(const int64_t* d_offsets, size_t num_dst, size_t num_edges, size_t heads,
                  const float* d_y_or_x, const float* d_grad_y, float* d_out, int device,
                  hipStream_t stream) {
  if (num_dst == 0 || num_edges == 0 || heads == 0) return;
  GF_REQUIRE(d_offsets && d_y_or_x && d_out, "edge_softmax: null pointer");
  GF_REQUIRE(heads < (1u << 16), "edge_softmax: too many heads");
  DeviceGuard dg(device);
  const uint64_t items = static_cast<uint64_t>(num_dst) * heads;
  const bool long_segments = num_edges > 32 * num_dst;
  const uint32_t H = static_cast<uint32_t>(heads);
  if (!d_grad_y) {
    if (long_segments)
      edge_softmax_fwd_wave<<<dim3(blocks_for(items * 64)), dim3(kThreads), 0, stream>>>(
          d_offsets, num_dst, H, d_y_or_x, d_out);
    else
      edge_softmax_fwd_thread<<<dim3(blocks_for(items)), dim3(kThreads), 0, stream>>>(
          d_offsets, num_dst, H, d_y_or_x, d_out);
  } else {
    if (long_segments)
      edge_softmax_bwd_wave<<<dim3(blocks_for(items * 64)), dim3(kThreads), 0, stream>>>(
          d_offsets, num_dst, H, d_y_or_x, d_grad_y, d_out);
    else
      edge_softmax_bwd_thread<<<dim3(blocks_for(items)), dim3(kThreads), 0, stream>>>(
          d_offsets, num_dst, H, d_y_or_x, d_grad_y, d_out);
  }
  GF_HIP(hipGetLastError());
}

void segment_reduce_forward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                            const float* d_src, size_t dim, const float* d_w, size_t heads,
                            bool mean, float* d_out, int device, hipStream_t stream) {
  if (num_dst == 0 || dim == 0) return;
  GF_REQUIRE(d_offsets && d_out, "segment_reduce: null pointer");
  GF_REQUIRE(!d_w || (heads > 0 && dim % heads == 0), "segment_reduce: dim must be a multiple of heads");
  DeviceGuard dg(device);
  segment_reduce_fwd<<<dim3(blocks_for(static_cast<uint64_t>(num_dst) * 64)), dim3(kThreads), 0,
                       stream>>>(d_offsets, num_dst, d_col, d_src, static_cast<uint32_t>(dim),
                                 d_w, static_cast<uint32_t>(heads ? heads : 1), mean ? 1 : 0,
                                 d_out);
  GF_HIP(hipGetLastError());
}

void segment_max_forward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                         const float* d_src, size_t dim, float* d_out, int64_t* d_arg, int device,
                         hipStream_t stream) {
  if (num_dst == 0 || dim == 0) return;
  GF_REQUIRE(d_offsets && d_src && d_out && d_arg, "segment_max: null pointer");
  DeviceGuard dg(device);
  segment_max_fwd<<<dim3(blocks_for(static_cast<uint64_t>(num_dst) * 64)), dim3(kThreads), 0,
                    stream>>>(d_offsets, num_dst, d_col, d_src, static_cast<uint32_t>(dim), d_out,
                              d_arg);
  GF_HIP(hipGetLastError());
}

void segment_max_backward(size_t num_dst, const int64_t* d_col, size_t dim,
                          const float* d_grad_out, const int64_t* d_arg, float* d_grad_src,
                          size_t num_src, int device, hipStream_t stream) {
  GF_REQUIRE(d_grad_src != nullptr || num_src == 0, "segment_max backward: null gradient");
  DeviceGuard dg(device);
  if (num_src && dim) GF_HIP(hipMemsetAsync(d_grad_src, 0, num_src * dim * sizeof(float), stream));
  if (num_dst == 0 || dim == 0) return;
  GF_REQUIRE(d_grad_out && d_arg, "segment_max backward: null pointer");
  segment_max_bwd<<<dim3(blocks_for(static_cast<uint64_t>(num_dst) * dim)), dim3(kThreads), 0,
                    stream>>>(num_dst, d_col, static_cast<uint32_t>(dim), d_grad_out, d_arg,
                              d_grad_src);
  GF_HIP(hipGetLastError());
}

void segment_reduce_backward(const int64_t* d_offsets, size_t num_dst, const int64_t* d_col,
                             const float* d_src, size_t dim, const float* d_w, size_t heads,
                             bool mean, const float* d_grad_out, float* d_grad_src,
                             size_t num_src, float* d_grad_w, int device, hipStream_t stream) {
  DeviceGuard dg(device);
  if (d_grad_src && num_src && dim) {
    // general blocks accumulate into zeros; the sampler layout writes every edge's row
    // exactly once, so only the rows of the destination nodes themselves are cleared
    const size_t rows = d_col ? num_src : std::min(num_dst, num_src);
    if (rows) GF_HIP(hipMemsetAsync(d_grad_src, 0, rows * dim * sizeof(float), stream));
  }
  if (num_dst == 0 || dim == 0) return;
  GF_REQUIRE(d_offsets && d_grad_out, "segment_reduce backward: null pointer");
  GF_REQUIRE(!d_grad_w || (d_w && d_src), "segment_reduce backward: weight gradient needs w and src");
  GF_REQUIRE(!d_w || (heads > 0 && dim % heads == 0), "segment_reduce: dim must be a multiple of heads");
  segment_reduce_bwd<<<dim3(blocks_for(static_cast<uint64_t>(num_dst) * 64)), dim3(kThreads), 0,
                       stream>>>(d_offsets, num_dst, d_col, d_src, static_cast<uint32_t>(dim),
                                 d_w, static_cast<uint32_t>(heads ? heads : 1), mean ? 1 : 0,
                                 d_grad_out, d_grad_src, d_grad_w);
  GF_HIP(hipGetLastError());
}

}  // namespace gf
