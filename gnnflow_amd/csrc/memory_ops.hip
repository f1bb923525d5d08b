// TGN memory / mailbox gather + scatter on MI355X — counterpart of
// gnnflow/models/modules/memory.py (Memory.prepare_input :156-190, update_mem_mail
// :192-269), the gather on the same node ids that follows the feature gather of mfgs[0].
//
// prepare_input: the reference does torch.unique on the CPU, four table gathers and an
// inverse scatter; the result is simply table[ids] for each of the four tables, so it is
// ONE launch of the feature-gather kernel with four cache-free contexts (memory rows,
// mailbox rows, and the two timestamp columns as 1-float rows).
//
// update_mem_mail: "last writer wins" per distinct node id, for the mailbox over the
// interleaved [src0,dst0,src1,dst1,...] order and for the memory over the [src.., dst..]
// order.  The reference picks the winner with `new_empty(...).scatter_(0, inv, perm)`, which
// torch documents as nondeterministic for duplicate indices (its CPU kernel is a sequential
// loop, i.e. the LAST occurrence wins); here the last occurrence wins deterministically:
// kernel 1 does atomicMax(winner[nid], epoch<<32 | position), kernel 2 lets the winning
// position write the row.  The epoch tag makes re-initialising the winner tables
// unnecessary.
#include <cstring>

#include "common.hpp"
#include "feature_cache.hpp"

namespace gf {

namespace {

constexpr int kThreads = 256;

struct UpdateArgs {
  float* node_memory;       // [N, dm]
  float* node_memory_ts;    // [N]
  float* mailbox;           // [N, 2*dm + de]
  float* mailbox_ts;        // [N]
  uint64_t num_nodes;
  uint32_t dm, de;
  const int64_t* nid;       // [n]  src(B) ++ dst(B) ++ neg(...)
  const float* memory;      // [n, dm]
  const float* ts;          // [n]
  const float* edge_feats;  // [B, de] or null (zeros)
  uint32_t B;               // n / (2 + neg_sample_ratio)
  unsigned long long* win_mail;  // [N]
  unsigned long long* win_mem;   // [N]
  unsigned long long epoch;      // > 0, increases every call
};

// mail position j (interleaved): node = (j even ? src : dst)[j / 2]
__device__ inline int64_t mail_node(const UpdateArgs& a, uint32_t j) {
  const uint32_t i = j >> 1;
  return a.nid[(j & 1) ? a.B + i : i];
}

__global__ void memory_claim_kernel(UpdateArgs a) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * a.B) return;
  const unsigned long long tag = (a.epoch << 32) | j;
  const int64_t m = mail_node(a, j);
  if (m >= 0 && static_cast<uint64_t>(m) < a.num_nodes) atomicMax(&a.win_mail[m], tag);
  const int64_t v = a.nid[j];   // memory order: src.. then dst..
  if (v >= 0 && static_cast<uint64_t>(v) < a.num_nodes) atomicMax(&a.win_mem[v], tag);
}

// one wave per position j in [0, 2B): writes the mailbox row / memory row it won
__global__ __launch_bounds__(kThreads) void memory_write_kernel(UpdateArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * kThreads) >> 6;
  const uint32_t dmail = 2 * a.dm + a.de;
  for (uint32_t j = wave; j < 2 * a.B; j += nwaves) {
    const unsigned long long tag = (a.epoch << 32) | j;
    const int64_t m = mail_node(a, j);
    if (m >= 0 && static_cast<uint64_t>(m) < a.num_nodes && a.win_mail[m] == tag) {
      // src_mail = [mem_src, mem_dst, ef], dst_mail = [mem_dst, mem_src, ef]  (:216-219)
      const uint32_t i = j >> 1;
      const float* first = a.memory + static_cast<uint64_t>((j & 1) ? a.B + i : i) * a.dm;
      const float* second = a.memory + static_cast<uint64_t>((j & 1) ? i : a.B + i) * a.dm;
      float* row = a.mailbox + static_cast<uint64_t>(m) * dmail;
      for (uint32_t c = lane; c < a.dm; c += 64) {
        row[c] = first[c];
        row[a.dm + c] = second[c];
      }
      for (uint32_t c = lane; c < a.de; c += 64)
        row[2 * a.dm + c] = a.edge_feats ? a.edge_feats[static_cast<uint64_t>(i) * a.de + c] : 0.0f;
      // `mail_ts = last_updated_ts[:len(nid)]` (:222): indexed by the interleaved position
      if (lane == 0) a.mailbox_ts[m] = a.ts[j];
    }
    const int64_t v = a.nid[j];
    if (v >= 0 && static_cast<uint64_t>(v) < a.num_nodes && a.win_mem[v] == tag) {
      const float* src = a.memory + static_cast<uint64_t>(j) * a.dm;
      float* row = a.node_memory + static_cast<uint64_t>(v) * a.dm;
      for (uint32_t c = lane; c < a.dm; c += 64) row[c] = src[c];
      if (lane == 0) a.node_memory_ts[v] = a.ts[j];
    }
  }
}

}  // namespace

void memory_update(float* node_memory, float* node_memory_ts, float* mailbox, float* mailbox_ts,
                   size_t num_nodes, size_t dim_memory, size_t dim_edge, const int64_t* nid,
                   const float* memory, const float* ts, const float* edge_feats, size_t n,
                   int neg_sample_ratio, unsigned long long* win_mail,
                   unsigned long long* win_mem, unsigned long long epoch, int device,
                   hipStream_t stream) {
  GF_REQUIRE(neg_sample_ratio >= 0, "memory_update: negative neg_sample_ratio");
  const size_t chunks = 2 + static_cast<size_t>(neg_sample_ratio);
  const size_t B = n / chunks;
  if (B == 0) return;
  GF_REQUIRE(node_memory && node_memory_ts && mailbox && mailbox_ts && nid && memory && ts &&
                 win_mail && win_mem,
             "memory_update: null pointer");
  GF_REQUIRE(epoch > 0 && epoch < (1ull << 31), "memory_update: bad epoch");
  GF_REQUIRE(2 * B < (1ull << 32), "memory_update: batch too large");
  DeviceGuard dg(device);
  UpdateArgs a;
  a.node_memory = node_memory; a.node_memory_ts = node_memory_ts;
  a.mailbox = mailbox; a.mailbox_ts = mailbox_ts;
  a.num_nodes = num_nodes;
  a.dm = static_cast<uint32_t>(dim_memory); a.de = static_cast<uint32_t>(dim_edge);
  a.nid = nid; a.memory = memory; a.ts = ts; a.edge_feats = edge_feats;
  a.B = static_cast<uint32_t>(B);
  a.win_mail = win_mail; a.win_mem = win_mem; a.epoch = epoch;
  const unsigned g1 = static_cast<unsigned>((2 * B + kThreads - 1) / kThreads);
  memory_claim_kernel<<<dim3(g1), dim3(kThreads), 0, stream>>>(a);
  const unsigned g2 = static_cast<unsigned>(std::min<size_t>((2 * B + 3) / 4, 4096));
  memory_write_kernel<<<dim3(g2), dim3(kThreads), 0, stream>>>(a);
  GF_HIP(hipGetLastError());
}

}  // namespace gf
