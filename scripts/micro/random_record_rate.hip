// Micro-benchmark: what bounds config 3's emit kernel (sample_emit_kernel at batch 600 000: one
// random 32-byte edge record read + 40 bytes of coalesced stores per sampled edge, 0.42 of 8 TB/s
// on its 60 algorithmic bytes)?  Three kernels over a 6.4 GB pool of 32-byte records (the 200 M-edge
// graph's nbr_pool), E = 141 M edges as in the layer-1 emit:
//   reads   : every thread reads 4 random records per trip (all four in flight), nothing written
//   writes  : the emit's six output arrays written in slot order (40 B per slot), nothing read
//   both    : the emit's shape — record reads, then the stores
// and the reads again with 64-byte and 128-byte records read whole.
//   hipcc --offload-arch=gfx950 -O3 random_record_rate.hip -o random_record_rate && ./random_record_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Rec { long long dst, eid; float ts; int pad[3]; };   // 32 B, as gf::EdgePair
__device__ inline uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }

template <int BYTES, bool READ, bool WRITE>
__global__ __launch_bounds__(256) void emit_like(const char* pool, uint64_t nrec, uint64_t total, long long* a8, float* a4, float* b4,
                                                  long long* c8, long long* d8, long long* e8, uint32_t salt) {
  const uint64_t stride = uint64_t{gridDim.x} * blockDim.x;
  constexpr int K = 4;
  long long acc = 0;
  for (uint64_t t0 = uint64_t{blockIdx.x} * blockDim.x + threadIdx.x; t0 < total; t0 += K * stride) {
    Rec r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint64_t t = t0 + k * stride;
      if (READ && t < total) {
        const char* at = pool + (mix(t + salt) % nrec) * BYTES;
        r[k] = *reinterpret_cast<const Rec*>(at);
        // (records of 64 / 128 bytes are read WHOLE: every further 32 bytes folded into the first)
        for (int part = 1; part < BYTES / 32; ++part) {
          const Rec more = reinterpret_cast<const Rec*>(at)[part];
          r[k].dst += more.dst; r[k].eid ^= more.eid;
        }
      } else { r[k].dst = t; r[k].eid = t; r[k].ts = 1.f; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint64_t t = t0 + k * stride;
      if (t >= total) continue;
      if (WRITE) { a8[t] = r[k].dst; a4[t] = r[k].ts; b4[t] = 2.f - r[k].ts; c8[t] = r[k].eid; d8[t] = static_cast<long long>(t / 10); e8[t] = static_cast<long long>(t); }
      else acc += r[k].dst + r[k].eid;
    }
  }
  if (!WRITE && acc == 0x7fffffffffffffffll) a8[0] = acc;
}
int main() {
  const uint64_t E = 141000000ull, pool_bytes = 6400000000ull;
  char* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
  long long *a8, *c8, *d8, *e8; float *a4, *b4;
  CK(hipMalloc(&a8, E * 8)); CK(hipMalloc(&c8, E * 8)); CK(hipMalloc(&d8, E * 8)); CK(hipMalloc(&e8, E * 8));
  CK(hipMalloc(&a4, E * 4)); CK(hipMalloc(&b4, E * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 16;
  auto run = [&](const char* name, auto kernel, uint64_t nrec, double bytes_read, double bytes_written) -> int {
    kernel<<<grid, 256>>>(pool, nrec, E, a8, a4, b4, c8, d8, e8, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; ++it) kernel<<<grid, 256>>>(pool, nrec, E, a8, a4, b4, c8, d8, e8, 7 + it);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("%-34s %8.1f us  %6.2f G records/s  read %5.2f TB/s  written %5.2f TB/s  together %5.2f TB/s\n", name, ms * 1e3,
           E / (ms * 1e-3) / 1e9, bytes_read / (ms * 1e-3) / 1e12, bytes_written / (ms * 1e-3) / 1e12, (bytes_read + bytes_written) / (ms * 1e-3) / 1e12);
    return 0;
  };
  if (run("reads only, 32-byte records", emit_like<32, true, false>, pool_bytes / 32, E * 32.0, 0)) return 1;
  if (run("reads only, 64-byte records", emit_like<64, true, false>, pool_bytes / 64, E * 64.0, 0)) return 1;
  if (run("reads only, 128-byte records", emit_like<128, true, false>, pool_bytes / 128, E * 128.0, 0)) return 1;
  if (run("writes only (40 B per slot)", emit_like<32, false, true>, pool_bytes / 32, 0, E * 40.0)) return 1;
  if (run("reads (32 B) + writes: the emit", emit_like<32, true, true>, pool_bytes / 32, E * 32.0, E * 40.0)) return 1;
  return 0;
}
