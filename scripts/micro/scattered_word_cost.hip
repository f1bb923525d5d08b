// Micro-benchmark: what does ONE scattered 4-byte access per thread cost as a function of the
// size of the array it lands in (the LRU bookkeeping of a 38 M-slot cache reads and writes
// words of 150-765 MB arrays at random)?  n threads, one access each, dependent chain of
// `hops` arrays (index -> a[index] -> b[a[index]] ...), timed per launch with events.
//   hipcc --offload-arch=gfx950 -O3 scattered_word_cost.hip -o scattered_word_cost && ./scattered_word_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill(uint32_t* a, size_t words, uint32_t seed) {
  for (size_t i = blockIdx.x * size_t{blockDim.x} + threadIdx.x; i < words; i += size_t{gridDim.x} * blockDim.x) {
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    a[i] = static_cast<uint32_t>(x % words);
  }
}
// hops dependent reads, then one scattered write
__global__ void chase(const uint32_t* a, uint32_t* w, size_t words, uint32_t n, int hops, int do_write, uint32_t salt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t{i} + salt) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  uint32_t idx = static_cast<uint32_t>(x % words);
  for (int h = 0; h < hops; ++h) idx = a[idx];
  if (do_write) w[idx] = i; else if (idx == 0xFFFFFFFFu) w[0] = 1;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes_mb[] = {4, 32, 152, 765, 3000, 12000};
  const uint32_t ns[] = {20000, 200000};
  for (size_t mb : sizes_mb) {
    const size_t words = mb * 1000000 / 4;
    uint32_t *a, *w; CK(hipMalloc(&a, words * 4)); CK(hipMalloc(&w, words * 4));
    fill<<<2048, 256, 0, s>>>(a, words, 7); CK(hipMemsetAsync(w, 0, words * 4, s));
    for (uint32_t n : ns) for (int hops = 0; hops <= 3; ++hops) for (int wr = 0; wr <= 1; ++wr) {
      if (hops == 0 && !wr) continue;
      for (int it = 0; it < 3; ++it) chase<<<(n + 255) / 256, 256, 0, s>>>(a, w, words, n, hops, wr, it);
      CK(hipEventRecord(e0, s));
      const int iters = 20;
      for (int it = 0; it < iters; ++it) chase<<<(n + 255) / 256, 256, 0, s>>>(a, w, words, n, hops, wr, 100 + it);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("array %6zu MB  n %6u  reads %d  write %d : %7.2f us per launch\n", mb, n, hops, wr, ms * 1e3 / iters);
    }
    CK(hipFree(a)); CK(hipFree(w));
  }
  return 0;
}
