// Micro-benchmark: what a kernel's reads of HOST memory cost on this box, and what they do to a
// latency chain in HBM running beside them (the staging ring's pull kernel next to the LRU update).
//  1. one wave chasing pointers through a pinned host array: latency per dependent 64-byte read,
//     random over the whole array / inside one 4 KB page;
//  2. the same chase through an HBM array, alone, and while another stream keeps W workgroups
//     pulling random 688-byte rows out of the host array.
//   hipcc --offload-arch=gfx950 -O3 host_read_latency.hip -o host_read_latency && ./host_read_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t x) {
  x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x;
}
// lane 0 of one wave: `hops` dependent reads of a[idx] (a[] holds the next index)
__global__ void chase(const uint32_t* a, uint32_t* out, uint32_t start, int hops) {
  uint32_t idx = start;
  if (threadIdx.x == 0) {
    for (int h = 0; h < hops; ++h) idx = a[idx];
    out[0] = idx;
  }
}
// W workgroups x 4 waves, each wave pulls 8 random rows of `rowf4` float4s per trip, `trips` trips
__global__ void pull(const float4* src, float4* dst, size_t rows, uint32_t rowf4, int trips, uint32_t salt) {
  const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int t = 0; t < trips; ++t) {
    const uint32_t total = 8 * rowf4;
    float4 v[6]; bool ok[6];
    for (int k = 0; k < 6; ++k) {
      const uint32_t f = lane + 64 * k; ok[k] = f < total;
      const uint32_t r = ok[k] ? f / rowf4 : 0;
      const size_t row = mix((uint64_t{wave} * 8 + r) * 131 + t * 7919 + salt) % rows;
      if (ok[k]) v[k] = src[row * rowf4 + (f - r * rowf4)];
    }
    for (int k = 0; k < 6; ++k) if (ok[k]) dst[(size_t{wave} * 8) * rowf4 + lane + 64 * k] = v[k];
  }
}
int main(int argc, char** argv) {
  const size_t host_bytes = size_t{1} << 30;
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t* out; CK(hipMalloc(&out, 64));
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: hipHostMalloc; mode 1: 2 MB-aligned anonymous memory with MADV_HUGEPAGE, hipHostRegister
    uint32_t* h = nullptr;
    if (mode == 0) CK(hipHostMalloc(&h, host_bytes, hipHostMallocDefault));
    else {
      void* p = mmap(nullptr, host_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (p == MAP_FAILED) { printf("mmap failed\n"); return 1; }
      madvise(p, host_bytes, MADV_HUGEPAGE);
      h = static_cast<uint32_t*>(p);
      for (size_t i = 0; i < host_bytes / 4; i += 1024) h[i] = 0;
      CK(hipHostRegister(h, host_bytes, hipHostRegisterDefault));
    }
    const size_t words = host_bytes / 4;
    // next-index chains: random over everything (16-word stride so every hop is a new line) ...
    for (size_t i = 0; i < words; i += 16) h[i] = static_cast<uint32_t>((mix(i) % (words / 16)) * 16);
    uint32_t* hd = nullptr; CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0));
    const char* name = mode == 0 ? "hipHostMalloc" : "mmap+THP+hipHostRegister";
    for (int hops : {64, 256}) {
      chase<<<1, 64, 0, s0>>>(hd, out, 0, hops); CK(hipStreamSynchronize(s0));
      CK(hipEventRecord(e0, s0));
      chase<<<1, 64, 0, s0>>>(hd, out, 16, hops);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-26s host chase, random over 1 GB : %d hops %8.1f us = %6.2f us per hop\n", name, hops, ms * 1e3, ms * 1e3 / hops);
    }
    // ... and a chain that stays inside the first 4 KB page
    static uint32_t saved[1024];
    for (int i = 0; i < 1024; ++i) saved[i] = h[i];
    for (int i = 0; i < 1024; i += 16) h[i] = ((i / 16 + 37) % 64) * 16;
    chase<<<1, 64, 0, s0>>>(hd, out, 0, 256); CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(e0, s0));
    chase<<<1, 64, 0, s0>>>(hd, out, 16, 256);
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-26s host chase, inside one 4 KB page: 256 hops %8.1f us = %6.2f us per hop\n", name, ms * 1e3, ms * 1e3 / 256);
    for (int i = 0; i < 1024; ++i) h[i] = saved[i];
    // HBM chase alone and beside the pull
    const size_t dwords = size_t{256} << 20;   // 1 GB in HBM
    uint32_t* d; CK(hipMalloc(&d, dwords * 4));
    {
      uint32_t* tmp = static_cast<uint32_t*>(malloc(dwords * 4));
      for (size_t i = 0; i < dwords; i += 16) tmp[i] = static_cast<uint32_t>((mix(i + 5) % (dwords / 16)) * 16);
      CK(hipMemcpy(d, tmp, dwords * 4, hipMemcpyHostToDevice)); free(tmp);
    }
    float4* dst; CK(hipMalloc(&dst, size_t{64} << 20));
    const uint32_t rowf4 = 43; const size_t rows = host_bytes / (rowf4 * 16);
    for (int W : {0, 8, 64, 256}) {
      chase<<<1, 64, 0, s0>>>(d, out, 0, 256); CK(hipStreamSynchronize(s0));
      if (W) pull<<<W, 256, 0, s1>>>(reinterpret_cast<const float4*>(hd), dst, rows, rowf4, 400, 1);
      CK(hipEventRecord(e0, s0));
      chase<<<1, 64, 0, s0>>>(d, out, 16, 256);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipDeviceSynchronize());
      printf("%-26s HBM chase (1 GB array) beside a pull of %3d workgroups: %6.2f us per hop\n", name, W, ms * 1e3 / 256);
    }
    // ... and beside DMA copies of the same host memory (hipMemcpyAsync, 413 KB each, back to back)
    {
      chase<<<1, 64, 0, s0>>>(d, out, 0, 4096); CK(hipStreamSynchronize(s0));
      for (int i = 0; i < 64; ++i)
        CK(hipMemcpyAsync(dst, reinterpret_cast<const char*>(h) + static_cast<size_t>(i) * 4000000, 412800, hipMemcpyHostToDevice, s1));
      CK(hipEventRecord(e0, s0));
      chase<<<1, 64, 0, s0>>>(d, out, 16, 4096);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const auto t0 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      const double tail_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%-26s HBM chase (4096 hops) beside 64 DMA copies of 413 KB: %6.2f us per hop (chase %.0f us; the copies went on for another %.0f us)\n",
             name, ms * 1e3 / 4096, ms * 1e3, tail_us);
      chase<<<1, 64, 0, s0>>>(d, out, 32, 4096); CK(hipStreamSynchronize(s0));
      CK(hipEventRecord(e0, s0));
      chase<<<1, 64, 0, s0>>>(d, out, 48, 4096);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-26s HBM chase (4096 hops) alone, clocks up: %6.2f us per hop\n", name, ms * 1e3 / 4096);
    }
    // the pull's own rate
    for (int W : {8, 64, 256}) {
      const int trips = 50;
      pull<<<W, 256, 0, s1>>>(reinterpret_cast<const float4*>(hd), dst, rows, rowf4, trips, 3); CK(hipStreamSynchronize(s1));
      CK(hipEventRecord(e0, s1));
      pull<<<W, 256, 0, s1>>>(reinterpret_cast<const float4*>(hd), dst, rows, rowf4, trips, 9);
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = double(W) * 4 * 8 * rowf4 * 16 * trips;
      printf("%-26s pull alone, %3d workgroups: %7.1f us per trip, %6.1f GB/s\n", name, W, ms * 1e3 / trips, bytes / (ms * 1e-3) / 1e9);
    }
    CK(hipFree(d)); CK(hipFree(dst));
    if (mode == 0) CK(hipHostFree(h)); else { CK(hipHostUnregister(h)); munmap(h, host_bytes); }
  }
  return 0;
}
