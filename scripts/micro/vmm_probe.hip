// Probe: does the HIP virtual-memory API work on this box, and what does mapping cost?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(char* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride * 4096) p[i] = 1;
}
int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu\n", gran);
  const size_t va = size_t(256) << 30, chunk = size_t(1) << 30;
  void* base = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  CK(hipMemAddressReserve(&base, va, gran, nullptr, 0));
  auto t1 = std::chrono::steady_clock::now();
  printf("reserve 256 GiB VA: %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int i = 0; i < 8; ++i) {
    auto a = std::chrono::steady_clock::now();
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)base + i * chunk, chunk, 0, h, 0));
    CK(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
    CK(hipMemRelease(h));
    auto b = std::chrono::steady_clock::now();
    touch<<<256, 256>>>((char*)base, (i + 1) * chunk);
    CK(hipDeviceSynchronize());
    auto c = std::chrono::steady_clock::now();
    printf("chunk %d: create+map+access %.3f ms, touch all %.3f ms\n", i,
           std::chrono::duration<double, std::milli>(b - a).count(),
           std::chrono::duration<double, std::milli>(c - b).count());
  }
  // compare: plain hipMalloc of 8 GiB
  auto a = std::chrono::steady_clock::now();
  void* p = nullptr;
  CK(hipMalloc(&p, size_t(8) << 30));
  auto b = std::chrono::steady_clock::now();
  CK(hipFree(p));
  auto c = std::chrono::steady_clock::now();
  printf("hipMalloc 8 GiB %.3f ms, hipFree %.3f ms\n",
         std::chrono::duration<double, std::milli>(b - a).count(),
         std::chrono::duration<double, std::milli>(c - b).count());
  for (int i = 0; i < 8; ++i) CK(hipMemUnmap((char*)base + i * chunk, chunk));
  CK(hipMemAddressFree(base, va));
  printf("ok\n");
  return 0;
}
