// Probe: uniform chunk sizes at chunk-aligned addresses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
int main() {
  hipSetDevice(0);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MB = 1 << 20;
  for (size_t chunk : {64 * MB, 256 * MB, 1024 * MB}) {
    for (int aligned = 0; aligned < 2; ++aligned) {
      void* raw = nullptr;
      hipError_t e = hipMemAddressReserve(&raw, (size_t(24) << 30) + chunk, 2 * MB, nullptr, 0);
      char* base = (char*)raw;
      if (aligned) base = (char*)(((uintptr_t)raw + chunk - 1) / chunk * chunk);
      int ok = 0, bad = 0;
      for (int i = 0; i < 20; ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t c = hipMemCreate(&h, chunk, &prop, 0);
        hipError_t m = c == hipSuccess ? hipMemMap(base + i * chunk, chunk, 0, h, 0) : c;
        hipError_t a = m == hipSuccess ? hipMemSetAccess(base + i * chunk, chunk, &acc, 1) : m;
        (void)hipGetLastError();
        if (a == hipSuccess) ++ok; else ++bad;
      }
      printf("chunk %4zu MB, base %s (%p): %d ok, %d failed (reserve %s)\n", chunk / MB,
             aligned ? "chunk-aligned" : "as returned", (void*)base, ok, bad, hipGetErrorString(e));
    }
  }
  return 0;
}
