// Cost of an atomics-only grid barrier (no __threadfence) among N resident workgroups of 1024
// threads, against an empty kernel of the same grid and against two back-to-back empty kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void empty_kernel(unsigned* sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && sink[1] == 12345u) sink[2] = 1;
}

__global__ __launch_bounds__(1024) void barrier_kernel(unsigned* ctr, unsigned target, unsigned* sink) {
  __shared__ unsigned ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0, v = 0;
    do {
      v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while (v < target && ++spins < 2000000u);
    ok = v >= target;
  }
  __syncthreads();
  if (!ok && threadIdx.x == 0) sink[0] = 1;   // bailed out
}

int main() {
  CK(hipSetDevice(0));
  unsigned *ctr, *sink;
  CK(hipMalloc(&ctr, 4096));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(sink, 0, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (unsigned n : {32u, 64u, 128u, 200u}) {
    float t_empty = 0, t_two = 0, t_bar = 0;
    const int reps = 200;
    for (int it = 0; it < 20; ++it) empty_kernel<<<n, 1024>>>(sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) empty_kernel<<<n, 1024>>>(sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_empty, e0, e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) { empty_kernel<<<n, 1024>>>(sink); empty_kernel<<<n, 1024>>>(sink); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_two, e0, e1));
    CK(hipMemset(ctr, 0, 4096));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) barrier_kernel<<<n, 1024>>>(ctr, n * (it + 1), sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_bar, e0, e1));
    unsigned h[2];
    CK(hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost));
    printf("%3u workgroups: empty %.2f us, two empties %.2f us, one kernel with a barrier %.2f us%s\n", n,
           t_empty * 1e3 / reps, t_two * 1e3 / reps, t_bar * 1e3 / reps, h[0] ? "  (BAILED OUT)" : "");
  }
  return 0;
}
