// Micro-benchmark: host cost of (a) 20 eager launches vs (b) 20 hipGraphExecKernelNodeSetParams
// + 1 hipGraphLaunch, for short dependent kernels (decides whether the fetch chain is worth
// capturing in a hipGraph).  hipcc --offload-arch=gfx950 graph_update_cost.hip -o graph_update_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(int* p, int v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += v; }
int main() {
  const int N = 20, n = 10000;
  int* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  // (a) eager
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipStreamSynchronize(s));
    auto t0 = now();
    for (int it = 0; it < 200; ++it)
      for (int i = 0; i < N; ++i) k<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(d, i + it, n);
    auto t1 = now();
    CK(hipStreamSynchronize(s));
    auto t2 = now();
    printf("eager : host %.1f us / %d launches (%.2f us each), +drain %.1f us\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 200 / N,
           std::chrono::duration<double, std::micro>(t2 - t1).count());
  }
  // (b) graph with per-replay param updates
  hipGraph_t g; CK(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> nodes(N);
  int v = 0, nn = n; int* dp = d;
  void* args[3] = {&dp, &v, &nn};
  hipKernelNodeParams kp{};
  kp.func = (void*)k; kp.gridDim = dim3((n + 255) / 256); kp.blockDim = dim3(256);
  kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
  for (int i = 0; i < N; ++i) {
    // two independent chains of N/2
    hipGraphNode_t* dep = (i == 0 || i == N / 2) ? nullptr : &nodes[i - 1];
    CK(hipGraphAddKernelNode(&nodes[i], g, dep, dep ? 1 : 0, &kp));
  }
  hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipStreamSynchronize(s));
    auto t0 = now();
    double upd = 0;
    for (int it = 0; it < 200; ++it) {
      auto u0 = now();
      if (rep < 2) for (int i = 0; i < N; ++i) { v = i + it; kp.gridDim = dim3((n + 255) / 256 + (it & 1)); CK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp)); }
      upd += std::chrono::duration<double, std::micro>(now() - u0).count();
      CK(hipGraphLaunch(ge, s));
    }
    auto t1 = now();
    CK(hipStreamSynchronize(s));
    auto t2 = now();
    printf("graph : host %.1f us / replay of %d nodes (updates %.1f us, launch %.1f us), +drain %.1f us\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, N, upd / 200,
           (std::chrono::duration<double, std::micro>(t1 - t0).count() - upd) / 200,
           std::chrono::duration<double, std::micro>(t2 - t1).count());
  }
  int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
  printf("check value %d\n", h);
  return 0;
}
