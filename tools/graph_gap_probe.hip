// dev probe: what a chain of DEPENDENT kernels costs per link on this box when it is enqueued launch by launch on a stream, as one
// hipGraph of the whole chain, and as a 3-node graph launched once per "iteration" (the shape of a warm-started ICP iteration:
// k_warm -> k_reduce_stage1 -> k_solve).  The kernels do next to nothing (one block, one dependent load + store): the time is the links.
// build: hipcc -O3 --offload-arch=gfx950 tools/graph_gap_probe.hip -o tools/bin/graph_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_link(const float* in, float* out, int blocks_work) {
  float v = in[threadIdx.x & 63];
  for (int i = 0; i < blocks_work; ++i) v = v * 1.0000001f + 1e-9f;
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = v;
}
int main() {
  float *a, *b; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int iters = 20, links = 3;
  for (int grid : {1, 32, 2048}) {
    auto chain = [&](int n) { for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_link, dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 64); } };
    auto timed = [&](auto f) {
      std::vector<double> t;
      for (int r = 0; r < 12; ++r) { CK(hipStreamSynchronize(s)); auto t0 = std::chrono::steady_clock::now(); f(); CK(hipStreamSynchronize(s)); t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
      std::sort(t.begin(), t.end()); return t[t.size() / 2];
    };
    const double t_stream = timed([&] { chain(iters * links); });
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(iters * links); CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double t_graph_all = timed([&] { CK(hipGraphLaunch(ge, s)); });
    hipGraph_t g3; hipGraphExec_t ge3;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(links); CK(hipStreamEndCapture(s, &g3));
    auto t0 = std::chrono::steady_clock::now(); CK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0)); const double t_inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const double t_graph3 = timed([&] { for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge3, s)); });
    printf("grid %4d: %d dependent kernels: stream %.1f us (%.2f per link) | one graph %.1f us (%.2f) | %d launches of a 3-node graph %.1f us (%.2f per link; instantiate %.0f us)\n",
           grid, iters * links, t_stream, t_stream / (iters * links), t_graph_all, t_graph_all / (iters * links), iters, t_graph3, t_graph3 / (iters * links), t_inst);
  }
  return 0;
}
