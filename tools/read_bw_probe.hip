// dev probe: what a read-only streaming kernel reaches on this box, in the access shape of the warm-started iteration
// (k_warm<ACC, 2>: per query 16 B + 12 B + 12 B from three arrays, coalesced, one query per lane and round) -- the practical
// ceiling next to the 8 TB/s spec peak and the device-copy figure on the bench line.
// build: hipcc -O3 --offload-arch=gfx950 tools/read_bw_probe.hip -o tools/bin/read_bw_probe ; run: tools/bin/read_bw_probe [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
struct F3 { float x, y, z; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// chunked like k_warm: block b owns a contiguous chunk, XCD-aware remap, one element per lane and round, DEPTH rounds in flight
template <int DEPTH, bool XCD>
__global__ __launch_bounds__(256) void k_read3(const float4* __restrict__ a, const F3* __restrict__ b, const F3* __restrict__ c, uint32_t n, float* out) {
  const uint32_t nb = gridDim.x;
  const uint32_t vb = XCD ? (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t chunk = (((n + nb - 1) / nb) + 255u) & ~255u;
  const uint64_t beg64 = (uint64_t)vb * chunk;
  const uint32_t beg = beg64 < n ? (uint32_t)beg64 : n, end = beg64 + chunk < n ? (uint32_t)(beg64 + chunk) : n;
  float s = 0.f;
  for (uint32_t i0 = beg + threadIdx.x; i0 < end; i0 += 256u * DEPTH) {
    float4 ra[DEPTH]; F3 rb[DEPTH], rc[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { const uint32_t i = i0 + 256u * d; if (i < end) { ra[d] = a[i]; rb[d] = b[i]; rc[d] = c[i]; } else { ra[d] = make_float4(0, 0, 0, 0); rb[d] = F3{0, 0, 0}; rc[d] = F3{0, 0, 0}; } }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) s += ra[d].x + ra[d].y + ra[d].z + ra[d].w + rb[d].x + rb[d].y + rb[d].z + rc[d].x + rc[d].y + rc[d].z;
  }
  if (s == 12345.678f) out[0] = s;
}
// one flat array of float4, grid-stride
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read1(const float4* __restrict__ a, size_t n4, float* out) {
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256u;
  for (size_t i0 = (size_t)blockIdx.x * 256u + threadIdx.x; i0 < n4; i0 += stride * DEPTH) {
    float4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { const size_t i = i0 + stride * d; r[d] = i < n4 ? a[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) s += r[d].x + r[d].y + r[d].z + r[d].w;
  }
  if (s == 12345.678f) out[0] = s;
}
// the same sweep from the END of the buffer (rev): what a pass that follows a forward pass finds in the memory-side cache (256 MB
// "infinity cache"): a forward pass over more than the cache leaves its TAIL there, a pass in the same direction starts at the head
// (every line evicted before it is reached again), a pass in the opposite direction starts where the last one ended
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read1_dir(const float4* __restrict__ a, size_t n4, int rev, float* out) {
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256u;
  for (size_t i0 = (size_t)blockIdx.x * 256u + threadIdx.x; i0 < n4; i0 += stride * DEPTH) {
    float4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { const size_t i = i0 + stride * d; r[d] = i < n4 ? a[rev ? n4 - 1 - i : i] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) s += r[d].x + r[d].y + r[d].z + r[d].w;
  }
  if (s == 12345.678f) out[0] = s;
}
template <class F>
static double time_ms(F f, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 10000000u;
  float4* a; F3 *b, *c; float* out; char *x, *y;
  CK(hipMalloc(&a, (size_t)n * 16)); CK(hipMalloc(&b, (size_t)n * 12)); CK(hipMalloc(&c, (size_t)n * 12)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, (size_t)n * 16)); CK(hipMemset(b, 0, (size_t)n * 12)); CK(hipMemset(c, 0, (size_t)n * 12));
  const size_t G = 1ull << 30;
  CK(hipMalloc(&x, G)); CK(hipMalloc(&y, G)); CK(hipMemset(x, 1, G));
  const double bytes3 = 40.0 * n;
  printf("n = %u: three-array stream %.1f MB per pass\n", n, bytes3 / 1e6);
  for (int nb : {1024, 2048, 4096, 8192, 16384}) {
    double t1 = time_ms([&] { hipLaunchKernelGGL((k_read3<1, true>), dim3(nb), dim3(256), 0, 0, a, b, c, n, out); });
    double t2 = time_ms([&] { hipLaunchKernelGGL((k_read3<2, true>), dim3(nb), dim3(256), 0, 0, a, b, c, n, out); });
    double t4 = time_ms([&] { hipLaunchKernelGGL((k_read3<4, true>), dim3(nb), dim3(256), 0, 0, a, b, c, n, out); });
    double t2n = time_ms([&] { hipLaunchKernelGGL((k_read3<2, false>), dim3(nb), dim3(256), 0, 0, a, b, c, n, out); });
    printf("read3 chunked blocks=%5d: depth1 %.4f ms %.2f TB/s | depth2 %.4f ms %.2f TB/s | depth4 %.4f ms %.2f TB/s | depth2 no-xcd-map %.4f ms %.2f TB/s\n", nb, t1, bytes3 / t1 / 1e9,
           t2, bytes3 / t2 / 1e9, t4, bytes3 / t4 / 1e9, t2n, bytes3 / t2n / 1e9);
  }
  for (size_t mb : {400ull, 1024ull}) {
    const size_t n4 = mb * (1ull << 20) / 16;
    for (int nb : {2048, 8192, 32768}) {
      double t2 = time_ms([&] { hipLaunchKernelGGL((k_read1<2>), dim3(nb), dim3(256), 0, 0, (const float4*)x, n4, out); });
      double t4 = time_ms([&] { hipLaunchKernelGGL((k_read1<4>), dim3(nb), dim3(256), 0, 0, (const float4*)x, n4, out); });
      printf("read1 %4zu MiB grid-stride blocks=%5d: depth2 %.4f ms %.2f TB/s | depth4 %.4f ms %.2f TB/s\n", mb, nb, t2, n4 * 16.0 / t2 / 1e9, t4, n4 * 16.0 / t4 / 1e9);
    }
  }
  // the memory-side cache: repeated passes over buffers below and above its size, and passes that alternate direction
  for (size_t mb : {64ull, 128ull, 192ull, 256ull, 320ull, 400ull, 800ull}) {
    const size_t n4 = mb * (1ull << 20) / 16;
    const int nb = 8192;
    double tf = time_ms([&] { hipLaunchKernelGGL((k_read1_dir<4>), dim3(nb), dim3(256), 0, 0, (const float4*)x, n4, 0, out); });
    int flip = 0;
    double ta = time_ms([&] { hipLaunchKernelGGL((k_read1_dir<4>), dim3(nb), dim3(256), 0, 0, (const float4*)x, n4, flip, out); flip ^= 1; });
    printf("read1 %4zu MiB repeated: same direction %.4f ms %.2f TB/s | alternating direction %.4f ms %.2f TB/s\n", mb, tf, n4 * 16.0 / tf / 1e9, ta, n4 * 16.0 / ta / 1e9);
  }
  double tc = time_ms([&] { CK(hipMemcpyAsync(y, x, G, hipMemcpyDeviceToDevice, 0)); }, 10);
  printf("device copy 1 GiB: %.4f ms = %.2f TB/s (read + write)\n", tc, 2.0 * G / tc / 1e9);
  return 0;
}
