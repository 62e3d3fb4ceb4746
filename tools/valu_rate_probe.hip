// dev probe: issue rate of the VALU instructions the search kernels are made of (wave-instructions per cycle per SIMD),
// 8 waves per SIMD, every CU busy.  build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[16]; f32x2 p[8]; uint32_t u[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; u[i] = (uint32_t)(i * 77 + threadIdx.x); }
  for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
  float b = seed * 1.5f; f32x2 pb = {b, b}; uint32_t ub = 12345u + threadIdx.x, uc = 0xFFFFFFE0u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pb));
      if (KIND == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
      if (KIND == 3) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
      if (KIND == 4) asm volatile("v_and_or_b32 %0, %0, %1, 5" : "+v"(u[i]) : "v"(uc));
      if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pb));
      if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub));
      if (KIND == 8) { unsigned long long m; asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(m) : "v"(*(unsigned long long*)&u[(i & 7) * 2]), "v"(*(unsigned long long*)&u[((i + 1) & 7) * 2])); }
      if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (KIND == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
      if (KIND == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
      if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(pb));
    }
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i]; for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* d) {
  const int iters = 4096, blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
  hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 16;      // wave-instructions
  const double per_simd = winstr / 1024.0;
  printf("%-14s %8.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_f32", d); run<6>("v_mul_f32", d); run<9>("v_fma_f32", d); run<1>("v_pk_add_f32", d); run<5>("v_pk_mul_f32", d); run<12>("v_pk_fma_f32", d);
  run<2>("v_min_u32", d); run<3>("v_med3_u32", d); run<4>("v_and_or_b32", d); run<7>("v_cndmask_b32", d); run<8>("v_cmp_lt_u64", d);
  run<10>("v_mul_u32_u24", d); run<11>("v_mul_lo_u32", d);
  return 0;
}
