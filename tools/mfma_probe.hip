// Dev probe (not product code): operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, for the planned in-tile
// accumulation (DESIGN.md section 5, next levers): D(16x16) += A(16x4) * B(4x16) per wave instruction.
// Hypothesis: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]; result register r of lane l holds
// D[i = (l / 16) + 4 * r][j = l % 16].  With A = B^T = X (16 x 4) the instruction adds sum_k x_k x_k^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double* __restrict__ X /*[nk][16]*/, int nk, double* __restrict__ D /*[16][16]*/) {
  const int l = threadIdx.x;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < nk; k0 += 4) {
    const double x = X[(k0 + l / 16) * 16 + (l % 16)];      // component l%16 of vector k0 + l/16: feeds A and B alike
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l / 16) + 4 * r) * 16 + (l % 16)] = acc[r];   // f64 form: row = (lane >> 4) + 4 * reg (NOT the f32 map)
}

int main() {
  const int nk = 64;
  std::vector<double> X(nk * 16), ref(256, 0.0), out(256, 0.0);
  unsigned s = 12345u;
  for (auto& v : X) { s = s * 1664525u + 1013904223u; v = (double)(float)((s >> 8) * (1.0 / 16777216.0) - 0.5); }
  for (int k = 0; k < nk; ++k)
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) ref[i * 16 + j] += X[k * 16 + i] * X[k * 16 + j];
  double *dX, *dD;
  if (hipMalloc(&dX, X.size() * 8) != hipSuccess || hipMalloc(&dD, 256 * 8) != hipSuccess) { std::printf("no device\n"); return 2; }
  (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dX, nk, dD);
  (void)hipMemcpy(out.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int i = 0; i < 256; ++i) worst = std::fmax(worst, std::fabs(out[i] - ref[i]));
  std::printf("mfma_f64_16x16x4 rank update with the hypothesised layout: max |D - ref| = %.3e (%s)\n", worst, worst < 1e-12 ? "layout confirmed" : "layout WRONG");
  return worst < 1e-12 ? 0 : 1;
}
