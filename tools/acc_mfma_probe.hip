// Dev probe (not product code) for DESIGN.md section 5, next lever 1: the point-to-plane accumulation as an f64 MFMA rank
// update.  Per pair the accumulation kernel forms v = (e0..e5, res, 1) in f32 exactly as the reference does
// (e = [(d+s) x n ; n], res = n.(d-s)); the 28 sums it needs are entries of sum_i v_i v_i^T:
//   [0] = D[7][7] (count), [1..21] = upper triangle of D[0..5][0..5], [22..27] = D[0..5][6].
// Here: one workgroup of 256 threads takes 1024 pairs, every lane forms the v of its 4 pairs, writes them to LDS
// ([pair][8] floats), then each wave feeds its 256 pairs to v_mfma_f64_16x16x4_f64, 4 pairs per instruction:
// lane l reads component l & 15 (zero beyond 7) of pair 4 g + (l >> 4) as BOTH operands.  The per-wave D tiles are then
// summed in LDS in a fixed order.  Checked against a scalar f64 host accumulation of the same f32 terms.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256, PAIRS_PER_BLOCK = 1024, WAVES = THREADS / 64;

__device__ __forceinline__ void form_v(const float* __restrict__ q, const float* __restrict__ p, const float* __restrict__ n,
                                       const float dm[3], const float sm[3], float v[8]) {
  const float d0 = __fsub_rn(p[0], dm[0]), d1 = __fsub_rn(p[1], dm[1]), d2 = __fsub_rn(p[2], dm[2]);
  const float s0 = __fsub_rn(q[0], sm[0]), s1 = __fsub_rn(q[1], sm[1]), s2 = __fsub_rn(q[2], sm[2]);
  const float a0 = __fadd_rn(d0, s0), a1 = __fadd_rn(d1, s1), a2 = __fadd_rn(d2, s2);
  const float r0 = __fsub_rn(d0, s0), r1 = __fsub_rn(d1, s1), r2 = __fsub_rn(d2, s2);
  v[0] = __fsub_rn(__fmul_rn(a1, n[2]), __fmul_rn(a2, n[1]));
  v[1] = __fsub_rn(__fmul_rn(a2, n[0]), __fmul_rn(a0, n[2]));
  v[2] = __fsub_rn(__fmul_rn(a0, n[1]), __fmul_rn(a1, n[0]));
  v[3] = n[0]; v[4] = n[1]; v[5] = n[2];
  v[6] = __fadd_rn(__fmul_rn(n[0], r0), __fadd_rn(__fmul_rn(n[1], r1), __fmul_rn(n[2], r2)));
  v[7] = 1.0f;
}

__global__ __launch_bounds__(THREADS) void k_acc_mfma(const float* __restrict__ q, const float* __restrict__ p, const float* __restrict__ n,
                                                      size_t npairs, const float* __restrict__ means /*dm(3), sm(3)*/, double* __restrict__ partials /*[blocks][28]*/) {
  __shared__ float vs[PAIRS_PER_BLOCK * 8];
  __shared__ double dt[WAVES][4][64];
  const float dm[3] = {means[0], means[1], means[2]}, sm[3] = {means[3], means[4], means[5]};
  const size_t base = (size_t)blockIdx.x * PAIRS_PER_BLOCK;
  for (int k = threadIdx.x; k < PAIRS_PER_BLOCK; k += THREADS) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t i = base + k;
    if (i < npairs) form_v(q + 3 * i, p + 3 * i, n + 3 * i, dm, sm, v);
    float4* dst = reinterpret_cast<float4*>(vs + 8 * k);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int comp = lane & 15, sub = lane >> 4;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int g = 0; g < PAIRS_PER_BLOCK / WAVES / 4; ++g) {
    const int pair = wave * (PAIRS_PER_BLOCK / WAVES) + 4 * g + sub;
    const double x = comp < 8 ? (double)vs[8 * pair + comp] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) dt[wave][r][lane] = acc[r];       // D[(lane >> 4) + 4 r][lane & 15]
  __syncthreads();
  if (threadIdx.x < 28) {
    // slot -> (row, col) of D
    int row, col;
    const int s = threadIdx.x;
    if (s == 0) { row = 7; col = 7; }
    else if (s >= 22) { row = s - 22; col = 6; }
    else { int k = s - 1; row = 0; while (k >= 6 - row) { k -= 6 - row; ++row; } col = row + k; }
    const int l = (row & 3) * 16 + col, r = row >> 2;               // row = (l >> 4) + 4 r  with l >> 4 = row & 3
    double sum = 0.0;
    for (int w = 0; w < WAVES; ++w) sum += dt[w][r][l];
    partials[(size_t)blockIdx.x * 28 + s] = sum;
  }
}

int main() {
  const size_t np = 4'000'003;
  std::vector<float> q(3 * np), p(3 * np), n(3 * np);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) * (1.0 / 16777216.0)); };
  for (size_t i = 0; i < np; ++i) {
    for (int c = 0; c < 3; ++c) { p[3 * i + c] = rnd(); q[3 * i + c] = p[3 * i + c] + 0.01f * (rnd() - 0.5f); }
    float nx = rnd() - 0.5f, ny = rnd() - 0.5f, nz = rnd() - 0.5f;
    const float inv = 1.0f / std::sqrt(nx * nx + ny * ny + nz * nz + 1e-12f);
    n[3 * i] = nx * inv; n[3 * i + 1] = ny * inv; n[3 * i + 2] = nz * inv;
  }
  const float means[6] = {0.5f, 0.5f, 0.5f, 0.49f, 0.51f, 0.5f};
  // host reference: the same f32 terms (compiled with -ffp-contract=off), products and sums in f64
  double ref[28] = {0};
  for (size_t i = 0; i < np; ++i) {
    const float *pp = &p[3 * i], *qq = &q[3 * i], *nn = &n[3 * i];
    const float d0 = pp[0] - means[0], d1 = pp[1] - means[1], d2 = pp[2] - means[2];
    const float s0 = qq[0] - means[3], s1 = qq[1] - means[4], s2 = qq[2] - means[5];
    const float a0 = d0 + s0, a1 = d1 + s1, a2 = d2 + s2, r0 = d0 - s0, r1 = d1 - s1, r2 = d2 - s2;
    const float e[6] = {a1 * nn[2] - a2 * nn[1], a2 * nn[0] - a0 * nn[2], a0 * nn[1] - a1 * nn[0], nn[0], nn[1], nn[2]};
    const float res = nn[0] * r0 + (nn[1] * r1 + nn[2] * r2);
    ref[0] += 1.0;
    int k = 1;
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) ref[k++] += (double)e[r] * (double)e[c];
    for (int r = 0; r < 6; ++r) ref[22 + r] += (double)res * (double)e[r];
  }
  float *dq, *dp, *dn, *dmeans; double* dpart;
  const int blocks = (int)((np + PAIRS_PER_BLOCK - 1) / PAIRS_PER_BLOCK);
  if (hipMalloc(&dq, q.size() * 4) != hipSuccess) { std::printf("no device\n"); return 2; }
  (void)hipMalloc(&dp, p.size() * 4); (void)hipMalloc(&dn, n.size() * 4); (void)hipMalloc(&dmeans, 24); (void)hipMalloc(&dpart, (size_t)blocks * 28 * 8);
  (void)hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dn, n.data(), n.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dmeans, means, 24, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_acc_mfma, dim3(blocks), dim3(THREADS), 0, 0, dq, dp, dn, np, dmeans, dpart);
  (void)hipEventRecord(e0, 0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k_acc_mfma, dim3(blocks), dim3(THREADS), 0, 0, dq, dp, dn, np, dmeans, dpart);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> part((size_t)blocks * 28);
  (void)hipMemcpy(part.data(), dpart, part.size() * 8, hipMemcpyDeviceToHost);
  double got[28] = {0};
  for (int b = 0; b < blocks; ++b) for (int k = 0; k < 28; ++k) got[k] += part[(size_t)b * 28 + k];
  double worst = 0.0;
  for (int k = 0; k < 28; ++k) worst = std::fmax(worst, std::fabs(got[k] - ref[k]) / (std::fabs(ref[k]) + 1.0));
  std::printf("MFMA accumulation of %zu pairs: max relative difference of the 28 sums vs scalar f64 = %.3e (%s); %.3f ms per pass = %.0f M pairs/s (36 B/pair read: %.0f GB/s)\n",
              np, worst, worst < 1e-12 ? "OK" : "MISMATCH", ms / 10, np / (ms / 10) / 1e3, 36.0 * np / (ms / 10) / 1e6);
  return worst < 1e-12 ? 0 : 1;
}
