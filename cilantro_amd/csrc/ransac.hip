// ransac.hip -- PlaneRANSACEstimator3f on the device.  Replaces cilantro's
//   model_estimation/ransac_base.hpp:64-131                       (estimate loop)
//   model_estimation/ransac_hyperplane_estimator.hpp:47-55, 70-85  (computeResiduals, estimate_params_)
//   core/principal_component_analysis.hpp:76-84, core/covariance.hpp:64-77 / 125-141
//
// The reference scores ONE hypothesis per pass over the points (residual vector + inlier list per
// iteration).  Here hypotheses are scored in rounds of RS_ROUND planes per pass: a lane keeps 8 points in
// registers and walks the round's planes through the scalar cache; only inlier COUNTS leave the kernel
// (per-block, summed in a fixed order), and a one-lane kernel replays the reference's sequential
// "better than best / target reached" decisions over them, so the outcome is the one the sequential loop
// would reach with the same samples.  Rounds after the stopping iteration early-exit on a device flag.
// Residuals and the ordered inlier list are produced once, for the final model.
//
// Numeric contract: absDistance = |(n0*x + (n1*y + n2*z)) + offset| in f32, no FMA contraction (packed
// v_pk_mul/add round each step like the scalar ops) => inlier sets are bit-identical to the reference's for
// the same plane.  Model fit: f32 per-term arithmetic, f64 accumulation in a fixed order (bitwise
// reproducible), f64 Jacobi eigen-solve (solve.hpp) in place of Eigen's f32 SelfAdjointEigenSolver.
#include "../../include/cilantro_hip/c_api.h"
#include "solve.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int RS_THREADS = 256;
constexpr int RS_PTS = 8;        // points per lane held in registers
constexpr int RS_ROUND = 128;    // hypotheses scored per pass over the points
constexpr int RS_MAX_BLOCKS = 1024;

struct RansacState {
  float best[4];
  unsigned int best_cnt;
  unsigned int iterations;
  int done;
  int have_model;
  float mean[3];
  unsigned int n_inliers;
  double moments[12];   // scratch between the two re-estimation passes
};

__device__ __forceinline__ float abs_distance(float n0, float n1, float n2, float off, float x, float y, float z) {
  return fabsf(__fadd_rn(__fadd_rn(__fmul_rn(n0, x), __fadd_rn(__fmul_rn(n1, y), __fmul_rn(n2, z))), off));
}

// estimate_params_ from m accumulated moments: plane through `mean` with the smallest-eigenvalue normal
__device__ void plane_from_cov(const double cs[6], double m, const float mean[3], float plane[4]) {
  if (!(m >= 2.0)) {   // covariance.hpp:93-96: fewer than min_sample_size_ (=2) points -> NaN
    plane[0] = plane[1] = plane[2] = plane[3] = NAN;
    return;
  }
  const double inv = m - 1.0;
  const double C[9] = {cs[0] / inv, cs[1] / inv, cs[2] / inv, cs[1] / inv, cs[3] / inv, cs[4] / inv, cs[2] / inv, cs[4] / inv, cs[5] / inv};
  double w[3], V[9];
  cilhip::sym_eig3(C, w, V);
  const float n0 = (float)V[2], n1 = (float)V[5], n2 = (float)V[8];
  plane[0] = n0; plane[1] = n1; plane[2] = n2;
  plane[3] = -__fadd_rn(__fmul_rn(n0, mean[0]), __fadd_rn(__fmul_rn(n1, mean[1]), __fmul_rn(n2, mean[2])));
}

// one lane per hypothesis: PCA of its sample (ransac_base.hpp:94 -> estimate_params_(sample_ind, .))
__global__ void k_models(const float* __restrict__ xyz, const uint32_t* __restrict__ samples, uint32_t sample_size, uint32_t m_total,
                         uint32_t m_padded, float4* planes) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= m_padded) return;
  if (h >= m_total) { planes[h] = make_float4(NAN, NAN, NAN, NAN); return; }   // padding: never an inlier
  float p[3][3];
  double s[3] = {0, 0, 0};
  for (uint32_t i = 0; i < sample_size; ++i) {
    const size_t idx = samples[3 * (size_t)h + i];
    for (int d = 0; d < 3; ++d) { p[i][d] = xyz[3 * idx + d]; s[d] += (double)p[i][d]; }
  }
  float mean[3];
  for (int d = 0; d < 3; ++d) mean[d] = (float)(s[d] / (double)sample_size);
  double cs[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t i = 0; i < sample_size; ++i) {
    const float t0 = __fsub_rn(p[i][0], mean[0]), t1 = __fsub_rn(p[i][1], mean[1]), t2 = __fsub_rn(p[i][2], mean[2]);
    cs[0] += (double)__fmul_rn(t0, t0); cs[1] += (double)__fmul_rn(t0, t1); cs[2] += (double)__fmul_rn(t0, t2);
    cs[3] += (double)__fmul_rn(t1, t1); cs[4] += (double)__fmul_rn(t1, t2); cs[5] += (double)__fmul_rn(t2, t2);
  }
  float pl[4];
  plane_from_cov(cs, (double)sample_size, mean, pl);
  planes[h] = make_float4(pl[0], pl[1], pl[2], pl[3]);
}

// inlier counts of `m` (multiple of 4, <= RS_ROUND) planes over all points -> partial[block][RS_ROUND]
__global__ __launch_bounds__(RS_THREADS) void k_score(const float* __restrict__ xyz, uint32_t n, const float* __restrict__ planes,
                                                      uint32_t m, float thr, uint32_t* __restrict__ partial,
                                                      const RansacState* __restrict__ st) {
  if (st && st->done) return;
  __shared__ uint32_t cnt[RS_ROUND];
  for (int t = threadIdx.x; t < RS_ROUND; t += RS_THREADS) cnt[t] = 0;
  __syncthreads();
  static_assert(RS_ROUND == 128, "two per-wave count registers: hypothesis h in lane h & 63 of register h >> 6");
  uint32_t cv0 = 0, cv1 = 0;
  constexpr uint32_t TILE = RS_THREADS * RS_PTS;
  for (size_t base = (size_t)blockIdx.x * TILE; base < n; base += (size_t)gridDim.x * TILE) {
    f32x2 px[RS_PTS / 2], py[RS_PTS / 2], pz[RS_PTS / 2];
#pragma unroll
    for (int k = 0; k < RS_PTS / 2; ++k) {
      const size_t i0 = base + (size_t)(2 * k) * RS_THREADS + threadIdx.x, i1 = i0 + RS_THREADS;
      const bool v0 = i0 < n, v1 = i1 < n;   // out of range: NaN coordinates, |NaN| <= thr is false
      px[k] = (f32x2){v0 ? xyz[3 * i0] : NAN, v1 ? xyz[3 * i1] : NAN};
      py[k] = (f32x2){v0 ? xyz[3 * i0 + 1] : NAN, v1 ? xyz[3 * i1 + 1] : NAN};
      pz[k] = (f32x2){v0 ? xyz[3 * i0 + 2] : NAN, v1 ? xyz[3 * i1 + 2] : NAN};
    }
    // (the four planes of the NEXT trip are requested before this trip's arithmetic: the scalar loads' latency used to sit in
    //  front of every trip)
    // The count of hypothesis h lives in LANE h & 63 of one of two per-wave registers across all the tiles of the block (read, add,
    // write back: three instructions instead of a wave-aggregated LDS atomic per hypothesis and tile); the waves' registers meet in
    // LDS once, at the end.
    float cn[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) cn[t] = planes[t];
    auto half_pass = [&](uint32_t jbeg, uint32_t jend, uint32_t& cv) {
      for (uint32_t j = jbeg; j < jend; j += 4) {
        float c[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) c[t] = cn[t];
        const uint32_t jn = j + 4 < m ? j + 4 : j;
#pragma unroll
        for (int t = 0; t < 16; ++t) cn[t] = planes[4 * jn + t];   // wave-uniform: wide scalar loads
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x2 n0 = {c[4 * u], c[4 * u]}, n1 = {c[4 * u + 1], c[4 * u + 1]}, n2 = {c[4 * u + 2], c[4 * u + 2]},
                      off = {c[4 * u + 3], c[4 * u + 3]};
          uint32_t tot = 0;
#pragma unroll
          for (int k = 0; k < RS_PTS / 2; ++k) {
            const f32x2 r = (n0 * px[k] + (n1 * py[k] + n2 * pz[k])) + off;   // -ffp-contract=off
            tot += (uint32_t)__popcll(__ballot(fabsf(r.x) <= thr)) + (uint32_t)__popcll(__ballot(fabsf(r.y) <= thr));
          }
          const int ln = (int)((j + u) & 63u);
          const uint32_t upd = (uint32_t)__builtin_amdgcn_readlane((int)cv, ln) + tot;      // (wave-uniform: scalar registers)
          asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(cv) : "s"(upd), "s"(ln) : "m0");      // (one scalar operand per VALU instruction on this target: the lane goes through M0)
        }
      }
    };
    half_pass(0, m < 64u ? m : 64u, cv0);
    if (m > 64u) half_pass(64, m, cv1);
  }
  {
    const int lane = (int)(threadIdx.x & 63);
    if (cv0) atomicAdd(&cnt[lane], cv0);
    if (cv1) atomicAdd(&cnt[64 + lane], cv1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < RS_ROUND; t += RS_THREADS) partial[(size_t)blockIdx.x * RS_ROUND + t] = cnt[t];
}

// sums the per-block counts and replays ransac_base.hpp:103-114 over this round's hypotheses, in order
constexpr int PICK_GROUPS = 8;
__global__ __launch_bounds__(RS_ROUND * PICK_GROUPS) void k_pick(const uint32_t* __restrict__ partial, int nblocks, const float4* __restrict__ planes, uint32_t m,
                       uint32_t sample_size, uint32_t target, RansacState* st, uint32_t* counts_out) {
  __shared__ uint32_t cnt[RS_ROUND];
  __shared__ uint32_t part[PICK_GROUPS][RS_ROUND];
  if (st && st->done) return;
  // (integer sums: any order gives the same counts -- PICK_GROUPS threads per hypothesis, four independent loads in flight each;
  //  one thread per hypothesis walking all the blocks' rows took 0.13 ms of a 1.6 ms pass)
  const int t = threadIdx.x & (RS_ROUND - 1), grp = threadIdx.x / RS_ROUND;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  int g = grp;
  for (; g + 3 * PICK_GROUPS < nblocks; g += 4 * PICK_GROUPS) {
    c0 += partial[(size_t)g * RS_ROUND + t]; c1 += partial[(size_t)(g + PICK_GROUPS) * RS_ROUND + t];
    c2 += partial[(size_t)(g + 2 * PICK_GROUPS) * RS_ROUND + t]; c3 += partial[(size_t)(g + 3 * PICK_GROUPS) * RS_ROUND + t];
  }
  for (; g < nblocks; g += PICK_GROUPS) c0 += partial[(size_t)g * RS_ROUND + t];
  part[grp][t] = (c0 + c1) + (c2 + c3);
  __syncthreads();
  if (grp == 0) {
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < PICK_GROUPS; ++k) c += part[k][t];
    cnt[t] = c;
    if (counts_out && (uint32_t)t < m) counts_out[t] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0 && st) {
    for (uint32_t h = 0; h < m; ++h) {
      st->iterations++;                                     // :103
      if (cnt[h] < sample_size) continue;                   // :104
      if (cnt[h] > st->best_cnt) {                          // :107-111
        const float4 p = planes[h];
        st->best[0] = p.x; st->best[1] = p.y; st->best[2] = p.z; st->best[3] = p.w;
        st->best_cnt = cnt[h];
        st->have_model = 1;
      }
      if (st->best_cnt >= target) { st->done = 1; break; }  // :114
    }
  }
}

// fixed-order block reduction of K f64 values per thread -> out[K] by thread 0..K-1
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double* out) {
  __shared__ double red[RS_THREADS / 64][K];
#pragma unroll
  for (int k = 0; k < K; ++k)
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < K; ++k) red[threadIdx.x >> 6][k] = v[k];
  __syncthreads();
  if (threadIdx.x < K) {
    double s = red[0][threadIdx.x];
    for (int w = 1; w < RS_THREADS / 64; ++w) s += red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// pass 0: sum of the inliers of st->best (x,y,z,count).  pass 1: upper triangle of sum (p-mean)(p-mean)^T.
// all_points != 0: every point takes part (estimateModel() without a subset).
template <int PASS>
__global__ __launch_bounds__(RS_THREADS) void k_moments(const float* __restrict__ xyz, uint32_t n, float thr, int all_points,
                                                        const RansacState* __restrict__ st, double* __restrict__ partial) {
  const float n0 = st->best[0], n1 = st->best[1], n2 = st->best[2], off = st->best[3];
  const float m0 = st->mean[0], m1 = st->mean[1], m2 = st->mean[2];
  constexpr int K = PASS == 0 ? 4 : 6;
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.0;
  for (size_t i = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RS_THREADS) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!all_points && !(abs_distance(n0, n1, n2, off, x, y, z) <= thr)) continue;
    if (PASS == 0) {
      acc[0] += (double)x; acc[1] += (double)y; acc[2] += (double)z; acc[3] += 1.0;
    } else {
      const float t0 = __fsub_rn(x, m0), t1 = __fsub_rn(y, m1), t2 = __fsub_rn(z, m2);
      acc[0] += (double)__fmul_rn(t0, t0); acc[1] += (double)__fmul_rn(t0, t1); acc[2] += (double)__fmul_rn(t0, t2);
      acc[3] += (double)__fmul_rn(t1, t1); acc[4] += (double)__fmul_rn(t1, t2); acc[5] += (double)__fmul_rn(t2, t2);
    }
  }
  block_reduce_store<K>(acc, partial + (size_t)blockIdx.x * 8);
}

// PASS 0: mean of the inliers -> st->mean.  PASS 1: covariance -> eigen-solve -> st->best.  One block.
template <int PASS>
__global__ void k_moments_finish(const double* __restrict__ partial, int nblocks, RansacState* st) {
  constexpr int K = PASS == 0 ? 4 : 6;
  __shared__ double tot[8];
  if (threadIdx.x < K) {
    double s = 0.0;
    for (int g = 0; g < nblocks; ++g) s += partial[(size_t)g * 8 + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (PASS == 0) {
      const double m = tot[3];
      st->moments[0] = m;
      for (int d = 0; d < 3; ++d) st->mean[d] = m > 0.0 ? (float)(tot[d] / m) : NAN;
    } else {
      float pl[4];
      const float mean[3] = {st->mean[0], st->mean[1], st->mean[2]};
      plane_from_cov(tot, st->moments[0], mean, pl);
      for (int d = 0; d < 4; ++d) st->best[d] = pl[d];
    }
  }
}

// final model: per-chunk inlier counts -> exclusive scan -> residuals + ordered inlier indices
__global__ __launch_bounds__(RS_THREADS) void k_chunk_counts(const float* __restrict__ xyz, uint32_t n, uint32_t chunk, float thr,
                                                             const RansacState* __restrict__ st, uint32_t* __restrict__ counts) {
  const float n0 = st->best[0], n1 = st->best[1], n2 = st->best[2], off = st->best[3];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  uint32_t c = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RS_THREADS)
    c += abs_distance(n0, n1, n2, off, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]) <= thr ? 1u : 0u;
  __shared__ uint32_t red[RS_THREADS / 64];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_scan_counts(uint32_t* counts, int nblocks, RansacState* st) {   // one lane; nblocks <= 1024
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t run = 0;
    for (int g = 0; g < nblocks; ++g) { const uint32_t c = counts[g]; counts[g] = run; run += c; }
    st->n_inliers = run;
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_write_final(const float* __restrict__ xyz, uint32_t n, uint32_t chunk, float thr,
                                                            const RansacState* __restrict__ st, const uint32_t* __restrict__ offsets,
                                                            float* __restrict__ residuals, uint32_t* __restrict__ inliers) {
  const float n0 = st->best[0], n1 = st->best[1], n2 = st->best[2], off = st->best[3];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  __shared__ uint32_t wave_cnt[RS_THREADS / 64];
  uint32_t run = offsets ? offsets[blockIdx.x] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t b = lo; b < hi; b += RS_THREADS) {
    const size_t i = b + threadIdx.x;
    bool in = false;
    if (i < hi) {
      const float r = abs_distance(n0, n1, n2, off, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      if (residuals) residuals[i] = r;
      in = r <= thr;
    }
    if (inliers) {   // block-uniform
      const unsigned long long bal = __ballot(in);
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int w = 0; w < RS_THREADS / 64; ++w) { before += w < wave ? wave_cnt[w] : 0; total += wave_cnt[w]; }
      if (in) inliers[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)i;
      run += total;
      __syncthreads();
    }
  }
}

#define RS_CK(x)                          \
  do {                                    \
    if ((x) != hipSuccess) {              \
      rc = CILHIP_ERR_HIP;                \
      goto done;                          \
    }                                     \
  } while (0)

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline uint64_t bounded(uint64_t& s, uint64_t bound) {   // uniform in [0, bound) (128-bit multiply, bias < 2^-32)
  return (uint64_t)(((unsigned __int128)splitmix64(s) * bound) >> 64);
}

struct Buffers {
  float* xyz = nullptr;
  bool own_xyz = false;
  uint32_t* samples = nullptr;
  float4* planes = nullptr;
  uint32_t* partial = nullptr;
  double* dpartial = nullptr;
  uint32_t* chunk_counts = nullptr;
  RansacState* st = nullptr;
  float* residuals = nullptr;
  uint32_t* inliers = nullptr;
  uint32_t* counts = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  void release() {
    if (own_xyz && xyz) (void)hipFree(xyz);
    if (samples) (void)hipFree(samples);
    if (planes) (void)hipFree(planes);
    if (partial) (void)hipFree(partial);
    if (dpartial) (void)hipFree(dpartial);
    if (chunk_counts) (void)hipFree(chunk_counts);
    if (st) (void)hipFree(st);
    if (residuals) (void)hipFree(residuals);
    if (inliers) (void)hipFree(inliers);
    if (counts) (void)hipFree(counts);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
  }
};

// blocks of the scoring pass: the kernel alternates vector arithmetic with scalar population counts, so it wants many waves per SIMD
// (measured at 50M points, ms per 128-hypothesis pass incl. the pick: 1024 blocks 1.51, 2048 1.59, 4096 1.84, 8192 2.56: one generation of blocks)
#ifndef CILHIP_RS_SCORE_BLOCKS
#define CILHIP_RS_SCORE_BLOCKS 1024
#endif
inline int score_blocks(size_t n) {
  static const size_t cap = [] { const char* e = getenv("CILHIP_EXP_RS_BLOCKS"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)CILHIP_RS_SCORE_BLOCKS; }();
  const size_t tiles = (n + (size_t)RS_THREADS * RS_PTS - 1) / ((size_t)RS_THREADS * RS_PTS);
  return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

}  // namespace

extern "C" {

int cilhip_plane_ransac3f(int device, const float* xyz, size_t n, int mem, const uint32_t* samples, uint64_t seed,
                          float max_residual, size_t target_inliers, size_t max_iter, int re_estimate,
                          cilhip_plane_model* out, float* residuals_out, uint32_t* inliers_out) {
  if (!out || (!xyz && n) || n > 0xFFFFFFF0ull || max_iter > 0x0FFFFFFFull) return CILHIP_ERR_INVALID;
  int rc = CILHIP_OK;
  Buffers b;
  RansacState hs;
  std::memset(&hs, 0, sizeof hs);
  for (int d = 0; d < 4; ++d) hs.best[d] = NAN;
  const uint32_t sample_size = n < 3 ? (uint32_t)n : 3u;          // ransac_base.hpp:67
  if (target_inliers > n) target_inliers = n;                     // :68
  float ms = 0.0f;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    RS_CK(hipSetDevice(device));
    RS_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
    RS_CK(hipEventCreate(&b.e0));
    RS_CK(hipEventCreate(&b.e1));
    RS_CK(hipMalloc(&b.st, sizeof(RansacState)));
    RS_CK(hipMemcpyAsync(b.st, &hs, sizeof hs, hipMemcpyHostToDevice, b.s));
    if (n > 0 && max_iter > 0) {
      if (mem == CILHIP_MEM_DEVICE) {
        b.xyz = const_cast<float*>(xyz);
      } else {
        b.own_xyz = true;
        RS_CK(hipMalloc(&b.xyz, 3 * n * sizeof(float)));
        RS_CK(hipMemcpyAsync(b.xyz, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, b.s));
      }
      // the random samples (ransac_base.hpp:83-91): 3 distinct indices per iteration; drawn on the host
      std::vector<uint32_t> hsamp;
      if (!samples) {
        hsamp.resize(3 * max_iter);
        uint64_t st = seed;
        for (size_t it = 0; it < max_iter; ++it) {
          uint32_t pick[3] = {0, 0, 0};
          for (uint32_t i = 0; i < sample_size; ++i) {
            uint32_t v = (uint32_t)bounded(st, n - i);   // i-th draw among the n-i indices not picked yet
            uint32_t srt[3];
            for (uint32_t a = 0; a < i; ++a) srt[a] = pick[a];
            for (uint32_t a = 0; a + 1 < i; ++a)
              if (srt[a] > srt[a + 1]) { const uint32_t t = srt[a]; srt[a] = srt[a + 1]; srt[a + 1] = t; }
            for (uint32_t a = 0; a < i; ++a) v += v >= srt[a] ? 1u : 0u;
            pick[i] = v;
          }
          for (int i = 0; i < 3; ++i) hsamp[3 * it + i] = pick[i];
        }
        samples = hsamp.data();
      } else {
        for (size_t i = 0; i < 3 * max_iter; ++i)
          if ((i % 3) < sample_size && samples[i] >= n) { rc = CILHIP_ERR_INVALID; goto done; }
      }
      const size_t mpad = (max_iter + RS_ROUND - 1) / RS_ROUND * RS_ROUND;
      const int nb = score_blocks(n);
      RS_CK(hipMalloc(&b.samples, 3 * max_iter * sizeof(uint32_t)));
      RS_CK(hipMalloc(&b.planes, mpad * sizeof(float4)));
      RS_CK(hipMalloc(&b.partial, (size_t)nb * RS_ROUND * sizeof(uint32_t)));
      RS_CK(hipMalloc(&b.dpartial, (size_t)RS_MAX_BLOCKS * 8 * sizeof(double)));
      RS_CK(hipMalloc(&b.chunk_counts, RS_MAX_BLOCKS * sizeof(uint32_t)));
      RS_CK(hipMemcpyAsync(b.samples, samples, 3 * max_iter * sizeof(uint32_t), hipMemcpyHostToDevice, b.s));
      RS_CK(hipEventRecord(b.e0, b.s));
      hipLaunchKernelGGL(k_models, dim3((unsigned)((mpad + 127) / 128)), dim3(128), 0, b.s, b.xyz, b.samples, sample_size,
                         (uint32_t)max_iter, (uint32_t)mpad, b.planes);
      for (size_t r0 = 0; r0 < max_iter; r0 += RS_ROUND) {
        const uint32_t m = (uint32_t)(max_iter - r0 < RS_ROUND ? max_iter - r0 : RS_ROUND);
        const uint32_t m4 = (m + 3u) & ~3u;
        hipLaunchKernelGGL(k_score, dim3(nb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, (const float*)(b.planes + r0), m4,
                           max_residual, b.partial, b.st);
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(RS_ROUND * PICK_GROUPS), 0, b.s, b.partial, nb, b.planes + r0, m, sample_size,
                           (uint32_t)target_inliers, b.st, (uint32_t*)nullptr);
      }
      const int mb = (int)std::min<size_t>((n + RS_THREADS - 1) / RS_THREADS, RS_MAX_BLOCKS);
      if (re_estimate) {   // ransac_base.hpp:118-128: PCA of the best model's inliers (NaN model when there are < 2)
        hipLaunchKernelGGL(k_moments<0>, dim3(mb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, max_residual, 0, b.st, b.dpartial);
        hipLaunchKernelGGL(k_moments_finish<0>, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
        hipLaunchKernelGGL(k_moments<1>, dim3(mb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, max_residual, 0, b.st, b.dpartial);
        hipLaunchKernelGGL(k_moments_finish<1>, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
      }
      const uint32_t chunk = (uint32_t)(((n + mb - 1) / mb + RS_THREADS - 1) / RS_THREADS * RS_THREADS);
      const int cb = (int)((n + chunk - 1) / chunk);
      hipLaunchKernelGGL(k_chunk_counts, dim3(cb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, chunk, max_residual, b.st, b.chunk_counts);
      hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(64), 0, b.s, b.chunk_counts, cb, b.st);
      if (residuals_out) RS_CK(hipMalloc(&b.residuals, n * sizeof(float)));
      if (inliers_out) RS_CK(hipMalloc(&b.inliers, n * sizeof(uint32_t)));
      if (residuals_out || inliers_out)
        hipLaunchKernelGGL(k_write_final, dim3(cb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, chunk, max_residual, b.st,
                           b.chunk_counts, b.residuals, b.inliers);
      RS_CK(hipEventRecord(b.e1, b.s));
      RS_CK(hipGetLastError());
    }
    RS_CK(hipMemcpyAsync(&hs, b.st, sizeof hs, hipMemcpyDeviceToHost, b.s));
    RS_CK(hipStreamSynchronize(b.s));
    if (n > 0 && max_iter > 0) {
      RS_CK(hipEventElapsedTime(&ms, b.e0, b.e1));
      // the reference keeps residuals / inliers of the best hypothesis; with no accepted model they are empty
      const bool any = hs.have_model != 0;
      if (residuals_out && b.residuals) RS_CK(hipMemcpy(residuals_out, b.residuals, n * sizeof(float), hipMemcpyDeviceToHost));
      if (inliers_out && b.inliers && hs.n_inliers)
        RS_CK(hipMemcpy(inliers_out, b.inliers, (size_t)hs.n_inliers * sizeof(uint32_t), hipMemcpyDeviceToHost));
      (void)any;
    }
  }
done:
  for (int d = 0; d < 3; ++d) out->normal[d] = hs.best[d];
  out->offset = hs.best[3];
  out->iterations = hs.iterations;
  out->n_inliers = hs.n_inliers;
  out->target_reached = hs.n_inliers >= target_inliers ? 1 : 0;   // ransac_base.hpp:172
  out->device_ms = (double)ms;
  b.release();
  return rc;
}

int cilhip_plane_score3f(int device, const float* xyz, size_t n, int mem, const float* planes, size_t m, float max_residual,
                         uint32_t* counts_out) {
  if ((!xyz && n) || (!planes && m) || (!counts_out && m) || n > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  if (m == 0) return CILHIP_OK;
  int rc = CILHIP_OK;
  Buffers b;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    RS_CK(hipSetDevice(device));
    RS_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
    if (n == 0) { std::memset(counts_out, 0, m * sizeof(uint32_t)); goto done; }
    if (mem == CILHIP_MEM_DEVICE) {
      b.xyz = const_cast<float*>(xyz);
    } else {
      b.own_xyz = true;
      RS_CK(hipMalloc(&b.xyz, 3 * n * sizeof(float)));
      RS_CK(hipMemcpyAsync(b.xyz, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, b.s));
    }
    const size_t mpad = (m + RS_ROUND - 1) / RS_ROUND * RS_ROUND;
    const int nb = score_blocks(n);
    std::vector<float> hp(4 * mpad, NAN);
    std::memcpy(hp.data(), planes, 4 * m * sizeof(float));
    RS_CK(hipMalloc(&b.planes, mpad * sizeof(float4)));
    RS_CK(hipMalloc(&b.partial, (size_t)nb * RS_ROUND * sizeof(uint32_t)));
    RS_CK(hipMalloc(&b.counts, mpad * sizeof(uint32_t)));
    RS_CK(hipMemcpyAsync(b.planes, hp.data(), 4 * mpad * sizeof(float), hipMemcpyHostToDevice, b.s));
    for (size_t r0 = 0; r0 < m; r0 += RS_ROUND) {
      const uint32_t mm = (uint32_t)(m - r0 < RS_ROUND ? m - r0 : RS_ROUND);
      hipLaunchKernelGGL(k_score, dim3(nb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, (const float*)(b.planes + r0), (mm + 3u) & ~3u,
                         max_residual, b.partial, (const RansacState*)nullptr);
      hipLaunchKernelGGL(k_pick, dim3(1), dim3(RS_ROUND * PICK_GROUPS), 0, b.s, b.partial, nb, b.planes + r0, mm, 0u, 0u, (RansacState*)nullptr,
                         b.counts + r0);
    }
    RS_CK(hipGetLastError());
    RS_CK(hipMemcpyAsync(counts_out, b.counts, m * sizeof(uint32_t), hipMemcpyDeviceToHost, b.s));
    RS_CK(hipStreamSynchronize(b.s));
  }
done:
  b.release();
  return rc;
}

int cilhip_plane_fit3f(int device, const float* xyz, size_t n, int mem, float plane_out[4]) {
  if (!plane_out || (!xyz && n) || n > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  int rc = CILHIP_OK;
  Buffers b;
  RansacState hs;
  std::memset(&hs, 0, sizeof hs);
  for (int d = 0; d < 4; ++d) hs.best[d] = NAN;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    RS_CK(hipSetDevice(device));
    if (n >= 2) {
      RS_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
      RS_CK(hipMalloc(&b.st, sizeof(RansacState)));
      RS_CK(hipMemcpyAsync(b.st, &hs, sizeof hs, hipMemcpyHostToDevice, b.s));
      if (mem == CILHIP_MEM_DEVICE) {
        b.xyz = const_cast<float*>(xyz);
      } else {
        b.own_xyz = true;
        RS_CK(hipMalloc(&b.xyz, 3 * n * sizeof(float)));
        RS_CK(hipMemcpyAsync(b.xyz, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, b.s));
      }
      RS_CK(hipMalloc(&b.dpartial, (size_t)RS_MAX_BLOCKS * 8 * sizeof(double)));
      const int mb = (int)std::min<size_t>((n + RS_THREADS - 1) / RS_THREADS, RS_MAX_BLOCKS);
      hipLaunchKernelGGL(k_moments<0>, dim3(mb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, 0.0f, 1, b.st, b.dpartial);
      hipLaunchKernelGGL(k_moments_finish<0>, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
      hipLaunchKernelGGL(k_moments<1>, dim3(mb), dim3(RS_THREADS), 0, b.s, b.xyz, (uint32_t)n, 0.0f, 1, b.st, b.dpartial);
      hipLaunchKernelGGL(k_moments_finish<1>, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
      RS_CK(hipGetLastError());
      RS_CK(hipMemcpyAsync(&hs, b.st, sizeof hs, hipMemcpyDeviceToHost, b.s));
      RS_CK(hipStreamSynchronize(b.s));
    }
  }
done:
  for (int d = 0; d < 4; ++d) plane_out[d] = hs.best[d];
  b.release();
  return rc;
}

}  // extern "C"
