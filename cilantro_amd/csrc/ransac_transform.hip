// ransac_transform.hip -- RigidTransformRANSACEstimator3f on the device.  Replaces cilantro's
//   model_estimation/ransac_transform_estimator.hpp:61-104   (estimateModel, computeResiduals)
//   model_estimation/ransac_base.hpp:64-131                   (estimate loop)
//   registration/transform_estimation.hpp:11-48               (estimateTransformPointToPointMetric: Kabsch without scale)
// for n point PAIRS (dst_i, src_i) -- the reference's constructors gather them from correspondences or index lists
// (:34-59); the gather is the caller's here.
//
// Same shape as ransac.hip (PlaneRANSACEstimator3f): the reference fits and scores ONE hypothesis per pass over the pairs
// and keeps a residual vector + inlier list per iteration; here one lane fits each hypothesis (closed-form moments of its 3
// pairs -> 3x3 SVD, solve.hpp), hypotheses are scored TR_ROUND at a time per pass (a lane keeps 4 pairs in registers, the
// round's transforms come through the scalar cache, only inlier COUNTS leave the kernel), and a one-lane kernel replays the
// reference's sequential "strictly better / target reached" decisions, so the outcome is the sequential loop's.  The
// re-estimation over the best model's inliers is one streaming pass of 16 f64 moments in a fixed order.
//
// Numeric contract: residual = |T * s - d| formed as the reference forms it in f32 -- T * s with the pinned pairing of
// transform_point (solve.hpp), the difference, squaredNorm as d0*d0 + (d1*d1 + d2*d2), sqrt -- no FMA contraction.  The
// inlier test `sqrt(x) <= thr` is evaluated as `x <= X`, X = the largest f32 whose correctly rounded square root is <= thr
// (computed on the host: sqrt is monotonic, so the two tests select the same pairs bit for bit).  Model fit: raw f64
// moments (products of f32 coordinates are exact in f64), f64 SVD -- Eigen's JacobiSVD<Matrix3f> round-off is "parity
// unpinned" (Eigen is absent from the build container), as for every estimator of this engine.
#include "../../include/cilantro_hip/c_api.h"
#include "solve.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int TR_THREADS = 256;
constexpr int TR_PAIRS = 4;       // pairs per lane held in registers
constexpr int TR_ROUND = 64;      // hypotheses scored per pass over the pairs
constexpr int TR_MAX_BLOCKS = 1024;

struct TState {
  float best[12];       // row-major L (9), t (3)
  unsigned int best_cnt;
  unsigned int iterations;
  int done;
  int have_model;
  unsigned int n_inliers;
  unsigned int pad;
};

__device__ __forceinline__ float sq_residual(const float* M, float sx, float sy, float sz, float dx, float dy, float dz) {
  const float qx = __fadd_rn(__fadd_rn(__fmul_rn(M[0], sx), __fadd_rn(__fmul_rn(M[1], sy), __fmul_rn(M[2], sz))), M[9]);
  const float qy = __fadd_rn(__fadd_rn(__fmul_rn(M[3], sx), __fadd_rn(__fmul_rn(M[4], sy), __fmul_rn(M[5], sz))), M[10]);
  const float qz = __fadd_rn(__fadd_rn(__fmul_rn(M[6], sx), __fadd_rn(__fmul_rn(M[7], sy), __fmul_rn(M[8], sz))), M[11]);
  const float e0 = __fsub_rn(qx, dx), e1 = __fsub_rn(qy, dy), e2 = __fsub_rn(qz, dz);
  return __fadd_rn(__fmul_rn(e0, e0), __fadd_rn(__fmul_rn(e1, e1), __fmul_rn(e2, e2)));
}

// rigid transform (row-major L, t) from the 16 raw moments n, sum d, sum s, sum d s^T; identity when n == 0
// (transform_estimation.hpp:20-23)
__device__ void model_from_sums(const double sums[16], float M[12]) {
  double L[9], t[3];
  cilhip::kabsch_from_sums(sums, L, t);
  for (int i = 0; i < 9; ++i) M[i] = (float)L[i];
  for (int i = 0; i < 3; ++i) M[9 + i] = (float)t[i];
}

// one lane per hypothesis: Kabsch of its sample (ransac_base.hpp:94 -> estimateModel(sample_ind, .), :75-83)
__global__ void k_tmodels(const float* __restrict__ dst, const float* __restrict__ src, const uint32_t* __restrict__ samples, uint32_t sample_size,
                          uint32_t m_total, uint32_t m_padded, float* __restrict__ models) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= m_padded) return;
  float M[12];
  if (h >= m_total) {
    for (int i = 0; i < 12; ++i) M[i] = NAN;      // padding: never an inlier
  } else {
    double sums[16];
    for (int i = 0; i < 16; ++i) sums[i] = 0.0;
    for (uint32_t i = 0; i < sample_size; ++i) {
      const size_t idx = samples[3 * (size_t)h + i];
      const double p[3] = {(double)dst[3 * idx], (double)dst[3 * idx + 1], (double)dst[3 * idx + 2]};
      const double q[3] = {(double)src[3 * idx], (double)src[3 * idx + 1], (double)src[3 * idx + 2]};
      sums[0] += 1.0;
      for (int c = 0; c < 3; ++c) { sums[1 + c] += p[c]; sums[4 + c] += q[c]; }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sums[7 + r * 3 + c] += p[r] * q[c];
    }
    model_from_sums(sums, M);
  }
  for (int i = 0; i < 12; ++i) models[12 * (size_t)h + i] = M[i];
}

// inlier counts of `m` (<= TR_ROUND) transforms over all pairs -> partial[block][TR_ROUND]
__global__ __launch_bounds__(TR_THREADS) void k_tscore(const float* __restrict__ dst, const float* __restrict__ src, uint32_t n,
                                                       const float* __restrict__ models, uint32_t m, float thr_sq, uint32_t* __restrict__ partial,
                                                       const TState* __restrict__ st) {
  if (st && st->done) return;
  __shared__ uint32_t cnt[TR_ROUND];
  for (int t = threadIdx.x; t < TR_ROUND; t += TR_THREADS) cnt[t] = 0;
  __syncthreads();
  const bool lane0 = (threadIdx.x & 63) == 0;
  constexpr uint32_t TILE = TR_THREADS * TR_PAIRS;
  for (size_t base = (size_t)blockIdx.x * TILE; base < n; base += (size_t)gridDim.x * TILE) {
    float s[TR_PAIRS][3], d[TR_PAIRS][3];
#pragma unroll
    for (int k = 0; k < TR_PAIRS; ++k) {
      const size_t i = base + (size_t)k * TR_THREADS + threadIdx.x;
      const bool v = i < n;      // out of range: NaN coordinates, NaN <= thr is false
#pragma unroll
      for (int c = 0; c < 3; ++c) { s[k][c] = v ? src[3 * i + c] : NAN; d[k][c] = v ? dst[3 * i + c] : NAN; }
    }
    for (uint32_t j = 0; j < m; ++j) {
      float M[12];
#pragma unroll
      for (int t = 0; t < 12; ++t) M[t] = models[12 * j + t];      // wave-uniform: scalar loads
      uint32_t tot = 0;
#pragma unroll
      for (int k = 0; k < TR_PAIRS; ++k)
        tot += (uint32_t)__popcll(__ballot(sq_residual(M, s[k][0], s[k][1], s[k][2], d[k][0], d[k][1], d[k][2]) <= thr_sq));
      if (lane0 && tot) atomicAdd(&cnt[j], tot);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < TR_ROUND; t += TR_THREADS) partial[(size_t)blockIdx.x * TR_ROUND + t] = cnt[t];
}

// sums the per-block counts and replays ransac_base.hpp:103-114 over this round's hypotheses, in order
__global__ void k_tpick(const uint32_t* __restrict__ partial, int nblocks, const float* __restrict__ models, uint32_t m, uint32_t sample_size,
                        uint32_t target, TState* st, uint32_t* counts_out) {
  __shared__ uint32_t cnt[TR_ROUND];
  if (st && st->done) return;
  const int t = threadIdx.x;
  uint32_t c = 0;
  for (int g = 0; g < nblocks; ++g) c += partial[(size_t)g * TR_ROUND + t];
  cnt[t] = c;
  if (counts_out && (uint32_t)t < m) counts_out[t] = c;
  __syncthreads();
  if (t == 0 && st) {
    for (uint32_t h = 0; h < m; ++h) {
      st->iterations++;                                     // :103
      if (cnt[h] < sample_size) continue;                   // :104
      if (cnt[h] > st->best_cnt) {                          // :107-111
        for (int i = 0; i < 12; ++i) st->best[i] = models[12 * h + i];
        st->best_cnt = cnt[h];
        st->have_model = 1;
      }
      if (st->best_cnt >= target) { st->done = 1; break; }  // :114
    }
  }
}

// the 16 raw moments over the inliers of st->best (all_pairs != 0: over every pair -- estimateModel() without a subset)
__global__ __launch_bounds__(TR_THREADS) void k_tmoments(const float* __restrict__ dst, const float* __restrict__ src, uint32_t n, float thr_sq,
                                                         int all_pairs, const TState* __restrict__ st, double* __restrict__ partial) {
  float M[12];
  for (int i = 0; i < 12; ++i) M[i] = st->best[i];
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  const bool none = !all_pairs && !st->have_model;      // no hypothesis was ever accepted: model_inliers_ is empty
  for (size_t i = (size_t)blockIdx.x * TR_THREADS + threadIdx.x; i < n && !none; i += (size_t)gridDim.x * TR_THREADS) {
    const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2], dx = dst[3 * i], dy = dst[3 * i + 1], dz = dst[3 * i + 2];
    if (!all_pairs && !(sq_residual(M, sx, sy, sz, dx, dy, dz) <= thr_sq)) continue;
    const double p[3] = {(double)dx, (double)dy, (double)dz}, q[3] = {(double)sx, (double)sy, (double)sz};
    acc[0] += 1.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc[1 + c] += p[c]; acc[4 + c] += q[c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[7 + r * 3 + c] = fma(p[r], q[c], acc[7 + r * 3 + c]);
  }
  __shared__ double red[TR_THREADS / 64][16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off, 64);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 16; ++k) red[threadIdx.x >> 6][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 16) partial[(size_t)blockIdx.x * 16 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void k_tmoments_finish(const double* __restrict__ partial, int nblocks, TState* st) {
  __shared__ double tot[16];
  if (threadIdx.x < 16) {
    double s = 0.0;
    for (int g = 0; g < nblocks; ++g) s += partial[(size_t)g * 16 + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) { model_from_sums(tot, st->best); st->have_model |= 2; }      // (re-estimated: residuals / inliers are formed for it)
}

// final model: per-chunk inlier counts -> exclusive scan -> residuals + ordered inlier indices
__global__ __launch_bounds__(TR_THREADS) void k_tchunk_counts(const float* __restrict__ dst, const float* __restrict__ src, uint32_t n, uint32_t chunk,
                                                              float thr_sq, const TState* __restrict__ st, uint32_t* __restrict__ counts) {
  float M[12];
  for (int i = 0; i < 12; ++i) M[i] = st->best[i];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  uint32_t c = 0;
  if (st->have_model)      // (no accepted model and no re-estimation: model_inliers_ stays empty)
  for (size_t i = lo + threadIdx.x; i < hi; i += TR_THREADS)
    c += sq_residual(M, src[3 * i], src[3 * i + 1], src[3 * i + 2], dst[3 * i], dst[3 * i + 1], dst[3 * i + 2]) <= thr_sq ? 1u : 0u;
  __shared__ uint32_t red[TR_THREADS / 64];
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void k_tscan_counts(uint32_t* counts, int nblocks, TState* st) {   // one lane; nblocks <= 1024
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t run = 0;
    for (int g = 0; g < nblocks; ++g) { const uint32_t c = counts[g]; counts[g] = run; run += c; }
    st->n_inliers = run;
  }
}

__global__ __launch_bounds__(TR_THREADS) void k_twrite_final(const float* __restrict__ dst, const float* __restrict__ src, uint32_t n, uint32_t chunk,
                                                             float thr_sq, const TState* __restrict__ st, const uint32_t* __restrict__ offsets,
                                                             float* __restrict__ residuals, uint32_t* __restrict__ inliers) {
  float M[12];
  for (int i = 0; i < 12; ++i) M[i] = st->best[i];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  __shared__ uint32_t wave_cnt[TR_THREADS / 64];
  uint32_t run = offsets ? offsets[blockIdx.x] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t b = lo; b < hi; b += TR_THREADS) {
    const size_t i = b + threadIdx.x;
    bool in = false;
    if (i < hi) {
      const float r2 = sq_residual(M, src[3 * i], src[3 * i + 1], src[3 * i + 2], dst[3 * i], dst[3 * i + 1], dst[3 * i + 2]);
      // .norm(): the correctly rounded f32 square root (HIP's __fsqrt_rn is the native, 1-ulp instruction; the f64 root of an
      // f32 value rounds to the correctly rounded f32 root -- 53 > 2 * 24 + 2 bits)
      if (residuals) residuals[i] = (float)sqrt((double)r2);
      in = st->have_model && r2 <= thr_sq;
    }
    if (inliers) {   // block-uniform
      const unsigned long long bal = __ballot(in);
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int w = 0; w < TR_THREADS / 64; ++w) { before += w < wave ? wave_cnt[w] : 0; total += wave_cnt[w]; }
      if (in) inliers[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)i;
      run += total;
      __syncthreads();
    }
  }
}

#define TR_CK(x)                          \
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
inline uint64_t bounded(uint64_t& s, uint64_t bound) { return (uint64_t)(((unsigned __int128)splitmix64(s) * bound) >> 64); }

// the largest f32 x with sqrtf(x) <= thr (sqrtf is correctly rounded and monotonic): `norm <= thr` <=> `squaredNorm <= x`
inline float sq_threshold(float thr) {
  if (!(thr >= 0.0f)) return -1.0f;                        // nothing is an inlier (NaN / negative threshold)
  if (std::isinf(thr)) return thr;
  float x = thr * thr;
  while (std::sqrt(x) > thr) x = std::nextafter(x, 0.0f);
  while (true) {
    const float up = std::nextafter(x, INFINITY);
    if (std::isinf(up) || std::sqrt(up) > thr) break;
    x = up;
  }
  return x;
}

struct TBuffers {
  float *dst = nullptr, *src = nullptr;
  bool own = false;
  uint32_t* samples = nullptr;
  float* models = nullptr;
  uint32_t* partial = nullptr;
  double* dpartial = nullptr;
  uint32_t* chunk_counts = nullptr;
  TState* st = nullptr;
  float* residuals = nullptr;
  uint32_t* inliers = nullptr;
  uint32_t* counts = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  void release() {
    if (own && dst) (void)hipFree(dst);
    if (own && src) (void)hipFree(src);
    if (samples) (void)hipFree(samples);
    if (models) (void)hipFree(models);
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
  hipError_t upload(const float* d, const float* sc, size_t n, int mem) {
    if (mem == CILHIP_MEM_DEVICE) { dst = const_cast<float*>(d); src = const_cast<float*>(sc); return hipSuccess; }
    own = true;
    hipError_t e = hipMalloc(&dst, 3 * n * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMalloc(&src, 3 * n * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, d, 3 * n * sizeof(float), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(src, sc, 3 * n * sizeof(float), hipMemcpyHostToDevice, s);
  }
};

inline int tscore_blocks(size_t n) {
  const size_t tiles = (n + (size_t)TR_THREADS * TR_PAIRS - 1) / ((size_t)TR_THREADS * TR_PAIRS);
  return (int)(tiles < 1 ? 1 : (tiles > TR_MAX_BLOCKS ? TR_MAX_BLOCKS : tiles));
}

// row-major (L, t) -> col-major 4x4
inline void pack_model(const float M[12], float T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = 0.0f;
  T[15] = 1.0f;
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[c * 4 + r] = M[r * 3 + c]; T[12 + r] = M[9 + r]; }
}

}  // namespace

extern "C" {

int cilhip_transform_ransac3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, const uint32_t* samples, uint64_t seed,
                              float max_residual, size_t target_inliers, size_t max_iter, int re_estimate, cilhip_transform_model* out,
                              float* residuals_out, uint32_t* inliers_out) {
  if (!out || ((!dst_xyz || !src_xyz) && n) || n > 0xFFFFFFF0ull || max_iter > 0x0FFFFFFFull) return CILHIP_ERR_INVALID;
  int rc = CILHIP_OK;
  TBuffers b;
  TState hs;
  std::memset(&hs, 0, sizeof hs);
  hs.best[0] = hs.best[4] = hs.best[8] = 1.0f;                    // model_params_ is default-constructed: identity
  const uint32_t sample_size = n < 3 ? (uint32_t)n : 3u;          // ransac_base.hpp:67 (MinSampleSize = Dim for rigid transforms)
  if (target_inliers > n) target_inliers = n;                     // :68
  const float thr_sq = sq_threshold(max_residual);
  float ms = 0.0f;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    TR_CK(hipSetDevice(device));
    TR_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
    TR_CK(hipEventCreate(&b.e0));
    TR_CK(hipEventCreate(&b.e1));
    TR_CK(hipMalloc(&b.st, sizeof(TState)));
    TR_CK(hipMemcpyAsync(b.st, &hs, sizeof hs, hipMemcpyHostToDevice, b.s));
    if (n > 0) {
      TR_CK(b.upload(dst_xyz, src_xyz, n, mem));
      std::vector<uint32_t> hsamp;
      if (!samples && max_iter) {      // the random samples (ransac_base.hpp:83-91): distinct indices per iteration, drawn on the host
        hsamp.resize(3 * max_iter);
        uint64_t st = seed;
        for (size_t it = 0; it < max_iter; ++it) {
          uint32_t pick[3] = {0, 0, 0};
          for (uint32_t i = 0; i < sample_size; ++i) {
            uint32_t v = (uint32_t)bounded(st, n - i);
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
      } else if (samples) {
        for (size_t i = 0; i < 3 * max_iter; ++i)
          if ((i % 3) < sample_size && samples[i] >= n) { rc = CILHIP_ERR_INVALID; goto done; }
      }
      const size_t mpad = ((max_iter ? max_iter : 1) + TR_ROUND - 1) / TR_ROUND * TR_ROUND;
      const int nb = tscore_blocks(n);
      TR_CK(hipMalloc(&b.samples, 3 * (max_iter ? max_iter : 1) * sizeof(uint32_t)));
      TR_CK(hipMalloc(&b.models, mpad * 12 * sizeof(float)));
      TR_CK(hipMalloc(&b.partial, (size_t)nb * TR_ROUND * sizeof(uint32_t)));
      TR_CK(hipMalloc(&b.dpartial, (size_t)TR_MAX_BLOCKS * 16 * sizeof(double)));
      TR_CK(hipMalloc(&b.chunk_counts, TR_MAX_BLOCKS * sizeof(uint32_t)));
      if (max_iter) TR_CK(hipMemcpy(b.samples, samples, 3 * max_iter * sizeof(uint32_t), hipMemcpyHostToDevice));      // (blocking: `samples` may be a local vector)
      TR_CK(hipEventRecord(b.e0, b.s));
      if (max_iter)
        hipLaunchKernelGGL(k_tmodels, dim3((unsigned)((mpad + 127) / 128)), dim3(128), 0, b.s, b.dst, b.src, b.samples, sample_size, (uint32_t)max_iter,
                           (uint32_t)mpad, b.models);
      for (size_t r0 = 0; r0 < max_iter; r0 += TR_ROUND) {
        const uint32_t m = (uint32_t)(max_iter - r0 < TR_ROUND ? max_iter - r0 : TR_ROUND);
        hipLaunchKernelGGL(k_tscore, dim3(nb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, (const float*)(b.models + 12 * r0), m, thr_sq,
                           b.partial, b.st);
        hipLaunchKernelGGL(k_tpick, dim3(1), dim3(TR_ROUND), 0, b.s, b.partial, nb, (const float*)(b.models + 12 * r0), m, sample_size,
                           (uint32_t)target_inliers, b.st, (uint32_t*)nullptr);
      }
      const int mb = (int)std::min<size_t>((n + TR_THREADS - 1) / TR_THREADS, TR_MAX_BLOCKS);
      const uint32_t chunk = (uint32_t)(((n + mb - 1) / mb + TR_THREADS - 1) / TR_THREADS * TR_THREADS);
      const int cb = (int)((n + chunk - 1) / chunk);
      if (re_estimate) {
        // ransac_base.hpp:118-128: the model of the best hypothesis' inliers (model_inliers_ is EMPTY when no hypothesis was
        // ever accepted: estimateTransformPointToPointMetric of nothing is the identity)
        hipLaunchKernelGGL(k_tmoments, dim3(mb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, thr_sq, 0, b.st, b.dpartial);
        hipLaunchKernelGGL(k_tmoments_finish, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
      }
      hipLaunchKernelGGL(k_tchunk_counts, dim3(cb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, chunk, thr_sq, b.st, b.chunk_counts);
      hipLaunchKernelGGL(k_tscan_counts, dim3(1), dim3(64), 0, b.s, b.chunk_counts, cb, b.st);
      if (residuals_out) TR_CK(hipMalloc(&b.residuals, n * sizeof(float)));
      if (inliers_out) TR_CK(hipMalloc(&b.inliers, n * sizeof(uint32_t)));
      if (residuals_out || inliers_out)
        hipLaunchKernelGGL(k_twrite_final, dim3(cb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, chunk, thr_sq, b.st, b.chunk_counts,
                           b.residuals, b.inliers);
      TR_CK(hipEventRecord(b.e1, b.s));
      TR_CK(hipGetLastError());
    }
    TR_CK(hipMemcpyAsync(&hs, b.st, sizeof hs, hipMemcpyDeviceToHost, b.s));
    TR_CK(hipStreamSynchronize(b.s));
    if (n > 0) {
      TR_CK(hipEventElapsedTime(&ms, b.e0, b.e1));
      if (residuals_out && b.residuals) TR_CK(hipMemcpy(residuals_out, b.residuals, n * sizeof(float), hipMemcpyDeviceToHost));
      if (inliers_out && b.inliers && hs.n_inliers)
        TR_CK(hipMemcpy(inliers_out, b.inliers, (size_t)hs.n_inliers * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
  }
done:
  pack_model(hs.best, out->T);
  out->iterations = hs.iterations;
  out->n_inliers = hs.n_inliers;
  out->have_model = hs.have_model;
  out->target_reached = hs.n_inliers >= target_inliers ? 1 : 0;   // ransac_base.hpp:172
  out->device_ms = (double)ms;
  b.release();
  return rc;
}

int cilhip_transform_score3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, const float* transforms, size_t m,
                             float max_residual, uint32_t* counts_out) {
  if (((!dst_xyz || !src_xyz) && n) || (!transforms && m) || (!counts_out && m) || n > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  if (m == 0) return CILHIP_OK;
  int rc = CILHIP_OK;
  TBuffers b;
  const float thr_sq = sq_threshold(max_residual);
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    TR_CK(hipSetDevice(device));
    TR_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
    if (n == 0) { std::memset(counts_out, 0, m * sizeof(uint32_t)); goto done; }
    TR_CK(b.upload(dst_xyz, src_xyz, n, mem));
    const size_t mpad = (m + TR_ROUND - 1) / TR_ROUND * TR_ROUND;
    const int nb = tscore_blocks(n);
    std::vector<float> hm(12 * mpad, NAN);
    for (size_t h = 0; h < m; ++h) {        // col-major 4x4 -> row-major L, t
      const float* T = transforms + 16 * h;
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) hm[12 * h + r * 3 + c] = T[c * 4 + r]; hm[12 * h + 9 + r] = T[12 + r]; }
    }
    TR_CK(hipMalloc(&b.models, mpad * 12 * sizeof(float)));
    TR_CK(hipMalloc(&b.partial, (size_t)nb * TR_ROUND * sizeof(uint32_t)));
    TR_CK(hipMalloc(&b.counts, mpad * sizeof(uint32_t)));
    TR_CK(hipMemcpyAsync(b.models, hm.data(), 12 * mpad * sizeof(float), hipMemcpyHostToDevice, b.s));
    for (size_t r0 = 0; r0 < m; r0 += TR_ROUND) {
      const uint32_t mm = (uint32_t)(m - r0 < TR_ROUND ? m - r0 : TR_ROUND);
      hipLaunchKernelGGL(k_tscore, dim3(nb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, (const float*)(b.models + 12 * r0), mm, thr_sq,
                         b.partial, (const TState*)nullptr);
      hipLaunchKernelGGL(k_tpick, dim3(1), dim3(TR_ROUND), 0, b.s, b.partial, nb, (const float*)(b.models + 12 * r0), mm, 0u, 0u, (TState*)nullptr,
                         b.counts + r0);
    }
    TR_CK(hipGetLastError());
    TR_CK(hipMemcpyAsync(counts_out, b.counts, m * sizeof(uint32_t), hipMemcpyDeviceToHost, b.s));
    TR_CK(hipStreamSynchronize(b.s));
  }
done:
  b.release();
  return rc;
}

int cilhip_transform_fit3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, float T_out[16]) {
  if (!T_out || ((!dst_xyz || !src_xyz) && n) || n > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  int rc = CILHIP_OK;
  TBuffers b;
  TState hs;
  std::memset(&hs, 0, sizeof hs);
  hs.best[0] = hs.best[4] = hs.best[8] = 1.0f;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    TR_CK(hipSetDevice(device));
    if (n > 0) {
      TR_CK(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
      TR_CK(hipMalloc(&b.st, sizeof(TState)));
      TR_CK(hipMemcpyAsync(b.st, &hs, sizeof hs, hipMemcpyHostToDevice, b.s));
      TR_CK(b.upload(dst_xyz, src_xyz, n, mem));
      TR_CK(hipMalloc(&b.dpartial, (size_t)TR_MAX_BLOCKS * 16 * sizeof(double)));
      const int mb = (int)std::min<size_t>((n + TR_THREADS - 1) / TR_THREADS, TR_MAX_BLOCKS);
      hipLaunchKernelGGL(k_tmoments, dim3(mb), dim3(TR_THREADS), 0, b.s, b.dst, b.src, (uint32_t)n, 0.0f, 1, b.st, b.dpartial);
      hipLaunchKernelGGL(k_tmoments_finish, dim3(1), dim3(64), 0, b.s, b.dpartial, mb, b.st);
      TR_CK(hipGetLastError());
      TR_CK(hipMemcpyAsync(&hs, b.st, sizeof hs, hipMemcpyDeviceToHost, b.s));
      TR_CK(hipStreamSynchronize(b.s));
    }
  }
done:
  pack_model(hs.best, T_out);
  b.release();
  return rc;
}

}  // extern "C"
