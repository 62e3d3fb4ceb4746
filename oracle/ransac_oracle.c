/*
 * ransac_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): plain-C restatement of cilantro's
 * PlaneRANSACEstimator3f = HyperplaneRANSACEstimator<float,3>
 *   /root/reference/include/cilantro/model_estimation/ransac_hyperplane_estimator.hpp:9-86
 *   /root/reference/include/cilantro/model_estimation/ransac_base.hpp:64-131         (estimate loop)
 *   /root/reference/include/cilantro/core/principal_component_analysis.hpp:76-84      (compute_)
 *   /root/reference/include/cilantro/core/covariance.hpp:64-77, :125-141              (mean / covariance)
 *
 * The random sample of each iteration (ransac_base.hpp:83-91: partial Fisher-Yates on a persistent
 * permutation, std::mt19937 seeded from std::random_device, i.e. not reproducible by design) is an INPUT
 * here: `samples` holds 3 point indices per iteration.
 *
 * Parity: the scoring half (absDistance, the <= test, the sequential best-model replay) is exact f32
 * arithmetic and is bit-exact restated.  The model half goes through Eigen::SelfAdjointEigenSolver, and
 * Eigen is absent from this container => "parity unpinned" for the eigenvector round-off; a cyclic Jacobi
 * eigen-solver stands in for it.
 *
 * mode 0 (F32)  : every step in f32 like the reference (serial f32 sums)
 * mode 1 (MIXED): f32 per-term arithmetic, f64 accumulation and f64 eigen-solve (what the HIP path mirrors)
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"

/* cyclic Jacobi for a symmetric 3x3 (row-major).  V columns = eigenvectors, sorted by DESCENDING eigenvalue
 * (principal_component_analysis.hpp:78,83 reverse Eigen's ascending order); if det(V) < 0 the LAST column
 * is negated (:79-82). */
#define DEF_SYM_EIG3(NAME, REAL, SQRT, FABS, EPS)                                                   \
  static void NAME(const REAL Ain[9], REAL w[3], REAL V[9]) {                                       \
    REAL A[9];                                                                                      \
    for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? (REAL)1 : (REAL)0; }         \
    for (int sweep = 0; sweep < 50; ++sweep) {                                                      \
      const REAL off = FABS(A[1]) + FABS(A[2]) + FABS(A[5]);                                        \
      const REAL dia = FABS(A[0]) + FABS(A[4]) + FABS(A[8]);                                        \
      if (off <= (REAL)EPS * (REAL)0.125 * dia || off == (REAL)0) break;                            \
      for (int p = 0; p < 2; ++p)                                                                   \
        for (int q = p + 1; q < 3; ++q) {                                                           \
          const REAL apq = A[p * 3 + q];                                                            \
          if (apq == (REAL)0) continue;                                                             \
          const REAL theta = (A[q * 3 + q] - A[p * 3 + p]) / ((REAL)2 * apq);                       \
          const REAL t = (theta >= 0 ? (REAL)1 : (REAL)-1) / (FABS(theta) + SQRT(theta * theta + (REAL)1)); \
          const REAL c = (REAL)1 / SQRT(t * t + (REAL)1), s = t * c;                                \
          for (int k = 0; k < 3; ++k) { /* A <- A J */                                              \
            const REAL akp = A[k * 3 + p], akq = A[k * 3 + q];                                      \
            A[k * 3 + p] = c * akp - s * akq;                                                       \
            A[k * 3 + q] = s * akp + c * akq;                                                       \
          }                                                                                         \
          for (int k = 0; k < 3; ++k) { /* A <- J^T A */                                            \
            const REAL apk = A[p * 3 + k], aqk = A[q * 3 + k];                                      \
            A[p * 3 + k] = c * apk - s * aqk;                                                       \
            A[q * 3 + k] = s * apk + c * aqk;                                                       \
          }                                                                                         \
          A[p * 3 + q] = A[q * 3 + p] = (REAL)0;                                                    \
          for (int k = 0; k < 3; ++k) { /* V <- V J */                                              \
            const REAL vkp = V[k * 3 + p], vkq = V[k * 3 + q];                                      \
            V[k * 3 + p] = c * vkp - s * vkq;                                                       \
            V[k * 3 + q] = s * vkp + c * vkq;                                                       \
          }                                                                                         \
        }                                                                                           \
    }                                                                                               \
    w[0] = A[0]; w[1] = A[4]; w[2] = A[8];                                                          \
    for (int i = 0; i < 2; ++i)          /* selection sort, descending, stable */                   \
      for (int j = i + 1; j < 3; ++j)                                                               \
        if (w[j] > w[i]) {                                                                          \
          REAL tw = w[i]; w[i] = w[j]; w[j] = tw;                                                   \
          for (int k = 0; k < 3; ++k) { REAL tv = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = tv; } \
        }                                                                                           \
    const REAL det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) +      \
                     V[2] * (V[3] * V[7] - V[4] * V[6]);                                            \
    if (det < (REAL)0) { V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8]; }                                \
  }

DEF_SYM_EIG3(sym_eig3_f, float, sqrtf, fabsf, FLT_EPSILON)
DEF_SYM_EIG3(sym_eig3_d, double, sqrt, fabs, DBL_EPSILON)

void orc_sym_eig3(const double A[9], double w[3], double V[9]) { sym_eig3_d(A, w, V); }

/* Eigen::Hyperplane::absDistance (ransac_hyperplane_estimator.hpp:52): |n.dot(p) + offset| in f32; the
 * 3-term dot pairs as t0 + (t1 + t2) (Eigen's unrolled redux), no FMA contraction (-ffp-contract=off). */
static inline float abs_distance(const float pl[4], const float* p) {
  const float t0 = pl[0] * p[0], t1 = pl[1] * p[1], t2 = pl[2] * p[2];
  return fabsf((t0 + (t1 + t2)) + pl[3]);
}

void orc_plane_residuals(const float* pts, size_t n, const float plane[4], float* res) {
  for (size_t i = 0; i < n; ++i) res[i] = abs_distance(plane, pts + 3 * i);
}

size_t orc_plane_count_inliers(const float* pts, size_t n, const float plane[4], float thresh) {
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) k += abs_distance(plane, pts + 3 * i) <= thresh;
  return k;
}

/* the same count on all host cores (bench.py's cpu_baseline: the reference's computeResiduals is a serial Eigen expression,
 * ransac_hyperplane_estimator.hpp:47-55 -- this is what an OpenMP build of it could reach; an integer count: order-independent) */
size_t orc_plane_count_inliers_mt(const float* pts, size_t n, const float plane[4], float thresh) {
  size_t k = 0;
#pragma omp parallel for reduction(+ : k) schedule(static)
  for (size_t i = 0; i < n; ++i) k += abs_distance(plane, pts + 3 * i) <= thresh;
  return k;
}

/* estimate_params_(sample_ind, model) (ransac_hyperplane_estimator.hpp:78-85): PCA of the subset
 * (covariance.hpp:125-141 serial branch), normal = eigenvector of the smallest eigenvalue,
 * offset = -normal.dot(mean).  idx == NULL => all points 0..m-1 (:70-76). */
void orc_plane_fit(const float* pts, const uint32_t* idx, size_t m, int mode, float plane[4]) {
  if (m < 2) { /* covariance.hpp:93-96 (min_sample_size_ = 2, :182) */
    plane[0] = plane[1] = plane[2] = plane[3] = NAN;
    return;
  }
  float mean[3];
  float n3[3];
  if (mode == 0) {
    float s[3] = {0, 0, 0};
    for (size_t i = 0; i < m; ++i) {
      const float* p = pts + 3 * (size_t)(idx ? idx[i] : i);
      s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
    }
    const float inv = 1.0f / (float)m;
    for (int d = 0; d < 3; ++d) mean[d] = inv * s[d];
    float c[9] = {0};
    for (size_t i = 0; i < m; ++i) {
      const float* p = pts + 3 * (size_t)(idx ? idx[i] : i);
      const float t[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) c[a * 3 + b] += t[a] * t[b];
    }
    const float invc = 1.0f / (float)(m - 1);
    for (int i = 0; i < 9; ++i) c[i] *= invc;
    float w[3], V[9];
    sym_eig3_f(c, w, V);
    n3[0] = V[2]; n3[1] = V[5]; n3[2] = V[8];
  } else {
    double s[3] = {0, 0, 0};
    for (size_t i = 0; i < m; ++i) {
      const float* p = pts + 3 * (size_t)(idx ? idx[i] : i);
      s[0] += (double)p[0]; s[1] += (double)p[1]; s[2] += (double)p[2];
    }
    for (int d = 0; d < 3; ++d) mean[d] = (float)(s[d] / (double)m);
    double c[9] = {0};
    for (size_t i = 0; i < m; ++i) {
      const float* p = pts + 3 * (size_t)(idx ? idx[i] : i);
      const float t[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
      for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) c[a * 3 + b] += (double)(t[a] * t[b]);
    }
    c[3] = c[1]; c[6] = c[2]; c[7] = c[5];
    for (int i = 0; i < 9; ++i) c[i] /= (double)(m - 1);
    double w[3], V[9];
    sym_eig3_d(c, w, V);
    n3[0] = (float)V[2]; n3[1] = (float)V[5]; n3[2] = (float)V[8];
  }
  plane[0] = n3[0]; plane[1] = n3[1]; plane[2] = n3[2];
  const float t0 = n3[0] * mean[0], t1 = n3[1] * mean[1], t2 = n3[2] * mean[2];
  plane[3] = -(t0 + (t1 + t2));
}

/* RandomSampleConsensusBase::estimate() (ransac_base.hpp:64-131) for the plane estimator.
 * samples: 3 indices per iteration (sample_size_ = 3, ransac_hyperplane_estimator.hpp:18).
 * residuals (n) / inliers (capacity n) may be NULL.  Returns the number of iterations performed;
 * *n_inliers = model_inliers_.size(); plane = NaN when no model was ever accepted and nothing re-estimated. */
size_t orc_plane_ransac(const float* pts, size_t n, const uint32_t* samples, size_t max_iter, float thresh,
                        size_t target_inliers, int re_estimate, int mode, float plane[4], float* residuals,
                        uint32_t* inliers, size_t* n_inliers) {
  size_t sample_size = 3;
  if (n < sample_size) sample_size = n;             /* :67 */
  if (target_inliers > n) target_inliers = n;       /* :68 */
  float best[4] = {NAN, NAN, NAN, NAN};
  size_t best_cnt = 0, it = 0;
  while (it < max_iter) {
    float cur[4];
    orc_plane_fit(pts, samples + 3 * it, sample_size, mode, cur);   /* :94 */
    const size_t cnt = orc_plane_count_inliers(pts, n, cur, thresh); /* :95-101 */
    ++it;                                                            /* :103 */
    if (cnt < sample_size) continue;                                 /* :104 */
    if (cnt > best_cnt) { memcpy(best, cur, sizeof best); best_cnt = cnt; } /* :107-111 */
    if (best_cnt >= target_inliers) break;                           /* :114 */
  }
  uint32_t* inl = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  size_t k = 0;
  for (size_t i = 0; i < n && best_cnt; ++i)
    if (abs_distance(best, pts + 3 * i) <= thresh) inl[k++] = (uint32_t)i;
  if (re_estimate) {                                                 /* :118-128 */
    orc_plane_fit(pts, inl, k, mode, best);
    k = 0;
    for (size_t i = 0; i < n; ++i)
      if (abs_distance(best, pts + 3 * i) <= thresh) inl[k++] = (uint32_t)i;
  }
  memcpy(plane, best, sizeof best);
  if (residuals) orc_plane_residuals(pts, n, best, residuals);
  if (inliers) memcpy(inliers, inl, k * sizeof(uint32_t));
  *n_inliers = k;
  free(inl);
  return it;
}

/* ---- RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp) over n point PAIRS ------------------
 * computeResiduals (:90-98): residual_i = (model_params * src_i - dst_i).norm() in f32 -- T * s with Eigen's unrolled 3-term
 * pairing (x L0 + (y L1 + z L2)) + t (the pinned transform expression of the whole engine), the difference, squaredNorm as
 * e0*e0 + (e1*e1 + e2*e2), sqrtf; no FMA contraction (-ffp-contract=off).  T: col-major 4x4. */
static inline float tf_residual(const float T[16], const float* s, const float* d) {
  const float qx = (T[0] * s[0] + (T[4] * s[1] + T[8] * s[2])) + T[12];
  const float qy = (T[1] * s[0] + (T[5] * s[1] + T[9] * s[2])) + T[13];
  const float qz = (T[2] * s[0] + (T[6] * s[1] + T[10] * s[2])) + T[14];
  const float e0 = qx - d[0], e1 = qy - d[1], e2 = qz - d[2];
  return sqrtf(e0 * e0 + (e1 * e1 + e2 * e2));
}

void orc_transform_residuals(const float* dst, const float* src, size_t n, const float T[16], float* res) {
  for (size_t i = 0; i < n; ++i) res[i] = tf_residual(T, src + 3 * i, dst + 3 * i);
}

size_t orc_transform_count_inliers(const float* dst, const float* src, size_t n, const float T[16], float thresh) {
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) k += tf_residual(T, src + 3 * i, dst + 3 * i) <= thresh;
  return k;
}

/* estimateModel(sample_ind, .) (:75-83) / estimateModel() (:61-64): estimateTransformPointToPointMetric of the selected pairs
 * (registration/transform_estimation.hpp:11-48, restated in estimator_impl.inc).  idx == NULL: all pairs 0..m-1. */
void orc_transform_fit(const float* dst, const float* src, const uint32_t* idx, size_t m, int mode, float T[16]) {
  int64_t* ii = (int64_t*)malloc((m ? m : 1) * sizeof(int64_t));
  for (size_t k = 0; k < m; ++k) ii[k] = idx ? (int64_t)idx[k] : (int64_t)k;
  orc_estimate_p2p(dst, src, ii, ii, m, mode, T, NULL);
  free(ii);
}

/* RandomSampleConsensusBase::estimate() (ransac_base.hpp:64-131) for the rigid transform estimator: sample size 3
 * (MinSampleSize = Dim, ransac_transform_estimator.hpp:20-23).  Returns the iterations performed; *have_model bit 0: some
 * hypothesis was accepted, bit 1: re-estimated.  No accepted model and no re-estimation: identity, no inliers (the
 * reference's model_params_ is then an uninitialised Eigen transform). */
size_t orc_transform_ransac(const float* dst, const float* src, size_t n, const uint32_t* samples, size_t max_iter, float thresh,
                            size_t target_inliers, int re_estimate, int mode, float T[16], float* residuals, uint32_t* inliers,
                            size_t* n_inliers, int* have_model) {
  size_t sample_size = 3;
  if (n < sample_size) sample_size = n;             /* :67 */
  if (target_inliers > n) target_inliers = n;       /* :68 */
  float best[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  size_t best_cnt = 0, it = 0;
  int have = 0;
  while (it < max_iter) {
    float cur[16];
    orc_transform_fit(dst, src, samples + 3 * it, sample_size, mode, cur);       /* :94 */
    const size_t cnt = orc_transform_count_inliers(dst, src, n, cur, thresh);     /* :95-101 */
    ++it;                                                                         /* :103 */
    if (cnt < sample_size) continue;                                              /* :104 */
    if (cnt > best_cnt) { memcpy(best, cur, sizeof best); best_cnt = cnt; have = 1; } /* :107-111 */
    if (best_cnt >= target_inliers) break;                                        /* :114 */
  }
  uint32_t* inl = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  size_t k = 0;
  for (size_t i = 0; i < n && have; ++i)
    if (tf_residual(best, src + 3 * i, dst + 3 * i) <= thresh) inl[k++] = (uint32_t)i;
  if (re_estimate) {                                                              /* :118-128 */
    orc_transform_fit(dst, src, inl, k, mode, best);
    have |= 2;
    k = 0;
    for (size_t i = 0; i < n; ++i)
      if (tf_residual(best, src + 3 * i, dst + 3 * i) <= thresh) inl[k++] = (uint32_t)i;
  }
  memcpy(T, best, sizeof best);
  if (residuals) orc_transform_residuals(dst, src, n, best, residuals);
  if (inliers) memcpy(inliers, inl, k * sizeof(uint32_t));
  if (n_inliers) *n_inliers = k;
  if (have_model) *have_model = have;
  free(inl);
  return it;
}
