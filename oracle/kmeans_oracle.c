/*
 * kmeans_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): plain-C restatement of cilantro's
 * KMeans<float,3>::cluster_ with the brute-force assignment (use_kd_tree = false),
 * /root/reference/include/cilantro/clustering/kmeans.hpp:67-194.
 *
 * mode 0: all-f32, serial centroid sums exactly as the reference (:126-131, :179-181)
 * mode 1: identical decisions, but cluster sums in f64 and mean = (float)(sum / count)
 *         (what the HIP path's exact fixed-point accumulation rounds to)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"

/* (c - x).squaredNorm(): Eigen's unrolled 3-term redux pairing d0*d0 + (d1*d1 + d2*d2), f32, no contraction */
static inline float sqn(const float* c, const float* x) {
  const float d0 = c[0] - x[0], d1 = c[1] - x[1], d2 = c[2] - x[2];
  return d0 * d0 + (d1 * d1 + d2 * d2);
}

/* kmeans.hpp:95-119: labels[i] = argmin_j ||c_j - x_i||^2, strict '<' over ascending j. Returns #changed. */
size_t orc_kmeans_assign(const float* x, size_t n, const float* c, size_t k, int64_t* labels) {
  size_t changed = 0;
#pragma omp parallel for reduction(+ : changed) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float best = INFINITY;
    int64_t bi = 0;
    for (size_t j = 0; j < k; ++j) {
      const float d = sqn(c + 3 * j, x + 3 * i);
      if (d < best) { best = d; bi = (int64_t)j; }
    }
    if (labels[i] != bi) ++changed;
    labels[i] = bi;
  }
  return changed;
}

/* kmeans.hpp:86-94, the use_kd_tree branch: a KDTree over the centroids (core/kd_tree.hpp, leaf 10), nearestNeighborSearch per
 * point (:181-185 -> nanoflann knnSearch, k = 1: first met wins ties); the pinned kd-tree restatement of icp_oracle.c. */
size_t orc_kmeans_assign_kd(const float* x, size_t n, const float* c, size_t k, int64_t* labels) {
  orc_kdtree* t = orc_kdtree_build(c, k, 10);
  int64_t* idx = (int64_t*)malloc((n ? n : 1) * sizeof(int64_t));
  float* d2 = (float*)malloc((n ? n : 1) * sizeof(float));
  uint32_t* cnt = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_knn_batch(t, x, n, 1, INFINITY, idx, d2, cnt);
  size_t changed = 0;
  for (size_t i = 0; i < n; ++i) {
    if (labels[i] != idx[i]) ++changed;
    labels[i] = idx[i];
  }
  free(idx); free(d2); free(cnt);
  orc_kdtree_free(t);
  return changed;
}

static size_t kmeans_impl(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int use_kd, int64_t* labels);
/* kmeans.hpp:67-194.  centroids: in = initial, out = final (3*k floats). labels: n, caller-zeroed (:80 resize). */
size_t orc_kmeans(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int64_t* labels) {
  return kmeans_impl(x, n, c, k, max_iter, tol, mode, 0, labels);
}
size_t orc_kmeans_kd(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int64_t* labels) {
  return kmeans_impl(x, n, c, k, max_iter, tol, mode, 1, labels);
}
static size_t kmeans_impl(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int use_kd, int64_t* labels) {
  const float tol_sq = tol * tol;
  float* c_old = (float*)malloc(3 * k * sizeof(float));
  float* sf = (float*)malloc(3 * k * sizeof(float));
  double* sd = (double*)malloc(3 * k * sizeof(double));
  size_t* cnt = (size_t*)malloc(k * sizeof(size_t));
  size_t iter = 0;
  while (iter < max_iter) {
    const size_t changed = use_kd ? orc_kmeans_assign_kd(x, n, c, k, labels) : orc_kmeans_assign(x, n, c, k, labels);
    if (changed == 0 && iter > 0) break;                                   /* :122 */
    if (tol > 0.0f) memcpy(c_old, c, 3 * k * sizeof(float));               /* :123 */
    memset(sf, 0, 3 * k * sizeof(float)); memset(sd, 0, 3 * k * sizeof(double)); memset(cnt, 0, k * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) {                                       /* :126-131 serial */
      const size_t l = (size_t)labels[i];
      for (int d = 0; d < 3; ++d) { sf[3 * l + d] += x[3 * i + d]; sd[3 * l + d] += (double)x[3 * i + d]; }
      cnt[l]++;
    }
    for (size_t i = 0; i < k; ++i) {                                       /* :134-176 empty clusters */
      if (cnt[i] != 0) continue;
      size_t mx = 0;
      for (size_t j = 1; j < k; ++j) if (cnt[j] > cnt[mx]) mx = j;
      float oc[3];
      for (int d = 0; d < 3; ++d)
        oc[d] = mode == 0 ? sf[3 * mx + d] * (1.0f / (float)cnt[mx]) : (float)(sd[3 * mx + d] / (double)cnt[mx]);
      float md = -1.0f; size_t mi = 0;
      for (size_t j = 0; j < n; ++j)                                       /* ties: lowest index (reference: unordered critical) */
        if ((size_t)labels[j] == mx) { const float dd = sqn(oc, x + 3 * j); if (dd > md) { md = dd; mi = j; } }
      labels[mi] = (int64_t)i;
      for (int d = 0; d < 3; ++d) {
        sf[3 * mx + d] -= x[3 * mi + d]; sd[3 * mx + d] -= (double)x[3 * mi + d];
        /* NB: the reference does NOT add the moved point to cluster i's sum (:171-175), so the re-seeded
           centroid becomes 0 * (1/1) = the origin for one iteration.  Restated faithfully. */
      }
      cnt[mx]--; cnt[i]++;
    }
    for (size_t i = 0; i < k; ++i)                                         /* :179-181 */
      for (int d = 0; d < 3; ++d)
        c[3 * i + d] = mode == 0 ? sf[3 * i + d] * (1.0f / (float)cnt[i]) : (float)(sd[3 * i + d] / (double)cnt[i]);
    ++iter;
    if (tol > 0.0f) {                                                      /* :186-188 */
      float mxs = 0.0f;
      for (size_t i = 0; i < k; ++i) { const float s = sqn(c + 3 * i, c_old + 3 * i); if (s > mxs) mxs = s; }
      if (mxs < tol_sq) break;
    }
  }
  free(c_old); free(sf); free(sd); free(cnt);
  return iter;
}
