/*
 * normals_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): plain-C restatement of cilantro's k-NN based
 * NormalEstimation<float,3>
 *   /root/reference/include/cilantro/core/normal_estimation.hpp:294-420 (compute_normals[_curvature][_view_point]_)
 *   /root/reference/include/cilantro/core/covariance.hpp:140-170          (mean / covariance of a neighbourhood, serial)
 * on top of the oracle's kd-tree (icp_oracle.c: nanoflann 1.7.1 restatement, pinned against the reference's own
 * nanoflann), i.e. KDTree::kNNSearch / kNNInRadiusSearch (core/kd_tree.hpp:216-256,:286-318).
 *
 * The neighbour sets are exact; the eigen-decomposition goes through Eigen::SelfAdjointEigenSolver in the reference
 * (Eigen absent here => "parity unpinned" for its round-off and for the SIGN of the eigenvector when no view point
 * is set); the cyclic Jacobi solver of ransac_oracle.c stands in.
 *
 * mode 0: f32 sums exactly as the reference (serial, neighbour order = ascending distance), f64 eigen-solve of the
 *         f32 covariance;  mode 1: f32 per-term arithmetic, f64 accumulation (what the HIP path mirrors).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "icp_oracle.h"

/* k-NN of every query: idx [nq*k] (-1 padded), d2 [nq*k] (+inf padded), cnt [nq] */
void orc_knn_batch(const orc_kdtree* t, const float* q, size_t nq, size_t k, float radius_sq, int64_t* idx, float* d2, uint32_t* cnt) {
#pragma omp parallel for schedule(dynamic, 256)
  for (size_t i = 0; i < nq; ++i) {
    uint64_t ii[64];
    float dd[64];
    const size_t m = orc_kdtree_knn_in_radius(t, q + 3 * i, k, radius_sq, ii, dd);
    for (size_t j = 0; j < k; ++j) {
      idx[i * k + j] = j < m ? (int64_t)ii[j] : -1;
      d2[i * k + j] = j < m ? dd[j] : INFINITY;
    }
    if (cnt) cnt[i] = (uint32_t)m;
  }
}

void orc_normals_knn(const float* pts, size_t n, size_t k, float radius_sq, const float* view_point, int mode, float* normals,
                     float* curvature) {
  orc_kdtree* t = orc_kdtree_build(pts, n, 10);   /* normal_estimation.hpp:19-21: own tree, max_leaf_size 10 */
  const int use_vp = view_point && isfinite(view_point[0]) && isfinite(view_point[1]) && isfinite(view_point[2]);   /* :366 */
#pragma omp parallel for schedule(dynamic, 256)
  for (size_t i = 0; i < n; ++i) {
    uint64_t ii[64];
    float dd[64];
    const size_t m = orc_kdtree_knn_in_radius(t, pts + 3 * i, k, radius_sq, ii, dd);
    float nrm[3] = {NAN, NAN, NAN}, curv = NAN;
    if (m >= 3) {   /* setMinValidSampleSize(points_.rows()) :28; covariance.hpp:150-154 */
      float mean[3];
      double C[9];
      if (mode == 0) {
        float s[3] = {0, 0, 0};
        for (size_t j = 0; j < m; ++j) for (int d = 0; d < 3; ++d) s[d] += pts[3 * ii[j] + d];
        const float inv = 1.0f / (float)m;
        for (int d = 0; d < 3; ++d) mean[d] = inv * s[d];
        float c[9] = {0};
        for (size_t j = 0; j < m; ++j) {
          const float tt[3] = {pts[3 * ii[j]] - mean[0], pts[3 * ii[j] + 1] - mean[1], pts[3 * ii[j] + 2] - mean[2]};
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c[a * 3 + b] += tt[a] * tt[b];
        }
        const float invc = 1.0f / (float)(m - 1);
        for (int a = 0; a < 9; ++a) C[a] = (double)(c[a] * invc);
      } else {
        double s[3] = {0, 0, 0};
        for (size_t j = 0; j < m; ++j) for (int d = 0; d < 3; ++d) s[d] += (double)pts[3 * ii[j] + d];
        for (int d = 0; d < 3; ++d) mean[d] = (float)(s[d] / (double)m);
        double c[9] = {0};
        for (size_t j = 0; j < m; ++j) {
          const float tt[3] = {pts[3 * ii[j]] - mean[0], pts[3 * ii[j] + 1] - mean[1], pts[3 * ii[j] + 2] - mean[2]};
          for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) c[a * 3 + b] += (double)(tt[a] * tt[b]);
        }
        c[3] = c[1]; c[6] = c[2]; c[7] = c[5];
        for (int a = 0; a < 9; ++a) C[a] = c[a] / (double)(m - 1);
      }
      double w[3], V[9];
      orc_sym_eig3(C, w, V);                       /* descending; smallest eigenvalue last */
      nrm[0] = (float)V[2]; nrm[1] = (float)V[5]; nrm[2] = (float)V[8];
      if (use_vp) {                                 /* :326-330 */
        const float* p = pts + 3 * i;
        const float t0 = nrm[0] * (view_point[0] - p[0]), t1 = nrm[1] * (view_point[1] - p[1]), t2 = nrm[2] * (view_point[2] - p[2]);
        if (t0 + (t1 + t2) < 0.0f) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
      }
      curv = (float)(w[2] / ((w[0] + w[1]) + w[2]));   /* :388 */
    }
    for (int d = 0; d < 3; ++d) normals[3 * i + d] = nrm[d];
    if (curvature) curvature[i] = curv;
  }
  orc_kdtree_free(t);
}

/* Radius neighbourhoods (normal_estimation.hpp:120-162 -> KDTree::radiusSearch, nanoflann RadiusResultSet: d2 < r2 strict,
 * sorted ascending).  The restated kd-tree serves them through its k-NN-in-radius search with k = n. */
void orc_normals_radius(const float* pts, size_t n, float radius_sq, const float* view_point, int mode, float* normals, float* curvature) {
  orc_kdtree* t = orc_kdtree_build(pts, n, 10);
  const int use_vp = view_point && isfinite(view_point[0]) && isfinite(view_point[1]) && isfinite(view_point[2]);
#pragma omp parallel
  {
    uint64_t* ii = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
    float* dd = (float*)malloc((n ? n : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) {
      const size_t m = orc_kdtree_knn_in_radius(t, pts + 3 * i, n, radius_sq, ii, dd);
      float nrm[3] = {NAN, NAN, NAN}, curv = NAN;
      if (m >= 3) {
        float mean[3];
        double C[9];
        if (mode == 0) {
          float s[3] = {0, 0, 0};
          for (size_t j = 0; j < m; ++j) for (int d = 0; d < 3; ++d) s[d] += pts[3 * ii[j] + d];
          const float inv = 1.0f / (float)m;
          for (int d = 0; d < 3; ++d) mean[d] = inv * s[d];
          float c[9] = {0};
          for (size_t j = 0; j < m; ++j) {
            const float tt[3] = {pts[3 * ii[j]] - mean[0], pts[3 * ii[j] + 1] - mean[1], pts[3 * ii[j] + 2] - mean[2]};
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) c[a * 3 + b] += tt[a] * tt[b];
          }
          const float invc = 1.0f / (float)(m - 1);
          for (int a = 0; a < 9; ++a) C[a] = (double)(c[a] * invc);
        } else {
          double s[3] = {0, 0, 0};
          for (size_t j = 0; j < m; ++j) for (int d = 0; d < 3; ++d) s[d] += (double)pts[3 * ii[j] + d];
          for (int d = 0; d < 3; ++d) mean[d] = (float)(s[d] / (double)m);
          double c[9] = {0};
          for (size_t j = 0; j < m; ++j) {
            const float tt[3] = {pts[3 * ii[j]] - mean[0], pts[3 * ii[j] + 1] - mean[1], pts[3 * ii[j] + 2] - mean[2]};
            for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) c[a * 3 + b] += (double)(tt[a] * tt[b]);
          }
          c[3] = c[1]; c[6] = c[2]; c[7] = c[5];
          for (int a = 0; a < 9; ++a) C[a] = c[a] / (double)(m - 1);
        }
        double w[3], V[9];
        orc_sym_eig3(C, w, V);
        nrm[0] = (float)V[2]; nrm[1] = (float)V[5]; nrm[2] = (float)V[8];
        if (use_vp) {
          const float* p = pts + 3 * i;
          const float t0 = nrm[0] * (view_point[0] - p[0]), t1 = nrm[1] * (view_point[1] - p[1]), t2 = nrm[2] * (view_point[2] - p[2]);
          if (t0 + (t1 + t2) < 0.0f) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
        }
        curv = (float)(w[2] / ((w[0] + w[1]) + w[2]));
      }
      for (int d = 0; d < 3; ++d) normals[3 * i + d] = nrm[d];
      if (curvature) curvature[i] = curv;
    }
    free(ii); free(dd);
  }
  orc_kdtree_free(t);
}

/* KDTree::radiusSearch (core/kd_tree.hpp:251-282) by exhaustive search: every point with d2 < radius_sq (strict),
 * ascending by (d2, index).  offsets[nq+1] always; idx / d2 written when cap >= total.  Returns the total. */
typedef struct { float d2; int64_t idx; } orc_rs_item;
static int orc_rs_cmp(const void* a, const void* b) {
  const orc_rs_item* x = (const orc_rs_item*)a; const orc_rs_item* y = (const orc_rs_item*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}
size_t orc_radius_search(const float* pts, size_t n, const float* q, size_t nq, float radius_sq, uint64_t* offsets, int64_t* idx,
                         float* d2, size_t cap) {
  offsets[0] = 0;
  for (size_t i = 0; i < nq; ++i) {
    size_t c = 0;
    for (size_t j = 0; j < n; ++j) {
      const float dx = q[3 * i] - pts[3 * j], dy = q[3 * i + 1] - pts[3 * j + 1], dz = q[3 * i + 2] - pts[3 * j + 2];
      const float v = ((dx * dx) + (dy * dy)) + (dz * dz);
      if (v < radius_sq) ++c;
    }
    offsets[i + 1] = offsets[i] + c;
  }
  const size_t total = (size_t)offsets[nq];
  if (!idx || cap < total) return total;
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < nq; ++i) {
    const size_t c = (size_t)(offsets[i + 1] - offsets[i]);
    if (!c) continue;
    orc_rs_item* it = (orc_rs_item*)malloc(c * sizeof(orc_rs_item));
    size_t k = 0;
    for (size_t j = 0; j < n; ++j) {
      const float dx = q[3 * i] - pts[3 * j], dy = q[3 * i + 1] - pts[3 * j + 1], dz = q[3 * i + 2] - pts[3 * j + 2];
      const float v = ((dx * dx) + (dy * dy)) + (dz * dz);
      if (v < radius_sq) { it[k].d2 = v; it[k].idx = (int64_t)j; ++k; }
    }
    qsort(it, c, sizeof(orc_rs_item), orc_rs_cmp);
    for (k = 0; k < c; ++k) { idx[offsets[i] + k] = it[k].idx; d2[offsets[i] + k] = it[k].d2; }
    free(it);
  }
  return total;
}
