// cilantro_hip/normal_estimation.hpp -- C++ host-side mirrors of the k-NN side of cilantro's KDTree3f and of
// NormalEstimation3f, header-only on top of the C ABI (c_api.h):
//
//   KDTree3f              core/kd_tree.hpp:144-388     kNNSearch :216-256, radiusSearch :251-282, kNNInRadiusSearch :286-318
//   NormalEstimation3f    core/normal_estimation.hpp   get/estimate Normals[AndCurvature]KNN[InRadius] :72-221
//
// Same method names, argument meaning and defaults as the reference -- including its asymmetry: KDTree's radii are
// SQUARED distances (kd_tree.hpp:286-318), NormalEstimation takes a plain radius and squares it itself
// (normal_estimation.hpp:126, :174).  Clouds are non-owning (pointer, count) views.
// No CPU fallback: a failing C-ABI call throws.
#pragma once

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "c_api.h"
#include "icp.hpp"

namespace cilantro_hip {

// core/nearest_neighbors.hpp:7-47
struct Neighbor {
  size_t index;
  float value;
};
typedef std::vector<Neighbor> Neighborhood;
typedef std::vector<Neighborhood> NeighborhoodSet;

class KDTree3f {
public:
  explicit KDTree3f(const ConstPointsView& points, int device = 0) : points_(points), device_(device) {}

  // kd_tree.hpp:233-240 / :303-311: one Neighborhood per query, ascending distance
  NeighborhoodSet kNNSearch(const ConstPointsView& queries, size_t k) const { return search(queries, k, std::numeric_limits<float>::infinity()); }
  NeighborhoodSet kNNInRadiusSearch(const ConstPointsView& queries, size_t k, float radius) const { return search(queries, k, radius); }
  // kd_tree.hpp:266-282: every neighbour with squared distance < radius, ascending distance (equal distances by index)
  NeighborhoodSet radiusSearch(const ConstPointsView& queries, float radius) const {
    const size_t nq = queries.cols();
    std::vector<uint64_t> off(nq + 1, 0);
    size_t total = 0;
    int rc = cilhip_radius_search3f(device_, points_.data(), points_.cols(), queries.data(), nq, CILHIP_MEM_HOST, radius, off.data(), nullptr, nullptr, 0, &total);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_radius_search3f failed (rc " + std::to_string(rc) + ")");
    std::vector<uint32_t> idx(total ? total : 1);
    std::vector<float> d2(total ? total : 1);
    if (total) {
      rc = cilhip_radius_search3f(device_, points_.data(), points_.cols(), queries.data(), nq, CILHIP_MEM_HOST, radius, off.data(), idx.data(), d2.data(), total, &total);
      if (rc != CILHIP_OK) throw std::runtime_error("cilhip_radius_search3f failed (rc " + std::to_string(rc) + ")");
    }
    NeighborhoodSet out(nq);
    for (size_t i = 0; i < nq; ++i) {
      out[i].resize((size_t)(off[i + 1] - off[i]));
      for (size_t j = 0; j < out[i].size(); ++j) out[i][j] = Neighbor{(size_t)idx[off[i] + j], d2[off[i] + j]};
    }
    return out;
  }
  const ConstPointsView& getPointsMatrixMap() const { return points_; }

private:
  NeighborhoodSet search(const ConstPointsView& queries, size_t k, float radius) const {
    const size_t nq = queries.cols();
    std::vector<uint32_t> idx(nq * k ? nq * k : 1), cnt(nq ? nq : 1);
    std::vector<float> d2(nq * k ? nq * k : 1);
    const int rc = cilhip_knn3f(device_, points_.data(), points_.cols(), queries.data(), nq, CILHIP_MEM_HOST, k, radius, idx.data(), d2.data(),
                                cnt.data());
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_knn3f failed (rc " + std::to_string(rc) + ")");
    NeighborhoodSet out(nq);
    for (size_t i = 0; i < nq; ++i) {
      out[i].resize(cnt[i]);
      for (size_t j = 0; j < cnt[i]; ++j) out[i][j] = Neighbor{(size_t)idx[i * k + j], d2[i * k + j]};
    }
    return out;
  }
  ConstPointsView points_;
  int device_;
};

class NormalEstimation3f {
public:
  explicit NormalEstimation3f(const ConstPointsView& points, int device = 0) : points_(points), device_(device) {
    const float nan = std::numeric_limits<float>::quiet_NaN();   // normal_estimation.hpp:24: no view point by default
    view_point_[0] = view_point_[1] = view_point_[2] = nan;
  }

  const float* getViewPoint() const { return view_point_; }
  NormalEstimation3f& setViewPoint(const float vp[3]) { for (int i = 0; i < 3; ++i) view_point_[i] = vp[i]; return *this; }
  NormalEstimation3f& setViewPoint(float x, float y, float z) { view_point_[0] = x; view_point_[1] = y; view_point_[2] = z; return *this; }

  // normals: packed xyz per point (VectorSet<float,3>), curvature: one float per point
  const NormalEstimation3f& getNormalsAndCurvatureKNN(std::vector<float>& normals, std::vector<float>& curvature, size_t k) const {
    run(normals, &curvature, k, std::numeric_limits<float>::infinity());
    return *this;
  }
  std::vector<float> getNormalsKNN(size_t k) const { std::vector<float> n; run(n, nullptr, k, std::numeric_limits<float>::infinity()); return n; }
  std::vector<float> getCurvatureKNN(size_t k) const { std::vector<float> n, c; run(n, &c, k, std::numeric_limits<float>::infinity()); return c; }
  const NormalEstimation3f& getNormalsAndCurvatureKNNInRadius(std::vector<float>& normals, std::vector<float>& curvature, size_t k, float radius) const {
    run(normals, &curvature, k, radius * radius);
    return *this;
  }
  std::vector<float> getNormalsKNNInRadius(size_t k, float radius) const { std::vector<float> n; run(n, nullptr, k, radius * radius); return n; }
  std::vector<float> getCurvatureKNNInRadius(size_t k, float radius) const { std::vector<float> n, c; run(n, &c, k, radius * radius); return c; }
  // :120-162: every point inside the radius takes part (unbounded neighbourhood; moments accumulated without listing it)
  const NormalEstimation3f& getNormalsAndCurvatureRadius(std::vector<float>& normals, std::vector<float>& curvature, float radius) const {
    run_radius(normals, &curvature, radius * radius);
    return *this;
  }
  std::vector<float> getNormalsRadius(float radius) const { std::vector<float> n; run_radius(n, nullptr, radius * radius); return n; }
  std::vector<float> getCurvatureRadius(float radius) const { std::vector<float> n, c; run_radius(n, &c, radius * radius); return c; }

private:
  void run(std::vector<float>& normals, std::vector<float>* curvature, size_t k, float radius) const {
    const size_t n = points_.cols();
    normals.assign(3 * n, 0.0f);
    if (curvature) curvature->assign(n, 0.0f);
    const int rc = cilhip_normals_knn3f(device_, points_.data(), n, CILHIP_MEM_HOST, k, radius, view_point_, normals.data(),
                                        curvature ? curvature->data() : nullptr);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_normals_knn3f failed (rc " + std::to_string(rc) + ")");
  }
  void run_radius(std::vector<float>& normals, std::vector<float>* curvature, float radius_sq) const {
    const size_t n = points_.cols();
    normals.assign(3 * n, 0.0f);
    if (curvature) curvature->assign(n, 0.0f);
    const int rc = cilhip_normals_radius3f(device_, points_.data(), n, CILHIP_MEM_HOST, radius_sq, view_point_, normals.data(),
                                           curvature ? curvature->data() : nullptr);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_normals_radius3f failed (rc " + std::to_string(rc) + ")");
  }
  ConstPointsView points_;
  int device_;
  float view_point_[3];
};

}  // namespace cilantro_hip
