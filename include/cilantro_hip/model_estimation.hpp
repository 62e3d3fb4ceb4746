// cilantro_hip/model_estimation.hpp -- C++ host-side mirrors of the two "next tier" consumers of the
// path, header-only on top of the C ABI (c_api.h):
//
//   PlaneRANSACEstimator3f   model_estimation/ransac_hyperplane_estimator.hpp:9-86 (HyperplaneRANSACEstimator<float,3>)
//                            model_estimation/ransac_base.hpp:16-184               (RandomSampleConsensusBase, CRTP setters)
//   KMeans3f                 clustering/kmeans.hpp:14-66 (KMeans<float,3>), clustering/clustering_base.hpp
//   RigidTransformRANSACEstimator3f  model_estimation/ransac_transform_estimator.hpp ("next tier")
//
// Same method names, argument meaning and defaults as the reference; clouds are non-owning
// (pointer, count) views (ConstPointsView, icp.hpp).  No CPU fallback: a failing C-ABI call throws.
#pragma once

#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <vector>

#include "c_api.h"
#include "icp.hpp"

namespace cilantro_hip {

// Eigen::Hyperplane<float,3> stand-in: normal() . x + offset() = 0
struct Hyperplane3f {
  float coeffs_[4] = {0, 0, 0, 0};
  const float* normal() const { return coeffs_; }
  float offset() const { return coeffs_[3]; }
  const float* coeffs() const { return coeffs_; }
  float absDistance(const float* p) const {
    const float d = coeffs_[0] * p[0] + (coeffs_[1] * p[1] + coeffs_[2] * p[2]) + coeffs_[3];
    return d < 0 ? -d : d;
  }
};

class PlaneRANSACEstimator3f {
public:
  typedef Hyperplane3f Model;
  typedef float ResidualScalar;
  typedef std::vector<float> ResidualVector;
  typedef std::vector<size_t> IndexVector;

  // ransac_hyperplane_estimator.hpp:17-19: sample size 3, target = ceil(n/2), 100 iterations, 0.1, re-estimate
  explicit PlaneRANSACEstimator3f(const ConstPointsView& points, int device = 0)
      : points_(points), device_(device), inlier_count_thresh_(points.cols() / 2 + points.cols() % 2) {}

  size_t getSampleSize() const { return 3; }
  size_t getTargetInlierCount() const { return inlier_count_thresh_; }
  PlaneRANSACEstimator3f& setTargetInlierCount(size_t v) { inlier_count_thresh_ = v; dirty_ = true; return *this; }
  size_t getMaxNumberOfIterations() const { return max_iter_; }
  PlaneRANSACEstimator3f& setMaxNumberOfIterations(size_t v) { max_iter_ = v; dirty_ = true; return *this; }
  float getMaxInlierResidual() const { return inlier_dist_thresh_; }
  PlaneRANSACEstimator3f& setMaxInlierResidual(float v) { inlier_dist_thresh_ = v; dirty_ = true; return *this; }
  bool getReEstimationStep() const { return re_estimate_; }
  PlaneRANSACEstimator3f& setReEstimationStep(bool v) { re_estimate_ = v; dirty_ = true; return *this; }

  // not in the reference (it seeds std::mt19937 from std::random_device): reproducible sampling
  PlaneRANSACEstimator3f& setSeed(uint64_t seed) { seed_ = seed; dirty_ = true; return *this; }
  PlaneRANSACEstimator3f& setSamples(const std::vector<uint32_t>& triples) { samples_ = triples; dirty_ = true; return *this; }

  // ransac_base.hpp:64-131
  PlaneRANSACEstimator3f& estimate() {
    if (!samples_.empty() && samples_.size() < 3 * max_iter_) throw std::invalid_argument("setSamples: 3 indices per iteration");
    const size_t n = points_.cols();
    cilhip_plane_model out;
    model_residuals_.assign(n, 0.0f);
    std::vector<uint32_t> inl(n ? n : 1);
    const int rc = cilhip_plane_ransac3f(device_, points_.data(), n, CILHIP_MEM_HOST, samples_.empty() ? nullptr : samples_.data(), seed_,
                                         inlier_dist_thresh_, inlier_count_thresh_, max_iter_, re_estimate_ ? 1 : 0, &out,
                                         model_residuals_.data(), inl.data());
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_plane_ransac3f failed (rc " + std::to_string(rc) + ")");
    for (int d = 0; d < 3; ++d) model_params_.coeffs_[d] = out.normal[d];
    model_params_.coeffs_[3] = out.offset;
    model_inliers_.assign(inl.begin(), inl.begin() + out.n_inliers);
    iteration_count_ = out.iterations;
    target_reached_ = out.target_reached != 0;
    device_ms_ = out.device_ms;
    dirty_ = false;
    return *this;
  }
  PlaneRANSACEstimator3f& estimate(float max_residual, size_t target_inlier_count, size_t max_iter) {   // :133-139
    inlier_dist_thresh_ = max_residual; inlier_count_thresh_ = target_inlier_count; max_iter_ = max_iter;
    return estimate();
  }

  // ransac_base.hpp:141-175 (lazy: run on first access)
  const Model& getModel() { ensure(); return model_params_; }
  PlaneRANSACEstimator3f& getModel(Model& m) { ensure(); m = model_params_; return *this; }
  const ResidualVector& getModelResiduals() { ensure(); return model_residuals_; }
  const IndexVector& getModelInliers() { ensure(); return model_inliers_; }
  bool targetInlierCountAchieved() { ensure(); return target_reached_; }
  size_t getNumberOfPerformedIterations() { ensure(); return iteration_count_; }
  size_t getNumberOfInliers() { ensure(); return model_inliers_.size(); }
  double getDeviceMilliseconds() const { return device_ms_; }

  // ransac_hyperplane_estimator.hpp:22-32: PCA plane through all points
  Model estimateModel() {
    Model m;
    const int rc = cilhip_plane_fit3f(device_, points_.data(), points_.cols(), CILHIP_MEM_HOST, m.coeffs_);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_plane_fit3f failed (rc " + std::to_string(rc) + ")");
    return m;
  }
  size_t getDataPointsCount() const { return points_.cols(); }

private:
  void ensure() { if (dirty_) estimate(); }
  ConstPointsView points_;
  int device_;
  size_t inlier_count_thresh_;
  size_t max_iter_ = 100;
  float inlier_dist_thresh_ = 0.1f;
  bool re_estimate_ = true;
  uint64_t seed_ = 0;
  std::vector<uint32_t> samples_;
  bool dirty_ = true;
  Model model_params_;
  ResidualVector model_residuals_;
  IndexVector model_inliers_;
  size_t iteration_count_ = 0;
  bool target_reached_ = false;
  double device_ms_ = 0.0;
};

// RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp:126; TransformRANSACEstimator<RigidTransform<float,3>>)
// over point PAIRS.  Constructors as the reference's (:25-59): paired clouds; clouds + a correspondence set; clouds + two index lists.
class RigidTransformRANSACEstimator3f {
public:
  typedef RigidTransform3f Model;
  typedef RigidTransform3f Transform;
  typedef float ResidualScalar;
  typedef std::vector<float> ResidualVector;
  typedef std::vector<size_t> IndexVector;
  enum { Dim = 3, MinSampleSize = 3 };

  RigidTransformRANSACEstimator3f(const ConstPointsView& dst_points, const ConstPointsView& src_points, int device = 0)
      : dst_(dst_points.data(), dst_points.data() + 3 * dst_points.cols()), src_(src_points.data(), src_points.data() + 3 * src_points.cols()),
        device_(device) { init(); }
  template <class CorrespondencesT>
  RigidTransformRANSACEstimator3f(const ConstPointsView& dst_points, const ConstPointsView& src_points, const CorrespondencesT& corr, int device = 0)
      : device_(device) {
    dst_.resize(3 * corr.size()); src_.resize(3 * corr.size());
    for (size_t i = 0; i < corr.size(); ++i)
      for (int d = 0; d < 3; ++d) { dst_[3 * i + d] = dst_points.data()[3 * corr[i].indexInFirst + d]; src_[3 * i + d] = src_points.data()[3 * corr[i].indexInSecond + d]; }
    init();
  }
  template <typename IdxT>
  RigidTransformRANSACEstimator3f(const ConstPointsView& dst_points, const ConstPointsView& src_points, const std::vector<IdxT>& dst_ind,
                                  const std::vector<IdxT>& src_ind, int device = 0)
      : device_(device) {
    if (dst_ind.size() != src_ind.size()) throw std::invalid_argument("dst / src pairs must have the same length");
    dst_.resize(3 * dst_ind.size()); src_.resize(3 * src_ind.size());
    for (size_t i = 0; i < dst_ind.size(); ++i)
      for (int d = 0; d < 3; ++d) { dst_[3 * i + d] = dst_points.data()[3 * dst_ind[i] + d]; src_[3 * i + d] = src_points.data()[3 * src_ind[i] + d]; }
    init();
  }

  size_t getSampleSize() const { return 3; }
  size_t getTargetInlierCount() const { return inlier_count_thresh_; }
  RigidTransformRANSACEstimator3f& setTargetInlierCount(size_t v) { inlier_count_thresh_ = v; dirty_ = true; return *this; }
  size_t getMaxNumberOfIterations() const { return max_iter_; }
  RigidTransformRANSACEstimator3f& setMaxNumberOfIterations(size_t v) { max_iter_ = v; dirty_ = true; return *this; }
  float getMaxInlierResidual() const { return inlier_dist_thresh_; }
  RigidTransformRANSACEstimator3f& setMaxInlierResidual(float v) { inlier_dist_thresh_ = v; dirty_ = true; return *this; }
  bool getReEstimationStep() const { return re_estimate_; }
  RigidTransformRANSACEstimator3f& setReEstimationStep(bool v) { re_estimate_ = v; dirty_ = true; return *this; }
  RigidTransformRANSACEstimator3f& setSeed(uint64_t seed) { seed_ = seed; dirty_ = true; return *this; }                       // (extension)
  RigidTransformRANSACEstimator3f& setSamples(const std::vector<uint32_t>& triples) { samples_ = triples; dirty_ = true; return *this; }

  RigidTransformRANSACEstimator3f& estimate() {      // ransac_base.hpp:64-131
    if (!samples_.empty() && samples_.size() < 3 * max_iter_) throw std::invalid_argument("setSamples: 3 indices per iteration");
    const size_t n = getDataPointsCount();
    cilhip_transform_model out;
    model_residuals_.assign(n, 0.0f);
    std::vector<uint32_t> inl(n ? n : 1);
    const int rc = cilhip_transform_ransac3f(device_, dst_.data(), src_.data(), n, CILHIP_MEM_HOST, samples_.empty() ? nullptr : samples_.data(), seed_,
                                             inlier_dist_thresh_, inlier_count_thresh_, max_iter_, re_estimate_ ? 1 : 0, &out, model_residuals_.data(),
                                             inl.data());
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_transform_ransac3f failed (rc " + std::to_string(rc) + ")");
    std::memcpy(model_params_.m, out.T, sizeof(out.T));
    if (!out.have_model) model_residuals_.clear();      // no accepted hypothesis, no re-estimation: the reference's vectors stay empty
    model_inliers_.assign(inl.begin(), inl.begin() + out.n_inliers);
    iteration_count_ = out.iterations;
    target_reached_ = out.target_reached != 0;
    dirty_ = false;
    return *this;
  }
  RigidTransformRANSACEstimator3f& estimate(float max_residual, size_t target_inlier_count, size_t max_iter) {
    inlier_dist_thresh_ = max_residual; inlier_count_thresh_ = target_inlier_count; max_iter_ = max_iter;
    return estimate();
  }
  const Model& getModel() { ensure(); return model_params_; }
  const ResidualVector& getModelResiduals() { ensure(); return model_residuals_; }
  const IndexVector& getModelInliers() { ensure(); return model_inliers_; }
  bool targetInlierCountAchieved() { ensure(); return target_reached_; }
  size_t getNumberOfPerformedIterations() { ensure(); return iteration_count_; }
  size_t getNumberOfInliers() { ensure(); return model_inliers_.size(); }

  Model estimateModel() {      // :61-72: the closed-form fit over all pairs
    Model m;
    const int rc = cilhip_transform_fit3f(device_, dst_.data(), src_.data(), getDataPointsCount(), CILHIP_MEM_HOST, m.m);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_transform_fit3f failed (rc " + std::to_string(rc) + ")");
    return m;
  }
  size_t getDataPointsCount() const { return dst_.size() / 3; }

private:
  void init() {
    if (dst_.size() != src_.size()) throw std::invalid_argument("dst / src pairs must have the same length");
    const size_t n = dst_.size() / 3;
    inlier_count_thresh_ = n / 2 + n % 2;      // :27-30
  }
  void ensure() { if (dirty_) estimate(); }
  std::vector<float> dst_, src_;
  int device_;
  size_t inlier_count_thresh_ = 0;
  size_t max_iter_ = 100;
  float inlier_dist_thresh_ = 0.01f;
  bool re_estimate_ = true;
  uint64_t seed_ = 0;
  std::vector<uint32_t> samples_;
  bool dirty_ = true;
  Model model_params_;
  ResidualVector model_residuals_;
  IndexVector model_inliers_;
  size_t iteration_count_ = 0;
  bool target_reached_ = false;
};

class KMeans3f {
public:
  explicit KMeans3f(const ConstPointsView& data, int device = 0) : data_(data), device_(device) {}

  // kmeans.hpp:24-30: cluster(initial centroids, max_iter, tol, use_kd_tree)
  KMeans3f& cluster(const ConstPointsView& centroids, size_t max_iter = 100, float tol = std::numeric_limits<float>::epsilon(),
                    bool use_kd_tree = false) {
    cluster_centroids_.assign(centroids.data(), centroids.data() + 3 * centroids.cols());
    return run(max_iter, tol, use_kd_tree);
  }
  // kmeans.hpp:32-53: k distinct random points as initial centroids
  KMeans3f& cluster(size_t num_clusters, size_t max_iter = 100, float tol = std::numeric_limits<float>::epsilon(), bool use_kd_tree = false) {
    const size_t n = data_.cols();
    if (num_clusters > n) num_clusters = n;
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    std::mt19937 rng(std::random_device{}());
    cluster_centroids_.resize(3 * num_clusters);
    for (size_t i = 0; i < num_clusters; ++i) {
      std::uniform_int_distribution<size_t> dist(i, n - 1);
      std::swap(perm[i], perm[dist(rng)]);
      for (int d = 0; d < 3; ++d) cluster_centroids_[3 * i + d] = data_.data()[3 * perm[i] + d];
    }
    return run(max_iter, tol, use_kd_tree);
  }

  const std::vector<float>& getClusterCentroids() const { return cluster_centroids_; }   // packed xyz, k points
  const std::vector<size_t>& getPointToClusterIndexMap() const { return point_to_cluster_index_map_; }
  size_t getNumberOfClusters() const { return cluster_centroids_.size() / 3; }
  size_t getNumberOfPerformedIterations() const { return iteration_count_; }
  // clustering_base.hpp:22-33
  std::vector<std::vector<size_t>> getClusterToPointIndicesMap() const {
    std::vector<std::vector<size_t>> m(getNumberOfClusters());
    for (size_t i = 0; i < point_to_cluster_index_map_.size(); ++i) m[point_to_cluster_index_map_[i]].push_back(i);
    return m;
  }

private:
  // use_kd_tree (kmeans.hpp:86-94): the same exhaustive nearest-centroid pass with nanoflann's rounding of the distance (c_api.h)
  KMeans3f& run(size_t max_iter, float tol, bool use_kd_tree) {
    const size_t n = data_.cols(), k = cluster_centroids_.size() / 3;
    std::vector<uint32_t> lab(n ? n : 1);
    const int rc = cilhip_kmeans3f_ex(device_, data_.data(), n, CILHIP_MEM_HOST, cluster_centroids_.data(), k, max_iter, tol, use_kd_tree ? 1 : 0,
                                      lab.data(), &iteration_count_);
    if (rc != CILHIP_OK) throw std::runtime_error("cilhip_kmeans3f failed (rc " + std::to_string(rc) + ")");
    point_to_cluster_index_map_.assign(lab.begin(), lab.begin() + n);
    return *this;
  }
  ConstPointsView data_;
  int device_;
  std::vector<float> cluster_centroids_;
  std::vector<size_t> point_to_cluster_index_map_;
  size_t iteration_count_ = 0;
};

}  // namespace cilantro_hip
