// cilantro_hip/icp.hpp -- C++ host-side mirror of cilantro's rigid-ICP template surface for this
// path, header-only, on top of the C ABI (c_api.h).  Same class / method names, argument meaning,
// defaults and result conventions as the reference, so code written against
//
//     cilantro::SimpleCombinedMetricRigidICP3f icp(dst.points, dst.normals, src.points);
//     icp.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0f)
//        .setPointToPlaneMetricWeight(1.0f);
//     icp.correspondenceSearchEngine().setMaxDistance(0.1f * 0.1f);
//     icp.setConvergenceTolerance(1e-4f).setMaxNumberOfIterations(30);
//     auto tf = icp.estimate().getTransform();                    (examples/rigid_icp.cpp:116-125)
//
// switches engines by changing the namespace.  Mirrors (paths relative to
// /root/reference/include/cilantro/):
//   registration/icp_base.hpp:8-123                        IterativeClosestPointBase (CRTP, setters return *this)
//   registration/icp_single_transform_point_to_point_metric.hpp
//   registration/icp_single_transform_combined_metric.hpp  (weights / GN step knobs :100-143)
//   registration/icp_common_instances.hpp:34-45,74-97,250,261   Simple* wrappers
//   correspondence_search/correspondence_search_kd_tree.hpp:23-307  engine concept + knobs :239-271
//   core/correspondence.hpp:9-55                            Correspondence / CorrespondenceSet
//
// Eigen is NOT required: clouds are passed as non-owning (pointer, count) views with the reference's
// memory layout (3xN column-major float == packed xyz).  When <Eigen/Dense> is available the
// overloads at the bottom accept cilantro's own ConstVectorSetMatrixMap / RigidTransform3f types.
//
// Every option combination the GPU path does not implement throws std::invalid_argument; results
// never silently differ from the reference.  There is no CPU fallback: without a usable HIP device
// construction throws std::runtime_error.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <limits>
#include <memory>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "c_api.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define CILANTRO_HIP_HAVE_EIGEN 1
#endif
#endif

namespace cilantro_hip {

// core/data_containers.hpp:73-122: non-owning view of a 3xN column-major float matrix
struct ConstPointsView {
  const float* data_ = nullptr;
  size_t cols_ = 0;
  ConstPointsView() = default;
  ConstPointsView(const float* d, size_t n) : data_(d), cols_(n) {}
  ConstPointsView(const std::vector<float>& xyz) : data_(xyz.data()), cols_(xyz.size() / 3) {}
  const float* data() const { return data_; }
  size_t cols() const { return cols_; }
  static constexpr size_t rows() { return 3; }
};

// core/space_transformations.hpp:54-55: Eigen::Transform<float,3,Isometry> storage (4x4 column-major)
struct RigidTransform3f {
  float m[16];
  RigidTransform3f() { setIdentity(); }
  void setIdentity() {
    std::memset(m, 0, sizeof(m));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
  }
  static RigidTransform3f Identity() { return RigidTransform3f(); }
  const float* data() const { return m; }
  float* data() { return m; }
  float linear(int r, int c) const { return m[c * 4 + r]; }
  float& linear(int r, int c) { return m[c * 4 + r]; }
  float translation(int r) const { return m[12 + r]; }
  float& translation(int r) { return m[12 + r]; }
  float operator()(int r, int c) const { return m[c * 4 + r]; }
};

enum struct CorrespondenceSearchDirection { FIRST_TO_SECOND, SECOND_TO_FIRST, BOTH };  // core/correspondence.hpp:7

template <typename ScalarT, typename IndexT = size_t>
struct Correspondence {  // core/correspondence.hpp:9-52
  using Scalar = ScalarT;
  using Index = IndexT;
  IndexT indexInFirst;
  IndexT indexInSecond;
  ScalarT value;
};
template <typename ScalarT, typename IndexT = size_t>
using CorrespondenceSet = std::vector<Correspondence<ScalarT, IndexT>>;

namespace internal {
struct CtxDeleter {
  void operator()(cilhip_ctx* c) const { cilhip_destroy(c); }
};
using CtxPtr = std::unique_ptr<cilhip_ctx, CtxDeleter>;

inline void check(cilhip_ctx* c, int rc, const char* what) {
  if (rc == CILHIP_OK) return;
  std::string msg = std::string(what) + ": " + (c ? cilhip_last_error(c) : "no context");
  if (rc == CILHIP_ERR_UNSUPPORTED || rc == CILHIP_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

inline CtxPtr make_ctx(int device) {
  cilhip_ctx* c = nullptr;
  const int rc = cilhip_create(&c, device);
  if (rc != CILHIP_OK)
    throw std::runtime_error("cilhip_create failed (no usable HIP device; libcilantro_hip has no CPU fallback)");
  return CtxPtr(c);
}

template <class TransformT>
inline void to_abi(const TransformT& t, float out[16]) {
  std::memcpy(out, t.data(), 16 * sizeof(float));  // Eigen::Transform::data() and RigidTransform3f::data(): col-major 4x4
}
}  // namespace internal

// Models cilantro's correspondence-search engine concept (the CorrespondenceSearchEngineT template
// parameter of the ICP classes, registration/icp_base.hpp:8-10) with the search on the GPU.
class CorrespondenceSearchHIP {
public:
  using CorrespondenceScalar = float;
  using CorrespondenceIndex = size_t;
  using SearchResult = CorrespondenceSet<CorrespondenceScalar, CorrespondenceIndex>;

  explicit CorrespondenceSearchHIP(cilhip_ctx* ctx)
      : ctx_(ctx),
        search_dir_(CorrespondenceSearchDirection::SECOND_TO_FIRST),  // correspondence_search_kd_tree.hpp:47
        max_distance_(0.01f * 0.01f),                                  // :48 (squared)
        inlier_fraction_(1.0),
        require_reciprocality_(false),
        one_to_one_(false),
        fetched_(false) {}

  CorrespondenceSearchHIP& findCorrespondences() { return findCorrespondences(RigidTransform3f::Identity()); }

  template <class TransformT>
  CorrespondenceSearchHIP& findCorrespondences(const TransformT& tform) {  // :107
    float T[16];
    internal::to_abi(tform, T);
    internal::check(ctx_, cilhip_find_correspondences(ctx_, T, max_distance_, nullptr), "findCorrespondences");
    fetched_ = false;
    return *this;
  }

  // ascending source index, as correspondence_search_kd_tree_utilities.hpp:45-50 leaves them
  const SearchResult& getCorrespondences() const {
    if (!fetched_) {
      size_t n = 0;
      const size_t cap = capacity_;  // = points of both clouds (direction BOTH: up to one match per point of either)
      std::vector<uint64_t> i1, i2;
      std::vector<float> v;
      i1.resize(cap ? cap : 1); i2.resize(cap ? cap : 1); v.resize(cap ? cap : 1);
      internal::check(ctx_, cilhip_get_correspondences(ctx_, i1.data(), i2.data(), v.data(), cap, &n), "getCorrespondences");
      correspondences_.resize(n);
      for (size_t k = 0; k < n; ++k) correspondences_[k] = {static_cast<size_t>(i1[k]), static_cast<size_t>(i2[k]), v[k]};
      fetched_ = true;
    }
    return correspondences_;
  }

  void invalidateFetched() { fetched_ = false; }      // (extension: the ICP loop replaced the context's correspondence set)
  const CorrespondenceSearchDirection& getSearchDirection() const { return search_dir_; }
  // correspondence_search_kd_tree.hpp:239-247.  FIRST_TO_SECOND / BOTH rebuild an index over the transformed source every
  // search (as the reference rebuilds its kd-tree) and hand the estimators a pair list.
  CorrespondenceSearchHIP& setSearchDirection(const CorrespondenceSearchDirection& d) {
    const double v = d == CorrespondenceSearchDirection::SECOND_TO_FIRST ? 0.0 : d == CorrespondenceSearchDirection::FIRST_TO_SECOND ? 1.0 : 2.0;
    internal::check(ctx_, cilhip_set_option(ctx_, "search_direction", v), "setSearchDirection");
    search_dir_ = d;
    fetched_ = false;
    return *this;
  }
  CorrespondenceScalar getMaxDistance() const { return max_distance_; }
  CorrespondenceSearchHIP& setMaxDistance(CorrespondenceScalar dist_thresh) {  // SQUARED, as the reference
    max_distance_ = dist_thresh;
    return *this;
  }
  double getInlierFraction() const { return inlier_fraction_; }
  CorrespondenceSearchHIP& setInlierFraction(double fraction) {
    // core/correspondence.hpp:57-66: the filter only acts for 0 < fraction < 1
    internal::check(ctx_, cilhip_set_option(ctx_, "inlier_fraction", fraction), "setInlierFraction");
    inlier_fraction_ = fraction;
    fetched_ = false;
    return *this;
  }
  bool getRequireReciprocality() const { return require_reciprocality_; }
  CorrespondenceSearchHIP& setRequireReciprocality(bool b) {  // :257-263: only read when the direction is BOTH
    internal::check(ctx_, cilhip_set_option(ctx_, "require_reciprocality", b ? 1.0 : 0.0), "setRequireReciprocality");
    require_reciprocality_ = b;
    fetched_ = false;
    return *this;
  }
  bool getOneToOne() const { return one_to_one_; }
  CorrespondenceSearchHIP& setOneToOne(bool b) {  // core/correspondence.hpp:68-100 (SECOND_TO_FIRST branch)
    internal::check(ctx_, cilhip_set_option(ctx_, "one_to_one", b ? 1.0 : 0.0), "setOneToOne");
    one_to_one_ = b;
    fetched_ = false;
    return *this;
  }

  // Feature adaptors of the engine (common_transformable_feature_adaptors.hpp): the default is PointFeaturesAdaptor3f on
  // both clouds; this switches both to PointNormalFeaturesAdaptor3f(points, normals, normal_weight) (:60-161) -- the
  // search then runs on the 6-D features (p, w n).  The target's normals are the ones the ICP object was built with;
  // src_normals: the source's (pass an empty view to reuse normals given to a four-cloud ICP constructor).  With
  // keep_metric = true (default) the combined metric stays the three-cloud one.  normal_weight = 0 switches back.
  CorrespondenceSearchHIP& setPointNormalFeatureAdaptors(const ConstPointsView& src_normals, float normal_weight, bool keep_metric = true) {
    if (src_normals.cols()) {
      internal::check(ctx_, cilhip_set_source_normals(ctx_, src_normals.data(), CILHIP_MEM_HOST), "set_source_normals");
      internal::check(ctx_, cilhip_set_option(ctx_, "symmetric_metric", keep_metric ? 0.0 : 1.0), "symmetric_metric");
    }
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_kind", 0.0), "feature_kind");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_normal_weight", (double)normal_weight), "feature_normal_weight");
    fetched_ = false;
    return *this;
  }
  // ... or PointColorFeaturesAdaptor3f(points, colors, color_weight) on both clouds (:164-252): features (p, w c), the colour part
  // does not move with the transform.  color_weight = 0 switches back to point features.
  CorrespondenceSearchHIP& setPointColorFeatureAdaptors(const ConstPointsView& dst_colors, const ConstPointsView& src_colors, float color_weight) {
    internal::check(ctx_, cilhip_set_color_features(ctx_, dst_colors.data(), src_colors.data(), CILHIP_MEM_HOST), "set_color_features");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_kind", 1.0), "feature_kind");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_normal_weight", (double)color_weight), "feature_normal_weight");
    fetched_ = false;
    return *this;
  }

  // ... or PointNormalColorFeaturesAdaptor3f(points, normals, colors, normal_weight, color_weight) on both clouds (:255-343): 9-D
  // features (p, wn n, wc c); the normal part follows the transform like the point+normal adaptor's, the colour part does not move.
  CorrespondenceSearchHIP& setPointNormalColorFeatureAdaptors(const ConstPointsView& src_normals, const ConstPointsView& dst_colors,
                                                              const ConstPointsView& src_colors, float normal_weight, float color_weight,
                                                              bool keep_metric = true) {
    if (src_normals.cols()) {
      internal::check(ctx_, cilhip_set_source_normals(ctx_, src_normals.data(), CILHIP_MEM_HOST), "set_source_normals");
      internal::check(ctx_, cilhip_set_option(ctx_, "symmetric_metric", keep_metric ? 0.0 : 1.0), "symmetric_metric");
    }
    internal::check(ctx_, cilhip_set_color_features(ctx_, dst_colors.data(), src_colors.data(), CILHIP_MEM_HOST), "set_color_features");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_kind", 2.0), "feature_kind");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_normal_weight", (double)normal_weight), "feature_normal_weight");
    internal::check(ctx_, cilhip_set_option(ctx_, "feature_color_weight", (double)color_weight), "feature_color_weight");
    fetched_ = false;
    return *this;
  }

  void setSourceCount_(size_t n) { capacity_ += n; }  // internal: result capacity (called with both cloud sizes)
  cilhip_ctx* context() const { return ctx_; }        // (extension)

private:
  cilhip_ctx* ctx_;
  CorrespondenceSearchDirection search_dir_;
  CorrespondenceScalar max_distance_;
  double inlier_fraction_;
  bool require_reciprocality_;
  bool one_to_one_;
  size_t capacity_ = 0;
  mutable bool fetched_;
  mutable SearchResult correspondences_;
};

// registration/icp_base.hpp: CRTP base with the reference's public interface.
template <class ICPInstanceT>
class IterativeClosestPointBase {
public:
  using Transform = RigidTransform3f;
  using Scalar = float;
  using CorrespondenceSearchEngine = CorrespondenceSearchHIP;
  using ResidualVector = std::vector<float>;

  inline const CorrespondenceSearchEngine& correspondenceSearchEngine() const { return engine_; }
  inline CorrespondenceSearchEngine& correspondenceSearchEngine() { return engine_; }

  inline size_t getMaxNumberOfIterations() const { return max_iterations_; }
  inline ICPInstanceT& setMaxNumberOfIterations(size_t max_iter) { max_iterations_ = max_iter; return self(); }
  inline size_t getNumberOfPerformedIterations() const { return iterations_; }
  inline float getConvergenceTolerance() const { return convergence_tol_; }
  inline ICPInstanceT& setConvergenceTolerance(float conv_tol) { convergence_tol_ = conv_tol; return self(); }
  inline const Transform& getInitialTransform() const { return transform_init_; }
  template <class TransformT>
  inline ICPInstanceT& setInitialTransform(const TransformT& tform_init) {
    internal::to_abi(tform_init, transform_init_.m);
    return self();
  }
  inline Transform& initialTransform() { return transform_init_; }
  inline float getLastUpdateNorm() const { return last_delta_norm_; }

  // Main ICP loop (icp_base.hpp:68-87): all iterations run on the device, one sync at the end.
  ICPInstanceT& estimate() {
    cilhip_icp_params p;
    cilhip_icp_default_params(&p);
    self().fillParams(p);
    p.max_iter = max_iterations_;
    p.conv_tol = convergence_tol_;
    p.max_sq_dist = engine_.getMaxDistance();
    cilhip_icp_result r;
    pending_exception_ = nullptr;
    const int run_rc = cilhip_icp_run(ctx_.get(), &p, transform_init_.m, &r);
    // (a functor evaluator that threw inside the library's callback: caught at the C boundary, rethrown here)
    if (pending_exception_) { std::exception_ptr e = pending_exception_; pending_exception_ = nullptr; std::rethrow_exception(e); }
    internal::check(ctx_.get(), run_rc, "estimate");
    engine_.invalidateFetched();      // the engine now holds the last iteration's set (correspondence_search_kd_tree.hpp:231)
    std::memcpy(transform_.m, r.T, sizeof(r.T));
    iterations_ = r.iterations;
    last_delta_norm_ = r.last_delta_norm;
    last_ncorr_ = r.last_ncorr;
    return self();
  }
  inline ICPInstanceT& estimate(size_t max_iter, float conv_tol) {
    max_iterations_ = max_iter;
    convergence_tol_ = conv_tol;
    return estimate();
  }
  inline const Transform& getTransform() const { return transform_; }
  template <class TransformT>
  inline const ICPInstanceT& getTransform(TransformT& tform) const {
    std::memcpy(tform.data(), transform_.m, sizeof(transform_.m));
    return static_cast<const ICPInstanceT&>(*this);
  }
  inline ResidualVector getResiduals() { return self().computeResiduals(); }
  inline bool hasConverged() const { return last_delta_norm_ < convergence_tol_; }
  inline size_t getNumberOfLastCorrespondences() const { return last_ncorr_; }  // (extension)
  inline cilhip_ctx* context() { return ctx_.get(); }                            // (extension)

protected:
  IterativeClosestPointBase(int device, size_t max_iter = 15, float conv_tol = 1e-5f)  // icp_base.hpp:24-25
      : ctx_(internal::make_ctx(device)),
        engine_(ctx_.get()),
        max_iterations_(max_iter),
        iterations_(0),
        convergence_tol_(conv_tol),
        last_delta_norm_(std::numeric_limits<float>::infinity()),
        last_ncorr_(0) {}

  ICPInstanceT& self() { return *static_cast<ICPInstanceT*>(this); }

  internal::CtxPtr ctx_;
  CorrespondenceSearchEngine engine_;
  size_t max_iterations_;
  size_t iterations_;
  float convergence_tol_;
  float last_delta_norm_;
  size_t last_ncorr_;
  std::exception_ptr pending_exception_;      // thrown by a caller's functor inside a library callback: rethrown by estimate()
  Transform transform_init_;
  Transform transform_;
  size_t n_src_ = 0;
};

// icp_single_transform_point_to_point_metric.hpp + icp_common_instances.hpp:34-45,250
class SimplePointToPointMetricRigidICP3f : public IterativeClosestPointBase<SimplePointToPointMetricRigidICP3f> {
  using Base = IterativeClosestPointBase<SimplePointToPointMetricRigidICP3f>;
  friend Base;

public:
  SimplePointToPointMetricRigidICP3f(const ConstPointsView& dst, const ConstPointsView& src, int device = 0)
      : Base(device) {
    internal::check(ctx_.get(), cilhip_set_target(ctx_.get(), dst.data(), nullptr, dst.cols(), CILHIP_MEM_HOST), "set_target");
    internal::check(ctx_.get(), cilhip_set_source(ctx_.get(), src.data(), src.cols(), CILHIP_MEM_HOST), "set_source");
    n_src_ = src.cols();
    engine_.setSourceCount_(n_src_);
    engine_.setSourceCount_(dst.cols());
  }

private:
  void fillParams(cilhip_icp_params& p) const { p.metric = CILHIP_METRIC_POINT_TO_POINT; }
  std::vector<float> computeResiduals() {
    std::vector<float> res(n_src_ ? n_src_ : 1);
    internal::check(ctx_.get(), cilhip_compute_residuals(ctx_.get(), CILHIP_METRIC_POINT_TO_POINT, 0.f, 0.f, transform_.m, res.data(), CILHIP_MEM_HOST), "getResiduals");
    res.resize(n_src_);
    return res;
  }
};

// The correspondence weight evaluators of the combined-metric classes (core/common_pair_evaluators.hpp:14-27 Identity,
// :30-43 Unity, :46-80 RBF kernel over squared distances).  The reference fixes their TYPES as template arguments
// (icp_single_transform_combined_metric.hpp:11-14); here one runtime object covers the three stock classes -- evaluated on the device --
// and any functor of the reference's call shape, evaluator(indexInFirst, indexInSecond, value) (transform_estimation.hpp:303, :332):
// such an object makes the estimates call back to the host once per estimate (cilhip_set_pair_weight_callback).
class CorrespondenceWeightEvaluator {
public:
  enum Kind { Unity = 0, Identity = 1, RBFKernel = 2, Custom = 3 };
  using Functor = std::function<float(size_t, size_t, float)>;
  CorrespondenceWeightEvaluator(Kind kind = Unity, float sigma = 1.0f) : kind_(kind), sigma_(sigma) {}
  CorrespondenceWeightEvaluator(Functor f) : kind_(Custom), sigma_(1.0f), fn_(std::move(f)) {}
  inline CorrespondenceWeightEvaluator& setKind(Kind k) { kind_ = k; return *this; }
  inline CorrespondenceWeightEvaluator& setSigma(float sigma) { sigma_ = sigma; return *this; }   // :55-58
  inline CorrespondenceWeightEvaluator& setFunctor(Functor f) { fn_ = std::move(f); kind_ = Custom; return *this; }
  inline Kind kind() const { return kind_; }
  inline float sigma() const { return sigma_; }
  // the host evaluation (what the reference's class computes): used when EITHER evaluator of an instance is a functor
  inline float operator()(size_t i, size_t j, float value) const {
    switch (kind_) {
      case Unity: return 1.0f;
      case Identity: return value;
      case RBFKernel: return std::exp((-0.5f / (sigma_ * sigma_)) * value);
      default: return fn_(i, j, value);
    }
  }

private:
  Kind kind_;
  float sigma_;
  Functor fn_;
};

// icp_single_transform_combined_metric.hpp + icp_common_instances.hpp:74-97,261
class SimpleCombinedMetricRigidICP3f : public IterativeClosestPointBase<SimpleCombinedMetricRigidICP3f> {
  using Base = IterativeClosestPointBase<SimpleCombinedMetricRigidICP3f>;
  friend Base;

public:
  SimpleCombinedMetricRigidICP3f(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p, int device = 0)
      : Base(device),
        max_optimization_iterations_(1),             // icp_single_transform_combined_metric.hpp:44-47
        optimization_convergence_tol_(1e-5f),
        point_to_point_weight_(0.0f),
        point_to_plane_weight_(1.0f) {
    if (dst_n.cols() != dst_p.cols()) throw std::invalid_argument("dst normals must match dst points");
    internal::check(ctx_.get(), cilhip_set_target(ctx_.get(), dst_p.data(), dst_n.data(), dst_p.cols(), CILHIP_MEM_HOST), "set_target");
    internal::check(ctx_.get(), cilhip_set_source(ctx_.get(), src_p.data(), src_p.cols(), CILHIP_MEM_HOST), "set_source");
    n_src_ = src_p.cols();
    engine_.setSourceCount_(n_src_);
    engine_.setSourceCount_(dst_p.cols());
  }

  // four-cloud form (icp_common_instances.hpp:88-97): source normals => symmetric metric
  SimpleCombinedMetricRigidICP3f(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p,
                                 const ConstPointsView& src_n, int device = 0)
      : SimpleCombinedMetricRigidICP3f(dst_p, dst_n, src_p, device) {
    if (src_n.cols() != src_p.cols()) throw std::invalid_argument("src normals must match src points");
    internal::check(ctx_.get(), cilhip_set_source_normals(ctx_.get(), src_n.data(), CILHIP_MEM_HOST), "set_source_normals");
  }

  inline float getPointToPointMetricWeight() const { return point_to_point_weight_; }
  inline SimpleCombinedMetricRigidICP3f& setPointToPointMetricWeight(float w) { point_to_point_weight_ = w; return *this; }
  inline float getPointToPlaneMetricWeight() const { return point_to_plane_weight_; }
  inline SimpleCombinedMetricRigidICP3f& setPointToPlaneMetricWeight(float w) { point_to_plane_weight_ = w; return *this; }
  inline size_t getMaxNumberOfOptimizationStepIterations() const { return max_optimization_iterations_; }
  inline SimpleCombinedMetricRigidICP3f& setMaxNumberOfOptimizationStepIterations(size_t n) { max_optimization_iterations_ = n; return *this; }
  inline float getOptimizationStepConvergenceTolerance() const { return optimization_convergence_tol_; }
  inline SimpleCombinedMetricRigidICP3f& setOptimizationStepConvergenceTolerance(float t) { optimization_convergence_tol_ = t; return *this; }
  // icp_single_transform_combined_metric.hpp:95-101
  inline CorrespondenceWeightEvaluator& pointToPointCorrespondenceWeightEvaluator() { return point_corr_eval_; }
  inline CorrespondenceWeightEvaluator& pointToPlaneCorrespondenceWeightEvaluator() { return plane_corr_eval_; }

private:
  static void weightTrampoline_(void* user, const uint64_t* i1, const uint64_t* i2, const float* value, size_t n, float* wq, float* wl) {
    SimpleCombinedMetricRigidICP3f* self = static_cast<SimpleCombinedMetricRigidICP3f*>(user);
    try {
      if (self->pending_exception_) return;      // (an earlier call threw: the run is abandoned by estimate(); weights stay 0)
      for (size_t k = 0; k < n; ++k) {
        wq[k] = self->point_corr_eval_((size_t)i1[k], (size_t)i2[k], value[k]);
        wl[k] = self->plane_corr_eval_((size_t)i1[k], (size_t)i2[k], value[k]);
      }
    } catch (...) {      // (no exception may cross the C frames of the library)
      self->pending_exception_ = std::current_exception();
      for (size_t k = 0; k < n; ++k) wq[k] = wl[k] = 0.0f;
    }
  }
  void fillParams(cilhip_icp_params& p) const {
    const bool custom = point_corr_eval_.kind() == CorrespondenceWeightEvaluator::Custom || plane_corr_eval_.kind() == CorrespondenceWeightEvaluator::Custom;
    internal::check(ctx_.get(), cilhip_set_pair_weight_callback(ctx_.get(), custom ? &weightTrampoline_ : nullptr, const_cast<SimpleCombinedMetricRigidICP3f*>(this)),
                    "set_pair_weight_callback");
    if (!custom) {
      internal::check(ctx_.get(), cilhip_set_option(ctx_.get(), "point_weight_evaluator", (double)point_corr_eval_.kind()), "point_weight_evaluator");
      internal::check(ctx_.get(), cilhip_set_option(ctx_.get(), "point_weight_sigma", (double)point_corr_eval_.sigma()), "point_weight_sigma");
      internal::check(ctx_.get(), cilhip_set_option(ctx_.get(), "plane_weight_evaluator", (double)plane_corr_eval_.kind()), "plane_weight_evaluator");
      internal::check(ctx_.get(), cilhip_set_option(ctx_.get(), "plane_weight_sigma", (double)plane_corr_eval_.sigma()), "plane_weight_sigma");
    }
    p.metric = CILHIP_METRIC_COMBINED;
    p.w_p2p = point_to_point_weight_;
    p.w_p2pl = point_to_plane_weight_;
    p.max_opt_iter = max_optimization_iterations_;
    p.opt_conv_tol = optimization_convergence_tol_;
  }
  std::vector<float> computeResiduals() {
    std::vector<float> res(n_src_ ? n_src_ : 1);
    internal::check(ctx_.get(), cilhip_compute_residuals(ctx_.get(), CILHIP_METRIC_COMBINED, point_to_point_weight_, point_to_plane_weight_, transform_.m, res.data(), CILHIP_MEM_HOST), "getResiduals");
    res.resize(n_src_);
    return res;
  }

  size_t max_optimization_iterations_;
  float optimization_convergence_tol_;
  float point_to_point_weight_;
  float point_to_plane_weight_;
  CorrespondenceWeightEvaluator point_corr_eval_, plane_corr_eval_;
};

// Affine instances (registration/icp_common_instances.hpp:255, :266): the same loop and correspondence engine with an
// AffineTransform -- the step is the 12-unknown closed form (transform_estimation.hpp:50-102 on the raw coordinates for
// the point-to-point class, :369-476 with the means for the combined class) and there is no rotation() polish.  The 4x4
// holder type is shared with the rigid instances (AffineTransform3f below): its linear part is then a general 3x3 matrix.
// A correspondence search engine that OWNS its context and clouds: what a user builds beside an ICP object's own engine, e.g. the
// two engines of a CorrespondenceSearchCombinedMetricCombiner (the reference constructs CorrespondenceSearchKDTree objects over
// feature adaptors of the same clouds, correspondence_search_kd_tree.hpp:25-45).
class CorrespondenceSearchHIPOwned : private internal::CtxPtr, public CorrespondenceSearchHIP {
public:
  CorrespondenceSearchHIPOwned(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p, int device = 0)
      : internal::CtxPtr(internal::make_ctx(device)), CorrespondenceSearchHIP(internal::CtxPtr::get()) {
    if (dst_n.cols() && dst_n.cols() != dst_p.cols()) throw std::invalid_argument("dst normals must match dst points");
    internal::check(context(), cilhip_set_target(context(), dst_p.data(), dst_n.cols() ? dst_n.data() : nullptr, dst_p.cols(), CILHIP_MEM_HOST), "set_target");
    internal::check(context(), cilhip_set_source(context(), src_p.data(), src_p.cols(), CILHIP_MEM_HOST), "set_source");
    setSourceCount_(src_p.cols());
    setSourceCount_(dst_p.cols());
  }
};

// registration/correspondence_search_combined_metric_combiner.hpp:8-81: the point-to-point terms of the combined metric read one
// engine's correspondences, the point-to-plane terms another's.  Both engines hold the same two clouds.
template <class PointToPointCorrespondenceSearchT, class PointToPlaneCorrespondenceSearchT>
class CorrespondenceSearchCombinedMetricCombiner {
public:
  using PointToPointCorrespondenceSearch = PointToPointCorrespondenceSearchT;
  using PointToPlaneCorrespondenceSearch = PointToPlaneCorrespondenceSearchT;
  using PointToPointCorrespondenceSearchResult = typename PointToPointCorrespondenceSearchT::SearchResult;
  using PointToPlaneCorrespondenceSearchResult = typename PointToPlaneCorrespondenceSearchT::SearchResult;

  CorrespondenceSearchCombinedMetricCombiner(PointToPointCorrespondenceSearch& point_to_point_corr_search,
                                             PointToPlaneCorrespondenceSearch& point_to_plane_corr_search)
      : point_to_point_corr_search_(point_to_point_corr_search), point_to_plane_corr_search_(point_to_plane_corr_search) {}

  bool sameEngine() const { return (const void*)&point_to_point_corr_search_ == (const void*)&point_to_plane_corr_search_; }
  CorrespondenceSearchCombinedMetricCombiner& findCorrespondences() { return findCorrespondences(RigidTransform3f::Identity()); }
  template <class TransformT>
  CorrespondenceSearchCombinedMetricCombiner& findCorrespondences(const TransformT& tform) {      // :46-58
    point_to_point_corr_search_.findCorrespondences(tform);
    if (!sameEngine()) point_to_plane_corr_search_.findCorrespondences(tform);
    return *this;
  }
  const PointToPointCorrespondenceSearchResult& getPointToPointCorrespondences() const { return point_to_point_corr_search_.getCorrespondences(); }
  const PointToPlaneCorrespondenceSearchResult& getPointToPlaneCorrespondences() const { return point_to_plane_corr_search_.getCorrespondences(); }
  PointToPointCorrespondenceSearch& pointToPointCorrespondenceSearchEngine() { return point_to_point_corr_search_; }
  PointToPlaneCorrespondenceSearch& pointToPlaneCorrespondenceSearchEngine() { return point_to_plane_corr_search_; }

private:
  PointToPointCorrespondenceSearch& point_to_point_corr_search_;
  PointToPlaneCorrespondenceSearch& point_to_plane_corr_search_;
};

// CombinedMetricSingleTransformICP handed its engine (icp_single_transform_combined_metric.hpp:8-243, rigid 3-D instance) -- here
// a Combiner over two CorrespondenceSearchHIP engines: cilhip_icp_run_two_sets runs the loop (both searches, the estimator over
// the two sets, compose, convergence test per iteration).  Defaults :44-47, icp_base.hpp:24-25.
template <class CombinerT>
class CombinedMetricRigidICP3f {
public:
  explicit CombinedMetricRigidICP3f(CombinerT& combiner) : combiner_(combiner) {}
  CombinerT& correspondenceSearchEngine() { return combiner_; }
  CombinedMetricRigidICP3f& setMaxNumberOfIterations(size_t n) { max_iterations_ = n; return *this; }
  CombinedMetricRigidICP3f& setConvergenceTolerance(float t) { convergence_tol_ = t; return *this; }
  CombinedMetricRigidICP3f& setPointToPointMetricWeight(float w) { point_to_point_weight_ = w; return *this; }
  CombinedMetricRigidICP3f& setPointToPlaneMetricWeight(float w) { point_to_plane_weight_ = w; return *this; }
  CombinedMetricRigidICP3f& setMaxNumberOfOptimizationStepIterations(size_t n) { max_optimization_iterations_ = n; return *this; }
  CombinedMetricRigidICP3f& setOptimizationStepConvergenceTolerance(float t) { optimization_convergence_tol_ = t; return *this; }
  template <class TransformT>
  CombinedMetricRigidICP3f& setInitialTransform(const TransformT& t) { internal::to_abi(t, transform_init_.m); return *this; }
  CombinedMetricRigidICP3f& estimate() {
    cilhip_icp_params p;
    cilhip_icp_default_params(&p);
    p.metric = CILHIP_METRIC_COMBINED;
    p.w_p2p = point_to_point_weight_; p.w_p2pl = point_to_plane_weight_;
    p.max_iter = max_iterations_; p.conv_tol = convergence_tol_;
    p.max_opt_iter = max_optimization_iterations_; p.opt_conv_tol = optimization_convergence_tol_;
    auto& e1 = combiner_.pointToPointCorrespondenceSearchEngine();
    auto& e2 = combiner_.pointToPlaneCorrespondenceSearchEngine();
    cilhip_icp_result r;
    internal::check(e1.context(), cilhip_icp_run_two_sets(e1.context(), e1.getMaxDistance(), e2.context(), e2.getMaxDistance(), &p, transform_init_.m, &r),
                    "estimate (two correspondence sets)");
    e1.invalidateFetched(); e2.invalidateFetched();
    std::memcpy(transform_.m, r.T, sizeof(r.T));
    iterations_ = r.iterations; last_delta_norm_ = r.last_delta_norm;
    return *this;
  }
  const RigidTransform3f& getTransform() const { return transform_; }
  size_t getNumberOfPerformedIterations() const { return iterations_; }
  float getLastUpdateNorm() const { return last_delta_norm_; }
  bool hasConverged() const { return last_delta_norm_ < convergence_tol_; }

private:
  CombinerT& combiner_;
  size_t max_iterations_ = 15, iterations_ = 0, max_optimization_iterations_ = 1;
  float convergence_tol_ = 1e-5f, optimization_convergence_tol_ = 1e-5f, point_to_point_weight_ = 0.0f, point_to_plane_weight_ = 1.0f;
  float last_delta_norm_ = std::numeric_limits<float>::infinity();
  RigidTransform3f transform_init_, transform_;
};

// The sharded ICP loop over several devices of one process (c_api.h: cilhip_multi_*): per iteration every device's context searches
// and accumulates over its shard, RCCL all-reduces the 48 partial sums on the devices' streams, every context applies them.
// SlabSharded (default): spatial slabs of target (+ halo) and source, exact nearest neighbours without any key exchange, the
// device-side guard and re-partitioning inside; SourceShards: the source in contiguous shards, the target on every device; TargetShards: the
// target in index shards (a target that does not fit one device), the whole source everywhere, a MIN all-reduce of packed keys per iteration.  The
// reference has no counterpart (single-process CPU); the loop is IterativeClosestPointBase::estimate (icp_base.hpp:68-87) for the
// rigid combined / point-to-point metrics with their defaults (icp_single_transform_combined_metric.hpp:44-47).
class MultiDeviceRigidICP {
public:
  enum struct Partition { SourceShards = 0, Slabs = 1, TargetShards = 2 };      // TargetShards: index shards of the target, MIN of packed keys per iteration
  explicit MultiDeviceRigidICP(const std::vector<int>& devices) {
    if (cilhip_multi_create(&m_, devices.data(), (int)devices.size()) != CILHIP_OK)
      throw std::runtime_error("cilhip_multi_create failed (no usable HIP devices / RCCL for these ordinals; there is no CPU fallback)");
  }
  ~MultiDeviceRigidICP() { cilhip_multi_destroy(m_); }
  MultiDeviceRigidICP(const MultiDeviceRigidICP&) = delete;
  MultiDeviceRigidICP& operator=(const MultiDeviceRigidICP&) = delete;

  // dst_n may be empty (point-to-point metric).  max_sq_dist: the engines' squared search radius (it sizes the slabs' halos).
  MultiDeviceRigidICP& setClouds(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p, float max_sq_dist,
                                 Partition part = Partition::Slabs) {
    if (dst_n.cols() && dst_n.cols() != dst_p.cols()) throw std::invalid_argument("dst normals must match dst points");
    ck(cilhip_multi_set_clouds(m_, dst_p.data(), dst_n.cols() ? dst_n.data() : nullptr, dst_p.cols(), src_p.data(), src_p.cols(), max_sq_dist, (int)part, nullptr));
    max_sq_ = max_sq_dist;
    return *this;
  }
  MultiDeviceRigidICP& setMaxNumberOfIterations(size_t n) { max_iterations_ = n; return *this; }
  MultiDeviceRigidICP& setConvergenceTolerance(float t) { convergence_tol_ = t; return *this; }
  MultiDeviceRigidICP& setPointToPointMetricWeight(float w) { w_p2p_ = w; return *this; }
  MultiDeviceRigidICP& setPointToPlaneMetricWeight(float w) { w_p2pl_ = w; return *this; }
  MultiDeviceRigidICP& usePointToPointMetric(bool b) { p2p_ = b; return *this; }      // SimplePointToPointMetricRigidICP3f instead of the combined metric
  template <class TransformT>
  MultiDeviceRigidICP& setInitialTransform(const TransformT& t) { internal::to_abi(t, transform_init_.m); return *this; }
  MultiDeviceRigidICP& estimate() {
    cilhip_icp_params p;
    cilhip_icp_default_params(&p);
    p.metric = p2p_ ? CILHIP_METRIC_POINT_TO_POINT : CILHIP_METRIC_COMBINED;
    p.w_p2p = w_p2p_; p.w_p2pl = w_p2pl_; p.max_iter = max_iterations_; p.conv_tol = convergence_tol_; p.max_sq_dist = max_sq_;
    cilhip_icp_result r;
    ck(cilhip_multi_icp_run(m_, &p, transform_init_.m, 0, &r));
    std::memcpy(transform_.m, r.T, sizeof(r.T));
    iterations_ = r.iterations; last_delta_norm_ = r.last_delta_norm; last_ncorr_ = r.last_ncorr;
    return *this;
  }
  const RigidTransform3f& getTransform() const { return transform_; }
  size_t getNumberOfPerformedIterations() const { return iterations_; }
  float getLastUpdateNorm() const { return last_delta_norm_; }
  bool hasConverged() const { return last_delta_norm_ < convergence_tol_; }
  size_t getNumberOfLastCorrespondences() const { return last_ncorr_; }
  int getNumberOfRepartitions() const { return cilhip_multi_repartitions(m_); }
  cilhip_ctx* context(int rank) { return cilhip_multi_context(m_, rank); }

private:
  void ck(int rc) const { if (rc != CILHIP_OK) throw std::runtime_error(std::string("MultiDeviceRigidICP: ") + cilhip_multi_last_error(m_)); }
  cilhip_multi* m_ = nullptr;
  size_t max_iterations_ = 15, iterations_ = 0, last_ncorr_ = 0;
  float convergence_tol_ = 1e-5f, w_p2p_ = 0.0f, w_p2pl_ = 1.0f, max_sq_ = 0.01f * 0.01f, last_delta_norm_ = std::numeric_limits<float>::infinity();
  bool p2p_ = false;
  RigidTransform3f transform_init_, transform_;
};

typedef RigidTransform3f AffineTransform3f;

class SimplePointToPointMetricAffineICP3f : public SimplePointToPointMetricRigidICP3f {
public:
  SimplePointToPointMetricAffineICP3f(const ConstPointsView& dst, const ConstPointsView& src, int device = 0)
      : SimplePointToPointMetricRigidICP3f(dst, src, device) {
    internal::check(context(), cilhip_set_option(context(), "transform_mode", 1.0), "transform_mode");
  }
};

class SimpleCombinedMetricAffineICP3f : public SimpleCombinedMetricRigidICP3f {
public:
  SimpleCombinedMetricAffineICP3f(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p, int device = 0)
      : SimpleCombinedMetricRigidICP3f(dst_p, dst_n, src_p, device) {
    internal::check(context(), cilhip_set_option(context(), "transform_mode", 1.0), "transform_mode");
  }
  // four-cloud form: accepted like the reference's wrapper, but the symmetric objective exists for the rigid instances
  // only (icp_single_transform_combined_metric.hpp:180-204): the source normals are not used
  SimpleCombinedMetricAffineICP3f(const ConstPointsView& dst_p, const ConstPointsView& dst_n, const ConstPointsView& src_p,
                                  const ConstPointsView& /*src_n*/, int device = 0)
      : SimpleCombinedMetricAffineICP3f(dst_p, dst_n, src_p, device) {}
};

#ifdef CILANTRO_HIP_HAVE_EIGEN
// Adaptors for cilantro's own types (compiled only where Eigen3 exists; it is absent from the build
// container, so these are exercised by downstream builds, not by this repository's tests).
template <class Derived>
inline ConstPointsView view(const Eigen::DenseBase<Derived>& m) {
  static_assert(Derived::RowsAtCompileTime == 3, "3xN matrix expected");
  return ConstPointsView(m.derived().data(), static_cast<size_t>(m.cols()));
}
inline Eigen::Transform<float, 3, Eigen::Isometry> toEigen(const RigidTransform3f& t) {
  Eigen::Transform<float, 3, Eigen::Isometry> e;
  std::memcpy(e.data(), t.m, sizeof(t.m));
  return e;
}
#endif

}  // namespace cilantro_hip
