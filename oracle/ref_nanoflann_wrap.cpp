// ref_nanoflann_wrap.cpp -- ORACLE / TEST INFRASTRUCTURE ONLY.
//
// Thin C wrapper that compiles the REFERENCE's OWN vendored nanoflann 1.7.1 from where it lies
// (/root/reference/include/cilantro/3rd_party/nanoflann/nanoflann.hpp, passed with -I; nothing
// is copied into this repository) into oracle/_ref/libref_nanoflann.so.
//
// cilantro's own wrapper (core/kd_tree.hpp) cannot be compiled here because it pulls in Eigen3
// (absent from the container), so the ~40 lines of Eigen-typed glue are restated
// Eigen-free below, each citing the lines it follows:
//   * data adaptor           core/kd_tree.hpp:11-37   (kdtree_get_pt(idx,dim) = obj(dim,idx))
//   * k-NN result set        core/kd_tree.hpp:63-109  (KNNSearchResultAdaptor)
//   * tree construction      core/kd_tree.hpp:162-170 (leaf 10, 1 build thread, eps 0, sorted)
//   * kNNInRadiusSearch      core/kd_tree.hpp:283-291
//   * the OMP query loop     correspondence_search/correspondence_search_kd_tree_utilities.hpp:7-51
// Used (a) to pin oracle/icp_oracle.c's kd-tree restatement, (b) as the "reference" CPU baseline
// for the kNN pass in bench.py.
#include <nanoflann.hpp>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

namespace {

struct PointsAdaptor {                       // core/kd_tree.hpp:11-37
  const float* data; size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(size_t idx, size_t dim) const { return data[3 * idx + dim]; }
  template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};

using Metric = nanoflann::L2_Adaptor<float, PointsAdaptor, float, size_t>;   // kd_tree.hpp:46-47
using Tree = nanoflann::KDTreeSingleIndexAdaptor<Metric, PointsAdaptor, 3, size_t>;

struct Neighbor { size_t index; float value; };

class KNNResult {                            // core/kd_tree.hpp:63-109
 public:
  using DistanceType = float;
  using IndexType = size_t;
  KNNResult(std::vector<Neighbor>& r, size_t k, float max_radius) : r_(r), k_(k), count_(0) {
    r_.resize(k_);
    r_[k_ - 1].value = max_radius;
  }
  size_t size() const { return count_; }
  bool full() const { return count_ == k_; }
  bool addPoint(float dist, size_t index) {
    size_t i;
    for (i = count_; i > 0; --i) {
      if (r_[i - 1].value > dist) {
        if (i < k_) { r_[i].index = r_[i - 1].index; r_[i].value = r_[i - 1].value; }
      } else break;
    }
    if (i < k_) { r_[i].index = index; r_[i].value = dist; }
    if (count_ < k_) count_++;
    return true;
  }
  float worstDist() const { return r_[k_ - 1].value; }
  void sort() const {}
 private:
  std::vector<Neighbor>& r_;
  const size_t k_;
  size_t count_;
};

template <int D>
struct AdaptorN {                            // the same data adaptor over row-major n x D features (D = 6: point + normal or
  const float* data; size_t n;               // point + colour, D = 9: point + normal + colour)
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(size_t idx, size_t dim) const { return data[D * idx + dim]; }
  template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
template <int D> using MetricN = nanoflann::L2_Adaptor<float, AdaptorN<D>, float, size_t>;
template <int D> using TreeN = nanoflann::KDTreeSingleIndexAdaptor<MetricN<D>, AdaptorN<D>, D, size_t>;

struct RefTree {
  PointsAdaptor adaptor;
  Tree tree;
  nanoflann::SearchParameters params;
  RefTree(const float* p, size_t n)
      : adaptor{p, n},
        tree(3, adaptor, nanoflann::KDTreeSingleIndexAdaptorParams(
                             10, nanoflann::KDTreeSingleIndexAdaptorFlags::None, 1)),
        params(0.0f, true) {}
};

// The same loop over D-dimensional features (PointNormalFeaturesAdaptor / PointColorFeaturesAdaptor: D = 6,
// PointNormalColorFeaturesAdaptor: D = 9; common_transformable_feature_adaptors.hpp:60-343): the reference's kd-tree instantiated
// for that DIM on row-major n x D feature arrays (tree built here, one call).
template <int D>
static size_t ref_find_correspondences_n(const float* dstf, size_t nd, const float* qf, size_t nq, float max_d,
                                         int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads) {
  if (nd == 0) return 0;
  AdaptorN<D> ad{dstf, nd};
  TreeN<D> tree(D, ad, nanoflann::KDTreeSingleIndexAdaptorParams(10, nanoflann::KDTreeSingleIndexAdaptorFlags::None, 1));
  nanoflann::SearchParameters params(0.0f, true);
  std::vector<int64_t> tmp_idx(nq);
  std::vector<float> tmp_d2(nq);
  std::vector<char> keep(nq);
  std::vector<Neighbor> nn;
#pragma omp parallel for private(nn) schedule(dynamic, 256) num_threads(num_threads)
  for (size_t i = 0; i < nq; i++) {
    KNNResult sra(nn, 1, max_d);
    tree.findNeighbors(sra, qf + (size_t)D * i, params);
    nn.resize(sra.size());
    float dist = 0.f;
    keep[i] = !nn.empty() && (dist = nn[0].value) < max_d;
    if (keep[i]) { tmp_idx[i] = (int64_t)nn[0].index; tmp_d2[i] = dist; }
  }
  size_t count = 0;
  for (size_t i = 0; i < nq; i++)
    if (keep[i]) { dst_idx[count] = tmp_idx[i]; src_idx[count] = (int64_t)i; d2[count] = tmp_d2[i]; ++count; }
  return count;
}

}  // namespace

extern "C" {

void* ref_kdtree_build(const float* pts_xyz, size_t n) { return new RefTree(pts_xyz, n); }
void ref_kdtree_free(void* t) { delete static_cast<RefTree*>(t); }

size_t ref_kdtree_knn_in_radius(const void* tp, const float q[3], size_t k, float radius_sq,
                                size_t* out_idx, float* out_d2) {
  const RefTree* t = static_cast<const RefTree*>(tp);
  if (t->adaptor.n == 0 || k == 0) return 0;
  std::vector<Neighbor> nn;
  KNNResult sra(nn, k, radius_sq);
  t->tree.findNeighbors(sra, q, t->params);
  for (size_t i = 0; i < sra.size(); ++i) { out_idx[i] = nn[i].index; out_d2[i] = nn[i].value; }
  return sra.size();
}

// KDTree::kNNInRadiusSearch for many queries (core/kd_tree.hpp:283-291 in the loop of e.g. core/normal_estimation.hpp:294-420): the
// reference's own knnSearch through cilantro's result adaptor -- its order among EXACTLY equal distances (first met stays ahead,
// at every slot and at the k-th place) is what the engine's k-NN is held to.  idx / d2: nq x k, -1 / +inf padded; cnt: nq or null.
void ref_kdtree_knn_batch(const void* tp, const float* q, size_t nq, size_t k, float radius_sq, int64_t* idx, float* d2, uint32_t* cnt,
                          int num_threads) {
  const RefTree* t = static_cast<const RefTree*>(tp);
  std::vector<Neighbor> nn;
#pragma omp parallel for private(nn) schedule(dynamic, 256) num_threads(num_threads)
  for (size_t i = 0; i < nq; i++) {
    size_t m = 0;
    if (t->adaptor.n != 0 && k != 0) {
      KNNResult sra(nn, k, radius_sq);
      t->tree.findNeighbors(sra, q + 3 * i, t->params);
      m = sra.size();
    }
    for (size_t j = 0; j < k; ++j) {
      idx[i * k + j] = j < m ? (int64_t)nn[j].index : -1;
      d2[i * k + j] = j < m ? nn[j].value : std::numeric_limits<float>::infinity();
    }
    if (cnt) cnt[i] = (uint32_t)m;
  }
}

// correspondence_search_kd_tree_utilities.hpp:7-51, ref_is_first, identity evaluator
size_t ref_find_correspondences(const void* tp, const float* q, size_t nq, float max_d,
                                int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads) {
  const RefTree* t = static_cast<const RefTree*>(tp);
  if (t->adaptor.n == 0) return 0;
  std::vector<int64_t> tmp_idx(nq);
  std::vector<float> tmp_d2(nq);
  std::vector<char> keep(nq);
  std::vector<Neighbor> nn;
#pragma omp parallel for private(nn) schedule(dynamic, 256) num_threads(num_threads)
  for (size_t i = 0; i < nq; i++) {
    KNNResult sra(nn, 1, max_d);
    t->tree.findNeighbors(sra, q + 3 * i, t->params);
    nn.resize(sra.size());
    float dist = 0.f;
    keep[i] = !nn.empty() && (dist = nn[0].value) < max_d;
    if (keep[i]) { tmp_idx[i] = (int64_t)nn[0].index; tmp_d2[i] = dist; }
  }
  size_t count = 0;
  for (size_t i = 0; i < nq; i++)
    if (keep[i]) { dst_idx[count] = tmp_idx[i]; src_idx[count] = (int64_t)i; d2[count] = tmp_d2[i]; ++count; }
  return count;
}

// KDTree::radiusSearch for one query (core/kd_tree.hpp:251-257 with RadiusSearchResultAdaptor :111-142), then the
// std::sort by value of :254; returns the number of neighbours (written up to `cap`).
size_t ref_kdtree_radius_search(const void* tp, const float q[3], float radius_sq, size_t* out_idx, float* out_d2, size_t cap) {
  const RefTree* t = static_cast<const RefTree*>(tp);
  if (t->adaptor.n == 0) return 0;
  struct RadiusResult {
    using DistanceType = float;
    using IndexType = size_t;
    std::vector<Neighbor>& r; float radius;
    size_t size() const { return r.size(); }
    bool full() const { return true; }
    bool addPoint(float dist, size_t index) { r.push_back(Neighbor{index, dist}); return true; }
    float worstDist() const { return radius; }
    void sort() const {}
  };
  std::vector<Neighbor> nn;
  RadiusResult rr{nn, radius_sq};
  t->tree.findNeighbors(rr, q, t->params);
  std::sort(nn.begin(), nn.end(), [](const Neighbor& a, const Neighbor& b) { return a.value < b.value; });
  for (size_t i = 0; i < nn.size() && i < cap; ++i) { out_idx[i] = nn[i].index; out_d2[i] = nn[i].value; }
  return nn.size();
}

size_t ref_find_correspondences6(const float* dst6, size_t nd, const float* q6, size_t nq, float max_d,
                                 int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads) {
  return ref_find_correspondences_n<6>(dst6, nd, q6, nq, max_d, dst_idx, src_idx, d2, num_threads);
}
size_t ref_find_correspondences9(const float* dst9, size_t nd, const float* q9, size_t nq, float max_d,
                                 int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads) {
  return ref_find_correspondences_n<9>(dst9, nd, q9, nq, max_d, dst_idx, src_idx, d2, num_threads);
}

}  // extern "C"
