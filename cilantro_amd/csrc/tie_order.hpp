// tie_order.hpp -- which of several EXACTLY equidistant target points the reference would return (host side, option "tie_rule" = 1).
//
// The engine's nearest neighbour is the brute-force argmin of the pinned f32 squared distance with the lowest target index on ties.
// The reference keeps the candidate its kd-tree traversal meets FIRST (core/kd_tree.hpp:82-90: a strict '<' insert into the k = 1
// result set; nanoflann's searchLevel, 3rd_party/nanoflann/nanoflann.hpp:1885-1961, descends into the child on the query's side of a
// split first and walks a leaf in the order of its slice of vAcc_).  Both are exact nearest neighbours; they name different points
// only where two or more target points are at exactly the smallest distance (duplicated points, a depth sensor's lattice).
//
// To reproduce the reference's choice for those (rare) queries the ORDER of that traversal is needed, i.e. the tree itself: this
// file builds the index nanoflann 1.7.1 builds for the reference's parameters (leaf_max_size 10, one build thread, core/kd_tree.hpp
// :162-170) -- computeBoundingBox (:1846-1877), divideTree (:1150-1212), middleSplit_ (:1321-1372), planeSplit (:1383-1428) -- keeps
// per point its leaf and its slot in vAcc_, per node its parent, and answers "which of these candidates does a query meet first":
// at the lowest common ancestor of two candidates' leaves the child on the query's side ((val - divlow) + (val - divhigh) < 0: child1)
// is visited first; inside one leaf the lower vAcc_ slot.  Nothing else of the search is replayed: the candidates handed in are
// already known to be the exact nearest points (the device found them).
#pragma once

#include <stdint.h>

#include <algorithm>
#include <vector>

namespace cilhip {

class TieOrderTree {
 public:
  // xyz: the target in its ORIGINAL order (index i at xyz[3 i ..]), as the reference's adaptor presents it (core/kd_tree.hpp:11-37)
  void build(const float* xyz, uint32_t n, uint32_t leaf_max = 10) {
    pts_ = xyz; n_ = n; leaf_max_ = leaf_max;
    order_.resize(n);
    for (uint32_t i = 0; i < n; ++i) order_[i] = i;      // init_vind
    nodes_.clear();
    leaf_of_.assign(n, 0); slot_of_.assign(n, 0);
    if (n == 0) return;
    Box box;
    for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = at(order_[0], d);
    for (uint32_t k = 1; k < n; ++k)
      for (int d = 0; d < 3; ++d) { const float v = at(order_[k], d); if (v < box.lo[d]) box.lo[d] = v; if (v > box.hi[d]) box.hi[d] = v; }
    divide(0, n, box, -1);
    for (uint32_t s = 0; s < n; ++s) slot_of_[order_[s]] = s;
  }
  bool built() const { return !nodes_.empty() || n_ == 0; }
  uint32_t size() const { return n_; }

  // q: the (transformed) query; cand: ORIGINAL target indices, all at the same distance from q.  Returns the one met first.
  uint32_t first_met(const float q[3], const uint32_t* cand, int k) const {
    uint32_t best = cand[0];
    for (int c = 1; c < k; ++c)
      if (cand[c] != best && before(q, cand[c], best)) best = cand[c];
    return best;
  }

 private:
  struct Box { float lo[3], hi[3]; };
  struct Node { int32_t parent, child1, child2, depth; int32_t feat; float divlow, divhigh; };

  float at(uint32_t idx, int d) const { return pts_[3 * (size_t)idx + d]; }

  // planeSplit: on return  [0, lim1) < cutval,  [lim1, lim2) == cutval,  [lim2, count) > cutval   (two Hoare-style passes)
  void plane_split(uint32_t ind, uint32_t count, int feat, float cutval, uint32_t& lim1, uint32_t& lim2) {
    uint32_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && at(order_[ind + left], feat) < cutval) ++left;
      while (right && left <= right && at(order_[ind + right], feat) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(order_[ind + left], order_[ind + right]);
      ++left; --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && at(order_[ind + left], feat) <= cutval) ++left;
      while (right && left <= right && at(order_[ind + right], feat) > cutval) --right;
      if (left > right || !right) break;
      std::swap(order_[ind + left], order_[ind + right]);
      ++left; --right;
    }
    lim2 = left;
  }

  // divideTree over order_[left, right); box: in = the node's box from above, out = the tight box of its points' sub-boxes
  int32_t divide(uint32_t left, uint32_t right, Box& box, int32_t parent) {
    const int32_t id = (int32_t)nodes_.size();
    nodes_.push_back(Node{parent, -1, -1, parent < 0 ? 0 : nodes_[parent].depth + 1, 0, 0.0f, 0.0f});
    if (right - left <= leaf_max_) {
      for (uint32_t k = left; k < right; ++k) leaf_of_[order_[k]] = (uint32_t)id;
      for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = at(order_[left], d);
      for (uint32_t k = left + 1; k < right; ++k)
        for (int d = 0; d < 3; ++d) { const float v = at(order_[k], d); if (box.lo[d] > v) box.lo[d] = v; if (box.hi[d] < v) box.hi[d] = v; }
      return id;
    }
    // middleSplit_: the dimension of largest point spread among those whose box span is within 1e-5 of the largest
    const uint32_t count = right - left;
    const float EPS = 0.00001f;
    float max_span = box.hi[0] - box.lo[0];
    for (int d = 1; d < 3; ++d) { const float span = box.hi[d] - box.lo[d]; if (span > max_span) max_span = span; }
    float max_spread = -1.0f, min_elem = 0.0f, max_elem = 0.0f;
    int feat = 0;
    for (int d = 0; d < 3; ++d) {
      const float span = box.hi[d] - box.lo[d];
      if (span >= (1 - EPS) * max_span) {
        float mn = at(order_[left], d), mx = mn;      // computeMinMax
        for (uint32_t k = 1; k < count; ++k) { const float v = at(order_[left + k], d); if (v < mn) mn = v; if (v > mx) mx = v; }
        const float spread = mx - mn;
        if (spread > max_spread) { feat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
      }
    }
    const float split_val = (box.lo[feat] + box.hi[feat]) / 2;
    const float cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    uint32_t lim1, lim2;
    plane_split(left, count, feat, cutval, lim1, lim2);
    const uint32_t idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
    Box lbox = box, rbox = box;
    lbox.hi[feat] = cutval;
    const int32_t c1 = divide(left, left + idx, lbox, id);
    rbox.lo[feat] = cutval;
    const int32_t c2 = divide(left + idx, right, rbox, id);
    Node& nd = nodes_[id];
    nd.child1 = c1; nd.child2 = c2; nd.feat = feat; nd.divlow = lbox.hi[feat]; nd.divhigh = rbox.lo[feat];
    for (int d = 0; d < 3; ++d) { box.lo[d] = std::min(lbox.lo[d], rbox.lo[d]); box.hi[d] = std::max(lbox.hi[d], rbox.hi[d]); }
    return id;
  }

  // does the traversal of query q reach point a before point b?
  bool before(const float q[3], uint32_t a, uint32_t b) const {
    int32_t na = (int32_t)leaf_of_[a], nb = (int32_t)leaf_of_[b];
    if (na == nb) return slot_of_[a] < slot_of_[b];
    int32_t ca = na, cb = nb;      // the children of the common ancestor on the two paths
    while (nodes_[na].depth > nodes_[nb].depth) { ca = na; na = nodes_[na].parent; }
    while (nodes_[nb].depth > nodes_[na].depth) { cb = nb; nb = nodes_[nb].parent; }
    while (na != nb) { ca = na; cb = nb; na = nodes_[na].parent; nb = nodes_[nb].parent; }
    const Node& nd = nodes_[na];
    const float val = q[nd.feat];
    const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    const int32_t first = (diff1 + diff2) < 0 ? nd.child1 : nd.child2;      // searchLevel: bestChild
    (void)cb;
    return ca == first;
  }

  const float* pts_ = nullptr;
  uint32_t n_ = 0, leaf_max_ = 10;
  std::vector<uint32_t> order_;      // vAcc_
  std::vector<Node> nodes_;
  std::vector<uint32_t> leaf_of_, slot_of_;
};

}  // namespace cilhip
