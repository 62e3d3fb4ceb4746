// tie_order_host.hpp -- TEST INFRASTRUCTURE (the CPU cross-check of cilantro_amd/csrc/tie_build.hip; the product builds these tables on the
// device and never includes this file) -- which of several EXACTLY equidistant target points the reference would return (option "tie_rule").
//
// The engine's nearest neighbour is the brute-force argmin of the pinned f32 squared distance.  The reference keeps the candidate
// its kd-tree traversal meets FIRST (core/kd_tree.hpp:82-90: a strict '<' insert into the k = 1 result set; nanoflann's
// searchLevel, 3rd_party/nanoflann/nanoflann.hpp:1885-1961, descends into the child on the query's side of a split first and walks
// a leaf in the order of its slice of vAcc_).  Both are exact nearest neighbours; they name different points only where two or
// more target points are at exactly the smallest distance (duplicated points, a depth sensor's lattice).
//
// To name the reference's choice for those (rare) queries the ORDER of that traversal is needed, i.e. the tree itself.  This file
// produces the ORDER TABLES of the index nanoflann 1.7.1 builds for the reference's parameters (leaf_max_size 10, core/kd_tree.hpp
// :162-170; computeBoundingBox :1846-1877, divideTree :1150-1212, middleSplit_ :1321-1372, planeSplit :1383-1428):
//   per point   its leaf and its slot in the reference's permutation vAcc_,
//   per node    its parent, its depth, whether it is the SECOND child, and the split (dimension, divlow, divhigh),
// which is all it takes to answer "which of two points does a query's traversal reach first": at the lowest common ancestor of
// their leaves the child on the query's side ((val - divlow) + (val - divhigh) < 0: the first child) is visited first; inside one
// leaf the lower slot.  The search itself is not replayed: the candidates are already known to be exact nearest points (the device
// found them).  The device evaluates the same tables (kernels.hip: tie_before); first_met() below is the host form of it.
//
// The permutation a node leaves depends only on the order of its own slice when it is reached and on the box handed down to it, so
// sub-trees are independent: they are built by a pool of threads (a node's slice is split by ONE thread with the reference's
// two-pass sweep -- that sweep IS the contract --, its children go back to the pool), the coordinates travel with the indices
// (16-byte records permuted in place: sequential sweeps instead of gathers through vAcc_).  divlow / divhigh are the tight bounds
// the reference reads off its children's bounding boxes after the recursion (:1196-1205): the largest coordinate of the first
// child's points and the smallest of the second's along the split dimension -- formed here right after the split.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace cilhip {

struct TieNode {          // 16 bytes: one load on the device
  int32_t parent;         // -1: the root
  uint32_t info;          // (depth << 3) | (split dimension << 1) | (1: this node is its parent's SECOND child)
  float divlow, divhigh;  // internal nodes: the split; a leaf: divlow = the slot of its first point (as bits)
};

class TieOrderTree {
 public:
  // xyz: the target in its ORIGINAL order (index i at xyz[3 i ..]), as the reference's adaptor presents it (core/kd_tree.hpp:11-37)
  void build(const float* xyz, uint32_t n, uint32_t leaf_max = 10, unsigned threads = 0) {
    n_ = n; leaf_max_ = leaf_max;
    nodes_.clear();
    leaf_of_.assign(n, 0); slot_of_.assign(n, 0);
    built_ = true;
    if (n == 0) return;
    if (threads == 0) threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (n < 200000u) threads = 1;
    threads_ = threads;
    recs_.resize(n);
    Box box;
    {      // init_vind + computeBoundingBox (min / max do not depend on the order they are taken in)
      const unsigned parts = threads > 1 ? threads : 1;
      std::vector<Box> pb(parts);
      chunks(n, parts, [&](unsigned k, uint32_t lo, uint32_t hi) {
        Box b;
        for (int d = 0; d < 3; ++d) { b.lo[d] = INFINITY; b.hi[d] = -INFINITY; }
        for (uint32_t i = lo; i < hi; ++i) {
          Rec& r = recs_[i];
          r.c[0] = xyz[3 * (size_t)i]; r.c[1] = xyz[3 * (size_t)i + 1]; r.c[2] = xyz[3 * (size_t)i + 2]; r.idx = i;
          for (int d = 0; d < 3; ++d) { if (r.c[d] < b.lo[d]) b.lo[d] = r.c[d]; if (r.c[d] > b.hi[d]) b.hi[d] = r.c[d]; }
        }
        pb[k] = b;
      });
      box = pb[0];
      for (unsigned k = 1; k < parts; ++k)
        for (int d = 0; d < 3; ++d) { if (pb[k].lo[d] < box.lo[d]) box.lo[d] = pb[k].lo[d]; if (pb[k].hi[d] > box.hi[d]) box.hi[d] = pb[k].hi[d]; }
    }
    // every worker appends to its own node list (ids local to the list); the lists are concatenated afterwards
    std::vector<std::vector<BNode>> lists(threads);
    Pool pool;
    pool.pending = 1;
    pool.q.push_back(Task{0, n, box, Ref{-1, 0}, 0, false});
    const uint32_t grain = std::max<uint32_t>(4096u, n / (threads * 16u));
    auto worker = [&](unsigned tid) {
      std::vector<BNode>& mine = lists[tid];
      Task t;
      while (pool.pop(t)) {
        // depth-first over this task's sub-tree; children that are still large go back to the pool
        std::vector<Task> stack;
        stack.push_back(t);
        while (!stack.empty()) {
          Task cur = stack.back();
          stack.pop_back();
          const int32_t id = (int32_t)mine.size();
          mine.push_back(BNode{cur.parent, cur.depth, cur.second, 0, 0.0f, 0.0f, cur.left, cur.right, true});
          if (cur.right - cur.left <= leaf_max_) continue;
          Task c1, c2;
          split(cur, mine[id], c1, c2);
          c1.parent = c2.parent = Ref{(int32_t)tid, id};
          // (second child first on the stack: the first child is built next, as the reference's recursion does -- no effect on the result)
          Task* kids[2] = {&c2, &c1};
          for (Task* k : kids) {
            if (threads > 1 && k->right - k->left > grain) pool.push(*k);
            else stack.push_back(*k);
          }
        }
        pool.done_one();
      }
    };
    if (threads == 1) worker(0);
    else {
      std::vector<std::thread> th;
      for (unsigned k = 0; k < threads; ++k) th.emplace_back(worker, k);
      for (auto& x : th) x.join();
    }
    // concatenate the lists; leaf tables
    std::vector<uint32_t> base(threads + 1, 0);
    for (unsigned k = 0; k < threads; ++k) base[k + 1] = base[k] + (uint32_t)lists[k].size();
    nodes_.resize(base[threads]);
    chunks(threads, threads, [&](unsigned k, uint32_t, uint32_t) {      // (a list's leaves cover slices no other list touches)
      for (size_t j = 0; j < lists[k].size(); ++j) {
        const BNode& b = lists[k][j];
        const uint32_t id = base[k] + (uint32_t)j;
        TieNode& nd = nodes_[id];
        nd.parent = b.parent.list < 0 ? -1 : (int32_t)(base[b.parent.list] + (uint32_t)b.parent.id);
        nd.info = ((uint32_t)b.depth << 3) | ((uint32_t)b.feat << 1) | (b.second ? 1u : 0u);
        nd.divlow = b.divlow; nd.divhigh = b.divhigh;
        if (b.leaf) {
          for (uint32_t s = b.left; s < b.right; ++s) { leaf_of_[recs_[s].idx] = id; slot_of_[recs_[s].idx] = s; }
          memcpy(&nd.divlow, &b.left, 4);      // (a leaf has no split: the field holds the slot of its first point, as bits -- slot - that = place inside the leaf)
        }
      }
    });
    recs_.clear(); recs_.shrink_to_fit();
  }
  bool built() const { return built_; }
  uint32_t size() const { return n_; }
  const std::vector<TieNode>& nodes() const { return nodes_; }
  const std::vector<uint32_t>& leaf_of() const { return leaf_of_; }      // by ORIGINAL target index
  const std::vector<uint32_t>& slot_of() const { return slot_of_; }      // ... its slot in the reference's vAcc_

  // q: the (transformed) query; cand: ORIGINAL target indices, all at the same distance from q.  Returns the one met first.
  uint32_t first_met(const float q[3], const uint32_t* cand, int k) const {
    uint32_t best = cand[0];
    for (int c = 1; c < k; ++c)
      if (cand[c] != best && before(q, cand[c], best)) best = cand[c];
    return best;
  }

 private:
  struct Rec { float c[3]; uint32_t idx; };
  struct Box { float lo[3], hi[3]; };
  struct Ref { int32_t list, id; };
  struct Task { uint32_t left, right; Box box; Ref parent; int32_t depth; bool second; };
  struct BNode { Ref parent; int32_t depth; bool second; int32_t feat; float divlow, divhigh; uint32_t left, right; bool leaf; };
  struct Pool {
    std::mutex m; std::condition_variable cv; std::deque<Task> q; size_t pending = 0;      // pending: tasks queued or being worked on
    void push(const Task& t) { { std::lock_guard<std::mutex> g(m); q.push_back(t); ++pending; } cv.notify_one(); }
    bool pop(Task& t) {
      std::unique_lock<std::mutex> g(m);
      cv.wait(g, [&] { return !q.empty() || pending == 0; });
      if (q.empty()) return false;
      t = q.front(); q.pop_front();
      return true;
    }
    void done_one() { bool last; { std::lock_guard<std::mutex> g(m); last = --pending == 0; } if (last) cv.notify_all(); }
  };

  // the reference's split of one node's slice: dimension (middleSplit_), two-pass sweep (planeSplit), split index; fills the node
  // and the two children's slices and (loose) boxes
  void split(const Task& t, BNode& nd, Task& c1, Task& c2) {
    Rec* const v = recs_.data() + t.left;
    const uint32_t count = t.right - t.left;
    const Box& box = t.box;
    const float EPS = 0.00001f;
    float max_span = box.hi[0] - box.lo[0];
    for (int d = 1; d < 3; ++d) { const float span = box.hi[d] - box.lo[d]; if (span > max_span) max_span = span; }
    float max_spread = -1.0f, min_elem = 0.0f, max_elem = 0.0f;
    int feat = 0;
    for (int d = 0; d < 3; ++d) {
      const float span = box.hi[d] - box.lo[d];
      if (span >= (1 - EPS) * max_span) {
        float mn, mx;      // computeMinMax
        min_max(v, count, d, mn, mx);
        const float spread = mx - mn;
        if (spread > max_spread) { feat = d; max_spread = spread; min_elem = mn; max_elem = mx; }
      }
    }
    const float split_val = (box.lo[feat] + box.hi[feat]) / 2;
    const float cutval = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    // planeSplit: afterwards [0, lim1) < cutval, [lim1, lim2) == cutval, [lim2, count) > cutval
    uint32_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && v[left].c[feat] < cutval) ++left;
      while (right && left <= right && v[right].c[feat] >= cutval) --right;
      if (left > right || !right) break;
      std::swap(v[left], v[right]);
      ++left; --right;
    }
    const uint32_t lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && v[left].c[feat] <= cutval) ++left;
      while (right && left <= right && v[right].c[feat] > cutval) --right;
      if (left > right || !right) break;
      std::swap(v[left], v[right]);
      ++left; --right;
    }
    const uint32_t lim2 = left;
    const uint32_t idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
    // the children's tight bounds along the split dimension (what the reference reads off their boxes after the recursion)
    float lo_hi, hi_lo, unused;
    min_max(v, idx, feat, unused, lo_hi);
    min_max(v + idx, count - idx, feat, hi_lo, unused);
    nd.leaf = false; nd.feat = feat; nd.divlow = lo_hi; nd.divhigh = hi_lo;
    c1 = Task{t.left, t.left + idx, box, Ref{0, 0}, t.depth + 1, false};
    c1.box.hi[feat] = cutval;
    c2 = Task{t.left + idx, t.right, box, Ref{0, 0}, t.depth + 1, true};
    c2.box.lo[feat] = cutval;
  }

  // fn(part, lo, hi) over [0, n) cut into `parts` ranges, one thread each (the caller's thread takes the first)
  template <typename F>
  static void chunks(uint32_t n, unsigned parts, F fn) {
    if (parts <= 1) { fn(0u, 0u, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = ((uint64_t)n + parts - 1) / parts;
    for (unsigned k = 1; k < parts; ++k) {
      const uint32_t lo = (uint32_t)std::min<uint64_t>(per * k, n), hi = (uint32_t)std::min<uint64_t>(per * (k + 1), n);
      th.emplace_back([=] { fn(k, lo, hi); });      // (an empty range is the callee's business)
    }
    fn(0u, 0u, (uint32_t)std::min<uint64_t>(per, n));
    for (auto& x : th) x.join();
  }
  // smallest and largest coordinate d of a slice; the few huge slices near the root are swept by several threads (exact: min / max)
  void min_max(const Rec* v, uint32_t count, int d, float& mn, float& mx) const {
    const unsigned parts = (threads_ > 1 && count >= (1u << 21)) ? std::min(threads_, 16u) : 1u;
    if (parts == 1) {
      mn = mx = v[0].c[d];
      for (uint32_t k = 1; k < count; ++k) { const float x = v[k].c[d]; if (x < mn) mn = x; if (x > mx) mx = x; }
      return;
    }
    std::vector<float> lo(parts), hi(parts);
    chunks(count, parts, [&](unsigned p, uint32_t a, uint32_t b) {
      float l = INFINITY, h = -INFINITY;
      for (uint32_t k = a; k < b; ++k) { const float x = v[k].c[d]; if (x < l) l = x; if (x > h) h = x; }
      lo[p] = l; hi[p] = h;
    });
    mn = lo[0]; mx = hi[0];
    for (unsigned p = 1; p < parts; ++p) { if (lo[p] < mn) mn = lo[p]; if (hi[p] > mx) mx = hi[p]; }
  }

  // does the traversal of query q reach point a before point b?
  bool before(const float q[3], uint32_t a, uint32_t b) const {
    uint32_t na = leaf_of_[a], nb = leaf_of_[b];
    if (na == nb) return slot_of_[a] < slot_of_[b];
    uint32_t a_second = 0;      // is the node on a's path below the common ancestor a SECOND child
    while ((nodes_[na].info >> 3) > (nodes_[nb].info >> 3)) { a_second = nodes_[na].info & 1u; na = (uint32_t)nodes_[na].parent; }
    while ((nodes_[nb].info >> 3) > (nodes_[na].info >> 3)) nb = (uint32_t)nodes_[nb].parent;
    while (na != nb) { a_second = nodes_[na].info & 1u; na = (uint32_t)nodes_[na].parent; nb = (uint32_t)nodes_[nb].parent; }
    const TieNode& nd = nodes_[na];
    const float val = q[(nd.info >> 1) & 3u];
    const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    const uint32_t first_is_second = (diff1 + diff2) < 0 ? 0u : 1u;      // searchLevel: bestChild
    return a_second == first_is_second;
  }

  // Where point `a` comes in query q's traversal, as ONE number (smaller = met earlier): per level from the root one bit -- 0 = the node on
  // a's path is the child searchLevel descends into first -- most significant first, then a's place inside its leaf.  Comparing two
  // points' keys is before(): the paths share their bits down to the lowest common ancestor and differ right below it.  What the device
  // publishes per query when a target is held in index shards (kernels.hip: tie_rank): the MIN over the shards is the first-met point of
  // the whole target.  Holds max_depth() <= 58 levels and leaves of up to 16 points.
 public:
  uint64_t traversal_key(const float q[3], uint32_t a) const {
    uint32_t n = leaf_of_[a];
    uint32_t left; memcpy(&left, &nodes_[n].divlow, 4);
    uint64_t key = (uint64_t)((slot_of_[a] - left) & 15u);
    while ((nodes_[n].info >> 3) != 0u) {
      const TieNode& p = nodes_[(uint32_t)nodes_[n].parent];
      const float val = q[(p.info >> 1) & 3u];
      const float diff1 = val - p.divlow, diff2 = val - p.divhigh;
      const uint32_t first_is_second = (diff1 + diff2) < 0 ? 0u : 1u;
      if ((nodes_[n].info & 1u) != first_is_second) key |= 1ull << (62u - (nodes_[n].info >> 3));
      n = (uint32_t)nodes_[n].parent;
    }
    return key;
  }
  uint32_t max_depth() const { uint32_t d = 0; for (const TieNode& t : nodes_) d = std::max(d, t.info >> 3); return d; }
 private:

  uint32_t n_ = 0, leaf_max_ = 10;
  unsigned threads_ = 1;
  bool built_ = false;
  std::vector<Rec> recs_;      // the reference's vAcc_ with the coordinates alongside (build only)
  std::vector<TieNode> nodes_;
  std::vector<uint32_t> leaf_of_, slot_of_;
};

}  // namespace cilhip
