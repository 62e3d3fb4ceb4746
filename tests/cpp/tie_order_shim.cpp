// Test shim: tests/cpp/tie_order_host.hpp (the host restatement of the reference's build: the CPU cross-check of the device build in
// csrc/tie_build.hip) behind a few C functions, so that the CPU suite can check the order tables against the reference's nanoflann
// (oracle/_ref) without a GPU, and the GPU suite the device's tables against these.  Built by tests/cpp/build.sh into tests/cpp/bin/libtie_order_shim.so.
#include "tie_order_host.hpp"

#include <cstring>

struct Shim { std::vector<float> xyz; cilhip::TieOrderTree tree; };

extern "C" {
void* tie_shim_build_mt(const float* xyz, uint32_t n, unsigned threads) {
  Shim* s = new Shim();
  s->xyz.assign(xyz, xyz + 3 * (size_t)n);
  s->tree.build(s->xyz.data(), n, 10, threads);
  return s;
}
void* tie_shim_build(const float* xyz, uint32_t n) { return tie_shim_build_mt(xyz, n, 1); }
void tie_shim_free(void* h) { delete static_cast<Shim*>(h); }
// per query k: cand[k*stride .. +count[k]) -> out[k]
void tie_shim_first_met(void* h, const float* q, const uint32_t* cand, const int* count, int stride, uint32_t nq, uint32_t* out) {
  const Shim* s = static_cast<const Shim*>(h);
  for (uint32_t k = 0; k < nq; ++k) out[k] = s->tree.first_met(q + 3 * (size_t)k, cand + (size_t)k * stride, count[k]);
}
// per query k: out[k] = the candidate with the smallest traversal key (what the MIN over index shards of a target leaves); *depth = the tree's
void tie_shim_min_key(void* h, const float* q, const uint32_t* cand, const int* count, int stride, uint32_t nq, uint32_t* out, uint32_t* depth) {
  const Shim* s = static_cast<const Shim*>(h);
  for (uint32_t k = 0; k < nq; ++k) {
    uint64_t best = ~0ull;
    for (int j = 0; j < count[k]; ++j) {
      const uint64_t key = s->tree.traversal_key(q + 3 * (size_t)k, cand[(size_t)k * stride + j]);
      if (key < best) { best = key; out[k] = cand[(size_t)k * stride + j]; }
    }
  }
  *depth = s->tree.max_depth();
}
// out[k] = traversal key of target point idx[k] for query k
void tie_shim_keys(void* h, const float* q, const uint32_t* idx, uint32_t nq, uint64_t* out) {
  const Shim* s = static_cast<const Shim*>(h);
  for (uint32_t k = 0; k < nq; ++k) out[k] = s->tree.traversal_key(q + 3 * (size_t)k, idx[k]);
}
// the reference's permutation (slot of every point), leaf populations and depths: what must not depend on how many threads built the tree
// (node ids do: they are list positions)
int tie_shim_same_order(void* ha, void* hb) {
  const Shim *a = static_cast<const Shim*>(ha), *b = static_cast<const Shim*>(hb);
  if (a->tree.size() != b->tree.size() || a->tree.nodes().size() != b->tree.nodes().size()) return 0;
  if (a->tree.slot_of() != b->tree.slot_of()) return 0;
  for (uint32_t i = 0; i < a->tree.size(); ++i) {
    // walk both leaf-to-root paths: same depths, same splits, same child sides
    uint32_t na = a->tree.leaf_of()[i], nb = b->tree.leaf_of()[i];
    for (;;) {
      const cilhip::TieNode &x = a->tree.nodes()[na], &y = b->tree.nodes()[nb];
      if (x.info != y.info) return 0;
      if ((x.parent < 0) != (y.parent < 0)) return 0;
      if (x.parent < 0) break;
      na = (uint32_t)x.parent; nb = (uint32_t)y.parent;
      const cilhip::TieNode &px = a->tree.nodes()[na], &py = b->tree.nodes()[nb];
      if (std::memcmp(&px.divlow, &py.divlow, 4) || std::memcmp(&px.divhigh, &py.divhigh, 4)) return 0;
    }
  }
  return 1;
}
// the same comparison against RAW tables (the device build's, copied out by cilhip_tie_order_tables): 1 = the same permutation and
// the same leaf-to-root path for every point; otherwise 0 and *first_bad = a point that differs
int tie_shim_same_as_tables(void* ha, uint32_t n, const uint32_t* leaf, const uint32_t* slot, const cilhip::TieNode* nodes, size_t n_nodes, uint32_t* first_bad) {
  const Shim* a = static_cast<const Shim*>(ha);
  *first_bad = 0xFFFFFFFFu;
  if (a->tree.size() != n || a->tree.nodes().size() != n_nodes) return 0;
  for (uint32_t i = 0; i < n; ++i) {
    bool ok = a->tree.slot_of()[i] == slot[i];
    uint32_t na = a->tree.leaf_of()[i], nb = leaf[i];
    if (ok && nb >= n_nodes) ok = false;
    // a leaf's record holds the slot of its first point
    if (ok && std::memcmp(&a->tree.nodes()[na].divlow, &nodes[nb].divlow, 4)) ok = false;
    while (ok) {
      const cilhip::TieNode &x = a->tree.nodes()[na], &y = nodes[nb];
      if (x.info != y.info || (x.parent < 0) != (y.parent < 0)) { ok = false; break; }
      if (x.parent < 0) break;
      na = (uint32_t)x.parent; nb = (uint32_t)y.parent;
      if (nb >= n_nodes) { ok = false; break; }
      const cilhip::TieNode &px = a->tree.nodes()[na], &py = nodes[nb];
      if (std::memcmp(&px.divlow, &py.divlow, 4) || std::memcmp(&px.divhigh, &py.divhigh, 4)) ok = false;
    }
    if (!ok) { *first_bad = i; return 0; }
  }
  return 1;
}
}
