// tie_build.hip -- the order tables of the reference's kd-tree (option "tie_rule"), built ON THE DEVICE.
//
// tests/cpp/tie_order_host.hpp states WHAT the tables are and builds them on the host with the reference's own sequential sweeps (test
// infrastructure: the CPU cross-check, itself pinned against the reference's nanoflann by tests/test_tie_order_cpu.py).  This file produces the same tables -- the same permutation slot for slot, the same
// splits -- level by level on the GPU: nanoflann's divideTree (3rd_party/nanoflann/nanoflann.hpp:1150-1212) recurses, but a node's
// result depends only on the order of its own slice when it is reached and on the box handed down to it, so ALL nodes of a level are
// independent segments of one array and a level is a handful of segmented passes over 16-byte records:
//
//   middleSplit_ (:1321-1372)   the cut dimension needs min / max of the slice per dimension (computeMinMax): one segmented reduction
//                               per level (rocPRIM reduce_by_key over the records' node keys), also the source of the children's tight
//                               bounds divlow / divhigh (:1196-1205);
//   planeSplit  (:1383-1428)    a two-pointer sweep, parallel in disguise.  First pass: with L = "value < cut", the sweep swaps the k-th
//                               element that is NOT L from the left with the k-th L from the right while the former lies left of the
//                               latter -- i.e. the k-th misplaced non-L (position < lim1 = #L, ascending) trades places with the k-th
//                               misplaced L (position >= lim1, descending); everything else stays.  One exclusive scan of the flags gives
//                               every element its rank, a scatter through two rank -> position tables moves the records.  Second pass: the
//                               same over [lim1, count) with "value <= cut".  (The sweep's `right != 0` guards only end it; they never
//                               leave an element on the wrong side: checked case by case in DESIGN 6.10.)
//
// Node ids are breadth-first here (the host build numbers depth-first per worker): ids are labels -- tie_before() follows parent links
// -- so the tables agree with the host's up to that relabelling; tests/test_gpu_tie_rule.py compares slot for slot and path for path.
#include "internal.hpp"

#include <hip/hip_runtime.h>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <vector>

namespace cilhip {

namespace {

constexpr uint32_t LEAF_MAX = 10;      // core/kd_tree.hpp:162-170
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int TB = 256;

struct MM { float mn[3], mx[3]; };
struct MMOp {
  __host__ __device__ MM operator()(const MM& a, const MM& b) const {
    MM r;
    for (int d = 0; d < 3; ++d) { r.mn[d] = a.mn[d] < b.mn[d] ? a.mn[d] : b.mn[d]; r.mx[d] = a.mx[d] > b.mx[d] ? a.mx[d] : b.mx[d]; }
    return r;
  }
};
struct ToMM {
  __host__ __device__ MM operator()(const float4& p) const { return MM{{p.x, p.y, p.z}, {p.x, p.y, p.z}}; }
};

// an internal node of the current level (more than LEAF_MAX points)
struct Act {
  uint32_t left, right;      // its slice of the record array
  float blo[3], bhi[3];      // the box handed down to it (loose: the parent's box cut at the parent's cutval)
  MM mm;                     // min / max of its points per dimension
  uint32_t node, depth;      // its TieNode id
  int feat; float cut;       // middleSplit_'s choice
  uint32_t lim1, lim2, idx;  // planeSplit's limits, the split index
};

__device__ __forceinline__ float coord(const float4& p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

__global__ void k_init_recs(const float* __restrict__ xyz, uint32_t n, float4* __restrict__ recs, uint32_t* __restrict__ node_of) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    recs[i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
    node_of[i] = 0u;
  }
}
// records given as {x, y, z, bits(original index)} in ANY order (a grid's sorted points): back to the original order
__global__ void k_init_recs_from_sorted(const float4* __restrict__ sorted, uint32_t n, float4* __restrict__ recs, uint32_t* __restrict__ node_of) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const float4 p = sorted[j];
    const uint32_t i = __float_as_uint(p.w);
    if (i < n) recs[i] = p;
    node_of[j] = 0u;
  }
}

// the root: computeBoundingBox (:1846-1877) = min / max over all points (the one run of the first reduce_by_key)
__global__ void k_root(const MM* __restrict__ agg, uint32_t n, Act* __restrict__ act, uint32_t* __restrict__ counts, uint4* __restrict__ nodes) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Act a{};
  a.left = 0; a.right = n; a.mm = agg[0]; a.node = 0; a.depth = 0;
  for (int d = 0; d < 3; ++d) { a.blo[d] = a.mm.mn[d]; a.bhi[d] = a.mm.mx[d]; }
  const bool leaf = n <= LEAF_MAX;
  counts[0] = leaf ? 0u : 1u;      // active nodes of level 0
  counts[1] = 1u;                  // nodes so far
  if (!leaf) act[0] = a;
  nodes[0] = make_uint4(0xFFFFFFFFu, 0u, leaf ? 0u : 0u, 0u);      // parent -1, depth 0; a leaf root: z = slot of its first point = 0
}

// middleSplit_: the cut dimension and value of every active node
__global__ void k_decide(Act* __restrict__ act, const uint32_t* __restrict__ counts) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= counts[0]) return;
  Act& A = act[a];
  const float EPS = 0.00001f;
  float max_span = __fsub_rn(A.bhi[0], A.blo[0]);
  for (int d = 1; d < 3; ++d) { const float span = __fsub_rn(A.bhi[d], A.blo[d]); if (span > max_span) max_span = span; }
  const float thr = __fmul_rn(__fsub_rn(1.0f, EPS), max_span);
  float max_spread = -1.0f, min_elem = 0.0f, max_elem = 0.0f;
  int feat = 0;
  for (int d = 0; d < 3; ++d) {
    const float span = __fsub_rn(A.bhi[d], A.blo[d]);
    if (span >= thr) {
      const float spread = __fsub_rn(A.mm.mx[d], A.mm.mn[d]);
      if (spread > max_spread) { feat = d; max_spread = spread; min_elem = A.mm.mn[d]; max_elem = A.mm.mx[d]; }
    }
  }
  const float split_val = __fadd_rn(A.blo[feat], A.bhi[feat]) / 2;
  A.feat = feat;
  A.cut = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
}

// flags of a planeSplit pass: PASS 1 -- "not (value < cut)" over the node's slice; PASS 2 -- "value > cut" over [lim1, count)
template <int PASS>
__global__ void k_flags(const float4* __restrict__ recs, const uint32_t* __restrict__ node_of, const Act* __restrict__ act, uint32_t n, uint32_t* __restrict__ flags) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gridDim.x * blockDim.x) {
    uint32_t f = 0;
    if (p < n) {
      const uint32_t a = node_of[p];
      if (a != NONE) {
        const Act& A = act[a];
        const float v = coord(recs[p], A.feat);
        f = PASS == 1 ? (v < A.cut ? 0u : 1u) : ((p - A.left >= A.lim1 && v > A.cut) ? 1u : 0u);
      }
    }
    flags[p] = f;      // (n + 1 entries: the scan's last value is the total)
  }
}

// ranks: S = exclusive scan of the flags.  Inside the pass's range [lo, hi) of a node: nF = flagged elements, nU = the others = the
// limit; a flagged element at relative position < nU is misplaced (rank = flagged elements before it), an unflagged one at >= nU is
// (rank from the right = unflagged elements behind it); rank -> position tables for the scatter.
template <int PASS>
__global__ void k_ranks(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ S, const uint32_t* __restrict__ node_of, Act* __restrict__ act, uint32_t n,
                        uint32_t* __restrict__ posF, uint32_t* __restrict__ posU) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t a = node_of[p];
    if (a == NONE) continue;
    Act& A = act[a];
    const uint32_t lo = PASS == 1 ? A.left : A.left + A.lim1, hi = A.right;
    if (p < lo) continue;
    const uint32_t nF = S[hi] - S[lo], nU = (hi - lo) - nF;
    if (p == lo) { if (PASS == 1) A.lim1 = nU; else A.lim2 = A.lim1 + nU; }
    if (PASS == 2 && p == A.left && lo != A.left) { /* (written by the thread at lo) */ }
    const uint32_t rel = p - lo, Fb = S[p] - S[lo];
    if (flags[p]) { if (rel < nU) posF[lo + Fb] = p; }
    else if (rel >= nU) { const uint32_t Ub = rel - Fb; posU[lo + (nU - Ub - 1u)] = p; }
  }
}
// (a pass whose range is EMPTY -- lim1 == count cannot happen, see k_children -- never reaches p == lo: lim2 is preset there)

template <int PASS>
__global__ void k_scatter(const float4* __restrict__ in, float4* __restrict__ out, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ S,
                          const uint32_t* __restrict__ node_of, const Act* __restrict__ act, uint32_t n, const uint32_t* __restrict__ posF, const uint32_t* __restrict__ posU) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    uint32_t dest = p;
    const uint32_t a = node_of[p];
    if (a != NONE) {
      const Act& A = act[a];
      const uint32_t lo = PASS == 1 ? A.left : A.left + A.lim1, hi = A.right;
      if (p >= lo) {
        const uint32_t nF = S[hi] - S[lo], nU = (hi - lo) - nF;
        const uint32_t rel = p - lo, Fb = S[p] - S[lo];
        if (flags[p]) { if (rel < nU) dest = posU[lo + Fb]; }
        else if (rel >= nU) { const uint32_t Ub = rel - Fb; dest = posF[lo + (nU - Ub - 1u)]; }
      }
    }
    out[dest] = in[p];
  }
}

// the split index (:1169-1176 via middleSplit_ :1360-1371), the children's ids (breadth-first: 2 per active node, in the nodes' order)
// and which of them go on (more than LEAF_MAX points): internal[2a + c]
__global__ void k_children(Act* __restrict__ act, const uint32_t* __restrict__ counts, uint32_t* __restrict__ internal) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= counts[0]) { if (a == counts[0]) { internal[2 * a] = 0u; } return; }      // (one entry behind the last: the scan's total)
  Act& A = act[a];
  const uint32_t count = A.right - A.left;
  const uint32_t idx = A.lim1 > count / 2 ? A.lim1 : (A.lim2 < count / 2 ? A.lim2 : count / 2);
  A.idx = idx;
  internal[2 * a] = idx > LEAF_MAX ? 1u : 0u;
  internal[2 * a + 1] = (count - idx) > LEAF_MAX ? 1u : 0u;
}

// per record: the key of the child it now belongs to (2a + c), for the children's min / max
__global__ void k_child_keys(const uint32_t* __restrict__ node_of, const Act* __restrict__ act, uint32_t n, uint32_t* __restrict__ keys) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t a = node_of[p];
    keys[p] = a == NONE ? NONE : 2u * a + (p >= act[a].left + act[a].idx ? 1u : 0u);
  }
}
__global__ void k_store_mm(const uint32_t* __restrict__ uk, const MM* __restrict__ agg, const uint32_t* __restrict__ nruns, uint32_t cap, MM* __restrict__ child_mm) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *nruns || r >= cap) return;
  if (uk[r] != NONE) child_mm[uk[r]] = agg[r];
}

// the node's record (split + the children's tight bounds along it, :1196-1205), the children's records, the next level's active nodes
__global__ void k_finish(const Act* __restrict__ act, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ internal_scan, const MM* __restrict__ child_mm,
                         uint4* __restrict__ nodes, uint32_t node_cap, Act* __restrict__ act_next, uint32_t* __restrict__ counts_next) {
  const uint32_t na = counts[0];
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a == 0) { counts_next[0] = internal_scan[2 * na]; counts_next[1] = counts[1] + 2u * na; }
  if (a >= na) return;
  const Act& A = act[a];
  const uint32_t base = counts[1];      // ids of this level's children start here
  const MM m1 = child_mm[2 * a], m2 = child_mm[2 * a + 1];
  if (A.node < node_cap) {
    uint4 nd = nodes[A.node];
    nd.y = (A.depth << 3) | ((uint32_t)A.feat << 1) | (nd.y & 1u);
    nd.z = __float_as_uint(m1.mx[A.feat]);      // divlow
    nd.w = __float_as_uint(m2.mn[A.feat]);      // divhigh
    nodes[A.node] = nd;
  }
  for (uint32_t c = 0; c < 2; ++c) {
    const uint32_t id = base + 2u * a + c;
    const uint32_t left = c == 0 ? A.left : A.left + A.idx, right = c == 0 ? A.left + A.idx : A.right;
    const bool internal = (right - left) > LEAF_MAX;
    if (id < node_cap) nodes[id] = make_uint4(A.node, ((A.depth + 1u) << 3) | c, internal ? 0u : left, 0u);      // (a leaf: z = the slot of its first point)
    if (internal) {
      Act B{};
      B.left = left; B.right = right; B.node = id; B.depth = A.depth + 1u;
      for (int d = 0; d < 3; ++d) { B.blo[d] = A.blo[d]; B.bhi[d] = A.bhi[d]; }
      if (c == 0) B.bhi[A.feat] = A.cut; else B.blo[A.feat] = A.cut;
      B.mm = c == 0 ? m1 : m2;
      B.lim1 = B.lim2 = B.idx = 0;
      act_next[internal_scan[2 * a + c]] = B;
    }
  }
}

// every record of this level's nodes moves to its child: the next level's active index, or -- a leaf -- its final leaf and slot
__global__ void k_descend(const float4* __restrict__ recs, uint32_t* __restrict__ node_of, const Act* __restrict__ act, const uint32_t* __restrict__ counts,
                          const uint32_t* __restrict__ internal_scan, const uint32_t* __restrict__ internal, uint32_t n, uint32_t* __restrict__ leaf_by_index,
                          uint32_t* __restrict__ slot_by_index) {
  const uint32_t base = counts[1];
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t a = node_of[p];
    if (a == NONE) continue;
    const uint32_t c = p >= act[a].left + act[a].idx ? 1u : 0u;
    if (internal[2 * a + c]) node_of[p] = internal_scan[2 * a + c];
    else {
      node_of[p] = NONE;
      const uint32_t i = __float_as_uint(recs[p].w);
      leaf_by_index[i] = base + 2u * a + c;
      slot_by_index[i] = p;
    }
  }
}
// a cloud of at most LEAF_MAX points: the root is the only leaf
__global__ void k_root_leaf(const float4* __restrict__ recs, uint32_t n, uint32_t* __restrict__ leaf_by_index, uint32_t* __restrict__ slot_by_index) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t i = __float_as_uint(recs[p].w);
  leaf_by_index[i] = 0u; slot_by_index[i] = p;
}

#define TB_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)

}  // namespace

// d_xyz: the cloud in its ORIGINAL order (3 floats per point), or d_sorted: {x, y, z, bits(original index)} records in any order (one
// of the two).  d_leaf_by_index / d_slot_by_index: [n], by ORIGINAL index, as TieOrderTree::leaf_of() / slot_of().  *d_nodes_out: the
// TieNode records (hipMalloc'ed here, the caller frees), *n_nodes_out how many; *max_depth_out the deepest node's depth.
hipError_t tie_order_build_device(const float* d_xyz, const float4* d_sorted, uint32_t n, hipStream_t s, uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index,
                                  uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out) {
  *d_nodes_out = nullptr; *n_nodes_out = 0; if (max_depth_out) *max_depth_out = 0;
  if (n == 0) return hipSuccess;
  hipError_t e = hipSuccess;
  const unsigned gp = (unsigned)std::min<size_t>(((size_t)n + TB) / TB, 65535u * 4u);
  // active nodes of a level hold more than LEAF_MAX points each; a level's children: twice that; all nodes: bounded by 2n (leaves of one
  // point), in practice ~0.3 n -- the node array grows by doubling when a level does not fit
  const uint32_t act_cap = n / (LEAF_MAX + 1) + 2;
  size_t node_cap = std::max<size_t>(1024, (size_t)n / 2 + 64);
  float4 *recA = nullptr, *recB = nullptr;
  uint32_t *node_of = nullptr, *flags = nullptr, *S = nullptr, *posF = nullptr, *posU = nullptr, *keys = nullptr, *uk = nullptr, *internal = nullptr, *iscan = nullptr;
  uint32_t *counts = nullptr, *nruns = nullptr;
  MM *agg = nullptr, *child_mm = nullptr;
  Act *act0 = nullptr, *act1 = nullptr;
  uint4* nodes = nullptr;
  void *tmp = nullptr, *ws = nullptr;
  size_t tmp_bytes = 0;
  uint32_t h_counts[2] = {0, 0};
  int depth = 0;
  {
    const size_t run_cap = 4 * (size_t)act_cap + 4;
    {      // scratch of the scans / the reduction: the largest request
      size_t b1 = 0, b2 = 0, b3 = 0;
      TB_TRY(rocprim::exclusive_scan(nullptr, b1, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s));
      TB_TRY(rocprim::exclusive_scan(nullptr, b2, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 2 * (size_t)act_cap + 1, rocprim::plus<uint32_t>(), s));
      auto vals = rocprim::make_transform_iterator((const float4*)nullptr, ToMM());
      TB_TRY(rocprim::reduce_by_key(nullptr, b3, (uint32_t*)nullptr, vals, (size_t)n, (uint32_t*)nullptr, (MM*)nullptr, (uint32_t*)nullptr, MMOp(), rocprim::equal_to<uint32_t>(), s));
      tmp_bytes = std::max(b1, std::max(b2, b3));
    }
    // ONE allocation for the whole workspace (twenty hipMalloc / hipFree pairs were a third of a small cloud's build)
    {
      size_t off = 0;
      auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
      const size_t o_recA = take((size_t)n * sizeof(float4)), o_recB = take((size_t)n * sizeof(float4)), o_node = take((size_t)n * 4), o_flags = take(((size_t)n + 1) * 4),
                   o_S = take(((size_t)n + 1) * 4), o_posF = take((size_t)n * 4), o_posU = take((size_t)n * 4), o_keys = take((size_t)n * 4), o_uk = take(run_cap * 4),
                   o_agg = take(run_cap * sizeof(MM)), o_cmm = take((2 * (size_t)act_cap + 2) * sizeof(MM)), o_int = take((2 * (size_t)act_cap + 2) * 4),
                   o_iscan = take((2 * (size_t)act_cap + 2) * 4), o_counts = take(16), o_nruns = take(4), o_act0 = take((size_t)act_cap * sizeof(Act)),
                   o_act1 = take((size_t)act_cap * sizeof(Act)), o_tmp = take(tmp_bytes ? tmp_bytes : 16);
      TB_TRY(hipMalloc(&ws, off));
      unsigned char* b = static_cast<unsigned char*>(ws);
      recA = (float4*)(b + o_recA); recB = (float4*)(b + o_recB); node_of = (uint32_t*)(b + o_node); flags = (uint32_t*)(b + o_flags); S = (uint32_t*)(b + o_S);
      posF = (uint32_t*)(b + o_posF); posU = (uint32_t*)(b + o_posU); keys = (uint32_t*)(b + o_keys); uk = (uint32_t*)(b + o_uk); agg = (MM*)(b + o_agg);
      child_mm = (MM*)(b + o_cmm); internal = (uint32_t*)(b + o_int); iscan = (uint32_t*)(b + o_iscan); counts = (uint32_t*)(b + o_counts); nruns = (uint32_t*)(b + o_nruns);
      act0 = (Act*)(b + o_act0); act1 = (Act*)(b + o_act1); tmp = b + o_tmp;
    }
    TB_TRY(hipMalloc(&nodes, node_cap * sizeof(uint4)));
    if (d_sorted) hipLaunchKernelGGL(k_init_recs_from_sorted, dim3(gp), dim3(TB), 0, s, d_sorted, n, recA, node_of);
    else hipLaunchKernelGGL(k_init_recs, dim3(gp), dim3(TB), 0, s, d_xyz, n, recA, node_of);
    // the root's box: one run of key 0
    TB_TRY(hipMemsetAsync(keys, 0, (size_t)n * 4, s));
    {
      auto vals = rocprim::make_transform_iterator(recA, ToMM());
      size_t b = tmp_bytes;
      TB_TRY(rocprim::reduce_by_key(tmp, b, keys, vals, (size_t)n, uk, agg, nruns, MMOp(), rocprim::equal_to<uint32_t>(), s));
    }
    hipLaunchKernelGGL(k_root, dim3(1), dim3(64), 0, s, agg, n, act0, counts, nodes);
    TB_TRY(hipMemcpyAsync(h_counts, counts, 8, hipMemcpyDeviceToHost, s));
    TB_TRY(hipStreamSynchronize(s));
    if (h_counts[0] == 0) hipLaunchKernelGGL(k_root_leaf, dim3((n + TB - 1) / TB), dim3(TB), 0, s, recA, n, d_leaf_by_index, d_slot_by_index);
    Act *cur = act0, *nxt = act1;
    uint32_t *cnt_cur = counts, *cnt_nxt = counts + 2;
    while (h_counts[0] != 0) {
      const uint32_t na = h_counts[0];
      if ((size_t)h_counts[1] + 2 * (size_t)na > node_cap) {      // the node array grows (degenerate clouds: many tiny leaves)
        const size_t ncap = std::max(node_cap * 2, (size_t)h_counts[1] + 2 * (size_t)na + 64);
        uint4* nn = nullptr;
        TB_TRY(hipMalloc(&nn, ncap * sizeof(uint4)));
        TB_TRY(hipMemcpyAsync(nn, nodes, (size_t)h_counts[1] * sizeof(uint4), hipMemcpyDeviceToDevice, s));
        TB_TRY(hipStreamSynchronize(s));
        (void)hipFree(nodes); nodes = nn; node_cap = ncap;
      }
      const unsigned ga = (na + TB) / TB + 1;
      hipLaunchKernelGGL(k_decide, dim3(ga), dim3(TB), 0, s, cur, cnt_cur);
      // planeSplit, first pass: recA -> recB
      hipLaunchKernelGGL(k_flags<1>, dim3(gp), dim3(TB), 0, s, recA, node_of, cur, n, flags);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, flags, S, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL(k_ranks<1>, dim3(gp), dim3(TB), 0, s, flags, S, node_of, cur, n, posF, posU);
      hipLaunchKernelGGL(k_scatter<1>, dim3(gp), dim3(TB), 0, s, recA, recB, flags, S, node_of, cur, n, posF, posU);
      // second pass: recB -> recA
      hipLaunchKernelGGL(k_flags<2>, dim3(gp), dim3(TB), 0, s, recB, node_of, cur, n, flags);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, flags, S, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL(k_ranks<2>, dim3(gp), dim3(TB), 0, s, flags, S, node_of, cur, n, posF, posU);
      hipLaunchKernelGGL(k_scatter<2>, dim3(gp), dim3(TB), 0, s, recB, recA, flags, S, node_of, cur, n, posF, posU);
      // children
      hipLaunchKernelGGL(k_children, dim3(ga), dim3(TB), 0, s, cur, cnt_cur, internal);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, internal, iscan, 0u, 2 * (size_t)na + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL(k_child_keys, dim3(gp), dim3(TB), 0, s, node_of, cur, n, keys);
      {
        auto vals = rocprim::make_transform_iterator(recA, ToMM());
        size_t b = tmp_bytes;
        TB_TRY(rocprim::reduce_by_key(tmp, b, keys, vals, (size_t)n, uk, agg, nruns, MMOp(), rocprim::equal_to<uint32_t>(), s));
      }
      {
        const uint32_t rc = (uint32_t)std::min<size_t>(4 * (size_t)na + 4, 4 * (size_t)act_cap + 4);
        hipLaunchKernelGGL(k_store_mm, dim3((rc + TB - 1) / TB), dim3(TB), 0, s, uk, agg, nruns, rc, child_mm);
      }
      hipLaunchKernelGGL(k_finish, dim3(ga), dim3(TB), 0, s, cur, cnt_cur, iscan, child_mm, nodes, (uint32_t)std::min<size_t>(node_cap, 0xFFFFFFFFu), nxt, cnt_nxt);
      hipLaunchKernelGGL(k_descend, dim3(gp), dim3(TB), 0, s, recA, node_of, cur, cnt_cur, iscan, internal, n, d_leaf_by_index, d_slot_by_index);
      TB_TRY(hipGetLastError());
      TB_TRY(hipMemcpyAsync(h_counts, cnt_nxt, 8, hipMemcpyDeviceToHost, s));
      TB_TRY(hipStreamSynchronize(s));
      std::swap(cur, nxt); std::swap(cnt_cur, cnt_nxt);
      ++depth;
      if (depth > 200) { e = hipErrorUnknown; goto done; }      // (cannot happen: every split leaves both children non-empty)
    }
    TB_TRY(hipGetLastError());
    TB_TRY(hipStreamSynchronize(s));
    *d_nodes_out = nodes; nodes = nullptr;
    *n_nodes_out = h_counts[1];
    if (max_depth_out) *max_depth_out = depth;
  }
done:
  if (ws) (void)hipFree(ws);
  if (nodes) (void)hipFree(nodes);
  return e;
}

}  // namespace cilhip
