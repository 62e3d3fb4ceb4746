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

// DIM = 3: the point tree of the ICP / k-NN searches; DIM = 6 / 9: the trees the reference's feature adaptors search (points + weighted
// normals or colours; points + normals + colours: correspondence_search/common_transformable_feature_adaptors.hpp:60-343 through a
// KDTree of that dimension).  The same build: nanoflann's code is generic in DIM.
template <int DIM> struct Rec { float c[DIM]; uint32_t idx; };      // a record of the reference's vAcc_ with the coordinates alongside (DIM = 3: a float4)
template <int DIM> struct MM { float mn[DIM], mx[DIM]; };
template <int DIM> struct MMOp {
  __host__ __device__ MM<DIM> operator()(const MM<DIM>& a, const MM<DIM>& b) const {
    MM<DIM> r;
    for (int d = 0; d < DIM; ++d) { r.mn[d] = a.mn[d] < b.mn[d] ? a.mn[d] : b.mn[d]; r.mx[d] = a.mx[d] > b.mx[d] ? a.mx[d] : b.mx[d]; }
    return r;
  }
};
template <int DIM> struct ToMM {
  __host__ __device__ MM<DIM> operator()(const Rec<DIM>& p) const { MM<DIM> m; for (int d = 0; d < DIM; ++d) m.mn[d] = m.mx[d] = p.c[d]; return m; }
};
// TieNode::info: (depth << SHIFT) | (split dimension << 1) | second child -- two dimension bits for the point tree (the format of
// tie_before / tie_rank), four for the feature trees (tie_before_nd)
template <int DIM> struct InfoShift { static constexpr uint32_t value = DIM == 3 ? 3u : 5u; };

// an internal node of the current level (more than LEAF_MAX points)
template <int DIM> struct Act {
  uint32_t left, right;        // its slice of the record array
  float blo[DIM], bhi[DIM];    // the box handed down to it (loose: the parent's box cut at the parent's cutval)
  MM<DIM> mm;                  // min / max of its points per dimension
  uint32_t node, depth;        // its TieNode id
  int feat; float cut;         // middleSplit_'s choice
  uint32_t lim1, lim2, idx;    // planeSplit's limits, the split index
};

template <int DIM> __device__ __forceinline__ float coord(const Rec<DIM>& p, int d) {
  float v = p.c[0];
#pragma unroll
  for (int k = 1; k < DIM; ++k) v = d == k ? p.c[k] : v;
  return v;
}

__global__ void k_init_recs(const float* __restrict__ xyz, uint32_t n, Rec<3>* __restrict__ recs, uint32_t* __restrict__ node_of) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    recs[i] = Rec<3>{{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]}, i};
    node_of[i] = 0u;
  }
}
// records given as {x, y, z, bits(original index)} in ANY order (a grid's sorted points), with the feature parts the adaptors append
// (w1 * att1[j], w2 * att2[j]: formed in f32 as the adaptors store them, :90 / :192 / :296): back to the original order
template <int DIM>
__global__ void k_init_recs_from_sorted(const float4* __restrict__ sorted, const float4* __restrict__ att1, float w1, const float4* __restrict__ att2, float w2,
                                        uint32_t n, Rec<DIM>* __restrict__ recs, uint32_t* __restrict__ node_of) {
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const float4 p = sorted[j];
    const uint32_t i = __float_as_uint(p.w);
    Rec<DIM> r;
    r.c[0] = p.x; r.c[1] = p.y; r.c[2] = p.z; r.idx = i;
    if (DIM >= 6) { const float4 a = att1[j]; r.c[3] = __fmul_rn(w1, a.x); r.c[4] = __fmul_rn(w1, a.y); r.c[DIM >= 6 ? 5 : 0] = __fmul_rn(w1, a.z); }
    if (DIM >= 9) { const float4 a = att2[j]; r.c[DIM >= 9 ? 6 : 0] = __fmul_rn(w2, a.x); r.c[DIM >= 9 ? 7 : 0] = __fmul_rn(w2, a.y); r.c[DIM >= 9 ? 8 : 0] = __fmul_rn(w2, a.z); }
    if (i < n) recs[i] = r;
    node_of[j] = 0u;
  }
}

// the root: computeBoundingBox (:1846-1877) = min / max over all points (the one run of the first reduce_by_key)
template <int DIM>
__global__ void k_root(const MM<DIM>* __restrict__ agg, uint32_t n, Act<DIM>* __restrict__ act, uint32_t* __restrict__ counts, uint4* __restrict__ nodes) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Act<DIM> a{};
  a.left = 0; a.right = n; a.mm = agg[0]; a.node = 0; a.depth = 0;
  for (int d = 0; d < DIM; ++d) { a.blo[d] = a.mm.mn[d]; a.bhi[d] = a.mm.mx[d]; }
  const bool leaf = n <= LEAF_MAX;
  counts[0] = leaf ? 0u : 1u;      // active nodes of level 0
  counts[1] = 1u;                  // nodes so far
  if (!leaf) act[0] = a;
  nodes[0] = make_uint4(0xFFFFFFFFu, 0u, leaf ? 0u : 0u, 0u);      // parent -1, depth 0; a leaf root: z = slot of its first point = 0
}

// middleSplit_: the cut dimension and value of every active node
template <int DIM>
__global__ void k_decide(Act<DIM>* __restrict__ act, const uint32_t* __restrict__ counts) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= counts[0]) return;
  Act<DIM>& A = act[a];
  const float EPS = 0.00001f;
  float max_span = __fsub_rn(A.bhi[0], A.blo[0]);
  for (int d = 1; d < DIM; ++d) { const float span = __fsub_rn(A.bhi[d], A.blo[d]); if (span > max_span) max_span = span; }
  const float thr = __fmul_rn(__fsub_rn(1.0f, EPS), max_span);
  float max_spread = -1.0f, min_elem = 0.0f, max_elem = 0.0f;
  int feat = 0;
  for (int d = 0; d < DIM; ++d) {
    const float span = __fsub_rn(A.bhi[d], A.blo[d]);
    if (span >= thr) {
      const float spread = __fsub_rn(A.mm.mx[d], A.mm.mn[d]);
      if (spread > max_spread) { feat = d; max_spread = spread; min_elem = A.mm.mn[d]; max_elem = A.mm.mx[d]; }
    }
  }
  float lo_f = A.blo[0], hi_f = A.bhi[0];
#pragma unroll
  for (int d = 1; d < DIM; ++d) { lo_f = feat == d ? A.blo[d] : lo_f; hi_f = feat == d ? A.bhi[d] : hi_f; }
  const float split_val = __fadd_rn(lo_f, hi_f) / 2;
  A.feat = feat;
  A.cut = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
}

// flags of a planeSplit pass: PASS 1 -- "not (value < cut)" over the node's slice; PASS 2 -- "value > cut" over [lim1, count)
template <int DIM, int PASS>
__global__ void k_flags(const Rec<DIM>* __restrict__ recs, const uint32_t* __restrict__ node_of, const Act<DIM>* __restrict__ act, uint32_t n, uint32_t* __restrict__ flags) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= n; p += gridDim.x * blockDim.x) {
    uint32_t f = 0;
    if (p < n) {
      const uint32_t a = node_of[p];
      if (a != NONE) {
        const Act<DIM>& A = act[a];
        const float v = coord<DIM>(recs[p], A.feat);
        f = PASS == 1 ? (v < A.cut ? 0u : 1u) : ((p - A.left >= A.lim1 && v > A.cut) ? 1u : 0u);
      }
    }
    flags[p] = f;      // (n + 1 entries: the scan's last value is the total)
  }
}

// ranks: S = exclusive scan of the flags.  Inside the pass's range [lo, hi) of a node: nF = flagged elements, nU = the others = the
// limit; a flagged element at relative position < nU is misplaced (rank = flagged elements before it), an unflagged one at >= nU is
// (rank from the right = unflagged elements behind it); rank -> position tables for the scatter.
template <int DIM, int PASS>
__global__ void k_ranks(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ S, const uint32_t* __restrict__ node_of, Act<DIM>* __restrict__ act, uint32_t n,
                        uint32_t* __restrict__ posF, uint32_t* __restrict__ posU) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t a = node_of[p];
    if (a == NONE) continue;
    Act<DIM>& A = act[a];
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

template <int DIM, int PASS>
__global__ void k_scatter(const Rec<DIM>* __restrict__ in, Rec<DIM>* __restrict__ out, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ S,
                          const uint32_t* __restrict__ node_of, const Act<DIM>* __restrict__ act, uint32_t n, const uint32_t* __restrict__ posF, const uint32_t* __restrict__ posU) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    uint32_t dest = p;
    const uint32_t a = node_of[p];
    if (a != NONE) {
      const Act<DIM>& A = act[a];
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
template <int DIM>
__global__ void k_children(Act<DIM>* __restrict__ act, const uint32_t* __restrict__ counts, uint32_t* __restrict__ internal) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= counts[0]) { if (a == counts[0]) { internal[2 * a] = 0u; } return; }      // (one entry behind the last: the scan's total)
  Act<DIM>& A = act[a];
  const uint32_t count = A.right - A.left;
  const uint32_t idx = A.lim1 > count / 2 ? A.lim1 : (A.lim2 < count / 2 ? A.lim2 : count / 2);
  A.idx = idx;
  internal[2 * a] = idx > LEAF_MAX ? 1u : 0u;
  internal[2 * a + 1] = (count - idx) > LEAF_MAX ? 1u : 0u;
}

// per record: the key of the child it now belongs to (2a + c), for the children's min / max
template <int DIM>
__global__ void k_child_keys(const uint32_t* __restrict__ node_of, const Act<DIM>* __restrict__ act, uint32_t n, uint32_t* __restrict__ keys) {
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t a = node_of[p];
    keys[p] = a == NONE ? NONE : 2u * a + (p >= act[a].left + act[a].idx ? 1u : 0u);
  }
}
template <int DIM>
__global__ void k_store_mm(const uint32_t* __restrict__ uk, const MM<DIM>* __restrict__ agg, const uint32_t* __restrict__ nruns, uint32_t cap, MM<DIM>* __restrict__ child_mm) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *nruns || r >= cap) return;
  if (uk[r] != NONE) child_mm[uk[r]] = agg[r];
}

// the node's record (split + the children's tight bounds along it, :1196-1205), the children's records, the next level's active nodes
template <int DIM>
__global__ void k_finish(const Act<DIM>* __restrict__ act, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ internal_scan, const MM<DIM>* __restrict__ child_mm,
                         uint4* __restrict__ nodes, uint32_t node_cap, Act<DIM>* __restrict__ act_next, uint32_t* __restrict__ counts_next) {
  constexpr uint32_t SH = InfoShift<DIM>::value;
  const uint32_t na = counts[0];
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a == 0) { counts_next[0] = internal_scan[2 * na]; counts_next[1] = counts[1] + 2u * na; }
  if (a >= na) return;
  const Act<DIM>& A = act[a];
  const uint32_t base = counts[1];      // ids of this level's children start here
  const MM<DIM> m1 = child_mm[2 * a], m2 = child_mm[2 * a + 1];
  float dlow = m1.mx[0], dhigh = m2.mn[0];
#pragma unroll
  for (int d = 1; d < DIM; ++d) { dlow = A.feat == d ? m1.mx[d] : dlow; dhigh = A.feat == d ? m2.mn[d] : dhigh; }
  if (A.node < node_cap) {
    uint4 nd = nodes[A.node];
    nd.y = (A.depth << SH) | ((uint32_t)A.feat << 1) | (nd.y & 1u);
    nd.z = __float_as_uint(dlow);       // divlow
    nd.w = __float_as_uint(dhigh);      // divhigh
    nodes[A.node] = nd;
  }
  for (uint32_t c = 0; c < 2; ++c) {
    const uint32_t id = base + 2u * a + c;
    const uint32_t left = c == 0 ? A.left : A.left + A.idx, right = c == 0 ? A.left + A.idx : A.right;
    const bool internal = (right - left) > LEAF_MAX;
    if (id < node_cap) nodes[id] = make_uint4(A.node, ((A.depth + 1u) << SH) | c, internal ? 0u : left, 0u);      // (a leaf: z = the slot of its first point)
    if (internal) {
      Act<DIM> B{};
      B.left = left; B.right = right; B.node = id; B.depth = A.depth + 1u;
      for (int d = 0; d < DIM; ++d) { B.blo[d] = A.blo[d]; B.bhi[d] = A.bhi[d]; }
#pragma unroll
      for (int d = 0; d < DIM; ++d) if (A.feat == d) { if (c == 0) B.bhi[d] = A.cut; else B.blo[d] = A.cut; }
      B.mm = c == 0 ? m1 : m2;
      B.lim1 = B.lim2 = B.idx = 0;
      act_next[internal_scan[2 * a + c]] = B;
    }
  }
}

// every record of this level's nodes moves to its child: the next level's active index, or -- a leaf -- its final leaf and slot
template <int DIM>
__global__ void k_descend(const Rec<DIM>* __restrict__ recs, uint32_t* __restrict__ node_of, const Act<DIM>* __restrict__ act, const uint32_t* __restrict__ counts,
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
      const uint32_t i = recs[p].idx;
      leaf_by_index[i] = base + 2u * a + c;
      slot_by_index[i] = p;
    }
  }
}
// a cloud of at most LEAF_MAX points: the root is the only leaf
template <int DIM>
__global__ void k_root_leaf(const Rec<DIM>* __restrict__ recs, uint32_t n, uint32_t* __restrict__ leaf_by_index, uint32_t* __restrict__ slot_by_index) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t i = recs[p].idx;
  leaf_by_index[i] = 0u; slot_by_index[i] = p;
}

// ---- small clouds: the whole build in ONE workgroup -------------------------------------------------------------------------------------
// A level of the general build is some twenty launches and a read-back of two counters: a cloud of a thousand points -- the centroids of
// a KMeans iteration, whose kd branch wants these tables in every iteration that meets exact ties (kmeans.hip) -- spends a millisecond on
// launch chains.  Up to SMALL_N points one workgroup runs the SAME passes (the same flags, ranks, scatters, the same node records) as
// phases between barriers, with block-wide prefix sums and min / max through LDS atomics on order-preserving integer images of the floats.
constexpr uint32_t SMALL_N = 2048;
constexpr uint32_t SMALL_T = 1024;
constexpr uint32_t SMALL_ACT = SMALL_N / (LEAF_MAX + 1) + 2;

struct SmallWs {
  Rec<3>*recA, *recB;
  uint32_t *node_of, *flags, *S, *posF, *posU, *internal, *iscan;
  MM<3>* child_mm;
  Act<3>*act0, *act1;
  uint32_t* out;      // [2]: nodes, depth
};

__device__ __forceinline__ uint32_t ord_of(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord_inv(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); }

// out[i] = in[0] + ... + in[i - 1], i < count (count <= a few thousand), by the whole block; tmp: SMALL_T words of LDS
__device__ void block_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t count, uint32_t* tmp) {
  const uint32_t T = blockDim.x, t = threadIdx.x;
  const uint32_t per = (count + T - 1u) / T;
  const uint32_t b = min(t * per, count), e = min(b + per, count);
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += in[i];
  tmp[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < T; off <<= 1) {
    const uint32_t v = t >= off ? tmp[t - off] : 0u;
    __syncthreads();
    tmp[t] += v;
    __syncthreads();
  }
  uint32_t run = tmp[t] - sum;
  for (uint32_t i = b; i < e; ++i) { const uint32_t f = in[i]; out[i] = run; run += f; }
  __syncthreads();
}

__global__ __launch_bounds__(SMALL_T) void k_build_small(const float* xyz, const float4* sorted, uint32_t n, SmallWs w, uint4* nodes,
                                                         uint32_t node_cap, uint32_t* leaf_by_index, uint32_t* slot_by_index) {      // (no __restrict__: the threads of the block hand data to each other through these arrays)
  __shared__ uint32_t tmp[SMALL_T];
  __shared__ uint32_t mmk[2 * SMALL_ACT + 2][6];      // min x, y, z, max x, y, z of a level's children (ordered integer images)
  __shared__ uint32_t s_na, s_nodes;
  const uint32_t T = blockDim.x, t = threadIdx.x;
  constexpr uint32_t SH = InfoShift<3>::value;
  Rec<3>*recA = w.recA, *recB = w.recB;
  // records in the ORIGINAL order (the reference's vAcc_ starts as 0 .. n-1)
  for (uint32_t p = t; p < n; p += T) {
    if (sorted) { const float4 q = sorted[p]; const uint32_t i = __float_as_uint(q.w); if (i < n) recA[i] = Rec<3>{{q.x, q.y, q.z}, i}; }
    else recA[p] = Rec<3>{{xyz[3 * (size_t)p], xyz[3 * (size_t)p + 1], xyz[3 * (size_t)p + 2]}, p};
    w.node_of[p] = 0u;
  }
  if (t < 6) mmk[0][t] = t < 3 ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  for (uint32_t p = t; p < n; p += T) {
    const Rec<3> r = recA[p];
#pragma unroll
    for (int d = 0; d < 3; ++d) { atomicMin(&mmk[0][d], ord_of(r.c[d])); atomicMax(&mmk[0][3 + d], ord_of(r.c[d])); }
  }
  __syncthreads();
  if (t == 0) {      // the root (k_root)
    Act<3> a{};
    a.left = 0; a.right = n; a.node = 0; a.depth = 0;
    for (int d = 0; d < 3; ++d) { a.mm.mn[d] = ord_inv(mmk[0][d]); a.mm.mx[d] = ord_inv(mmk[0][3 + d]); a.blo[d] = a.mm.mn[d]; a.bhi[d] = a.mm.mx[d]; }
    const bool leaf = n <= LEAF_MAX;
    s_na = leaf ? 0u : 1u; s_nodes = 1u;
    if (!leaf) w.act0[0] = a;
    nodes[0] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  }
  __syncthreads();
  if (s_na == 0u) {      // at most LEAF_MAX points: the root is the only leaf (k_root_leaf)
    for (uint32_t p = t; p < n; p += T) { const uint32_t i = recA[p].idx; leaf_by_index[i] = 0u; slot_by_index[i] = p; }
    if (t == 0) { w.out[0] = 1u; w.out[1] = 0u; }
    return;
  }
  Act<3>*cur = w.act0, *nxt = w.act1;
  uint32_t depth = 0;
  for (;;) {
    const uint32_t na = s_na, base = s_nodes;      // (block-uniform: read after a barrier)
    // middleSplit_ (k_decide)
    if (t < na) {
      Act<3>& A = cur[t];
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
      const float lo_f = feat == 0 ? A.blo[0] : feat == 1 ? A.blo[1] : A.blo[2], hi_f = feat == 0 ? A.bhi[0] : feat == 1 ? A.bhi[1] : A.bhi[2];
      const float split_val = __fadd_rn(lo_f, hi_f) / 2;
      A.feat = feat;
      A.cut = split_val < min_elem ? min_elem : (split_val > max_elem ? max_elem : split_val);
    }
    __syncthreads();
    // planeSplit: two passes of {flags, prefix sums, ranks, scatter} (k_flags / k_ranks / k_scatter)
    for (int pass = 1; pass <= 2; ++pass) {
      const Rec<3>* in = pass == 1 ? recA : recB;
      Rec<3>* outp = pass == 1 ? recB : recA;
      for (uint32_t p = t; p <= n; p += T) {
        uint32_t f = 0;
        if (p < n) {
          const uint32_t a = w.node_of[p];
          if (a != NONE) {
            const Act<3>& A = cur[a];
            const float v = coord<3>(in[p], A.feat);
            f = pass == 1 ? (v < A.cut ? 0u : 1u) : ((p - A.left >= A.lim1 && v > A.cut) ? 1u : 0u);
          }
        }
        w.flags[p] = f;
      }
      __syncthreads();
      block_exclusive_scan(w.flags, w.S, n + 1u, tmp);
      for (uint32_t p = t; p < n; p += T) {
        const uint32_t a = w.node_of[p];
        if (a == NONE) continue;
        Act<3>& A = cur[a];
        const uint32_t lo = pass == 1 ? A.left : A.left + A.lim1, hi = A.right;
        if (p < lo) continue;
        const uint32_t nF = w.S[hi] - w.S[lo], nU = (hi - lo) - nF;
        const uint32_t rel = p - lo, Fb = w.S[p] - w.S[lo];
        if (w.flags[p]) { if (rel < nU) w.posF[lo + Fb] = p; }
        else if (rel >= nU) { const uint32_t Ub = rel - Fb; w.posU[lo + (nU - Ub - 1u)] = p; }
      }
      __syncthreads();
      for (uint32_t p = t; p < n; p += T) {
        uint32_t dest = p;
        const uint32_t a = w.node_of[p];
        if (a != NONE) {
          const Act<3>& A = cur[a];
          // (pass 1 scatters with the range [left, right); pass 2 with [left + lim1, right): lim1 is not touched by pass 2's limit write)
          const uint32_t lo = pass == 1 ? A.left : A.left + A.lim1, hi = A.right;
          if (p >= lo) {
            const uint32_t nF = w.S[hi] - w.S[lo], nU = (hi - lo) - nF;
            const uint32_t rel = p - lo, Fb = w.S[p] - w.S[lo];
            if (w.flags[p]) { if (rel < nU) dest = w.posU[lo + Fb]; }
            else if (rel >= nU) { const uint32_t Ub = rel - Fb; dest = w.posF[lo + (nU - Ub - 1u)]; }
          }
        }
        outp[dest] = in[p];
      }
      __syncthreads();
      // (the limits are written in a phase of their own: no thread of the block reads a node's record while one writes it)
      for (uint32_t p = t; p < n; p += T) {
        const uint32_t a = w.node_of[p];
        if (a == NONE) continue;
        Act<3>& A = cur[a];
        const uint32_t lo = pass == 1 ? A.left : A.left + A.lim1;
        if (p != lo) continue;
        const uint32_t nF = w.S[A.right] - w.S[lo], nU = (A.right - lo) - nF;
        if (pass == 1) A.lim1 = nU; else A.lim2 = A.lim1 + nU;
      }
      __syncthreads();
    }
    // the split index, which children go on (k_children), their active indices
    if (t < na) {
      Act<3>& A = cur[t];
      const uint32_t count = A.right - A.left;
      const uint32_t idx = A.lim1 > count / 2 ? A.lim1 : (A.lim2 < count / 2 ? A.lim2 : count / 2);
      A.idx = idx;
      w.internal[2 * t] = idx > LEAF_MAX ? 1u : 0u;
      w.internal[2 * t + 1] = (count - idx) > LEAF_MAX ? 1u : 0u;
    }
    if (t == na) w.internal[2 * na] = 0u;
    for (uint32_t c = t; c < 2u * na; c += T) { mmk[c][0] = mmk[c][1] = mmk[c][2] = 0xFFFFFFFFu; mmk[c][3] = mmk[c][4] = mmk[c][5] = 0u; }
    __syncthreads();
    block_exclusive_scan(w.internal, w.iscan, 2u * na + 1u, tmp);
    // the children's min / max (the general build: reduce_by_key over the child keys)
    for (uint32_t p = t; p < n; p += T) {
      const uint32_t a = w.node_of[p];
      if (a == NONE) continue;
      const uint32_t key = 2u * a + (p >= cur[a].left + cur[a].idx ? 1u : 0u);
      const Rec<3> r = recA[p];
#pragma unroll
      for (int d = 0; d < 3; ++d) { atomicMin(&mmk[key][d], ord_of(r.c[d])); atomicMax(&mmk[key][3 + d], ord_of(r.c[d])); }
    }
    __syncthreads();
    for (uint32_t c = t; c < 2u * na; c += T) {
      MM<3> m;
      for (int d = 0; d < 3; ++d) { m.mn[d] = ord_inv(mmk[c][d]); m.mx[d] = ord_inv(mmk[c][3 + d]); }
      w.child_mm[c] = m;
    }
    __syncthreads();
    // node records, the next level's active nodes (k_finish)
    if (t < na) {
      const Act<3>& A = cur[t];
      const MM<3> m1 = w.child_mm[2 * t], m2 = w.child_mm[2 * t + 1];
      const float dlow = A.feat == 0 ? m1.mx[0] : A.feat == 1 ? m1.mx[1] : m1.mx[2], dhigh = A.feat == 0 ? m2.mn[0] : A.feat == 1 ? m2.mn[1] : m2.mn[2];
      if (A.node < node_cap) {
        uint4 nd = nodes[A.node];
        nd.y = (A.depth << SH) | ((uint32_t)A.feat << 1) | (nd.y & 1u);
        nd.z = __float_as_uint(dlow);
        nd.w = __float_as_uint(dhigh);
        nodes[A.node] = nd;
      }
      for (uint32_t c = 0; c < 2; ++c) {
        const uint32_t id = base + 2u * t + c;
        const uint32_t left = c == 0 ? A.left : A.left + A.idx, right = c == 0 ? A.left + A.idx : A.right;
        const bool internal = (right - left) > LEAF_MAX;
        if (id < node_cap) nodes[id] = make_uint4(A.node, ((A.depth + 1u) << SH) | c, internal ? 0u : left, 0u);
        if (internal) {
          Act<3> B{};
          B.left = left; B.right = right; B.node = id; B.depth = A.depth + 1u;
          for (int d = 0; d < 3; ++d) { B.blo[d] = A.blo[d]; B.bhi[d] = A.bhi[d]; }
          for (int d = 0; d < 3; ++d) if (A.feat == d) { if (c == 0) B.bhi[d] = A.cut; else B.blo[d] = A.cut; }
          B.mm = c == 0 ? m1 : m2;
          B.lim1 = B.lim2 = B.idx = 0;
          nxt[w.iscan[2 * t + c]] = B;
        }
      }
    }
    // every record to its child: the next level's active index, or its final leaf and slot (k_descend)
    for (uint32_t p = t; p < n; p += T) {
      const uint32_t a = w.node_of[p];
      if (a == NONE) continue;
      const uint32_t c = p >= cur[a].left + cur[a].idx ? 1u : 0u;
      if (w.internal[2 * a + c]) w.node_of[p] = w.iscan[2 * a + c];
      else {
        w.node_of[p] = NONE;
        const uint32_t i = recA[p].idx;
        leaf_by_index[i] = base + 2u * a + c;
        slot_by_index[i] = p;
      }
    }
    __syncthreads();
    if (t == 0) { s_na = w.iscan[2 * na]; s_nodes = base + 2u * na; }
    __syncthreads();
    { Act<3>* x = cur; cur = nxt; nxt = x; }
    ++depth;
    if (s_na == 0u || depth > 200u) break;
  }
  __syncthreads();
  if (t == 0) { w.out[0] = s_nodes; w.out[1] = depth; }
}

// the build of a cloud of at most SMALL_N points: one allocation, one launch, one read-back
hipError_t build_small(const float* d_xyz, const float4* d_sorted, uint32_t n, hipStream_t s, uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index, uint4** d_nodes_out,
                       size_t* n_nodes_out, int* max_depth_out) {
  hipError_t e = hipSuccess;
  void* ws = nullptr;
  uint4* nodes = nullptr;
  const size_t node_cap = 2 * (size_t)n + 2;      // (every split leaves both children non-empty: at most 2n - 1 nodes)
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_recA = take((size_t)n * sizeof(Rec<3>)), o_recB = take((size_t)n * sizeof(Rec<3>)), o_node = take((size_t)n * 4), o_flags = take(((size_t)n + 1) * 4),
               o_S = take(((size_t)n + 1) * 4), o_posF = take((size_t)n * 4), o_posU = take((size_t)n * 4), o_int = take((2 * (size_t)SMALL_ACT + 2) * 4),
               o_iscan = take((2 * (size_t)SMALL_ACT + 2) * 4), o_cmm = take((2 * (size_t)SMALL_ACT + 2) * sizeof(MM<3>)), o_act0 = take((size_t)SMALL_ACT * sizeof(Act<3>)),
               o_act1 = take((size_t)SMALL_ACT * sizeof(Act<3>)), o_out = take(16);
  uint32_t h_out[2] = {0, 0};
  do {
    if ((e = hipMalloc(&ws, off)) != hipSuccess) break;
    if ((e = hipMalloc(&nodes, node_cap * sizeof(uint4))) != hipSuccess) break;
    unsigned char* b = static_cast<unsigned char*>(ws);
    SmallWs w{(Rec<3>*)(b + o_recA), (Rec<3>*)(b + o_recB), (uint32_t*)(b + o_node), (uint32_t*)(b + o_flags), (uint32_t*)(b + o_S), (uint32_t*)(b + o_posF), (uint32_t*)(b + o_posU),
              (uint32_t*)(b + o_int), (uint32_t*)(b + o_iscan), (MM<3>*)(b + o_cmm), (Act<3>*)(b + o_act0), (Act<3>*)(b + o_act1), (uint32_t*)(b + o_out)};
    hipLaunchKernelGGL(k_build_small, dim3(1), dim3(SMALL_T), 0, s, d_xyz, d_sorted, n, w, nodes, (uint32_t)node_cap, d_leaf_by_index, d_slot_by_index);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipMemcpyAsync(h_out, w.out, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) break;
    if (h_out[1] > 200u) { e = hipErrorUnknown; break; }
    *d_nodes_out = nodes; nodes = nullptr;
    *n_nodes_out = h_out[0];
    if (max_depth_out) *max_depth_out = (int)h_out[1];
  } while (0);
  if (ws) (void)hipFree(ws);
  if (nodes) (void)hipFree(nodes);
  return e;
}

#define TB_TRY(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)

// The build.  Records come from d_xyz (DIM = 3: the cloud in its ORIGINAL order) or from d_sorted (+ the feature parts att1 / att2 by the
// same positions, weighted): back to the original order first -- the reference's vAcc_ starts as 0 .. n-1.
template <int DIM>
hipError_t build_impl(const float* d_xyz, const float4* d_sorted, const float4* att1, float w1, const float4* att2, float w2, uint32_t n, hipStream_t s,
                      uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index, uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out) {
  typedef Rec<DIM> R;
  typedef MM<DIM> M;
  typedef Act<DIM> A;
  *d_nodes_out = nullptr; *n_nodes_out = 0; if (max_depth_out) *max_depth_out = 0;
  if (n == 0) return hipSuccess;
  hipError_t e = hipSuccess;
  const unsigned gp = (unsigned)std::min<size_t>(((size_t)n + TB) / TB, 65535u * 4u);
  // active nodes of a level hold more than LEAF_MAX points each; a level's children: twice that; all nodes: bounded by 2n (leaves of one
  // point), in practice ~0.3 n -- the node array grows by doubling when a level does not fit
  const uint32_t act_cap = n / (LEAF_MAX + 1) + 2;
  size_t node_cap = std::max<size_t>(1024, (size_t)n / 2 + 64);
  R *recA = nullptr, *recB = nullptr;
  uint32_t *node_of = nullptr, *flags = nullptr, *S = nullptr, *posF = nullptr, *posU = nullptr, *keys = nullptr, *uk = nullptr, *internal = nullptr, *iscan = nullptr;
  uint32_t *counts = nullptr, *nruns = nullptr;
  M *agg = nullptr, *child_mm = nullptr;
  A *act0 = nullptr, *act1 = nullptr;
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
      auto vals = rocprim::make_transform_iterator((const R*)nullptr, ToMM<DIM>());
      TB_TRY(rocprim::reduce_by_key(nullptr, b3, (uint32_t*)nullptr, vals, (size_t)n, (uint32_t*)nullptr, (M*)nullptr, (uint32_t*)nullptr, MMOp<DIM>(), rocprim::equal_to<uint32_t>(), s));
      tmp_bytes = std::max(b1, std::max(b2, b3));
    }
    // ONE allocation for the whole workspace (twenty hipMalloc / hipFree pairs were a third of a small cloud's build)
    {
      size_t off = 0;
      auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
      const size_t o_recA = take((size_t)n * sizeof(R)), o_recB = take((size_t)n * sizeof(R)), o_node = take((size_t)n * 4), o_flags = take(((size_t)n + 1) * 4),
                   o_S = take(((size_t)n + 1) * 4), o_posF = take((size_t)n * 4), o_posU = take((size_t)n * 4), o_keys = take((size_t)n * 4), o_uk = take(run_cap * 4),
                   o_agg = take(run_cap * sizeof(M)), o_cmm = take((2 * (size_t)act_cap + 2) * sizeof(M)), o_int = take((2 * (size_t)act_cap + 2) * 4),
                   o_iscan = take((2 * (size_t)act_cap + 2) * 4), o_counts = take(16), o_nruns = take(4), o_act0 = take((size_t)act_cap * sizeof(A)),
                   o_act1 = take((size_t)act_cap * sizeof(A)), o_tmp = take(tmp_bytes ? tmp_bytes : 16);
      TB_TRY(hipMalloc(&ws, off));
      unsigned char* b = static_cast<unsigned char*>(ws);
      recA = (R*)(b + o_recA); recB = (R*)(b + o_recB); node_of = (uint32_t*)(b + o_node); flags = (uint32_t*)(b + o_flags); S = (uint32_t*)(b + o_S);
      posF = (uint32_t*)(b + o_posF); posU = (uint32_t*)(b + o_posU); keys = (uint32_t*)(b + o_keys); uk = (uint32_t*)(b + o_uk); agg = (M*)(b + o_agg);
      child_mm = (M*)(b + o_cmm); internal = (uint32_t*)(b + o_int); iscan = (uint32_t*)(b + o_iscan); counts = (uint32_t*)(b + o_counts); nruns = (uint32_t*)(b + o_nruns);
      act0 = (A*)(b + o_act0); act1 = (A*)(b + o_act1); tmp = b + o_tmp;
    }
    TB_TRY(hipMalloc(&nodes, node_cap * sizeof(uint4)));
    if (d_sorted) hipLaunchKernelGGL((k_init_recs_from_sorted<DIM>), dim3(gp), dim3(TB), 0, s, d_sorted, att1, w1, att2, w2, n, recA, node_of);
    else if (DIM == 3) hipLaunchKernelGGL(k_init_recs, dim3(gp), dim3(TB), 0, s, d_xyz, n, reinterpret_cast<Rec<3>*>(recA), node_of);
    else { e = hipErrorInvalidValue; goto done; }
    // the root's box: one run of key 0
    TB_TRY(hipMemsetAsync(keys, 0, (size_t)n * 4, s));
    {
      auto vals = rocprim::make_transform_iterator((const R*)recA, ToMM<DIM>());
      size_t b = tmp_bytes;
      TB_TRY(rocprim::reduce_by_key(tmp, b, keys, vals, (size_t)n, uk, agg, nruns, MMOp<DIM>(), rocprim::equal_to<uint32_t>(), s));
    }
    hipLaunchKernelGGL((k_root<DIM>), dim3(1), dim3(64), 0, s, (const M*)agg, n, act0, counts, nodes);
    TB_TRY(hipMemcpyAsync(h_counts, counts, 8, hipMemcpyDeviceToHost, s));
    TB_TRY(hipStreamSynchronize(s));
    if (h_counts[0] == 0) hipLaunchKernelGGL((k_root_leaf<DIM>), dim3((n + TB - 1) / TB), dim3(TB), 0, s, (const R*)recA, n, d_leaf_by_index, d_slot_by_index);
    A *cur = act0, *nxt = act1;
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
      hipLaunchKernelGGL((k_decide<DIM>), dim3(ga), dim3(TB), 0, s, cur, (const uint32_t*)cnt_cur);
      // planeSplit, first pass: recA -> recB
      hipLaunchKernelGGL((k_flags<DIM, 1>), dim3(gp), dim3(TB), 0, s, (const R*)recA, (const uint32_t*)node_of, (const A*)cur, n, flags);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, flags, S, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL((k_ranks<DIM, 1>), dim3(gp), dim3(TB), 0, s, (const uint32_t*)flags, (const uint32_t*)S, (const uint32_t*)node_of, cur, n, posF, posU);
      hipLaunchKernelGGL((k_scatter<DIM, 1>), dim3(gp), dim3(TB), 0, s, (const R*)recA, recB, (const uint32_t*)flags, (const uint32_t*)S, (const uint32_t*)node_of, (const A*)cur, n,
                         (const uint32_t*)posF, (const uint32_t*)posU);
      // second pass: recB -> recA
      hipLaunchKernelGGL((k_flags<DIM, 2>), dim3(gp), dim3(TB), 0, s, (const R*)recB, (const uint32_t*)node_of, (const A*)cur, n, flags);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, flags, S, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL((k_ranks<DIM, 2>), dim3(gp), dim3(TB), 0, s, (const uint32_t*)flags, (const uint32_t*)S, (const uint32_t*)node_of, cur, n, posF, posU);
      hipLaunchKernelGGL((k_scatter<DIM, 2>), dim3(gp), dim3(TB), 0, s, (const R*)recB, recA, (const uint32_t*)flags, (const uint32_t*)S, (const uint32_t*)node_of, (const A*)cur, n,
                         (const uint32_t*)posF, (const uint32_t*)posU);
      // children
      hipLaunchKernelGGL((k_children<DIM>), dim3(ga), dim3(TB), 0, s, cur, (const uint32_t*)cnt_cur, internal);
      { size_t b = tmp_bytes; TB_TRY(rocprim::exclusive_scan(tmp, b, internal, iscan, 0u, 2 * (size_t)na + 1, rocprim::plus<uint32_t>(), s)); }
      hipLaunchKernelGGL((k_child_keys<DIM>), dim3(gp), dim3(TB), 0, s, (const uint32_t*)node_of, (const A*)cur, n, keys);
      {
        auto vals = rocprim::make_transform_iterator((const R*)recA, ToMM<DIM>());
        size_t b = tmp_bytes;
        TB_TRY(rocprim::reduce_by_key(tmp, b, keys, vals, (size_t)n, uk, agg, nruns, MMOp<DIM>(), rocprim::equal_to<uint32_t>(), s));
      }
      {
        const uint32_t rc = (uint32_t)std::min<size_t>(4 * (size_t)na + 4, 4 * (size_t)act_cap + 4);
        hipLaunchKernelGGL((k_store_mm<DIM>), dim3((rc + TB - 1) / TB), dim3(TB), 0, s, (const uint32_t*)uk, (const M*)agg, (const uint32_t*)nruns, rc, child_mm);
      }
      hipLaunchKernelGGL((k_finish<DIM>), dim3(ga), dim3(TB), 0, s, (const A*)cur, (const uint32_t*)cnt_cur, (const uint32_t*)iscan, (const M*)child_mm, nodes,
                         (uint32_t)std::min<size_t>(node_cap, 0xFFFFFFFFu), nxt, cnt_nxt);
      hipLaunchKernelGGL((k_descend<DIM>), dim3(gp), dim3(TB), 0, s, (const R*)recA, node_of, (const A*)cur, (const uint32_t*)cnt_cur, (const uint32_t*)iscan, (const uint32_t*)internal, n,
                         d_leaf_by_index, d_slot_by_index);
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

}  // namespace

// d_xyz: the cloud in its ORIGINAL order (3 floats per point), or d_sorted: {x, y, z, bits(original index)} records in any order (one
// of the two).  d_leaf_by_index / d_slot_by_index: [n], by ORIGINAL index.  *d_nodes_out: the TieNode records (hipMalloc'ed here, the
// caller frees), *n_nodes_out how many; *max_depth_out the deepest node's depth.
hipError_t tie_order_build_device(const float* d_xyz, const float4* d_sorted, uint32_t n, hipStream_t s, uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index,
                                  uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out) {
  if (n != 0 && n <= SMALL_N && (d_xyz != nullptr) != (d_sorted != nullptr)) {      // one workgroup: a launch instead of twenty per level
    *d_nodes_out = nullptr; *n_nodes_out = 0; if (max_depth_out) *max_depth_out = 0;
    return build_small(d_xyz, d_sorted, n, s, d_leaf_by_index, d_slot_by_index, d_nodes_out, n_nodes_out, max_depth_out);
  }
  return build_impl<3>(d_xyz, d_sorted, nullptr, 0.0f, nullptr, 0.0f, n, s, d_leaf_by_index, d_slot_by_index, d_nodes_out, n_nodes_out, max_depth_out);
}
// The tree of a feature adaptor's search: records (p, w1 * att1[, w2 * att2]) from the grid's sorted points and the attributes at the same
// positions; dim = 6 or 9.  Node records carry the split dimension in four bits: TieNode::info = (depth << 5) | (dimension << 1) | second child.
hipError_t tie_order_build_device_features(int dim, const float4* d_sorted, const float4* att1, float w1, const float4* att2, float w2, uint32_t n, hipStream_t s,
                                           uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index, uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out) {
  if (dim == 6) return build_impl<6>(nullptr, d_sorted, att1, w1, nullptr, 0.0f, n, s, d_leaf_by_index, d_slot_by_index, d_nodes_out, n_nodes_out, max_depth_out);
  if (dim == 9) return build_impl<9>(nullptr, d_sorted, att1, w1, att2, w2, n, s, d_leaf_by_index, d_slot_by_index, d_nodes_out, n_nodes_out, max_depth_out);
  return hipErrorInvalidValue;
}

}  // namespace cilhip
