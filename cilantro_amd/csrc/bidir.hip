// bidir.hip -- the engine's other search directions:
//   correspondence_search/correspondence_search_kd_tree.hpp:185-222   FIRST_TO_SECOND and BOTH (the reference
//       REBUILDS a kd-tree over the transformed source every call, :188-190 / :209-211)
//   correspondence_search/correspondence_search_kd_tree_utilities.hpp:65-101   BOTH = the two unidirectional sets
//       sorted by (indexInFirst, indexInSecond), then set_union, or set_intersection with require_reciprocality_
//   core/correspondence.hpp:57-100   post-filters (one-to-one: FIRST_TO_SECOND branch; BOTH: no-op)
//
// Here: q = T*s for the sorted source -> a second uniform grid over q (the same grid builder as for the target)
// -> reverse search (queries = the target points in their grid order, per-lane search kernel) -> every directed match
// becomes a 64-bit key (first << 32 | second, ORIGINAL indices) -> one radix sort -> union = first of each run of equal
// keys, intersection = keys that occur twice -> ordered compaction into a PAIR LIST.  The accumulation kernels then run
// unchanged over a gathered view of that list (one "query" per pair).
#include <hip/hip_runtime.h>

#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "internal.hpp"

namespace cilhip {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

namespace {

constexpr unsigned long long KEY_INVALID = 0xFFFFFFFFFFFFFFFFull;
inline int nblk(size_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096) + (n == 0); }

struct TfDev { float m[16]; };

// q (packed xyz, index = sorted-source position) = T * s with the pinned arithmetic of the search kernels
__global__ void k_transform_sorted(const float4* __restrict__ src_sorted, uint32_t ns, const IcpState* __restrict__ st, float* __restrict__ out) {
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const float4 s = src_sorted[i];
    float qx, qy, qz;
    transform_point(T, s.x, s.y, s.z, qx, qy, qz);
    out[3 * (size_t)i] = qx; out[3 * (size_t)i + 1] = qy; out[3 * (size_t)i + 2] = qz;
  }
}

// candidate slots [0, nd): reverse matches (target point jd -> nearest transformed source point)
__global__ void k_cand_reverse(const float4* __restrict__ dst_sorted, uint32_t nd, const uint32_t* __restrict__ rev_pos,
                               const float* __restrict__ rev_d2, const float4* __restrict__ q_sorted /*w = sorted-source position*/,
                               const float4* __restrict__ src_sorted, unsigned long long* keys, uint32_t* slots, uint32_t* posd, uint32_t* poss,
                               float* d2) {
  for (uint32_t jd = blockIdx.x * blockDim.x + threadIdx.x; jd < nd; jd += gridDim.x * blockDim.x) {
    const uint32_t rp = rev_pos[jd];
    unsigned long long key = KEY_INVALID;
    uint32_t sp = NONE_U32;
    if (rp != NONE_U32) {
      sp = __float_as_uint(q_sorted[rp].w);
      key = ((unsigned long long)__float_as_uint(dst_sorted[jd].w) << 32) | (unsigned long long)__float_as_uint(src_sorted[sp].w);
    }
    keys[jd] = key; slots[jd] = jd;
    posd[jd] = jd; poss[jd] = sp; d2[jd] = rev_d2[jd];
  }
}

// candidate slots [nd, nd + ns): forward matches (source point -> nearest target point)
__global__ void k_cand_forward(const float4* __restrict__ dst_sorted, uint32_t nd, const float4* __restrict__ src_sorted, uint32_t ns,
                               const uint32_t* __restrict__ nn_pos, const float* __restrict__ nn_d2, unsigned long long* keys, uint32_t* slots,
                               uint32_t* posd, uint32_t* poss, float* d2) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t pos = nn_pos[i];
    const uint32_t t = nd + i;
    keys[t] = pos != NONE_U32 ? (((unsigned long long)__float_as_uint(dst_sorted[pos].w) << 32) | (unsigned long long)__float_as_uint(src_sorted[i].w))
                              : KEY_INVALID;
    slots[t] = t;
    posd[t] = pos; poss[t] = i; d2[t] = nn_d2[i];
  }
}

// mode 0: keep every valid key (single direction); 1: union (first of each run); 2: intersection (keys that occur twice)
__global__ void k_mark(const unsigned long long* __restrict__ keys, uint32_t n, int mode, uint32_t* flags) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const unsigned long long k = keys[t];
    bool keep = k != KEY_INVALID;
    if (keep && mode == 1) keep = (t == 0) || keys[t - 1] != k;
    if (keep && mode == 2) keep = (t + 1 < n) && keys[t + 1] == k;
    flags[t] = keep ? 1u : 0u;
  }
}

__global__ void k_compact_sorted(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ flags,
                                 const uint32_t* __restrict__ offs, uint32_t n, const uint32_t* __restrict__ c_posd, const uint32_t* __restrict__ c_poss,
                                 const float* __restrict__ c_d2, uint32_t* first, uint32_t* second, uint32_t* posd, uint32_t* poss, float* d2) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    if (!flags[t]) continue;
    const uint32_t o = offs[t], sl = slots[t];
    const unsigned long long k = keys[t];
    first[o] = (uint32_t)(k >> 32); second[o] = (uint32_t)(k & 0xFFFFFFFFull);
    posd[o] = c_posd[sl]; poss[o] = c_poss[sl]; d2[o] = c_d2[sl];
  }
}

// in-place-safe compaction of a pair list by flags (offs = exclusive scan of flags); out arrays distinct from in arrays
__global__ void k_compact_pairs(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs, uint32_t n, const uint32_t* f_in,
                                const uint32_t* s_in, const uint32_t* pd_in, const uint32_t* ps_in, const float* d_in, uint32_t* f_out, uint32_t* s_out,
                                uint32_t* pd_out, uint32_t* ps_out, float* d_out) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    if (!flags[t]) continue;
    const uint32_t o = offs[t];
    f_out[o] = f_in[t]; s_out[o] = s_in[t]; pd_out[o] = pd_in[t]; ps_out[o] = ps_in[t]; d_out[o] = d_in[t];
  }
}

// one-to-one, FIRST_TO_SECOND branch (correspondence.hpp:72-82): per source point the match with the smallest value
// (ties: lowest target index -- the reference's unstable sort leaves that open)
__global__ void k_o2o_min_pairs(const uint32_t* __restrict__ first, const uint32_t* __restrict__ poss, const float* __restrict__ d2, uint32_t n,
                                unsigned long long* winner) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    atomicMin(&winner[poss[t]], ((unsigned long long)__float_as_uint(d2[t]) << 32) | (unsigned long long)first[t]);
}
__global__ void k_o2o_flags_pairs(const uint32_t* __restrict__ first, const uint32_t* __restrict__ poss, const float* __restrict__ d2, uint32_t n,
                                  const unsigned long long* __restrict__ winner, uint32_t* flags) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
    flags[t] = winner[poss[t]] == (((unsigned long long)__float_as_uint(d2[t]) << 32) | (unsigned long long)first[t]) ? 1u : 0u;
}

__global__ void k_gather_pair_view(const float4* __restrict__ src_sorted, const float4* __restrict__ src_nrm_sorted, const uint32_t* __restrict__ poss,
                                   uint32_t n, float4* src_view, float4* nrm_view) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const uint32_t sp = poss[t];
    src_view[t] = src_sorted[sp];
    if (nrm_view) nrm_view[t] = src_nrm_sorted[sp];
  }
}

hipError_t scan_flags(const uint32_t* flags, uint32_t* offs, uint32_t n, uint32_t* total_out, hipStream_t s) {
  *total_out = 0;
  if (n == 0) return hipSuccess;
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, flags, offs, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  void* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, flags, offs, 0u, (size_t)n, rocprim::plus<uint32_t>(), s);
  uint32_t last_off = 0, last_flag = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&last_off, offs + (n - 1), 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&last_flag, flags + (n - 1), 4, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  if (e != hipSuccess) return e;
  if (e2 != hipSuccess) return e2;
  *total_out = last_off + last_flag;
  return hipSuccess;
}

}  // namespace

void free_pairs(PairSet& p) {
  uint32_t** u[] = {&p.first, &p.second, &p.posd, &p.poss, &p.first2, &p.second2, &p.posd2, &p.poss2};
  for (auto q : u) { if (*q) (void)hipFree(*q); *q = nullptr; }
  if (p.d2) (void)hipFree(p.d2);
  if (p.d2b) (void)hipFree(p.d2b);
  if (p.src_view) (void)hipFree(p.src_view);
  if (p.nrm_view) (void)hipFree(p.nrm_view);
  p.d2 = p.d2b = nullptr; p.src_view = p.nrm_view = nullptr;
  p.cap = 0; p.count = 0;
}

static hipError_t ensure_pairs(PairSet& p, size_t cap, bool with_normals) {
  if (cap <= p.cap && (!with_normals || p.nrm_view)) return hipSuccess;
  free_pairs(p);
  const size_t c = cap ? cap : 1;
  uint32_t** u[] = {&p.first, &p.second, &p.posd, &p.poss, &p.first2, &p.second2, &p.posd2, &p.poss2};
  for (auto q : u) HIP_TRY(hipMalloc(q, c * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&p.d2, c * sizeof(float)));
  HIP_TRY(hipMalloc(&p.d2b, c * sizeof(float)));
  HIP_TRY(hipMalloc(&p.src_view, c * sizeof(float4)));
  if (with_normals) HIP_TRY(hipMalloc(&p.nrm_view, c * sizeof(float4)));
  p.cap = c;
  return hipSuccess;
}

// direction: 1 = FIRST_TO_SECOND, 2 = BOTH.  For BOTH the caller has already run the forward search with the state's
// transform (fwd_pos / fwd_d2 by sorted-source position).  id_state: a device IcpState holding the identity.
hipError_t find_pairs(const GridDev& g, const float4* src_sorted, const float4* src_nrm_sorted, uint32_t ns, const IcpState* state,
                      const IcpState* id_state, float max_sq, int direction, bool reciprocal, double inlier_fraction, bool one_to_one,
                      const uint32_t* fwd_pos, const float* fwd_d2, PairSet& out, hipStream_t s) {
  const uint32_t nd = g.n;
  out.count = 0;
  const size_t ncand = (size_t)nd + (direction == 2 ? ns : 0);
  HIP_TRY(ensure_pairs(out, ncand, src_nrm_sorted != nullptr));
  if (nd == 0 || ns == 0) return hipSuccess;   // kd_tree_utilities.hpp:16-19: an empty side gives no correspondences

  // 1. q = T*s (sorted-source order) and a grid over it
  float* d_q = nullptr;
  GridBuildResult qg{};
  uint32_t *rev_pos = nullptr, *slots_in = nullptr, *slots_out = nullptr, *flags = nullptr, *offs = nullptr, *c_posd = nullptr, *c_poss = nullptr;
  float *rev_d2 = nullptr, *c_d2 = nullptr;
  unsigned long long *keys_in = nullptr, *keys_out = nullptr, *winner = nullptr, *sel_keys = nullptr;
  void *tmp = nullptr, *sel_state = nullptr;
  bool have_grid = false;
  hipError_t e = hipSuccess;
  do {
    if ((e = hipMalloc(&d_q, 3 * (size_t)ns * sizeof(float))) != hipSuccess) break;
    hipLaunchKernelGGL(k_transform_sorted, dim3(nblk(ns)), dim3(256), 0, s, src_sorted, ns, state, d_q);
    double mean[3];
    if ((e = build_grid(d_q, nullptr, ns, s, &qg, mean, 1.0)) != hipSuccess) break;
    have_grid = true;
    // 2. reverse search: the target points (in their grid order) against the grid over q
    if ((e = hipMalloc(&rev_pos, (size_t)nd * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&rev_d2, (size_t)nd * 4)) != hipSuccess) break;
    IterArgs r{};
    r.grid = qg.grid; r.src = g.pts; r.src_nrm = nullptr; r.ns = nd; r.max_sq = max_sq; r.state = id_state;
    r.nn_pos = rev_pos; r.nn_d2 = rev_d2; r.partials = nullptr; r.defer_mask = nullptr; r.tile_partials = nullptr; r.store_matches = 1;
    r.skip_if_inner_done = 0;
    launch_iter(r, IM_NONE, true, true, iter_num_blocks(nd), s);
    // 3. candidates -> keys (original indices) -> sort
    if ((e = hipMalloc(&keys_in, ncand * 8)) != hipSuccess) break;
    if ((e = hipMalloc(&keys_out, ncand * 8)) != hipSuccess) break;
    if ((e = hipMalloc(&slots_in, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&slots_out, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&c_posd, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&c_poss, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&c_d2, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&flags, ncand * 4)) != hipSuccess) break;
    if ((e = hipMalloc(&offs, ncand * 4)) != hipSuccess) break;
    hipLaunchKernelGGL(k_cand_reverse, dim3(nblk(nd)), dim3(256), 0, s, g.pts, nd, rev_pos, rev_d2, qg.grid.pts, src_sorted, keys_in, slots_in, c_posd,
                       c_poss, c_d2);
    if (direction == 2)
      hipLaunchKernelGGL(k_cand_forward, dim3(nblk(ns)), dim3(256), 0, s, g.pts, nd, src_sorted, ns, fwd_pos, fwd_d2, keys_in, slots_in, c_posd, c_poss,
                         c_d2);
    size_t tmp_bytes = 0;
    if ((e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, slots_in, slots_out, ncand, 0u, 64u, s)) != hipSuccess) break;
    if ((e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16)) != hipSuccess) break;
    if ((e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, slots_in, slots_out, ncand, 0u, 64u, s)) != hipSuccess) break;
    // 4. union / intersection / plain, then ordered compaction
    hipLaunchKernelGGL(k_mark, dim3(nblk(ncand)), dim3(256), 0, s, keys_out, (uint32_t)ncand, direction == 2 ? (reciprocal ? 2 : 1) : 0, flags);
    uint32_t m = 0;
    if ((e = scan_flags(flags, offs, (uint32_t)ncand, &m, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_compact_sorted, dim3(nblk(ncand)), dim3(256), 0, s, keys_out, slots_out, flags, offs, (uint32_t)ncand, c_posd, c_poss, c_d2,
                       out.first, out.second, out.posd, out.poss, out.d2);
    // 5. post-filters on the pair list (correspondence_search_kd_tree.hpp:224-225)
    if (m > 0 && inlier_fraction > 0.0 && inlier_fraction < 1.0) {
      if ((e = hipMalloc(&sel_keys, (size_t)m * 8)) != hipSuccess) break;
      if ((e = hipMalloc(&sel_state, filter_state_bytes())) != hipSuccess) break;
      launch_select_fraction(out.d2, m, inlier_fraction, sel_keys, sel_state, flags, s);
      uint32_t m2 = 0;
      if ((e = scan_flags(flags, offs, m, &m2, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_compact_pairs, dim3(nblk(m)), dim3(256), 0, s, flags, offs, m, out.first, out.second, out.posd, out.poss, out.d2, out.first2,
                         out.second2, out.posd2, out.poss2, out.d2b);
      std::swap(out.first, out.first2); std::swap(out.second, out.second2); std::swap(out.posd, out.posd2); std::swap(out.poss, out.poss2);
      std::swap(out.d2, out.d2b);
      m = m2;
    }
    if (m > 0 && one_to_one && direction == 1) {
      if ((e = hipMalloc(&winner, (size_t)ns * 8)) != hipSuccess) break;
      if ((e = hipMemsetAsync(winner, 0xFF, (size_t)ns * 8, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_o2o_min_pairs, dim3(nblk(m)), dim3(256), 0, s, out.first, out.poss, out.d2, m, winner);
      hipLaunchKernelGGL(k_o2o_flags_pairs, dim3(nblk(m)), dim3(256), 0, s, out.first, out.poss, out.d2, m, winner, flags);
      uint32_t m2 = 0;
      if ((e = scan_flags(flags, offs, m, &m2, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_compact_pairs, dim3(nblk(m)), dim3(256), 0, s, flags, offs, m, out.first, out.second, out.posd, out.poss, out.d2, out.first2,
                         out.second2, out.posd2, out.poss2, out.d2b);
      std::swap(out.first, out.first2); std::swap(out.second, out.second2); std::swap(out.posd, out.posd2); std::swap(out.poss, out.poss2);
      std::swap(out.d2, out.d2b);
      m = m2;
    }
    // 6. the view the accumulation kernels stream over: one "query" per pair
    if (m > 0)
      hipLaunchKernelGGL(k_gather_pair_view, dim3(nblk(m)), dim3(256), 0, s, src_sorted, src_nrm_sorted, out.poss, m, out.src_view,
                         src_nrm_sorted ? out.nrm_view : (float4*)nullptr);
    e = hipStreamSynchronize(s);
    out.count = m;
  } while (0);
  if (have_grid) free_grid(qg.grid);
  void* frees[] = {d_q, rev_pos, rev_d2, keys_in, keys_out, slots_in, slots_out, c_posd, c_poss, c_d2, flags, offs, tmp, winner, sel_keys, sel_state};
  for (void* f : frees) if (f) (void)hipFree(f);
  return e;
}

}  // namespace cilhip
