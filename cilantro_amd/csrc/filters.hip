// filters.hip -- the correspondence post-filters of the engine, on the device:
//   filterCorrespondencesFraction   core/correspondence.hpp:57-66   (inlier_fraction_ in (0,1))
//   filterCorrespondencesOneToOne   core/correspondence.hpp:68-100  (SECOND_TO_FIRST branch)
// applied, in this order, after the search (correspondence_search_kd_tree.hpp:224-225).
//
// The reference sorts the whole correspondence set (std::sort, unstable) for each filter.  Here nothing is
// sorted: every match gets the unique 64-bit key  (bits(d2) << 32) | original source index  and
//   * fraction: an 8-pass MSB-first radix SELECT finds the k-th smallest key, k = llround(f * n); matches
//     with a larger key are dropped.  Among equal values the reference keeps an unspecified subset; the
//     key order pins it to "lowest source index first", independent of grid / launch configuration.
//   * one-to-one: atomicMin of the key per target point; a match survives iff it holds the minimum.
// Dropped matches are simply marked NONE in nn_pos, so the accumulation kernels need no change.
#include "internal.hpp"

namespace cilhip {

constexpr unsigned long long KEY_DROPPED = 0xFFFFFFFFFFFFFFFFull;

struct SelectState {
  unsigned long long prefix;      // bits decided so far (high bytes)
  unsigned long long k;           // 1-based rank still to locate inside the current prefix bucket
  unsigned long long n_found;
  unsigned long long threshold;   // result: k-th smallest key (KEY_DROPPED-1 semantics: keep key <= threshold)
  int keep_none;                  // k == 0
  int pad;
  unsigned int hist[256];
};

__global__ void k_build_keys(const float4* __restrict__ src_sorted, const uint32_t* __restrict__ nn_pos,
                             const float* __restrict__ nn_d2, uint32_t ns, unsigned long long* keys, SelectState* st) {
  unsigned long long cnt = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const bool found = nn_pos[i] != NONE_U32;
    keys[i] = found ? (((unsigned long long)__float_as_uint(nn_d2[i]) << 32) | __float_as_uint(src_sorted[i].w)) : KEY_DROPPED;
    cnt += found ? 1ull : 0ull;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&st->n_found, cnt);
}

__global__ void k_select_init(SelectState* st, double fraction) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->prefix = 0;
    const long long k = llround(fraction * (double)st->n_found);      // correspondence.hpp:63
    st->k = k > 0 ? (unsigned long long)k : 0ull;
    if (st->k > st->n_found) st->k = st->n_found;
    st->keep_none = (st->k == 0);
    st->threshold = 0;
  }
  if (blockIdx.x == 0) st->hist[threadIdx.x & 255] = 0;
}

// histogram of byte `byte` (7 = most significant) over the keys that match the decided prefix
__global__ void k_select_hist(const unsigned long long* __restrict__ keys, uint32_t ns, SelectState* st, int byte) {
  __shared__ unsigned int h[256];
  h[threadIdx.x & 255] = 0;
  __syncthreads();
  const unsigned long long prefix = st->prefix;
  const int shift = byte * 8;
  const unsigned long long himask = (byte == 7) ? 0ull : (~0ull << (shift + 8));
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    if (k != KEY_DROPPED && (k & himask) == prefix) atomicAdd(&h[(k >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  const unsigned int v = h[threadIdx.x & 255];
  if (threadIdx.x < 256 && v) atomicAdd(&st->hist[threadIdx.x], v);
}

__global__ void k_select_pick(SelectState* st, int byte) {
  if (threadIdx.x == 0) {
    if (!st->keep_none) {
      unsigned long long k = st->k, acc = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (acc + st->hist[b] >= k) break;
        acc += st->hist[b];
      }
      if (b > 255) b = 255;
      st->k = k - acc;
      st->prefix |= ((unsigned long long)b) << (byte * 8);
      if (byte == 0) st->threshold = st->prefix;
    }
  }
  __syncthreads();
  st->hist[threadIdx.x & 255] = 0;
}

__global__ void k_apply_fraction(const unsigned long long* __restrict__ keys, uint32_t ns, const SelectState* st, uint32_t* nn_pos) {
  const bool none = st->keep_none != 0;
  const unsigned long long thr = st->threshold;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    if (k != KEY_DROPPED && (none || k > thr)) nn_pos[i] = NONE_U32;
  }
}

__global__ void k_o2o_min(const float4* __restrict__ src_sorted, const uint32_t* __restrict__ nn_pos, const float* __restrict__ nn_d2,
                          uint32_t ns, unsigned long long* winner) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t pos = nn_pos[i];
    if (pos != NONE_U32)
      atomicMin(&winner[pos], ((unsigned long long)__float_as_uint(nn_d2[i]) << 32) | __float_as_uint(src_sorted[i].w));
  }
}

__global__ void k_o2o_apply(const float4* __restrict__ src_sorted, const float* __restrict__ nn_d2, uint32_t ns,
                            const unsigned long long* __restrict__ winner, uint32_t* nn_pos) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t pos = nn_pos[i];
    if (pos != NONE_U32 &&
        winner[pos] != (((unsigned long long)__float_as_uint(nn_d2[i]) << 32) | __float_as_uint(src_sorted[i].w)))
      nn_pos[i] = NONE_U32;
  }
}

static inline int nblk(uint32_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048) + (n == 0); }

// scratch: keys [ns] u64, state [1] SelectState (both owned by the caller)
void launch_filter_fraction(const float4* src_sorted, uint32_t* nn_pos, const float* nn_d2, uint32_t ns, double fraction,
                            unsigned long long* keys, void* state, hipStream_t s) {
  if (ns == 0 || !(fraction > 0.0 && fraction < 1.0)) return;
  SelectState* st = static_cast<SelectState*>(state);
  (void)hipMemsetAsync(st, 0, sizeof(SelectState), s);
  hipLaunchKernelGGL(k_build_keys, dim3(nblk(ns)), dim3(256), 0, s, src_sorted, nn_pos, nn_d2, ns, keys, st);
  hipLaunchKernelGGL(k_select_init, dim3(1), dim3(256), 0, s, st, fraction);
  for (int byte = 7; byte >= 0; --byte) {
    hipLaunchKernelGGL(k_select_hist, dim3(nblk(ns)), dim3(256), 0, s, keys, ns, st, byte);
    hipLaunchKernelGGL(k_select_pick, dim3(1), dim3(256), 0, s, st, byte);
  }
  hipLaunchKernelGGL(k_apply_fraction, dim3(nblk(ns)), dim3(256), 0, s, keys, ns, st, nn_pos);
}

size_t filter_state_bytes() { return sizeof(SelectState); }

// Generic form for a plain list of values (the pair lists of the other search directions): keep the
// k = llround(fraction * n) smallest, ties in list order.  flags[i] = 1 for kept entries.
__global__ void k_build_keys_pos(const float* __restrict__ d2, uint32_t n, unsigned long long* keys, SelectState* st) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    keys[i] = ((unsigned long long)__float_as_uint(d2[i]) << 32) | (unsigned long long)i;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_found = n;
}
__global__ void k_flags_fraction(const unsigned long long* __restrict__ keys, uint32_t n, const SelectState* st, uint32_t* flags) {
  const bool none = st->keep_none != 0;
  const unsigned long long thr = st->threshold;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) flags[i] = (!none && keys[i] <= thr) ? 1u : 0u;
}
void launch_select_fraction(const float* d2, uint32_t n, double fraction, unsigned long long* keys, void* state, uint32_t* flags, hipStream_t s) {
  if (n == 0) return;
  SelectState* st = static_cast<SelectState*>(state);
  (void)hipMemsetAsync(st, 0, sizeof(SelectState), s);
  hipLaunchKernelGGL(k_build_keys_pos, dim3(nblk(n)), dim3(256), 0, s, d2, n, keys, st);
  hipLaunchKernelGGL(k_select_init, dim3(1), dim3(256), 0, s, st, fraction);
  for (int byte = 7; byte >= 0; --byte) {
    hipLaunchKernelGGL(k_select_hist, dim3(nblk(n)), dim3(256), 0, s, keys, n, st, byte);
    hipLaunchKernelGGL(k_select_pick, dim3(1), dim3(256), 0, s, st, byte);
  }
  hipLaunchKernelGGL(k_flags_fraction, dim3(nblk(n)), dim3(256), 0, s, keys, n, st, flags);
}

// winner: [n_target] u64 scratch
void launch_filter_one_to_one(const float4* src_sorted, uint32_t* nn_pos, const float* nn_d2, uint32_t ns,
                              unsigned long long* winner, uint32_t n_target, hipStream_t s) {
  if (ns == 0 || n_target == 0) return;
  (void)hipMemsetAsync(winner, 0xFF, (size_t)n_target * sizeof(unsigned long long), s);
  hipLaunchKernelGGL(k_o2o_min, dim3(nblk(ns)), dim3(256), 0, s, src_sorted, nn_pos, nn_d2, ns, winner);
  hipLaunchKernelGGL(k_o2o_apply, dim3(nblk(ns)), dim3(256), 0, s, src_sorted, nn_d2, ns, winner, nn_pos);
}

}  // namespace cilhip
