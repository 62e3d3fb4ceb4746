// extract.hip -- result extraction (matches back to original order, packed keys of the target-sharded protocol, distances on demand), the tie counters and order tables by position, computeResiduals(); split from kernels.hip.
#include "search_device.hpp"

namespace cilhip {

// ---- result extraction ----------------------------------------------------------------------------
__global__ void k_scatter_nn(const float4* __restrict__ src_sorted, const float4* __restrict__ dst_sorted,
                             const uint32_t* __restrict__ nn_pos, const float* __restrict__ nn_d2, uint32_t ns,
                             uint32_t* out_idx, float* out_d2) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t orig = __float_as_uint(src_sorted[i].w);
    const uint32_t pos = nn_pos[i];
    if (out_idx) out_idx[orig] = (pos == NONE_U32) ? NONE_U32 : __float_as_uint(dst_sorted[pos].w);
    if (out_d2) out_d2[orig] = nn_d2[i];
  }
}

void launch_scatter_nn(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos,
                       const float* nn_d2, uint32_t ns, uint32_t* out_idx, float* out_d2, hipStream_t s) {
  if (ns == 0) return;
  const int nb = (int)((ns + 255) / 256 < 4096 ? (ns + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_scatter_nn, dim3(nb), dim3(256), 0, s, src_sorted, dst_sorted, nn_pos, nn_d2, ns, out_idx, out_d2);
}

__global__ void k_count_found(const uint32_t* __restrict__ nn_pos, uint32_t ns, unsigned long long* out) {
  unsigned long long c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x)
    c += (nn_pos[i] != NONE_U32) ? 1ull : 0ull;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// reorder per-source attributes (normals) into the sorted-source order: out[i] = {in[orig(i)], 0}
__global__ void k_gather_by_w(const float4* __restrict__ src_sorted, const float* __restrict__ in_xyz, uint32_t ns, float4* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t o = __float_as_uint(src_sorted[i].w);
    out[i] = make_float4(in_xyz[3 * (size_t)o], in_xyz[3 * (size_t)o + 1], in_xyz[3 * (size_t)o + 2], 0.0f);
  }
}

__global__ void k_gather1_by_w(const float4* __restrict__ src_sorted, const float* __restrict__ in, uint32_t ns, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) out[i] = in[__float_as_uint(src_sorted[i].w)];
}
void launch_gather1_by_w(const float4* src_sorted, const float* in, uint32_t ns, float* out, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_gather1_by_w, dim3((ns + 255u) / 256u), dim3(256), 0, s, src_sorted, in, ns, out);
}
void launch_gather_by_w(const float4* src_sorted, const float* in_xyz, uint32_t ns, float4* out, hipStream_t s) {
  if (ns == 0) return;
  const int nb = (int)((ns + 255) / 256 < 4096 ? (ns + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_gather_by_w, dim3(nb), dim3(256), 0, s, src_sorted, in_xyz, ns, out);
}

// ---- target-sharded runs (SURVEY.md 8(e) partitioning A) -------------------------------------------
// Every rank searches ALL source points against its own target shard and publishes, per source point
// (ORIGINAL source order, so the ranks' arrays line up), the packed key (bits(d2) << 32) | GLOBAL target
// index; "none" = 0x7fff...f so that a signed-int64 MIN all-reduce picks the globally nearest target
// (ties -> lowest global index, exactly the single-GPU rule).
constexpr unsigned long long KEY_NONE = 0x7fffffffffffffffull;

__global__ void k_pack_keys(const float4* __restrict__ src_sorted, const float4* __restrict__ dst_sorted,
                            const uint32_t* __restrict__ nn_pos, const float* __restrict__ nn_d2, uint32_t ns,
                            uint32_t index_offset, unsigned long long* keys) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t orig = __float_as_uint(src_sorted[i].w);
    const uint32_t pos = nn_pos[i];
    keys[orig] = (pos == NONE_U32) ? KEY_NONE
                                   : (((unsigned long long)__float_as_uint(nn_d2[i]) << 32) |
                                      (unsigned long long)(__float_as_uint(dst_sorted[pos].w) + index_offset));
  }
}

// after the MIN all-reduce: keep the pairs whose winning target lives in THIS rank's shard
// (tie_counter != null -- option "tie_rule" in force, no order tables yet: a shard whose own nearest point is exactly as far as the
//  winner's but is not the winner has met a tie ACROSS shards; counted like the ties a search notices inside its shard)
__global__ void k_keys_to_pos(const float4* __restrict__ src_sorted, const unsigned long long* __restrict__ keys,
                              const uint32_t* __restrict__ inv_perm, uint32_t ns, uint32_t index_offset, uint32_t n_local,
                              uint32_t* nn_pos, float* nn_d2, unsigned int* tie_counter) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const unsigned long long k = keys[__float_as_uint(src_sorted[i].w)];
    uint32_t pos = NONE_U32;
    if (k != KEY_NONE) {
      const uint32_t gidx = (uint32_t)k;
      if (gidx >= index_offset && gidx - index_offset < n_local) pos = inv_perm[gidx - index_offset];
      if (tie_counter != nullptr && pos == NONE_U32 && nn_pos[i] != NONE_U32 && __float_as_uint(nn_d2[i]) == (uint32_t)(k >> 32)) atomicAdd(tie_counter, 1u);
    }
    nn_pos[i] = pos;
    nn_d2[i] = __uint_as_float((uint32_t)(k >> 32));
  }
}

// ---- the reference's tie order ACROSS target shards ---------------------------------------------------------------------------------
// Inside a shard tie_settle() leaves the shard's first-met point among the equidistant ones (the traversal order of one query is a
// total order over the WHOLE target's tree: the first of a subset is well defined).  Between shards the MIN of (d2, global index)
// would pick the lowest index instead.  So a second key per query: the position of the shard's match in the query's traversal -- per
// level of the tree one bit, 0 = the child searchLevel descends into first (nanoflann.hpp:1931-1947), most significant = the root's
// children, then the slot inside the leaf (leaf_max_size 10 < 16) -- published by every shard whose match is at the winning distance;
// the MIN over the shards is the first-met point of the whole target, its owner recognises its own key.  Depth <= 58 (checked when
// the tables are loaded).
__device__ __forceinline__ unsigned long long tie_rank(const TieDev& tt, float qx, float qy, float qz, uint32_t pos) {
  const uint2 ls = tt.leaf_slot[pos];
  uint4 N = tt.nodes[ls.x];
  unsigned long long key = (unsigned long long)((ls.y - N.z) & 15u);      // (a leaf's record: z = the slot of its first point)
  while ((N.y >> 3) != 0u) {
    const uint4 P = tt.nodes[N.x];
    const uint32_t feat = (P.y >> 1) & 3u;
    const float val = feat == 0u ? qx : (feat == 1u ? qy : qz);
    const float diff1 = __fsub_rn(val, __uint_as_float(P.z)), diff2 = __fsub_rn(val, __uint_as_float(P.w));
    const uint32_t first_is_second = __fadd_rn(diff1, diff2) < 0.0f ? 0u : 1u;
    if ((N.y & 1u) != first_is_second) key |= 1ull << (62u - (N.y >> 3));
    N = P;
  }
  return key;
}
// own[orig] = out[orig] = the traversal key of this shard's match if it is at the winning distance, "none" otherwise
__global__ void k_order_keys(const float4* __restrict__ src_sorted, const IcpState* __restrict__ state, const unsigned long long* __restrict__ win,
                             const uint32_t* __restrict__ nn_pos, const float* __restrict__ nn_d2, uint32_t ns, TieDev tt,
                             unsigned long long* own, unsigned long long* out) {
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = state->T[i];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const float4 s4 = src_sorted[i];
    const uint32_t orig = __float_as_uint(s4.w);
    const unsigned long long k = win[orig];
    const uint32_t lp = nn_pos[i];
    unsigned long long ok = KEY_NONE;
    if (lp != NONE_U32 && k != KEY_NONE && __float_as_uint(nn_d2[i]) == (uint32_t)(k >> 32)) {
      float qx, qy, qz;
      transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
      ok = tie_rank(tt, qx, qy, qz, lp);
    }
    own[orig] = ok;
    out[orig] = ok;
  }
}
// after the MIN all-reduce of the traversal keys: this shard keeps the matches whose key came back
__global__ void k_select_ordered(const float4* __restrict__ src_sorted, const unsigned long long* __restrict__ own,
                                 const unsigned long long* __restrict__ reduced, const unsigned long long* __restrict__ win, uint32_t ns,
                                 uint32_t* nn_pos, float* nn_d2) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t orig = __float_as_uint(src_sorted[i].w);
    const unsigned long long o = own[orig];
    if (o == KEY_NONE || o != reduced[orig]) nn_pos[i] = NONE_U32;
    nn_d2[i] = __uint_as_float((uint32_t)(win[orig] >> 32));
  }
}

__global__ void k_inv_perm(const float4* __restrict__ dst_sorted, uint32_t n, uint32_t* inv) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    inv[__float_as_uint(dst_sorted[i].w)] = i;
}

static inline int blocks_for(uint32_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096) + (n == 0); }

void launch_pack_keys(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos, const float* nn_d2,
                      uint32_t ns, uint32_t index_offset, unsigned long long* keys, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_pack_keys, dim3(blocks_for(ns)), dim3(256), 0, s, src_sorted, dst_sorted, nn_pos, nn_d2, ns, index_offset, keys);
}
void launch_keys_to_pos(const float4* src_sorted, const unsigned long long* keys, const uint32_t* inv_perm, uint32_t ns,
                        uint32_t index_offset, uint32_t n_local, uint32_t* nn_pos, float* nn_d2, hipStream_t s, unsigned int* tie_counter) {
  if (ns) hipLaunchKernelGGL(k_keys_to_pos, dim3(blocks_for(ns)), dim3(256), 0, s, src_sorted, keys, inv_perm, ns, index_offset, n_local, nn_pos, nn_d2, tie_counter);
}
void launch_order_keys(const float4* src_sorted, const IcpState* state, const unsigned long long* win, const uint32_t* nn_pos, const float* nn_d2,
                       uint32_t ns, const TieDev& tt, unsigned long long* own, unsigned long long* out, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_order_keys, dim3(blocks_for(ns)), dim3(256), 0, s, src_sorted, state, win, nn_pos, nn_d2, ns, tt, own, out);
}
void launch_select_ordered(const float4* src_sorted, const unsigned long long* own, const unsigned long long* reduced, const unsigned long long* win,
                           uint32_t ns, uint32_t* nn_pos, float* nn_d2, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_select_ordered, dim3(blocks_for(ns)), dim3(256), 0, s, src_sorted, own, reduced, win, ns, nn_pos, nn_d2);
}
void launch_inv_perm(const float4* dst_sorted, uint32_t n, uint32_t* inv, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_inv_perm, dim3(blocks_for(n)), dim3(256), 0, s, dst_sorted, n, inv);
}

struct T16 { float v[16]; };
__global__ void k_fill_d2(const float4* __restrict__ src_sorted, const float4* __restrict__ dst_sorted, const uint32_t* __restrict__ nn_pos, T16 T, uint32_t ns,
                          float* __restrict__ nn_d2) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const uint32_t pos = nn_pos[i];
    float e = 0.0f;
    if (pos != NONE_U32) {
      const float4 s4 = src_sorted[i], p = dst_sorted[pos];
      float qx, qy, qz;
      transform_point(T.v, s4.x, s4.y, s4.z, qx, qy, qz);
      e = d2_pinned(qx, qy, qz, p.x, p.y, p.z);
    }
    nn_d2[i] = e;
  }
}
void launch_fill_d2(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos, const float T[16], uint32_t ns, float* nn_d2, hipStream_t s) {
  if (ns == 0) return;
  T16 t;
  for (int i = 0; i < 16; ++i) t.v[i] = T[i];
  hipLaunchKernelGGL(k_fill_d2, dim3(blocks_for(ns)), dim3(256), 0, s, src_sorted, dst_sorted, nn_pos, t, ns, nn_d2);
}

// ---- ties: queries whose nearest target point is not unique ---------------------------------------------------------------
// Two target points at EXACTLY the same pinned f32 distance from a query (duplicated points; a sensor's lattice) are the one
// place where this engine and the reference may name different correspondences: the engine keeps the lowest target index, the
// reference's nanoflann the candidate its kd-tree traversal meets first (core/kd_tree.hpp:82-90) -- both are exact nearest
// neighbours.  This diagnostic counts such queries under a transform: the exact search once more (shells; a cell whose gap equals
// the best distance is scanned, so every tied candidate is met), remembering whether the winning distance was met on a second point.
__device__ __forceinline__ void scan_range4_tie(const float4* __restrict__ pts, uint32_t beg, uint32_t end, float qx, float qy, float qz, NN& best, bool& tie) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t jj = min(j + (uint32_t)k, last);
      const float4 p = pts[jj];
      const float e = d2_pinned(qx, qy, qz, p.x, p.y, p.z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(p.w);
      const bool same_d = (uint32_t)(key >> 32) == (uint32_t)(best.key >> 32);
      if (key < best.key) { tie = same_d && best.pos != NONE_U32; best.key = key; best.pos = jj; }
      else if (same_d && key != best.key && best.pos != NONE_U32) tie = true;
    }
  }
}
__global__ __launch_bounds__(256) void k_count_ties(GridDev g, const float4* __restrict__ src, uint32_t ns, T16c T, float max_sq, unsigned long long* out) {
  unsigned int mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    const float4 s4 = src[i];
    float qx, qy, qz;
    transform_point(T.v, s4.x, s4.y, s4.z, qx, qy, qz);
    NN best;
    best.key = ((unsigned long long)__float_as_uint(max_sq) << 32);
    best.pos = NONE_U32;
    bool tie = false;
    const float BIG = 1.0e9f;
    const int cx = (int)floorf(fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG)), cy = (int)floorf(fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG)),
              cz = (int)floorf(fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG));
    const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin), gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin),
                gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
    if ((gx * gx + gy * gy + gz * gz) * KSHRINK >= max_sq) continue;      // farther than the radius from the whole grid
    for (int s = max(0, max(max(-cx, cx - (g.nx - 1)), max(max(-cy, cy - (g.ny - 1)), max(-cz, cz - (g.nz - 1)))));; ++s) {
      const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1), y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
      for (int z = z0; z <= z1; ++z) {
        const float zl = g.oz + (float)z * g.cell;
        const float az = axis_gap(qz, zl, zl + g.cell, g.margin);
        for (int y = y0; y <= y1; ++y) {
          const bool face = (z == cz - s) || (z == cz + s) || (y == cy - s) || (y == cy + s);
          const float yl = g.oy + (float)y * g.cell;
          const float ay = axis_gap(qy, yl, yl + g.cell, g.margin);
          if ((az * az + ay * ay) * KSHRINK > __uint_as_float((uint32_t)(best.key >> 32))) continue;
          const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          if (face) {
            const int xa = max(cx - s, 0), xb = min(cx + s, g.nx - 1);
            if (xa <= xb) scan_range4_tie(g.pts, g.cell_start[row + xa], g.cell_start[row + xb + 1], qx, qy, qz, best, tie);
          } else {
            if (cx - s >= 0 && cx - s < g.nx) scan_range4_tie(g.pts, g.cell_start[row + cx - s], g.cell_start[row + cx - s + 1], qx, qy, qz, best, tie);
            if (s > 0 && cx + s >= 0 && cx + s < g.nx) scan_range4_tie(g.pts, g.cell_start[row + cx + s], g.cell_start[row + cx + s + 1], qx, qy, qz, best, tie);
          }
        }
      }
      float b = INFINITY;      // lower bound on the distance to anything not yet scanned
      if (cx - s > 0) b = fminf(b, qx - (g.ox + (float)(cx - s) * g.cell));
      if (cx + s + 1 < g.nx) b = fminf(b, (g.ox + (float)(cx + s + 1) * g.cell) - qx);
      if (cy - s > 0) b = fminf(b, qy - (g.oy + (float)(cy - s) * g.cell));
      if (cy + s + 1 < g.ny) b = fminf(b, (g.oy + (float)(cy + s + 1) * g.cell) - qy);
      if (cz - s > 0) b = fminf(b, qz - (g.oz + (float)(cz - s) * g.cell));
      if (cz + s + 1 < g.nz) b = fminf(b, (g.oz + (float)(cz + s + 1) * g.cell) - qz);
      if (b == INFINITY) break;
      b -= g.margin;
      if (b > 0.0f && __uint_as_float((uint32_t)(best.key >> 32)) < b * b * KSHRINK) break;
    }
    mine += (best.pos != NONE_U32 && tie) ? 1u : 0u;
  }
  const double tot = wave_sum((double)mine);
  if ((threadIdx.x & 63) == 0 && tot > 0.0) atomicAdd(out, (unsigned long long)tot);
}
void launch_count_ties(const GridDev& g, const float4* src_sorted, uint32_t ns, const float T[16], float max_sq, unsigned long long* out, hipStream_t s) {
  (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
  if (ns == 0 || g.n == 0) return;
  T16c t;
  for (int i = 0; i < 16; ++i) t.v[i] = T[i];
  hipLaunchKernelGGL(k_count_ties, dim3((int)((ns + 255u) / 256u < 4096u ? (ns + 255u) / 256u : 4096u)), dim3(256), 0, s, g, src_sorted, ns, t, max_sq, out);
}

// the order tables of option "tie_rule" (tie_build.hip) arrive by ORIGINAL target index; the searches know sorted positions
__global__ void k_tie_tables_by_position(const float4* __restrict__ dst_sorted, uint32_t n, const uint32_t* __restrict__ leaf_by_index,
                                         const uint32_t* __restrict__ slot_by_index, uint2* __restrict__ leaf_slot) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t o = __float_as_uint(dst_sorted[j].w);
  leaf_slot[j] = make_uint2(leaf_by_index[o], slot_by_index[o]);
}
void launch_tie_tables_by_position(const float4* dst_sorted, uint32_t n, const uint32_t* leaf_by_index, const uint32_t* slot_by_index, uint2* leaf_slot, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_tie_tables_by_position, dim3((n + 255u) / 256u), dim3(256), 0, s, dst_sorted, n, leaf_by_index, slot_by_index, leaf_slot);
}

void launch_count_found(const uint32_t* nn_pos, uint32_t ns, unsigned long long* out, hipStream_t s) {
  (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
  if (ns == 0) return;
  const int nb = (int)((ns + 255) / 256 < 2048 ? (ns + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_count_found, dim3(nb), dim3(256), 0, s, nn_pos, ns, out);
}

// computeResiduals() of both ICP classes (icp_single_transform_combined_metric.hpp:220-243,
// icp_single_transform_point_to_point_metric.hpp:68-85): unbounded exact 1-NN, then the metric value.
__global__ __launch_bounds__(256) void k_residuals(IterArgs a, int metric, float w_p2p, float w_p2pl, float* out) {
  __shared__ uint2 worklist[LIST_CAP * ITER_THREADS];
  uint2* lst = worklist + threadIdx.x;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = a.state->T[i];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.ns; i += gridDim.x * blockDim.x) {
    const float4 s4 = a.src[i];
    float qx, qy, qz;
    transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
    NN best;
    nn_search(a.grid, qx, qy, qz, a.max_sq, best, lst);
    if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)
      best.pos = tie_settle(a.grid, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)));
    float v = __uint_as_float(0x7fc00000u);  // NaN when the target is empty (:221-224)
    if (best.pos != NONE_U32) {
      const float4 p = a.grid.pts[best.pos];
      const float dx = __fsub_rn(p.x, qx), dy = __fsub_rn(p.y, qy), dz = __fsub_rn(p.z, qz);
      const float sq = __fadd_rn(__fmul_rn(dx, dx), __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dz, dz)));  // squaredNorm()
      if (metric == 0) {
        v = sq;
      } else {
        float4 nv = a.grid.nrm[best.pos];
        if (a.src_nrm) {  // `normal += src_normals_.col(i)` -- the UNtransformed source normal, as the reference (:237)
          const float4 sn = a.src_nrm[i];
          nv.x = __fadd_rn(nv.x, sn.x); nv.y = __fadd_rn(nv.y, sn.y); nv.z = __fadd_rn(nv.z, sn.z);
        }
        const float pd =__fadd_rn(__fmul_rn(nv.x, dx), __fadd_rn(__fmul_rn(nv.y, dy), __fmul_rn(nv.z, dz)));
        v = __fadd_rn(__fmul_rn(w_p2p, sq), __fmul_rn(__fmul_rn(w_p2pl, pd), pd));
      }
    }
    out[__float_as_uint(s4.w)] = v;
  }
}

void launch_residuals(const IterArgs& a, int metric, float w_p2p, float w_p2pl, float* out, hipStream_t s) {
  if (a.ns == 0) return;
  const int nb = iter_num_blocks(a.ns);
  hipLaunchKernelGGL(k_residuals, dim3(nb), dim3(256), 0, s, a, metric, w_p2p, w_p2pl, out);
}

}  // namespace cilhip
