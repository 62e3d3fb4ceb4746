// kernels.hip -- the per-iteration hot path of rigid ICP as hand-written HIP for gfx950 (CDNA4).
//
// Per ICP iteration these kernels replace four OpenMP loops of the reference:
//   q_i = T*s_i                         correspondence_search/common_transformable_feature_adaptors.hpp:28-33
//   1-NN of q_i in dst within r^2       correspondence_search/correspondence_search_kd_tree_utilities.hpp:26-33
//                                       (nanoflann searchLevel, 3rd_party/nanoflann/nanoflann.hpp:1885-1961)
//   second transform of src             core/space_transformations.hpp:203-216   (eliminated: q is re-formed in registers)
//   normal-equation / moment sums       registration/transform_estimation.hpp:25-34, :298-320, :328-343
// as, per iteration, ONE of three forms of the search (+ accumulation):
//   * k_tile_boxes + k_search_tiled<metric> + k_search_deferred<metric>: the LDS-tiled search with the accumulation inside
//     the tile (f64 MFMA rank update) -- the first iterations of a run on a large cloud;
//   * k_warm<metric, rec>: the search warm-started from the previous iteration's matches (settled by k_self_nn's
//     nearest-other-point table for nearly all queries, listed and searched densely for the rest) + the same MFMA
//     accumulation, streaming 40 B per query -- every later iteration near alignment;
//   * k_search_tiled<none> / k_iter<none, search, store> (per-lane search) followed by the streaming accumulation
//     k_iter<metric, no search>: sources far from alignment, engine post-filters, weight evaluators, later Gauss-Newton
//     steps, small clouds (k_iter<metric, search> is the per-lane fused form, option "fused");
// variants: k_search_tiled<none, feat6> (6-D point+normal features), k_acc_reverse (reverse matches of the other search
// directions, accumulated where they are found).  Then k_reduce_stage1 and the one-block epilogue k_solve, which reduces
// the partial sums in a fixed order, performs the 3x3 SVD / 6x6 LDL^T solve + compose on the device (solve.hpp) and
// publishes the loop state to the host, so all iterations of IterativeClosestPointBase::estimate()
// (registration/icp_base.hpp:68-87) are enqueued without a host round trip on the critical path.  DESIGN.md section 5
// has the kernel table, the measurements and what was tried.
//
// Design notes (MI355X):
//   * no MFMA in the search: K=3 contraction, and the -2q.p+|p|^2 form would change the rounding of d2 and break
//     index parity with the reference.
//   * wave64: one query per lane; queries are pre-sorted by target-grid cube / cell so the 64 lanes of a
//     wave walk the same few cell runs (LDS tile, or loads that coalesce / broadcast in the TA and hit L1/L2).
//   * blockIdx -> work mapping is XCD-aware: hardware places block b on XCD b%8, so virtual block
//     (b%8)*(nb/8)+b/8 gives every XCD one contiguous eighth of the (spatially sorted) queries and
//     each private 4 MiB L2 caches one slab of the target instead of all of it.
//   * the normal-equation sums are a rank update Z += z z^T of per-correspondence f32 term vectors: v_mfma_f64_16x16x4_f64
//     work (products of f32 terms exact in f64, sums in f64) with the wave's 16x16 tile in registers; the streaming
//     kernel keeps per-lane f64 accumulators instead.  Fixed orders everywhere (per wave, per block, across blocks)
//     => bitwise run-to-run reproducible (the reference's OpenMP reduction is not).
#include "internal.hpp"
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>

namespace cilhip {

// Kernel timing without extra packets in the queue: a launcher that supports it attaches the caller's events to its kernels' OWN
// dispatch packets (hipExtLaunchKernelGGL: start of the first kernel, stop of the last) instead of the caller recording events
// around it -- an event recorded between two dependent kernels costs the device ~6 us of idle time each (measured: 11.5 us per
// warm-started iteration of 110).  set_launch_events() arms the NEXT such launcher call of this thread.
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }
template <typename... Args, typename F = void (*)(Args...)>
static inline void launch_ev(F kernel, dim3 grid, dim3 block, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, Args... args) {
  if (ev_start != nullptr || ev_stop != nullptr) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, ev_start, ev_stop, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
}

// LDS-tiled search geometry: cube of 2^L cells per axis, <= TILE_QUERIES queries per tile,
// TILE_THREADS threads per workgroup.  (4^3 cells / 256 queries / 256 threads, or 8^3 / 2048 / 1024.)

#ifndef CILHIP_CAND
#define CILHIP_CAND 4 /* candidates per lane per trip of the flattened work-list loop */
#endif

#define KSHRINK 0.99999905f /* 1 - 2^-20: covers the <= 2^-22 relative rounding of the f32 d2 */

// d2 exactly as nanoflann's L2_Adaptor::evalMetric computes it for DIM=3
// (nanoflann.hpp:570-604: only the tail loop runs): ((dx*dx)+(dy*dy))+(dz*dz), dx = q.x - p.x,
// every operation individually rounded (no FMA contraction).
__device__ __forceinline__ float d2_pinned(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- margin keys (IterArgs::nn_lb, the match records' fourth component) ------------------------------------
// A search that has PROVEN its result for a query q also knows a lower bound on the distance from q to every target point but
// the match: the second smallest squared distance it evaluated (every point of the searched block was evaluated) and the gap
// from q to the block's faces (everything else lies beyond), whichever is smaller.  Without a match the same bound holds for
// every target point.  The key stores it relative to the run's motion clock: B = +-((lb - eps) + acc), rounded DOWN, so that
// under a later transform of the same run  B - (acc' + eps')  is still such a bound (a query moves by at most acc' - acc
// between the two searches; eps, eps': the rounding of the two computed queries).
struct MotionRef { float acc, eps; };      // IcpState::motion_acc / ::motion_eps under the transform being searched
// Would a query with this bound (lb on the other points, d2 to its match) have to be searched again by a warm-started iteration
// if the next update moves the source as far as the last one did?  The LB forms of the cold kernels count these (and the queries
// that leave without a bound) into the `listed` counters: the host enters the warm-started form only where it will pay.
__device__ __forceinline__ bool margin_is_small(bool found, float second_sq, float gap, float best_sq, float max_sq, float step) {
  const float lb = fminf(__fsqrt_rn(second_sq), gap);
  const float need = found ? __fsqrt_rn(best_sq) : __fsqrt_rn(max_sq);      // (no match: the bound has to stay beyond the radius)
  return !(lb - need > 2.0f * step);
}
#define MARGIN_NONE_NO_MATCH (-1.17549435e-38f) /* -FLT_MIN: no match, no bound known (+0: a match, no bound known) */
__device__ __forceinline__ float margin_key(bool found, float second_sq, float gap, const MotionRef& m) {
  // sqrt of a pinned squared distance: the true distance is at least that times (1 - 2^-22), the device's square root is within 1 ulp
  const float lb = fminf(__fsqrt_rn(second_sq) * 0.999999f, gap);
  const float b = __fmul_rn(__fadd_rn(__fsub_rn(lb, m.eps), m.acc), 0.9999995f);
  return found ? fmaxf(b, 0.0f) : -fmaxf(b, 1.17549435e-38f);
}

// The accumulating tile kernel carries its two queries' keys through the second search and the barrier packed into ONE register:
// 16 bits each -- sign = no match, 15 bits = (lb - eps) in units of cell / 8192, rounded DOWN (values beyond 4 cells: clamped; a
// bound may always be smaller) -- and forms the key proper where it writes the record.
__device__ __forceinline__ uint32_t margin_q15(bool found, float second_sq, float gap, const MotionRef& m, float inv_cell) {
  const float lb = __fsub_rn(fminf(__fsqrt_rn(second_sq) * 0.999999f, gap), m.eps);
  const float u = fminf(fmaxf(lb * inv_cell * 8192.0f, 0.0f), 32767.0f);
  return (uint32_t)floorf(u) | (found ? 0u : 0x8000u);
}
__device__ __forceinline__ float margin_from_q15(uint32_t q, float cell, const MotionRef& m) {
  const float b = __fmul_rn(__fadd_rn((float)(q & 0x7FFFu) * (cell * (1.0f / 8192.0f)), m.acc), 0.9999995f);
  return (q & 0x8000u) ? -fmaxf(b, 1.17549435e-38f) : b;
}

struct T16c { float v[16]; };

struct NN {
  unsigned long long key;  // (bits(d2) << 32) | original target index : strict '<' + lowest-index tie-break
  uint32_t pos;            // position in the sorted target array, NONE_U32 if nothing within the radius
  uint32_t tie = 0;        // some squared distance that was the smallest so far has been met on a second target point (sticky: may be set
                           // for a distance that was beaten later -- tie_settle() looks again, exactly --, never missing for the final one)
};
// one candidate against the running best (the per-lane searches out of global memory: latency-bound, the two compares are free)
__device__ __forceinline__ void nn_take(NN& best, unsigned long long k, uint32_t pos) {
  best.tie |= (uint32_t)(((uint32_t)(k >> 32) == (uint32_t)(best.key >> 32)) & (k != best.key));      // same distance, another point (a clamped re-read has the same key)
  if (k < best.key) { best.key = k; best.pos = pos; }
}

// distance from q to the interval [lo,hi], shrunk by the grid margin (never over-estimates)
__device__ __forceinline__ float axis_gap(float q, float lo, float hi, float margin) {
  return fmaxf(fmaxf(lo - q, q - hi) - margin, 0.0f);
}

// Batched candidate scan: 4 independent 16-byte loads in flight per trip (indices clamped to the
// last element of the range: re-evaluating a candidate never changes the result), so a wave pays
// one memory round trip per 4 candidates instead of one per candidate.
__device__ __forceinline__ void scan_range4(const float4* __restrict__ pts, uint32_t beg, uint32_t end,
                                            float qx, float qy, float qz, NN& best) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
    const uint32_t j1 = min(j + 1, last), j2 = min(j + 2, last), j3 = min(j + 3, last);
    const float4 p0 = pts[j], p1 = pts[j1], p2 = pts[j2], p3 = pts[j3];
    const float e0 = d2_pinned(qx, qy, qz, p0.x, p0.y, p0.z), e1 = d2_pinned(qx, qy, qz, p1.x, p1.y, p1.z);
    const float e2 = d2_pinned(qx, qy, qz, p2.x, p2.y, p2.z), e3 = d2_pinned(qx, qy, qz, p3.x, p3.y, p3.z);
    const unsigned long long k0 = ((unsigned long long)__float_as_uint(e0) << 32) | __float_as_uint(p0.w);
    const unsigned long long k1 = ((unsigned long long)__float_as_uint(e1) << 32) | __float_as_uint(p1.w);
    const unsigned long long k2 = ((unsigned long long)__float_as_uint(e2) << 32) | __float_as_uint(p2.w);
    const unsigned long long k3 = ((unsigned long long)__float_as_uint(e3) << 32) | __float_as_uint(p3.w);
    nn_take(best, k0, j);
    nn_take(best, k1, j1);
    nn_take(best, k2, j2);
    nn_take(best, k3, j3);
  }
}

// The same scan over 6-D point+normal features (PointNormalFeaturesAdaptor, common_transformable_feature_adaptors.hpp:60-161):
// feature = (p, w n); squared distance exactly as nanoflann's L2_Adaptor::evalMetric forms it for DIM = 6
// (nanoflann.hpp:570-604): one group of four, result = ((d0*d0 + d1*d1) + d2*d2) + d3*d3, then the tail loop adds
// d4*d4 and d5*d5 one by one.  The target's feature normals are formed as w * n in f32, as the adaptor stores them (:90).
struct Feat6 {
  float fx, fy, fz;   // the query's (transformed) feature normal
  float w;            // normal weight
  const float4* nrm;  // sorted target normals
  // 9-D point + normal + colour features (DIM = 9: two groups of four, then one tail term): the query's w2 * colour, the
  // colour weight and the sorted target colours; att2 == nullptr: 6-D
  float gx, gy, gz;
  float w2;
  const float4* att2;
};
__device__ __forceinline__ float d6_pinned(float qx, float qy, float qz, const Feat6& f, const float4 p, const float4 n, const float4 c = make_float4(0.f, 0.f, 0.f, 0.f)) {
  const float d0 = __fsub_rn(qx, p.x), d1 = __fsub_rn(qy, p.y), d2 = __fsub_rn(qz, p.z);
  const float d3 = __fsub_rn(f.fx, __fmul_rn(f.w, n.x)), d4 = __fsub_rn(f.fy, __fmul_rn(f.w, n.y)), d5 = __fsub_rn(f.fz, __fmul_rn(f.w, n.z));
  if (f.att2 != nullptr) {
    const float d6 = __fsub_rn(f.gx, __fmul_rn(f.w2, c.x)), d7 = __fsub_rn(f.gy, __fmul_rn(f.w2, c.y)), d8 = __fsub_rn(f.gz, __fmul_rn(f.w2, c.z));
    const float g1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
    const float g2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d4, d4), __fmul_rn(d5, d5)), __fmul_rn(d6, d6)), __fmul_rn(d7, d7));
    return __fadd_rn(__fadd_rn(g1, g2), __fmul_rn(d8, d8));
  }
  float r = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
  r = __fadd_rn(r, __fmul_rn(d4, d4));
  return __fadd_rn(r, __fmul_rn(d5, d5));
}
// the query side of a feature search: the source point's feature parts under the current transform (i = index in a.src's order)
__device__ __forceinline__ void query_features(const IterArgs& a, const float* T, uint32_t i, bool with_targets, Feat6& f) {
  source_feature(a.feat, T, a.feat.src[i], f.fx, f.fy, f.fz);
  f.w = a.feat.w;
  f.nrm = with_targets ? a.feat.dst : nullptr;
  f.gx = f.gy = f.gz = 0.0f; f.w2 = a.feat.w2;
  f.att2 = a.feat.dst2;
  if (a.feat.dst2 != nullptr) {
    const float4 sc = a.feat.src2[i];
    f.gx = __fmul_rn(a.feat.w2, sc.x); f.gy = __fmul_rn(a.feat.w2, sc.y); f.gz = __fmul_rn(a.feat.w2, sc.z);
  }
}
__device__ __forceinline__ const float4* target_features(const IterArgs& a) { return a.feat.dst; }

__device__ __forceinline__ void scan_range_f6(const float4* __restrict__ pts, uint32_t beg, uint32_t end,
                                              float qx, float qy, float qz, const Feat6& f, NN& best) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 2) {
    const uint32_t j1 = min(j + 1, last);
    const float4 p0 = pts[j], p1 = pts[j1], n0 = f.nrm[j], n1 = f.nrm[j1];
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
    if (f.att2 != nullptr) { c0 = f.att2[j]; c1 = f.att2[j1]; }
    const float e0 = d6_pinned(qx, qy, qz, f, p0, n0, c0), e1 = d6_pinned(qx, qy, qz, f, p1, n1, c1);
    const unsigned long long k0 = ((unsigned long long)__float_as_uint(e0) << 32) | __float_as_uint(p0.w);
    const unsigned long long k1 = ((unsigned long long)__float_as_uint(e1) << 32) | __float_as_uint(p1.w);
    nn_take(best, k0, j);      // (the tie flag of option "tie_rule": the same 6-D / 9-D distance met on another point)
    nn_take(best, k1, j1);
  }
}

// scan_range4 for a caller that keeps the second smallest squared distance it evaluated beside the best key (m2: a median-of-three
// beside every compare; the clamped re-reads past the end of a range are not counted): the margin keys' bound on every other point,
// and the tie test of option "tie_rule" for free (m2 == the best distance).
__device__ __forceinline__ void scan_range4_m2(const float4* __restrict__ pts, uint32_t beg, uint32_t end, float qx, float qy, float qz, NN& best, float& m2) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
    uint32_t jj[4];
    float4 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { jj[k] = min(j + (uint32_t)k, last); p[k] = pts[jj[k]]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float e = d2_pinned(qx, qy, qz, p[k].x, p[k].y, p[k].z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(p[k].w);
      if (j + (uint32_t)k <= last) m2 = __builtin_amdgcn_fmed3f(__uint_as_float((uint32_t)(best.key >> 32)), m2, e);
      if (key < best.key) { best.key = key; best.pos = jj[k]; }      // (ties: m2 == the best distance at the end -- the caller looks)
    }
  }
}
// Generic exact search: expanding Chebyshev shells s = s_start, s_start+1, ... around cell (cx,cy,cz)
// (which may lie outside the grid), each shell scanned as runs of cells along x (contiguous in
// memory), with conservative box-distance pruning.  `best` carries what inner shells already found.
// Terminates when the pruning bound proves that no unscanned point can beat or tie the best.
template <bool M2 = false>
__device__ __forceinline__ void nn_search_shells(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                              int s_start, NN& best, float* m2p = nullptr) {
  float m2 = M2 ? *m2p : 0.0f;
  for (int s = s_start;; ++s) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1);
    const int y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
    const int xlo = cx - s, xhi = cx + s;
    for (int z = z0; z <= z1; ++z) {
      const bool zface = (z == cz - s) || (z == cz + s);
      const float zl = g.oz + (float)z * g.cell;
      const float gz = axis_gap(qz, zl, zl + g.cell, g.margin);
      const float gz2 = gz * gz;
      if (gz2 * KSHRINK > __uint_as_float((uint32_t)(best.key >> 32))) continue;
      for (int y = y0; y <= y1; ++y) {
        const bool face = zface || (y == cy - s) || (y == cy + s);
        const float yl = g.oy + (float)y * g.cell;
        const float gy = axis_gap(qy, yl, yl + g.cell, g.margin);
        const float gyz2 = gz2 + gy * gy;
        const float bd = __uint_as_float((uint32_t)(best.key >> 32));
        if (gyz2 * KSHRINK > bd) continue;
        const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
        if (face) {
          const int xa = max(xlo, 0), xb = min(xhi, g.nx - 1);
          if (xa <= xb) {
            const float gx = axis_gap(qx, g.ox + (float)xa * g.cell, g.ox + (float)(xb + 1) * g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= bd)
              { if (M2) scan_range4_m2(g.pts, g.cell_start[row + xa], g.cell_start[row + xb + 1], qx, qy, qz, best, m2); else scan_range4(g.pts, g.cell_start[row + xa], g.cell_start[row + xb + 1], qx, qy, qz, best); }
          }
        } else {
          if (xlo >= 0 && xlo < g.nx) {
            const float xl = g.ox + (float)xlo * g.cell;
            const float gx = axis_gap(qx, xl, xl + g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= bd)
              { if (M2) scan_range4_m2(g.pts, g.cell_start[row + xlo], g.cell_start[row + xlo + 1], qx, qy, qz, best, m2); else scan_range4(g.pts, g.cell_start[row + xlo], g.cell_start[row + xlo + 1], qx, qy, qz, best); }
          }
          if (xhi >= 0 && xhi < g.nx) {
            const float xl = g.ox + (float)xhi * g.cell;
            const float gx = axis_gap(qx, xl, xl + g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= __uint_as_float((uint32_t)(best.key >> 32)))
              { if (M2) scan_range4_m2(g.pts, g.cell_start[row + xhi], g.cell_start[row + xhi + 1], qx, qy, qz, best, m2); else scan_range4(g.pts, g.cell_start[row + xhi], g.cell_start[row + xhi + 1], qx, qy, qz, best); }
          }
        }
      }
    }
    // lower bound on the distance to anything not yet scanned (outside the (2s+1)^3 block, inside the grid)
    float b = INFINITY;
    if (cx - s > 0) b = fminf(b, qx - (g.ox + (float)(cx - s) * g.cell));
    if (cx + s + 1 < g.nx) b = fminf(b, (g.ox + (float)(cx + s + 1) * g.cell) - qx);
    if (cy - s > 0) b = fminf(b, qy - (g.oy + (float)(cy - s) * g.cell));
    if (cy + s + 1 < g.ny) b = fminf(b, (g.oy + (float)(cy + s + 1) * g.cell) - qy);
    if (cz - s > 0) b = fminf(b, qz - (g.oz + (float)(cz - s) * g.cell));
    if (cz + s + 1 < g.nz) b = fminf(b, (g.oz + (float)(cz + s + 1) * g.cell) - qz);
    if (b == INFINITY) break;  // block covers the grid: everything scanned
    b -= g.margin;
    if (b > 0.0f && __uint_as_float((uint32_t)(best.key >> 32)) < b * b * KSHRINK) break;
  }
  if (M2) *m2p = m2;
}

// The shell search once more, for a caller that wants a MARGIN with its result (the warm-started kernel's listed queries, DESIGN.md
// 6.2): it looks `extra` further than the best found so far requires -- a cell or row is skipped only when its gap exceeds
// sqrt(best) + extra, the shells end when the next one lies beyond that -- evaluates every point of the cells it does look at and
// keeps the two smallest squared distances a1 <= b2 met (carried in from the blocks the caller has already scanned completely).
// On return every target point that was not evaluated is at least sqrt(best d2, or the radius without a match) + extra away.
__device__ __forceinline__ void scan_range4_track2(const float4* __restrict__ pts, uint32_t beg, uint32_t end, float qx, float qy, float qz, NN& best,
                                                   float& a1, float& b2, float4& bp) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
    uint32_t jj[4];
    float4 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { jj[k] = min(j + (uint32_t)k, last); p[k] = pts[jj[k]]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float e = d2_pinned(qx, qy, qz, p[k].x, p[k].y, p[k].z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(p[k].w);
      if (key < best.key) { best.key = key; best.pos = jj[k]; bp = p[k]; }
      if (j + (uint32_t)k <= last) { b2 = __builtin_amdgcn_fmed3f(a1, b2, e); a1 = fminf(a1, e); }
    }
  }
}
__device__ __forceinline__ void nn_search_shells_margin(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz, int s_start, NN& best,
                                                        float& a1, float& b2, float4& bp, float extra) {
  for (int s = s_start;; ++s) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1);
    const int y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
    const int xlo = cx - s, xhi = cx + s;
    for (int z = z0; z <= z1; ++z) {
      const bool zface = (z == cz - s) || (z == cz + s);
      const float zl = g.oz + (float)z * g.cell;
      const float gz = axis_gap(qz, zl, zl + g.cell, g.margin);
      const float gz2 = gz * gz;
      for (int y = y0; y <= y1; ++y) {
        const bool face = zface || (y == cy - s) || (y == cy + s);
        const float yl = g.oy + (float)y * g.cell;
        const float gy = axis_gap(qy, yl, yl + g.cell, g.margin);
        const float gyz2 = gz2 + gy * gy;
        // (the limit follows the best found so far: rounded UP, so that what is skipped really lies beyond sqrt(best) + extra)
        float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
        float lim2 = lim * lim * 1.000001f;
        if (gyz2 * KSHRINK > lim2) continue;
        const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
        if (face) {
          const int xa = max(xlo, 0), xb = min(xhi, g.nx - 1);
          if (xa <= xb) {
            const float gx = axis_gap(qx, g.ox + (float)xa * g.cell, g.ox + (float)(xb + 1) * g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= lim2)
              scan_range4_track2(g.pts, g.cell_start[row + xa], g.cell_start[row + xb + 1], qx, qy, qz, best, a1, b2, bp);
          }
        } else {
          if (xlo >= 0 && xlo < g.nx) {
            const float xl = g.ox + (float)xlo * g.cell;
            const float gx = axis_gap(qx, xl, xl + g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= lim2)
              scan_range4_track2(g.pts, g.cell_start[row + xlo], g.cell_start[row + xlo + 1], qx, qy, qz, best, a1, b2, bp);
          }
          if (xhi >= 0 && xhi < g.nx) {
            lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
            lim2 = lim * lim * 1.000001f;
            const float xl = g.ox + (float)xhi * g.cell;
            const float gx = axis_gap(qx, xl, xl + g.cell, g.margin);
            if ((gyz2 + gx * gx) * KSHRINK <= lim2)
              scan_range4_track2(g.pts, g.cell_start[row + xhi], g.cell_start[row + xhi + 1], qx, qy, qz, best, a1, b2, bp);
          }
        }
      }
    }
    // lower bound on the distance to anything not yet scanned (outside the (2s+1)^3 block, inside the grid)
    float b = INFINITY;
    if (cx - s > 0) b = fminf(b, qx - (g.ox + (float)(cx - s) * g.cell));
    if (cx + s + 1 < g.nx) b = fminf(b, (g.ox + (float)(cx + s + 1) * g.cell) - qx);
    if (cy - s > 0) b = fminf(b, qy - (g.oy + (float)(cy - s) * g.cell));
    if (cy + s + 1 < g.ny) b = fminf(b, (g.oy + (float)(cy + s + 1) * g.cell) - qy);
    if (cz - s > 0) b = fminf(b, qz - (g.oz + (float)(cz - s) * g.cell));
    if (cz + s + 1 < g.nz) b = fminf(b, (g.oz + (float)(cz + s + 1) * g.cell) - qz);
    if (b == INFINITY) break;  // block covers the grid: everything scanned
    b -= g.margin;
    const float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
    if (b > 0.0f && lim * lim * 1.000001f < b * b * KSHRINK) break;
  }
}

constexpr int ITER_THREADS = 256;
constexpr int ITER_WAVES = ITER_THREADS / 64;
constexpr int LIST_CAP = 10;  // 8 neighbour rows + the two x-neighbours of the own cell

// Exact 1-NN in radius (== brute-force argmin of the pinned f32 d2, lowest-index tie-break, d2 < max_sq
// strict as nanoflann.hpp:1901 / kd_tree_utilities.hpp:29).
//
// Fast path (query inside the grid): scan the own cell; from the best so far decide, per neighbour
// row of the 3x3x3 block, which run of cells along x can still hold a nearer point; fetch all run
// boundaries with independent loads (one round trip); push the non-empty runs on a per-lane LDS
// work list and scan them in ONE loop (dense trips: lanes do not wait on each other's culled rows).
// If the 3x3x3 block does not prove exactness (sparse data / large radius) or the query lies outside
// the grid, continue with the generic shell search.
// lst: this lane's column of the LDS work list, entries at lst[k * ITER_THREADS].
// nn_search_from(): `best` comes in initialised -- (radius, none), or a point KNOWN to lie within the radius (a warm start:
// the search then only looks where something nearer, or as near with a lower index, can be; the result is the same).
__device__ __forceinline__ void nn_search_from(const GridDev& g, float qx, float qy, float qz, float max_sq, NN& best,
                                               uint2* lst) {
  const float BIG = 1.0e9f;
  const float fx = fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG);
  const float fy = fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG);
  const float fz = fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG);
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  const bool inside = (cx >= 0) & (cx < g.nx) & (cy >= 0) & (cy < g.ny) & (cz >= 0) & (cz < g.nz);
  int s_shells = 2;      // (ONE call site of the shell search for both ways into it: it is inlined, and the kernels that hold this search live on their registers)
  if (!inside) {
    // query farther than the radius from the whole grid: nothing to find
    const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin);
    const float gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin);
    const float gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
    if ((gx * gx + gy * gy + gz * gz) * KSHRINK >= max_sq) return;
    s_shells = max(0, max(max(-cx, cx - (g.nx - 1)), max(max(-cy, cy - (g.ny - 1)), max(-cz, cz - (g.nz - 1)))));
  } else {
  const uint32_t cid = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
  const uint32_t b0 = g.cell_start[cid], e0 = g.cell_start[cid + 1];
  scan_range4(g.pts, b0, e0, qx, qy, qz, best);
  const float bd = __uint_as_float((uint32_t)(best.key >> 32));

  // shrunk distances from q to the six faces of its own cell
  const float xl = g.ox + (float)cx * g.cell, yl = g.oy + (float)cy * g.cell, zl = g.oz + (float)cz * g.cell;
  const float gmx = fmaxf(qx - xl - g.margin, 0.0f), gpx = fmaxf(xl + g.cell - qx - g.margin, 0.0f);
  const float gmy = fmaxf(qy - yl - g.margin, 0.0f), gpy = fmaxf(yl + g.cell - qy - g.margin, 0.0f);
  const float gmz = fmaxf(qz - zl - g.margin, 0.0f), gpz = fmaxf(zl + g.cell - qz - g.margin, 0.0f);
  const bool hmx = cx > 0, hpx = cx + 1 < g.nx, hmy = cy > 0, hpy = cy + 1 < g.ny, hmz = cz > 0, hpz = cz + 1 < g.nz;
  {  // nothing outside the own cell can beat or tie the best: done (no neighbour = no constraint)
    float b = INFINITY;
    if (hmx) b = fminf(b, gmx);
    if (hpx) b = fminf(b, gpx);
    if (hmy) b = fminf(b, gmy);
    if (hpy) b = fminf(b, gpy);
    if (hmz) b = fminf(b, gmz);
    if (hpz) b = fminf(b, gpz);
    if (b == INFINITY || bd < b * b * KSHRINK) return;
  }
  const float ax2[3] = {gmx * gmx, 0.0f, gpx * gpx};
  const float ay2[3] = {gmy * gmy, 0.0f, gpy * gpy};
  const float az2[3] = {gmz * gmz, 0.0f, gpz * gpz};
  const bool okx[3] = {hmx, true, hpx}, oky[3] = {hmy, true, hpy}, okz[3] = {hmz, true, hpz};

  // run boundaries of the 9 rows: unconditional independent loads (index 0 when the row is culled)
  uint32_t ia[9], ib[9];
  bool pass[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int dz = r / 3, dy = r % 3;  // 0,1,2 <-> -1,0,+1
    const float gyz2 = az2[dz] + ay2[dy];
    const bool p = okz[dz] && oky[dy] && (gyz2 * KSHRINK <= bd);
    const bool left = p && okx[0] && ((gyz2 + ax2[0]) * KSHRINK <= bd);
    const bool right = p && okx[2] && ((gyz2 + ax2[2]) * KSHRINK <= bd);
    const uint32_t row = cid + (uint32_t)((dz - 1) * g.ny * g.nx + (dy - 1) * g.nx);  // wraps harmlessly when !p
    pass[r] = (r == 4) ? (left || right) : p;
    ia[r] = pass[r] ? (row - (left ? 1u : 0u)) : 0u;
    ib[r] = pass[r] ? (row + 1u + (right ? 1u : 0u)) : 0u;
  }
  uint32_t va[9], vb[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) { va[r] = g.cell_start[ia[r]]; vb[r] = g.cell_start[ib[r]]; }
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    if (r == 4) {  // own row: the own cell [b0,e0) is already scanned -> up to two side runs
      if (pass[r] && b0 > va[r]) { lst[cnt * ITER_THREADS] = make_uint2(va[r], b0); ++cnt; }
      if (pass[r] && vb[r] > e0) { lst[cnt * ITER_THREADS] = make_uint2(e0, vb[r]); ++cnt; }
    } else {
      if (pass[r] && vb[r] > va[r]) { lst[cnt * ITER_THREADS] = make_uint2(va[r], vb[r]); ++cnt; }
    }
  }
  {
    // ONE flattened loop over this lane's work list: every trip each lane evaluates its next
    // CILHIP_CAND candidates, popping the next range when the current one is exhausted, so the
    // wave runs max_lane(total trips) instead of sum_k max_lane(trips of range k).
    int k = 0;
    uint32_t j = 0, e = 0;
    for (;;) {
      if (j >= e) {
        if (k >= cnt) break;
        const uint2 r = lst[k * ITER_THREADS];
        ++k;
        j = r.x; e = r.y;
      }
      const uint32_t last = e - 1;
      const float4 p0 = g.pts[j];
      const uint32_t j1 = min(j + 1, last);
      const float4 p1 = g.pts[j1];
#if CILHIP_CAND == 4
      const uint32_t j2 = min(j + 2, last), j3 = min(j + 3, last);
      const float4 p2 = g.pts[j2], p3 = g.pts[j3];
#endif
      const float e0 = d2_pinned(qx, qy, qz, p0.x, p0.y, p0.z), e1 = d2_pinned(qx, qy, qz, p1.x, p1.y, p1.z);
      const unsigned long long k0 = ((unsigned long long)__float_as_uint(e0) << 32) | __float_as_uint(p0.w);
      const unsigned long long k1 = ((unsigned long long)__float_as_uint(e1) << 32) | __float_as_uint(p1.w);
      nn_take(best, k0, j);
      nn_take(best, k1, j1);
#if CILHIP_CAND == 4
      const float e2 = d2_pinned(qx, qy, qz, p2.x, p2.y, p2.z), e3 = d2_pinned(qx, qy, qz, p3.x, p3.y, p3.z);
      const unsigned long long k2 = ((unsigned long long)__float_as_uint(e2) << 32) | __float_as_uint(p2.w);
      const unsigned long long k3 = ((unsigned long long)__float_as_uint(e3) << 32) | __float_as_uint(p3.w);
      nn_take(best, k2, j2);
      nn_take(best, k3, j3);
#endif
      j += CILHIP_CAND;
    }
  }
  {  // does the 3x3x3 block prove exactness?  faces of the block that still have cells beyond them
    float b = INFINITY;
    if (cx - 1 > 0) b = fminf(b, gmx + g.cell);
    if (cx + 2 < g.nx) b = fminf(b, gpx + g.cell);
    if (cy - 1 > 0) b = fminf(b, gmy + g.cell);
    if (cy + 2 < g.ny) b = fminf(b, gpy + g.cell);
    if (cz - 1 > 0) b = fminf(b, gmz + g.cell);
    if (cz + 2 < g.nz) b = fminf(b, gpz + g.cell);
    if (b == INFINITY) return;
    b -= g.margin;
    if (b > 0.0f && __uint_as_float((uint32_t)(best.key >> 32)) < b * b * KSHRINK) return;
  }
  }
  nn_search_shells(g, qx, qy, qz, cx, cy, cz, s_shells, best);
}
// The same search for a caller that wants the MARGIN with the result (the cold per-lane iterations of a run whose later
// iterations may be warm-started): besides the best key it keeps the second smallest squared distance it evaluated
// (m2: a median-of-three beside every compare; the radius stands in for "best" while nothing has been found, as in the tiles'
// octant search) and the smallest squared gap of anything it SKIPPED (cull2: rows and side cells of the 3x3x3 block culled against
// the best so far).  *lb_out = a lower bound on the distance from q to every target point but the match (to every target point
// without one): min(sqrt(m2), sqrt(cull2), gap to the faces of the block that proved the result); 0 when the result came from the
// shell search (no bound kept).
__device__ __forceinline__ void nn_search_lb(const GridDev& g, float qx, float qy, float qz, float max_sq, NN& best, uint2* lst, float* lb_out) {
  best.key = ((unsigned long long)__float_as_uint(max_sq) << 32);
  best.pos = NONE_U32;
  best.tie = 0;
  *lb_out = 0.0f;
  const float BIG = 1.0e9f;
  const float fx = fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG);
  const float fy = fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG);
  const float fz = fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG);
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  const bool inside = (cx >= 0) & (cx < g.nx) & (cy >= 0) & (cy < g.ny) & (cz >= 0) & (cz < g.nz);
  if (!inside) {
    const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin);
    const float gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin);
    const float gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
    const float gg = (gx * gx + gy * gy + gz * gz) * KSHRINK;
    if (gg >= max_sq) { *lb_out = __fsqrt_rn(gg) * 0.999999f; return; }      // every target point lies inside the grid
    const int s0 = max(0, max(max(-cx, cx - (g.nx - 1)), max(max(-cy, cy - (g.ny - 1)), max(-cz, cz - (g.nz - 1)))));
    // (queries outside the grid: the plain shell search -- no second-smallest tracking here; a tie among them is noticed by the keys)
    nn_search_shells(g, qx, qy, qz, cx, cy, cz, s0, best);
    return;
  }
  float m2 = INFINITY;
  const uint32_t cid = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
  const uint32_t b0 = g.cell_start[cid], e0 = g.cell_start[cid + 1];
  scan_range4_m2(g.pts, b0, e0, qx, qy, qz, best, m2);
  const float bd = __uint_as_float((uint32_t)(best.key >> 32));
  const float xl = g.ox + (float)cx * g.cell, yl = g.oy + (float)cy * g.cell, zl = g.oz + (float)cz * g.cell;
  const float gmx = fmaxf(qx - xl - g.margin, 0.0f), gpx = fmaxf(xl + g.cell - qx - g.margin, 0.0f);
  const float gmy = fmaxf(qy - yl - g.margin, 0.0f), gpy = fmaxf(yl + g.cell - qy - g.margin, 0.0f);
  const float gmz = fmaxf(qz - zl - g.margin, 0.0f), gpz = fmaxf(zl + g.cell - qz - g.margin, 0.0f);
  const bool hmx = cx > 0, hpx = cx + 1 < g.nx, hmy = cy > 0, hpy = cy + 1 < g.ny, hmz = cz > 0, hpz = cz + 1 < g.nz;
  {  // nothing outside the own cell can beat or tie the best: done
    float b = INFINITY;
    if (hmx) b = fminf(b, gmx);
    if (hpx) b = fminf(b, gpx);
    if (hmy) b = fminf(b, gmy);
    if (hpy) b = fminf(b, gpy);
    if (hmz) b = fminf(b, gmz);
    if (hpz) b = fminf(b, gpz);
    if (b == INFINITY || bd < b * b * KSHRINK) { *lb_out = fminf(__fsqrt_rn(m2) * 0.999999f, b); best.tie = (best.pos != NONE_U32 && m2 == bd) ? 1u : 0u; return; }
  }
  const float ax2[3] = {gmx * gmx, 0.0f, gpx * gpx};
  const float ay2[3] = {gmy * gmy, 0.0f, gpy * gpy};
  const float az2[3] = {gmz * gmz, 0.0f, gpz * gpz};
  const bool okx[3] = {hmx, true, hpx}, oky[3] = {hmy, true, hpy}, okz[3] = {hmz, true, hpz};
  float cull2 = INFINITY;      // smallest squared gap of a row / side cell that exists and was skipped
  uint32_t ia[9], ib[9];
  bool pass[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int dz = r / 3, dy = r % 3;
    const float gyz2 = az2[dz] + ay2[dy];
    const bool ex = okz[dz] && oky[dy];
    const bool p = ex && (gyz2 * KSHRINK <= bd);
    const bool left = p && okx[0] && ((gyz2 + ax2[0]) * KSHRINK <= bd);
    const bool right = p && okx[2] && ((gyz2 + ax2[2]) * KSHRINK <= bd);
    if (ex && !p) cull2 = fminf(cull2, gyz2);
    if (p && okx[0] && !left) cull2 = fminf(cull2, gyz2 + ax2[0]);
    if (p && okx[2] && !right) cull2 = fminf(cull2, gyz2 + ax2[2]);
    const uint32_t row = cid + (uint32_t)((dz - 1) * g.ny * g.nx + (dy - 1) * g.nx);
    pass[r] = (r == 4) ? (left || right) : p;
    ia[r] = pass[r] ? (row - (left ? 1u : 0u)) : 0u;
    ib[r] = pass[r] ? (row + 1u + (right ? 1u : 0u)) : 0u;
  }
  uint32_t va[9], vb[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) { va[r] = g.cell_start[ia[r]]; vb[r] = g.cell_start[ib[r]]; }
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    if (r == 4) {
      if (pass[r] && b0 > va[r]) { lst[cnt * ITER_THREADS] = make_uint2(va[r], b0); ++cnt; }
      if (pass[r] && vb[r] > e0) { lst[cnt * ITER_THREADS] = make_uint2(e0, vb[r]); ++cnt; }
    } else {
      if (pass[r] && vb[r] > va[r]) { lst[cnt * ITER_THREADS] = make_uint2(va[r], vb[r]); ++cnt; }
    }
  }
  for (int k = 0; k < cnt; ++k) {
    const uint2 r = lst[k * ITER_THREADS];
    scan_range4_m2(g.pts, r.x, r.y, qx, qy, qz, best, m2);
  }
  {  // does the 3x3x3 block prove exactness?
    float b = INFINITY;
    if (cx - 1 > 0) b = fminf(b, gmx + g.cell);
    if (cx + 2 < g.nx) b = fminf(b, gpx + g.cell);
    if (cy - 1 > 0) b = fminf(b, gmy + g.cell);
    if (cy + 2 < g.ny) b = fminf(b, gpy + g.cell);
    if (cz - 1 > 0) b = fminf(b, gmz + g.cell);
    if (cz + 2 < g.nz) b = fminf(b, gpz + g.cell);
    if (b != INFINITY) b -= g.margin;
    if (b == INFINITY || (b > 0.0f && __uint_as_float((uint32_t)(best.key >> 32)) < b * b * KSHRINK)) {
      *lb_out = fminf(fminf(__fsqrt_rn(m2) * 0.999999f, __fsqrt_rn(cull2)), b);
      best.tie = (best.pos != NONE_U32 && m2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
      return;
    }
  }
  nn_search_shells<true>(g, qx, qy, qz, cx, cy, cz, 2, best, &m2);
  best.tie = (best.pos != NONE_U32 && m2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
}
__device__ __forceinline__ void nn_search(const GridDev& g, float qx, float qy, float qz, float max_sq, NN& best,
                                          uint2* lst) {
  best.key = ((unsigned long long)__float_as_uint(max_sq) << 32);
  best.pos = NONE_U32;
  best.tie = 0;
  nn_search_from(g, qx, qy, qz, max_sq, best, lst);
}

// The same exact search by a GROUP of G adjacent lanes for ONE query (the clean-up pass of the tiled search: few queries,
// each with a large block of cells to look at -- one lane per query leaves the chip idle behind long dependent chains).
// The rows of the (2s+1)^2 x (2s+1) block around the query's cell are dealt round-robin to the lanes, each row one run of
// the sorted target array; the group then takes the minimum key (keys are unique: they carry the target index).  If the
// block does not prove the result, s grows straight to the size the best found so far needs.  All control flow is
// uniform within a group.  `sub` = lane index inside the group; every lane of the group returns the same result.
// FEAT6: candidates are compared by the 6-D feature distance (the proof still uses the 3-D geometry: d6 >= d3).
// INIT: `best` comes in holding a target point KNOWN to lie within the radius (the previous iteration's match under the current
// transform, the same on every lane of the group): the first block is the one that point's distance needs, and rows beyond that
// distance are never opened -- the search looks only where something nearer, or as near, can be; the result is the same.
template <int G, bool FEAT6 = false, bool INIT = false>
__device__ __forceinline__ void nn_search_group(const GridDev& g, float qx, float qy, float qz, float max_sq, int sub, int s_start, NN& best,
                                                const Feat6* f6 = nullptr) {
  if (!INIT) {
    best.key = ((unsigned long long)__float_as_uint(max_sq) << 32);
    best.pos = NONE_U32;
  }
  best.tie = 0;
  const float BIG = 1.0e9f;
  const float fx = fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG);
  const float fy = fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG);
  const float fz = fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG);
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  {  // query farther than the radius from the whole grid: nothing to find
    const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin);
    const float gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin);
    const float gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
    if ((gx * gx + gy * gy + gz * gz) * KSHRINK >= max_sq) return;
  }
  // first block size that reaches the grid at all
  int s = max(s_start, max(max(-cx, cx - (g.nx - 1)), max(max(-cy, cy - (g.ny - 1)), max(-cz, cz - (g.nz - 1)))));
  if (INIT && best.pos != NONE_U32)      // ... and that holds the ball of the known point's distance whatever the offset of q in its cell
    s = max(s, (int)fminf(sqrtf(__uint_as_float((uint32_t)(best.key >> 32))) * g.inv_cell + 1.0f, (float)(g.nx + g.ny + g.nz)));
  for (;;) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1);
    const int y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
    const int xa = max(cx - s, 0), xb = min(cx + s, g.nx - 1);
    if (xa <= xb && y0 <= y1 && z0 <= z1) {
      const float gx = axis_gap(qx, g.ox + (float)xa * g.cell, g.ox + (float)(xb + 1) * g.cell, g.margin);
      const float gx2 = gx * gx;
      const int wy = y1 - y0 + 1, nrows = wy * (z1 - z0 + 1);
      int z = z0, y = y0 + sub;                    // row `sub` of the block, then every G-th
      while (y > y1) { y -= wy; ++z; }
      for (int k = sub; k < nrows; k += G) {
        const float zl = g.oz + (float)z * g.cell, yl = g.oy + (float)y * g.cell;
        const float gz = axis_gap(qz, zl, zl + g.cell, g.margin), gy = axis_gap(qy, yl, yl + g.cell, g.margin);
        const float gyz2 = gz * gz + gy * gy, bd0 = __uint_as_float((uint32_t)(best.key >> 32));
        if ((gyz2 + gx2) * KSHRINK <= bd0) {
          const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          // the row clipped to the cells the ball of the best distance so far can reach along x (a surface's rows hold many points the
          // ball does not come near): reach = sqrt(bd / KSHRINK - gyz2), rounded UP, plus the grid margin -- a superset of the cells
          // whose gap admits a candidate, so nothing that could win or tie is skipped
          int xa_r = xa, xb_r = xb;
          const float w2 = bd0 * (1.0f / KSHRINK) * 1.000001f - gyz2;
          if (w2 < 1.0e30f) {
            const float w = sqrtf(fmaxf(w2, 0.0f)) * 1.000001f + 2.0f * g.margin;
            xa_r = max(xa, (int)floorf(fminf(fmaxf((qx - w - g.ox) * g.inv_cell, -BIG), BIG)) - 0);
            xb_r = min(xb, (int)floorf(fminf(fmaxf((qx + w - g.ox) * g.inv_cell, -BIG), BIG)) + 0);
          }
          if (xa_r <= xb_r) {
            if (FEAT6) scan_range_f6(g.pts, g.cell_start[row + xa_r], g.cell_start[row + xb_r + 1], qx, qy, qz, *f6, best);
            else scan_range4(g.pts, g.cell_start[row + xa_r], g.cell_start[row + xb_r + 1], qx, qy, qz, best);
          }
        }
        y += G;
        while (y > y1) { y -= wy; ++z; }
      }
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
      const unsigned long long ok = __shfl_xor(best.key, off, 64);
      const uint32_t op = __shfl_xor(best.pos, off, 64);
      best.tie |= __shfl_xor(best.tie, off, 64);      // (what a lane noticed in its rows; and the same distance on two lanes' points:)
      nn_take(best, ok, op);
    }
    // lower bound on the distance to anything outside the block (and inside the grid)
    float b = INFINITY;
    if (cx - s > 0) b = fminf(b, qx - (g.ox + (float)(cx - s) * g.cell));
    if (cx + s + 1 < g.nx) b = fminf(b, (g.ox + (float)(cx + s + 1) * g.cell) - qx);
    if (cy - s > 0) b = fminf(b, qy - (g.oy + (float)(cy - s) * g.cell));
    if (cy + s + 1 < g.ny) b = fminf(b, (g.oy + (float)(cy + s + 1) * g.cell) - qy);
    if (cz - s > 0) b = fminf(b, qz - (g.oz + (float)(cz - s) * g.cell));
    if (cz + s + 1 < g.nz) b = fminf(b, (g.oz + (float)(cz + s + 1) * g.cell) - qz);
    if (b == INFINITY) break;  // block covers the grid: everything scanned
    b -= g.margin;
    const float bd = __uint_as_float((uint32_t)(best.key >> 32));   // the radius while nothing is found
    if (b > 0.0f && bd < b * b * KSHRINK) break;
    // the block size that proves a result at distance sqrt(bd) whatever the offset of q in its cell (at least one more)
    // (nothing found yet: grow geometrically -- the radius may be infinite)
    const float need = sqrtf(bd) * g.inv_cell + 1.0f;
    s = (best.pos == NONE_U32) ? s + max(1, s >> 1) : max(s + 1, (int)fminf(need, (float)(g.nx + g.ny + g.nz)));
  }
}

// (tie_before(): internal.hpp -- shared with the reverse searches of bidir.hip)
// A query whose search noticed a tie (NN::tie, or second smallest distance == smallest): pos / bd = its match by the lowest-index rule
// and that match's squared distance.  Returns the sorted position of the match the option asks for: every target point at EXACTLY
// bd is enumerated -- the closed ball of that radius, shells of cells around the query's, rows beyond the distance skipped, ends when
// the next shell lies strictly beyond it (a cell at exactly the distance is looked at) -- and the first-met one kept as they stream by
// (the traversal order of one query is a total order: pairwise comparisons suffice, any number of candidates).  Without tables the
// query is counted for the host and keeps its match.
// FEAT6: the same over 6-D / 9-D feature distances (every point at feature distance bd lies inside the 3-D ball of that radius: d6 >= d3), ordered by
// the feature tree's tables (tie_before_nd; the query's feature vector = its point and transformed feature parts).
__device__ __forceinline__ bool tie_before_nd(const TieDev& tt, const float* qf, uint32_t pa, uint32_t pb) {
  const uint2 la = tt.leaf_slot[pa], lb = tt.leaf_slot[pb];
  if (la.x == lb.x) return la.y < lb.y;
  uint32_t na = la.x, nb = lb.x;
  uint4 A = tt.nodes[na], B = tt.nodes[nb];
  uint32_t a_second = 0;      // (TieNode::info of a feature tree: (depth << 5) | (split dimension << 1) | second child)
  while ((A.y >> 5) > (B.y >> 5)) { a_second = A.y & 1u; na = A.x; A = tt.nodes[na]; }
  while ((B.y >> 5) > (A.y >> 5)) { nb = B.x; B = tt.nodes[nb]; }
  while (na != nb) { a_second = A.y & 1u; na = A.x; A = tt.nodes[na]; nb = B.x; B = tt.nodes[nb]; }
  const uint32_t feat = (A.y >> 1) & 15u;
  float val = qf[0];
#pragma unroll
  for (uint32_t d = 1; d < 9; ++d) val = feat == d ? qf[d] : val;
  const float diff1 = __fsub_rn(val, __uint_as_float(A.z)), diff2 = __fsub_rn(val, __uint_as_float(A.w));
  const uint32_t first_is_second = __fadd_rn(diff1, diff2) < 0.0f ? 0u : 1u;
  return a_second == first_is_second;
}
template <bool FEAT6 = false>
__device__ __forceinline__ uint32_t tie_settle(const GridDev& g, const TieDev& tt, float qx, float qy, float qz, uint32_t pos, float bd, const Feat6* f6 = nullptr) {
  if (tt.leaf_slot == nullptr) { atomicAdd(tt.counters, 1u); return pos; }
  float qf[9] = {qx, qy, qz, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (FEAT6) { qf[3] = f6->fx; qf[4] = f6->fy; qf[5] = f6->fz; qf[6] = f6->gx; qf[7] = f6->gy; qf[8] = f6->gz; }
  const float BIG = 1.0e9f;
  const int cx = (int)floorf(fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG)), cy = (int)floorf(fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG)),
            cz = (int)floorf(fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG));
  uint32_t cur = pos, ncand = 0;
  for (int s = max(0, max(max(-cx, cx - (g.nx - 1)), max(max(-cy, cy - (g.ny - 1)), max(-cz, cz - (g.nz - 1)))));; ++s) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1), y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
    for (int z = z0; z <= z1; ++z) {
      const float zl = g.oz + (float)z * g.cell;
      const float az = axis_gap(qz, zl, zl + g.cell, g.margin);
      for (int y = y0; y <= y1; ++y) {
        const bool face = (z == cz - s) || (z == cz + s) || (y == cy - s) || (y == cy + s);
        const float yl = g.oy + (float)y * g.cell;
        const float ay = axis_gap(qy, yl, yl + g.cell, g.margin);
        if ((az * az + ay * ay) * KSHRINK > bd) continue;
        const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
        uint32_t rb[2] = {0, 0}, re[2] = {0, 0};
        if (face) {
          const int xa = max(cx - s, 0), xb = min(cx + s, g.nx - 1);
          if (xa <= xb) { rb[0] = g.cell_start[row + xa]; re[0] = g.cell_start[row + xb + 1]; }
        } else {
          if (cx - s >= 0 && cx - s < g.nx) { rb[0] = g.cell_start[row + cx - s]; re[0] = g.cell_start[row + cx - s + 1]; }
          if (s > 0 && cx + s >= 0 && cx + s < g.nx) { rb[1] = g.cell_start[row + cx + s]; re[1] = g.cell_start[row + cx + s + 1]; }
        }
        for (int r = 0; r < 2; ++r)
          for (uint32_t j = rb[r]; j < re[r]; ++j) {
            const float4 p = g.pts[j];
            float e;
            if (FEAT6) e = d6_pinned(qx, qy, qz, *f6, p, f6->nrm[j], f6->att2 != nullptr ? f6->att2[j] : make_float4(0.f, 0.f, 0.f, 0.f));
            else e = d2_pinned(qx, qy, qz, p.x, p.y, p.z);
            if (e == bd) {
              ++ncand;
              if (j != cur && (FEAT6 ? tie_before_nd(tt, qf, j, cur) : tie_before(tt, qx, qy, qz, j, cur))) cur = j;
            }
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
    if (b > 0.0f && bd < b * b * KSHRINK) break;
  }
  if (ncand >= 2u) { atomicAdd(tt.counters + 1, 1u); if (cur != pos) atomicAdd(tt.counters + 2, 1u); }      // (a flag raised for a distance that was beaten later: one candidate)
  return cur;
}

// ---- accumulation helpers ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// exp() for the RBF weight evaluator in PINNED f32 arithmetic (one fixed sequence of correctly rounded operations, the same
// in oracle/icp_oracle.c): round-to-nearest argument reduction by ln 2 (two-part constant), degree-6 polynomial, exact
// scaling.  Within 1 ulp of the correctly rounded value on [-80, 0]; the reference calls std::exp(float), whose last
// bit depends on its libm.  Arguments below -80 give 0 (the true value is < 2e-35).
__device__ __forceinline__ float pinned_expf(float x) {
  if (!(x >= -80.0f)) return x != x ? x : 0.0f;
  if (x > 80.0f) x = 80.0f;
  const float n = rintf(__fmul_rn(x, 1.44269504f));
  float r = __fmaf_rn(n, -0.693359375f, x);
  r = __fmaf_rn(n, 2.12194440e-4f, r);
  float q = 1.9875691500e-4f;
  q = __fmaf_rn(q, r, 1.3981999507e-3f);
  q = __fmaf_rn(q, r, 8.3334519073e-3f);
  q = __fmaf_rn(q, r, 4.1665795894e-2f);
  q = __fmaf_rn(q, r, 1.6666665459e-1f);
  q = __fmaf_rn(q, r, 5.0000001201e-1f);
  q = __fmaf_rn(q, __fmul_rn(r, r), r);
  q = __fadd_rn(q, 1.0f);
  return ldexpf(q, (int)n);
}
__device__ __forceinline__ float corr_weight(int kind, float coeff, float value) {
  return kind == CW_UNITY ? 1.0f : kind == CW_IDENTITY ? value : pinned_expf(__fmul_rn(coeff, value));
}
// per-pair weights (point term, plane term) of a correspondence with search distance `value`
__device__ __forceinline__ void pair_weights(const CorrWeights& cw, float value, float& wq, float& wp) {
  wq = wp = 1.0f;
  if (cw.enabled) {
    wq = __fmul_rn(cw.w_p2p, corr_weight(cw.point_kind, cw.point_coeff, value));
    wp = __fmul_rn(cw.w_p2pl, corr_weight(cw.plane_kind, cw.plane_coeff, value));
  }
}

template <int METRIC>
struct AccTraits {
  static constexpr bool plane = (METRIC == IM_PLANE || METRIC == IM_BOTH);
  static constexpr bool point = (METRIC == IM_POINT || METRIC == IM_BOTH);
  static constexpr bool kabsch = (METRIC == IM_KABSCH);
  static constexpr bool affine = (METRIC == IM_AFF0 || METRIC == IM_AFF1 || METRIC == IM_AFF2);
  static constexpr int NA = affine ? (METRIC == IM_AFF0 ? 35 : 30) : kabsch ? 16 : (plane ? 28 : 1);  // slots [0, NA)
  static constexpr int NB = point ? 16 : 0;                  // slots [28, 28+NB)
};

// The accumulation of one matched pair (q = T*s already formed): what every accumulating kernel adds per correspondence.
// accA / accB are the caller's per-lane f64 accumulators (slots [0, NA) and [28, 28 + NB) of a partial-sum row).
template <int METRIC>
__device__ __forceinline__ void accumulate_pair(double* __restrict__ accA, double* __restrict__ accB, const float* T, const float* iL, const float* it,
                                                const float* smt, const float* dmean, const bool sym, const bool has_nrm, float qx, float qy, float qz,
                                                uint32_t pos, const float4 p, const float4 nvp, const float4 snp, const float wq = 1.0f,
                                                const float wp = 1.0f) {
  // wq / wp: the per-pair weights of the point and plane terms (pair_weights(); 1 = unity evaluators, where the metric
  // weights are applied to the sums by the solver instead).  Only the rigid combined-metric forms take them.
  using TR = AccTraits<METRIC>;
  if (METRIC != IM_NONE && pos != NONE_U32) {
    if (TR::kabsch) {
      // raw moments for the closed-form estimator (transform_estimation.hpp:25-34)
      const double pd[3] = {(double)p.x, (double)p.y, (double)p.z};
      const double qd[3] = {(double)qx, (double)qy, (double)qz};
      accA[0] += 1.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) { accA[1 + c] += pd[c]; accA[4 + c] += qd[c]; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) accA[7 + r * 3 + c] = fma(pd[r], qd[c], accA[7 + r * 3 + c]);
    } else if (TR::affine) {
      // Affine closed form (transform_estimation.hpp:369-476; :50-102 for the point-to-point class): per-term
      // quantities in f32 as the reference forms them -- s = q - src_mean', d = p - dst_mean -- their products and
      // sums in f64.  eq_vec = (n_0 s, n_1 s, n_2 s, n): every entry of eq_vec eq_vec^T is n_j n_k (s,1)_a (s,1)_b.
      const float d0 = __fsub_rn(p.x, dmean[0]), d1 = __fsub_rn(p.y, dmean[1]), d2 = __fsub_rn(p.z, dmean[2]);
      const float s0 = __fsub_rn(qx, smt[0]), s1 = __fsub_rn(qy, smt[1]), s2 = __fsub_rn(qz, smt[2]);
      const double sd[4] = {(double)s0, (double)s1, (double)s2, 1.0};
      // wq / wp: per-pair weights of the point and plane terms (weight evaluators of the affine combined-metric class, :432-434,
      // :453-455; 1 = unity, where the metric weights are applied to the sums by the solver).  Every sum is linear in its weight.
      const double wqd = (double)wq, wpd = (double)wp;
      if (METRIC == IM_AFF0) {
        const double dd[3] = {(double)d0, (double)d1, (double)d2};
        accA[0] += 1.0;
        accA[34] += wqd;                 // sum of the point weights: the translation block of the point terms
        int k = 1;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = r; c < 3; ++c) { accA[k] = fma(wqd * sd[r], sd[c], accA[k]); ++k; }
#pragma unroll
        for (int c = 0; c < 3; ++c) accA[7 + c] = fma(wqd, sd[c], accA[7 + c]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) accA[10 + r * 3 + c] = fma(wqd * sd[r], dd[c], accA[10 + r * 3 + c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) accA[19 + c] = fma(wqd, dd[c], accA[19 + c]);
        if (has_nrm) {
          // n.dot(dst - dst_mean)  (:464), f32 like the reference's dot product
          const float res = __fadd_rn(__fadd_rn(__fmul_rn(nvp.x, d0), __fmul_rn(nvp.y, d1)), __fmul_rn(nvp.z, d2));
          const double nd[3] = {(double)nvp.x, (double)nvp.y, (double)nvp.z};
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const double rn = wpd * (double)res * nd[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) accA[22 + j * 4 + c] = fma(rn, sd[c], accA[22 + j * 4 + c]);
          }
        }
      } else {
        const double nd[3] = {(double)nvp.x, (double)nvp.y, (double)nvp.z};
        int k = 0;
#pragma unroll
        for (int jk = 0; jk < 3; ++jk) {
          // (j,k): AFF1 -> (0,0),(0,1),(0,2); AFF2 -> (1,1),(1,2),(2,2)
          const int j = (METRIC == IM_AFF1) ? 0 : (jk == 2 ? 2 : 1);
          const int kk = (METRIC == IM_AFF1) ? jk : (jk == 0 ? 1 : 2);
          const double nn = wpd * nd[j] * nd[kk];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = r; c < 4; ++c) { accA[k] = fma(nn * sd[r], sd[c], accA[k]); ++k; }
        }
      }
    } else {
      // per-term quantities in f32 exactly as the reference forms them (transform_estimation.hpp:302-304,:333-335)
      const float d0 = __fsub_rn(p.x, dmean[0]), d1 = __fsub_rn(p.y, dmean[1]), d2 = __fsub_rn(p.z, dmean[2]);
      const float u0 = __fsub_rn(qx, smt[0]), u1 = __fsub_rn(qy, smt[1]), u2 = __fsub_rn(qz, smt[2]);
      // s = inner_tform * (q - T*src_mean); identity on the first Gauss-Newton step
      const float s0 = __fadd_rn(__fadd_rn(__fmul_rn(iL[0], u0), __fadd_rn(__fmul_rn(iL[1], u1), __fmul_rn(iL[2], u2))), it[0]);
      const float s1 = __fadd_rn(__fadd_rn(__fmul_rn(iL[3], u0), __fadd_rn(__fmul_rn(iL[4], u1), __fmul_rn(iL[5], u2))), it[1]);
      const float s2 = __fadd_rn(__fadd_rn(__fmul_rn(iL[6], u0), __fadd_rn(__fmul_rn(iL[7], u1), __fmul_rn(iL[8], u2))), it[2]);
      const float a0 = __fadd_rn(d0, s0), a1 = __fadd_rn(d1, s1), a2 = __fadd_rn(d2, s2);
      const float r0 = __fsub_rn(d0, s0), r1 = __fsub_rn(d1, s1), r2 = __fsub_rn(d2, s2);
      accA[0] += 1.0;
      if (TR::plane) {
        float4 nv = nvp;
        if (sym) {
          // symmetric metric (transform_estimation.hpp:705-706): n = n_dst + tform.linear() * n_src', with
          // n_src' = transform_.linear() * n_src (transformNormals, core/space_transformations.hpp:374-390)
          const float4 sn = snp;
          const float t0 = __fadd_rn(__fmul_rn(T[0], sn.x), __fadd_rn(__fmul_rn(T[4], sn.y), __fmul_rn(T[8], sn.z)));
          const float t1 = __fadd_rn(__fmul_rn(T[1], sn.x), __fadd_rn(__fmul_rn(T[5], sn.y), __fmul_rn(T[9], sn.z)));
          const float t2 = __fadd_rn(__fmul_rn(T[2], sn.x), __fadd_rn(__fmul_rn(T[6], sn.y), __fmul_rn(T[10], sn.z)));
          nv.x = __fadd_rn(nv.x, __fadd_rn(__fmul_rn(iL[0], t0), __fadd_rn(__fmul_rn(iL[1], t1), __fmul_rn(iL[2], t2))));
          nv.y = __fadd_rn(nv.y, __fadd_rn(__fmul_rn(iL[3], t0), __fadd_rn(__fmul_rn(iL[4], t1), __fmul_rn(iL[5], t2))));
          nv.z = __fadd_rn(nv.z, __fadd_rn(__fmul_rn(iL[6], t0), __fadd_rn(__fmul_rn(iL[7], t1), __fmul_rn(iL[8], t2))));
        }
        float e[6];
        e[0] = __fsub_rn(__fmul_rn(a1, nv.z), __fmul_rn(a2, nv.y));   // (d+s).cross(n)  :337
        e[1] = __fsub_rn(__fmul_rn(a2, nv.x), __fmul_rn(a0, nv.z));
        e[2] = __fsub_rn(__fmul_rn(a0, nv.y), __fmul_rn(a1, nv.x));
        e[3] = nv.x; e[4] = nv.y; e[5] = nv.z;
        const float res = __fadd_rn(__fmul_rn(nv.x, r0), __fadd_rn(__fmul_rn(nv.y, r1), __fmul_rn(nv.z, r2)));  // n.dot(d-s)
        // weight * eq_vec and weight * residual rounded to f32 as the reference forms them (:340-341); entry (r, c), r <= c,
        // is the LOWER-triangle product (w e_c) e_r -- the triangle LDLT reads.  wp = 1 changes nothing.
        double ed[6], wed[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { ed[k] = (double)e[k]; wed[k] = (double)__fmul_rn(wp, e[k]); }
        int k = 1;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) { accA[k] = fma(wed[c], ed[r], accA[k]); ++k; }
        const double rd = (double)__fmul_rn(wp, res);
#pragma unroll
        for (int r = 0; r < 6; ++r) accA[22 + r] = fma(rd, ed[r], accA[22 + r]);
      }
      if (TR::point) {
        // every point-term sum is linear in the weight: w a as the first factor, w r on the right-hand side (:318-319)
        const double wd = (double)wq;
        const double ad[3] = {(double)a0, (double)a1, (double)a2};
        const double wa[3] = {wd * ad[0], wd * ad[1], wd * ad[2]};
        const double rd[3] = {wd * (double)r0, wd * (double)r1, wd * (double)r2};
        accB[0] += wa[0]; accB[1] += wa[1]; accB[2] += wa[2];
        accB[3] = fma(wa[0], ad[0], accB[3]); accB[4] = fma(wa[0], ad[1], accB[4]); accB[5] = fma(wa[0], ad[2], accB[5]);
        accB[6] = fma(wa[1], ad[1], accB[6]); accB[7] = fma(wa[1], ad[2], accB[7]); accB[8] = fma(wa[2], ad[2], accB[8]);
        accB[9] += ad[1] * rd[2] - ad[2] * rd[1];
        accB[10] += ad[2] * rd[0] - ad[0] * rd[2];
        accB[11] += ad[0] * rd[1] - ad[1] * rd[0];
        accB[12] += rd[0]; accB[13] += rd[1]; accB[14] += rd[2];
        accB[15] += wd;      // sum of the weights: the translation block of E E^T
      }
    }
  }
}


// ---- in-tile accumulation (k_search_tiled<ACC>): the per-correspondence vector z and where the sums sit in z z^T ----
// PLANE : z = (e0..e5, res, 1)              e = [(d+s) x n ; n], res = n.(d-s)          transform_estimation.hpp:333-341
// POINT : z = (a0,a1,a2, r0,r1,r2, 1, 0)    a = d+s, r = d-s                             :302-319
// BOTH  : z = (PLANE's 8, a0,a1,a2, r0,r1,r2)
// KABSCH: z = (p0,p1,p2, q0,q1,q2, 1, 0)    raw coordinates                              :25-34
// with d = p - dst_mean, s = q - T*src_mean (the inner Gauss-Newton transform is the identity on the first step), every
// term formed in f32 exactly as accumulate_pair forms it.  slot_terms(): partial-sum slot = Z[i1][j1] - Z[i2][j2]
// (i2 < 0: one term) in accumulate_pair's slot layout; the point-to-point cross product a x r becomes a difference of two
// accumulated products (f64: the cancellation costs ~1e-16 of sum |a_i r_j|, far below the estimator's own round-off).
template <int ACC>
struct FusedZ {
  static constexpr bool plane = (ACC == IM_PLANE || ACC == IM_BOTH);
  static constexpr int NC = (ACC == IM_BOTH) ? 14 : 8;
  static constexpr bool needs_normal = plane;
  __device__ static bool slot_terms(int s, int& i1, int& j1, int& i2, int& j2) {
    i1 = j1 = 0; i2 = j2 = -1;
    if (ACC == IM_KABSCH) {
      if (s == 0) { i1 = 6; j1 = 6; return true; }
      if (s < 4) { i1 = s - 1; j1 = 6; return true; }
      if (s < 7) { i1 = 3 + (s - 4); j1 = 6; return true; }
      if (s < 16) { i1 = (s - 7) / 3; j1 = 3 + (s - 7) % 3; return true; }
      return false;
    }
    if (plane && s < 28) {
      if (s == 0) { i1 = 7; j1 = 7; return true; }
      if (s >= 22) { i1 = s - 22; j1 = 6; return true; }
      int k = s - 1, r = 0;
      while (k >= 6 - r) { k -= 6 - r; ++r; }
      i1 = r; j1 = r + k;
      return true;
    }
    if (ACC == IM_POINT && s == 0) { i1 = 6; j1 = 6; return true; }
    if ((ACC == IM_POINT || ACC == IM_BOTH) && s >= 28 && s < 43) {
      const int A0 = (ACC == IM_BOTH) ? 8 : 0, R0 = A0 + 3, ONE = (ACC == IM_BOTH) ? 7 : 6;
      const int b = s - 28;
      if (b < 3) { i1 = A0 + b; j1 = ONE; return true; }
      if (b < 9) { int k = b - 3, r = 0; while (k >= 3 - r) { k -= 3 - r; ++r; } i1 = A0 + r; j1 = A0 + r + k; return true; }
      if (b == 9) { i1 = A0 + 1; j1 = R0 + 2; i2 = A0 + 2; j2 = R0 + 1; return true; }     // a1 r2 - a2 r1
      if (b == 10) { i1 = A0 + 2; j1 = R0 + 0; i2 = A0 + 0; j2 = R0 + 2; return true; }    // a2 r0 - a0 r2
      if (b == 11) { i1 = A0 + 0; j1 = R0 + 1; i2 = A0 + 1; j2 = R0 + 0; return true; }    // a0 r1 - a1 r0
      i1 = R0 + (b - 12); j1 = ONE;
      return true;
    }
    if ((ACC == IM_POINT || ACC == IM_BOTH) && s == 43) { i1 = j1 = (ACC == IM_BOTH) ? 7 : 6; return true; }   // sum of the (unit) weights = n
    return false;
  }
};

template <int ACC>
__device__ __forceinline__ void fused_z(bool has, float qx, float qy, float qz, const float4 p, const float4 nv, const float* dmean, const float* smt,
                                        float* z) {
#pragma unroll
  for (int k = 0; k < 16; ++k) z[k] = 0.0f;
  if (!has) return;
  if (ACC == IM_KABSCH) {
    z[0] = p.x; z[1] = p.y; z[2] = p.z; z[3] = qx; z[4] = qy; z[5] = qz; z[6] = 1.0f;
    return;
  }
  const float d0 = __fsub_rn(p.x, dmean[0]), d1 = __fsub_rn(p.y, dmean[1]), d2 = __fsub_rn(p.z, dmean[2]);
  const float s0 = __fsub_rn(qx, smt[0]), s1 = __fsub_rn(qy, smt[1]), s2 = __fsub_rn(qz, smt[2]);
  const float a0 = __fadd_rn(d0, s0), a1 = __fadd_rn(d1, s1), a2 = __fadd_rn(d2, s2);
  const float r0 = __fsub_rn(d0, s0), r1 = __fsub_rn(d1, s1), r2 = __fsub_rn(d2, s2);
  if (FusedZ<ACC>::plane) {
    z[0] = __fsub_rn(__fmul_rn(a1, nv.z), __fmul_rn(a2, nv.y));   // (d+s).cross(n)
    z[1] = __fsub_rn(__fmul_rn(a2, nv.x), __fmul_rn(a0, nv.z));
    z[2] = __fsub_rn(__fmul_rn(a0, nv.y), __fmul_rn(a1, nv.x));
    z[3] = nv.x; z[4] = nv.y; z[5] = nv.z;
    z[6] = __fadd_rn(__fmul_rn(nv.x, r0), __fadd_rn(__fmul_rn(nv.y, r1), __fmul_rn(nv.z, r2)));   // n.dot(d-s)
    z[7] = 1.0f;
    if (ACC == IM_BOTH) { z[8] = a0; z[9] = a1; z[10] = a2; z[11] = r0; z[12] = r1; z[13] = r2; }
  } else {
    z[0] = a0; z[1] = a1; z[2] = a2; z[3] = r0; z[4] = r1; z[5] = r2; z[6] = 1.0f;
  }
}

// =====================================================================================================
// LDS-tiled search kernel.
//
// Profiling the per-lane global-memory search showed it bound by the texture-address / L1 path
// (TA busy ~70 %: every candidate is a per-lane 16-byte gather at 64 B/clk/CU, and neighbouring lanes
// fetch the same points again and again).  Here one workgroup owns one TILE = up to TILE_QUERIES queries
// whose sort-time cells share one cube of CUBE_EDGE^3 target-grid cells (the source is sorted cube-major).
// Per tile:
//   1. the REGION of target cells the tile can touch comes from the tile alone, with no pass over its
//      queries: the cube is an oriented box in source space (centre per tile, half-axes common to all
//      tiles); its image under the current transform is bounded by centre' +- |R A| 1, converted to
//      cells, grown by one cell and clipped to the grid (14^3 cells when the transform has not moved
//      since the sort).  Queries whose current cell lies outside that box (f32 rounding, clamped
//      out-of-grid queries that moved in) are handed to the clean-up pass, so the box only has to be
//      right for performance, never for correctness;
//   2. the region's cell table (rows x (RX+1) cell_start values) is fetched into LDS -- its loads and the
//      query loads are issued together at kernel start, nothing waits on the queries; one wave turns the
//      row lengths into LDS offsets while the others transform their queries and pick their octant
//      blocks; then all waves copy the rows into LDS with 16 lanes per row (each row of the region is
//      ONE contiguous run of the sorted target array) -- every target point is fetched once per tile
//      instead of once per lane;
//   3. every lane searches out of LDS (ds_read_b128): (a) octant-first in straight-line code; the queries
//      the octant does not prove are queued in LDS and (b) searched in the full 3x3x3 block, again in
//      straight-line code, packed densely over the lanes (so a wave never runs the long block for a few
//      of its lanes, and a source still far from its final pose -- most octant proofs failing -- costs
//      one dense second pass instead of a divergent one).
// Exactness is unchanged: queries whose 3x3x3 block does not prove the result (sparse data, large
// radius), queries outside the grid or outside the tile's box, the slabs of a region that exceed the LDS
// budget (and whole tiles whose region cannot be staged at all: queries that drifted far from their
// sort-time cells) fall back to the global-memory search of the clean-up pass, k_search_todo.
constexpr int TILE_MAXE = CILHIP_TILE_MAXE;                  // entries of the staged cell table: rows * (RX + 1)
constexpr int TILE_QPT = TILE_QUERIES / TILE_THREADS;        // queries per thread
constexpr int TILE_WAVES = TILE_THREADS / 64;
constexpr uint32_t DEFER_MARK = 0xFFFFFFFEu;                 // nn_pos value: "the LDS tile could not settle this query" (between a tile's 3x3x3 pass and its home lanes)
constexpr int FUSED_WAVE_BYTES = 3584;                       // per-wave scratch of the in-tile accumulation (64 correspondences x 14 floats) carved from the point buffer
static_assert(FUSED_WAVE_BYTES * TILE_WAVES <= TILE_BYTES, "the accumulation scratch reuses the tile's point buffer");
static_assert(FUSED_WAVE_BYTES >= 64 * 8 * 4 + 64 && FUSED_WAVE_BYTES >= 4 * 64 * 8, "scratch holds the padded 8-float layout and the wave's 16x16 f64 tile");
static_assert(TILE_CAP + 8 < 65536, "LDS slots are packed into 16 bits");
static_assert(TILE_MAXE <= 65536 && TILE_MAXROWS <= 32767, "OctQuery packs a table index and a row into 16 bits each");
static_assert(TILE_MAXROWS <= 64 * 8, "the row scan holds at most 8 rows per lane of one wave");

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- The staged tile in LDS: PAIRS of records {x0, x1, y0, y1, z0, z1}, 24 bytes, slots in the order of the sorted target array
// row by row (slot j = half j & 1 of pair j >> 1). ----
// (No index: a match is named by its LDS slot -> sorted position; which of several EXACTLY equidistant points wins is never
//  decided in the tile -- a tie is noticed, confirmed and handed to the clean-up pass, which applies the tie rule with full keys.)
// One ds_read_b64 delivers the SAME coordinate of the two records of a pair as an aligned register pair -- the operands of the packed
// f32 instructions (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations per component, so d2 still rounds exactly as
// ((dx*dx)+(dy*dy))+(dz*dz)) -- at the LDS's full rate (256 B/clk; ds_read2_b32 over 12-byte records measured half of it, and the
// LDS busy for half of the kernel's time).  A run that starts on an odd slot is read from the even slot before it: one more real
// target point among the candidates -- a superset of the block never hurts the minimum or the proof.
//
// Candidates are ranked by a 32-bit key: the bits of d2 with the low KB bits replaced by the candidate's SLOT CODE (a compile-time
// constant per straight-line slot).  The running smallest and second smallest keys cost one v_min_u32 and one v_med3_u32 per
// candidate, nothing is selected or compared in 64 bits, and the winner's slot comes out of the key.  The truncation is monotone, so
// the smallest key belongs to a candidate whose d2 is within 2^(KB-23) of the smallest; whenever the two smallest keys agree in
// their distance bits (two candidates nearer to each other than the truncation, a genuine tie, a record read through two runs),
// or the smallest lies within the truncation of the search radius, the block is scanned again EXACTLY (clamped runs, full f32
// compares: a few lanes of some waves).  The truncated second smallest key is a valid LOWER bound of every other candidate's d2,
// which is all the margin keys and the feature search need of it.
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

struct Run4 { f32x2 x0, y0, z0, x1, y1, z1; };   // four consecutive records as coordinate pairs (slots 0,1 and 2,3)

// request two consecutive pairs (four slots) starting at the pair at LDS byte address `addr` (six ds_read_b64; nothing waits here)
__device__ __forceinline__ void lds_issue_run4(uint32_t addr, Run4& r) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(r.x0) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(r.y0) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(r.z0) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(r.x1) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(r.y1) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:40" : "=v"(r.z1) : "v"(addr));
}
// one pair
__device__ __forceinline__ void lds_issue_pair(uint32_t addr, f32x2& x, f32x2& y, f32x2& z) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(y) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(z) : "v"(addr));
}
// slot -> LDS byte address of its pair / dword index of its x inside the float array
__device__ __forceinline__ uint32_t lds_pair_addr(uint32_t base, uint32_t slot) { return base + __umul24(slot >> 1, 24u); }
__device__ __forceinline__ uint32_t lds_slot_x(uint32_t slot) { return __umul24(slot >> 1, 6u) + (slot & 1u); }
// wait until at most N of the LDS operations issued so far are outstanding (they return in order, so everything requested
// before the youngest N has arrived); the registers go through the statement so that no use is scheduled ahead of it
template <int N>
__device__ __forceinline__ void lds_wait_run4(Run4& r) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(r.x0), "+v"(r.y0), "+v"(r.z0), "+v"(r.x1), "+v"(r.y1), "+v"(r.z1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait_pair(f32x2& x, f32x2& y, f32x2& z) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x), "+v"(y), "+v"(z) : "n"(N));
}

// two candidates against the two smallest keys (b <= s); `code` is a compile-time constant at every call site (an inline operand)
__device__ __forceinline__ void eval_pair(uint32_t code, const f32x2 X, const f32x2 Y, const f32x2 Z, const f32x2 qx, const f32x2 qy, const f32x2 qz, uint32_t kmask,
                                          uint32_t& b, uint32_t& s) {
  const f32x2 dx = qx - X, dy = qy - Y, dz = qz - Z;
  const f32x2 e = (dx * dx + dy * dy) + dz * dz;
  const uint32_t k0 = (__float_as_uint(e.x) & kmask) | code, k1 = (__float_as_uint(e.y) & kmask) | (code + 1u);
  s = umed3(b, s, k0); b = min(b, k0);
  s = umed3(b, s, k1); b = min(b, k1);
}
__device__ __forceinline__ void eval_run4(uint32_t code, const Run4& r, const f32x2 qx, const f32x2 qy, const f32x2 qz, uint32_t kmask, uint32_t& b, uint32_t& s) {
  eval_pair(code, r.x0, r.y0, r.z0, qx, qy, qz, kmask, b, s);
  eval_pair(code + 2u, r.x1, r.y1, r.z1, qx, qy, qz, kmask, b, s);
}

// LDS tile index -> position in the global sorted target array (row found by binary search over rowbase)
__device__ __forceinline__ uint32_t lds_to_global(uint32_t l, const uint32_t* rowbase, const uint32_t* rowdelta, int rows) {
  if (l == NONE_U32) return NONE_U32;
  int lo = 0, hi = rows;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rowbase[mid] <= l) lo = mid; else hi = mid;
  }
  return l + rowdelta[lo];
}

struct TileLds {
  const float* lp;            // the staged pairs {x0, x1, y0, y1, z0, z1}
  uint32_t lp_addr;           // the same, as an LDS byte address
  uint32_t pad;               // slot of the first of the 8 pad records (d2 = inf) behind the staged points
  const uint32_t* lcs;        // [rows][W1] cell_start values (GLOBAL sorted positions) of the region's cells, +1 end column
  const uint32_t* rowbase;    // [rows+1] LDS index of the first staged point of each region row
  const uint32_t* rowdelta;   // [rows]   global sorted position - LDS index, per row (mod 2^32)
  int lox, loy, loz, RY, W1, rows;
};

__device__ __forceinline__ float lds_d2(const float* lp, uint32_t j, float qx, float qy, float qz) {
  const float* const w = lp + lds_slot_x(j);
  return d2_pinned(qx, qy, qz, w[0], w[2], w[4]);
}

// the exact scan of a block's runs (clamped): smallest d2 (lowest slot among equals), the second smallest over the OTHER slots,
// whether another slot repeats the smallest exactly
struct ExactScan { float e1, e2; uint32_t bl; bool tie; };
__device__ __forceinline__ void exact_take(ExactScan& x, float e, uint32_t j) {
  if (e < x.e1) { x.e2 = x.e1; x.e1 = e; x.bl = j; x.tie = false; }
  else { x.tie |= e == x.e1; x.e2 = fminf(x.e2, e); }
}

#ifndef CILHIP_OCT_EXTRA
#define CILHIP_OCT_EXTRA 1  /* further quads taken in straight-line code before the overflow loop */
#endif
constexpr int OCT_CAND = 4;   // candidates per run evaluated unconditionally (a run = 2 cells, ~2 points at the default occupancy)
constexpr int OCT_EXTRA = CILHIP_OCT_EXTRA;
constexpr uint32_t OCT_KMASK = 0xFFFFFFE0u, OCT_NOCODE = 31u, OCT_OVER = 16u;   // 5 code bits: 16 straight-line slots, 4 of the current overflow quad

// What a lane keeps of one query between its preparation (while the target points are still in flight) and the search.
struct OctQuery {
  float qx, qy, qz;
  int ebrow;     // low 16 bits: cell-table index of the first cell of run 0 (runs 1..3: + W1, + RY*W1, + (RY+1)*W1);
                 // high 16 bits: region row of run 0 (runs 1..3: +1, +RY, +RY+1)
};

// Octant-first search (the common case): the 2x2x2 block of cells on the side of q's own cell that q
// leans towards contains every target point closer than the distance from q to that block's faces, which
// is at least half a cell.  The block is 4 runs of the sorted target array (2 x-adjacent cells each).
// The grid carries GRID_PAD layers of empty cells around the data and the fast path only takes queries whose
// cell is not in the outermost layer, so the block never leaves the grid: no clipping, no validity flags.
// octant_prepare() only needs the query; octant_search() runs out of LDS in STRAIGHT-LINE code: every lane
// evaluates exactly OCT_CAND unclamped candidates per run (reading past a short run only evaluates further
// real target points or the far-away pad records -- never wrong), no per-lane loop or branch, so the wave
// executes each instruction once with all lanes busy.
// Returns whether the block lies inside the staged region [lo, hi] (cells, inclusive) -- all the fast path needs.  The
// block of the other queries is clamped into the region so that the code below stays branch-free; their result is dropped.
__device__ __forceinline__ bool octant_prepare(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                               int lox, int loy, int loz, int hix, int hiy, int hiz, int RY, int W1, OctQuery& o) {
  o.qx = qx; o.qy = qy; o.qz = qz;
  // offsets of q inside its cell; q leans to the low side of an axis when the offset is below half a cell
  const float ux = qx - (g.ox + (float)cx * g.cell), uy = qy - (g.oy + (float)cy * g.cell), uz = qz - (g.oz + (float)cz * g.cell);
  const float half = 0.5f * g.cell;
  const int bx = cx + ((ux >= half) ? 0 : -1), by = cy + ((uy >= half) ? 0 : -1), bz = cz + ((uz >= half) ? 0 : -1);   // low corner of the block
  const bool inside = (bx >= lox) & (bx < hix) & (by >= loy) & (by < hiy) & (bz >= loz) & (bz < hiz);
  const int kx = min(max(bx, lox), hix - 1), ky = min(max(by, loy), hiy - 1), kz = min(max(bz, loz), hiz - 1);
  const int row00 = (kz - loz) * RY + (ky - loy);
  o.ebrow = (row00 * W1 + (kx - lox)) | (row00 << 16);
  return inside;
}

// distance from q to the nearest face of its octant block: along an axis the two-cell span's nearest face is at
// max(u, cell - u), u = offset of q inside its cell (the far face of the own cell on the side q leans away from; the
// other face of the span is a full cell further).  Recomputed at search time rather than carried in registers.
__device__ __forceinline__ float octant_bound(const GridDev& g, float qx, float qy, float qz) {
  const float ux = qx - (g.ox + floorf((qx - g.ox) * g.inv_cell) * g.cell), uy = qy - (g.oy + floorf((qy - g.oy) * g.inv_cell) * g.cell),
              uz = qz - (g.oz + floorf((qz - g.oz) * g.inv_cell) * g.cell);
  return fminf(fminf(fmaxf(ux, g.cell - ux), fmaxf(uy, g.cell - uy)), fmaxf(uz, g.cell - uz));
}

// Returns true (result proven exact) iff the best found is strictly nearer than any point outside the block can be and no other
// record lies at exactly its distance.  best.key = bits(d2) << 32 (exact d2 of the winner; the radius when there is none), best.pos its
// sorted position, bl_out its LDS slot; *second_out = a lower bound of the squared distance of every OTHER record evaluated (at most the
// radius); *bound_out = the (shrunk) distance from q to the nearest face of the block.
__device__ __forceinline__ bool octant_search(const GridDev& g, const TileLds& t, const OctQuery& o, float max_sq, NN& best, uint32_t& bl_out,
                                              float* second_out = nullptr, float* bound_out = nullptr) {
  const f32x2 qx = {o.qx, o.qx}, qy = {o.qy, o.qy}, qz = {o.qz, o.qz};
  const uint32_t init = (__float_as_uint(max_sq) & OCT_KMASK) | OCT_NOCODE;
  uint32_t b = init, s = init;
  uint32_t rj[4], re[4];
  const int row00 = o.ebrow >> 16, eb00 = o.ebrow & 0xFFFF;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = row00 + (k >> 1) * t.RY + (k & 1);
    const int eb = eb00 + ((k >> 1) * t.RY + (k & 1)) * t.W1;
    const uint32_t dl = t.rowdelta[row];
    rj[k] = (t.lcs[eb] - dl) & ~1u;      // (the even slot at or before the run's first)
    re[k] = t.lcs[eb + 2] - dl;
  }
  {
    Run4 r0, r1;
    lds_issue_run4(lds_pair_addr(t.lp_addr, rj[0]), r0);
    lds_issue_run4(lds_pair_addr(t.lp_addr, rj[1]), r1);
    lds_wait_run4<6>(r0);
    eval_run4(0u, r0, qx, qy, qz, OCT_KMASK, b, s);
    lds_issue_run4(lds_pair_addr(t.lp_addr, rj[2]), r0);
    lds_wait_run4<6>(r1);
    eval_run4(4u, r1, qx, qy, qz, OCT_KMASK, b, s);
    lds_issue_run4(lds_pair_addr(t.lp_addr, rj[3]), r1);
    lds_wait_run4<6>(r0);
    eval_run4(8u, r0, qx, qy, qz, OCT_KMASK, b, s);
    lds_wait_run4<0>(r1);
    eval_run4(12u, r1, qx, qy, qz, OCT_KMASK, b, s);
  }
  // Runs longer than OCT_CAND.  A flattened per-lane loop costs the whole wave its longest lane, and some lane of almost every
  // wave has one long run.  So: OCT_EXTRA more quads in straight-line code -- every lane takes the first run it has not finished
  // (the others read the pad records: d2 = inf) -- and only then the loop, which few waves enter.  The overflow quad's slots carry
  // the codes 16..19; `bov` remembers the quad that last improved the minimum.
  uint32_t nj[4], bov = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) nj[k] = rj[k] + OCT_CAND;
#pragma unroll
  for (int x = 0; x < OCT_EXTRA; ++x) {
    const bool c0 = nj[0] < re[0], c1 = nj[1] < re[1], c2 = nj[2] < re[2], c3 = nj[3] < re[3];
    const uint32_t jx = c0 ? nj[0] : c1 ? nj[1] : c2 ? nj[2] : c3 ? nj[3] : t.pad;      // (t.pad: even)
    nj[0] += c0 ? 4u : 0u;
    nj[1] += (!c0 & c1) ? 4u : 0u;
    nj[2] += (!c0 & !c1 & c2) ? 4u : 0u;
    nj[3] += (!c0 & !c1 & !c2 & c3) ? 4u : 0u;
    Run4 r;
    lds_issue_run4(lds_pair_addr(t.lp_addr, jx), r);
    const uint32_t bprev = b;
    lds_wait_run4<0>(r);
    eval_run4(OCT_OVER, r, qx, qy, qz, OCT_KMASK, b, s);
    bov = b != bprev ? jx : bov;
  }
  for (;;) {
    const bool c0 = nj[0] < re[0], c1 = nj[1] < re[1], c2 = nj[2] < re[2], c3 = nj[3] < re[3];
    if (!(c0 | c1 | c2 | c3)) break;
    const uint32_t jx = c0 ? nj[0] : c1 ? nj[1] : c2 ? nj[2] : nj[3];
    nj[0] += c0 ? 4u : 0u;
    nj[1] += (!c0 & c1) ? 4u : 0u;
    nj[2] += (!c0 & !c1 & c2) ? 4u : 0u;
    nj[3] += (!c0 & !c1 & !c2 & c3) ? 4u : 0u;
    Run4 r;
    lds_issue_run4(lds_pair_addr(t.lp_addr, jx), r);
    const uint32_t bprev = b;
    lds_wait_run4<0>(r);
    eval_run4(OCT_OVER, r, qx, qy, qz, OCT_KMASK, b, s);
    bov = b != bprev ? jx : bov;
  }
  // the winner's slot out of its code
  const uint32_t code = b & ~OCT_KMASK;
  uint32_t bl = NONE_U32;
  float e1 = max_sq, second = __uint_as_float(s & OCT_KMASK);
  bool tie = false;
  if (code != OCT_NOCODE) {
    const uint32_t k = code >> 2;
    bl = (code >= OCT_OVER ? bov : (k == 0 ? rj[0] : k == 1 ? rj[1] : k == 2 ? rj[2] : rj[3])) + (code & 3u);
    e1 = lds_d2(t.lp, bl, o.qx, o.qy, o.qz);
    if ((((b ^ s) & OCT_KMASK) == 0u) | (((b ^ init) & OCT_KMASK) == 0u)) {
      // the two smallest keys agree in their distance bits, or the smallest lies within the truncation of the radius: decided exactly,
      // over the block's own four runs, clamped (the runs are looked up again HERE: nothing of them stays live for this rare branch)
      ExactScan x{INFINITY, INFINITY, NONE_U32, false};
      int ebr = o.ebrow;
      asm volatile("" : "+v"(ebr));
      const int crow = ebr >> 16, ceb = ebr & 0xFFFF;
      for (int k2 = 0; k2 < 4; ++k2) {
        const int krow = crow + (k2 >> 1) * t.RY + (k2 & 1);
        const int keb = ceb + ((k2 >> 1) * t.RY + (k2 & 1)) * t.W1;
        const uint32_t kdl = t.rowdelta[krow];
        const uint32_t j1 = t.lcs[keb + 2] - kdl;
        for (uint32_t j = t.lcs[keb] - kdl; j < j1; ++j) exact_take(x, lds_d2(t.lp, j, o.qx, o.qy, o.qz), j);
      }
      bl = x.bl; e1 = x.e1; second = x.e2; tie = x.tie;      // (x.bl == NONE: the key's winner was an over-read record outside the block)
    }
    if (!(e1 < max_sq)) { second = fminf(second, e1); bl = NONE_U32; e1 = max_sq; tie = false; }
  }
  second = fminf(second, max_sq);
  best.key = ((unsigned long long)__float_as_uint(e1) << 32);
  // LDS index -> global position: the winner normally lies in the row of its run (one table read); an
  // over-read winner past the end of that row (or one picked up through a clipped run) takes the binary search
  uint32_t pos = NONE_U32;
  if (bl != NONE_U32) {
    const int k = (int)(bl >= rj[1]) + (int)(bl >= rj[2]) + (int)(bl >= rj[3]);   // the runs ascend in LDS
    const int row = row00 + (k >> 1) * t.RY + (k & 1);
    if (bl >= t.rowbase[row] && bl < t.rowbase[row + 1]) pos = bl + t.rowdelta[row];
    else pos = lds_to_global(bl, t.rowbase, t.rowdelta, t.rows);      // (the even slot before an odd run start belongs to the row before)
  }
  best.pos = pos;
  if (second_out) *second_out = second;
  bl_out = bl;      // LDS index of the winner (NONE_U32: nothing within the radius): the in-tile accumulation reads the point from there
  const float bd = octant_bound(g, o.qx, o.qy, o.qz) - g.margin;
  if (bound_out) *bound_out = bd;      // (shrunk) distance from q to the nearest face of the block: every point outside it is at least that far
  return bd > 0.0f && e1 < bd * bd * KSHRINK && !tie;
}

// The full 3x3x3 block of cells around the query's cell, for the queries the octant block did not prove, in
// STRAIGHT-LINE code: 9 runs of 3 x-adjacent cells, B27_CAND unclamped candidates each, no culling, no per-lane loop --
// the wave executes each instruction once with all its lanes busy (the queued queries are packed densely over the
// lanes, see phase 3b of the kernel).  Longer runs go through a per-lane list afterwards (most lanes: none or one).
// Needs cx, cy, cz one cell inside the region on every side (the fast range).  Returns false if the block does not
// prove the result (the query then goes to the clean-up pass).  Keys as in octant_search, six code bits.
constexpr int B27_CAND = 6;
constexpr uint32_t B27_KMASK = 0xFFFFFFC0u, B27_NOCODE = 63u, B27_OVER = 56u;   // 54 straight-line slots (run * 6 + slot), 56 / 57: the current overflow pair
__device__ __forceinline__ bool block27_search(const GridDev& g, const TileLds& t, float qx_, float qy_, float qz_,
                                               int cx, int cy, int cz, float max_sq, NN& best, float* second_out = nullptr, float* bound_out = nullptr) {
  const f32x2 qx = {qx_, qx_}, qy = {qy_, qy_}, qz = {qz_, qz_};
  const uint32_t init = (__float_as_uint(max_sq) & B27_KMASK) | B27_NOCODE;
  uint32_t b = init, s = init;
  const int row0 = (cz - t.loz) * t.RY + (cy - t.loy);       // region row of the own cell
  const int e0 = row0 * t.W1 + (cx - t.lox) - 1;             // table entry of the x-1 cell of that row
  uint32_t over = 0;            // bit r: run r is longer than B27_CAND
  {
    // two register sets: the records of run r + 1 fly while run r is evaluated and the table entries of run r + 2 are read
    struct Run6 { f32x2 x0, y0, z0, x1, y1, z1, x2, y2, z2; } A, B;
    auto issue6 = [&](uint32_t rj, Run6& q) {
      const uint32_t addr = lds_pair_addr(t.lp_addr, rj);
      lds_issue_pair(addr, q.x0, q.y0, q.z0);
      lds_issue_pair(addr + 24u, q.x1, q.y1, q.z1);
      lds_issue_pair(addr + 48u, q.x2, q.y2, q.z2);
    };
    auto eval6 = [&](uint32_t code, const Run6& q) {
      eval_pair(code, q.x0, q.y0, q.z0, qx, qy, qz, B27_KMASK, b, s);
      eval_pair(code + 2u, q.x1, q.y1, q.z1, qx, qy, qz, B27_KMASK, b, s);
      eval_pair(code + 4u, q.x2, q.y2, q.z2, qx, qy, qz, B27_KMASK, b, s);
    };
    auto table = [&](int r) -> uint32_t {
      const int off = (r / 3 - 1) * t.RY + (r % 3 - 1);
      const uint32_t dl = t.rowdelta[row0 + off];
      const uint32_t rj = (t.lcs[e0 + off * t.W1] - dl) & ~1u;      // (the even slot at or before the run's first)
      over |= (t.lcs[e0 + off * t.W1 + 3] - dl > rj + (uint32_t)B27_CAND) ? (1u << r) : 0u;
      return rj;
    };
    uint32_t rj0 = table(0), rj1 = table(1);
    issue6(rj0, A);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      Run6& cur = (r & 1) ? B : A;
      Run6& nxt = (r & 1) ? A : B;
      if (r < 8) issue6((r & 1) ? rj0 : rj1, nxt);
      if (r < 8) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(cur.x0), "+v"(cur.y0), "+v"(cur.z0), "+v"(cur.x1), "+v"(cur.y1), "+v"(cur.z1), "+v"(cur.x2), "+v"(cur.y2), "+v"(cur.z2));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur.x0), "+v"(cur.y0), "+v"(cur.z0), "+v"(cur.x1), "+v"(cur.y1), "+v"(cur.z1), "+v"(cur.x2), "+v"(cur.y2), "+v"(cur.z2));
      eval6((uint32_t)(r * 6), cur);
      if (r < 7) { if (r & 1) rj1 = table(r + 2); else rj0 = table(r + 2); }
    }
  }
  // the rest of the long runs: each lane walks its own list of them (most lanes: none or one), so the wave pays
  // the longest list, not one pass per run of the block
  uint32_t bov = 0;
  while (over) {
    const int r = __ffs(over) - 1;
    over &= over - 1;
    const int dz = (r * 11) >> 5, dy = r - 3 * dz;
    const int off = (dz - 1) * t.RY + (dy - 1);
    const uint32_t dl = t.rowdelta[row0 + off];
    const uint32_t re = t.lcs[e0 + off * t.W1 + 3] - dl;
    for (uint32_t j = ((t.lcs[e0 + off * t.W1] - dl) & ~1u) + (uint32_t)B27_CAND; j < re; j += 2) {
      f32x2 ax, ay, az;
      lds_issue_pair(lds_pair_addr(t.lp_addr, j), ax, ay, az);
      const uint32_t bprev = b;
      lds_wait_pair<0>(ax, ay, az);
      eval_pair(B27_OVER, ax, ay, az, qx, qy, qz, B27_KMASK, b, s);
      bov = b != bprev ? j : bov;
    }
  }
  const uint32_t code = b & ~B27_KMASK;
  uint32_t bl = NONE_U32;
  int brow = row0;
  float e1 = max_sq, second = __uint_as_float(s & B27_KMASK);
  bool tie = false;
  if (code != B27_NOCODE) {
    if (code >= B27_OVER) bl = bov + (code - B27_OVER);
    else {
      const int r = (int)((code * 43u) >> 8);                   // code / 6 for code < 54
      const int dz = (r * 11) >> 5, dy = r - 3 * dz;            // r/3, r%3 for r in 0..8
      const int off = (dz - 1) * t.RY + (dy - 1);
      brow = row0 + off;
      bl = ((t.lcs[e0 + off * t.W1] - t.rowdelta[brow]) & ~1u) + (code - (uint32_t)r * 6u);
    }
    e1 = lds_d2(t.lp, bl, qx_, qy_, qz_);
    if ((((b ^ s) & B27_KMASK) == 0u) | (((b ^ init) & B27_KMASK) == 0u)) {
      // decided exactly over the block's own nine runs, clamped (see octant_search)
      ExactScan x{INFINITY, INFINITY, NONE_U32, false};
      int crow = (cz - t.loz) * t.RY + (cy - t.loy), ce0;
      asm volatile("" : "+v"(crow));
      ce0 = crow * t.W1 + (cx - t.lox) - 1;
      for (int r = 0; r < 9; ++r) {
        const int off = (r / 3 - 1) * t.RY + (r % 3 - 1);
        const uint32_t dl = t.rowdelta[crow + off];
        const uint32_t rj = t.lcs[ce0 + off * t.W1] - dl, re = t.lcs[ce0 + off * t.W1 + 3] - dl;
        for (uint32_t j = rj; j < re; ++j) exact_take(x, lds_d2(t.lp, j, qx_, qy_, qz_), j);
      }
      bl = x.bl; e1 = x.e1; second = x.e2; tie = x.tie;
    }
    if (!(e1 < max_sq)) { second = fminf(second, e1); bl = NONE_U32; e1 = max_sq; tie = false; }
  }
  second = fminf(second, max_sq);
  best.key = ((unsigned long long)__float_as_uint(e1) << 32);
  uint32_t pos = NONE_U32;
  if (bl != NONE_U32) {
    if (bl >= t.rowbase[brow] && bl < t.rowbase[brow + 1]) pos = bl + t.rowdelta[brow];   // (an over-read / overflow / re-scanned winner: the binary search)
    else pos = lds_to_global(bl, t.rowbase, t.rowdelta, t.rows);
  }
  best.pos = pos;
  // does the 3x3x3 block prove exactness?  (no bound from a side where the block reaches the edge of the grid)
  const float ux = qx_ - (g.ox + (float)cx * g.cell), uy = qy_ - (g.oy + (float)cy * g.cell), uz = qz_ - (g.oz + (float)cz * g.cell);
  float bd = INFINITY;
  if (cx - 1 > 0) bd = fminf(bd, ux);
  if (cx + 2 < g.nx) bd = fminf(bd, g.cell - ux);
  if (cy - 1 > 0) bd = fminf(bd, uy);
  if (cy + 2 < g.ny) bd = fminf(bd, g.cell - uy);
  if (cz - 1 > 0) bd = fminf(bd, uz);
  if (cz + 2 < g.nz) bd = fminf(bd, g.cell - uz);
  if (second_out) *second_out = second;
  if (tie) { if (bound_out) *bound_out = 0.0f; return false; }      // a confirmed tie goes to the clean-up pass
  if (bd == INFINITY) { if (bound_out) *bound_out = INFINITY; return true; }
  bd = fmaxf(bd, 0.0f) + g.cell - 2.0f * g.margin;
  if (bound_out) *bound_out = bd;
  return bd > 0.0f && e1 < bd * bd * KSHRINK;
}

#ifdef CILHIP_EXP_PHASE_CLOCKS
__device__ unsigned long long g_phase_clk[8];
#define PHASE_CLK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase_clk[k], now_ - tprev_); tprev_ = now_; } } while (0)
__device__ unsigned long long g_warm_clk[16];
__device__ unsigned long long g_warm_stamp[4096][3];      // per block of the LAST k_warm launch: start / end (100 MHz wall clock)
#define WARM_CLK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_warm_clk[k], now_ - tprev_); atomicAdd(&g_warm_clk[8 + (k)], 1ull); tprev_ = now_; } } while (0)
static void debug_dump_warm_clocks() {
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_warm_clk), sizeof h) != hipSuccess) return;
  if (h[8] == 0) return;
  fprintf(stderr, "[warm clocks, 100 MHz ticks per block (thread 0: wave 0), %llu blocks] prologue=%.1f stream=%.1f levelA=%.1f (%.2f rounds) levelB=%.1f (%.2f) -=%.1f (%.2f) end=%.1f\n",
          h[8], (double)h[0] / h[8], (double)h[1] / h[8], (double)h[2] / h[8], (double)h[10] / h[8], (double)h[3] / h[8], (double)h[11] / h[8], (double)h[4] / h[8],
          (double)h[12] / h[8], (double)h[5] / h[8]);
  fprintf(stderr, "[warm list] entries at level A %llu, left open %llu (bound = radius: %llu, bound > cell: %llu)\n", h[6], h[7], h[14], h[15]);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warm_clk), h, sizeof h);
  // the last launch, block by block: when it started / ended relative to the first start; by XCD (blockIdx & 7)
  static unsigned long long st[4096][3];
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_warm_stamp), sizeof st) != hipSuccess) return;
  unsigned long long t0 = ~0ull; int nb = 0;
  for (int b = 0; b < 4096; ++b) if (st[b][1]) { nb = b + 1; if (st[b][0] < t0) t0 = st[b][0]; }
  if (!nb) return;
  double smax = 0, emax = 0, emin = 1e30, dsum = 0, dmin = 1e30, dmax = 0;
  double xe[8] = {0}, xd[8] = {0}; int xn[8] = {0};
  for (int b = 0; b < nb; ++b) {
    const double s0 = (double)(st[b][0] - t0) / 100.0, e0 = (double)(st[b][1] - t0) / 100.0, d = e0 - s0;
    if (s0 > smax) smax = s0; if (e0 > emax) emax = e0; if (e0 < emin) emin = e0; dsum += d; if (d < dmin) dmin = d; if (d > dmax) dmax = d;
    if (e0 > xe[b & 7]) xe[b & 7] = e0; xd[b & 7] += d; ++xn[b & 7];
  }
  {
    int hist[24] = {0}; double lw = 0, lwo = 0; int nw = 0, nwo = 0; double pos[8] = {0}; int posn[8] = {0};
    for (int b = 0; b < nb; ++b) {
      const double d = (double)(st[b][1] - st[b][0]) / 100.0;
      int k = (int)(d / 5.0); if (k > 23) k = 23; ++hist[k];
      if (st[b][2]) { lw += d; ++nw; } else { lwo += d; ++nwo; }
      const int oct = ((b >> 3) * 8) / ((nb + 7) / 8); pos[oct < 8 ? oct : 7] += d; ++posn[oct < 8 ? oct : 7];
    }
    fprintf(stderr, "[warm stamps] lifetime histogram (5 us bins):");
    for (int k = 0; k < 24; ++k) fprintf(stderr, " %d", hist[k]);
    fprintf(stderr, "\n[warm stamps] blocks with listed queries: %d, avg life %.1f; without: %d, avg life %.1f; avg life by position of the chunk inside its XCD's share (eighths):", nw, nw ? lw / nw : 0.0, nwo, nwo ? lwo / nwo : 0.0);
    for (int k = 0; k < 8; ++k) fprintf(stderr, " %.1f", posn[k] ? pos[k] / posn[k] : 0.0);
    fprintf(stderr, "\n");
  }
  fprintf(stderr, "[warm stamps, last launch, %d blocks, us] last start=%.1f  first end=%.1f  last end=%.1f  block lifetime min/avg/max=%.1f/%.1f/%.1f  per XCD (avg life, last end):", nb, smax, emin, emax, dmin, dsum / nb, dmax);
  for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f,%.1f", xn[x] ? xd[x] / xn[x] : 0.0, xe[x]);
  fprintf(stderr, "\n");
  memset(st, 0, sizeof st);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warm_stamp), st, sizeof st);
}
void debug_dump_phase_clocks() {
  debug_dump_warm_clocks();
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_clk), sizeof h) != hipSuccess) return;
  unsigned long long tot = 0;
  for (int k = 0; k < 8; ++k) tot += h[k];
  fprintf(stderr, "[phase clocks, 100 MHz ticks summed over blocks, thread 0] table=%llu scan+prep=%llu stage=%llu search(+3b, masks)=%llu tail loads+z=%llu mfma+D=%llu row=%llu total=%llu\n",
          h[0], h[1], h[2], h[3], h[4], h[5], h[6], tot);
  memset(h, 0, sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clk), h, sizeof h);
}
#else
#define PHASE_CLK(k)
#define WARM_CLK(k)
#endif

// Region of every tile under the CURRENT transform, once per search instead of once per wave of the search kernel (the
// arithmetic is wave-uniform there, but still costs every wave ~100 vector instructions): the image of the tile's cube
// (oriented box in source space: centre per tile, half-axes common to all tiles) is T c +- |R A| 1, converted to cells.
// box[8t..8t+7] = bx0, bx1, by0, by1, bz0, bz1 (cell range of the box, already extended into the empty layer next to the
// data where it touches the first / last data cells), then the two reciprocals the search kernel's table fill uses.
__device__ __forceinline__ void compute_tile_box(const float* T, const BoxArgs& g, uint32_t t) {
  const bool trim = g.trim != 0;
  const float4 c4 = g.tile_center[t];
  float ccx, ccy, ccz;
  transform_point(T, c4.x, c4.y, c4.z, ccx, ccy, ccz);
  float ext[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float e = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      e += fabsf(T[i] * g.tile_axes[k] + T[i + 4] * g.tile_axes[3 + k] + T[i + 8] * g.tile_axes[6 + k]);   // |(R A)_ik|
    ext[i] = e;
  }
  const float SHR = 1.0e-3f, BIG = 1.0e9f;   // the cube is half-open: shrink by 1e-3 cell so that an unmoved cube maps to itself
  // Per axis: the cells [b0, b1] the image covers, and the REGION the tile stages = those cells plus a halo cell on a
  // side only where a query can lean that way (trim: the accumulating form of the search; the search-only form keeps both
  // halos -- its second, 3x3x3 pass needs the whole neighbourhood of a cell): the octant block of a query is the 2 cells
  // on the side of its own cell it leans to, so the low halo is needed only if the first cell can hold a query in its lower half (the image starts
  // below the cell's middle), the high halo only if the last cell can hold one in its upper half.  A cube of 12 cells
  // shifted by a fraction of a cell covers 13 cells and needs ONE of the two halos: 14 cells per axis instead of 15 --
  // a fifth fewer points to stage, and regions that stay inside the LDS budget.  (Only a matter of speed: the search
  // kernel tests every query's block against the region and hands what does not fit to the clean-up pass.)
  const float HALF_SLACK = 0.01f;
  int lo[3], hi[3];
  const float cc3[3] = {ccx, ccy, ccz}, o3[3] = {g.ox, g.oy, g.oz};
  const int n3[3] = {g.nx, g.ny, g.nz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a0 = fminf(fmaxf((cc3[i] - ext[i] - o3[i]) * g.inv_cell + SHR, -BIG), BIG);
    const float a1 = fminf(fmaxf((cc3[i] + ext[i] - o3[i]) * g.inv_cell - SHR, -BIG), BIG);
    int b0 = (int)floorf(a0), b1 = (int)floorf(a1);
    bool halo0 = !trim || (a0 - (float)b0) < 0.5f + HALF_SLACK, halo1 = !trim || (a1 - (float)b1) > 0.5f - HALF_SLACK;
    // a box that reaches the first / last layer of data cells also takes the empty layer next to it: the queries the
    // current transform (or noise) pushed just outside the data's bounding box stay on the fast path
    if (b0 <= GRID_PAD) { b0 = min(b0, GRID_PAD - 1); halo0 = true; }
    if (b1 >= n3[i] - 1 - GRID_PAD) { b1 = max(b1, n3[i] - GRID_PAD); halo1 = true; }
    // (stored as the region shrunk by one cell on both sides, the form the search kernel grows back and clips)
    lo[i] = b0 - (halo0 ? 1 : 0) + 1;
    hi[i] = b1 + (halo1 ? 1 : 0) - 1;
  }
  const int bx0 = lo[0], bx1 = hi[0], by0 = lo[1], by1 = hi[1], bz0 = lo[2], bz1 = hi[2];
  // reciprocals for the flat cell-table fill of the search kernel (division by a run-time width there would be ~20
  // emulated instructions per wave): e / W1 == (e * inv_w1) >> 20 for e < 2^20 / W1, r / RY == (r * inv_ry) >> 16 for r < 3855
  const int W1 = (min(bx1 + 1, g.nx - 1) - max(bx0 - 1, 0) + 1) + 1, RY = min(by1 + 1, g.ny - 1) - max(by0 - 1, 0) + 1;
  const uint32_t inv_w1 = W1 > 0 ? ((1u << 20) + (uint32_t)W1 - 1u) / (uint32_t)W1 : 0u;
  const uint32_t inv_ry = RY > 0 ? (65536u + (uint32_t)RY - 1u) / (uint32_t)RY : 0u;
  int* b = g.tile_box + 8 * (size_t)t;
  b[0] = bx0; b[1] = bx1; b[2] = by0; b[3] = by1; b[4] = bz0; b[5] = bz1; b[6] = (int)inv_w1; b[7] = (int)inv_ry;
}


__global__ void k_tile_boxes(BoxArgs g, const IcpState* __restrict__ st) {
  if (st->done) return;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) *g.defer_flag = 0u;       // (the tiles of the search that follows set it when they defer a query)
  if (t >= g.ntiles) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  compute_tile_box(T, g, t);
}

BoxArgs make_box_args(const IterArgs& a, const float4* tile_center, int* tile_box, uint32_t ntiles, bool trim) {
  BoxArgs b{};
  b.tile_center = tile_center; b.tile_box = tile_box; b.ntiles = ntiles; b.defer_flag = a.defer_flag;
  b.ox = a.grid.ox; b.oy = a.grid.oy; b.oz = a.grid.oz; b.inv_cell = a.grid.inv_cell;
  b.nx = a.grid.nx; b.ny = a.grid.ny; b.nz = a.grid.nz;
  for (int i = 0; i < 9; ++i) b.tile_axes[i] = a.tile_axes[i];
  b.trim = trim ? 1 : 0;
  return b;
}

// A tile that cannot be staged at all hands every query to the clean-up pass (all-ones masks; the pass clips to the
// tile's range) and contributes a zero partial row.
template <int ACC>
__device__ __forceinline__ void defer_whole_tile(const IterArgs& a, uint32_t vb) {
  if ((threadIdx.x & 63u) == 0) {
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) a.defer_mask[(size_t)vb * (2 * TILE_WAVES) + u * TILE_WAVES + (threadIdx.x >> 6)] = ~0ull;
    if (threadIdx.x == 0) { if (__hip_atomic_load(a.defer_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __hip_atomic_store(a.defer_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(a.unproven_cnt + (vb & 63u), (uint32_t)TILE_QUERIES); atomicAdd(a.unproven_cnt + 64u + (vb & 63u), (uint32_t)TILE_QUERIES); }
  }
  if (ACC != IM_NONE && threadIdx.x < SUMS_MAX) a.tile_partials[(size_t)vb * SUMS_MAX + threadIdx.x] = 0.0;
}

// FEAT6 (search-only form): the correspondence search over 6-D point+normal features (PointNormalFeaturesAdaptor,
// common_transformable_feature_adaptors.hpp:60-161).  Candidates are compared by the feature distance d6 = d3 + |w dn|^2,
// which is never below the 3-D distance d3 -- so the tile searches by d3 out of LDS exactly as for points, remembers the
// second smallest d3 it met, fetches the WINNER's normal (one gather) and forms its d6: if that is still below the
// second smallest d3, no other candidate can win (their d6 >= their d3), and the usual geometric proof -- now with d6 --
// settles the query.  Otherwise (normals that disagree by more than the spacing of the candidates) the query goes to the
// clean-up pass, which searches by d6 outright.  Normals are not staged: the LDS budget holds the points.
// LB: the tile also leaves what the warm-started iterations start from -- per settled query the margin key of
// its search (second smallest distance in the block it searched, capped by the gap to the block's faces): the search-only form
// into a.nn_lb next to a.nn_pos; the accumulating form straight into the MATCH RECORDS {matched point, key} {normal} the
// record-reading warm kernel streams (a.warm_rec / a.warm_rec_n), so that the iteration after a tile iteration can already run
// warm-started without a record-writing pass in between.
template <int ACC, bool FEAT6 = false, bool LB = false>
__global__ __launch_bounds__(TILE_THREADS, CILHIP_TILE_WAVES_PER_SIMD) void k_search_tiled(IterArgs a, const uint2* __restrict__ tiles,
                                                                  const int* __restrict__ tile_box, uint32_t ntiles) {
  static_assert(!(LB && FEAT6), "margin keys are a property of the 3-D point search");
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  // XCD-aware tile order: block b runs on XCD b%8 -> each XCD gets one contiguous eighth of the tiles
  const uint32_t per = (ntiles + 7u) >> 3;
  const uint32_t vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || vb >= ntiles) return;

  __shared__ __attribute__((aligned(16))) unsigned char raw[TILE_BYTES];
  __shared__ uint32_t lcs[TILE_MAXE];
  __shared__ uint32_t rowbase[TILE_MAXROWS + 1];
  __shared__ uint32_t rowdelta[TILE_MAXROWS];
  __shared__ uint32_t queue_count;            // queries queued for the 3x3x3 pass (phase 3)
  __shared__ float tform_lds[19];             // the transform (and the motion clock), for phase 3b
  __shared__ uint32_t small_count;            // (LB) queries that leave this tile with a margin a warm-started iteration could not use, or with none
  __shared__ int geom_lds[8];                 // the region's geometry, for phase 3b (so that nothing it derives is kept live from here)
  float* const lp = reinterpret_cast<float*>(raw);      // staged pairs {x0, x1, y0, y1, z0, z1}
  if (threadIdx.x == 0) { queue_count = 0; small_count = 0; }      // (several barriers before their first use)
  if (threadIdx.x < 16) tform_lds[threadIdx.x] = st->T[threadIdx.x];
  if (LB && threadIdx.x == 16) { tform_lds[16] = st->motion_acc; tform_lds[17] = st->motion_eps; tform_lds[18] = st->motion_pred; }
#ifdef CILHIP_EXP_PHASE_CLOCKS
  unsigned long long tprev_ = wall_clock64();
#endif

  const GridDev& g = a.grid;
  const uint2 tile = tiles[vb];
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  const float smt[3] = {st->smt[0], st->smt[1], st->smt[2]};     // (scalar loads here: read in the tail they are vector loads with a round trip each)
  const MotionRef mref = {LB ? st->motion_acc : 0.0f, LB ? st->motion_eps : 0.0f};
  const float mstep = LB ? st->motion_pred : 0.0f;

  // the lane's queries: issue the loads first, they fly while the region's cell table is fetched
  float4 s4[TILE_QPT];
#pragma unroll
  for (int u = 0; u < TILE_QPT; ++u) {
    const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
    s4[u] = i < tile.y ? a.src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- 1. region of the tile: the cell range of its cube's image under the current transform (k_tile_boxes) ----
  const int* tb = tile_box + 8 * (size_t)vb;
  const int bx0 = tb[0], bx1 = tb[1], by0 = tb[2], by1 = tb[3], bz0 = tb[4], bz1 = tb[5];
  const int lox = max(bx0 - 1, 0), loy = max(by0 - 1, 0), loz = max(bz0 - 1, 0);
  const int hix = min(bx1 + 1, g.nx - 1), hiy = min(by1 + 1, g.ny - 1), hiz = min(bz1 + 1, g.nz - 1);
  const int RX = hix - lox + 1, RY = hiy - loy + 1, RZ = hiz - loz + 1;
  int rows = RY * RZ;               // (shrinks if the region's points exceed the LDS budget)
  const int W1 = RX + 1, E = rows * W1;
  // cells of the box whose whole 3x3x3 neighbourhood is inside the grid: the only ones the fast path takes
  const int fx0 = max(bx0, 1), fx1 = min(bx1, g.nx - 2), fy0 = max(by0, 1), fy1 = min(by1, g.ny - 2), fz0 = max(bz0, 1), fz1 = min(bz1, g.nz - 2);
  bool ok = (fx0 <= fx1) & (fy0 <= fy1) & (fz0 <= fz1) & (RY <= TILE_MAXSPAN) & (RZ <= TILE_MAXSPAN) & (E <= TILE_MAXE);   // block-uniform
  if (!ok) {
    // whole-tile fallback (the cube's image is outside the grid or too large for the LDS budget: the transform
    // moved far from the sort-time one): the clean-up pass searches this tile's queries in their sorted order
    defer_whole_tile<ACC>(a, vb);
    return;
  }

  if (threadIdx.x == 0) { geom_lds[0] = lox; geom_lds[1] = loy; geom_lds[2] = loz; geom_lds[3] = RY; geom_lds[4] = W1; geom_lds[5] = rows; }

  // ---- 2a. cell table of the region: rows x (RX+1) cell_start values, flat over the block.  Buffer loads:
  //          32-bit offsets (one shift per address) and out-of-range lanes simply read 0 ----
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)g.cell_start, 0, ((uint32_t)g.nx * (uint32_t)g.ny * (uint32_t)g.nz + 1u) * 4u, 0x00020000);
  {
    const uint32_t inv_w1 = (uint32_t)tb[6], inv_ry = (uint32_t)tb[7];   // e / W1 == (e * inv_w1) >> 20, r / RY == (r * inv_ry) >> 16 (k_tile_boxes)
    constexpr int TRIPS = (TILE_MAXE + TILE_THREADS - 1) / TILE_THREADS;
    const uint32_t rowstride = (uint32_t)g.nx, slab = (uint32_t)g.ny * (uint32_t)g.nx;
    const uint32_t gbase = ((uint32_t)loz * (uint32_t)g.ny + (uint32_t)loy) * (uint32_t)g.nx + (uint32_t)lox;
    uint32_t v[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      if (k * TILE_THREADS < E) {   // block-uniform
        const uint32_t e = (uint32_t)k * TILE_THREADS + threadIdx.x;
        // (24-bit multiplies: full rate, and every factor here is far below 2^24 -- e < 2^13, inv_w1 <= 2^17,
        //  rows < 2^9, inv_ry <= 2^16, grid dims <= 2^11 per axis)
        const uint32_t r = __umul24(e, inv_w1) >> 20;
        const uint32_t x = e - __umul24(r, (uint32_t)W1);
        const uint32_t zr = __umul24(r, inv_ry) >> 16;
        const uint32_t gi = gbase + __umul24(zr, slab) + __umul24(r - __umul24(zr, (uint32_t)RY), rowstride) + x;
        v[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_cs, e < (uint32_t)E ? gi * 4u : 0xFFFFFFFFu, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      if (k * TILE_THREADS < E) {
        const uint32_t e = (uint32_t)k * TILE_THREADS + threadIdx.x;
        if (e < (uint32_t)E) lcs[e] = v[k];
      }
    }
  }
  __syncthreads();
  PHASE_CLK(0);

  // ---- 2b. row lengths -> LDS offsets (one wave, each lane a block of consecutive rows) ----
  if (threadIdx.x < 64) {
    const int K = (rows + 63) >> 6;              // rows per lane, <= 8
    const int r0 = (int)threadIdx.x * K;
    uint32_t len[8], fst[8], tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + k;
      const bool v = (k < K) & (r < rows);
      fst[k] = v ? lcs[r * W1] : 0u;
      len[k] = v ? lcs[r * W1 + RX] - fst[k] : 0u;
      tot += len[k];
    }
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t tt = __shfl_up(incl, off, 64);
      if ((int)threadIdx.x >= off) incl += tt;
    }
    uint32_t run = incl - tot;                    // exclusive prefix of this lane's block
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = r0 + k;
      if ((k < K) & (r < rows)) {
        rowbase[r] = run;
        rowdelta[r] = fst[k] - run;               // global position - LDS index (mod 2^32)
        run += len[k];
      }
    }
    if (threadIdx.x == 63) rowbase[rows] = incl;
  }

  // ---- 2c. (all waves, overlapping the scan) the lane's queries: transform, current cell, octant block ----
  OctQuery oq[TILE_QPT];
  uint32_t flags = 0;   // per query u: bit u = active, bit 8+u = fast path (cell inside the tile's box, not in the grid's outer layer)
#pragma unroll
  for (int u = 0; u < TILE_QPT; ++u) {
    const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
    float qx, qy, qz;
    transform_point(T, s4[u].x, s4[u].y, s4[u].z, qx, qy, qz);
    const float BIG = 1.0e9f;
    const int cx = (int)floorf(fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG));
    const int cy = (int)floorf(fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG));
    const int cz = (int)floorf(fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG));
    const bool active = i < tile.y;
    // fast path: the query's octant block is staged (the region holds a halo cell only on the sides some query of the
    // tile can lean to, k_tile_boxes: every query is checked against what was actually staged)
    const bool fast = active & octant_prepare(g, qx, qy, qz, cx, cy, cz, lox, loy, loz, hix, hiy, hiz, RY, W1, oq[u]);
    flags |= (active ? (1u << u) : 0u) | (fast ? (1u << (8 + u)) : 0u);
  }
  __syncthreads();
  PHASE_CLK(1);
  uint32_t P = rowbase[rows];
  if (P > (uint32_t)TILE_CAP) {   // block-uniform: the region holds more points than the LDS budget
    // Drop z-slabs off the top of the region until it fits (rows are z-major, so rowbase[k * RY] = points of the first k
    // slabs); the queries whose 3x3x3 block needs a dropped slab leave the fast path and go to the clean-up pass one by
    // one.  Only a region that does not even fit three slabs sends the whole tile there.
    int rz = RZ;
    while (rz > 3 && rowbase[rz * RY] > (uint32_t)TILE_CAP) --rz;
    if (rowbase[rz * RY] > (uint32_t)TILE_CAP) {
      defer_whole_tile<ACC>(a, vb);
      return;
    }
    rows = rz * RY;
    P = rowbase[rows];
    const int fz1n = loz + rz - 2;   // last cell whose z+1 slab is still staged
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u)
      if ((int)floorf((oq[u].qz - g.oz) * g.inv_cell) > fz1n) flags &= ~(1u << (8 + u));
    if (threadIdx.x == 0) geom_lds[5] = rows;
  }
  // ---- 2d. stage the points: 16 lanes per row (every row is one contiguous run of the sorted target array),
  //          four rows in flight per lane; buffer loads (32-bit offsets, idle lanes read out of range = nothing) ----
  {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_pts = __builtin_amdgcn_make_buffer_rsrc((void*)g.pts, 0, g.n * 16u, 0x00020000);
    const uint32_t o0 = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    constexpr int GR = TILE_THREADS / 16;   // rows per batch step
#ifndef CILHIP_STAGE_ROWS
#define CILHIP_STAGE_ROWS 4
#endif
    constexpr int STAGE_ROWS = CILHIP_STAGE_ROWS;   // rows in flight per lane
    for (int rb = 0; rb < rows; rb += STAGE_ROWS * GR) {   // block-uniform rounds of STAGE_ROWS * GR rows
#pragma unroll
      for (int half = 0; half < 2; ++half) {   // points [0,16) of every row, then points [16,32)
        u32x4 v[STAGE_ROWS];
        uint32_t dst[STAGE_ROWS];
#pragma unroll
        for (int m = 0; m < STAGE_ROWS; ++m) {
          const int r = rb + grp + GR * m;
          const bool rv = r < rows;
          const uint32_t f = rv ? rowbase[r] : 0u, l = rv ? rowbase[r + 1] - f : 0u, d = rv ? rowdelta[r] : 0u;
          const uint32_t o = o0 + 16u * (uint32_t)half;
          const bool has = o < l;
          dst[m] = has ? f + o : NONE_U32;
          v[m] = __builtin_amdgcn_raw_buffer_load_b128(rs_pts, has ? (f + o + d) * 16u : 0xFFFFFFFFu, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < STAGE_ROWS; ++m)
          if (dst[m] != NONE_U32) {
            float* const w = lp + lds_slot_x(dst[m]);
            w[0] = __uint_as_float(v[m].x); w[2] = __uint_as_float(v[m].y); w[4] = __uint_as_float(v[m].z);
          }
      }
    }
    // rare leftovers: rows longer than 32 points
    for (int r = grp; r < rows; r += GR) {
      const uint32_t f = rowbase[r], l = rowbase[r + 1] - f, d = rowdelta[r];
      for (uint32_t o = o0 + 32u; o < l; o += 16) { const float4 q = g.pts[f + o + d]; float* const w = lp + lds_slot_x(f + o); w[0] = q.x; w[2] = q.y; w[4] = q.z; }
    }
    if (threadIdx.x < 30) lp[lds_slot_x(P + threadIdx.x / 3u) + 2u * (threadIdx.x % 3u)] = 1.0e30f;  // 10 pad records (d2 = inf): the unclamped reads end at most 4 slots behind the even slot at or after P
  }
  __syncthreads();
  PHASE_CLK(2);
  // ---- 3. per-lane exact search out of LDS ----
  // 3a: the octant block, every lane, straight-line.  Queries it does not prove are QUEUED in LDS (16-bit slot ids in
  // the unused tail of the point buffer) instead of being finished in place: finishing them in place costs a wave the
  // whole 3x3x3 search even when one of its lanes needs it.
  TileLds tl{lp, (uint32_t)reinterpret_cast<uintptr_t>(raw), (P + 1u) & ~1u, lcs, rowbase, rowdelta, lox, loy, loz, RY, W1, rows};      // (low 32 bits of a generic pointer into LDS = the LDS byte address)
  // (queue base and capacity are re-derived from the LDS row table where needed rather than kept in registers across
  //  the search: P = rowbase[rows], block-uniform)
  uint32_t mpos[TILE_QPT];   // per query: sorted-target position of the match (NONE: none / not settled here)
  uint32_t mbl = 0;          // (accumulating form, feature search) the matches' LDS indices, 16 bits each: the matched points are read from the staged tile
  uint32_t mkeys = 0;        // (accumulating form, LB) the settled queries' margin keys, 16 bits each
  uint32_t f6_pos[TILE_QPT];  // (feature search) the 3-D winners and the second smallest 3-D distances, until the lane's searches are done
  float f6_second[TILE_QPT];
#pragma unroll
  for (int u = 0; u < TILE_QPT; ++u) {
    const bool active = (flags >> u) & 1u, fast = (flags >> (8 + u)) & 1u;
    const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
    NN best;
    best.key = ((unsigned long long)__float_as_uint(a.max_sq) << 32);
    best.pos = NONE_U32;
    uint32_t bl = NONE_U32;
    bool defer = false, unproven = false, pending = false;
    float mkey = 0.0f;      // (LB, search-only form) the margin key of what this search settles
    uint32_t mq = 0;        // (LB, accumulating form) ... packed (margin_q15)
    if (active) {
      if (fast && FEAT6) {
        // searched by the 3-D distance now; settled after BOTH of the lane's searches, with the winners' normals gathered
        // in one round trip (below)
        (void)octant_search(g, tl, oq[u], a.max_sq, best, bl, &f6_second[u]);
        f6_pos[u] = best.pos;
        mbl |= (bl & 0xFFFFu) << (16 * u);
        pending = true;
      } else if (fast) {
        float second = INFINITY, gapb = 0.0f;
        unproven = !octant_search(g, tl, oq[u], a.max_sq, best, bl, &second, &gapb);
        if (LB && ACC == IM_NONE) mkey = margin_key(best.pos != NONE_U32, second, gapb, mref);
        if (LB && ACC != IM_NONE) mq = margin_q15(best.pos != NONE_U32, second, gapb, mref, g.inv_cell);
        if (LB) {
          // (the accumulating form hands its unproven queries to the clean-up pass, which keeps no bound; the search-only form
          //  counts them where its 3x3x3 pass settles them)
          const bool small = unproven ? (ACC != IM_NONE) : margin_is_small(best.pos != NONE_U32, second, gapb, __uint_as_float((uint32_t)(best.key >> 32)), a.max_sq, mstep);
          const unsigned long long ms = __ballot(small);
          if (ms != 0ull && (threadIdx.x & 63u) == 0) atomicAdd(&small_count, (uint32_t)__popcll(ms));
        }
      } else {
        // outside the tile's box or in the grid's outer layer (or beyond): nothing to find if the query is farther
        // from the grid than the radius, else the clean-up pass (generic search) takes it
        const float gx = axis_gap(oq[u].qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin);
        const float gy = axis_gap(oq[u].qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin);
        const float gz = axis_gap(oq[u].qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
        defer = (gx * gx + gy * gy + gz * gz) * KSHRINK < a.max_sq;
        const float ggap = __fsqrt_rn((gx * gx + gy * gy + gz * gz) * KSHRINK) * 0.999999f;      // every target point lies inside the grid
        if (LB && ACC == IM_NONE) mkey = margin_key(false, INFINITY, ggap, mref);
        if (LB && ACC != IM_NONE) mq = margin_q15(false, INFINITY, ggap, mref, g.inv_cell);
      }
    }
    if (ACC != IM_NONE || FEAT6) {   // (accumulating form / feature search: no second pass in the tile, see 3b; the unproven ones are only counted)
      const unsigned long long mu = __ballot(unproven);
      if (mu != 0ull && (threadIdx.x & 63u) == 0) atomicAdd(&queue_count, (uint32_t)__popcll(mu));
      if (unproven) { defer = true; unproven = false; }
    }
    const unsigned long long m = (ACC == IM_NONE && !FEAT6) ? __ballot(unproven) : 0ull;
    if (m) {   // one LDS atomic per wave
      const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&queue_count, (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      const uint32_t Pq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowbase[rows]);
      uint16_t* const queue = reinterpret_cast<uint16_t*>(raw + ((Pq + 12u) >> 1) * 24u);
      const uint32_t queue_cap = min((uint32_t)TILE_QUERIES, ((uint32_t)TILE_CAP - Pq) * 6u);
      if (unproven) {
        if (slot < queue_cap) {
          queue[slot] = (uint16_t)(u * TILE_THREADS + threadIdx.x);
        }          // (no room: not written; the tile then defers all its unproven queries, see below)
      }
    }
    if (active && !unproven && !defer && !pending && a.store_matches) {
      a.nn_pos[i] = best.pos;
      if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
    }
    if (LB && ACC == IM_NONE && active && !unproven && !defer) a.nn_lb[i] = mkey;
    if (LB && ACC != IM_NONE) mkeys |= ((active && !defer) ? mq : 0u) << (16 * u);      // (the record is written in the tail)
    mpos[u] = (unproven | defer) ? NONE_U32 : best.pos;
    if (ACC != IM_NONE) mbl |= (bl & 0xFFFFu) << (16 * u);      // (bl < TILE_CAP + 8 < 2^16; NONE's low bits are never used: mpos says so)
    flags |= (unproven ? (1u << (16 + u)) : 0u) | (defer ? (1u << (24 + u)) : 0u) | (pending ? (1u << (20 + u)) : 0u);
  }
  if (FEAT6) {
    // settle the pending queries: feature distance of the 3-D winner against the second smallest 3-D distance met
    float4 np[TILE_QPT], cp[TILE_QPT], rr[TILE_QPT];
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) {
      np[u] = cp[u] = rr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (((flags >> (20 + u)) & 1u) && f6_pos[u] != NONE_U32) {
        np[u] = target_features(a)[f6_pos[u]];
        if (a.feat.dst2 != nullptr) cp[u] = a.feat.dst2[f6_pos[u]];
        { const float* const w = lp + lds_slot_x((mbl >> (16 * u)) & 0xFFFFu); rr[u] = make_float4(w[0], w[2], w[4], 0.f); }
      }
    }
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) {
      const bool pend = (flags >> (20 + u)) & 1u;
      const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
      float dbest = a.max_sq;                       // min(d6 of the winner, radius): what the proof compares
      uint32_t pos = NONE_U32;
      bool ambiguous = false;
      if (pend && f6_pos[u] != NONE_U32) {
        Feat6 f;
        query_features(a, T, i, false, f);
        const float d6 = d6_pinned(oq[u].qx, oq[u].qy, oq[u].qz, f, make_float4(rr[u].x, rr[u].y, rr[u].z, 0.f), np[u], cp[u]);
        ambiguous = !(f6_second[u] > d6);           // another candidate's d6 (>= its d3 >= second) could be <= d6: not settled here
        if (d6 < a.max_sq) { dbest = d6; pos = f6_pos[u]; }
      }
      const float b = octant_bound(g, oq[u].qx, oq[u].qy, oq[u].qz) - g.margin;
      const bool unproven = pend && (ambiguous || !(b > 0.0f && dbest < b * b * KSHRINK));
      const unsigned long long mu = __ballot(unproven);
      if (mu != 0ull && (threadIdx.x & 63u) == 0) atomicAdd(&queue_count, (uint32_t)__popcll(mu));
      if (unproven) flags |= 1u << (24 + u);
      if (pend && !unproven && a.store_matches) {
        a.nn_pos[i] = pos;
        if (a.nn_d2) a.nn_d2[i] = dbest;
      }
    }
  }
  // (ACC) What the accumulation needs for the queries settled above is fetched NOW, before the barrier: the matched normal
  // (the one gather from HBM), the matched point out of the staged tile, and the lane's FIRST query again (the last one is
  // still in registers; the first is not kept alive through the second search) -- the loads fly while the slower waves of
  // the tile finish their searches (a barrier does not wait for outstanding loads).
  float4 p4t[TILE_QPT], n4t[TILE_QPT];
  float qt[TILE_QPT][3];
  if (ACC != IM_NONE) {
    static_assert(!FEAT6 || ACC == IM_NONE, "the feature search has no accumulating form");
    static_assert(TILE_QPT == 2, "the tail keeps the LAST query of a lane in registers and fetches the first one again");
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) {
      p4t[u] = n4t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      qt[u][0] = qt[u][1] = qt[u][2] = 0.0f;
      if (mpos[u] != NONE_U32) {
        if (FusedZ<ACC>::needs_normal) n4t[u] = g.nrm[mpos[u]];                 // the one gather from HBM
        const float* const w = lp + lds_slot_x((mbl >> (16 * u)) & 0xFFFFu);    // the matched point out of the staged tile
        p4t[u] = make_float4(w[0], w[2], w[4], 0.f);
        if (u == TILE_QPT - 1) {
          qt[u][0] = oq[u].qx; qt[u][1] = oq[u].qy; qt[u][2] = oq[u].qz;        // searched last: still in registers
        } else {
          uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
          asm volatile("" : "+v"(i));   // (a fresh load: do not keep the kernel-start copy of the query alive through the search)
          const float4 s4r = a.src[i];
          qt[u][0] = s4r.x; qt[u][1] = s4r.y; qt[u][2] = s4r.z;                   // (transformed after the barrier: using it here would wait for the load here)
        }
      }
    }
  }
  __syncthreads();
  // 3b (search-only form): the queued queries, densely packed over the lanes: the full 3x3x3 block in straight-line code.
  // The query is fetched and transformed again.  Results go through nn_pos (DEFER_MARK: not proven either) and the
  // query's home lane picks them up below.  The ACCUMULATING form has no 3b: what the octant block does not prove goes to
  // the clean-up pass, which accumulates what it settles -- in a converged registration that is nothing, and the
  // register-hungry 3x3x3 pass between the search and the accumulation would cost every tile its in-flight pair loads
  // (the allocator spills them around it); a source far from its sort-time cells is what the re-sort is for.
  if (threadIdx.x == 0) {   // what the octant block did not prove, for the host's choice of the next iteration's form
    const uint32_t cu = queue_count;
    if (cu != 0u) atomicAdd(a.unproven_cnt + (vb & 63u), cu);
    if (LB && ACC != IM_NONE) { const uint32_t cs = small_count; if (cs != 0u) atomicAdd(a.unproven_cnt + 64u + (vb & 63u), cs); }
  }
  uint32_t nqueued = (ACC == IM_NONE && !FEAT6) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)queue_count) : 0u;   // block-uniform
  if (ACC == IM_NONE && !FEAT6) {
    // A queue that cannot hold every unproven query of the tile (a region near the LDS budget AND a source far from its
    // sort-time cells): WHICH queries found room depends on the order the waves arrived in, so none of them is taken --
    // all unproven queries of the tile go to the clean-up pass (the set is then the same in every run).
    const uint32_t Pq0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowbase[rows]);
    if (nqueued > min((uint32_t)TILE_QUERIES, ((uint32_t)TILE_CAP - Pq0) * 6u)) {
#pragma unroll
      for (int u = 0; u < TILE_QPT; ++u) if ((flags >> (16 + u)) & 1u) flags = (flags & ~(1u << (16 + u))) | (1u << (24 + u));
      nqueued = 0;
    }
  }
  if (ACC == IM_NONE && !FEAT6 && nqueued != 0) {
    const uint32_t Pq = (uint32_t)__builtin_amdgcn_readfirstlane((int)rowbase[__builtin_amdgcn_readfirstlane(geom_lds[5])]);
    TileLds tq{lp, (uint32_t)reinterpret_cast<uintptr_t>(raw), (Pq + 1u) & ~1u, lcs, rowbase, rowdelta, __builtin_amdgcn_readfirstlane(geom_lds[0]), __builtin_amdgcn_readfirstlane(geom_lds[1]),
               __builtin_amdgcn_readfirstlane(geom_lds[2]), __builtin_amdgcn_readfirstlane(geom_lds[3]),
               __builtin_amdgcn_readfirstlane(geom_lds[4]), __builtin_amdgcn_readfirstlane(geom_lds[5])};
    const uint16_t* const queue = reinterpret_cast<const uint16_t*>(raw + ((Pq + 12u) >> 1) * 24u);
    const uint32_t nq = min(nqueued, min((uint32_t)TILE_QUERIES, ((uint32_t)TILE_CAP - Pq) * 6u));
    const int hx27 = tq.lox + tq.W1 - 2, hy27 = tq.loy + tq.RY - 1, hz27 = tq.loz + tq.rows / tq.RY - 1;   // last staged cell per axis
    if (threadIdx.x < nq) {   // wave-uniform except in the last wave
      float Tq[16];        // from LDS rather than kept live across the kernel
#pragma unroll
      for (int k = 0; k < 16; ++k) Tq[k] = tform_lds[k];
      for (uint32_t k = threadIdx.x; k < nq; k += TILE_THREADS) {
        const uint32_t i = tile.x + queue[k];
        const float4 sq = a.src[i];
        float qx, qy, qz;
        transform_point(Tq, sq.x, sq.y, sq.z, qx, qy, qz);
        const int cx = (int)floorf((qx - g.ox) * g.inv_cell), cy = (int)floorf((qy - g.oy) * g.inv_cell), cz = (int)floorf((qz - g.oz) * g.inv_cell);
        NN best;
        // (the region holds the octant blocks of the tile's queries, not necessarily all of this query's 3x3x3 block)
        const bool in27 = (cx - 1 >= tq.lox) & (cx + 1 <= hx27) & (cy - 1 >= tq.loy) & (cy + 1 <= hy27) & (cz - 1 >= tq.loz) & (cz + 1 <= hz27);
        float second = INFINITY, gapb = 0.0f;
        const bool proven = in27 && block27_search(g, tq, qx, qy, qz, cx, cy, cz, a.max_sq, best, &second, &gapb);
        a.nn_pos[i] = proven ? best.pos : DEFER_MARK;
        if (proven && a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
        if (LB && proven) a.nn_lb[i] = margin_key(best.pos != NONE_U32, second, gapb, MotionRef{tform_lds[16], tform_lds[17]});
        if (LB && (!proven || margin_is_small(best.pos != NONE_U32, second, gapb, __uint_as_float((uint32_t)(best.key >> 32)), a.max_sq, tform_lds[18])))
          atomicAdd(&small_count, 1u);
      }
    }
    __syncthreads();
  }
  if (LB && ACC == IM_NONE && threadIdx.x == 0) {      // (after the barriers that order every wave's counts)
    const uint32_t cs = small_count;
    if (cs != 0u) atomicAdd(a.unproven_cnt + 64u + (vb & 63u), cs);
  }
  // ---- 4. home lanes: results of their queued queries; the deferred ones are published as one mask word per wave and
  //         query slot (bit = lane): the clean-up pass walks the masks in a fixed order ----
#pragma unroll
  for (int u = 0; u < TILE_QPT; ++u) {
    bool defer = (flags >> (24 + u)) & 1u;
    if (nqueued != 0 && ((flags >> (16 + u)) & 1u)) {
      const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
      // (a store by another wave of this workgroup, ordered by the barrier above: workgroup scope is all it takes)
      const uint32_t v = __hip_atomic_load(a.nn_pos + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (v == DEFER_MARK) defer = true; else mpos[u] = v;
    }
    const unsigned long long dm = __ballot(defer);
    if ((threadIdx.x & 63u) == 0) {
      a.defer_mask[(size_t)vb * (2 * TILE_WAVES) + u * TILE_WAVES + (threadIdx.x >> 6)] = dm;
      // tells the clean-up pass that it has anything to do at all.  Read first: far from convergence nearly every wave defers
      // something, and 10^5 atomics on one address serialise in L2 (measured: 1.6 ms in one search); all writers store 1.
      if (dm != 0ull && __hip_atomic_load(a.defer_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        __hip_atomic_store(a.defer_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  PHASE_CLK(3);
  if (ACC == IM_NONE) return;

  // ---- 5. accumulation inside the tile (ACC != IM_NONE; first Gauss-Newton step: the inner transform is the identity).
  // Every sum the estimators need is an entry of  Z = sum_i z_i z_i^T  for a per-correspondence vector z of f32 terms
  // (fused_z): a rank update with K = number of correspondences -- v_mfma_f64_16x16x4_f64 work, the one place on the
  // path where the matrix cores fit the arithmetic contract (products of f32 terms are exact in f64, sums in f64).
  // Nothing but the match position is carried through the search in registers: the query was fetched again and the
  // matched point and its normal gathered right after the search (above) -- all of it touched by this tile a moment ago.
  // From here on nobody reads the staged points or the cell table: the point buffer becomes per-wave scratch.
  {
    constexpr int NC = FusedZ<ACC>::NC;
    constexpr bool DUAL = NC <= 8;        // two groups of 4 correspondences per instruction: rows/cols 0-7 and 8-15
    float z[TILE_QPT][16];
#pragma unroll
    for (int u = 0; u < TILE_QPT - 1; ++u) {
      const float sx = qt[u][0], sy_ = qt[u][1], sz_ = qt[u][2];
      transform_point(T, sx, sy_, sz_, qt[u][0], qt[u][1], qt[u][2]);
    }
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) fused_z<ACC>(mpos[u] != NONE_U32, qt[u][0], qt[u][1], qt[u][2], p4t[u], n4t[u], a.dst_mean, smt, z[u]);
    if (LB) {
      // the match records of the queries this tile settled (the deferred ones: the clean-up pass)
#pragma unroll
      for (int u = 0; u < TILE_QPT; ++u)
        if (((flags >> u) & 1u) && !((flags >> (24 + u)) & 1u)) {
          const uint32_t i = tile.x + u * TILE_THREADS + threadIdx.x;
          a.warm_rec[i] = make_float4(p4t[u].x, p4t[u].y, p4t[u].z, margin_from_q15((mkeys >> (16 * u)) & 0xFFFFu, g.cell, mref));
          if (FusedZ<ACC>::needs_normal && mpos[u] != NONE_U32) a.warm_rec_n[i] = F3{n4t[u].x, n4t[u].y, n4t[u].z};
        }
    }
    PHASE_CLK(4);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    float* const zb = reinterpret_cast<float*>(raw) + wave * (FUSED_WAVE_BYTES / 4);
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < TILE_QPT; ++u) {
      if (DUAL) {
        // [64 correspondences][8 floats]; the second half of the wave 16 floats further (so that the two groups an
        // instruction reads sit on different banks)
        float4* w4 = reinterpret_cast<float4*>(zb + lane * 8 + (lane >= 32 ? 16 : 0));
        w4[0] = make_float4(z[u][0], z[u][1], z[u][2], z[u][3]);
        w4[1] = make_float4(z[u][4], z[u][5], z[u][6], z[u][7]);
      } else {
        float2* w2 = reinterpret_cast<float2*>(zb + lane * NC);
#pragma unroll
        for (int c = 0; c < NC / 2; ++c) w2[c] = make_float2(z[u][2 * c], z[u][2 * c + 1]);
      }
      __builtin_amdgcn_wave_barrier();    // (DS operations of one wave execute in order: the reads below see the writes)
      if (DUAL) {
        const int comp = lane & 7, half = (lane >> 3) & 1, k4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int qi = half * 32 + 4 * j + k4;
          const double x = (double)zb[qi * 8 + half * 16 + comp];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
      } else {
        const int comp = lane & 15, k4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float f = zb[(4 * j + k4) * NC + (comp < NC ? comp : 0)];
          const double x = comp < NC ? (double)f : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // D[(lane >> 4) + 4 r][lane & 15] = acc[r]  ->  this wave's 16x16 tile in its scratch, then one fixed-order sum
    double* const db = reinterpret_cast<double*>(raw + wave * FUSED_WAVE_BYTES);
#pragma unroll
    for (int r = 0; r < 4; ++r) db[r * 64 + lane] = acc[r];
    __syncthreads();
    PHASE_CLK(5);
    if (threadIdx.x < SUMS_MAX) {
      int i1, j1, i2, j2;
      const bool used = FusedZ<ACC>::slot_terms((int)threadIdx.x, i1, j1, i2, j2);
      double v1 = 0.0, v2 = 0.0;
      if (used) {
        const int e1 = (i1 >> 2) * 64 + 16 * (i1 & 3) + j1, e1b = ((i1 + 8) >> 2) * 64 + 16 * ((i1 + 8) & 3) + j1 + 8;
        const int e2 = i2 >= 0 ? (i2 >> 2) * 64 + 16 * (i2 & 3) + j2 : 0, e2b = i2 >= 0 ? ((i2 + 8) >> 2) * 64 + 16 * ((i2 + 8) & 3) + j2 + 8 : 0;
        for (int w = 0; w < TILE_WAVES; ++w) {
          const double* dw = reinterpret_cast<const double*>(raw + w * FUSED_WAVE_BYTES);
          v1 += dw[e1];
          if (DUAL) v1 += dw[e1b];
          if (i2 >= 0) { v2 += dw[e2]; if (DUAL) v2 += dw[e2b]; }
        }
      }
      a.tile_partials[(size_t)vb * SUMS_MAX + threadIdx.x] = v1 - v2;
    }
    PHASE_CLK(6);
  }
}

// Clean-up pass of the tiled search: the (few) queries the LDS tile could not settle -- sparse data or a radius beyond
// the 3x3x3 block, queries outside the grid, tiles whose region exceeded the LDS budget -- run the generic exact search
// out of global memory.  The tiles publish them as bit masks (one 64-bit word per wave and query slot); a block lists
// the queries of 64 words at a time and deals the list to its lanes in a fixed order, so that with ACC != IM_NONE what a
// lane accumulates -- and with it every partial sum -- is the same in every run.  Short lists: TODO_GROUP lanes per query;
// long lists (whole deferred tiles): one lane per query.  With ACC the block also folds its share of the tiles' partial rows.
constexpr int TODO_GROUP = 8;   // lanes per deferred query
#ifndef CILHIP_FEAT6_GROUP
#define CILHIP_FEAT6_GROUP 1
#endif
constexpr int FEAT6_GROUP = CILHIP_FEAT6_GROUP;   // lanes per query of the feature search: every query takes this path, so one lane each fills the chip best
                                                  // (10M<->10M iteration: 8 lanes 1.69 ms, 4: 1.22, 2: 1.07, 1: 0.93)
template <int ACC, bool FEAT6 = false>
__global__ __launch_bounds__(ITER_THREADS) void k_search_deferred(IterArgs a, const uint2* __restrict__ tiles, uint32_t ntiles) {
  using TR = AccTraits<ACC>;
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  if (*a.defer_flag == 0u) {     // no tile deferred anything (the usual case near convergence): a zero row, nothing else
    if (ACC != IM_NONE && threadIdx.x < SUMS_MAX) a.partials[(size_t)blockIdx.x * SUMS_MAX + threadIdx.x] = 0.0;
    return;
  }
  __shared__ uint2 worklist[LIST_CAP * ITER_THREADS];
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  const float iL[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, it[3] = {0.f, 0.f, 0.f};   // first Gauss-Newton step
  const float smt[3] = {st->smt[0], st->smt[1], st->smt[2]}, dmean[3] = {a.dst_mean[0], a.dst_mean[1], a.dst_mean[2]};
  double accA[TR::NA];
  double accB[TR::NB > 0 ? TR::NB : 1];
#pragma unroll
  for (int i = 0; i < TR::NA; ++i) accA[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (TR::NB > 0 ? TR::NB : 1); ++i) accB[i] = 0.0;

  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  const uint32_t W = ntiles * (2 * TILE_WAVES), nchunks = (W + 63u) >> 6;
  // (margin keys / match records of the warm-started iterations: the generic search proves its result but keeps no bound on
  //  the other points -- "no bound known"; the warm kernel searches such a query itself and then has one)
  const float key_unknown_has = 0.0f, key_unknown_none = MARGIN_NONE_NO_MATCH;
  auto finish = [&](uint32_t i, float qx, float qy, float qz, NN& best, const Feat6* f6 = nullptr) {
    // (option "tie_rule": a query whose nearest distance was met on two points -- the tiles send theirs here -- takes the reference's pick)
    if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)
      best.pos = tie_settle<FEAT6>(a.grid, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)), f6);
    a.nn_pos[i] = best.pos;
    if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
    if (ACC == IM_NONE && a.nn_lb) a.nn_lb[i] = best.pos != NONE_U32 ? key_unknown_has : key_unknown_none;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), nv = p;
    if (ACC != IM_NONE && best.pos != NONE_U32) {
      p = a.grid.pts[best.pos];
      if (TR::plane) nv = a.grid.nrm[best.pos];
      accumulate_pair<ACC>(accA, accB, T, iL, it, smt, dmean, false, true, qx, qy, qz, best.pos, p, nv, nv);
    }
    if (ACC != IM_NONE && a.warm_rec) {
      a.warm_rec[i] = make_float4(p.x, p.y, p.z, best.pos != NONE_U32 ? key_unknown_has : key_unknown_none);
      if (TR::plane) a.warm_rec_n[i] = F3{nv.x, nv.y, nv.z};
    }
  };
  // One chunk = 64 mask words, STRIDED through the mask array (slot j of chunk c = word j * nchunks + c): the 32 words of a
  // tile land in 32 different chunks, so a few tiles that defer a slab of queries each (an over-budget region) are spread
  // over as many blocks.  The block lists the chunk's deferred queries in LDS (ascending slot, bit) and deals the LIST to
  // its lanes -- a static assignment: every lane accumulates the same queries in the same order in every run.
  __shared__ unsigned long long words[64];
  __shared__ uint32_t wpre[65];
  __shared__ uint16_t entries[64 * 64];           // (word << 6) | bit
  __shared__ uint32_t block_has_work;
  // most blocks have nothing to search (a converged registration defers a few hundred queries in all): find out with
  // all mask loads of the block in flight at once
  __shared__ unsigned char chunk_flag[64];      // the block's k-th chunk holds a deferred query (k < 64; later ones are looked at anyway)
  if (threadIdx.x < 64) {
    unsigned long long any = 0ull;
    uint32_t kk = 0;
    for (uint32_t c0 = blockIdx.x; c0 < nchunks; c0 += 8u * gridDim.x, kk += 8) {
      unsigned long long mw[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint64_t cc = (uint64_t)c0 + (uint64_t)k * gridDim.x;
        const uint64_t widx = (uint64_t)threadIdx.x * nchunks + cc;
        mw[k] = (cc < nchunks && widx < W) ? a.defer_mask[widx] : 0ull;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned long long bk = __ballot(mw[k] != 0ull);
        any |= bk;
        if (threadIdx.x == 0 && kk + k < 64) chunk_flag[kk + k] = bk != 0ull ? 1 : 0;
      }
    }
    if (threadIdx.x == 0) block_has_work = any != 0ull ? 1u : 0u;
  }
  __syncthreads();
  const bool has_work = block_has_work != 0u;     // block-uniform
  uint32_t kc = 0;
  for (uint32_t chunk = blockIdx.x; has_work && chunk < nchunks; chunk += gridDim.x, ++kc) {      // block-uniform
    if (kc < 64 && !chunk_flag[kc]) continue;
    if (threadIdx.x < 64) {
      const uint32_t widx = threadIdx.x * nchunks + chunk;
      const unsigned long long mw = widx < W ? a.defer_mask[widx] : 0ull;
      words[threadIdx.x] = mw;
      uint32_t incl = (uint32_t)__popcll(mw);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if ((int)threadIdx.x >= off) incl += t;
      }
      wpre[threadIdx.x + 1] = incl;
      if (threadIdx.x == 0) wpre[0] = 0;
    }
    __syncthreads();
    const uint32_t total = wpre[64];
    if (total != 0) {
      if (threadIdx.x < 64) {
        unsigned long long mw = words[threadIdx.x];
        uint32_t o = wpre[threadIdx.x];
        while (mw) { entries[o++] = (uint16_t)((threadIdx.x << 6) | (uint32_t)(__ffsll((long long)mw) - 1)); mw &= mw - 1; }
      }
      __syncthreads();
      auto query_of = [&](uint32_t e, uint2& tile) -> uint32_t {
        const uint32_t ent = entries[e], wi = (ent >> 6) * nchunks + chunk;
        tile = tiles[wi / (2 * TILE_WAVES)];
        return tile.x + ((wi / TILE_WAVES) & 1u) * TILE_THREADS + (wi % TILE_WAVES) * 64u + (ent & 63u);
      };
      if (!FEAT6 && total > ITER_THREADS / TODO_GROUP) {
        // more queries than lane groups (an over-budget tile's dropped slab, whole deferred tiles, a source far from its
        // sort-time cells): one lane per query -- one trip per 256 queries instead of one per 32
        for (uint32_t e = threadIdx.x; e < total; e += ITER_THREADS) {
          uint2 tile;
          const uint32_t i = query_of(e, tile);
          if (i < tile.y) {
            const float4 s4 = a.src[i];
            float qx, qy, qz;
            transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
            NN best;
            nn_search(a.grid, qx, qy, qz, a.max_sq, best, worklist + threadIdx.x);
            finish(i, qx, qy, qz, best);
          }
        }
      } else {
        // a handful of queries: TODO_GROUP lanes each (shorter dependent chains per query)
        const uint32_t grp = threadIdx.x / TODO_GROUP;
        const int sub = (int)(threadIdx.x % TODO_GROUP);
        for (uint32_t e = grp; e < total; e += ITER_THREADS / TODO_GROUP) {      // (uniform within a group)
          uint2 tile;
          const uint32_t i = query_of(e, tile);
          if (i < tile.y) {
            const float4 s4 = a.src[i];
            float qx, qy, qz;
            transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
            NN best;
            if (FEAT6) {
              Feat6 f;
              query_features(a, T, i, true, f);
              nn_search_group<TODO_GROUP, true>(a.grid, qx, qy, qz, a.max_sq, sub, 1, best, &f);
              if (sub == 0) finish(i, qx, qy, qz, best, &f);
            } else {
              nn_search_group<TODO_GROUP>(a.grid, qx, qy, qz, a.max_sq, sub, 1, best);
              if (sub == 0) finish(i, qx, qy, qz, best);
            }
          }
        }
      }
    }
    __syncthreads();    // (the lists are rewritten by the next chunk)
  }
  if (ACC != IM_NONE) {
    __shared__ double sh[ITER_WAVES][SUMS_MAX];
    if (has_work) {         // (block-uniform; a block that searched nothing contributes exact zeros, as the sums below would)
      if (threadIdx.x < SUMS_MAX) {
#pragma unroll
        for (int w = 0; w < ITER_WAVES; ++w) sh[w][threadIdx.x] = 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < TR::NA; ++k) {
        const double v = wave_sum(accA[k]);
        if (lane == 0) sh[wave][k] = v;
      }
#pragma unroll
      for (int k = 0; k < TR::NB; ++k) {
        const double v = wave_sum(accB[k]);
        if (lane == 0) sh[wave][28 + k] = v;
      }
      __syncthreads();
    }
    if (threadIdx.x < SUMS_MAX)
      a.partials[(size_t)blockIdx.x * SUMS_MAX + threadIdx.x] =
          has_work ? (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]) : 0.0;
  }
}

// blocks of the clean-up pass (64 mask words = two tiles per chunk; at most 4096 blocks, then several chunks each -- a block
// walks its chunks one after the other, each a chain of dependent memory trips: 1024 -> 3072 blocks: 0.438 -> 0.427 ms per
// search of an independently sampled 10M source)
static uint32_t deferred_blocks(uint32_t ntiles) {
  const uint32_t nchunks = (ntiles * (2u * TILE_WAVES) + 63u) >> 6;
  static const uint32_t cap = [] { const char* e = getenv("CILHIP_EXP_DEFER_BLOCKS"); return e ? (uint32_t)atoi(e) : 4096u; }();
  return nchunks < 1u ? 1u : (nchunks > cap ? cap : nchunks);
}
// rows of partial sums the tiled path with in-tile accumulation leaves in a.tile_partials: one per tile, then one per
// block of the clean-up pass (a.partials = a.tile_partials + ntiles rows)
int tiled_partial_rows(uint32_t ntiles) { return (int)(ntiles + deferred_blocks(ntiles)); }

template <int ACC>
static void launch_search_tiled_m(const IterArgs& a, const uint2* tiles, const int* tile_box, uint32_t ntiles, hipStream_t s, hipEvent_t ev_stop) {
  const uint32_t nb = ((ntiles + 7u) >> 3) << 3;
  // (margin keys / match records for the warm-started iterations: the LB variants -- the search-only form when a.nn_lb is set, the
  //  accumulating form when a.warm_rec is)
  if (ACC == IM_NONE ? a.nn_lb != nullptr : a.warm_rec != nullptr)
    hipLaunchKernelGGL((k_search_tiled<ACC, false, true>), dim3(nb), dim3(TILE_THREADS), 0, s, a, tiles, tile_box, ntiles);
  else hipLaunchKernelGGL((k_search_tiled<ACC>), dim3(nb), dim3(TILE_THREADS), 0, s, a, tiles, tile_box, ntiles);
  launch_ev((k_search_deferred<ACC>), dim3(deferred_blocks(ntiles)), dim3(ITER_THREADS), s, (hipEvent_t) nullptr, ev_stop, a, tiles, ntiles);
}

// acc_metric: IM_NONE = search only (matches stored); IM_KABSCH / IM_PLANE / IM_POINT / IM_BOTH = search + accumulation of
// the first Gauss-Newton step's sums in one pass (a.partials[0 .. tiled_partial_rows) rows afterwards).
void launch_search_tiled(const IterArgs& a, int acc_metric, const uint2* tiles, const float4* tile_center, int* tile_box, uint32_t ntiles, hipStream_t s) {
  const hipEvent_t ev_start = g_ev_start, ev_stop = g_ev_stop;      // (armed by set_launch_events: consumed here)
  g_ev_start = g_ev_stop = nullptr;
  if (ntiles == 0) { if (ev_start) (void)hipEventRecord(ev_start, s); if (ev_stop) (void)hipEventRecord(ev_stop, s); return; }
  launch_ev(k_tile_boxes, dim3((ntiles + 255) / 256), dim3(256), s, ev_start, (hipEvent_t) nullptr, make_box_args(a, tile_center, tile_box, ntiles, acc_metric != IM_NONE), a.state);
  switch (acc_metric) {
    case IM_KABSCH: launch_search_tiled_m<IM_KABSCH>(a, tiles, tile_box, ntiles, s, ev_stop); break;
    case IM_PLANE: launch_search_tiled_m<IM_PLANE>(a, tiles, tile_box, ntiles, s, ev_stop); break;
    case IM_POINT: launch_search_tiled_m<IM_POINT>(a, tiles, tile_box, ntiles, s, ev_stop); break;
    case IM_BOTH: launch_search_tiled_m<IM_BOTH>(a, tiles, tile_box, ntiles, s, ev_stop); break;
    default: launch_search_tiled_m<IM_NONE>(a, tiles, tile_box, ntiles, s, ev_stop); break;
  }
}

// the tiled form of the 6-D point+normal feature search (matches stored; SECOND_TO_FIRST, rigid transforms)
void launch_search_tiled_feat6(const IterArgs& a, const uint2* tiles, const float4* tile_center, int* tile_box, uint32_t ntiles, hipStream_t s) {
  if (ntiles == 0) return;
  hipLaunchKernelGGL(k_tile_boxes, dim3((ntiles + 255) / 256), dim3(256), 0, s, make_box_args(a, tile_center, tile_box, ntiles, true), a.state);
  const uint32_t nb = ((ntiles + 7u) >> 3) << 3;
  hipLaunchKernelGGL((k_search_tiled<IM_NONE, true>), dim3(nb), dim3(TILE_THREADS), 0, s, a, tiles, (const int*)tile_box, ntiles);
  hipLaunchKernelGGL((k_search_deferred<IM_NONE, true>), dim3(deferred_blocks(ntiles)), dim3(ITER_THREADS), 0, s, a, tiles, ntiles);
}

// deferred queries / wholly deferred tiles of the last tiled search (introspection: tests, dev tools)
__global__ void k_count_deferred(const unsigned long long* __restrict__ mask, uint32_t ntiles, uint32_t* out) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
    uint32_t bits = 0;
    bool all = true;
    for (int k = 0; k < 2 * TILE_WAVES; ++k) { const unsigned long long w = mask[(size_t)t * (2 * TILE_WAVES) + k]; bits += (uint32_t)__popcll(w); all &= (w == ~0ull); }
    if (all) atomicAdd(out + 1, 1u); else if (bits) atomicAdd(out, bits);
  }
}
void launch_count_deferred(const unsigned long long* mask, uint32_t ntiles, uint32_t* out2, hipStream_t s) {
  (void)hipMemsetAsync(out2, 0, 2 * sizeof(uint32_t), s);
  if (ntiles) hipLaunchKernelGGL(k_count_deferred, dim3((ntiles + 255) / 256), dim3(256), 0, s, mask, ntiles, out2);
}

// Correspondence search over 6-D point+normal features (SECOND_TO_FIRST): FEAT6_GROUP lane(s) per query, the generic exact
// search out of global memory with the feature distance.  Rigid transforms only (:104-111: the normal part is L * (w n)).
__global__ __launch_bounds__(ITER_THREADS) void k_search_feat6(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  const int sub = threadIdx.x & (FEAT6_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * ITER_THREADS + threadIdx.x) / FEAT6_GROUP;
  if (gid >= a.ns) return;      // (whole groups leave together)
  const uint32_t i = (uint32_t)gid;
  const float4 s4 = a.src[i];
  float qx, qy, qz;
  transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
  Feat6 f;
  query_features(a, T, i, true, f);
  NN best;
  nn_search_group<FEAT6_GROUP, true>(a.grid, qx, qy, qz, a.max_sq, sub, 1, best, &f);
  if (sub == 0) {
    // (option "tie_rule": exactly equal feature distances take the pick of the reference's DIM = 6 / 9 tree)
    if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)
      best.pos = tie_settle<true>(a.grid, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)), &f);
    a.nn_pos[i] = best.pos;
    if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
  }
}

// The exact search with SEVERAL lanes per query (small clouds, sources far from alignment: one lane per query leaves the chip idle
// behind chains of dependent trips -- the reference's 120k-point sensor frames are 1 900 waves for 1 024 SIMDs, a tenth of the lanes
// walking every shell inside the radius): G adjacent lanes share a query, the rows of the block around its cell are dealt to them,
// the minimum key goes round the group (nn_search_group: the clean-up pass's search), the block grows straight to the size the
// best found so far needs.  Same keys, same tie rule; no margin key (the generic search keeps no bound on the other points).
template <int G>
__global__ __launch_bounds__(ITER_THREADS) void k_search_group(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  const int sub = threadIdx.x & (G - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * ITER_THREADS + threadIdx.x) / G;
  if (gid >= a.ns) return;      // (whole groups leave together)
  const uint32_t i = (uint32_t)gid;
  const float4 s4 = a.src[i];
  float qx, qy, qz;
  transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
  NN best;
  best.key = ((unsigned long long)__float_as_uint(a.max_sq) << 32);
  best.pos = NONE_U32;
  if (a.warm_pos != nullptr) {      // the previous iteration's match bounds the search (every lane of the group reads the same record)
    const uint32_t w = a.warm_pos[i];
    if (w != NONE_U32) {
      const float4 pw = a.grid.pts[w];
      const float e = d2_pinned(qx, qy, qz, pw.x, pw.y, pw.z);
      if (e < a.max_sq) { best.key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(pw.w); best.pos = w; }
    }
  }
  nn_search_group<G, false, true>(a.grid, qx, qy, qz, a.max_sq, sub, 1, best);
  if (sub == 0) {
    if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)
      best.pos = tie_settle(a.grid, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)));
    a.nn_pos[i] = best.pos;
    if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32));
    if (a.nn_lb) a.nn_lb[i] = best.pos != NONE_U32 ? 0.0f : MARGIN_NONE_NO_MATCH;      // (no bound known)
  }
}
void launch_search_group(const IterArgs& a, int lanes, hipStream_t s) {
  if (a.ns == 0) return;
  const uint64_t threads = (uint64_t)a.ns * (uint64_t)lanes;
  const dim3 grid((unsigned)((threads + ITER_THREADS - 1) / ITER_THREADS)), block(ITER_THREADS);
  if (lanes == 64) hipLaunchKernelGGL((k_search_group<64>), grid, block, 0, s, a);
  else if (lanes == 32) hipLaunchKernelGGL((k_search_group<32>), grid, block, 0, s, a);
  else if (lanes == 16) hipLaunchKernelGGL((k_search_group<16>), grid, block, 0, s, a);
  else if (lanes == 4) hipLaunchKernelGGL((k_search_group<4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((k_search_group<8>), grid, block, 0, s, a);
}

void launch_search_feat6(const IterArgs& a, hipStream_t s) {
  if (a.ns == 0) return;
  const uint64_t lanes = (uint64_t)a.ns * FEAT6_GROUP;
  hipLaunchKernelGGL(k_search_feat6, dim3((unsigned)((lanes + ITER_THREADS - 1) / ITER_THREADS)), dim3(ITER_THREADS), 0, s, a);
}

// The fused iteration kernel.  METRIC: what to accumulate; SEARCH: run the grid search (else reuse the
// stored matches: Gauss-Newton steps >= 1); STORE: keep (pos,d2) per query for later steps / the host.
// (the search-only instantiation lives on its four waves per SIMD -- measured: three cost the far-from-alignment regimes a fifth --: held to 128 registers)
template <int METRIC, bool SEARCH, bool STORE>
__global__ __launch_bounds__(ITER_THREADS, (METRIC == IM_NONE && SEARCH) ? 4 : 1) void k_iter(IterArgs a) {
  using TR = AccTraits<METRIC>;
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  if (a.skip_if_inner_done && st->inner_done) return;

  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = st->T[i];
  float iL[9], it[3], smt[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) iL[i] = st->innerL[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { it[i] = st->innert[i]; smt[i] = st->smt[i]; }
  float dmean[3] = {a.dst_mean[0], a.dst_mean[1], a.dst_mean[2]};
  if (a.no_centering) { smt[0] = smt[1] = smt[2] = 0.0f; dmean[0] = dmean[1] = dmean[2] = 0.0f; }

  __shared__ uint2 worklist[SEARCH ? LIST_CAP * ITER_THREADS : 1];
  uint2* lst = worklist + (SEARCH ? threadIdx.x : 0);

  double accA[TR::NA];
  double accB[TR::NB > 0 ? TR::NB : 1];
#pragma unroll
  for (int i = 0; i < TR::NA; ++i) accA[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (TR::NB > 0 ? TR::NB : 1); ++i) accB[i] = 0.0;

  // XCD-aware virtual block id (gridDim.x is a multiple of 8)
  const uint32_t nb = gridDim.x;
  // The streaming accumulation pass (!SEARCH) walks each XCD's eighth BACKWARDS: the search pass that ran just before
  // walked it forwards, so the queries / matched points it touched last are the ones still in the XCD's L2 and in the
  // 256 MB Infinity Cache -- read those first, before this pass's own traffic evicts them.
#ifndef CILHIP_ACC_REVERSE
#define CILHIP_ACC_REVERSE 1
#endif
  const uint32_t slot = (!SEARCH && CILHIP_ACC_REVERSE) ? ((nb >> 3) - 1u - (blockIdx.x >> 3)) : (blockIdx.x >> 3);
  const uint32_t vb = (blockIdx.x & 7u) * (nb >> 3) + slot;
  const uint32_t chunk = (((a.ns + nb - 1) / nb) + 63u) & ~63u;
  const uint64_t beg64 = (uint64_t)vb * chunk;
  const uint32_t beg = beg64 < a.ns ? (uint32_t)beg64 : a.ns;
  const uint32_t end = (beg64 + chunk < a.ns) ? (uint32_t)(beg64 + chunk) : a.ns;

  // the accumulation of one matched pair (q = T*s already formed); shared by the loops below
  // (value: the correspondence's search distance, read by the weight evaluators only)
  // (idx: the correspondence's position in the stream -- what a caller's own evaluators' weight tables are indexed by)
  auto accumulate = [&](float value, uint32_t idx, float qx, float qy, float qz, uint32_t pos, const float4 p, const float4 nvp, const float4 snp) {
    float wq = 1.0f, wp = 1.0f;
    if (a.cw.enabled) {
      if (a.cw.point_table != nullptr) {
        const uint32_t ii = pos != NONE_U32 ? idx : 0u;
        wq = __fmul_rn(a.cw.w_p2p, a.cw.point_table[ii]); wp = __fmul_rn(a.cw.w_p2pl, a.cw.plane_table[ii]);
      } else {
        pair_weights(a.cw, value, wq, wp);
      }
    }
    accumulate_pair<METRIC>(accA, accB, T, iL, it, smt, dmean, a.src_nrm != nullptr, a.grid.nrm != nullptr, qx, qy, qz, pos, p, nvp, snp, wq, wp);
  };

  if (!SEARCH) {
    // Streaming pass over the stored matches, TWO elements per lane per trip and the next trip's source points /
    // match indices requested before the current gathers (matched point, normal) are consumed: every lane keeps
    // 4 coalesced loads + 4 gathers in flight.  Per-lane accumulation order is unchanged (i, i+T, i+2T, ...).
    // (k_warm's three rules, DESIGN.md section 5: every load UNCONDITIONAL -- indices clamped into the chunk, positions clamped into
    //  the target, the results masked afterwards: a load under a divergent branch "may not have been issued" for the compiler's
    //  in-order vmcnt bookkeeping and its wait then drains the younger prefetch too --; the gathers of a trip leave before the next
    //  trip's prefetch, so the wait for them leaves the prefetch in flight)
    uint32_t i0 = beg + threadIdx.x;
    const uint32_t last = end > beg ? end - 1u : 0u;
    const bool nrm_a = TR::plane || (TR::affine && a.grid.nrm != nullptr);      // (uniform)
    float4 sa = a.src[min(i0, last)], sb = a.src[min(i0 + ITER_THREADS, last)];
    uint32_t pa = a.nn_pos[min(i0, last)], pb = a.nn_pos[min(i0 + ITER_THREADS, last)];
    if (!(i0 < end)) pa = NONE_U32;
    if (!(i0 + ITER_THREADS < end)) pb = NONE_U32;
    while (i0 < end) {
      const float4 s4a = sa, s4b = sb;
      const uint32_t posa = pa, posb = pb;
      const uint32_t ia = i0, ib = i0 + ITER_THREADS;
      float4 p_a = make_float4(0.f, 0.f, 0.f, 0.f), nv_a = p_a, sn_a = p_a, p_b = p_a, nv_b = p_a, sn_b = p_a;
      if (METRIC != IM_NONE) {
        const uint32_t ga = posa != NONE_U32 ? posa : 0u, gb = posb != NONE_U32 ? posb : 0u;
        __builtin_amdgcn_sched_barrier(0);
        if (nrm_a && a.grid.pn != nullptr) {      // (uniform) point and normal of a match from ONE 32-byte record
          p_a = a.grid.pn[2 * (size_t)ga]; nv_a = a.grid.pn[2 * (size_t)ga + 1];
          p_b = a.grid.pn[2 * (size_t)gb]; nv_b = a.grid.pn[2 * (size_t)gb + 1];
        } else {
          p_a = a.grid.pts[ga]; p_b = a.grid.pts[gb];
          if (nrm_a) { nv_a = a.grid.nrm[ga]; nv_b = a.grid.nrm[gb]; }
        }
        if (TR::plane && a.src_nrm) { sn_a = a.src_nrm[min(ia, last)]; sn_b = a.src_nrm[min(ib, last)]; }
        __builtin_amdgcn_sched_barrier(0);
      }
      i0 += 2 * ITER_THREADS;
      sa = a.src[min(i0, last)]; pa = a.nn_pos[min(i0, last)];
      sb = a.src[min(i0 + ITER_THREADS, last)]; pb = a.nn_pos[min(i0 + ITER_THREADS, last)];
      __builtin_amdgcn_sched_barrier(0);
      if (!(i0 < end)) pa = NONE_U32;
      if (!(i0 + ITER_THREADS < end)) pb = NONE_U32;
      float qx, qy, qz;
      transform_point(T, s4a.x, s4a.y, s4a.z, qx, qy, qz);
      // stored matches: the stored distance (the feature search's is the 6-D one) or, where none is kept, formed again
      float va = 0.0f, vb2 = 0.0f;
      if (a.cw.enabled && posa != NONE_U32) va = a.nn_d2 ? a.nn_d2[ia] : d2_pinned(qx, qy, qz, p_a.x, p_a.y, p_a.z);
      accumulate(va, ia, qx, qy, qz, posa, p_a, nv_a, sn_a);
      transform_point(T, s4b.x, s4b.y, s4b.z, qx, qy, qz);
      if (a.cw.enabled && posb != NONE_U32) vb2 = a.nn_d2 ? a.nn_d2[ib] : d2_pinned(qx, qy, qz, p_b.x, p_b.y, p_b.z);
      accumulate(vb2, ib, qx, qy, qz, posb, p_b, nv_b, sn_b);
    }
  } else {
  // Warm start (a.warm_pos: the matches of the PREVIOUS iteration, may alias nn_pos): the old match is a real target point,
  // so its distance from the new q bounds the search -- near convergence that ball lies inside q's own cell for most
  // queries and the search is one cell scan.  Lanes whose bound exceeds a.warm_far_sq (or that have none) are counted:
  // the host falls back to the tiled kernels when they are many.
  uint32_t inext = beg + threadIdx.x;
  float4 s4n = inext < end ? a.src[inext] : make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t wn = (a.warm_pos && inext < end) ? a.warm_pos[inext] : NONE_U32;
  uint32_t nfar = 0, nsmall = 0;
  const MotionRef mref = {st->motion_acc, st->motion_eps};
  const float mstep = st->motion_pred;
  while (inext < end) {
    const uint32_t i = inext;
    const float4 s4 = s4n;
    const uint32_t w = wn;
    uint32_t pos = NONE_U32;
    float value = 0.0f;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), nvp = p, snp = p;
    inext += ITER_THREADS;
    if (inext < end) { s4n = a.src[inext]; if (a.warm_pos) wn = a.warm_pos[inext]; }
    float qx, qy, qz;
    transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
    {
      NN best;
      best.key = ((unsigned long long)__float_as_uint(a.max_sq) << 32);
      best.pos = NONE_U32;
      if (a.warm_pos) {
        bool far = true;
        if (w != NONE_U32) {
          const float4 pw = a.grid.pts[w];
          const float e = d2_pinned(qx, qy, qz, pw.x, pw.y, pw.z);
          if (e < a.max_sq) { best.key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(pw.w); best.pos = w; far = !(e < a.warm_far_sq); }
        }
        nfar += far ? 1u : 0u;
      }
      if (STORE && a.nn_lb != nullptr) {
        // (a run whose later iterations may be warm-started: the margin key of this search next to the match, and the count of
        //  the queries whose margin the next update would already have spent)
        float lb = 0.0f;
        nn_search_lb(a.grid, qx, qy, qz, a.max_sq, best, lst, &lb);
        const bool found = best.pos != NONE_U32;
        a.nn_lb[i] = lb > 0.0f ? margin_key(found, INFINITY, lb, mref) : (found ? 0.0f : MARGIN_NONE_NO_MATCH);
        nsmall += (lb > 0.0f && !margin_is_small(found, INFINITY, lb, __uint_as_float((uint32_t)(best.key >> 32)), a.max_sq, mstep)) ? 0u : 1u;
      } else {
        nn_search_from(a.grid, qx, qy, qz, a.max_sq, best, lst);
      }
      if (a.tie.mode != 0 && best.tie != 0u && best.pos != NONE_U32)      // (option "tie_rule")
        best.pos = tie_settle(a.grid, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)));
      pos = best.pos;
      value = __uint_as_float((uint32_t)(best.key >> 32));
      if (STORE) { a.nn_pos[i] = pos; if (a.nn_d2) a.nn_d2[i] = __uint_as_float((uint32_t)(best.key >> 32)); }
      if (METRIC != IM_NONE && pos != NONE_U32) {
        p = a.grid.pts[pos];
        if (TR::plane) { nvp = a.grid.nrm[pos]; if (a.src_nrm) snp = a.src_nrm[i]; } else if (TR::affine && a.grid.nrm) nvp = a.grid.nrm[pos];
      }
    }
    accumulate(value, i, qx, qy, qz, pos, p, nvp, snp);
  }
  if (a.warm_pos && a.unproven_cnt) {
    const double tot = wave_sum((double)nfar);
    if ((threadIdx.x & 63) == 0 && tot > 0.0) atomicAdd(a.unproven_cnt + (vb & 63u), (uint32_t)tot);
  }
  if (STORE && a.nn_lb != nullptr && a.unproven_cnt) {
    const double tot = wave_sum((double)nsmall);
    if ((threadIdx.x & 63) == 0 && tot > 0.0) atomicAdd(a.unproven_cnt + 64u + (vb & 63u), (uint32_t)tot);
  }
  }

  if (METRIC != IM_NONE) {
    __shared__ double sh[ITER_WAVES][SUMS_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < SUMS_MAX) {
#pragma unroll
      for (int w = 0; w < ITER_WAVES; ++w) sh[w][threadIdx.x] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TR::NA; ++k) {
      const double v = wave_sum(accA[k]);
      if (lane == 0) sh[wave][k] = v;
    }
#pragma unroll
    for (int k = 0; k < TR::NB; ++k) {
      const double v = wave_sum(accB[k]);
      if (lane == 0) sh[wave][28 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < SUMS_MAX)
      a.partials[(size_t)vb * SUMS_MAX + threadIdx.x] =
          (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  }
}

// ---- per target point: a lower bound on the squared distance to its nearest OTHER target point -------------------------
// What the warm-started iteration settles most queries with: if |q - p| < nnd(p) / 2 then p is the one nearest target
// point of q (any other p' has |q - p'| >= nnd(p) - |q - p| > |q - p|) -- no neighbour has to be looked at.  Computed once
// per target: minimum over the 3x3x3 block of cells around the point (itself excluded by position: a duplicate gives 0),
// capped by the distance to the faces of that block (whatever lies beyond is at least that far) -- a LOWER bound is all
// the test needs.
__global__ __launch_bounds__(256) void k_self_nn(GridDev g, float* __restrict__ safe2) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= g.n) return;
  const float4 p = g.pts[j];
  const int cx = min(max((int)floorf((p.x - g.ox) * g.inv_cell), 0), g.nx - 1), cy = min(max((int)floorf((p.y - g.oy) * g.inv_cell), 0), g.ny - 1),
            cz = min(max((int)floorf((p.z - g.oz) * g.inv_cell), 0), g.nz - 1);
  float best = INFINITY;
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
  for (int z = max(cz - 1, 0); z <= min(cz + 1, g.nz - 1); ++z)
    for (int y = max(cy - 1, 0); y <= min(cy + 1, g.ny - 1); ++y) {
      const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
      const uint32_t beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
      for (uint32_t k = beg; k < end; ++k) {
        const float4 o = g.pts[k];
        const float e = d2_pinned(p.x, p.y, p.z, o.x, o.y, o.z);
        if (k != j) best = fminf(best, e);
      }
    }
  float b = INFINITY;      // faces of the block that still have cells beyond them
  if (cx - 1 > 0) b = fminf(b, p.x - (g.ox + (float)(cx - 1) * g.cell));
  if (cx + 2 < g.nx) b = fminf(b, (g.ox + (float)(cx + 2) * g.cell) - p.x);
  if (cy - 1 > 0) b = fminf(b, p.y - (g.oy + (float)(cy - 1) * g.cell));
  if (cy + 2 < g.ny) b = fminf(b, (g.oy + (float)(cy + 2) * g.cell) - p.y);
  if (cz - 1 > 0) b = fminf(b, p.z - (g.oz + (float)(cz - 1) * g.cell));
  if (cz + 2 < g.nz) b = fminf(b, (g.oz + (float)(cz + 2) * g.cell) - p.z);
  if (b != INFINITY) { b = fmaxf(b - g.margin, 0.0f); best = fminf(best, b * b * KSHRINK); }
  safe2[j] = best;
}
void launch_self_nn(const GridDev& g, float* safe2, hipStream_t s) {
  if (g.n == 0) return;
  hipLaunchKernelGGL(k_self_nn, dim3((g.n + 255u) / 256u), dim3(256), 0, s, g, safe2);
}

// ---- the WARM-STARTED iteration: search + accumulation from the previous iteration's matches -------------------------
// From the second iteration on every query has a match from the iteration before.  That match is a real target point, so
// its distance from the NEW q = T s bounds the search: anything nearer (or as near, with a lower index) lies in the ball
// of that radius around q.  Near alignment the radius is a small fraction of a cell and the ball stays inside q's octant
// block (the 2x2x2 cells q leans towards) -- usually inside q's own cell: per axis the neighbour is looked at only when
// the ball reaches its face.  No tile is staged: a lane reads its old match and the one to three cells its ball touches
// straight from memory (neighbouring lanes read neighbouring lines).  Two to three memory round trips per query:
// {old match, its normal, the run boundaries} -> {candidates, 4 per trip} -> done.  Queries without a usable bound (no old
// match, bound beyond the octant block, cell in the grid's outer layer) take the generic shell search -- exact as well --
// and are counted: the host goes back to the tiled kernels when they are many.  The matches and therefore the sums are
// the ones every other form finds; the accumulation is the tiles' rank update Z += z z^T on the matrix cores (per-wave
// 16x16 f64 tile kept in registers across the whole chunk, fixed order => bitwise reproducible run to run).
constexpr int WARM_THREADS = 256;
constexpr int WARM_WAVES = WARM_THREADS / 64;

// NR runs of the sorted target array, EVERY point evaluated (nothing culled: the caller wants a bound on all the points it did
// not choose), eight independent loads in flight per trip over the flattened index space of the runs; keeps the best key and the
// two smallest squared distances a1 <= b2 over the DISTINCT points met (the clamped re-reads past the end are not counted), and the
// best point's record (bp) so that the caller need not fetch it again.
template <int NR>
__device__ __forceinline__ void scan_runs_track2(const float4* __restrict__ pts, const uint32_t (&rb)[NR], const uint32_t (&re)[NR], float qx, float qy, float qz,
                                                 NN& best, float& a1, float& b2, float4& bp) {
  uint32_t pre[NR];      // inclusive prefix sums of the run lengths
  uint32_t total = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) { total += re[r] - rb[r]; pre[r] = total; }
  for (uint32_t t = 0; t < total; t += 8) {
    uint32_t j[8];
    float4 pc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t tt = min(t + (uint32_t)k, total - 1u);
      uint32_t jj = rb[0] + tt;
#pragma unroll
      for (int r = 1; r < NR; ++r) jj = tt >= pre[r - 1] ? rb[r] + (tt - pre[r - 1]) : jj;
      j[k] = jj;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) pc[k] = pts[j[k]];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float e = d2_pinned(qx, qy, qz, pc[k].x, pc[k].y, pc[k].z);
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(pc[k].w);
      if (key < best.key) { best.key = key; best.pos = j[k]; bp = pc[k]; }
      if (t + (uint32_t)k < total) { b2 = __builtin_amdgcn_fmed3f(a1, b2, e); a1 = fminf(a1, e); }
    }
  }
}
// REC: 0 = the old match, its normal and its table entry are gathered through warm_pos; 1 = the same, and every query's
// match record {point, table entry} {normal} (16 + 12 B, two arrays in query order) and a 12-byte copy of its source point
// are written; 2 = those are READ instead -- 40 B per query in three coalesced loads, no gather at all for the queries the
// table settles (nearly all of them); a query whose match changes rewrites its record.
// The queries the table does NOT settle (a percent or so) are not searched where they turn up -- nearly every wave holds
// one, and the whole wave would walk the search code for it: each wave lists them in LDS (ballot order: no atomics, the
// same list in every run) and searches the list afterwards, densely packed (the list holds all of the wave's queries if
// need be: a source far from alignment).
constexpr int WARM_QCAP = 256;                              // listed queries per wave (16 B each); a list that could not take another round is searched at once
#define Z4 make_float4(0.f, 0.f, 0.f, 0.f)
template <int ACC, int REC>
__global__ __launch_bounds__(WARM_THREADS, 4) void k_warm(IterArgs a) {
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = st->T[(i / 3) * 4 + (i % 3)];     // columns 0..3, rows 0..2
  // (loop state read HERE, into scalar registers: a load of it inside the streaming loop is a vector-memory load whose wait
  //  -- vmcnt counts in order -- also waits for the next round's prefetch, i.e. serialises memory latency and arithmetic)
  const float smt[3] = {st->smt[0], st->smt[1], st->smt[2]};
  const MotionRef mref = {st->motion_acc, st->motion_eps};
  const float Dk = __fadd_rn(mref.acc, mref.eps) * 1.000001f;      // the motion clock now (rounded up): what a key is compared against
  const GridDev& g = a.grid;
  __shared__ __attribute__((aligned(16))) unsigned char raw[WARM_WAVES * FUSED_WAVE_BYTES];
  __shared__ float4 dq[WARM_WAVES][WARM_QCAP];             // listed queries: {q = T s, index}
  __shared__ float dr[WARM_WAVES][WARM_QCAP];              // ... and their bounds (squared)
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  float* const zb = reinterpret_cast<float*>(raw) + wave * (FUSED_WAVE_BYTES / 4);
  float4* const wq = dq[wave];
  float* const wr = dr[wave];
#ifdef CILHIP_EXP_PHASE_CLOCKS
  unsigned long long tprev_ = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x < 4096u) g_warm_stamp[blockIdx.x][0] = tprev_;
#endif
  typedef double double4_t __attribute__((ext_vector_type(4)));
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  constexpr int NC = FusedZ<ACC>::NC;
  constexpr bool DUAL = NC <= 8;
  constexpr bool NRM = FusedZ<ACC>::needs_normal;

  const uint32_t nb = gridDim.x;
  const uint32_t vb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);      // XCD-aware (gridDim.x is a multiple of 8)
  // The source in ROUNDS of 256 queries, dealt out evenly: every block gets floor(R / nb) rounds, the first R mod nb blocks IN DISPATCH
  // ORDER (blockIdx: round-robin over the XCDs) one more -- no block without work (a chunk rounded up to whole rounds left the last
  // 2 % of the blocks idle at 10M), the heavier blocks spread over the XCDs.  vb's range starts after the rounds of the chunks before it.
  const uint32_t rounds_total = (a.ns + WARM_THREADS - 1) / WARM_THREADS, rbase = rounds_total / nb, rrem = rounds_total % nb;
  const uint32_t per_x = nb >> 3, xme = blockIdx.x & 7u, jme = blockIdx.x >> 3;
  uint32_t heavy_before = 0;      // heavier chunks among vb' < vb: chunk (x, j) is heavier iff its block index j * 8 + x < rrem
  for (uint32_t x = 0; x < xme; ++x) heavy_before += rrem > x ? min((rrem - x + 7u) >> 3, per_x) : 0u;
  heavy_before += rrem > xme ? min((rrem - xme + 7u) >> 3, jme) : 0u;
  const uint64_t beg64 = ((uint64_t)vb * rbase + heavy_before) * WARM_THREADS;
  const uint32_t my_rounds = rbase + (blockIdx.x < rrem ? 1u : 0u);
  const uint32_t beg = beg64 < a.ns ? (uint32_t)beg64 : a.ns;
  const uint32_t end = (beg64 + (uint64_t)my_rounds * WARM_THREADS < a.ns) ? (uint32_t)(beg64 + (uint64_t)my_rounds * WARM_THREADS) : a.ns;
  const int sy = g.nx, sz = g.nx * g.ny;
  uint32_t nfar = 0;

  auto transform = [&](const float4 s4, float& qx, float& qy, float& qz) {
    qx = __fadd_rn(__fadd_rn(__fmul_rn(T[0], s4.x), __fadd_rn(__fmul_rn(T[3], s4.y), __fmul_rn(T[6], s4.z))), T[9]);
    qy = __fadd_rn(__fadd_rn(__fmul_rn(T[1], s4.x), __fadd_rn(__fmul_rn(T[4], s4.y), __fmul_rn(T[7], s4.z))), T[10]);
    qz = __fadd_rn(__fadd_rn(__fmul_rn(T[2], s4.x), __fadd_rn(__fmul_rn(T[5], s4.y), __fmul_rn(T[8], s4.z))), T[11]);
  };

  // rank update of the wave's 16x16 tile with one round of (up to) 64 correspondences (k_search_tiled, step 5)
  // (two halves: the terms z of the wave's correspondences -> LDS; then LDS -> f64 operands -> the matrix cores.  Between them
  //  a round's streamed registers are dead, which is where the streaming loop requests the data of the round after next.)
  auto z_to_lds = [&](bool has, float qx, float qy, float qz, const float4 pm, const float4 nm) {
    float z[16];
    fused_z<ACC>(has, qx, qy, qz, pm, nm, a.dst_mean, smt, z);
    if (DUAL) {
      float4* w4 = reinterpret_cast<float4*>(zb + lane * 8 + (lane >= 32 ? 16 : 0));
      w4[0] = make_float4(z[0], z[1], z[2], z[3]);
      w4[1] = make_float4(z[4], z[5], z[6], z[7]);
    } else {
      float2* w2 = reinterpret_cast<float2*>(zb + lane * NC);
#pragma unroll
      for (int c = 0; c < NC / 2; ++c) w2[c] = make_float2(z[2 * c], z[2 * c + 1]);
    }
  };
  auto lds_to_mfma = [&]() {
    __builtin_amdgcn_wave_barrier();
    if (DUAL) {
      const int comp = lane & 7, hf = (lane >> 3) & 1, k4 = lane >> 4;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int qi = hf * 32 + 4 * jj + k4;
        const double x = (double)zb[qi * 8 + hf * 16 + comp];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
      }
    } else {
      const int comp = lane & 15, k4 = lane >> 4;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const float f = zb[(4 * jj + k4) * NC + (comp < NC ? comp : 0)];
        const double x = comp < NC ? (double)f : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto rank_update = [&](bool has, float qx, float qy, float qz, const float4 pm, const float4 nm) {
    z_to_lds(has, qx, qy, qz, pm, nm);
    lds_to_mfma();
  };

  // The search of the queries their margin did not settle.  A listed query comes with a BOUND: the squared distance R2 from its
  // new position to its old match (a real target point: the nearest one is no farther), or the radius without one.  Every target
  // point inside the ball of radius sqrt(R2) + extra around the query is evaluated (extra = a quarter of a cell) -- the cells the
  // ball does not reach are skipped -- keeping the best key, the best point and the two smallest squared distances: the best is
  // the exact match (it lies inside the ball), and every other target point is at least min(second smallest, sqrt(R2) + extra)
  // away: the query leaves with a fresh margin key of up to a quarter of a cell.  Two LEVELS, each run over the wave's list 64
  // entries at a time, what level A cannot take packed to the front of the list for level B:
  //   A: the ball lies inside the 3x3x3 block around the query's cell: nine rows, each clipped to the cells the ball reaches -- the
  //      18 run boundaries leave together, then one trip per eight candidates, one for the match's normal;
  //   B: the 5x5x5 block slab by slab, rows beyond sqrt(best so far) + extra skipped, then (rarely) the shell search with the
  //      same margin.
  // Stores the match, its record and the key.
  const float extra = a.warm_extra * g.cell;
  auto slow_finish = [&](bool v, uint32_t i, float qx, float qy, float qz, NN& best, float4& bp, float key, float4& pm, float4& nm) -> bool {
    const bool has = v && best.pos != NONE_U32;
    // (option "tie_rule": the two smallest distances of the search were equal -- the reference's pick among the points at that distance.
    //  The key stays: every point but the match is at least the match's own distance away, whichever of them the match is.)
    if (a.tie.mode != 0 && has && best.tie != 0u) {
      const uint32_t w = tie_settle(g, a.tie, qx, qy, qz, best.pos, __uint_as_float((uint32_t)(best.key >> 32)));
      if (w != best.pos) { best.pos = w; bp = g.pts[w]; }
    }
    nm = Z4;
    if (NRM) nm = g.nrm[has ? best.pos : 0u];      // (unconditional: one trip for the whole wave)
    pm = has ? make_float4(bp.x, bp.y, bp.z, 0.f) : Z4;
    if (!has) nm = Z4;
    if (v) {
      a.nn_pos[i] = best.pos;
      a.warm_rec[i] = make_float4(pm.x, pm.y, pm.z, key);
      if (NRM) a.warm_rec_n[i] = F3{nm.x, nm.y, nm.z};
    }
    return has;
  };
  struct SlowGeom { int cx, cy, cz; float ux, uy, uz; bool inner, inside; uint32_t cid; };
  auto slow_geom = [&](bool v, float qx, float qy, float qz) -> SlowGeom {
    SlowGeom s;
    const float BIG = 1.0e9f;
    const float fx = fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG), fy = fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG),
                fz = fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG);
    s.cx = (int)floorf(fx); s.cy = (int)floorf(fy); s.cz = (int)floorf(fz);
    s.inner = v & (s.cx >= 1) & (s.cx <= g.nx - 2) & (s.cy >= 1) & (s.cy <= g.ny - 2) & (s.cz >= 1) & (s.cz <= g.nz - 2);
    s.inside = v & (s.cx >= 0) & (s.cx < g.nx) & (s.cy >= 0) & (s.cy < g.ny) & (s.cz >= 0) & (s.cz < g.nz);
    s.ux = qx - (g.ox + (float)s.cx * g.cell); s.uy = qy - (g.oy + (float)s.cy * g.cell); s.uz = qz - (g.oz + (float)s.cz * g.cell);
    s.cid = ((uint32_t)s.cz * (uint32_t)g.ny + (uint32_t)s.cy) * (uint32_t)g.nx + (uint32_t)s.cx;
    return s;
  };
  // level A: returns whether it took the query (the ball fits the 3x3x3 block)
  auto slow_levelA = [&](bool v, float qx, float qy, float qz, float R2, NN& best, float4& bp, float& key) -> bool {
    const SlowGeom s = slow_geom(v, qx, qy, qz);
    const float Rr = __fsqrt_rn(R2) * 1.000001f + extra;       // the ball's radius, rounded up
    const float R2c = Rr * Rr * 1.000001f;
    float b = INFINITY;      // faces of the 3x3x3 block that still have cells beyond them
    if (s.cx - 1 > 0) b = fminf(b, s.ux);
    if (s.cx + 2 < g.nx) b = fminf(b, g.cell - s.ux);
    if (s.cy - 1 > 0) b = fminf(b, s.uy);
    if (s.cy + 2 < g.ny) b = fminf(b, g.cell - s.uy);
    if (s.cz - 1 > 0) b = fminf(b, s.uz);
    if (s.cz + 2 < g.nz) b = fminf(b, g.cell - s.uz);
    // (a query in the grid's OUTER layer is taken too: the rows and cells of its block that lie outside the grid do not exist -- every
    //  target point is inside the grid --, they are skipped; 2.7 % of the queries of a 220^3 grid, which used to go to the shells)
    const bool fits = s.inside && (b == INFINITY || Rr < (fmaxf(b, 0.0f) + g.cell - 2.0f * g.margin) * 0.999999f);
    // (addresses valid for every lane: a lane that is not taken reads the rows of cell (1,1,1) and is masked afterwards, so that
    //  the loads leave together instead of one exec-masked group after the other)
    const int c0 = fits ? (int)s.cid : sz + sy + 1;
    const float gm[3] = {fmaxf(s.uz - g.margin, 0.0f), 0.0f, fmaxf(g.cell - s.uz - g.margin, 0.0f)};
    const float gn[3] = {fmaxf(s.uy - g.margin, 0.0f), 0.0f, fmaxf(g.cell - s.uy - g.margin, 0.0f)};
    const float gxl = fmaxf(s.ux - g.margin, 0.0f), gxr = fmaxf(g.cell - s.ux - g.margin, 0.0f);
    uint32_t rb9[9], re9[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const float gyz2 = gm[r / 3] * gm[r / 3] + gn[r % 3] * gn[r % 3];
      const bool exists = (unsigned)(s.cz + r / 3 - 1) < (unsigned)g.nz && (unsigned)(s.cy + r % 3 - 1) < (unsigned)g.ny;
      const bool take = fits && exists && gyz2 * KSHRINK <= R2c;
      const bool left = take && s.cx > 0 && (gyz2 + gxl * gxl) * KSHRINK <= R2c, right = take && s.cx + 1 < g.nx && (gyz2 + gxr * gxr) * KSHRINK <= R2c;
      const int row = take ? c0 + (r / 3 - 1) * sz + (r % 3 - 1) * sy : sz + sy + 1;
      const uint32_t va = g.cell_start[row - (left ? 1 : 0)], vb2 = g.cell_start[row + 1 + (right ? 1 : 0)];
      rb9[r] = take ? va : 0u; re9[r] = take ? vb2 : 0u;
    }
    best.key = ((unsigned long long)__float_as_uint(a.max_sq) << 32);
    best.pos = NONE_U32;
    if (fits) {
      float a1 = INFINITY, b2 = INFINITY;
      scan_runs_track2<9>(g.pts, rb9, re9, qx, qy, qz, best, a1, b2, bp);
      best.tie = (best.pos != NONE_U32 && b2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
      // every point that was not evaluated lies beyond the ball
      key = margin_key(best.pos != NONE_U32, best.pos != NONE_U32 ? b2 : a1, (__fsqrt_rn(R2) + extra) * 0.999999f, mref);
    }
    return fits;
  };
  // level B: settles every query it is given
  auto slow_levelB = [&](bool v, float qx, float qy, float qz, float R2, NN& best, float4& bp, float& key) {
    const SlowGeom s = slow_geom(v, qx, qy, qz);
    // (the bound enters as a key with a placeholder index that loses every tie: the old match itself lies inside what is scanned
    //  and is met again with its own)
    best.key = ((unsigned long long)__float_as_uint(fminf(R2, a.max_sq)) << 32) | 0xFFFFFFFFull;
    best.pos = NONE_U32;
    float a1 = INFINITY, b2 = INFINITY;
    bool proven = false;
    if (s.inner) {
      // the 5x5x5 block, one z-slab at a time from the middle outwards: five rows (runs of five x-adjacent cells, clipped to the
      // grid), a row skipped when its gap exceeds sqrt(best so far) + extra
      const int xa = max(s.cx - 2, 0), xb = min(s.cx + 2, g.nx - 1);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int dzz = (k == 0) ? 0 : (k & 1) ? (k + 1) / 2 : -(k / 2);      // 0, +1, -1, +2, -2
        const int z = s.cz + dzz;
        const bool zin = z >= 0 && z < g.nz;
        const float zl = g.oz + (float)z * g.cell;
        const float gz = axis_gap(qz, zl, zl + g.cell, g.margin);
        const float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
        const float lim2 = lim * lim * 1.000001f;
        uint32_t rb5[5], re5[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const int y = s.cy + r - 2;
          const bool on = zin && y >= 0 && y < g.ny;
          const float yl = g.oy + (float)y * g.cell;
          const float gy = axis_gap(qy, yl, yl + g.cell, g.margin);
          const bool take = on && (gz * gz + gy * gy) * KSHRINK <= lim2;
          const uint32_t row = take ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx : 0u;
          const uint32_t va = g.cell_start[row + (uint32_t)xa], vb2 = g.cell_start[row + (uint32_t)xb + 1u];
          rb5[r] = take ? va : 0u; re5[r] = take ? vb2 : 0u;
        }
        scan_runs_track2<5>(g.pts, rb5, re5, qx, qy, qz, best, a1, b2, bp);
      }
      float b = INFINITY;      // faces of the 5x5x5 block that still have cells beyond them
      if (s.cx - 2 > 0) b = fminf(b, s.ux);
      if (s.cx + 3 < g.nx) b = fminf(b, g.cell - s.ux);
      if (s.cy - 2 > 0) b = fminf(b, s.uy);
      if (s.cy + 3 < g.ny) b = fminf(b, g.cell - s.uy);
      if (s.cz - 2 > 0) b = fminf(b, s.uz);
      if (s.cz + 3 < g.nz) b = fminf(b, g.cell - s.uz);
      if (b != INFINITY) b = fmaxf(b, 0.0f) + 2.0f * g.cell - 2.0f * g.margin;
      const float lim = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) + extra) * 1.000001f;
      proven = b > 0.0f && lim * lim * 1.000001f < b * b * KSHRINK;      // everything outside the block lies beyond sqrt(best) + extra as well
    }
    bool skip = false;
    if (v && !proven) {
      // the shell search with the same margin (from the 5x5x5 block's result for an inner cell).  Counted: the host goes back to
      // the tiled kernels when these are many.
      ++nfar;
      const bool inside = (s.cx >= 0) & (s.cx < g.nx) & (s.cy >= 0) & (s.cy < g.ny) & (s.cz >= 0) & (s.cz < g.nz);
      int s0 = s.inner ? 3 : 0;
      if (!inside) {
        const float gx = axis_gap(qx, g.ox, g.ox + (float)g.nx * g.cell, g.margin), gy = axis_gap(qy, g.oy, g.oy + (float)g.ny * g.cell, g.margin),
                    gz = axis_gap(qz, g.oz, g.oz + (float)g.nz * g.cell, g.margin);
        skip = (gx * gx + gy * gy + gz * gz) * KSHRINK >= a.max_sq;      // farther than the radius from the whole grid
        s0 = max(0, max(max(-s.cx, s.cx - (g.nx - 1)), max(max(-s.cy, s.cy - (g.ny - 1)), max(-s.cz, s.cz - (g.nz - 1)))));
        if (skip) { best.pos = NONE_U32; key = margin_key(false, INFINITY, __fsqrt_rn((gx * gx + gy * gy + gz * gz) * KSHRINK) * 0.999999f, mref); }   // every target point lies inside the grid
      }
      if (!skip) nn_search_shells_margin(g, qx, qy, qz, s.cx, s.cy, s.cz, s0, best, a1, b2, bp, extra);
    }
    best.tie = (best.pos != NONE_U32 && b2 == __uint_as_float((uint32_t)(best.key >> 32))) ? 1u : 0u;
    if (!skip) {
      // every point that was not evaluated lies beyond sqrt(best) + extra (the radius + extra without a match)
      const float reach = (__fsqrt_rn(__uint_as_float((uint32_t)(best.key >> 32))) * 0.999999f + extra) * 0.999999f;
      key = margin_key(best.pos != NONE_U32, best.pos != NONE_U32 ? b2 : a1, reach, mref);
    }
  };

  uint32_t qcount = 0;      // (wave-uniform)
  // One round of the streaming loop for the query whose data has arrived: transform, the margin test, what a settled
  // query stores, the list entry of an unsettled one, the rank update.
  // The margin test: the record's key says that when the match p was established every OTHER target point was at
  // least |key| - (motion clock then) away from the query; the query has moved by at most (motion clock now) - (then) since, so
  // every other point is still at least mrg = |key| - Dk away -- if p is strictly nearer than that it is THE nearest target
  // point (ties excluded by the strictness), and nothing is looked at: not even the query's cell.  A negative key is the same
  // bound for a query WITHOUT a match, over all target points: if mrg still exceeds the radius there is still none.
  // lbv / s2 (REC 1): the key the search left in nn_lb (a.lb_valid) and the old match's nearest-other-point table entry -- any other
  // target point p' has |q - p'| >= nnd(p) - |q - p|: a second lower bound, the larger key wins.
  auto round = [&](uint32_t i, bool valid, const F3 s3c, uint32_t w, float4 pm, float4 nm, float keyv, float s2) {
    float qx, qy, qz;
    transform(make_float4(s3c.x, s3c.y, s3c.z, 0.f), qx, qy, qz);
    const float e_old = d2_pinned(qx, qy, qz, pm.x, pm.y, pm.z);
    if (REC != 2) {
      if (w != NONE_U32) {
        const float alt = margin_key(true, INFINITY, __fsub_rn(__fsqrt_rn(fmaxf(s2, 0.0f)) * 0.999999f, __fsqrt_rn(e_old) * 1.000001f), mref);
        keyv = fmaxf(a.lb_valid ? fmaxf(keyv, 0.0f) : 0.0f, alt);
      } else {
        keyv = a.lb_valid ? fminf(keyv, MARGIN_NONE_NO_MATCH) : MARGIN_NONE_NO_MATCH;
      }
    }
    const float mrg = __fsub_rn(fabsf(keyv), Dk);
    const float m2 = mrg * mrg * KSHRINK;
    const bool ok = valid && mrg > 0.0f;
    const bool shas = ok && keyv > 0.0f && e_old < m2 && e_old < a.max_sq;
    const bool settled = shas || (ok && keyv < 0.0f && m2 >= a.max_sq);
    if (settled) {
      if (REC != 2 && a.nn_pos != a.warm_pos) a.nn_pos[i] = w;
      if (REC == 1) { a.warm_rec[i] = make_float4(pm.x, pm.y, pm.z, keyv); if (NRM) a.warm_rec_n[i] = F3{nm.x, nm.y, nm.z}; }
    }
    const bool todo = valid && !settled;
    const unsigned long long um = __ballot(todo);
    if (todo) {
      // the list entry and its bound: the squared distance to the old match if it has one inside the radius, else the radius
      const uint32_t o = qcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u));
      wq[o] = make_float4(qx, qy, qz, __uint_as_float(i));
      wr[o] = ((REC == 2 ? !(keyv < 0.0f) : w != NONE_U32) && e_old < a.max_sq) ? e_old : a.max_sq;
    }
    qcount += (uint32_t)__popcll(um);
    z_to_lds(shas, qx, qy, qz, pm, nm);
  };
  // What a round streams in.  REC 2: the 12-byte copy of the source point and the match record {point, margin key} {normal}
  // -- 40 B per query (28 without normals), all of it coalesced, TWO rounds in flight per wave (sets A and B; vmcnt retires in
  // order, so the wait for A leaves B's loads flying).  REC 1: the sorted source record, the stored position and the search's
  // margin key one round, the gathers through that position (old match, its normal, its table entry) the next: a three-stage
  // pipeline with one wait per round for loads that were issued a whole round earlier.
  uint32_t base = beg, qlisted = 0;
  F3 sA = F3{0.f, 0.f, 0.f}, nA = F3{0.f, 0.f, 0.f}, sB = F3{0.f, 0.f, 0.f}, nB = F3{0.f, 0.f, 0.f};
  float4 rA = Z4, rB = Z4;
  uint32_t iA = beg + threadIdx.x, iB = iA + WARM_THREADS;
  // (REC 0 / 1) stage 1 -> 2: source record + position of the round after next; stage 2 -> 3: what was gathered for the next round
  uint32_t w1 = NONE_U32, w2 = NONE_U32;
  float l1 = 0.0f, l2 = 0.0f;
  F3 s1 = F3{0.f, 0.f, 0.f}, s2_ = F3{0.f, 0.f, 0.f};
  float4 gp = Z4, gn = Z4;
  float gs = -1.0f;
  // (the three loads leave in THIS order everywhere -- scheduling barriers -- : the wait for a set is computed from the
  //  position of its loads in the in-order vmcnt queue, merged over all paths into the loop)
  auto load2 = [&](uint32_t k, F3& sv, float4& rv, F3& nv) {
    __builtin_amdgcn_sched_barrier(0);
    sv = a.warm_src3[k];
    __builtin_amdgcn_sched_barrier(0);
    rv = a.warm_rec[k];
    __builtin_amdgcn_sched_barrier(0);
    if (NRM) nv = a.warm_rec_n[k];
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load1 = [&](uint32_t k, F3& sv, uint32_t& wv, float& lv) { const float4 t4 = a.src[k]; sv = F3{t4.x, t4.y, t4.z}; wv = a.warm_pos[k]; lv = a.nn_lb[k]; };
  // (every load of the streaming loop is UNCONDITIONAL, from an index clamped into the chunk / a position clamped into the
  //  target: a load under a divergent branch may or may not have been issued as far as the compiler's vmcnt bookkeeping
  //  is concerned, and the waits it then inserts drain the younger prefetches as well)
  const uint32_t last = end > beg ? end - 1u : 0u;
  auto gather = [&](uint32_t wv) {
    const uint32_t wc = wv != NONE_U32 ? wv : 0u;
    if (NRM && g.pn != nullptr) { gp = g.pn[2 * (size_t)wc]; gn = g.pn[2 * (size_t)wc + 1]; }      // (uniform) point and normal from one 32-byte record
    else { gp = g.pts[wc]; if (NRM) gn = g.nrm[wc]; }
    gs = a.safe2[wc];
    if (wv == NONE_U32) { gp = gn = Z4; gs = -1.0f; }
  };
  WARM_CLK(0);
  for (;;) {
  // (the pipeline is filled HERE, at every entry of the streaming loop -- also after a list that had to be searched early:
  //  registers with loads in flight must not live across that search, where they would be spilled and reloaded)
  // (unconditionally: an empty chunk reads element 0, which exists)
  if (REC == 2) {
    load2(min(iA, last), sA, rA, nA);
    load2(min(iB, last), sB, rB, nB);
  } else {
    load1(min(iA, last), s2_, w2, l2);         // next round: record, then (dependent) its gathers
    gather(w2);
    load1(min(iA + WARM_THREADS, last), s1, w1, l1);      // the round after: record
  }
  // stream rounds until the chunk is done -- or the wave's list could not take two more rounds' queries (a source far from
  // alignment lists most of them): then the list is searched first
  if (REC == 2) {
    for (; base < end && qcount <= (uint32_t)(WARM_QCAP - 128); base += 2 * WARM_THREADS) {
      // (a set's registers are consumed -- down to the terms in LDS -- BEFORE the set is requested again, so that the new
      //  loads can land in the same registers: no copy at the loop's end that would have to wait for them)
      round(iA, iA < end, sA, NONE_U32, make_float4(rA.x, rA.y, rA.z, 0.f), make_float4(nA.x, nA.y, nA.z, 0.f), rA.w, 0.0f);
      __builtin_amdgcn_sched_barrier(0);
      iA += 2 * WARM_THREADS;
      load2(min(iA, last), sA, rA, nA);
      __builtin_amdgcn_sched_barrier(0);
      lds_to_mfma();
      round(iB, iB < end, sB, NONE_U32, make_float4(rB.x, rB.y, rB.z, 0.f), make_float4(nB.x, nB.y, nB.z, 0.f), rB.w, 0.0f);
      __builtin_amdgcn_sched_barrier(0);
      iB += 2 * WARM_THREADS;
      load2(min(iB, last), sB, rB, nB);
      __builtin_amdgcn_sched_barrier(0);
      lds_to_mfma();
    }
  } else {
    for (; base < end && qcount <= (uint32_t)(WARM_QCAP - 64); base += WARM_THREADS) {
      const uint32_t i = iA;
      const F3 sc = s2_;
      const uint32_t wc = w2;
      const float4 pc = gp, nc = gn;
      const float gc = gs, lc = l2;
      // next round: its record has arrived, its gathers leave now; the round after: its record leaves now
      iA += WARM_THREADS;
      s2_ = s1; w2 = w1; l2 = l1;
      gather(w2);
      load1(min(iA + WARM_THREADS, last), s1, w1, l1);
      round(i, i < end, sc, wc, pc, nc, lc, gc);
      lds_to_mfma();
    }
  }
  // the listed queries: two levels, 64 entries per round, what level A cannot take packed to the front of the list for level B
  __builtin_amdgcn_wave_barrier();
  WARM_CLK(1);
  qlisted += qcount;
  for (int level = 0; level < 2 && qcount != 0u; ++level) {
    uint32_t nopen = 0;
    for (uint32_t b0 = 0; b0 < qcount; b0 += 64u) {
      const bool v = b0 + (uint32_t)lane < qcount;
      const uint32_t e = min(b0 + (uint32_t)lane, (uint32_t)(WARM_QCAP - 1));      // (lanes beyond the list: masked by v)
      const float4 ent = wq[e];
      const float R2 = wr[e];
      NN best;
      float4 bp = Z4;
      float key = 0.0f;
      bool taken = true;
      if (level == 0) taken = slow_levelA(v, ent.x, ent.y, ent.z, R2, best, bp, key);
      else slow_levelB(v, ent.x, ent.y, ent.z, R2, best, bp, key);
      const bool open = v && !taken;
      const unsigned long long om = __ballot(open);
#ifdef CILHIP_EXP_PHASE_CLOCKS
      if (level == 0) {      // who is left open by level A: all / bound = the radius / bound beyond a cell
        if (v) atomicAdd(&g_warm_clk[6], 1ull);
        if (open) atomicAdd(&g_warm_clk[7], 1ull);
        if (open && R2 >= a.max_sq) atomicAdd(&g_warm_clk[14], 1ull);
        if (open && R2 < a.max_sq && R2 > g.cell * g.cell) atomicAdd(&g_warm_clk[15], 1ull);
      }
#endif
      // (this round's entries are in registers: the front of the list up to b0 + 64 is free)
      if (open) {
        const uint32_t o = nopen + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
        wq[o] = ent; wr[o] = R2;
      }
      nopen += (uint32_t)__popcll(om);
      float4 pm = Z4, nm = Z4;
      const bool has = slow_finish(v && taken, __float_as_uint(ent.w), ent.x, ent.y, ent.z, best, bp, key, pm, nm);
      rank_update(has, ent.x, ent.y, ent.z, pm, nm);
      WARM_CLK(2 + level);
    }
    qcount = nopen;
    __builtin_amdgcn_wave_barrier();
  }
  qcount = 0;
  __builtin_amdgcn_wave_barrier();
  if (base >= end) break;
  }
  if (a.unproven_cnt) {
    const double tot = wave_sum((double)nfar);
    if (lane == 0 && tot > 0.0) atomicAdd(a.unproven_cnt + (vb & 63u), (uint32_t)tot);
    if (lane == 0 && qlisted != 0u) atomicAdd(a.unproven_cnt + 64u + ((vb * WARM_WAVES + (uint32_t)wave) & 63u), qlisted);   // listed queries: is the form paying?
  }
  double* const db = reinterpret_cast<double*>(raw + wave * FUSED_WAVE_BYTES);
#pragma unroll
  for (int r = 0; r < 4; ++r) db[r * 64 + lane] = acc[r];
  __syncthreads();
  if (threadIdx.x < SUMS_MAX) {
    int i1, j1, i2, j2;
    const bool used = FusedZ<ACC>::slot_terms((int)threadIdx.x, i1, j1, i2, j2);
    double v1 = 0.0, v2 = 0.0;
    if (used) {
      const int e1 = (i1 >> 2) * 64 + 16 * (i1 & 3) + j1, e1b = ((i1 + 8) >> 2) * 64 + 16 * ((i1 + 8) & 3) + j1 + 8;
      const int e2 = i2 >= 0 ? (i2 >> 2) * 64 + 16 * (i2 & 3) + j2 : 0, e2b = i2 >= 0 ? ((i2 + 8) >> 2) * 64 + 16 * ((i2 + 8) & 3) + j2 + 8 : 0;
      for (int w = 0; w < WARM_WAVES; ++w) {
        const double* dw = reinterpret_cast<const double*>(raw + w * FUSED_WAVE_BYTES);
        v1 += dw[e1];
        if (DUAL) v1 += dw[e1b];
        if (i2 >= 0) { v2 += dw[e2]; if (DUAL) v2 += dw[e2b]; }
      }
    }
    a.partials[(size_t)vb * SUMS_MAX + threadIdx.x] = v1 - v2;
  }
  WARM_CLK(5);
#ifdef CILHIP_EXP_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 4096u) g_warm_stamp[blockIdx.x][1] = wall_clock64();
  if (lane == 0 && blockIdx.x < 4096u && qlisted) atomicAdd(&g_warm_stamp[blockIdx.x][2], (unsigned long long)qlisted);
#endif
}

#undef Z4
template <int ACC>
static void launch_warm_m(const IterArgs& a, int rec, int nblocks, hipStream_t s) {
  const dim3 g(nblocks), b(WARM_THREADS);
  const hipEvent_t ev_start = g_ev_start, ev_stop = g_ev_stop;      // (armed by set_launch_events: consumed here)
  g_ev_start = g_ev_stop = nullptr;
  if (rec == 2) launch_ev((k_warm<ACC, 2>), g, b, s, ev_start, ev_stop, a);
  else launch_ev((k_warm<ACC, 1>), g, b, s, ev_start, ev_stop, a);
}
int warm_num_blocks(uint32_t ns) {
  // ONE generation of blocks (4 resident per CU: registers, LDS): every wave searches its list once, at the end of its chunk --
  // with more, shorter blocks those latency-bound tails take wave slots from the streaming ones (measured: 2048 / 4096 / 8192
  // blocks 0.106 / 0.124 / 0.142 ms at 10M)
  static const long exp_nb = [] { const char* e = getenv("CILHIP_EXP_WARM_BLOCKS"); return e ? atol(e) : 0L; }();
  // Small clouds: at least eight rounds per wave (a block's fixed costs -- pipeline fill, list search, row -- against its share
  // of the stream), and 64 blocks are few enough for the epilogue to fold their rows itself, without the stage-1 kernel
  // (measured per step: 100k points 64 blocks 0.0317 ms, 392 blocks 0.0350; 1M points 512 blocks 0.0511, 1024 blocks 0.0530).
  if (exp_nb > 0) return (int)((exp_nb + 7) & ~7L);
  long nb = (long)ns / (8 * WARM_THREADS);
  if (nb > 1024) nb = 1024;
  if (nb < 64) nb = 64;
  return (int)((nb + 7) & ~7L);
}
// (squared distances are NOT written by this form -- a store inside the streaming loop shares the in-order vmcnt counter with
//  the prefetched loads; nobody reads them in the configurations that run warm-started (no post-filters, no weight
//  evaluators), and launch_fill_d2 recomputes them from the stored matches on demand)
void launch_warm(const IterArgs& a, int metric, int rec, int nblocks, hipStream_t s) {
  switch (metric) {
    case IM_KABSCH: launch_warm_m<IM_KABSCH>(a, rec, nblocks, s); break;
    case IM_PLANE: launch_warm_m<IM_PLANE>(a, rec, nblocks, s); break;
    case IM_POINT: launch_warm_m<IM_POINT>(a, rec, nblocks, s); break;
    default: launch_warm_m<IM_BOTH>(a, rec, nblocks, s); break;
  }
}

// the 12-byte copy of the sorted source the record-reading warm kernel streams (once per sort of a source)
__global__ void k_copy_src3(const float4* __restrict__ src, uint32_t ns, F3* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) { const float4 v = src[i]; out[i] = F3{v.x, v.y, v.z}; }
}
__global__ void k_interleave_pn(const float4* __restrict__ pts, const float4* __restrict__ nrm, uint32_t n, float4* __restrict__ pn) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { pn[2 * (size_t)i] = pts[i]; pn[2 * (size_t)i + 1] = nrm[i]; }
}
void launch_interleave_pn(const float4* pts, const float4* nrm, uint32_t n, float4* pn, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_interleave_pn, dim3((n + 255u) / 256u < 8192u ? (n + 255u) / 256u : 8192u), dim3(256), 0, s, pts, nrm, n, pn);
}
void launch_copy_src3(const float4* src_sorted, uint32_t ns, F3* out, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_copy_src3, dim3((int)((ns + 255u) / 256u < 4096u ? (ns + 255u) / 256u : 4096u)), dim3(256), 0, s, src_sorted, ns, out);
}

// ---- accumulation over REVERSE matches (search directions FIRST_TO_SECOND / BOTH without post-filters) ----------------
// The pair list of those directions (bidir.hip: sort, union / intersection, ordered compaction) is what a caller of
// getCorrespondences() sees; the ICP loop only needs its SUMS, and a sum does not care about the list's order: the
// reverse matches are accumulated where they are found -- element i = target point at sorted position i (read in order),
// its match = a record of the source's own grid (gathered; neighbours match neighbours) -- and BOTH is the forward pass
// plus the reverse matches that are not reciprocal duplicates (mode 2), its reciprocal form the duplicates alone (mode 3).
// A reverse match (target i -> source s) duplicates a forward one iff the forward match of s is i.
template <int METRIC>
__global__ __launch_bounds__(ITER_THREADS) void k_acc_reverse(IterArgs a, const float4* __restrict__ sgrid_pts, const uint32_t* __restrict__ rev_pos, uint32_t nd,
                                                              int mode, const uint32_t* __restrict__ fwd_pos, const uint32_t* __restrict__ src_inv) {
  using TR = AccTraits<METRIC>;
  const IcpState* __restrict__ st = a.state;
  if (st->done) return;
  if (a.skip_if_inner_done && st->inner_done) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = st->T[i];
  float iL[9], it[3], smt[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) iL[i] = st->innerL[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { it[i] = st->innert[i]; smt[i] = st->smt[i]; }
  const float dmean[3] = {a.dst_mean[0], a.dst_mean[1], a.dst_mean[2]};
  double accA[TR::NA];
  double accB[TR::NB > 0 ? TR::NB : 1];
#pragma unroll
  for (int i = 0; i < TR::NA; ++i) accA[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (TR::NB > 0 ? TR::NB : 1); ++i) accB[i] = 0.0;
  const uint32_t nb = gridDim.x;
  const uint32_t vb = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);      // XCD-aware (gridDim.x is a multiple of 8)
  const uint32_t chunk = (((nd + nb - 1) / nb) + 63u) & ~63u;
  const uint64_t beg64 = (uint64_t)vb * chunk;
  const uint32_t beg = beg64 < nd ? (uint32_t)beg64 : nd;
  const uint32_t end = (beg64 + chunk < nd) ? (uint32_t)(beg64 + chunk) : nd;
  for (uint32_t i = beg + threadIdx.x; i < end; i += ITER_THREADS) {
    const uint32_t pos = rev_pos[i];
    if (pos == NONE_U32) continue;
    const float4 s4 = sgrid_pts[pos];
    if (mode >= 2) {
      const bool dup = fwd_pos[src_inv[__float_as_uint(s4.w)]] == i;
      if (dup == (mode == 2)) continue;
    }
    const float4 p = a.grid.pts[i];
    float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TR::plane) nv = a.grid.nrm[i];
    float qx, qy, qz;
    transform_point(T, s4.x, s4.y, s4.z, qx, qy, qz);
    float wq = 1.0f, wp = 1.0f;
    if (a.cw.enabled) pair_weights(a.cw, d2_pinned(qx, qy, qz, p.x, p.y, p.z), wq, wp);     // the reverse search's distance, formed again
    accumulate_pair<METRIC>(accA, accB, T, iL, it, smt, dmean, false, true, qx, qy, qz, i, p, nv, nv, wq, wp);
  }
  __shared__ double sh[ITER_WAVES][SUMS_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < SUMS_MAX) {
#pragma unroll
    for (int w = 0; w < ITER_WAVES; ++w) sh[w][threadIdx.x] = 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TR::NA; ++k) {
    const double v = wave_sum(accA[k]);
    if (lane == 0) sh[wave][k] = v;
  }
#pragma unroll
  for (int k = 0; k < TR::NB; ++k) {
    const double v = wave_sum(accB[k]);
    if (lane == 0) sh[wave][28 + k] = v;
  }
  __syncthreads();
  if (threadIdx.x < SUMS_MAX)
    a.partials[(size_t)vb * SUMS_MAX + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

void launch_acc_reverse(const IterArgs& a, int metric, const float4* sgrid_pts, const uint32_t* rev_pos, uint32_t nd, int mode, const uint32_t* fwd_pos,
                        const uint32_t* src_inv, int nblocks, hipStream_t s) {
  const dim3 g(nblocks), b(ITER_THREADS);
  switch (metric) {
    case IM_KABSCH: hipLaunchKernelGGL((k_acc_reverse<IM_KABSCH>), g, b, 0, s, a, sgrid_pts, rev_pos, nd, mode, fwd_pos, src_inv); break;
    case IM_PLANE: hipLaunchKernelGGL((k_acc_reverse<IM_PLANE>), g, b, 0, s, a, sgrid_pts, rev_pos, nd, mode, fwd_pos, src_inv); break;
    case IM_POINT: hipLaunchKernelGGL((k_acc_reverse<IM_POINT>), g, b, 0, s, a, sgrid_pts, rev_pos, nd, mode, fwd_pos, src_inv); break;
    default: hipLaunchKernelGGL((k_acc_reverse<IM_BOTH>), g, b, 0, s, a, sgrid_pts, rev_pos, nd, mode, fwd_pos, src_inv); break;
  }
}

int iter_num_blocks(uint32_t ns) {
  // >= 8 blocks per CU on 256 CUs when there is enough work; multiple of 8 for the XCD mapping;
  // at least one wave of work per block.
#ifndef CILHIP_ITER_BLOCKS
#define CILHIP_ITER_BLOCKS 2048
#endif
  long want = ((long)ns + 255) / 256;
  long nb = want < CILHIP_ITER_BLOCKS ? want : CILHIP_ITER_BLOCKS;
  nb = (nb + 7) & ~7L;
  if (nb < 8) nb = 8;
  return (int)nb;
}

template <int METRIC>
static void launch_iter_m(const IterArgs& a, bool search, bool store, int nblocks, hipStream_t s) {
  dim3 g(nblocks), b(ITER_THREADS);
  if (search && store) hipLaunchKernelGGL((k_iter<METRIC, true, true>), g, b, 0, s, a);
  else if (search) hipLaunchKernelGGL((k_iter<METRIC, true, false>), g, b, 0, s, a);
  else hipLaunchKernelGGL((k_iter<METRIC, false, false>), g, b, 0, s, a);
}

void launch_iter(const IterArgs& a, int metric, bool search, bool store, int nblocks, hipStream_t s) {
  switch (metric) {
    case IM_NONE: launch_iter_m<IM_NONE>(a, search, store, nblocks, s); break;
    case IM_KABSCH: launch_iter_m<IM_KABSCH>(a, search, store, nblocks, s); break;
    case IM_PLANE: launch_iter_m<IM_PLANE>(a, search, store, nblocks, s); break;
    case IM_POINT: launch_iter_m<IM_POINT>(a, search, store, nblocks, s); break;
    case IM_AFF0: hipLaunchKernelGGL((k_iter<IM_AFF0, false, false>), dim3(nblocks), dim3(ITER_THREADS), 0, s, a); break;   // stored matches only
    case IM_AFF1: hipLaunchKernelGGL((k_iter<IM_AFF1, false, false>), dim3(nblocks), dim3(ITER_THREADS), 0, s, a); break;
    case IM_AFF2: hipLaunchKernelGGL((k_iter<IM_AFF2, false, false>), dim3(nblocks), dim3(ITER_THREADS), 0, s, a); break;
    default: launch_iter_m<IM_BOTH>(a, search, store, nblocks, s); break;
  }
}

// ---- epilogue: fixed-order reduction of block partials + solve + state update --------------------
__device__ void reduce_partials_block(const double* __restrict__ partials, int nblocks, double* sums /*shared*/) {
  __shared__ double sh[4][64];
  const int slot = threadIdx.x & 63, grp = threadIdx.x >> 6;
  double v = 0.0;
  if (slot < SUMS_MAX) {
    int b = grp;
    for (; b + 28 < nblocks; b += 32) {          // 8 independent loads in flight, added in ascending order
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = partials[(size_t)(b + 4 * k) * SUMS_MAX + slot];
#pragma unroll
      for (int k = 0; k < 8; ++k) v += r[k];
    }
    for (; b < nblocks; b += 4) v += partials[(size_t)b * SUMS_MAX + slot];
  }
  sh[grp][slot] = v;
  __syncthreads();
  if (threadIdx.x < SUMS_MAX)
    sums[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
  __syncthreads();
}

// Stage 1 of the cross-block reduction: REDUCE_GROUPS blocks, each folding a contiguous slice of the
// per-block partials (fixed order => deterministic).  A single block reading all 2048 x 48 doubles
// is latency-bound (~170 us measured); 32 blocks do it in a few us.
constexpr int REDUCE_GROUPS_MAX = 128;
static inline int reduce_groups(int nblocks) { return nblocks > 4096 ? REDUCE_GROUPS_MAX : 32; }   // ~50 rows per group at most

__global__ __launch_bounds__(256) void k_reduce_stage1(const double* __restrict__ partials, int nblocks, double* __restrict__ stage) {
  __shared__ double sums[SUMS_MAX];
  const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b0 = blockIdx.x * per;
  const int b1 = min(b0 + per, nblocks);
  reduce_partials_block(partials + (size_t)b0 * SUMS_MAX, max(b1 - b0, 0), sums);
  if (threadIdx.x < SUMS_MAX) stage[blockIdx.x * SUMS_MAX + threadIdx.x] = sums[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_reduce_partials(const double* partials, int nblocks, double* out) {
  __shared__ double sums[SUMS_MAX];
  reduce_partials_block(partials, nblocks, sums);
  if (threadIdx.x < SUMS_MAX) out[threadIdx.x] = sums[threadIdx.x];
}

// partials[nblocks][SUMS_MAX] -> out[SUMS_MAX]; `stage` is scratch of REDUCE_GROUPS*SUMS_MAX doubles.
void launch_reduce_partials(const double* partials, int nblocks, double* stage, double* out, hipStream_t s) {
  if (nblocks > 64) {
    const int G = reduce_groups(nblocks);
    hipLaunchKernelGGL(k_reduce_stage1, dim3(G), dim3(256), 0, s, partials, nblocks, stage);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, (const double*)stage, G, out);
  } else {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, partials, nblocks, out);
  }
}

// Stage 1 with a number of groups the CALLER fixes (every group's row is written, empty groups as zeros)
void launch_reduce_stage1_groups(const double* partials, int nblocks, double* stage, int groups, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_stage1, dim3(groups), dim3(256), 0, s, partials, nblocks, stage);
}

// Stage 1 only (the epilogue kernel k_solve folds the REDUCE_GROUPS rows itself).  Returns the number
// of rows k_solve has to read from `stage`, or 0 if it should read `partials` directly.
int launch_reduce_stage1(const double* partials, int nblocks, double* stage, hipStream_t s) {
  if (nblocks <= 64) return 0;
  const int G = reduce_groups(nblocks);
  hipLaunchKernelGGL(k_reduce_stage1, dim3(G), dim3(256), 0, s, partials, nblocks, stage);
  return G;
}

// The warm-started iteration's margin test needs two numbers about the queries q = T s (s in the source's bounding box: centre c,
// half extents h, source coordinates):
//   * how far any query moves when T becomes T' :  |(T' - T)(s, 1)| = |A s + b| <= |A c + b| + sum_j |A e_j| h_j   (A = L' - L, b = t' - t;
//     an affine function of s, the norm of its linear part bounded column by column) -- f64 of f32 entries, then rounded UP;
//   * the rounding error of a computed query: three products and three sums per component, each within 2^-24 relative of
//     |L_r0 x| + |L_r1 y| + |L_r2 z| + |t_r|: at most 2^-22 of that sum per component, sqrt(3) 2^-22 < 2^-21 for the norm of the three.
__device__ __forceinline__ float motion_eps_of(const float* T, const float* c, const float* h) {
  float m = 0.0f;
  for (int r = 0; r < 3; ++r) {
    float v = fabsf(T[12 + r]);
    for (int j = 0; j < 3; ++j) v += fabsf(T[j * 4 + r]) * (fabsf(c[j]) + h[j]);
    m = fmaxf(m, v);
  }
  return m * 6.0e-7f;      // > 2^-21
}
__device__ __forceinline__ float motion_step_of(const float* Told, const float* Tnew, const float* c, const float* h) {
  double v2 = 0.0, spread = 0.0;
  for (int r = 0; r < 3; ++r) {
    double v = (double)Tnew[12 + r] - (double)Told[12 + r];
    for (int j = 0; j < 3; ++j) v += ((double)Tnew[j * 4 + r] - (double)Told[j * 4 + r]) * (double)c[j];
    v2 += v * v;
  }
  for (int j = 0; j < 3; ++j) {
    double col = 0.0;
    for (int r = 0; r < 3; ++r) { const double d = (double)Tnew[j * 4 + r] - (double)Told[j * 4 + r]; col += d * d; }
    spread += sqrt(col) * (double)h[j];
  }
  return (float)((sqrt(v2) + spread) * 1.000001);      // (the conversion rounds to nearest: 2^-24 relative, covered)
}

__device__ void reset_inner(IcpState* st) {
  for (int i = 0; i < 9; ++i) { st->dLd[i] = (i % 4 == 0) ? 1.0 : 0.0; st->innerL[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
  for (int i = 0; i < 3; ++i) { st->dtd[i] = 0.0; st->innert[i] = 0.0f; }
  st->inner_done = 0;
  st->pad0 = 0;
}

// The epilogue proper (one block of 256 threads): k_solve's body, also the tail of k_reduce_solve's last block.
__device__ __forceinline__ void solve_body(const SolveArgs& a) {
  __shared__ double sums[SUMS_MAX];
  __shared__ IcpState lst;   // the state is pulled into LDS in one coalesced pass, updated by one lane, written back in one pass:
                             // the serial epilogue then pays one global round trip instead of one per field it touches
  static_assert(sizeof(IcpState) % 4 == 0, "IcpState is copied as dwords");
  constexpr int ST_DWORDS = (int)(sizeof(IcpState) / 4);
  if (a.state->done) return;
  // (everything the kernel reads from global memory is REQUESTED before anything is waited for: the state -- one dword per thread --,
  //  the search kernels' counters and the partial rows travel together: one far round trip instead of three in a row)
  static_assert(ST_DWORDS <= 256, "one dword of the state per thread");
  const uint32_t sreg = (int)threadIdx.x < ST_DWORDS ? reinterpret_cast<const uint32_t*>(a.state)[threadIdx.x] : 0u;
  __shared__ unsigned int unproven_total;
  __shared__ unsigned int listed_total;
  const bool counters = a.unproven_cnt != nullptr && a.gn_last_step && threadIdx.x < 128;   // (once per iteration; wave 0: unproven, wave 1: listed)
  unsigned int cv = counters ? a.unproven_cnt[threadIdx.x] : 0u;
  if (a.nblocks > 0) {
    reduce_partials_block(a.partials, a.nblocks, sums);
  } else {
    if (threadIdx.x < SUMS_MAX) sums[threadIdx.x] = a.reduced[threadIdx.x];
  }
  if ((int)threadIdx.x < ST_DWORDS) reinterpret_cast<uint32_t*>(&lst)[threadIdx.x] = sreg;
  if (counters) {
    a.unproven_cnt[threadIdx.x] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cv += __shfl_down(cv, off, 64);
    if (threadIdx.x == 0) unproven_total = cv;
    if (threadIdx.x == 64) listed_total = cv;
  }
  __syncthreads();
  IcpState* st = &lst;
  if (threadIdx.x == 0) {
  if (a.unproven_cnt != nullptr && a.gn_last_step) { st->unproven = unproven_total; st->listed = listed_total; }

  const double n = sums[0];
  double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  bool finalize = false;
  if (a.metric == IM_KABSCH) {
    for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
    kabsch_from_sums(sums, L, t);
    finalize = true;
  } else {
    // transform_estimation.hpp:264-272: no usable terms => tform stays identity, return false
    const bool has_p2p = (n > 0.0) && (a.w_p2p > 0.0f);
    const bool has_p2pl = (n > 0.0) && (a.w_p2pl > 0.0f);
    if (!st->inner_done) {
      if ((!has_p2p && !has_p2pl) || (has_p2pl && !a.has_normals)) {
        st->pad0 = 1;  // identity step
        st->inner_done = 1;
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      } else if (a.gn_zero_steps) {
        // no Gauss-Newton step at all: the inner transform stays the identity and is un-centred below
        st->inner_done = 1;
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      } else {
        // (registers for the usual case -- every index below is a compile-time constant once the loops are unrolled; the pivoted
        //  solve indexes its arrays dynamically: LDS for that one, scratch would be global memory)
        double AtA[36], Atb[6], dth[6];
        gn_normal_equations(sums, has_p2p ? (double)a.w_p2p : 0.0, has_p2pl ? (double)a.w_p2pl : 0.0, AtA, Atb, a.point_weighted != 0);
        if (!ldlt6_solve_fast(AtA, Atb, dth)) {      // (pivoted: rank-deficient systems only)
          __shared__ double sA[36], sb[6], sx[6], wsA[36], wsy[6];
          __shared__ int wsperm[6];
#pragma unroll
          for (int i = 0; i < 36; ++i) sA[i] = AtA[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) sb[i] = Atb[i];
          ldlt6_solve_ws(sA, sb, sx, wsA, wsy, wsperm);
#pragma unroll
          for (int i = 0; i < 6; ++i) dth[i] = sx[i];
        }
        rigid_gn_update(dth, st->dLd, st->dtd);
        for (int i = 0; i < 9; ++i) st->innerL[i] = (float)st->dLd[i];
        for (int i = 0; i < 3; ++i) st->innert[i] = (float)st->dtd[i];
        double nrm = 0.0;
        for (int i = 0; i < 6; ++i) nrm += dth[i] * dth[i];
        if (sqrt(nrm) < (double)a.opt_conv_tol) st->inner_done = 1;     // :360
        for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = sums[i];
      }
    }
    if (a.gn_last_step) {
      if (st->pad0 == 0) {
        for (int i = 0; i < 9; ++i) L[i] = st->dLd[i];
        // tform = t_dst * tform * t_src, t_src = Translation(-(transform_*src_mean_))   :361/:365
        for (int r = 0; r < 3; ++r)
          t[r] = st->dtd[r] - (L[r * 3] * (double)st->smt[0] + L[r * 3 + 1] * (double)st->smt[1] + L[r * 3 + 2] * (double)st->smt[2]) +
                 (double)a.dst_mean[r];
      }
      finalize = true;
    }
  }
  if (finalize) {
    float Tn[16];
    const float delta = compose_update(L, t, st->T, Tn);
    {
      const float step = motion_step_of(st->T, Tn, a.src_center, a.src_half);
      const float prev = st->motion_step;
      st->motion_pred = (prev < INFINITY && prev > 0.0f) ? step * fminf(1.0f, step / prev) : 0.0f;
      st->motion_step = step;
      st->motion_acc = (float)(((double)st->motion_acc + (double)step) * 1.000001);
      st->motion_eps = motion_eps_of(Tn, a.src_center, a.src_half);
    }
    for (int i = 0; i < 16; ++i) { st->Tprev[i] = st->T[i]; st->T[i] = Tn[i]; }
    float mx, my, mz;
    transform_point(Tn, a.src_mean[0], a.src_mean[1], a.src_mean[2], mx, my, mz);
    st->smt[0] = mx; st->smt[1] = my; st->smt[2] = mz;
    st->prev_delta = st->delta;
    st->delta = delta;
    st->iterations += 1;
    st->ncorr = (unsigned long long)(st->sums[0] + 0.5);   // (the sums of the last accumulation that ran: a converged inner loop skips the later ones)
    st->done = (delta < a.conv_tol) ? 1 : 0;                            // icp_base.hpp:83
    reset_inner(st);
    if (a.guard_axis >= 0) {
      // |((T - T_part) p)_axis| over the source's bounding box: an affine function of p, extreme at a corner
      const int ax = a.guard_axis;
      float d = Tn[12 + ax] - a.guard_T[12 + ax], spread = 0.0f;
      for (int j = 0; j < 3; ++j) {
        const float dl = Tn[j * 4 + ax] - a.guard_T[j * 4 + ax];
        d += dl * a.guard_center[j];
        spread += fabsf(dl) * a.guard_half[j];
      }
      if (!(fabsf(d) + spread <= a.guard_slack) && st->slab_violation == 0) {
        st->slab_violation = 1;
        st->violation_iter = st->iterations; st->violation_delta = delta; st->violation_ncorr = st->ncorr;
        for (int i = 0; i < 16; ++i) st->violation_T[i] = Tn[i];
      }
    }
  }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ST_DWORDS; k += 256) reinterpret_cast<uint32_t*>(a.state)[k] = reinterpret_cast<const uint32_t*>(&lst)[k];
  if (a.trace != nullptr && a.gn_last_step && threadIdx.x == 0 && lst.iterations >= 1 && lst.iterations <= RUN_TRACE_CAP)
    a.trace[lst.iterations - 1] = make_uint4(lst.unproven, lst.listed, __float_as_uint(lst.motion_step), __float_as_uint(lst.delta));
  if (a.feedback != nullptr && a.gn_last_step && threadIdx.x == 0) {
    FeedbackSlot* sl = &a.feedback->slot[(unsigned int)lst.iterations & 3u];
    sl->unproven = lst.unproven;
    sl->listed = lst.listed;
    sl->delta = lst.delta;
    sl->prev_delta = lst.prev_delta;
    sl->step = lst.motion_step;
    sl->commit = ((unsigned long long)a.run_tag << 32) | (unsigned long long)(unsigned int)lst.iterations;
    __threadfence_system();
    a.feedback->latest = ((unsigned long long)a.run_tag << 32) | (lst.done ? 0x80000000ull : 0ull) | (unsigned long long)((unsigned int)lst.iterations & 0x7fffffffu);
  }
}
__global__ __launch_bounds__(256) void k_solve(SolveArgs a) { solve_body(a); }

void launch_solve(const SolveArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_solve, dim3(1), dim3(256), 0, s, a);
}

// Stage 1 of the cross-block reduction AND the epilogue in one launch: the G blocks fold their slices as k_reduce_stage1 does (same
// rows, same order: the sums are bitwise the two-kernel path's), publish their row -- a device-scope release: the row leaves this
// XCD's L2 before the ticket is taken -- and the block that takes the LAST ticket (device-scope acquire: its caches are invalidated,
// the other XCDs' rows are read from memory) runs the epilogue over the G rows.  One launch and one kernel boundary less per
// iteration.  The same hand-over inside the 1024-block accumulation kernels was measured three times slower than they are
// (NOTEBOOK.md: every one of their blocks pays the write-back while the others are still streaming); here it is paid by 32 blocks
// whose only stores are their rows.
__global__ __launch_bounds__(256) void k_reduce_solve(const double* __restrict__ partials, int nblocks, double* __restrict__ stage, unsigned int* ticket, SolveArgs a) {
  if (a.state->done) return;      // (read by every block before any block can change it: the epilogue runs after the last ticket)
  {
    __shared__ double rsums[SUMS_MAX];
    const int per = (nblocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = blockIdx.x * per;
    const int b1 = min(b0 + per, nblocks);
    reduce_partials_block(partials + (size_t)b0 * SUMS_MAX, max(b1 - b0, 0), rsums);
    if (threadIdx.x < SUMS_MAX) stage[blockIdx.x * SUMS_MAX + threadIdx.x] = rsums[threadIdx.x];
  }
  __shared__ unsigned int last_block;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    last_block = (t == gridDim.x - 1u) ? 1u : 0u;
    if (t == gridDim.x - 1u) *ticket = 0u;      // (for the next launch: nobody else touches it any more)
  }
  __syncthreads();
  if (last_block == 0u) return;
  __threadfence();
  a.partials = stage;
  a.nblocks = (int)gridDim.x;
  a.reduced = nullptr;
  solve_body(a);
}
// partials[nblocks] -> epilogue.  Few rows: the epilogue folds them itself; many: one launch does both stages (above).
void launch_reduce_and_solve(const double* partials, int nblocks, double* stage, unsigned int* ticket, SolveArgs a, hipStream_t s) {
  if (nblocks <= 64 || ticket == nullptr) {
    const int rows = ticket == nullptr ? launch_reduce_stage1(partials, nblocks, stage, s) : 0;
    a.partials = rows ? stage : partials; a.nblocks = rows ? rows : nblocks; a.reduced = nullptr;
    hipLaunchKernelGGL(k_solve, dim3(1), dim3(256), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_reduce_solve, dim3(reduce_groups(nblocks)), dim3(256), 0, s, partials, nblocks, stage, ticket, a);
}

struct InitArgs { float T[16]; float src_mean[3]; Feedback* fb; unsigned int run_tag; float src_center[3], src_half[3]; unsigned int* tie_counters; };

__global__ void k_init_state(IcpState* st, InitArgs ia) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (ia.tie_counters != nullptr) { for (int k = 0; k < 4; ++k) ia.tie_counters[k] = 0u; }
  if (ia.fb != nullptr) {
    for (int k = 0; k < 4; ++k) { ia.fb->slot[k].unproven = 0u; ia.fb->slot[k].listed = 0u; ia.fb->slot[k].delta = 0.0f; ia.fb->slot[k].prev_delta = 0.0f; ia.fb->slot[k].step = 0.0f; ia.fb->slot[k].pad = 0.0f; ia.fb->slot[k].commit = 0ull; }
    __threadfence_system();
    ia.fb->latest = (unsigned long long)ia.run_tag << 32;
  }
  for (int i = 0; i < 16; ++i) st->T[i] = st->Tprev[i] = ia.T[i];
  float mx, my, mz;
  transform_point(ia.T, ia.src_mean[0], ia.src_mean[1], ia.src_mean[2], mx, my, mz);
  st->smt[0] = mx; st->smt[1] = my; st->smt[2] = mz;
  st->delta = INFINITY;
  st->prev_delta = INFINITY;
  st->iterations = 0;
  st->done = 0;
  st->ncorr = 0;
  for (int i = 0; i < SUMS_MAX; ++i) st->sums[i] = 0.0;
  st->slab_violation = 0; st->unproven = 0; st->listed = 0;
  st->violation_iter = 0; st->violation_delta = 0.0f; st->violation_ncorr = 0ull;
  for (int i = 0; i < 16; ++i) st->violation_T[i] = ia.T[i];
  st->motion_acc = 0.0f; st->motion_step = INFINITY; st->motion_pred = 0.0f;
  st->motion_eps = motion_eps_of(ia.T, ia.src_center, ia.src_half);
  reset_inner(st);
}

void launch_init_state(IcpState* st, const float T0[16], const float src_mean[3], hipStream_t s, Feedback* fb, unsigned int run_tag,
                       const float* src_center, const float* src_half, unsigned int* tie_counters) {
  InitArgs ia;
  ia.fb = fb; ia.run_tag = run_tag; ia.tie_counters = tie_counters;
  // (no bounding box given: a huge one -- the margin test then settles nothing)
  for (int i = 0; i < 3; ++i) { ia.src_center[i] = src_center ? src_center[i] : 0.0f; ia.src_half[i] = src_half ? src_half[i] : 1.0e30f; }
  for (int i = 0; i < 16; ++i) ia.T[i] = T0[i];
  for (int i = 0; i < 3; ++i) ia.src_mean[i] = src_mean[i];
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, s, st, ia);
}

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
