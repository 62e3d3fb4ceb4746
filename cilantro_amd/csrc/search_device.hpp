// search_device.hpp -- what the kernels of kernels.hip / warm.hip / extract.hip share: the search primitives out of global memory
// (keys, margins, shells, the reference's tie order), the accumulation helpers, the per-correspondence vector z of the matrix-core
// rank update.  Device code only (every function inline / forceinline); split from kernels.hip, which has the kernel overview.
#pragma once
#include "internal.hpp"
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>

namespace cilhip {

// Kernel timing without extra packets in the queue: a launcher that supports it attaches the caller's events to its kernels' OWN
// dispatch packets (hipExtLaunchKernelGGL: start of the first kernel, stop of the last) instead of the caller recording events
// around it -- an event recorded between two dependent kernels costs the device ~6 us of idle time each (measured: 11.5 us per
// warm-started iteration of 110).  set_launch_events() arms the NEXT such launcher call of this thread.
inline thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;      // (one pair per thread for all translation units: kernels.hip defines set_launch_events)
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
  static constexpr bool needs_normal = plane || ACC == IM_AFFC;      // (IM_AFFC / IM_AFFP: affine_device.hpp forms the terms; only this flag is read)
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

constexpr int FUSED_WAVE_BYTES = 3584;                       // per-wave scratch of the matrix-core accumulation (64 correspondences x 14 floats): carved from the tile's point buffer (k_search_tiled), a block's own LDS (k_warm)
static_assert(FUSED_WAVE_BYTES >= 64 * 8 * 4 + 64 && FUSED_WAVE_BYTES >= 4 * 64 * 8, "scratch holds the padded 8-float layout and the wave's 16x16 f64 tile");

}  // namespace cilhip
