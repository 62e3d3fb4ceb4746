// kmeans.hip -- KMeans<float,3> with the brute-force assignment, on the device (SURVEY.md section 8(f) rank 1;
// BASELINE configs[4]).  Replaces cilantro's clustering/kmeans.hpp:67-194 (cluster_, use_kd_tree = false):
//   assignment   :95-119   argmin_j ||c_j - x_i||^2, strict '<' over ascending j   -> k_assign_accumulate
//   centroid sums :126-131 serial f32 in the reference                              -> exact fixed-point int64
//                                                                                     sums in LDS, flushed with
//                                                                                     integer atomics (order-
//                                                                                     independent => deterministic)
//   empty clusters :134-176, new centroids :179-181, convergence :186-188           -> host, from k*(3+1) values
// The assignment is VALU-bound (n*k distance evaluations, no reuse to tile): two points per lane so the
// distance math runs on packed f32 (v_pk_*), centroids come through the scalar cache (wave-uniform index).
// d2 is (c-x).squaredNorm() with Eigen's 3-term redux pairing d0*d0 + (d1*d1 + d2*d2), no FMA contraction,
// so labels are bit-identical to the reference given identical centroids.
#include "../../include/cilantro_hip/c_api.h"
#include "internal.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int KM_THREADS = 256;
constexpr int KM_TIE_LIST = 1 << 16;   // tied points a pruned kd pass lists for k_fix_ties (more: the pass runs again with the tables)
constexpr int KM_MAX_K = 2048;   // LDS accumulators: k * 4 * 8 B <= 64 KiB of dynamic LDS

struct KmArgs {
  const float* xyz;        // [3n]
  const float* centroids;  // [3k]
  uint32_t n, k;
  uint32_t* labels;        // [n] in: previous, out: new
  long long* sums;         // [k*4] fixed-point sums x,y,z and count
  unsigned int* changed;   // [1]
  double scale;            // 2^S
  int accumulate;
  cilhip::TieDev tie;      // KD only: order tables of the reference's tree over THIS iteration's centroids, by centroid index (leaf_slot null: none)
  unsigned int* tie_count; // KD only, no tables: points whose best distance was met on two centroids are COUNTED here (they keep the lowest index; the host
                           // builds the tables and runs the pass again -- a cloud without exact ties never pays for a tree); null: not counted
  uint2* tie_list;         // ... and (the pruned pass) LISTED: {point, its label before the pass}, so that only they are looked at again (k_fix_ties)
  uint32_t tie_cap;        // entries the list holds (more tied points than that: the pass runs again instead)
};

// KD: the distance the reference's kd-tree branch compares (use_kd_tree = true, kmeans.hpp:86-94: a KDTree over the centroids,
// nanoflann's L2 metric) -- ((dx*dx) + (dy*dy)) + (dz*dz) -- instead of the brute-force branch's Eigen squaredNorm pairing
// dx*dx + (dy*dy + dz*dz) (:107).  Same argmin except where two centroids are equidistant to within that rounding.
// Among centroids at EXACTLY the same distance the tree returns the one its traversal meets first (nanoflann keeps a candidate on a
// strict '<' only): the pass notices such points -- a later block's minimum equal to the best so far, or two hits inside the winning
// block -- and settles them with tie_before() over the order tables of the tree the reference would have built on these centroids
// (tie_build.hip, rebuilt every Lloyd iteration like the reference's KDTree, kmeans.hpp:87).  No tables: the lowest index.
template <bool KD>
__global__ __launch_bounds__(KM_THREADS) void k_assign_accumulate(KmArgs a) {
  extern __shared__ long long lsum[];  // [k*4]
  if (a.accumulate) {
    for (uint32_t t = threadIdx.x; t < a.k * 4; t += KM_THREADS) lsum[t] = 0;   // a.k = padded count
    __syncthreads();
  }
  const uint32_t pairs = (a.n + 1) / 2;
  unsigned int changed = 0;
  for (uint32_t pidx = blockIdx.x * KM_THREADS + threadIdx.x; pidx < pairs; pidx += gridDim.x * KM_THREADS) {
    const uint32_t i0 = 2 * pidx, i1 = min(2 * pidx + 1, a.n - 1);
    const f32x2 px = {a.xyz[3 * (size_t)i0], a.xyz[3 * (size_t)i1]};
    const f32x2 py = {a.xyz[3 * (size_t)i0 + 1], a.xyz[3 * (size_t)i1 + 1]};
    const f32x2 pz = {a.xyz[3 * (size_t)i0 + 2], a.xyz[3 * (size_t)i1 + 2]};
    // argmin with strict '<' over ascending j = the FIRST index that attains the minimum.  Found in two steps so that the inner loop
    // carries one v_min per distance instead of a compare and two selects: (1) per block of 8 centroids the block's minimum (a chain
    // of mins), compared ONCE with the best so far -- strictly less: the earliest block wins ties --, (2) afterwards the winning
    // block is evaluated again and its first centroid at exactly that distance is the label.  The same distances bit for bit.
    f32x2 best = {INFINITY, INFINITY};
    uint32_t c0 = 0, c1 = 0;      // first centroid of the winning block, per point
    bool t0 = false, t1 = false;  // (KD) the best distance was met more than once
    // centroids are padded to a multiple of 8 (pad = +inf: never the minimum); 24 consecutive floats per
    // block of 8 come through the scalar cache with wide s_load's, one wait per 8 candidates
    for (uint32_t j = 0; j < a.k; j += 8) {
      float c[24];
#pragma unroll
      for (int t = 0; t < 24; ++t) c[t] = a.centroids[3 * j + t];
      f32x2 m = {INFINITY, INFINITY};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float cx = c[3 * u], cy = c[3 * u + 1], cz = c[3 * u + 2];
        const f32x2 dx = (f32x2){cx, cx} - px, dy = (f32x2){cy, cy} - py, dz = (f32x2){cz, cz} - pz;
        const f32x2 d = KD ? (dx * dx + dy * dy) + dz * dz : dx * dx + (dy * dy + dz * dz);     // (-ffp-contract=off)
        m.x = fminf(m.x, d.x); m.y = fminf(m.y, d.y);
      }
      if (m.x < best.x) { best.x = m.x; c0 = j; t0 = false; }
      else if (KD && m.x == best.x && m.x < INFINITY) t0 = true;
      if (m.y < best.y) { best.y = m.y; c1 = j; t1 = false; }
      else if (KD && m.y == best.y && m.y < INFINITY) t1 = true;
    }
    uint32_t b0 = 0, b1 = 0;
    {
      // the winning block once more, per point (the two points of a lane usually won in different blocks); nothing below the
      // smallest distance exists, so the first centroid AT it is the argmin; no finite minimum at all (non-finite data): label 0,
      // as a chain of strict compares from (inf, 0) leaves it
      uint32_t f0 = 8, f1 = 8, h0 = 0, h1 = 0;
#pragma unroll
      for (int u = 7; u >= 0; --u) {
        const float ax = a.centroids[3 * (c0 + u)], ay = a.centroids[3 * (c0 + u) + 1], az = a.centroids[3 * (c0 + u) + 2];
        const float bx = a.centroids[3 * (c1 + u)], by = a.centroids[3 * (c1 + u) + 1], bz = a.centroids[3 * (c1 + u) + 2];
        const float dx0 = ax - px.x, dy0 = ay - py.x, dz0 = az - pz.x, dx1 = bx - px.y, dy1 = by - py.y, dz1 = bz - pz.y;
        const float d0 = KD ? (dx0 * dx0 + dy0 * dy0) + dz0 * dz0 : dx0 * dx0 + (dy0 * dy0 + dz0 * dz0);
        const float d1 = KD ? (dx1 * dx1 + dy1 * dy1) + dz1 * dz1 : dx1 * dx1 + (dy1 * dy1 + dz1 * dz1);
        if (d0 == best.x) { f0 = (uint32_t)u; ++h0; }
        if (d1 == best.y) { f1 = (uint32_t)u; ++h1; }
      }
      b0 = (best.x < INFINITY && f0 < 8) ? c0 + f0 : 0u;
      b1 = (best.y < INFINITY && f1 < 8) ? c1 + f1 : 0u;
      if (KD && a.tie.leaf_slot != nullptr) {
        // every centroid at the best distance, the one the reference's traversal reaches first (rare: the loop is scalar per point)
        auto settle = [&](float qx, float qy, float qz, float bd, uint32_t cur) {
          for (uint32_t j = 0; j < a.k; ++j) {
            const float dx = a.centroids[3 * j] - qx, dy = a.centroids[3 * j + 1] - qy, dz = a.centroids[3 * j + 2] - qz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d == bd && j != cur && cilhip::tie_before(a.tie, qx, qy, qz, j, cur)) cur = j;
          }
          return cur;
        };
        if ((t0 || h0 > 1) && best.x < INFINITY) b0 = settle(px.x, py.x, pz.x, best.x, b0);
        if ((t1 || h1 > 1) && best.y < INFINITY) b1 = settle(px.y, py.y, pz.y, best.y, b1);
      } else if (KD && a.tie_count != nullptr) {
        const bool two = (2 * pidx + 1) < a.n;
        if (((t0 || h0 > 1) && best.x < INFINITY) || (two && (t1 || h1 > 1) && best.y < INFINITY)) atomicAdd(a.tie_count, 1u);
      }
    }
    const bool two = (2 * pidx + 1) < a.n;
    changed += (a.labels[i0] != b0) ? 1u : 0u;
    a.labels[i0] = b0;
    if (two) { changed += (a.labels[i1] != b1) ? 1u : 0u; a.labels[i1] = b1; }
    if (a.accumulate) {
      atomicAdd((unsigned long long*)&lsum[b0 * 4 + 0], (unsigned long long)llrint((double)px.x * a.scale));
      atomicAdd((unsigned long long*)&lsum[b0 * 4 + 1], (unsigned long long)llrint((double)py.x * a.scale));
      atomicAdd((unsigned long long*)&lsum[b0 * 4 + 2], (unsigned long long)llrint((double)pz.x * a.scale));
      atomicAdd((unsigned long long*)&lsum[b0 * 4 + 3], 1ull);
      if (two) {
        atomicAdd((unsigned long long*)&lsum[b1 * 4 + 0], (unsigned long long)llrint((double)px.y * a.scale));
        atomicAdd((unsigned long long*)&lsum[b1 * 4 + 1], (unsigned long long)llrint((double)py.y * a.scale));
        atomicAdd((unsigned long long*)&lsum[b1 * 4 + 2], (unsigned long long)llrint((double)pz.y * a.scale));
        atomicAdd((unsigned long long*)&lsum[b1 * 4 + 3], 1ull);
      }
    }
  }
  if (changed) atomicAdd(a.changed, changed);
  if (a.accumulate) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < a.k * 4; t += KM_THREADS)
      if (lsum[t] != 0) atomicAdd((unsigned long long*)&a.sums[t], (unsigned long long)lsum[t]);
  }
}

// ---- the same assignment, PRUNED exactly -------------------------------------------------------------------------------------
// The brute-force pass evaluates n*k distances (5e10 at 50M x 1024: 8 ms, within 15 % of the instruction floor of doing that).  The
// argmin itself only needs the centroids NEAR a point: the k centroids are binned into a uniform grid (about one per cell, built on the
// host every Lloyd iteration -- k is a thousand --, sorted by cell, x fastest), the grid lives in LDS, and a point looks at the 3x3x3
// block of cells around its own: nine runs of the sorted list.  The distance is the brute-force branch's, operation for operation
// (dx*dx + (dy*dy + dz*dz), d = c - x, no contraction), the winner the lexicographic minimum of (distance, index) -- what a chain of
// strict '<' over ascending j leaves.  PROOF of a result: every centroid outside the block lies beyond the block's faces (on the sides
// where the grid goes on), so if the best distance is strictly below (gap - margin)^2 (1 - 2^-20) -- margin = cell / 1024 for the
// rounding of the binning, the factor for the <= 2^-22 relative rounding of a computed distance -- nothing outside can win or tie.
// A point whose block does not prove its result (a centroid-free neighbourhood: clustered centroids, outliers), and every point with a
// non-finite coordinate, takes the exhaustive loop over all k centroids, out of LDS, with the same compare: exactness never depends on
// the grid, only speed does.  Labels are bit-identical to the brute-force kernel's (tests/test_gpu_parity.py: both run, compared).
struct KmGrid { float ox, oy, oz, cell, inv_cell, margin; int g; uint32_t kreal; };

// KD: the kd-tree branch's distance rounding ((dx*dx + dy*dy) + dz*dz, k_assign_accumulate<true>) and its choice among EXACTLY equidistant
// centroids: a point whose best distance was met on a second centroid is settled from the order tables of the reference's tree over the
// centroids (all centroids at exactly that distance, tie_before), or counted when there are none (the host builds them and runs again).
template <bool KD>
__global__ __launch_bounds__(KM_THREADS) void k_assign_grid(KmArgs a, KmGrid gr, const float4* __restrict__ cs_g, const uint32_t* __restrict__ cstart_g) {
  extern __shared__ long long lsum[];                                     // [kpad*4] sums, then the grid
  float4* const cs = reinterpret_cast<float4*>(lsum + (size_t)a.k * 4);   // [kpad + 8] sorted centroids {x, y, z, bits(j)}, +inf pad behind them
  uint32_t* const cstart = reinterpret_cast<uint32_t*>(cs + a.k + 8);     // [g^3 + 1]
  const int g = gr.g, ncell = g * g * g;
  for (uint32_t t = threadIdx.x; t < a.k * 4; t += KM_THREADS) lsum[t] = 0;
  for (uint32_t t = threadIdx.x; t < a.k + 8; t += KM_THREADS) cs[t] = cs_g[t];
  for (int t = threadIdx.x; t <= ncell; t += KM_THREADS) cstart[t] = cstart_g[t];
  __syncthreads();
  unsigned int changed = 0;

  auto finish = [&](uint32_t i, float px, float py, float pz, float bd, uint32_t bj, bool tied) {
    uint32_t b = (bd < INFINITY && bj != 0xFFFFFFFFu) ? bj : 0u;      // (no finite minimum at all: label 0, as a chain of strict compares from (inf, 0) leaves it)
    if (KD && tied && bd < INFINITY && bj != 0xFFFFFFFFu) {
      if (a.tie.leaf_slot != nullptr) {
        // every centroid at exactly the best distance (the flag may be a beaten distance's: then there is one), the first the reference's traversal meets
        for (uint32_t t = 0; t < gr.kreal; ++t) {
          const float4 c = cs[t];
          const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
          const float d = (dx * dx + dy * dy) + dz * dz;
          const uint32_t j = __float_as_uint(c.w);
          if (d == bd && j != b && cilhip::tie_before(a.tie, px, py, pz, j, b)) b = j;
        }
      } else if (a.tie_count != nullptr) {
        const uint32_t slot = atomicAdd(a.tie_count, 1u);
        if (slot < a.tie_cap) a.tie_list[slot] = make_uint2(i, a.labels[i]);
      }
    }
    changed += (a.labels[i] != b) ? 1u : 0u;
    a.labels[i] = b;
    if (a.accumulate) {
      atomicAdd((unsigned long long*)&lsum[b * 4 + 0], (unsigned long long)llrint((double)px * a.scale));
      atomicAdd((unsigned long long*)&lsum[b * 4 + 1], (unsigned long long)llrint((double)py * a.scale));
      atomicAdd((unsigned long long*)&lsum[b * 4 + 2], (unsigned long long)llrint((double)pz * a.scale));
      atomicAdd((unsigned long long*)&lsum[b * 4 + 3], 1ull);
    }
  };
  // a point its 3x3x3 block did not prove (one in seventy): the 5x5x5 block, then -- whatever that does not prove, and every point with a
  // non-finite coordinate -- all k centroids: the definition.  (Queueing these per wave and searching them 64 at a time was measured
  // slower, 1.80 against 1.42 ms: the wide search is not what the pass spends its time on.)
  auto slow = [&](uint32_t i, float px, float py, float pz) {
    float bd = INFINITY;
    uint32_t bj = 0xFFFFFFFFu;
    bool tied = false;
    auto take = [&](const float4 c) {
      const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
      const float d = KD ? (dx * dx + dy * dy) + dz * dz : dx * dx + (dy * dy + dz * dz);
      const uint32_t j = __float_as_uint(c.w);
      if (KD) tied = tied | ((d == bd) & (j != bj) & (d < INFINITY));      // (sticky: finish() looks again, exactly; a row read twice at the grid's edge is the same j)
      const bool better = (d < bd) | ((d == bd) & (j < bj));
      bd = better ? d : bd;
      bj = better ? j : bj;
    };
    bool proven = false;
    if ((fabsf(px) < INFINITY) & (fabsf(py) < INFINITY) & (fabsf(pz) < INFINITY)) {
      const int cx = min(max((int)floorf((px - gr.ox) * gr.inv_cell), 0), g - 1), cy = min(max((int)floorf((py - gr.oy) * gr.inv_cell), 0), g - 1),
                cz = min(max((int)floorf((pz - gr.oz) * gr.inv_cell), 0), g - 1);
      const int xa = max(cx - 2, 0), xb = min(cx + 2, g - 1);
      for (int zc = max(cz - 2, 0); zc <= min(cz + 2, g - 1); ++zc)
        for (int yc = max(cy - 2, 0); yc <= min(cy + 2, g - 1); ++yc) {
          const int row = (zc * g + yc) * g;
          const uint32_t t1 = cstart[row + xb + 1];
          for (uint32_t t = cstart[row + xa]; t < t1; ++t) take(cs[t]);
        }
      float gap = INFINITY;
      if (cx - 2 > 0) gap = fminf(gap, px - (gr.ox + (float)(cx - 2) * gr.cell));
      if (cx + 2 < g - 1) gap = fminf(gap, (gr.ox + (float)(cx + 3) * gr.cell) - px);
      if (cy - 2 > 0) gap = fminf(gap, py - (gr.oy + (float)(cy - 2) * gr.cell));
      if (cy + 2 < g - 1) gap = fminf(gap, (gr.oy + (float)(cy + 3) * gr.cell) - py);
      if (cz - 2 > 0) gap = fminf(gap, pz - (gr.oz + (float)(cz - 2) * gr.cell));
      if (cz + 2 < g - 1) gap = fminf(gap, (gr.oz + (float)(cz + 3) * gr.cell) - pz);
      const float gb = gap - gr.margin;
      proven = gap == INFINITY || (gb > 0.0f && bd < gb * gb * 0.99999905f);
    }
    if (!proven) {
      bd = INFINITY; bj = 0xFFFFFFFFu; tied = false;
      for (uint32_t t = 0; t < gr.kreal; ++t) take(cs[t]);
    }
    finish(i, px, py, pz, bd, bj, tied);
  };

  for (uint32_t i = blockIdx.x * KM_THREADS + threadIdx.x; i < a.n; i += gridDim.x * KM_THREADS) {
    const float px = a.xyz[3 * (size_t)i], py = a.xyz[3 * (size_t)i + 1], pz = a.xyz[3 * (size_t)i + 2];
    float bd = INFINITY;
    uint32_t bj = 0xFFFFFFFFu;
    bool tied = false;
    auto take = [&](const float4 c) {
      const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
      const float d = KD ? (dx * dx + dy * dy) + dz * dz : dx * dx + (dy * dy + dz * dz);      // (-ffp-contract=off: the branch's own value, bit for bit)
      const uint32_t j = __float_as_uint(c.w);
      if (KD) tied = tied | ((d == bd) & (j != bj) & (d < INFINITY));
      const bool better = (d < bd) | ((d == bd) & (j < bj));
      bd = better ? d : bd;
      bj = better ? j : bj;
    };
    bool proven = false;
    const bool finite = (fabsf(px) < INFINITY) & (fabsf(py) < INFINITY) & (fabsf(pz) < INFINITY);
    if (finite) {
      const int cx = min(max((int)floorf((px - gr.ox) * gr.inv_cell), 0), g - 1), cy = min(max((int)floorf((py - gr.oy) * gr.inv_cell), 0), g - 1),
                cz = min(max((int)floorf((pz - gr.oz) * gr.inv_cell), 0), g - 1);
      // Nine runs of the sorted list, in STRAIGHT-LINE code: four records of every run unconditionally -- reading past a short run only
      // evaluates further real centroids (any centroid is a legitimate candidate of the argmin) or the +inf pad behind the list, a row
      // clamped at the grid's edge is read twice (the compare is idempotent) -- so that the 36 reads of a point are independent and the
      // wave executes each instruction once; only runs longer than four go through a loop afterwards.
      const int xa = max(cx - 1, 0), xb = min(cx + 1, g - 1);
      uint32_t r0[9], r1[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int zc = min(max(cz + r / 3 - 1, 0), g - 1), yc = min(max(cy + r % 3 - 1, 0), g - 1);
        const int row = (zc * g + yc) * g;
        r0[r] = cstart[row + xa]; r1[r] = cstart[row + xb + 1];
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const float4 c0 = cs[r0[r]], c1 = cs[r0[r] + 1], c2 = cs[r0[r] + 2], c3 = cs[r0[r] + 3];
        take(c0); take(c1); take(c2); take(c3);
      }
#pragma unroll
      for (int r = 0; r < 9; ++r)
        for (uint32_t t = r0[r] + 4; t < r1[r]; ++t) take(cs[t]);
      // distance from the point to the faces of the block beyond which the grid goes on
      float gap = INFINITY;
      if (cx - 1 > 0) gap = fminf(gap, px - (gr.ox + (float)(cx - 1) * gr.cell));
      if (cx + 1 < g - 1) gap = fminf(gap, (gr.ox + (float)(cx + 2) * gr.cell) - px);
      if (cy - 1 > 0) gap = fminf(gap, py - (gr.oy + (float)(cy - 1) * gr.cell));
      if (cy + 1 < g - 1) gap = fminf(gap, (gr.oy + (float)(cy + 2) * gr.cell) - py);
      if (cz - 1 > 0) gap = fminf(gap, pz - (gr.oz + (float)(cz - 1) * gr.cell));
      if (cz + 1 < g - 1) gap = fminf(gap, (gr.oz + (float)(cz + 2) * gr.cell) - pz);
      const float gb = gap - gr.margin;
      proven = gap == INFINITY || (gb > 0.0f && bd < gb * gb * 0.99999905f);
    }
    if (proven) finish(i, px, py, pz, bd, bj, tied);
    else slow(i, px, py, pz);
  }
  if (changed) atomicAdd(a.changed, changed);
  if (a.accumulate) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < a.k * 4; t += KM_THREADS)
      if (lsum[t] != 0) atomicAdd((unsigned long long*)&a.sums[t], (unsigned long long)lsum[t]);
  }
}

// The tied points of a kd-branch pass, looked at again once the order tables of the reference's tree over the centroids exist: the point's
// nearest centroids once more (all k: a handful of points), the first one the reference's traversal meets among those at exactly the
// smallest distance; where that is not the lowest index the pass left, the label, the two clusters' exact sums and the count of changed
// labels are corrected in place -- integers: the result is what a pass WITH the tables gives, bit for bit.
__global__ void k_fix_ties(KmArgs a, const uint2* __restrict__ list, uint32_t count) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const uint32_t i = list[e].x, prev = list[e].y;
  const float px = a.xyz[3 * (size_t)i], py = a.xyz[3 * (size_t)i + 1], pz = a.xyz[3 * (size_t)i + 2];
  float bd = INFINITY;
  uint32_t bj = 0;
  for (uint32_t j = 0; j < a.k; ++j) {      // (a.k = padded count; the pads are +inf)
    const float dx = a.centroids[3 * j] - px, dy = a.centroids[3 * j + 1] - py, dz = a.centroids[3 * j + 2] - pz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < bd) { bd = d; bj = j; }
  }
  if (!(bd < INFINITY)) return;
  uint32_t cur = bj;
  for (uint32_t j = 0; j < a.k; ++j) {
    const float dx = a.centroids[3 * j] - px, dy = a.centroids[3 * j + 1] - py, dz = a.centroids[3 * j + 2] - pz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d == bd && j != cur && cilhip::tie_before(a.tie, px, py, pz, j, cur)) cur = j;
  }
  if (cur == bj) return;
  a.labels[i] = cur;
  if (a.accumulate) {
    const unsigned long long fx = (unsigned long long)llrint((double)px * a.scale), fy = (unsigned long long)llrint((double)py * a.scale),
                             fz = (unsigned long long)llrint((double)pz * a.scale);
    unsigned long long* const sm = reinterpret_cast<unsigned long long*>(a.sums);
    atomicAdd(&sm[bj * 4 + 0], 0ull - fx); atomicAdd(&sm[bj * 4 + 1], 0ull - fy); atomicAdd(&sm[bj * 4 + 2], 0ull - fz); atomicAdd(&sm[bj * 4 + 3], 0ull - 1ull);
    atomicAdd(&sm[cur * 4 + 0], fx); atomicAdd(&sm[cur * 4 + 1], fy); atomicAdd(&sm[cur * 4 + 2], fz); atomicAdd(&sm[cur * 4 + 3], 1ull);
  }
  const unsigned int was = prev != bj ? 1u : 0u, is = prev != cur ? 1u : 0u;
  if (was != is) atomicAdd(a.changed, is - was);      // (unsigned wrap-around: -1)
}

// The centroids' grid (host; k <= 2048): about one centroid per cell.  false: no usable grid (a non-finite centroid, fewer than 64
// centroids, all of them in one spot) -- the brute-force pass takes the iteration.
static bool build_centroid_grid(const float* c, size_t k, KmGrid& gr, std::vector<float4>& cs, std::vector<uint32_t>& cstart) {
  if (k < 64) return false;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t j = 0; j < k; ++j)
    for (int d = 0; d < 3; ++d) {
      const double v = c[3 * j + d];
      if (!std::isfinite(v)) return false;
      lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v);
    }
  const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  if (!(ext > 0.0) || !std::isfinite(ext)) return false;
  // about ONE centroid per cell (measured best: the pass pays per candidate read)
  static const double per_cell = [] { const char* e = getenv("CILHIP_KMEANS_PER_CELL"); return e ? atof(e) : 1.0; }();      // (dev sweep at 50M x 1024: 0.5: 2.33 ms, 0.7: 1.58, 1: 1.47, 1.5: 1.72, 2: 2.34, 4: 4.14)
  int g = (int)std::lround(std::cbrt((double)k / per_cell));
  g = std::min(std::max(g, 2), 16);
  const double cell = ext / g * 1.0001;      // (the largest coordinate falls inside the last cell)
  gr.ox = (float)lo[0]; gr.oy = (float)lo[1]; gr.oz = (float)lo[2];
  gr.cell = (float)cell; gr.inv_cell = (float)(1.0 / cell); gr.margin = (float)(cell / 1024.0); gr.g = g; gr.kreal = (uint32_t)k;
  // (binned with the grid's own f32 numbers, in double: a centroid lies inside the box of its cell up to the rounding the margin covers)
  const int ncell = g * g * g;
  std::vector<uint32_t> cnt(ncell + 1, 0), cellof(k);
  for (size_t j = 0; j < k; ++j) {
    int ci[3];
    const float o[3] = {gr.ox, gr.oy, gr.oz};
    for (int d = 0; d < 3; ++d) ci[d] = std::min(std::max((int)std::floor(((double)c[3 * j + d] - (double)o[d]) / (double)gr.cell), 0), g - 1);
    cellof[j] = (uint32_t)((ci[2] * g + ci[1]) * g + ci[0]);
    ++cnt[cellof[j] + 1];
  }
  for (int t = 0; t < ncell; ++t) cnt[t + 1] += cnt[t];
  cstart.assign(cnt.begin(), cnt.end());
  std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
  for (size_t j = 0; j < k; ++j) {      // ascending j inside a cell
    float4 v; v.x = c[3 * j]; v.y = c[3 * j + 1]; v.z = c[3 * j + 2];
    const uint32_t jj = (uint32_t)j; std::memcpy(&v.w, &jj, 4);
    cs[fill[cellof[j]]++] = v;
  }
  return true;
}

// farthest member of one cluster from a given point (empty-cluster repair, kmeans.hpp:145-168);
// ties -> lowest point index (the reference's omp-critical order is unspecified)
__global__ void k_farthest_member(const float* __restrict__ xyz, const uint32_t* __restrict__ labels, uint32_t n, uint32_t cluster,
                                  float ox, float oy, float oz, unsigned long long* best) {
  unsigned long long loc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (labels[i] != cluster) continue;
    const float d0 = ox - xyz[3 * (size_t)i], d1 = oy - xyz[3 * (size_t)i + 1], d2 = oz - xyz[3 * (size_t)i + 2];
    const float d = d0 * d0 + (d1 * d1 + d2 * d2);
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - i);
    loc = key > loc ? key : loc;   // +1 offset below keeps "no member" (0) distinguishable
  }
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(loc, off, 64); loc = o > loc ? o : loc; }
  if ((threadIdx.x & 63) == 0 && loc) atomicMax(best, loc);
}

__global__ void k_maxabs_bits(const float* __restrict__ v, size_t count, unsigned int* out) {
  unsigned int m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(fabsf(v[i])));
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned int)__shfl_down((int)m, off, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ void k_zip_tables(const uint32_t* __restrict__ leaf, const uint32_t* __restrict__ slot, uint2* __restrict__ out, uint32_t k) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < k) out[j] = make_uint2(leaf[j], slot[j]);
}

__global__ void k_set_label(uint32_t* labels, uint32_t i, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) labels[i] = v; }

// the pruned pass is the default; cilhip_kmeans_set_pruning(0) (or CILHIP_KMEANS_PRUNE=0 in the environment) keeps the brute-force pass
static bool g_kmeans_prune = [] { const char* e = getenv("CILHIP_KMEANS_PRUNE"); return !(e && e[0] == '0'); }();

}  // namespace

// ---- one shard of the points on one device: the state a Lloyd loop works on ---------------------------------------------------------
// The single-device entry points (kmeans_impl below) drive ONE shard that holds all points; a multi-GPU run (points
// sharded, centroids replicated, all-reduce of k x (3 sums + count)) drives one shard per rank through the same calls
// (cilhip_kmeans_shard_*; cilantro_amd/distributed_models.py is the loop).  The cluster sums are exact fixed-point integers, so the
// all-reduced sums -- and with them the centroids and every later assignment -- are the single-device run's bit for bit.
struct cilhip_kmeans_shard {
  int device = 0;
  hipStream_t s = nullptr;
  float* d_xyz = nullptr;
  bool own_xyz = false;
  size_t n = 0, k = 0, kpad = 0;
  uint64_t index_offset = 0;        // global index of this shard's point 0 (the empty-cluster repair's tie rule: lowest GLOBAL index)
  float* d_c = nullptr;
  uint32_t* d_lab = nullptr;
  long long* d_sums = nullptr;
  unsigned int* d_changed = nullptr;
  unsigned long long* d_best = nullptr;
  float4* d_cs = nullptr;
  uint32_t* d_cstart = nullptr;
  uint32_t* d_tleaf = nullptr;      // kd branch: order tables of the reference's tree over the centroids (leaf, slot by centroid index)
  uint2* d_tls = nullptr;
  uint4* d_tnodes = nullptr;
  uint32_t* d_lab_prev = nullptr;   // kd branch: the labels before a pass (put back when the pass has to run again with the tables)
  uint2* d_tie_list = nullptr;      // kd branch, pruned pass: the tied points {index, label before the pass}
  int tie_table_builds = 0;         // passes that met exact ties (diagnostics)
  std::vector<float4> cs_host;
  std::vector<uint32_t> cstart_host;
  std::vector<float> cpad;

  int init(int dev, const float* xyz, size_t n_, int mem, size_t k_, uint64_t offset) {
    device = dev; n = n_; k = k_; kpad = (k + 7) & ~(size_t)7; index_offset = offset;
    float4 pad4; pad4.x = pad4.y = pad4.z = INFINITY; { const uint32_t none = 0xFFFFFFFFu; std::memcpy(&pad4.w, &none, 4); }
    cs_host.assign(kpad + 8, pad4);
    cpad.assign(3 * kpad, INFINITY);                              // device copy padded with +inf centroids
#define KS_CK(x) do { if ((x) != hipSuccess) return CILHIP_ERR_HIP; } while (0)
    KS_CK(hipSetDevice(device));
    KS_CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (mem == CILHIP_MEM_DEVICE) {
      d_xyz = const_cast<float*>(xyz);
    } else {
      KS_CK(hipMalloc(&d_xyz, (n ? 3 * n : 1) * sizeof(float)));
      own_xyz = true;
      if (n) KS_CK(hipMemcpyAsync(d_xyz, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice, s));
    }
    KS_CK(hipMalloc(&d_c, 3 * kpad * sizeof(float)));
    KS_CK(hipMalloc(&d_lab, (n ? n : 1) * sizeof(uint32_t)));
    KS_CK(hipMalloc(&d_sums, kpad * 4 * sizeof(long long)));
    KS_CK(hipMalloc(&d_changed, 2 * sizeof(unsigned int)));      // [0] labels changed, [1] (kd branch) tied points met without tables
    KS_CK(hipMalloc(&d_best, sizeof(unsigned long long)));
    KS_CK(hipMemsetAsync(d_lab, 0, (n ? n : 1) * sizeof(uint32_t), s));   // point_to_cluster_index_map_.resize(n): zeros (:80)
    return CILHIP_OK;
  }
  // max |coordinate| of the shard (f32; 0 for an empty one)
  int maxabs(float* out) {
    KS_CK(hipSetDevice(device));
    unsigned int hmax = 0;   // max |x| as f32 bits (non-negative floats order like unsigned ints)
    KS_CK(hipMemsetAsync(d_changed, 0, sizeof(unsigned int), s));
    if (n) hipLaunchKernelGGL(k_maxabs_bits, dim3(1024), dim3(256), 0, s, d_xyz, 3 * n, d_changed);
    KS_CK(hipMemcpyAsync(&hmax, d_changed, sizeof(hmax), hipMemcpyDeviceToHost, s));
    KS_CK(hipStreamSynchronize(s));
    std::memcpy(out, &hmax, sizeof(float));
    return CILHIP_OK;
  }
  // one assignment pass over the shard under `centroids` (kmeans.hpp:95-119 / :86-94): labels updated in place, the shard's exact
  // fixed-point sums {x, y, z, count} per cluster (scale 2^S) and the number of labels that changed
  int assign(const float* centroids, double scale, bool kd_order, bool assign_only, long long* sums_out, unsigned int* changed_out) {
    KS_CK(hipSetDevice(device));
    std::memcpy(cpad.data(), centroids, 3 * k * sizeof(float));
    KS_CK(hipMemcpyAsync(d_c, cpad.data(), 3 * kpad * sizeof(float), hipMemcpyHostToDevice, s));
    KS_CK(hipMemsetAsync(d_changed, 0, sizeof(unsigned int), s));
    KS_CK(hipMemsetAsync(d_sums, 0, kpad * 4 * sizeof(long long), s));
    if (n) {
      const int nblocks = (int)std::min<size_t>((n / 2 + KM_THREADS - 1) / KM_THREADS + 1, 1024);
      KmArgs a{d_xyz, d_c, (uint32_t)n, (uint32_t)kpad, d_lab, d_sums, d_changed, scale, assign_only ? 0 : 1, {nullptr, nullptr, nullptr, 0}, nullptr, nullptr, 0u};
      // The kd branch's choice among exactly equidistant centroids needs the order tables of the tree the reference builds over THIS
      // iteration's centroids (kmeans.hpp:87) -- about a millisecond, as much as the whole pruned pass.  So the pass first runs WITHOUT
      // them and counts the points whose best distance was met on two centroids (none on data without exact ties); only then the
      // tables are built, the labels put back (the count of changed labels is against the previous iteration's) and the pass runs again.
      bool finite = true;
      if (kd_order) for (size_t t = 0; t < 3 * k; ++t) finite = finite && std::isfinite(centroids[t]);
      const bool ties_matter = kd_order && cilhip::g_knn_tie_rule != 0 && k > 1 && finite;
      if (ties_matter) {
        if (!d_lab_prev) KS_CK(hipMalloc(&d_lab_prev, n * sizeof(uint32_t)));
        KS_CK(hipMemcpyAsync(d_lab_prev, d_lab, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        KS_CK(hipMemsetAsync(d_changed + 1, 0, sizeof(unsigned int), s));
        a.tie_count = d_changed + 1;
        if (!d_tie_list) KS_CK(hipMalloc(&d_tie_list, (size_t)KM_TIE_LIST * sizeof(uint2)));
        a.tie_list = d_tie_list; a.tie_cap = KM_TIE_LIST;
      }
      KmGrid gr{};
      const bool pruned = g_kmeans_prune && build_centroid_grid(centroids, k, gr, cs_host, cstart_host);
      const size_t ncell1 = pruned ? (size_t)gr.g * gr.g * gr.g + 1 : 0;
      if (pruned) {
        // the grid of THIS iteration's centroids: sorted list + cell table, 20 KB
        if (!d_cs) { KS_CK(hipMalloc(&d_cs, (kpad + 8) * sizeof(float4))); KS_CK(hipMalloc(&d_cstart, (16 * 16 * 16 + 1) * sizeof(uint32_t))); }
        KS_CK(hipMemcpyAsync(d_cs, cs_host.data(), (kpad + 8) * sizeof(float4), hipMemcpyHostToDevice, s));
        KS_CK(hipMemcpyAsync(d_cstart, cstart_host.data(), ncell1 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      }
      auto pass = [&]() {
        if (pruned) {
          const size_t lds = kpad * 4 * sizeof(long long) + (kpad + 8) * sizeof(float4) + ncell1 * sizeof(uint32_t);
          const int nb_g = (int)std::min<size_t>((n + KM_THREADS - 1) / KM_THREADS, 2048);
          if (kd_order) hipLaunchKernelGGL(k_assign_grid<true>, dim3(nb_g), dim3(KM_THREADS), lds, s, a, gr, (const float4*)d_cs, (const uint32_t*)d_cstart);
          else hipLaunchKernelGGL(k_assign_grid<false>, dim3(nb_g), dim3(KM_THREADS), lds, s, a, gr, (const float4*)d_cs, (const uint32_t*)d_cstart);
        }
        else if (kd_order) hipLaunchKernelGGL(k_assign_accumulate<true>, dim3(nblocks), dim3(KM_THREADS), assign_only ? 0 : kpad * 4 * sizeof(long long), s, a);
        else hipLaunchKernelGGL(k_assign_accumulate<false>, dim3(nblocks), dim3(KM_THREADS), assign_only ? 0 : kpad * 4 * sizeof(long long), s, a);
      };
      pass();
      KS_CK(hipGetLastError());
      if (ties_matter) {
        unsigned int tied = 0;
        KS_CK(hipMemcpyAsync(&tied, d_changed + 1, sizeof(tied), hipMemcpyDeviceToHost, s));
        KS_CK(hipStreamSynchronize(s));
        if (tied != 0) {
          if (!d_tleaf) { KS_CK(hipMalloc(&d_tleaf, 2 * kpad * sizeof(uint32_t))); KS_CK(hipMalloc(&d_tls, kpad * sizeof(uint2))); }
          if (d_tnodes) { (void)hipFree(d_tnodes); d_tnodes = nullptr; }
          size_t nn = 0; int depth = 0;
          KS_CK(cilhip::tie_order_build_device(d_c, nullptr, (uint32_t)k, s, d_tleaf, d_tleaf + kpad, &d_tnodes, &nn, &depth));
          hipLaunchKernelGGL(k_zip_tables, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_tleaf, (const uint32_t*)(d_tleaf + kpad), d_tls, (uint32_t)k);
          a.tie.leaf_slot = d_tls; a.tie.nodes = d_tnodes; a.tie.mode = 1; a.tie_count = nullptr;
          ++tie_table_builds;
          if (pruned && tied <= (unsigned int)KM_TIE_LIST) {
            // the pruned pass listed its tied points: only they are looked at again
            hipLaunchKernelGGL(k_fix_ties, dim3((tied + 255u) / 256u), dim3(256), 0, s, a, (const uint2*)d_tie_list, (uint32_t)tied);
          } else {
            KS_CK(hipMemcpyAsync(d_lab, d_lab_prev, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
            KS_CK(hipMemsetAsync(d_changed, 0, sizeof(unsigned int), s));
            KS_CK(hipMemsetAsync(d_sums, 0, kpad * 4 * sizeof(long long), s));
            pass();
          }
          KS_CK(hipGetLastError());
        }
      }
    }
    if (assign_only) return CILHIP_OK;
    unsigned int changed = 0;
    KS_CK(hipMemcpyAsync(&changed, d_changed, sizeof(changed), hipMemcpyDeviceToHost, s));
    KS_CK(hipMemcpyAsync(sums_out, d_sums, k * 4 * sizeof(long long), hipMemcpyDeviceToHost, s));
    KS_CK(hipStreamSynchronize(s));
    *changed_out = changed;
    return CILHIP_OK;
  }
  // farthest member of `cluster` from `center` among the shard's points: key = (bits(d) << 32) | (0xFFFFFFFF - GLOBAL index), 0 = no
  // member; the maximum over shards names the point the reference's sweep keeps (ties: lowest index)
  int farthest(uint32_t cluster, const float center[3], unsigned long long* key_out) {
    KS_CK(hipSetDevice(device));
    KS_CK(hipMemsetAsync(d_best, 0, sizeof(unsigned long long), s));
    if (n) hipLaunchKernelGGL(k_farthest_member, dim3(1024), dim3(256), 0, s, d_xyz, d_lab, (uint32_t)n, cluster, center[0], center[1], center[2], d_best);
    unsigned long long best = 0;
    KS_CK(hipMemcpyAsync(&best, d_best, sizeof(best), hipMemcpyDeviceToHost, s));
    KS_CK(hipStreamSynchronize(s));
    if (best) {      // local -> global index
      const uint32_t li = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
      best = (best & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(index_offset + li));
    }
    *key_out = best;
    return CILHIP_OK;
  }
  // the point with LOCAL index li moves to `cluster`; its coordinates
  int move_point(uint32_t li, uint32_t cluster, float p[3]) {
    if (li >= n) return CILHIP_ERR_INVALID;
    KS_CK(hipSetDevice(device));
    hipLaunchKernelGGL(k_set_label, dim3(1), dim3(64), 0, s, d_lab, li, cluster);
    KS_CK(hipMemcpyAsync(p, d_xyz + 3 * (size_t)li, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    KS_CK(hipStreamSynchronize(s));
    return CILHIP_OK;
  }
  int labels(uint32_t* out) {
    KS_CK(hipSetDevice(device));
    if (n) KS_CK(hipMemcpyAsync(out, d_lab, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    KS_CK(hipStreamSynchronize(s));
    return CILHIP_OK;
  }
#undef KS_CK
  void release() {
    (void)hipSetDevice(device);
    if (s) (void)hipStreamSynchronize(s);
    if (own_xyz && d_xyz) (void)hipFree(d_xyz);
    if (d_c) (void)hipFree(d_c);
    if (d_lab) (void)hipFree(d_lab);
    if (d_sums) (void)hipFree(d_sums);
    if (d_changed) (void)hipFree(d_changed);
    if (d_best) (void)hipFree(d_best);
    if (d_cs) (void)hipFree(d_cs);
    if (d_cstart) (void)hipFree(d_cstart);
    if (d_tleaf) (void)hipFree(d_tleaf);
    if (d_tls) (void)hipFree(d_tls);
    if (d_tnodes) (void)hipFree(d_tnodes);
    if (d_lab_prev) (void)hipFree(d_lab_prev);
    if (d_tie_list) (void)hipFree(d_tie_list);
    if (s) (void)hipStreamDestroy(s);
    d_xyz = nullptr; s = nullptr;
  }
};

namespace {

// the fixed-point scale of the cluster sums: |x| * 2^S < 2^(62 - ceil(log2 n)) so that a whole cluster's sum cannot overflow int64
// (n = ALL points, maxabs = the largest |coordinate| of all of them: every shard of a run uses the same S)
int kmeans_scale_exponent(double maxabs, size_t n) {
  int e = 0;
  (void)std::frexp(maxabs > 0.0 ? maxabs : 1.0, &e);           // maxabs < 2^e
  int nbits = 0;
  while (((size_t)1 << nbits) < n) ++nbits;
  return 62 - nbits - e;
}

int kmeans_impl(int device, const float* xyz, size_t n, int mem, float* centroids, size_t k, size_t max_iter, float tol,
                uint32_t* labels_out, size_t* iterations_out, bool assign_only, bool kd_order = false) {
  if (!xyz || !centroids || k == 0 || n == 0 || n >= 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  if (k > KM_MAX_K) return CILHIP_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CILHIP_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return CILHIP_ERR_INVALID;
  int rc = CILHIP_OK;
  std::vector<long long> hs(k * 4);
  std::vector<float> c_old(3 * k);
  size_t iter = 0;
  cilhip_kmeans_shard sh;
#define KM_RC(x) do { rc = (x); if (rc != CILHIP_OK) goto done; } while (0)
  {
    KM_RC(sh.init(device, xyz, n, mem, k, 0));
    float fmax = 0.0f;
    KM_RC(sh.maxabs(&fmax));
    const double scale = std::ldexp(1.0, kmeans_scale_exponent((double)fmax, n));
    const float tol_sq = tol * tol;
    const size_t rounds = assign_only ? 1 : max_iter;
    while (iter < rounds) {
      unsigned int changed = 0;
      KM_RC(sh.assign(centroids, scale, kd_order, assign_only, hs.data(), &changed));
      if (assign_only) break;
      if (changed == 0 && iter > 0) break;                                            // kmeans.hpp:122
      if (tol > 0.0f) std::memcpy(c_old.data(), centroids, 3 * k * sizeof(float));   // :123
      // empty clusters (:134-176), processed in ascending cluster index like the reference
      for (size_t i = 0; i < k; ++i) {
        if (hs[i * 4 + 3] != 0) continue;
        size_t mx = 0;
        for (size_t j = 1; j < k; ++j) if (hs[j * 4 + 3] > hs[mx * 4 + 3]) mx = j;
        const double cm = (double)hs[mx * 4 + 3];
        const float oc[3] = {(float)((double)hs[mx * 4] / scale / cm), (float)((double)hs[mx * 4 + 1] / scale / cm), (float)((double)hs[mx * 4 + 2] / scale / cm)};
        unsigned long long best = 0;
        KM_RC(sh.farthest((uint32_t)mx, oc, &best));
        const uint32_t mi = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu);
        float p[3];
        KM_RC(sh.move_point(mi, (uint32_t)i, p));
        for (int d = 0; d < 3; ++d) hs[mx * 4 + d] -= (long long)std::llrint((double)p[d] * scale);
        hs[mx * 4 + 3]--; hs[i * 4 + 3]++;   // the reference does not add the point to cluster i's sum (:171-175)
      }
      for (size_t i = 0; i < k; ++i)                                                    // :179-181
        for (int d = 0; d < 3; ++d) centroids[3 * i + d] = (float)((double)hs[i * 4 + d] / scale / (double)hs[i * 4 + 3]);
      ++iter;
      if (tol > 0.0f) {                                                                  // :186-188
        float mxs = 0.0f;
        for (size_t i = 0; i < k; ++i) {
          const float d0 = centroids[3 * i] - c_old[3 * i], d1 = centroids[3 * i + 1] - c_old[3 * i + 1], d2 = centroids[3 * i + 2] - c_old[3 * i + 2];
          const float sq = d0 * d0 + (d1 * d1 + d2 * d2);
          if (sq > mxs) mxs = sq;
        }
        if (mxs < tol_sq) break;
      }
    }
    if (labels_out) KM_RC(sh.labels(labels_out));
  }
done:
#undef KM_RC
  if (iterations_out) *iterations_out = iter;
  sh.release();
  return rc;
}

}  // namespace

extern "C" {

int cilhip_kmeans_set_pruning(int on) { g_kmeans_prune = on != 0; return CILHIP_OK; }

int cilhip_kmeans_shard_create(int device, const float* xyz, size_t n, int mem, size_t k, uint64_t index_offset, cilhip_kmeans_shard** out) {
  if (!out || (!xyz && n) || k == 0 || n >= 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  *out = nullptr;
  if (k > KM_MAX_K) return CILHIP_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CILHIP_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return CILHIP_ERR_INVALID;
  cilhip_kmeans_shard* h = new (std::nothrow) cilhip_kmeans_shard();
  if (!h) return CILHIP_ERR_HIP;
  int rc = CILHIP_ERR_HIP;
  try { rc = h->init(device, xyz, n, mem, k, index_offset); } catch (...) { rc = CILHIP_ERR_HIP; }
  if (rc != CILHIP_OK) { h->release(); delete h; return rc; }
  *out = h;
  return CILHIP_OK;
}
void cilhip_kmeans_shard_destroy(cilhip_kmeans_shard* h) { if (h) { h->release(); delete h; } }
int cilhip_kmeans_shard_maxabs(cilhip_kmeans_shard* h, float* maxabs_out) { return (h && maxabs_out) ? h->maxabs(maxabs_out) : CILHIP_ERR_INVALID; }
int cilhip_kmeans_scale_exponent(double maxabs_all, size_t n_all) { return kmeans_scale_exponent(maxabs_all, n_all); }
int cilhip_kmeans_shard_assign(cilhip_kmeans_shard* h, const float* centroids, int scale_exponent, int use_kd_tree, int64_t* sums_out, uint64_t* changed_out) {
  if (!h || !centroids || !sums_out || !changed_out) return CILHIP_ERR_INVALID;
  unsigned int ch = 0;
  static_assert(sizeof(long long) == sizeof(int64_t), "the fixed-point sums are 64-bit");
  int rc = CILHIP_ERR_HIP;
  try { rc = h->assign(centroids, std::ldexp(1.0, scale_exponent), use_kd_tree != 0, false, reinterpret_cast<long long*>(sums_out), &ch); } catch (...) { rc = CILHIP_ERR_HIP; }
  *changed_out = ch;
  return rc;
}
int cilhip_kmeans_shard_farthest(cilhip_kmeans_shard* h, uint32_t cluster, const float center[3], uint64_t* key_out) {
  if (!h || !center || !key_out) return CILHIP_ERR_INVALID;
  unsigned long long key = 0;
  const int rc = h->farthest(cluster, center, &key);
  *key_out = key;
  return rc;
}
int cilhip_kmeans_shard_move_point(cilhip_kmeans_shard* h, uint64_t global_index, uint32_t to_cluster, float xyz_out[3]) {
  if (!h || !xyz_out || global_index < h->index_offset || global_index - h->index_offset >= h->n) return CILHIP_ERR_INVALID;
  return h->move_point((uint32_t)(global_index - h->index_offset), to_cluster, xyz_out);
}
int cilhip_kmeans_shard_labels(cilhip_kmeans_shard* h, uint32_t* labels_out) { return (h && (labels_out || h->n == 0)) ? h->labels(labels_out) : CILHIP_ERR_INVALID; }

int cilhip_kmeans3f(int device, const float* xyz, size_t n, int mem, float* centroids, size_t k, size_t max_iter, float tol,
                    uint32_t* labels_out, size_t* iterations_out) {
  return kmeans_impl(device, xyz, n, mem, centroids, k, max_iter, tol, labels_out, iterations_out, false);
}

int cilhip_kmeans3f_assign(int device, const float* xyz, size_t n, int mem, const float* centroids, size_t k, uint32_t* labels_out) {
  return kmeans_impl(device, xyz, n, mem, const_cast<float*>(centroids), k, 1, 0.0f, labels_out, nullptr, true);
}

int cilhip_kmeans3f_ex(int device, const float* xyz, size_t n, int mem, float* centroids, size_t k, size_t max_iter, float tol, int use_kd_tree,
                       uint32_t* labels_out, size_t* iterations_out) {
  return kmeans_impl(device, xyz, n, mem, centroids, k, max_iter, tol, labels_out, iterations_out, false, use_kd_tree != 0);
}

int cilhip_kmeans3f_assign_ex(int device, const float* xyz, size_t n, int mem, const float* centroids, size_t k, int use_kd_tree, uint32_t* labels_out) {
  return kmeans_impl(device, xyz, n, mem, const_cast<float*>(centroids), k, 1, 0.0f, labels_out, nullptr, true, use_kd_tree != 0);
}

}  // extern "C"
