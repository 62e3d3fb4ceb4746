// knn.hip -- exact k-nearest-neighbour search (k > 1) on the uniform grid, and the consumer that sits right
// before the ICP path: NormalEstimation.  Replaces
//   core/kd_tree.hpp:216-256, :286-318     KDTree::kNNSearch / kNNInRadiusSearch (nanoflann findNeighbors with
//                                           cilantro's KNNSearchResultAdaptor :63-109: k smallest, d2 < r2 strict)
//   core/normal_estimation.hpp:294-420      per point: k-NN -> mean / covariance of the neighbourhood
//                                           (core/covariance.hpp:140-170) -> eigenvector of the smallest eigenvalue,
//                                           optional flip towards the view point, curvature = l0 / (l0+l1+l2)
//
// One lane per query.  The k best candidates live in LDS as a per-lane sorted column of 64-bit keys
// (bits(d2) << 32 | original index).  The grid is searched in expanding Chebyshev shells; a row / cell is skipped when
// its box distance already exceeds the current k-th best, and the search stops when every unscanned point is provably
// farther than the k-th best.  d2 is the pinned ((dx*dx)+(dy*dy))+(dz*dz): neighbour sets and distances are those of the
// reference bit for bit.
// Exactly equal distances.  The reference's result set (core/kd_tree.hpp:80-99: insertion with a strict '>' shift, candidates
// admitted by nanoflann only while dist < worstDist) keeps, among equal distances, the candidates its kd-tree traversal meets
// FIRST, in that order -- inside the list and at the k-th place.  Here (cilhip_knn_set_tie_rule, default 2): the search notices a
// query whose list holds equal distances or whose k-th distance was met on a further point; the order tables of the reference's
// tree over the searched cloud are then built once (csrc/tie_build.hip, as for the 1-NN path), the search runs again with sorted
// POSITIONS in the keys, every group of equal distances inside a list is ordered by tie_before(), and a tied k-th place is refilled
// from ALL points at exactly that distance, first met first.  Rule 0: lowest index (the keys' own order).
// Queries are processed in target-grid cell order (neighbouring lanes scan the same cells).
#include "../../include/cilantro_hip/c_api.h"
#include "internal.hpp"

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cmath>
#include <cstring>
#include <vector>

namespace cilhip {
int g_knn_tie_rule = 2;      // cilhip_knn_set_tie_rule (k-NN lists; kmeans.hip: the kd branch)

namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_MAX_K = 32;
constexpr float KNN_SHRINK = 0.99999905f;   // 1 - 2^-20, as in the 1-NN path

struct KnnArgs {
  GridDev g;
  const float4* queries;   // [nq] {x,y,z,bitcast(original query index)}, in target-grid cell order
  uint32_t nq, k;
  float radius_sq;         // accept d2 < radius_sq (INFINITY: no radius)
  uint32_t* out_idx;       // [nq*k] rows by ORIGINAL query index, ascending (d2, index), NONE-padded; may be null
  float* out_d2;           // [nq*k] or null
  uint32_t* out_cnt;       // [nq] or null
  // normal estimation (do_pca): neighbourhood PCA
  int do_pca;
  const float* ref_xyz;    // [3*n_ref] reference points in ORIGINAL order (gathered by neighbour index)
  float* normals;          // [3*nq] by original query index
  float* curvature;        // [nq] or null
  float vp[3];
  int use_vp;
  // equal distances (see the header comment): tie.mode != 0 -- the reference's order; tie.leaf_slot == null -- only count the queries
  // that would need it (tie_count); by_pos -- the keys carry sorted positions instead of original indices (set with the tables)
  TieDev tie;
  unsigned int* tie_count;
  int by_pos;
};

__device__ __forceinline__ float d2_pinned(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__device__ __forceinline__ float gap(float q, float lo, float hi, float margin) { return fmaxf(fmaxf(lo - q, q - hi) - margin, 0.0f); }

// per-lane sorted list of the k best keys, column `tid` of lists[k][KNN_THREADS]
struct KList {
  unsigned long long* col;   // &lists[threadIdx.x]
  uint32_t k, cnt;
  unsigned long long worst;  // a candidate enters iff key < worst
  unsigned long long none_key;
  uint32_t tie_bits;         // bits of the distance at which a candidate last fell off (or stayed out of) a FULL list while an entry at exactly that
                             // distance stayed in: the k-th place is tied iff this is the final k-th distance (the k-th distance only shrinks)
  __device__ __forceinline__ float worst_d2() const { return __uint_as_float((uint32_t)(worst >> 32)); }
  __device__ __forceinline__ void insert(unsigned long long key) {
    if (key >= worst) { if (cnt == k && (uint32_t)(key >> 32) == (uint32_t)(worst >> 32)) tie_bits = (uint32_t)(key >> 32); return; }
    const bool full = cnt == k;
    const unsigned long long out = full ? col[(size_t)(k - 1) * KNN_THREADS] : 0ull;
    uint32_t j = cnt < k ? cnt : k - 1;
    while (j > 0 && col[(size_t)(j - 1) * KNN_THREADS] > key) { col[(size_t)j * KNN_THREADS] = col[(size_t)(j - 1) * KNN_THREADS]; --j; }
    col[(size_t)j * KNN_THREADS] = key;
    if (cnt < k) ++cnt;
    worst = cnt < k ? none_key : col[(size_t)(k - 1) * KNN_THREADS];
    if (full && (uint32_t)(out >> 32) == (uint32_t)(worst >> 32)) tie_bits = (uint32_t)(out >> 32);
  }
};

__device__ __forceinline__ void scan_run(const float4* __restrict__ pts, uint32_t beg, uint32_t end, float qx, float qy, float qz, KList& L, bool by_pos) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
    const float4 p0 = pts[j], p1 = pts[min(j + 1, last)], p2 = pts[min(j + 2, last)], p3 = pts[min(j + 3, last)];
    const unsigned long long k0 = ((unsigned long long)__float_as_uint(d2_pinned(qx, qy, qz, p0.x, p0.y, p0.z)) << 32) | (by_pos ? j : __float_as_uint(p0.w));
    const unsigned long long k1 = ((unsigned long long)__float_as_uint(d2_pinned(qx, qy, qz, p1.x, p1.y, p1.z)) << 32) | (by_pos ? j + 1 : __float_as_uint(p1.w));
    const unsigned long long k2 = ((unsigned long long)__float_as_uint(d2_pinned(qx, qy, qz, p2.x, p2.y, p2.z)) << 32) | (by_pos ? j + 2 : __float_as_uint(p2.w));
    const unsigned long long k3 = ((unsigned long long)__float_as_uint(d2_pinned(qx, qy, qz, p3.x, p3.y, p3.z)) << 32) | (by_pos ? j + 3 : __float_as_uint(p3.w));
    L.insert(k0);
    if (j + 1 <= last) L.insert(k1);   // (a clamped duplicate must not enter twice)
    if (j + 2 <= last) L.insert(k2);
    if (j + 3 <= last) L.insert(k3);
  }
}

__global__ __launch_bounds__(KNN_THREADS) void k_knn(KnnArgs a) {
  extern __shared__ unsigned long long lists[];   // [k][KNN_THREADS]
  const GridDev& g = a.g;
  const uint32_t qi = blockIdx.x * KNN_THREADS + threadIdx.x;
  if (qi >= a.nq) return;
  const float4 q4 = a.queries[qi];
  const float qx = q4.x, qy = q4.y, qz = q4.z;
  const uint32_t orig = __float_as_uint(q4.w);
  KList L;
  L.col = lists + threadIdx.x;
  L.k = a.k; L.cnt = 0;
  L.none_key = ((unsigned long long)__float_as_uint(a.radius_sq) << 32);
  L.worst = L.none_key;
  L.tie_bits = 0xFFFFFFFFu;
  const bool by_pos = a.by_pos != 0;
  const float BIG = 1.0e9f;
  const int cx = (int)floorf(fminf(fmaxf((qx - g.ox) * g.inv_cell, -BIG), BIG));
  const int cy = (int)floorf(fminf(fmaxf((qy - g.oy) * g.inv_cell, -BIG), BIG));
  const int cz = (int)floorf(fminf(fmaxf((qz - g.oz) * g.inv_cell, -BIG), BIG));
  const bool finite = fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY;   // false for NaN too
  // start at the first shell that can touch the grid (queries far outside would otherwise walk empty shells)
  int s0 = 0;
  s0 = max(s0, max(-cx, cx - (g.nx - 1)));
  s0 = max(s0, max(-cy, cy - (g.ny - 1)));
  s0 = max(s0, max(-cz, cz - (g.nz - 1)));
  if (finite && g.n > 0) {
    for (int s = s0;; ++s) {
      const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1);
      const int y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
      const int xlo = cx - s, xhi = cx + s;
      for (int z = z0; z <= z1; ++z) {
        const bool zface = (z == cz - s) || (z == cz + s);
        const float zl = g.oz + (float)z * g.cell;
        const float gz = gap(qz, zl, zl + g.cell, g.margin);
        const float gz2 = gz * gz;
        if (gz2 * KNN_SHRINK > L.worst_d2()) continue;
        for (int y = y0; y <= y1; ++y) {
          const bool face = zface || (y == cy - s) || (y == cy + s);
          const float yl = g.oy + (float)y * g.cell;
          const float gy = gap(qy, yl, yl + g.cell, g.margin);
          const float gyz2 = gz2 + gy * gy;
          if (gyz2 * KNN_SHRINK > L.worst_d2()) continue;
          const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          if (face) {
            const int xa = max(xlo, 0), xb = min(xhi, g.nx - 1);
            if (xa <= xb) {
              const float gx = gap(qx, g.ox + (float)xa * g.cell, g.ox + (float)(xb + 1) * g.cell, g.margin);
              if ((gyz2 + gx * gx) * KNN_SHRINK <= L.worst_d2()) scan_run(g.pts, g.cell_start[row + xa], g.cell_start[row + xb + 1], qx, qy, qz, L, by_pos);
            }
          } else {
            if (xlo >= 0 && xlo < g.nx) {
              const float xl = g.ox + (float)xlo * g.cell;
              const float gx = gap(qx, xl, xl + g.cell, g.margin);
              if ((gyz2 + gx * gx) * KNN_SHRINK <= L.worst_d2()) scan_run(g.pts, g.cell_start[row + xlo], g.cell_start[row + xlo + 1], qx, qy, qz, L, by_pos);
            }
            if (xhi >= 0 && xhi < g.nx && xhi != xlo) {
              const float xl = g.ox + (float)xhi * g.cell;
              const float gx = gap(qx, xl, xl + g.cell, g.margin);
              if ((gyz2 + gx * gx) * KNN_SHRINK <= L.worst_d2()) scan_run(g.pts, g.cell_start[row + xhi], g.cell_start[row + xhi + 1], qx, qy, qz, L, by_pos);
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
      if (b > 0.0f && L.worst_d2() < b * b * KNN_SHRINK) break;   // (also ends radius searches once the shell passed the radius)
    }
  }
  const uint32_t m = L.cnt;
  if (a.tie.mode != 0 && m > 0) {
    // equal distances inside the list, or at its k-th place (header comment)
    const uint32_t wbits = (uint32_t)(L.col[(size_t)(m - 1) * KNN_THREADS] >> 32);
    const bool boundary = m == a.k && L.tie_bits == wbits;
    bool interior = false;
    for (uint32_t j = 1; j < m; ++j) interior |= (uint32_t)(L.col[(size_t)j * KNN_THREADS] >> 32) == (uint32_t)(L.col[(size_t)(j - 1) * KNN_THREADS] >> 32);
    if (boundary || interior) {
      if (a.tie.leaf_slot == nullptr) atomicAdd(a.tie_count, 1u);
      else {      // (by_pos: the keys' low words are sorted positions)
        uint32_t end = m;
        if (boundary) {
          // the k-th place: every point at exactly the k-th distance competes for the slots [first, k), first met first
          uint32_t first = m - 1;
          while (first > 0 && (uint32_t)(L.col[(size_t)(first - 1) * KNN_THREADS] >> 32) == wbits) --first;
          const uint32_t cap = a.k - first;
          uint32_t filled = 0;
          const float bd = __uint_as_float(wbits);
          const unsigned long long hi = (unsigned long long)wbits << 32;
          for (int s = s0;; ++s) {
            const int z0 = max(cz - s, 0), z1 = min(cz + s, g.nz - 1), y0 = max(cy - s, 0), y1 = min(cy + s, g.ny - 1);
            for (int z = z0; z <= z1; ++z) {
              const float zl = g.oz + (float)z * g.cell;
              const float az = gap(qz, zl, zl + g.cell, g.margin);
              for (int y = y0; y <= y1; ++y) {
                const bool face = (z == cz - s) || (z == cz + s) || (y == cy - s) || (y == cy + s);
                const float yl = g.oy + (float)y * g.cell;
                const float ay = gap(qy, yl, yl + g.cell, g.margin);
                if ((az * az + ay * ay) * KNN_SHRINK > bd) continue;
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
                    if (d2_pinned(qx, qy, qz, p.x, p.y, p.z) != bd) continue;
                    // the adaptor's insertion among equal distances: behind everything met earlier (core/kd_tree.hpp:82-96)
                    uint32_t i = filled;
                    while (i > 0 && tie_before(a.tie, qx, qy, qz, j, (uint32_t)L.col[(size_t)(first + i - 1) * KNN_THREADS])) {
                      if (i < cap) L.col[(size_t)(first + i) * KNN_THREADS] = L.col[(size_t)(first + i - 1) * KNN_THREADS];
                      --i;
                    }
                    if (i < cap) L.col[(size_t)(first + i) * KNN_THREADS] = hi | j;
                    if (filled < cap) ++filled;
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
            if (b > 0.0f && bd < b * b * KNN_SHRINK) break;
          }
          end = first;
        }
        // groups of equal distances inside the list: all their points are in it, ordered as the traversal meets them
        for (uint32_t j = 1; j < end; ++j) {
          const unsigned long long key = L.col[(size_t)j * KNN_THREADS];
          uint32_t i = j;
          while (i > 0 && (uint32_t)(L.col[(size_t)(i - 1) * KNN_THREADS] >> 32) == (uint32_t)(key >> 32) &&
                 tie_before(a.tie, qx, qy, qz, (uint32_t)key, (uint32_t)L.col[(size_t)(i - 1) * KNN_THREADS])) {
            L.col[(size_t)i * KNN_THREADS] = L.col[(size_t)(i - 1) * KNN_THREADS];
            --i;
          }
          L.col[(size_t)i * KNN_THREADS] = key;
        }
      }
    }
  }
  if (by_pos)      // positions -> original indices, for everything that follows
    for (uint32_t j = 0; j < m; ++j) {
      const unsigned long long key = L.col[(size_t)j * KNN_THREADS];
      L.col[(size_t)j * KNN_THREADS] = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)__float_as_uint(g.pts[(uint32_t)key].w);
    }
  if (a.out_cnt) a.out_cnt[orig] = m;
  if (a.out_idx) {
    for (uint32_t j = 0; j < a.k; ++j) {
      const unsigned long long key = j < m ? L.col[(size_t)j * KNN_THREADS] : 0ull;
      a.out_idx[(size_t)orig * a.k + j] = j < m ? (uint32_t)(key & 0xFFFFFFFFull) : NONE_U32;
      if (a.out_d2) a.out_d2[(size_t)orig * a.k + j] = j < m ? __uint_as_float((uint32_t)(key >> 32)) : INFINITY;
    }
  }
  if (a.do_pca) {
    float n0 = NAN, n1 = NAN, n2 = NAN, curv = NAN;
    if (m >= 3) {   // setMinValidSampleSize(3), core/normal_estimation.hpp:28
      // mean: neighbours in ascending-distance order (the order of the reference's result set); f64 accumulation
      double s0d = 0.0, s1d = 0.0, s2d = 0.0;
      for (uint32_t j = 0; j < m; ++j) {
        const size_t id = (size_t)(L.col[(size_t)j * KNN_THREADS] & 0xFFFFFFFFull);
        s0d += (double)a.ref_xyz[3 * id]; s1d += (double)a.ref_xyz[3 * id + 1]; s2d += (double)a.ref_xyz[3 * id + 2];
      }
      const float m0 = (float)(s0d / (double)m), m1 = (float)(s1d / (double)m), m2 = (float)(s2d / (double)m);
      double cs[6] = {0, 0, 0, 0, 0, 0};
      for (uint32_t j = 0; j < m; ++j) {
        const size_t id = (size_t)(L.col[(size_t)j * KNN_THREADS] & 0xFFFFFFFFull);
        const float t0 = __fsub_rn(a.ref_xyz[3 * id], m0), t1 = __fsub_rn(a.ref_xyz[3 * id + 1], m1), t2 = __fsub_rn(a.ref_xyz[3 * id + 2], m2);
        cs[0] += (double)__fmul_rn(t0, t0); cs[1] += (double)__fmul_rn(t0, t1); cs[2] += (double)__fmul_rn(t0, t2);
        cs[3] += (double)__fmul_rn(t1, t1); cs[4] += (double)__fmul_rn(t1, t2); cs[5] += (double)__fmul_rn(t2, t2);
      }
      const double inv = (double)m - 1.0;
      const double C[9] = {cs[0] / inv, cs[1] / inv, cs[2] / inv, cs[1] / inv, cs[3] / inv, cs[4] / inv, cs[2] / inv, cs[4] / inv, cs[5] / inv};
      double w[3], V[9];
      sym_eig3(C, w, V);   // descending; the normal is the eigenvector of the SMALLEST eigenvalue (eigenvectors().col(0) of Eigen's ascending order)
      n0 = (float)V[2]; n1 = (float)V[5]; n2 = (float)V[8];
      if (a.use_vp) {      // normal_estimation.hpp:326-330: flip when it points away from the view point
        const float d = __fadd_rn(__fmul_rn(n0, __fsub_rn(a.vp[0], qx)), __fadd_rn(__fmul_rn(n1, __fsub_rn(a.vp[1], qy)), __fmul_rn(n2, __fsub_rn(a.vp[2], qz))));
        if (d < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
      }
      curv = (float)(w[2] / ((w[0] + w[1]) + w[2]));   // :388
    }
    a.normals[3 * (size_t)orig] = n0; a.normals[3 * (size_t)orig + 1] = n1; a.normals[3 * (size_t)orig + 2] = n2;
    if (a.curvature) a.curvature[orig] = curv;
  }
}

// Radius neighbourhoods (normal_estimation.hpp:120-162 with RadiusNeighborhoodSpecification -> KDTree::radiusSearch,
// nanoflann RadiusResultSet: d2 < r2, strict): the neighbourhood is unbounded, so nothing is listed -- the moments are
// accumulated in two passes over the cells the ball overlaps (f64; the reference sums in ascending-distance order in f32).
__global__ __launch_bounds__(KNN_THREADS) void k_radius_pca(KnnArgs a) {
  const GridDev& g = a.g;
  const uint32_t qi = blockIdx.x * KNN_THREADS + threadIdx.x;
  if (qi >= a.nq) return;
  const float4 q4 = a.queries[qi];
  const float qx = q4.x, qy = q4.y, qz = q4.z;
  const uint32_t orig = __float_as_uint(q4.w);
  float n0 = NAN, n1 = NAN, n2 = NAN, curv = NAN;
  const bool finite = fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY;
  if (finite && g.n > 0 && a.radius_sq > 0.0f) {
    const float r = sqrtf(a.radius_sq) * 1.000001f + g.margin;   // cells the ball can touch (never fewer)
    const float BIG = 1.0e9f;
    const int x0 = max((int)floorf(fminf(fmaxf((qx - r - g.ox) * g.inv_cell, -BIG), BIG)), 0), x1 = min((int)floorf(fminf(fmaxf((qx + r - g.ox) * g.inv_cell, -BIG), BIG)), g.nx - 1);
    const int y0 = max((int)floorf(fminf(fmaxf((qy - r - g.oy) * g.inv_cell, -BIG), BIG)), 0), y1 = min((int)floorf(fminf(fmaxf((qy + r - g.oy) * g.inv_cell, -BIG), BIG)), g.ny - 1);
    const int z0 = max((int)floorf(fminf(fmaxf((qz - r - g.oz) * g.inv_cell, -BIG), BIG)), 0), z1 = min((int)floorf(fminf(fmaxf((qz + r - g.oz) * g.inv_cell, -BIG), BIG)), g.nz - 1);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, cnt = 0.0;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    double cs[6] = {0, 0, 0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
      if (x0 <= x1)
        for (int z = z0; z <= z1; ++z)
          for (int y = y0; y <= y1; ++y) {
            const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
            const uint32_t beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
            for (uint32_t j = beg; j < end; ++j) {
              const float4 p = g.pts[j];
              if (!(d2_pinned(qx, qy, qz, p.x, p.y, p.z) < a.radius_sq)) continue;
              if (pass == 0) { s0 += (double)p.x; s1 += (double)p.y; s2 += (double)p.z; cnt += 1.0; }
              else {
                const float t0 = __fsub_rn(p.x, m0), t1 = __fsub_rn(p.y, m1), t2 = __fsub_rn(p.z, m2);
                cs[0] += (double)__fmul_rn(t0, t0); cs[1] += (double)__fmul_rn(t0, t1); cs[2] += (double)__fmul_rn(t0, t2);
                cs[3] += (double)__fmul_rn(t1, t1); cs[4] += (double)__fmul_rn(t1, t2); cs[5] += (double)__fmul_rn(t2, t2);
              }
            }
          }
      if (pass == 0) {
        if (cnt < 3.0) break;
        m0 = (float)(s0 / cnt); m1 = (float)(s1 / cnt); m2 = (float)(s2 / cnt);
      }
    }
    if (cnt >= 3.0) {
      const double inv = cnt - 1.0;
      const double C[9] = {cs[0] / inv, cs[1] / inv, cs[2] / inv, cs[1] / inv, cs[3] / inv, cs[4] / inv, cs[2] / inv, cs[4] / inv, cs[5] / inv};
      double w[3], V[9];
      sym_eig3(C, w, V);
      n0 = (float)V[2]; n1 = (float)V[5]; n2 = (float)V[8];
      if (a.use_vp) {
        const float d = __fadd_rn(__fmul_rn(n0, __fsub_rn(a.vp[0], qx)), __fadd_rn(__fmul_rn(n1, __fsub_rn(a.vp[1], qy)), __fmul_rn(n2, __fsub_rn(a.vp[2], qz))));
        if (d < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
      }
      curv = (float)(w[2] / ((w[0] + w[1]) + w[2]));
    }
    if (a.out_cnt) a.out_cnt[orig] = (uint32_t)cnt;
  } else if (a.out_cnt) {
    a.out_cnt[orig] = 0;
  }
  a.normals[3 * (size_t)orig] = n0; a.normals[3 * (size_t)orig + 1] = n1; a.normals[3 * (size_t)orig + 2] = n2;
  if (a.curvature) a.curvature[orig] = curv;
}


#define KN_CK(x)               \
  do {                         \
    if ((x) != hipSuccess) {   \
      rc = CILHIP_ERR_HIP;     \
      goto done;               \
    }                          \
  } while (0)

int knn_impl(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem, size_t k, float max_sq_dist,
             uint32_t* idx_out, float* d2_out, uint32_t* cnt_out, bool do_pca, const float* view_point, float* normals_out,
             float* curvature_out) {
  const bool radius_only = do_pca && k == 0;   // unbounded neighbourhood: moments only, no list
  if ((!ref_xyz && n_ref) || (k == 0 && !radius_only) || k > (size_t)KNN_MAX_K || n_ref > 0xFFFFFFF0ull || n_query > 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  if (radius_only && !std::isfinite(max_sq_dist)) return CILHIP_ERR_INVALID;
  if (!(max_sq_dist > 0.0f)) max_sq_dist = 0.0f;   // NaN / negative radius: nothing is inside
  const bool self = query_xyz == nullptr;
  if (self) n_query = n_ref;
  int rc = CILHIP_OK;
  hipStream_t s = nullptr;
  float *d_ref = nullptr, *d_q = nullptr, *d_d2 = nullptr, *d_nrm = nullptr, *d_curv = nullptr;
  bool own_ref = false, own_q = false;
  float4* d_qs = nullptr;
  uint2* d_tiles = nullptr;
  float4* d_tc = nullptr;
  uint32_t *d_idx = nullptr, *d_cnt = nullptr;
  unsigned int* d_tiecnt = nullptr;
  uint2* d_tie_ls = nullptr;
  uint4* d_tie_nodes = nullptr;
  uint32_t *d_tie_leaf = nullptr, *d_tie_slot = nullptr;
  GridBuildResult gr{};
  bool have_grid = false;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    KN_CK(hipSetDevice(device));
    if (n_query == 0) return CILHIP_OK;
    KN_CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (mem == CILHIP_MEM_DEVICE) {
      d_ref = const_cast<float*>(ref_xyz);
      d_q = self ? d_ref : const_cast<float*>(query_xyz);
    } else {
      if (n_ref) {
        own_ref = true;
        KN_CK(hipMalloc(&d_ref, 3 * n_ref * sizeof(float)));
        KN_CK(hipMemcpyAsync(d_ref, ref_xyz, 3 * n_ref * sizeof(float), hipMemcpyHostToDevice, s));
      }
      if (self) d_q = d_ref;
      else {
        own_q = true;
        KN_CK(hipMalloc(&d_q, 3 * n_query * sizeof(float)));
        KN_CK(hipMemcpyAsync(d_q, query_xyz, 3 * n_query * sizeof(float), hipMemcpyHostToDevice, s));
      }
    }
    double mean[3];
    // ~k/4 points per cell: the k-th neighbour then normally lies inside the 3x3x3 block of cells
    KN_CK(build_grid(d_ref, nullptr, (uint32_t)n_ref, s, &gr, mean, std::max(1.0, (double)k / 4.0)));   // (radius-only: 1 point per cell)
    have_grid = true;
    // queries in target-grid cell order (identity transform)
    KN_CK(hipMalloc(&d_qs, n_query * sizeof(float4)));
    {
      const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      float axes[9];
      uint32_t nt = 0;
      KN_CK(sort_source(d_q, (uint32_t)n_query, gr.grid, I, d_qs, s, &d_tiles, &d_tc, axes, &nt));
    }
    KnnArgs a{};
    a.g = gr.grid; a.queries = d_qs; a.nq = (uint32_t)n_query; a.k = (uint32_t)k; a.radius_sq = max_sq_dist;
    if (idx_out) { KN_CK(hipMalloc(&d_idx, n_query * k * sizeof(uint32_t))); a.out_idx = d_idx; }
    if (idx_out && d2_out) { KN_CK(hipMalloc(&d_d2, n_query * k * sizeof(float))); a.out_d2 = d_d2; }
    if (cnt_out) { KN_CK(hipMalloc(&d_cnt, n_query * sizeof(uint32_t))); a.out_cnt = d_cnt; }
    a.do_pca = do_pca ? 1 : 0;
    if (do_pca) {
      KN_CK(hipMalloc(&d_nrm, 3 * n_query * sizeof(float)));
      a.ref_xyz = d_ref; a.normals = d_nrm;
      if (curvature_out) { KN_CK(hipMalloc(&d_curv, n_query * sizeof(float))); a.curvature = d_curv; }
      a.use_vp = 0;
      if (view_point && std::isfinite(view_point[0]) && std::isfinite(view_point[1]) && std::isfinite(view_point[2])) {   // normal_estimation.hpp:366
        a.use_vp = 1;
        for (int i = 0; i < 3; ++i) a.vp[i] = view_point[i];
      }
    }
    if (radius_only)
      hipLaunchKernelGGL(k_radius_pca, dim3((unsigned)((n_query + KNN_THREADS - 1) / KNN_THREADS)), dim3(KNN_THREADS), 0, s, a);
    else {
      // equal distances (header comment): rule 2 -- search, and only if some list needs the reference's order build its tables and search
      // again; rule 1 -- tables first; rule 0 -- the keys' own order (lowest index)
      const int rule = g_knn_tie_rule;
      a.tie = TieDev{}; a.tie.mode = rule != 0 ? 1 : 0; a.by_pos = 0;
      if (rule != 0) { KN_CK(hipMalloc(&d_tiecnt, sizeof(unsigned int))); KN_CK(hipMemsetAsync(d_tiecnt, 0, sizeof(unsigned int), s)); a.tie_count = d_tiecnt; }
      for (int pass = 0; pass < 2; ++pass) {
        bool need_tables = rule == 1 && pass == 0;
        if (!need_tables) {
          hipLaunchKernelGGL(k_knn, dim3((unsigned)((n_query + KNN_THREADS - 1) / KNN_THREADS)), dim3(KNN_THREADS), k * KNN_THREADS * sizeof(unsigned long long), s, a);
          KN_CK(hipGetLastError());
          if (rule != 2 || pass == 1 || a.tie.leaf_slot != nullptr) break;
          unsigned int tied = 0;
          KN_CK(hipMemcpyAsync(&tied, d_tiecnt, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
          KN_CK(hipStreamSynchronize(s));
          if (tied == 0) break;
          need_tables = true;
        }
        if (need_tables && n_ref) {
          // the order tables of the tree the reference builds over the searched cloud (nanoflann 1.7.1, leaf size 10: core/kd_tree.hpp:162-170),
          // built on the device (tie_build.hip)
          size_t nn = 0;
          KN_CK(hipMalloc(&d_tie_ls, n_ref * sizeof(uint2)));
          KN_CK(hipMalloc(&d_tie_leaf, n_ref * sizeof(uint32_t)));
          KN_CK(hipMalloc(&d_tie_slot, n_ref * sizeof(uint32_t)));
          KN_CK(tie_order_build_device(d_ref, nullptr, (uint32_t)n_ref, s, d_tie_leaf, d_tie_slot, &d_tie_nodes, &nn, nullptr));
          launch_tie_tables_by_position(gr.grid.pts, gr.grid.n, d_tie_leaf, d_tie_slot, d_tie_ls, s);
          KN_CK(hipGetLastError());
          a.tie.leaf_slot = d_tie_ls; a.tie.nodes = d_tie_nodes; a.by_pos = 1;
        }
      }
    }
    KN_CK(hipGetLastError());
    if (idx_out) KN_CK(hipMemcpyAsync(idx_out, d_idx, n_query * k * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (idx_out && d2_out) KN_CK(hipMemcpyAsync(d2_out, d_d2, n_query * k * sizeof(float), hipMemcpyDeviceToHost, s));
    if (cnt_out) KN_CK(hipMemcpyAsync(cnt_out, d_cnt, n_query * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (do_pca) KN_CK(hipMemcpyAsync(normals_out, d_nrm, 3 * n_query * sizeof(float), hipMemcpyDeviceToHost, s));
    if (do_pca && curvature_out) KN_CK(hipMemcpyAsync(curvature_out, d_curv, n_query * sizeof(float), hipMemcpyDeviceToHost, s));
    KN_CK(hipStreamSynchronize(s));
  }
done:
  if (have_grid) free_grid(gr.grid);
  if (own_ref && d_ref) (void)hipFree(d_ref);
  if (own_q && d_q) (void)hipFree(d_q);
  if (d_qs) (void)hipFree(d_qs);
  if (d_tiles) (void)hipFree(d_tiles);
  if (d_tc) (void)hipFree(d_tc);
  if (d_idx) (void)hipFree(d_idx);
  if (d_d2) (void)hipFree(d_d2);
  if (d_cnt) (void)hipFree(d_cnt);
  if (d_nrm) (void)hipFree(d_nrm);
  if (d_curv) (void)hipFree(d_curv);
  if (d_tiecnt) (void)hipFree(d_tiecnt);
  if (d_tie_ls) (void)hipFree(d_tie_ls);
  if (d_tie_nodes) (void)hipFree(d_tie_nodes);
  if (d_tie_leaf) (void)hipFree(d_tie_leaf);
  if (d_tie_slot) (void)hipFree(d_tie_slot);
  if (s) (void)hipStreamDestroy(s);
  return rc;
}


// ---- KDTree::radiusSearch as a list-returning call (core/kd_tree.hpp:251-282) ---------------------------------
// Every target point with squared distance < radius (strict, RadiusSearchResultAdaptor :111-142 / nanoflann.hpp:1901),
// per query, ascending by distance.  The reference orders equal distances as std::sort leaves them (unspecified); here
// ties are ordered by index.  Three passes over the cells a query's ball can touch: count, then -- after a prefix sum of
// the counts in the callers' query order -- fill packed (d2, index) keys, then one segmented radix sort of all lists.
__device__ __forceinline__ void ball_cells(const GridDev& g, float qx, float qy, float qz, float radius_sq, int& x0, int& x1, int& y0, int& y1,
                                           int& z0, int& z1) {
  const float r = sqrtf(radius_sq) * 1.000001f + g.margin;   // cells the ball can touch (never fewer)
  const float BIG = 1.0e9f;
  x0 = max((int)floorf(fminf(fmaxf((qx - r - g.ox) * g.inv_cell, -BIG), BIG)), 0); x1 = min((int)floorf(fminf(fmaxf((qx + r - g.ox) * g.inv_cell, -BIG), BIG)), g.nx - 1);
  y0 = max((int)floorf(fminf(fmaxf((qy - r - g.oy) * g.inv_cell, -BIG), BIG)), 0); y1 = min((int)floorf(fminf(fmaxf((qy + r - g.oy) * g.inv_cell, -BIG), BIG)), g.ny - 1);
  z0 = max((int)floorf(fminf(fmaxf((qz - r - g.oz) * g.inv_cell, -BIG), BIG)), 0); z1 = min((int)floorf(fminf(fmaxf((qz + r - g.oz) * g.inv_cell, -BIG), BIG)), g.nz - 1);
}

// FILL = false: counts[orig] = neighbours of the query; FILL = true: keys[offsets[orig] + j] = (bits(d2) << 32) | index
template <bool FILL>
__global__ __launch_bounds__(KNN_THREADS) void k_radius_lists(GridDev g, const float4* __restrict__ queries, uint32_t nq, float radius_sq,
                                                              unsigned long long* __restrict__ counts_or_offsets, unsigned long long* __restrict__ keys) {
  const uint32_t qi = blockIdx.x * KNN_THREADS + threadIdx.x;
  if (qi >= nq) return;
  const float4 q4 = queries[qi];
  const float qx = q4.x, qy = q4.y, qz = q4.z;
  const uint32_t orig = __float_as_uint(q4.w);
  unsigned long long cnt = 0;
  const unsigned long long base = FILL ? counts_or_offsets[orig] : 0ull;
  const bool finite = fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY;
  if (finite && g.n > 0 && radius_sq > 0.0f) {
    int x0, x1, y0, y1, z0, z1;
    ball_cells(g, qx, qy, qz, radius_sq, x0, x1, y0, y1, z0, z1);
    if (x0 <= x1)
      for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
          const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          const uint32_t beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
          for (uint32_t j = beg; j < end; ++j) {
            const float4 p = g.pts[j];
            const float d2 = d2_pinned(qx, qy, qz, p.x, p.y, p.z);
            if (!(d2 < radius_sq)) continue;
            if (FILL) keys[base + cnt] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(p.w);
            ++cnt;
          }
        }
  }
  if (!FILL) counts_or_offsets[orig] = cnt;
}

__global__ void k_unpack_radius(const unsigned long long* __restrict__ keys, size_t total, uint32_t* __restrict__ idx, float* __restrict__ d2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    idx[i] = (uint32_t)k;
    if (d2) d2[i] = __uint_as_float((uint32_t)(k >> 32));
  }
}
__global__ void k_offsets32(const unsigned long long* __restrict__ off64, uint32_t n, uint32_t* __restrict__ off32) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) off32[i] = (uint32_t)off64[i];
}

int radius_impl(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem, float radius_sq,
                uint64_t* offsets_out, uint32_t* idx_out, float* d2_out, size_t capacity, size_t* total_out) {
  if ((!ref_xyz && n_ref) || !offsets_out || n_ref > 0xFFFFFFF0ull || n_query > 0xFFFFFFF0ull || !std::isfinite(radius_sq)) return CILHIP_ERR_INVALID;
  if (!(radius_sq > 0.0f)) radius_sq = 0.0f;
  const bool self = query_xyz == nullptr;
  if (self) n_query = n_ref;
  if (total_out) *total_out = 0;
  int rc = CILHIP_OK;
  hipStream_t s = nullptr;
  float *d_ref = nullptr, *d_q = nullptr, *d_d2 = nullptr;
  bool own_ref = false, own_q = false;
  float4* d_qs = nullptr;
  uint2* d_tiles = nullptr;
  float4* d_tc = nullptr;
  unsigned long long *d_cnt = nullptr, *d_keys = nullptr, *d_keys2 = nullptr;
  uint32_t *d_off32 = nullptr, *d_idx = nullptr;
  void* d_tmp = nullptr;
  GridBuildResult gr{};
  bool have_grid = false;
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CILHIP_ERR_NO_DEVICE;
    KN_CK(hipSetDevice(device));
    offsets_out[0] = 0;
    if (n_query == 0) return CILHIP_OK;
    KN_CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (mem == CILHIP_MEM_DEVICE) {
      d_ref = const_cast<float*>(ref_xyz);
      d_q = self ? d_ref : const_cast<float*>(query_xyz);
    } else {
      if (n_ref) {
        own_ref = true;
        KN_CK(hipMalloc(&d_ref, 3 * n_ref * sizeof(float)));
        KN_CK(hipMemcpyAsync(d_ref, ref_xyz, 3 * n_ref * sizeof(float), hipMemcpyHostToDevice, s));
      }
      if (self) d_q = d_ref;
      else {
        own_q = true;
        KN_CK(hipMalloc(&d_q, 3 * n_query * sizeof(float)));
        KN_CK(hipMemcpyAsync(d_q, query_xyz, 3 * n_query * sizeof(float), hipMemcpyHostToDevice, s));
      }
    }
    double mean[3];
    KN_CK(build_grid(d_ref, nullptr, (uint32_t)n_ref, s, &gr, mean, 2.0));
    have_grid = true;
    KN_CK(hipMalloc(&d_qs, n_query * sizeof(float4)));
    {
      const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      float axes[9];
      uint32_t nt = 0;
      KN_CK(sort_source(d_q, (uint32_t)n_query, gr.grid, I, d_qs, s, &d_tiles, &d_tc, axes, &nt));
    }
    const unsigned nblk = (unsigned)((n_query + KNN_THREADS - 1) / KNN_THREADS);
    KN_CK(hipMalloc(&d_cnt, (n_query + 1) * sizeof(unsigned long long)));
    KN_CK(hipMemsetAsync(d_cnt, 0, (n_query + 1) * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((k_radius_lists<false>), dim3(nblk), dim3(KNN_THREADS), 0, s, gr.grid, (const float4*)d_qs, (uint32_t)n_query, radius_sq, d_cnt,
                       (unsigned long long*)nullptr);
    {  // counts -> offsets (exclusive scan over n_query + 1 entries: the last one is the total), in place
      size_t tmp_bytes = 0;
      KN_CK(rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt, d_cnt, 0ull, n_query + 1, rocprim::plus<unsigned long long>(), s));
      KN_CK(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
      KN_CK(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_cnt, d_cnt, 0ull, n_query + 1, rocprim::plus<unsigned long long>(), s));
      (void)hipFree(d_tmp); d_tmp = nullptr;
    }
    KN_CK(hipMemcpyAsync(offsets_out, d_cnt, (n_query + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    KN_CK(hipStreamSynchronize(s));
    const size_t total = (size_t)offsets_out[n_query];
    if (total_out) *total_out = total;
    if (idx_out && capacity >= total && total > 0) {
      if (total > 0xFFFFFFF0ull) { rc = CILHIP_ERR_UNSUPPORTED; goto done; }   // one segmented sort call takes 32-bit sizes
      KN_CK(hipMalloc(&d_keys, total * sizeof(unsigned long long)));
      KN_CK(hipMalloc(&d_keys2, total * sizeof(unsigned long long)));
      hipLaunchKernelGGL((k_radius_lists<true>), dim3(nblk), dim3(KNN_THREADS), 0, s, gr.grid, (const float4*)d_qs, (uint32_t)n_query, radius_sq, d_cnt, d_keys);
      KN_CK(hipMalloc(&d_off32, (n_query + 1) * sizeof(uint32_t)));
      hipLaunchKernelGGL(k_offsets32, dim3(256), dim3(256), 0, s, (const unsigned long long*)d_cnt, (uint32_t)(n_query + 1), d_off32);
      {
        size_t tmp_bytes = 0;
        KN_CK(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, d_keys, d_keys2, (unsigned int)total, (unsigned int)n_query, d_off32, d_off32 + 1, 0, 64, s));
        KN_CK(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
        KN_CK(rocprim::segmented_radix_sort_keys(d_tmp, tmp_bytes, d_keys, d_keys2, (unsigned int)total, (unsigned int)n_query, d_off32, d_off32 + 1, 0, 64, s));
      }
      KN_CK(hipMalloc(&d_idx, total * sizeof(uint32_t)));
      if (d2_out) KN_CK(hipMalloc(&d_d2, total * sizeof(float)));
      hipLaunchKernelGGL(k_unpack_radius, dim3(2048), dim3(256), 0, s, (const unsigned long long*)d_keys2, total, d_idx, d_d2);
      KN_CK(hipGetLastError());
      KN_CK(hipMemcpyAsync(idx_out, d_idx, total * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      if (d2_out) KN_CK(hipMemcpyAsync(d2_out, d_d2, total * sizeof(float), hipMemcpyDeviceToHost, s));
      KN_CK(hipStreamSynchronize(s));
    }
  }
done:
  if (have_grid) free_grid(gr.grid);
  if (own_ref && d_ref) (void)hipFree(d_ref);
  if (own_q && d_q) (void)hipFree(d_q);
  if (d_qs) (void)hipFree(d_qs);
  if (d_tiles) (void)hipFree(d_tiles);
  if (d_tc) (void)hipFree(d_tc);
  if (d_cnt) (void)hipFree(d_cnt);
  if (d_keys) (void)hipFree(d_keys);
  if (d_keys2) (void)hipFree(d_keys2);
  if (d_off32) (void)hipFree(d_off32);
  if (d_idx) (void)hipFree(d_idx);
  if (d_d2) (void)hipFree(d_d2);
  if (d_tmp) (void)hipFree(d_tmp);
  if (s) (void)hipStreamDestroy(s);
  return rc;
}

}  // namespace
}  // namespace cilhip

extern "C" {

int cilhip_radius_search3f(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem, float radius_sq,
                           uint64_t* offsets_out, uint32_t* idx_out, float* d2_out, size_t capacity, size_t* total_out) {
  return cilhip::radius_impl(device, ref_xyz, n_ref, query_xyz, n_query, mem, radius_sq, offsets_out, idx_out, d2_out, capacity, total_out);
}

int cilhip_knn_set_tie_rule(int rule) {
  if (rule < 0 || rule > 2) return CILHIP_ERR_INVALID;
  cilhip::g_knn_tie_rule = rule;
  return CILHIP_OK;
}

int cilhip_knn3f(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem, size_t k,
                 float max_sq_dist, uint32_t* idx_out, float* d2_out, uint32_t* counts_out) {
  if (!idx_out && !counts_out) return CILHIP_ERR_INVALID;
  return cilhip::knn_impl(device, ref_xyz, n_ref, query_xyz, n_query, mem, k, max_sq_dist, idx_out, d2_out, counts_out, false, nullptr, nullptr,
                          nullptr);
}

int cilhip_normals_radius3f(int device, const float* xyz, size_t n, int mem, float radius_sq, const float* view_point, float* normals_out,
                            float* curvature_out) {
  if (!normals_out && n) return CILHIP_ERR_INVALID;
  return cilhip::knn_impl(device, xyz, n, nullptr, n, mem, 0, radius_sq, nullptr, nullptr, nullptr, true, view_point, normals_out, curvature_out);
}

int cilhip_normals_knn3f(int device, const float* xyz, size_t n, int mem, size_t k, float max_sq_dist, const float* view_point,
                         float* normals_out, float* curvature_out) {
  if (!normals_out && n) return CILHIP_ERR_INVALID;
  return cilhip::knn_impl(device, xyz, n, nullptr, n, mem, k, max_sq_dist, nullptr, nullptr, nullptr, true, view_point, normals_out, curvature_out);
}

}  // extern "C"
