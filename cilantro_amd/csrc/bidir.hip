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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "internal.hpp"
#include "rank_update.hpp"

namespace cilhip {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

namespace {

constexpr unsigned long long KEY_INVALID = 0xFFFFFFFFFFFFFFFFull;
inline int nblk(size_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096) + (n == 0); }

struct TfDev { float m[16]; };

// q (packed xyz, ORIGINAL source order: the grid built over it then carries original indices) = T * s with the pinned
// arithmetic of the search kernels -- only for transforms that cannot be searched through their inverse
__global__ void k_transform_original(const float* __restrict__ src_xyz, uint32_t ns, const IcpState* __restrict__ st, float* __restrict__ out) {
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    float qx, qy, qz;
    transform_point(T, src_xyz[3 * (size_t)i], src_xyz[3 * (size_t)i + 1], src_xyz[3 * (size_t)i + 2], qx, qy, qz);
    out[3 * (size_t)i] = qx; out[3 * (size_t)i + 1] = qy; out[3 * (size_t)i + 2] = qz;
  }
}

// the same under a transform the HOST holds (the order tables of the tree the reference builds over the transformed source: c_api.hip)
__global__ void k_transform_original_T(const float* __restrict__ src_xyz, uint32_t ns, TfDev Tf, float* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
    float qx, qy, qz;
    transform_point(Tf.m, src_xyz[3 * (size_t)i], src_xyz[3 * (size_t)i + 1], src_xyz[3 * (size_t)i + 2], qx, qy, qz);
    out[3 * (size_t)i] = qx; out[3 * (size_t)i + 1] = qy; out[3 * (size_t)i + 2] = qz;
  }
}
// ---- reverse search without a per-search index -------------------------------------------------------------------
// The reference rebuilds a kd-tree over the transformed source q = T s for every FIRST_TO_SECOND / BOTH search
// (correspondence_search_kd_tree.hpp:188-190, :209-211).  Here the source is indexed ONCE, in its own coordinates
// (a grid over s, records carrying the ORIGINAL source index), and a target point p is searched through the inverse
// transform: p' = T^-1 p picks the cells, every candidate is still compared by the pinned d2(p, T s) -- so the result is
// the exact argmin over the transformed source (ties: lowest original source index) -- and the geometric bounds are
// mapped back: a point at distance >= D from p' in source space lies at >= smin * D - eps from p in target space
// (smin = smallest singular value of T's linear part; eps covers the f32 rounding of p' and of T s).
struct InvArgs {
  int rigid_on_device;  // the state's transform is a rotation + translation: T^-1 = [L^T | -L^T t] is formed in the kernel (device-resident loops)
  float Ti[16];       // T^-1 (col-major), computed in f64 on the host
  float smin;         // lower bound on the smallest singular value of the linear part (slightly shrunk)
  float eps;          // absolute slack of the mapped bound
};

__device__ __forceinline__ float mapped_bound(float gap, const InvArgs& iv) { return fmaxf(iv.smin * gap - iv.eps, 0.0f); }

// FEAT6: candidates are compared by the 6-D feature distance (point, w v) -- the same value, bit for bit, as the forward search's
// (d6_features is symmetric in its two sides); pf = the target point's feature part w * v_p.  The geometric pruning around this
// scan stays 3-D: d6 >= d3.
template <bool FEAT6>
__device__ __forceinline__ void scan_range_inv(const float4* __restrict__ pts, uint32_t beg, uint32_t end, const float* T, float px, float py, float pz,
                                               unsigned long long& bkey, uint32_t& bpos, uint32_t& tie, const FeatSpec* fs = nullptr, float pfx = 0.f, float pfy = 0.f,
                                               float pfz = 0.f, float pgx = 0.f, float pgy = 0.f, float pgz = 0.f) {
  if (beg >= end) return;
  const uint32_t last = end - 1;
  for (uint32_t j = beg; j < end; j += 4) {
    uint32_t jj[4];
    float4 c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { jj[k] = min(j + (uint32_t)k, last); c[k] = pts[jj[k]]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float qx, qy, qz;
      transform_point(T, c[k].x, c[k].y, c[k].z, qx, qy, qz);
      // nanoflann's L2_Adaptor with the TARGET point as the query and q as the data point: dx = p.x - q.x
      float e;
      if (FEAT6) {
        float fx, fy, fz;
        source_feature(*fs, T, fs->src[jj[k]], fx, fy, fz);
        if (fs->dst2 != nullptr) {      // 9-D: the colour part does not follow the transform
          const float4 sc = fs->src2[jj[k]];
          e = d9_features(px, py, pz, pfx, pfy, pfz, pgx, pgy, pgz, qx, qy, qz, fx, fy, fz, __fmul_rn(fs->w2, sc.x), __fmul_rn(fs->w2, sc.y), __fmul_rn(fs->w2, sc.z));
        } else {
          e = d6_features(px, py, pz, pfx, pfy, pfz, qx, qy, qz, fx, fy, fz);
        }
      } else {
        const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
        e = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      }
      const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(c[k].w);
      // (another source point at exactly the best distance so far -- not the clamped re-read of the same record: settled after the search,
      //  where the candidates at the FINAL distance are enumerated again; a flag raised for a distance that is beaten later costs that look)
      tie |= ((uint32_t)(key >> 32) == (uint32_t)(bkey >> 32) && key != bkey) ? 1u : 0u;
      if (key < bkey) { bkey = key; bpos = jj[k]; }
    }
  }
}

// settling a tie of the reverse search: every source point of [beg, end) whose transformed image is at EXACTLY bd from the target point
// p is a candidate; the first one the reference's traversal of its tree over the transformed source meets (query = p) is kept
__device__ __forceinline__ void scan_range_inv_ties(const float4* __restrict__ pts, uint32_t beg, uint32_t end, const float* T, float px, float py, float pz,
                                                    float bd, const TieDev& tt, uint32_t& cur, uint32_t& ncand) {
  for (uint32_t j = beg; j < end; ++j) {
    const float4 c = pts[j];
    float qx, qy, qz;
    transform_point(T, c.x, c.y, c.z, qx, qy, qz);
    const float dx = __fsub_rn(px, qx), dy = __fsub_rn(py, qy), dz = __fsub_rn(pz, qz);
    const float e = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    if (__float_as_uint(e) == __float_as_uint(bd)) {
      ++ncand;
      if (tt.leaf_slot != nullptr && j != cur && tie_before(tt, px, py, pz, j, cur)) cur = j;
    }
  }
}

// the reverse match of ONE target point (sorted position jd): shells of the source grid around p' = T^-1 p, candidates compared by the
// pinned d2(p, T s); T / iv as set up by inverse_for_state()
template <bool FEAT6>
__device__ __forceinline__ void reverse_search_point(const GridDev& sg, const float4* __restrict__ dst_sorted, uint32_t jd, const float* T, const InvArgs& iv, float max_sq,
                                                     uint32_t* __restrict__ rev_pos, float* __restrict__ rev_d2, const FeatSpec& fs, const TieDev& tt) {
  const float KS = 0.99999905f;
  const float4 p = dst_sorted[jd];
  float pfx = 0.f, pfy = 0.f, pfz = 0.f;
  float pgx = 0.f, pgy = 0.f, pgz = 0.f;
  if (FEAT6) { const float4 v = fs.dst[jd]; pfx = __fmul_rn(fs.w, v.x); pfy = __fmul_rn(fs.w, v.y); pfz = __fmul_rn(fs.w, v.z); }      // the adaptor stores w * v (:90)
  if (FEAT6 && fs.dst2 != nullptr) { const float4 v = fs.dst2[jd]; pgx = __fmul_rn(fs.w2, v.x); pgy = __fmul_rn(fs.w2, v.y); pgz = __fmul_rn(fs.w2, v.z); }
  float sx, sy, sz;
  transform_point(iv.Ti, p.x, p.y, p.z, sx, sy, sz);
  unsigned long long bkey = (unsigned long long)__float_as_uint(max_sq) << 32;
  uint32_t bpos = NONE_U32, tie = 0u;
  const float BIG = 1.0e9f;
  const int cx = (int)floorf(fminf(fmaxf((sx - sg.ox) * sg.inv_cell, -BIG), BIG));
  const int cy = (int)floorf(fminf(fmaxf((sy - sg.oy) * sg.inv_cell, -BIG), BIG));
  const int cz = (int)floorf(fminf(fmaxf((sz - sg.oz) * sg.inv_cell, -BIG), BIG));
  {  // farther than the radius from the whole source grid: nothing to find
    const float gx = fmaxf(fmaxf(sg.ox - sx, sx - (sg.ox + (float)sg.nx * sg.cell)) - sg.margin, 0.0f);
    const float gy = fmaxf(fmaxf(sg.oy - sy, sy - (sg.oy + (float)sg.ny * sg.cell)) - sg.margin, 0.0f);
    const float gz = fmaxf(fmaxf(sg.oz - sz, sz - (sg.oz + (float)sg.nz * sg.cell)) - sg.margin, 0.0f);
    const float lb = mapped_bound(sqrtf(gx * gx + gy * gy + gz * gz), iv);
    if (lb * lb * KS >= max_sq) { rev_pos[jd] = NONE_U32; rev_d2[jd] = max_sq; return; }
  }
  int s = max(0, max(max(-cx, cx - (sg.nx - 1)), max(max(-cy, cy - (sg.ny - 1)), max(-cz, cz - (sg.nz - 1)))));   // first shell that reaches the grid
  for (;; ++s) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, sg.nz - 1);
    const int y0 = max(cy - s, 0), y1 = min(cy + s, sg.ny - 1);
    const int xlo = cx - s, xhi = cx + s;
    for (int z = z0; z <= z1; ++z) {
      const bool zface = (z == cz - s) || (z == cz + s);
      const float zl = sg.oz + (float)z * sg.cell;
      const float gz = fmaxf(fmaxf(zl - sz, sz - (zl + sg.cell)) - sg.margin, 0.0f);
      for (int y = y0; y <= y1; ++y) {
        const bool face = zface || (y == cy - s) || (y == cy + s);
        const float yl = sg.oy + (float)y * sg.cell;
        const float gy = fmaxf(fmaxf(yl - sy, sy - (yl + sg.cell)) - sg.margin, 0.0f);
        const float bd = __uint_as_float((uint32_t)(bkey >> 32));
        {
          const float lb = mapped_bound(sqrtf(gz * gz + gy * gy), iv);
          if (lb * lb * KS > bd) continue;
        }
        const uint32_t row = ((uint32_t)z * (uint32_t)sg.ny + (uint32_t)y) * (uint32_t)sg.nx;
        if (face) {
          const int xa = max(xlo, 0), xb = min(xhi, sg.nx - 1);
          if (xa <= xb) scan_range_inv<FEAT6>(sg.pts, sg.cell_start[row + xa], sg.cell_start[row + xb + 1], T, p.x, p.y, p.z, bkey, bpos, tie, &fs, pfx, pfy, pfz, pgx, pgy, pgz);
        } else {
          if (xlo >= 0 && xlo < sg.nx)
            scan_range_inv<FEAT6>(sg.pts, sg.cell_start[row + xlo], sg.cell_start[row + xlo + 1], T, p.x, p.y, p.z, bkey, bpos, tie, &fs, pfx, pfy, pfz, pgx, pgy, pgz);
          if (xhi >= 0 && xhi < sg.nx)
            scan_range_inv<FEAT6>(sg.pts, sg.cell_start[row + xhi], sg.cell_start[row + xhi + 1], T, p.x, p.y, p.z, bkey, bpos, tie, &fs, pfx, pfy, pfz, pgx, pgy, pgz);
        }
      }
    }
    // lower bound (source space) on the distance to anything outside the (2s+1)^3 block and inside the grid
    float b = INFINITY;
    if (cx - s > 0) b = fminf(b, sx - (sg.ox + (float)(cx - s) * sg.cell));
    if (cx + s + 1 < sg.nx) b = fminf(b, (sg.ox + (float)(cx + s + 1) * sg.cell) - sx);
    if (cy - s > 0) b = fminf(b, sy - (sg.oy + (float)(cy - s) * sg.cell));
    if (cy + s + 1 < sg.ny) b = fminf(b, (sg.oy + (float)(cy + s + 1) * sg.cell) - sy);
    if (cz - s > 0) b = fminf(b, sz - (sg.oz + (float)(cz - s) * sg.cell));
    if (cz + s + 1 < sg.nz) b = fminf(b, (sg.oz + (float)(cz + s + 1) * sg.cell) - sz);
    if (b == INFINITY) break;  // the block covers the grid: everything scanned
    const float lb = mapped_bound(b - sg.margin, iv);
    if (lb > 0.0f && __uint_as_float((uint32_t)(bkey >> 32)) < lb * lb * KS) break;
  }
  if (!FEAT6 && tt.mode != 0 && tie != 0u && bpos != NONE_U32) {
    // option "tie_rule": the same shells again with the distance fixed (a row or a shell AT the distance is looked at: the bounds are strict).
    // The flag may be stale (raised for a distance that was beaten later): the candidates at the FINAL distance are counted either way --
    // without the tables a target point with two or more of them is reported (counters[3]) and keeps the lowest source index.
    {
      const float bd = __uint_as_float((uint32_t)(bkey >> 32));
      uint32_t cur = bpos, ncand = 0u;
      for (int s2 = max(0, max(max(-cx, cx - (sg.nx - 1)), max(max(-cy, cy - (sg.ny - 1)), max(-cz, cz - (sg.nz - 1)))));; ++s2) {
        const int z0 = max(cz - s2, 0), z1 = min(cz + s2, sg.nz - 1);
        const int y0 = max(cy - s2, 0), y1 = min(cy + s2, sg.ny - 1);
        const int xlo = cx - s2, xhi = cx + s2;
        for (int z = z0; z <= z1; ++z) {
          const bool zface = (z == cz - s2) || (z == cz + s2);
          const float zl = sg.oz + (float)z * sg.cell;
          const float gz = fmaxf(fmaxf(zl - sz, sz - (zl + sg.cell)) - sg.margin, 0.0f);
          for (int y = y0; y <= y1; ++y) {
            const bool face = zface || (y == cy - s2) || (y == cy + s2);
            const float yl = sg.oy + (float)y * sg.cell;
            const float gy = fmaxf(fmaxf(yl - sy, sy - (yl + sg.cell)) - sg.margin, 0.0f);
            const float lbr = mapped_bound(sqrtf(gz * gz + gy * gy), iv);
            if (lbr * lbr * KS > bd) continue;
            const uint32_t row = ((uint32_t)z * (uint32_t)sg.ny + (uint32_t)y) * (uint32_t)sg.nx;
            if (face) {
              const int xa = max(xlo, 0), xb = min(xhi, sg.nx - 1);
              if (xa <= xb) scan_range_inv_ties(sg.pts, sg.cell_start[row + xa], sg.cell_start[row + xb + 1], T, p.x, p.y, p.z, bd, tt, cur, ncand);
            } else {
              if (xlo >= 0 && xlo < sg.nx) scan_range_inv_ties(sg.pts, sg.cell_start[row + xlo], sg.cell_start[row + xlo + 1], T, p.x, p.y, p.z, bd, tt, cur, ncand);
              if (s2 > 0 && xhi >= 0 && xhi < sg.nx) scan_range_inv_ties(sg.pts, sg.cell_start[row + xhi], sg.cell_start[row + xhi + 1], T, p.x, p.y, p.z, bd, tt, cur, ncand);
            }
          }
        }
        float b2 = INFINITY;
        if (cx - s2 > 0) b2 = fminf(b2, sx - (sg.ox + (float)(cx - s2) * sg.cell));
        if (cx + s2 + 1 < sg.nx) b2 = fminf(b2, (sg.ox + (float)(cx + s2 + 1) * sg.cell) - sx);
        if (cy - s2 > 0) b2 = fminf(b2, sy - (sg.oy + (float)(cy - s2) * sg.cell));
        if (cy + s2 + 1 < sg.ny) b2 = fminf(b2, (sg.oy + (float)(cy + s2 + 1) * sg.cell) - sy);
        if (cz - s2 > 0) b2 = fminf(b2, sz - (sg.oz + (float)(cz - s2) * sg.cell));
        if (cz + s2 + 1 < sg.nz) b2 = fminf(b2, (sg.oz + (float)(cz + s2 + 1) * sg.cell) - sz);
        if (b2 == INFINITY) break;
        const float lb2 = mapped_bound(b2 - sg.margin, iv);
        if (lb2 > 0.0f && bd < lb2 * lb2 * KS) break;
      }
      if (ncand >= 2u) {
        if (tt.leaf_slot == nullptr) atomicAdd(tt.counters + 3, 1u);
        else { atomicAdd(tt.counters + 1, 1u); if (cur != bpos) atomicAdd(tt.counters + 2, 1u); }
      }
      bpos = cur;
    }
  }
  rev_pos[jd] = bpos;
  rev_d2[jd] = __uint_as_float((uint32_t)(bkey >> 32));
}

// the state's transform and, for the device-resident (rigid) loops, its inverse [L^T | -L^T t]
__device__ __forceinline__ void inverse_for_state(const IcpState* __restrict__ st, float* T, InvArgs& iv) {
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T[k];
  if (iv.rigid_on_device) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) iv.Ti[c * 4 + r] = T[r * 4 + c];                                   // L^T
      iv.Ti[12 + r] = -(T[r * 4 + 0] * T[12] + T[r * 4 + 1] * T[13] + T[r * 4 + 2] * T[14]);       // -L^T t
    }
    iv.Ti[3] = iv.Ti[7] = iv.Ti[11] = 0.0f; iv.Ti[15] = 1.0f;
  }
}

template <bool FEAT6>
__global__ __launch_bounds__(256) void k_reverse_search(GridDev sg /*grid over the source, source space*/, const float4* __restrict__ dst_sorted, uint32_t nd,
                                                        const IcpState* __restrict__ st, InvArgs iv, float max_sq, uint32_t* __restrict__ rev_pos,
                                                        float* __restrict__ rev_d2, FeatSpec fs, TieDev tt) {
  if (st->done) return;
  float T[16];
  inverse_for_state(st, T, iv);
  for (uint32_t jd = blockIdx.x * blockDim.x + threadIdx.x; jd < nd; jd += gridDim.x * blockDim.x)
    reverse_search_point<FEAT6>(sg, dst_sorted, jd, T, iv, max_sq, rev_pos, rev_d2, fs, tt);
}

// The reverse search WARM-STARTED from the previous iteration's reverse matches (the device-resident FIRST_TO_SECOND / BOTH loops, from
// their second iteration on).  A target point p whose old match s_i still satisfies |p - T s_i| < half the distance from T s_i to its
// nearest other transformed source point has s_i as its one nearest source point (any other T s_j is at least nnd - |p - T s_i| away):
// no cell is looked at.  nnd in target space from a table over the SOURCE grid (k_self_nn: a lower bound on the squared distance to the
// nearest other source point, 0 for a duplicate): |T s_i - T s_j| >= smin |s_i - s_j| - 2 eps_q for the computed images (eps_q =
// IcpState::motion_eps: the rounding error of one computed T s).  In pinned arithmetic: m = smin sqrt(safe2) - 2 eps_q (rounded down),
// settled iff m > 0 and 4 e (1 + 1e-5) < m^2 with e = the pinned d2(p, T s_i) -- the value the full search would return for that pair;
// the strict inequality excludes ties.  Everything else (no old match, test failed) is LISTED in LDS and searched by the block's
// lanes densely packed -- the full shell search above, so the result is the exact argmin either way.
// Wave-centric: every wave lists its own unsettled points (ballot order: the same list in every run) and searches its list after
// RW_ROUNDS rounds -- no block-wide barrier in the loop.  ACC != IM_NONE: the first Gauss-Newton step's sums over the reverse matches
// are accumulated HERE, on the matrix cores (rank_update.hpp: the terms of k_warm / the tiles), for the settled points as they stream by
// and for the listed ones after their search -- one pass over the target instead of search + k_acc_reverse; mode (k_acc_reverse's): 1 = all
// reverse matches, 2 = those that are not reciprocal duplicates of a forward match, 3 = only those; one row of SUMS_MAX sums per block.
constexpr int RW_ROUNDS = 16;                  // rounds between two searches of a wave's list
constexpr int RW_WCAP = RW_ROUNDS * 64;        // ... which therefore holds all of the wave's points if need be
constexpr int RW_WAVES = 4;
struct RevAcc {
  float dst_mean[3];
  const float4* dst_nrm;        // target normals by sorted position (plane terms)
  int mode;
  const uint32_t* fwd_pos;      // forward matches by sorted source position (modes 2 / 3)
  const uint32_t* src_inv;      // original source index -> sorted source position
  const uint32_t* grid_to_sorted;   // ... composed with the source grid's order: source-grid position -> sorted source position (gathered beside the record itself)
  double* partials;             // [gridDim.x * SUMS_MAX]
};
template <int ACC>
__global__ __launch_bounds__(256) void k_reverse_warm(GridDev sg, const float4* __restrict__ dst_sorted, uint32_t nd, const IcpState* __restrict__ st, InvArgs iv, float max_sq,
                                                      uint32_t* __restrict__ rev_pos, float* __restrict__ rev_d2, const float* __restrict__ src_safe2, TieDev tt, RevAcc ra) {
  if (st->done) return;
  float T[16];
  inverse_for_state(st, T, iv);
  const float eps2 = st->motion_eps * 2.0002f;
  const float smt[3] = {st->smt[0], st->smt[1], st->smt[2]};
  __shared__ uint32_t list[RW_WAVES * RW_WCAP];
  __shared__ __attribute__((aligned(16))) unsigned char raw[ACC != IM_NONE ? RW_WAVES * FUSED_WAVE_BYTES : 16];
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  uint32_t* const wl = list + wave * RW_WCAP;
  float* const zb = reinterpret_cast<float*>(raw) + (ACC != IM_NONE ? wave * (FUSED_WAVE_BYTES / 4) : 0);
  WaveRank<ACC == IM_NONE ? IM_KABSCH : ACC> rank;
  constexpr bool NRM = FusedZ<ACC>::needs_normal;
  const FeatSpec none{};
  const uint32_t rounds = (nd + 255u) / 256u, last = nd - 1u;
  // a correspondence (target position jd, source-grid record c) counts under `mode`
  // (evaluated by every lane, from a valid record: no divergent branch around the two dependent loads)
  auto counts = [&](uint32_t jd, uint32_t sorted_pos) -> bool {
    if (ra.mode < 2) return true;      // (uniform)
    const bool dup = ra.fwd_pos[sorted_pos] == jd;
    return dup == (ra.mode == 3);
  };
  uint32_t wcnt = 0;      // (wave-uniform)
  auto flush = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k0 = 0; k0 < wcnt; k0 += 64u) {
      const bool active = k0 + (uint32_t)lane < wcnt;
      const uint32_t jd = wl[min(k0 + (uint32_t)lane, (uint32_t)(RW_WCAP - 1))];
      if (active) reverse_search_point<false>(sg, dst_sorted, jd, T, iv, max_sq, rev_pos, rev_d2, none, tt);
      if (ACC != IM_NONE) {
        const uint32_t rp = active ? rev_pos[jd] : NONE_U32;      // (the lane's own store)
        const bool has0 = rp != NONE_U32;
        const float4 c = sg.pts[has0 ? rp : 0u];
        const uint32_t sp = ra.mode >= 2 ? ra.grid_to_sorted[has0 ? rp : 0u] : 0u;
        const float4 p = dst_sorted[active ? jd : 0u];
        float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NRM) nv = ra.dst_nrm[active ? jd : 0u];
        float qx, qy, qz;
        transform_point(T, c.x, c.y, c.z, qx, qy, qz);
        const bool cnt = counts(jd, sp);
        rank.update(zb, lane, has0 && cnt, qx, qy, qz, p, nv, ra.dst_mean, smt);
      }
    }
    wcnt = 0;
    __builtin_amdgcn_wave_barrier();
  };
  // Software-pipelined by one round: the coalesced loads of the NEXT round (point, old match, normal) leave right behind the two gathers
  // the current round depends on (the old match's source record and its table entry), so that a wave always has two rounds' trips in flight.
  uint32_t since = 0;
  uint32_t r = blockIdx.x;
  float4 pN = make_float4(0.f, 0.f, 0.f, 0.f), nvN = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t rpN = NONE_U32;
  if (r < rounds) {
    const uint32_t jc = min(r * 256u + threadIdx.x, last);
    pN = dst_sorted[jc]; rpN = rev_pos[jc];
    if (ACC != IM_NONE && NRM) nvN = ra.dst_nrm[jc];
  }
  for (; r < rounds; r += gridDim.x) {
    if (since == (uint32_t)RW_ROUNDS) { flush(); since = 0; }
    ++since;
    const uint32_t jd = r * 256u + threadIdx.x;
    const bool valid = jd < nd;
    const float4 p = pN, nv = nvN;
    const uint32_t rp = rpN;
    const bool has = valid && rp != NONE_U32;
    const uint32_t rc = has ? rp : 0u;
    const float4 c = sg.pts[rc];
    const float sf = src_safe2[rc];
    const uint32_t sp = (ACC != IM_NONE && ra.mode >= 2) ? ra.grid_to_sorted[rc] : 0u;      // (uniform condition; the same trip as the record)
    if (r + gridDim.x < rounds) {      // (uniform)
      const uint32_t jn = min((r + gridDim.x) * 256u + threadIdx.x, last);
      pN = dst_sorted[jn]; rpN = rev_pos[jn];
      if (ACC != IM_NONE && NRM) nvN = ra.dst_nrm[jn];
    }
    float qx, qy, qz;
    transform_point(T, c.x, c.y, c.z, qx, qy, qz);
    const float dx = __fsub_rn(p.x, qx), dy = __fsub_rn(p.y, qy), dz = __fsub_rn(p.z, qz);
    const float e = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float m = iv.smin * sqrtf(sf) * 0.99999f - eps2;
    const bool settled = has && m > 0.0f && 4.0f * e * 1.00001f < m * m && e < max_sq;
    if (settled) rev_d2[jd] = e;
    const bool todo = valid && !settled;
    const unsigned long long um = __ballot(todo);
    if (todo) wl[wcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u))] = jd;
    wcnt += (uint32_t)__popcll(um);
    if (ACC != IM_NONE) { const bool cnt = counts(jd, sp); rank.update(zb, lane, settled && cnt, qx, qy, qz, p, nv, ra.dst_mean, smt); }
  }
  flush();
  if (ACC != IM_NONE) rank.template write_row<RW_WAVES>(raw, wave, lane, ra.partials + (size_t)blockIdx.x * SUMS_MAX);
}

// candidate slots [0, nd): reverse matches (target point jd -> nearest transformed source point)
// (poss = ORIGINAL source index of the pair throughout this file: the pair view is gathered from the original arrays)
__global__ void k_cand_reverse(const float4* __restrict__ dst_sorted, uint32_t nd, const uint32_t* __restrict__ rev_pos,
                               const float* __restrict__ rev_d2, const float4* __restrict__ sgrid_pts /*w = original source index*/,
                               unsigned long long* keys, uint32_t* slots, uint32_t* posd, uint32_t* poss, float* d2) {
  for (uint32_t jd = blockIdx.x * blockDim.x + threadIdx.x; jd < nd; jd += gridDim.x * blockDim.x) {
    const uint32_t rp = rev_pos[jd];
    unsigned long long key = KEY_INVALID;
    uint32_t so = NONE_U32;
    if (rp != NONE_U32) {
      so = __float_as_uint(sgrid_pts[rp].w);
      key = ((unsigned long long)__float_as_uint(dst_sorted[jd].w) << 32) | (unsigned long long)so;
    }
    keys[jd] = key; slots[jd] = jd;
    posd[jd] = jd; poss[jd] = so; d2[jd] = rev_d2[jd];
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
    posd[t] = pos; poss[t] = __float_as_uint(src_sorted[i].w); d2[t] = nn_d2[i];
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

__global__ void k_gather_pair_view(const float* __restrict__ src_xyz, const float* __restrict__ src_nrm, const uint32_t* __restrict__ poss,
                                   uint32_t n, float4* src_view, float4* nrm_view) {
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const size_t o = poss[t];
    src_view[t] = make_float4(src_xyz[3 * o], src_xyz[3 * o + 1], src_xyz[3 * o + 2], __uint_as_float((uint32_t)o));
    if (nrm_view) nrm_view[t] = make_float4(src_nrm[3 * o], src_nrm[3 * o + 1], src_nrm[3 * o + 2], 0.0f);
  }
}

}  // namespace

void launch_transform_original_host_T(const float* d_src_xyz, uint32_t ns, const float T[16], float* d_out, hipStream_t s) {
  TfDev tf;
  for (int k = 0; k < 16; ++k) tf.m[k] = T[k];
  hipLaunchKernelGGL(k_transform_original_T, dim3(nblk(ns)), dim3(256), 0, s, d_src_xyz, ns, tf, d_out);
}


void free_pairs(PairSet& p) {
  uint32_t** u[] = {&p.first, &p.second, &p.posd, &p.poss, &p.first2, &p.second2, &p.posd2, &p.poss2};
  for (auto q : u) { if (*q) (void)hipFree(*q); *q = nullptr; }
  if (p.d2) (void)hipFree(p.d2);
  if (p.d2b) (void)hipFree(p.d2b);
  if (p.src_view) (void)hipFree(p.src_view);
  if (p.nrm_view) (void)hipFree(p.nrm_view);
  p.d2 = p.d2b = nullptr; p.src_view = p.nrm_view = nullptr;
  p.cap = 0; p.count = 0;
  for (void*& w : p.ws) { if (w) (void)hipFree(w); w = nullptr; }
  p.ws_cand = 0; p.ws_tmp_bytes = 0; p.ws_nd = 0; p.ws_ns = 0;
}

static hipError_t ensure_pairs(PairSet& p, size_t cap, bool with_normals) {
  if (cap <= p.cap && (!with_normals || p.nrm_view)) return hipSuccess;
  void* keep_ws[PairSet::WS_COUNT];
  for (int k = 0; k < PairSet::WS_COUNT; ++k) { keep_ws[k] = p.ws[k]; p.ws[k] = nullptr; }   // (free_pairs would drop the workspace too)
  const size_t wc = p.ws_cand, wt = p.ws_tmp_bytes, wnd = p.ws_nd, wns = p.ws_ns;
  free_pairs(p);
  for (int k = 0; k < PairSet::WS_COUNT; ++k) p.ws[k] = keep_ws[k];
  p.ws_cand = wc; p.ws_tmp_bytes = wt; p.ws_nd = wnd; p.ws_ns = wns;
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

// The workspace of a pair search lives with the pair set: a search allocates nothing once it has run at its size
// (a dozen hipMalloc / hipFree pairs per search -- each free a device synchronisation -- were a third of its time).
static hipError_t ensure_ws(PairSet& p, size_t ncand, size_t nd, size_t ns, size_t tmp_bytes) {
  enum { REV_POS, REV_D2, KEYS_IN, KEYS_OUT, SLOTS_IN, SLOTS_OUT, C_POSD, C_POSS, C_D2, FLAGS, OFFS, WINNER, SEL_KEYS, SEL_STATE, TMP };
  if (ncand > p.ws_cand) {
    const int cand_slots[] = {KEYS_IN, KEYS_OUT, SLOTS_IN, SLOTS_OUT, C_POSD, C_POSS, C_D2, FLAGS, OFFS, SEL_KEYS};
    for (int k : cand_slots) { if (p.ws[k]) (void)hipFree(p.ws[k]); p.ws[k] = nullptr; }
    const size_t c = ncand;
    HIP_TRY(hipMalloc(&p.ws[KEYS_IN], c * 8)); HIP_TRY(hipMalloc(&p.ws[KEYS_OUT], c * 8)); HIP_TRY(hipMalloc(&p.ws[SEL_KEYS], c * 8));
    HIP_TRY(hipMalloc(&p.ws[SLOTS_IN], c * 4)); HIP_TRY(hipMalloc(&p.ws[SLOTS_OUT], c * 4));
    HIP_TRY(hipMalloc(&p.ws[C_POSD], c * 4)); HIP_TRY(hipMalloc(&p.ws[C_POSS], c * 4)); HIP_TRY(hipMalloc(&p.ws[C_D2], c * 4));
    HIP_TRY(hipMalloc(&p.ws[FLAGS], c * 4)); HIP_TRY(hipMalloc(&p.ws[OFFS], c * 4));
    if (!p.ws[SEL_STATE]) HIP_TRY(hipMalloc(&p.ws[SEL_STATE], filter_state_bytes()));
    p.ws_cand = ncand;
  }
  // (sized by the clouds, not by the candidates: a FIRST_TO_SECOND search -- nd candidates -- after a BOTH search of a smaller target
  //  -- nd' + ns' >= nd candidates -- keeps the candidate arrays and still needs longer per-target arrays)
  if (nd > p.ws_nd || !p.ws[REV_POS]) {
    for (int k : {REV_POS, REV_D2}) { if (p.ws[k]) (void)hipFree(p.ws[k]); p.ws[k] = nullptr; }
    HIP_TRY(hipMalloc(&p.ws[REV_POS], (nd ? nd : 1) * 4)); HIP_TRY(hipMalloc(&p.ws[REV_D2], (nd ? nd : 1) * 4));
    p.ws_nd = nd;
  }
  if (ns > p.ws_ns || !p.ws[WINNER]) {
    if (p.ws[WINNER]) (void)hipFree(p.ws[WINNER]);
    p.ws[WINNER] = nullptr;
    HIP_TRY(hipMalloc(&p.ws[WINNER], (ns ? ns : 1) * 8));
    p.ws_ns = ns;
  }
  if (tmp_bytes > p.ws_tmp_bytes) {
    if (p.ws[TMP]) (void)hipFree(p.ws[TMP]);
    p.ws[TMP] = nullptr;
    HIP_TRY(hipMalloc(&p.ws[TMP], tmp_bytes));
    p.ws_tmp_bytes = tmp_bytes;
  }
  return hipSuccess;
}

// exclusive scan of flags into offs; the total comes back to the host (one synchronisation: the next launches are sized by it)
static hipError_t scan_flags_ws(PairSet& p, const uint32_t* flags, uint32_t* offs, uint32_t n, uint32_t* total_out, hipStream_t s) {
  *total_out = 0;
  if (n == 0) return hipSuccess;
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, flags, offs, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  HIP_TRY(ensure_ws(p, p.ws_cand, 0, 0, tmp_bytes ? tmp_bytes : 16));
  HIP_TRY(rocprim::exclusive_scan(p.ws[14], tmp_bytes, flags, offs, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  uint32_t last_off = 0, last_flag = 0;
  HIP_TRY(hipMemcpyAsync(&last_off, offs + (n - 1), 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&last_flag, flags + (n - 1), 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *total_out = last_off + last_flag;
  return hipSuccess;
}

static unsigned bits_for_u32(uint32_t n) {
  unsigned b = 1;
  while (b < 32 && (1ull << b) < (unsigned long long)n) ++b;
  return b;
}

// direction: 1 = FIRST_TO_SECOND, 2 = BOTH.  For BOTH the caller has already run the forward search with the state's
// transform (fwd_pos / fwd_d2 by sorted-source position).  sgrid: the grid over the source in SOURCE coordinates (records
// carry original source indices); T_host: the state's transform.  A transform whose linear part is (nearly) singular
// cannot be searched through its inverse: then -- and only then -- a grid over the transformed source is built for this
// one search, as the reference builds its kd-tree.
hipError_t find_pairs(const FeatSpec& feat, const GridDev& g, const GridDev& sgrid, const float* d_src_xyz, const float* d_src_nrm, const float4* src_sorted, uint32_t ns,
                      const IcpState* state, const IcpState* id_state, const float T_host[16], float max_sq, int direction, bool reciprocal,
                      double inlier_fraction, bool one_to_one, const uint32_t* fwd_pos, const float* fwd_d2, PairSet& out, hipStream_t s, const TieDev* rev_tie) {
  enum { REV_POS, REV_D2, KEYS_IN, KEYS_OUT, SLOTS_IN, SLOTS_OUT, C_POSD, C_POSS, C_D2, FLAGS, OFFS, WINNER, SEL_KEYS, SEL_STATE, TMP };
  const uint32_t nd = g.n;
  out.count = 0;
  const size_t ncand = (size_t)nd + (direction == 2 ? ns : 0);
  HIP_TRY(ensure_pairs(out, ncand, d_src_nrm != nullptr));
  if (nd == 0 || ns == 0) return hipSuccess;   // kd_tree_utilities.hpp:16-19: an empty side gives no correspondences
  const unsigned end_bit = 32u + bits_for_u32(nd);            // keys = first << 32 | second, first < nd; the invalid key (all ones) still sorts last
  size_t sort_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ncand, 0u, end_bit, s));
  HIP_TRY(ensure_ws(out, ncand, nd, ns, sort_bytes ? sort_bytes : 16));
  uint32_t* rev_pos = (uint32_t*)out.ws[REV_POS]; float* rev_d2 = (float*)out.ws[REV_D2];
  unsigned long long *keys_in = (unsigned long long*)out.ws[KEYS_IN], *keys_out = (unsigned long long*)out.ws[KEYS_OUT];
  uint32_t *slots_in = (uint32_t*)out.ws[SLOTS_IN], *slots_out = (uint32_t*)out.ws[SLOTS_OUT], *c_posd = (uint32_t*)out.ws[C_POSD], *c_poss = (uint32_t*)out.ws[C_POSS];
  float* c_d2 = (float*)out.ws[C_D2];
  uint32_t *flags = (uint32_t*)out.ws[FLAGS], *offs = (uint32_t*)out.ws[OFFS];

  // 1. reverse search: the target points (in their grid order) against the source
  InvArgs iv{};
  bool through_inverse = false;
  {
    const double a00 = T_host[0], a01 = T_host[4], a02 = T_host[8], a10 = T_host[1], a11 = T_host[5], a12 = T_host[9], a20 = T_host[2], a21 = T_host[6], a22 = T_host[10];
    const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    const double M[9] = {a00 * a00 + a10 * a10 + a20 * a20, a00 * a01 + a10 * a11 + a20 * a21, a00 * a02 + a10 * a12 + a20 * a22,
                         a00 * a01 + a10 * a11 + a20 * a21, a01 * a01 + a11 * a11 + a21 * a21, a01 * a02 + a11 * a12 + a21 * a22,
                         a00 * a02 + a10 * a12 + a20 * a22, a01 * a02 + a11 * a12 + a21 * a22, a02 * a02 + a12 * a12 + a22 * a22};   // L^T L
    double w[3], V[9];
    sym_eig3(M, w, V);
    const double wmin = std::min(w[0], std::min(w[1], w[2])), wmax = std::max(w[0], std::max(w[1], w[2]));
    if (std::isfinite(det) && wmin > 0.0 && wmin > 1e-6 * wmax) {
      const double id = 1.0 / det, smin = std::sqrt(wmin), smax = std::sqrt(wmax);
      const double i00 = (a11 * a22 - a12 * a21) * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
      const double i10 = (a12 * a20 - a10 * a22) * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
      const double i20 = (a10 * a21 - a11 * a20) * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
      const double t0 = T_host[12], t1 = T_host[13], t2 = T_host[14];
      const double inv[16] = {i00, i10, i20, 0, i01, i11, i21, 0, i02, i12, i22, 0,
                              -(i00 * t0 + i01 * t1 + i02 * t2), -(i10 * t0 + i11 * t1 + i12 * t2), -(i20 * t0 + i21 * t1 + i22 * t2), 1};
      for (int i = 0; i < 16; ++i) iv.Ti[i] = (float)inv[i];
      iv.smin = (float)(smin * (1.0 - 1e-5));
      // |T p' - p| and the rounding of T s: a few f32 ulps of the coordinates involved, through the larger of T and T^-1
      const double ext_t = std::max({std::fabs((double)g.ox), std::fabs((double)g.oy), std::fabs((double)g.oz)}) + (double)std::max(g.nx, std::max(g.ny, g.nz)) * g.cell;
      const double ext_s = std::max({std::fabs((double)sgrid.ox), std::fabs((double)sgrid.oy), std::fabs((double)sgrid.oz)}) + (double)std::max(sgrid.nx, std::max(sgrid.ny, sgrid.nz)) * sgrid.cell;
      const double scale = std::max(ext_t, smax * ext_s) + std::fabs(t0) + std::fabs(t1) + std::fabs(t2);
      iv.eps = (float)(4e-6 * scale * std::max(1.0, smax / smin));
      through_inverse = sgrid.pts != nullptr;
    }
  }
  float* d_q = nullptr;
  GridBuildResult qg{};
  bool have_grid = false;
  hipError_t e = hipSuccess;
  do {
    const float4* cand_pts = sgrid.pts;
    if (through_inverse) {
      if (feat.enabled) hipLaunchKernelGGL(k_reverse_search<true>, dim3(iter_num_blocks(nd)), dim3(256), 0, s, sgrid, g.pts, nd, state, iv, max_sq, rev_pos, rev_d2, feat, TieDev{});
      else hipLaunchKernelGGL(k_reverse_search<false>, dim3(iter_num_blocks(nd)), dim3(256), 0, s, sgrid, g.pts, nd, state, iv, max_sq, rev_pos, rev_d2, feat, rev_tie ? *rev_tie : TieDev{});
    } else {
      if (feat.enabled) { e = hipErrorNotSupported; break; }      // (a feature search under a (nearly) singular transform: not implemented)
      if ((e = hipMalloc(&d_q, 3 * (size_t)ns * sizeof(float))) != hipSuccess) break;
      hipLaunchKernelGGL(k_transform_original, dim3(nblk(ns)), dim3(256), 0, s, d_src_xyz, ns, state, d_q);
      double mean[3];
      if ((e = build_grid(d_q, nullptr, ns, s, &qg, mean, 1.0)) != hipSuccess) break;
      have_grid = true;
      IterArgs r{};
      r.grid = qg.grid; r.src = g.pts; r.src_nrm = nullptr; r.ns = nd; r.max_sq = max_sq; r.state = id_state;
      r.nn_pos = rev_pos; r.nn_d2 = rev_d2; r.store_matches = 1;
      launch_iter(r, IM_NONE, true, true, iter_num_blocks(nd), s);
      cand_pts = qg.grid.pts;       // (records carry original source indices here too: q was formed in the original order)
    }
    // 2. candidates -> keys (original indices) -> sort
    hipLaunchKernelGGL(k_cand_reverse, dim3(nblk(nd)), dim3(256), 0, s, g.pts, nd, rev_pos, rev_d2, cand_pts, keys_in, slots_in, c_posd, c_poss, c_d2);
    if (direction == 2)
      hipLaunchKernelGGL(k_cand_forward, dim3(nblk(ns)), dim3(256), 0, s, g.pts, nd, src_sorted, ns, fwd_pos, fwd_d2, keys_in, slots_in, c_posd, c_poss,
                         c_d2);
    if ((e = rocprim::radix_sort_pairs(out.ws[TMP], sort_bytes, keys_in, keys_out, slots_in, slots_out, ncand, 0u, end_bit, s)) != hipSuccess) break;
    // 3. union / intersection / plain, then ordered compaction
    hipLaunchKernelGGL(k_mark, dim3(nblk(ncand)), dim3(256), 0, s, keys_out, (uint32_t)ncand, direction == 2 ? (reciprocal ? 2 : 1) : 0, flags);
    uint32_t m = 0;
    if ((e = scan_flags_ws(out, flags, offs, (uint32_t)ncand, &m, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_compact_sorted, dim3(nblk(ncand)), dim3(256), 0, s, keys_out, slots_out, flags, offs, (uint32_t)ncand, c_posd, c_poss, c_d2,
                       out.first, out.second, out.posd, out.poss, out.d2);
    // 4. post-filters on the pair list (correspondence_search_kd_tree.hpp:224-225)
    if (m > 0 && inlier_fraction > 0.0 && inlier_fraction < 1.0) {
      launch_select_fraction(out.d2, m, inlier_fraction, (unsigned long long*)out.ws[SEL_KEYS], out.ws[SEL_STATE], flags, s);
      uint32_t m2 = 0;
      if ((e = scan_flags_ws(out, flags, offs, m, &m2, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_compact_pairs, dim3(nblk(m)), dim3(256), 0, s, flags, offs, m, out.first, out.second, out.posd, out.poss, out.d2, out.first2,
                         out.second2, out.posd2, out.poss2, out.d2b);
      std::swap(out.first, out.first2); std::swap(out.second, out.second2); std::swap(out.posd, out.posd2); std::swap(out.poss, out.poss2);
      std::swap(out.d2, out.d2b);
      m = m2;
    }
    if (m > 0 && one_to_one && direction == 1) {
      unsigned long long* winner = (unsigned long long*)out.ws[WINNER];
      if ((e = hipMemsetAsync(winner, 0xFF, (size_t)ns * 8, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_o2o_min_pairs, dim3(nblk(m)), dim3(256), 0, s, out.first, out.poss, out.d2, m, winner);
      hipLaunchKernelGGL(k_o2o_flags_pairs, dim3(nblk(m)), dim3(256), 0, s, out.first, out.poss, out.d2, m, winner, flags);
      uint32_t m2 = 0;
      if ((e = scan_flags_ws(out, flags, offs, m, &m2, s)) != hipSuccess) break;
      hipLaunchKernelGGL(k_compact_pairs, dim3(nblk(m)), dim3(256), 0, s, flags, offs, m, out.first, out.second, out.posd, out.poss, out.d2, out.first2,
                         out.second2, out.posd2, out.poss2, out.d2b);
      std::swap(out.first, out.first2); std::swap(out.second, out.second2); std::swap(out.posd, out.posd2); std::swap(out.poss, out.poss2);
      std::swap(out.d2, out.d2b);
      m = m2;
    }
    // 5. the view the accumulation kernels stream over: one "query" per pair
    if (m > 0)
      hipLaunchKernelGGL(k_gather_pair_view, dim3(nblk(m)), dim3(256), 0, s, d_src_xyz, d_src_nrm, out.poss, m, out.src_view,
                         d_src_nrm ? out.nrm_view : (float4*)nullptr);
    if (have_grid) e = hipStreamSynchronize(s);      // (the one-off grid is freed below)
    out.count = m;
  } while (0);
  if (have_grid) free_grid(qg.grid);
  if (d_q) (void)hipFree(d_q);
  return e;
}

__global__ void k_grid_to_sorted(const float4* __restrict__ sgrid_pts, uint32_t ns, const uint32_t* __restrict__ src_inv, uint32_t* __restrict__ out) {
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ns; g += gridDim.x * blockDim.x) out[g] = src_inv[__float_as_uint(sgrid_pts[g].w)];
}
void launch_grid_to_sorted(const float4* sgrid_pts, uint32_t ns, const uint32_t* src_inv, uint32_t* out, hipStream_t s) {
  if (ns) hipLaunchKernelGGL(k_grid_to_sorted, dim3(std::min<uint32_t>((ns + 255u) / 256u, 4096u)), dim3(256), 0, s, sgrid_pts, ns, src_inv, out);
}
int reverse_warm_blocks(uint32_t nd) {      // at least eight rounds of 256 target points per block
  long nb = ((long)nd + 8 * 256 - 1) / (8 * 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}
void launch_reverse_search_rigid(const GridDev& g, const GridDev& sgrid, const IcpState* state, float max_sq, uint32_t* rev_pos, float* rev_d2, hipStream_t s,
                                 const FeatSpec* feat, const TieDev* rev_tie, const float* warm_src_safe2, const RevFused* fused) {
  if (g.n == 0) return;
  InvArgs iv{};
  iv.rigid_on_device = 1;
  iv.smin = 1.0f - 1e-4f;        // (the loop's transforms are polar-projected rotations: singular values within ~1e-6 of 1)
  const double ext_t = std::max({std::fabs((double)g.ox), std::fabs((double)g.oy), std::fabs((double)g.oz)}) + (double)std::max(g.nx, std::max(g.ny, g.nz)) * g.cell;
  const double ext_s = std::max({std::fabs((double)sgrid.ox), std::fabs((double)sgrid.oy), std::fabs((double)sgrid.oz)}) + (double)std::max(sgrid.nx, std::max(sgrid.ny, sgrid.nz)) * sgrid.cell;
  iv.eps = (float)(8e-6 * (ext_t + ext_s) + 1e-4 * ext_s);   // rounding of p' and of T s, and |T p' - p| for a linear part up to 1e-4 off orthonormal
  FeatSpec none{};
  if (warm_src_safe2 != nullptr && !(feat && feat->enabled)) {      // rev_pos holds the previous iteration's reverse matches
    const dim3 gb(reverse_warm_blocks(g.n)), tb(256);
    const TieDev tt = rev_tie ? *rev_tie : TieDev{};
    RevAcc ra{};
    int acc = IM_NONE;
    if (fused) {
      acc = fused->metric; ra.mode = fused->mode; ra.fwd_pos = fused->fwd_pos; ra.src_inv = fused->src_inv; ra.grid_to_sorted = fused->grid_to_sorted; ra.partials = fused->partials; ra.dst_nrm = g.nrm;
      for (int k = 0; k < 3; ++k) ra.dst_mean[k] = fused->dst_mean[k];
    }
    switch (acc) {
      case IM_KABSCH: hipLaunchKernelGGL((k_reverse_warm<IM_KABSCH>), gb, tb, 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, warm_src_safe2, tt, ra); break;
      case IM_PLANE: hipLaunchKernelGGL((k_reverse_warm<IM_PLANE>), gb, tb, 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, warm_src_safe2, tt, ra); break;
      case IM_POINT: hipLaunchKernelGGL((k_reverse_warm<IM_POINT>), gb, tb, 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, warm_src_safe2, tt, ra); break;
      case IM_BOTH: hipLaunchKernelGGL((k_reverse_warm<IM_BOTH>), gb, tb, 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, warm_src_safe2, tt, ra); break;
      default: hipLaunchKernelGGL((k_reverse_warm<IM_NONE>), gb, tb, 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, warm_src_safe2, tt, ra); break;
    }
    return;
  }
  if (feat && feat->enabled) hipLaunchKernelGGL(k_reverse_search<true>, dim3(iter_num_blocks(g.n)), dim3(256), 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, *feat, TieDev{});
  else hipLaunchKernelGGL(k_reverse_search<false>, dim3(iter_num_blocks(g.n)), dim3(256), 0, s, sgrid, g.pts, g.n, state, iv, max_sq, rev_pos, rev_d2, none, rev_tie ? *rev_tie : TieDev{});
}

}  // namespace cilhip
