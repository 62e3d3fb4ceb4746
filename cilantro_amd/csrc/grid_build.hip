// grid_build.hip -- one-off construction of the target index (replaces the single-threaded nanoflann
// kd-tree build: core/kd_tree.hpp:162-170 -> 3rd_party/nanoflann/nanoflann.hpp:1661-1687,1150-1212,
// 5.7 s for 10M points on the reference's CPU path) and the spatial pre-sort of the source cloud.
//
//   bbox + f64 mean (one pass)  ->  cell size from density (adaptive: shrink until the expected
//   own-cell population is small)  ->  cell key per point  ->  stable LSD radix sort of (key, index)
//   (rocPRIM; stable => within a cell points stay in ascending original index)  ->  cell_start table
//   ->  gather into 16-byte {x,y,z,idx} records (+ normals in the same order).
//
// The search radius is NOT baked into the grid (the kd-tree it replaces is radius-agnostic too):
// the query kernel expands shells until its pruning bound proves exactness.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace cilhip {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

constexpr int RB = 1024;  // reduction blocks

// per-block min/max (f32) and sum (f64) of xyz
__global__ __launch_bounds__(256) void k_bbox_sum(const float* __restrict__ xyz, uint32_t n, float* bmin, float* bmax, double* bsum) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  double sm[3] = {0, 0, 0};
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t beg = blockIdx.x * per, end = min(beg + per, n);
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = xyz[3 * (size_t)i + c];
      mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); sm[c] += (double)v;
    }
  }
  __shared__ float smn[4][3], smx[4][3];
  __shared__ double ssm[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = mn[c], b = mx[c]; double s = sm[c];
    for (int off = 32; off > 0; off >>= 1) {
      a = fminf(a, __shfl_down(a, off, 64)); b = fmaxf(b, __shfl_down(b, off, 64)); s += __shfl_down(s, off, 64);
    }
    if (lane == 0) { smn[wave][c] = a; smx[wave][c] = b; ssm[wave][c] = s; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    bmin[blockIdx.x * 3 + c] = fminf(fminf(smn[0][c], smn[1][c]), fminf(smn[2][c], smn[3][c]));
    bmax[blockIdx.x * 3 + c] = fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]));
    bsum[blockIdx.x * 3 + c] = (ssm[0][c] + ssm[1][c]) + (ssm[2][c] + ssm[3][c]);
  }
}

static hipError_t bbox_mean(const float* d_xyz, uint32_t n, hipStream_t s, float lo[3], float hi[3], double mean[3]) {
  float *d_min = nullptr, *d_max = nullptr; double* d_sum = nullptr;
  HIP_TRY(hipMalloc(&d_min, RB * 3 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_max, RB * 3 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_sum, RB * 3 * sizeof(double)));
  hipLaunchKernelGGL(k_bbox_sum, dim3(RB), dim3(256), 0, s, d_xyz, n, d_min, d_max, d_sum);
  std::vector<float> hmin(RB * 3), hmax(RB * 3); std::vector<double> hsum(RB * 3);
  HIP_TRY(hipMemcpyAsync(hmin.data(), d_min, RB * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hmax.data(), d_max, RB * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hsum.data(), d_sum, RB * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(d_min); (void)hipFree(d_max); (void)hipFree(d_sum);
  for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; mean[c] = 0.0; }
  for (int b = 0; b < RB; ++b)
    for (int c = 0; c < 3; ++c) {
      lo[c] = std::min(lo[c], hmin[b * 3 + c]); hi[c] = std::max(hi[c], hmax[b * 3 + c]); mean[c] += hsum[b * 3 + c];
    }
  for (int c = 0; c < 3; ++c) mean[c] = n ? mean[c] / (double)n : 0.0;
  return hipSuccess;
}

hipError_t mean3_device(const float* d_xyz, uint32_t n, hipStream_t s, double mean_out[3], float* lo_out, float* hi_out) {
  float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
  hipError_t e = hipSuccess;
  if (n == 0) mean_out[0] = mean_out[1] = mean_out[2] = 0.0;
  else e = bbox_mean(d_xyz, n, s, lo, hi, mean_out);
  for (int c = 0; c < 3; ++c) { if (lo_out) lo_out[c] = lo[c]; if (hi_out) hi_out[c] = hi[c]; }
  return e;
}

__device__ __forceinline__ uint32_t cell_of(const GridDev& g, float x, float y, float z) {
  int cx = (int)floorf(fminf(fmaxf((x - g.ox) * g.inv_cell, -1.0f), 1.0e9f));
  int cy = (int)floorf(fminf(fmaxf((y - g.oy) * g.inv_cell, -1.0f), 1.0e9f));
  int cz = (int)floorf(fminf(fmaxf((z - g.oz) * g.inv_cell, -1.0f), 1.0e9f));
  cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
  return ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
}

__global__ void k_cell_keys(const float* __restrict__ xyz, uint32_t n, GridDev g, uint32_t* keys, uint32_t* vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    keys[i] = cell_of(g, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
    vals[i] = i;
  }
}

struct Tf { float m[16]; };

__global__ void k_cell_keys_tf(const float* __restrict__ xyz, uint32_t n, GridDev g, Tf T, uint32_t* keys, uint32_t* vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float qx, qy, qz;
    transform_point(T.m, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], qx, qy, qz);
    keys[i] = cell_of(g, qx, qy, qz);
    vals[i] = i;
  }
}

// cell_start[k] = first sorted position whose key >= k, for k in [0, ncells]
__global__ void k_cell_start(const uint32_t* __restrict__ keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
    const uint32_t prev = (i == 0) ? 0u : keys_sorted[i - 1] + 1u;           // first key not yet closed
    const uint32_t cur = (i == n) ? ncells + 1u : keys_sorted[i] + 1u;      // exclusive end to fill
    for (uint32_t k = prev; k < cur; ++k) cell_start[k] = i;
  }
}

// The same table for MANY cells with long empty stretches (the target grid: two layers of empty cells around the data make
// the first and the last key's gap ~10^5 cells, which one lane of k_cell_start fills alone: 1.4 ms at 10M): mark the first
// sorted position of every non-empty cell, then a reverse running minimum hands every empty cell the start of the next
// non-empty one (0.1 ms).
__global__ void k_run_starts(const uint32_t* __restrict__ keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t k = keys_sorted[i];
    if (i == 0 || keys_sorted[i - 1] != k) cell_start[k] = i;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cell_start[ncells] = n;
}
static inline int grid_blocks(uint32_t n);
static hipError_t cell_start_table(const uint32_t* keys_sorted, uint32_t n, uint32_t ncells, uint32_t* cell_start, hipStream_t s) {
  HIP_TRY(hipMemsetAsync(cell_start, 0xFF, ((size_t)ncells + 1) * sizeof(uint32_t), s));
  hipLaunchKernelGGL(k_run_starts, dim3(grid_blocks(n)), dim3(256), 0, s, keys_sorted, n, ncells, cell_start);
  auto rit = rocprim::make_reverse_iterator(cell_start + (size_t)ncells + 1);
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, rit, rit, (size_t)ncells + 1, rocprim::minimum<uint32_t>(), s));
  void* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  hipError_t e = rocprim::inclusive_scan(tmp, tmp_bytes, rit, rit, (size_t)ncells + 1, rocprim::minimum<uint32_t>(), s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

__global__ void k_gather(const float* __restrict__ xyz, const float* __restrict__ nrm, const uint32_t* __restrict__ perm,
                         uint32_t n, float4* pts_out, float4* nrm_out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t j = perm[i];
    pts_out[i] = make_float4(xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2], __uint_as_float(j));
    if (nrm_out) nrm_out[i] = make_float4(nrm[3 * (size_t)j], nrm[3 * (size_t)j + 1], nrm[3 * (size_t)j + 2], 0.0f);
  }
}

// sum over cells of count^2 (f64): expected own-cell population seen by a random target point is this / n
__global__ void k_occupancy(const uint32_t* __restrict__ cell_start, uint32_t ncells, double* out) {
  double s = 0.0;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < ncells; k += gridDim.x * blockDim.x) {
    const double c = (double)(cell_start[k + 1] - cell_start[k]);
    s += c * c;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(out, s);
}

static inline int grid_blocks(uint32_t n) { return (int)std::min<uint32_t>((n + 255) / 256, 8192u) + (n == 0); }

static hipError_t sort_pairs(uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, uint32_t n, unsigned bits, hipStream_t s) {
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, bits, s));
  void* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, bits, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(tmp);
  return e != hipSuccess ? e : e2;
}

static unsigned bits_for(uint32_t ncells) {
  unsigned b = 1;
  while (b < 32 && (1ull << b) < (unsigned long long)ncells) ++b;
  return b;
}

static void set_dims(GridDev& g, const float lo[3], const float hi[3], double cell) {
  const int MAXD = 2048;
  const double MAXCELLS = 67108864.0;  // 2^26
  double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
  double maxext = std::max(ext[0], std::max(ext[1], ext[2]));
  if (!(maxext > 0.0) || !std::isfinite(maxext)) maxext = 1.0;
  if (!(cell > 0.0) || !std::isfinite(cell)) cell = maxext;
  cell = std::max(cell, maxext / (MAXD - 1));
  // GRID_PAD layers of (empty) cells around the data: every cell that can hold a target point (and the first layer
  // around them) has all 26 neighbours inside the grid, so the fast search path needs no boundary cases (queries in
  // the outermost layer or beyond take the generic path).
  for (;;) {
    double nxd = std::floor(ext[0] / cell) + 1 + 2 * GRID_PAD, nyd = std::floor(ext[1] / cell) + 1 + 2 * GRID_PAD, nzd = std::floor(ext[2] / cell) + 1 + 2 * GRID_PAD;
    if (nxd * nyd * nzd <= MAXCELLS && nxd <= MAXD && nyd <= MAXD && nzd <= MAXD) {
      g.nx = (int)nxd; g.ny = (int)nyd; g.nz = (int)nzd;
      break;
    }
    cell *= 1.1;
  }
  g.cell = (float)cell;
  g.ox = lo[0] - (float)GRID_PAD * g.cell; g.oy = lo[1] - (float)GRID_PAD * g.cell; g.oz = lo[2] - (float)GRID_PAD * g.cell;
  g.inv_cell = 1.0f / g.cell;
  g.margin = g.cell * (1.0f / 512.0f);
}

hipError_t build_grid(const float* d_xyz, const float* d_nrm, uint32_t n, hipStream_t s, GridBuildResult* out, double mean_out[3],
                      double target_occupancy, double refined_factor) {
  GridDev g{};
  g.n = n; g.pts = nullptr; g.nrm = nullptr; g.pn = nullptr; g.cell_start = nullptr;
  out->avg_occupancy = 0.0;
  if (n == 0) {
    // empty target: a minimal grid of empty cells; every search returns "none" (kd_tree_utilities.hpp:16-19)
    const float z3[3] = {0, 0, 0};
    set_dims(g, z3, z3, 1.0);
    const size_t nc = (size_t)g.nx * g.ny * g.nz;
    uint32_t* cs = nullptr;
    HIP_TRY(hipMalloc(&cs, (nc + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(cs, 0, (nc + 1) * sizeof(uint32_t), s));
    float4* dummy = nullptr;
    HIP_TRY(hipMalloc(&dummy, sizeof(float4)));
    g.cell_start = cs; g.pts = dummy;
    out->grid = g; out->n_cells = nc;
    mean_out[0] = mean_out[1] = mean_out[2] = 0.0;
    return hipSuccess;
  }
  float lo[3], hi[3];
  HIP_TRY(bbox_mean(d_xyz, n, s, lo, hi, mean_out));
  double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
  const double maxext = std::max(ext[0], std::max(ext[1], ext[2]));
  // volume with degenerate extents floored so planar / linear clouds still get a sane first guess
  double vol = 1.0;
  for (int c = 0; c < 3; ++c) vol *= std::max(ext[c], maxext * 1e-3);
  const double TARGET = target_occupancy > 0.0 ? target_occupancy : 1.0;  // points per cell for a volumetric cloud
  double cell = std::cbrt(vol * TARGET / (double)n);

  uint32_t *k_in = nullptr, *k_out = nullptr, *v_in = nullptr, *v_out = nullptr, *cs = nullptr;
  double* d_occ = nullptr;
  HIP_TRY(hipMalloc(&k_in, (size_t)n * 4)); HIP_TRY(hipMalloc(&k_out, (size_t)n * 4));
  HIP_TRY(hipMalloc(&v_in, (size_t)n * 4)); HIP_TRY(hipMalloc(&v_out, (size_t)n * 4));
  HIP_TRY(hipMalloc(&d_occ, sizeof(double)));
  double occ = 0.0;
  size_t ncells = 0;
  for (int attempt = 0; attempt < 8; ++attempt) {
    set_dims(g, lo, hi, cell);
    ncells = (size_t)g.nx * g.ny * g.nz;
    if (cs) { (void)hipFree(cs); cs = nullptr; }
    HIP_TRY(hipMalloc(&cs, (ncells + 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_cell_keys, dim3(grid_blocks(n)), dim3(256), 0, s, d_xyz, n, g, k_in, v_in);
    HIP_TRY(sort_pairs(k_in, k_out, v_in, v_out, n, bits_for((uint32_t)ncells), s));
    HIP_TRY(cell_start_table(k_out, n, (uint32_t)ncells, cs, s));
    HIP_TRY(hipMemsetAsync(d_occ, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_occupancy, dim3(grid_blocks((uint32_t)ncells)), dim3(256), 0, s, cs, (uint32_t)ncells, d_occ);
    HIP_TRY(hipMemcpyAsync(&occ, d_occ, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    occ /= (double)n;
    // adaptive refinement for surface-like / clustered clouds: too many candidates per cell -> shrink.  Such a cloud fills a small
    // part of its cells; a search that has to go beyond the first block of cells (a source far from alignment, residuals of several
    // cells: two frames of a depth sensor) walks shells of mostly EMPTY cells, and what that costs goes with the number of cells, not
    // of points: refined grids stop at a population RF times that of a volumetric cloud's grid (measured on the reference's frames:
    // frame_1 vs frame_2 0.67 -> 0.27 ms per iteration at 3x the cell edge's third power, the near-aligned pair 0.032 -> 0.038).
    const double RF = refined_factor >= 1.0 ? refined_factor : 1.0;
    if (occ <= 3.0 * TARGET * (attempt == 0 ? 1.0 : RF)) break;
    const double shrink = std::min(0.85, std::max(0.3, std::pow(2.0 * TARGET * RF / occ, 1.0 / 2.5)));
    const double new_cell = (double)g.cell * shrink;
    GridDev probe = g;
    set_dims(probe, lo, hi, new_cell);
    if (probe.cell >= g.cell * 0.97f) break;  // dims / cell-count caps reached: keep the current grid
    cell = new_cell;
  }
  float4 *pts = nullptr, *nrm = nullptr;
  HIP_TRY(hipMalloc(&pts, (size_t)n * sizeof(float4)));
  if (d_nrm) HIP_TRY(hipMalloc(&nrm, (size_t)n * sizeof(float4)));
  hipLaunchKernelGGL(k_gather, dim3(grid_blocks(n)), dim3(256), 0, s, d_xyz, d_nrm, v_out, n, pts, nrm);
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(v_out); (void)hipFree(d_occ);
  g.pts = pts; g.nrm = nrm; g.cell_start = cs;
  out->grid = g; out->avg_occupancy = occ; out->n_cells = ncells;
  return hipSuccess;
}

void free_grid(GridDev& g) {
  if (g.pts) (void)hipFree((void*)g.pts);
  if (g.nrm) (void)hipFree((void*)g.nrm);
  if (g.pn) (void)hipFree((void*)g.pn);
  if (g.cell_start) (void)hipFree((void*)g.cell_start);
  g.pts = nullptr; g.nrm = nullptr; g.pn = nullptr; g.cell_start = nullptr;
}

// ---- source ordering + tiles ------------------------------------------------------------------------
// The source is sorted CUBE-major: cube = CUBE_EDGE^3 block of target-grid cells, key = cube_id * CUBE_EDGE^3 +
// cell within the cube.  Queries of one cube become one (or, above TILE_QUERIES queries, several) TILE(s) of
// the LDS-tiled search kernel: the octant blocks of a cube's queries fit a (CUBE_EDGE+2..3)^3-cell region that
// one workgroup stages in LDS once.
__device__ __forceinline__ uint32_t cube_key_of(const GridDev& g, float x, float y, float z) {
  int cx = (int)floorf(fminf(fmaxf((x - g.ox) * g.inv_cell, -1.0f), 1.0e9f));
  int cy = (int)floorf(fminf(fmaxf((y - g.oy) * g.inv_cell, -1.0f), 1.0e9f));
  int cz = (int)floorf(fminf(fmaxf((z - g.oz) * g.inv_cell, -1.0f), 1.0e9f));
  // cubes tile the DATA cells GRID_PAD..n-1-GRID_PAD (the outer layers hold no target points): shift by the padding,
  // so a cloud registered onto itself fills whole cubes; queries in the low outer layers join cube 0
  cx = min(max(cx - GRID_PAD, 0), g.nx - 1 - GRID_PAD); cy = min(max(cy - GRID_PAD, 0), g.ny - 1 - GRID_PAD); cz = min(max(cz - GRID_PAD, 0), g.nz - 1 - GRID_PAD);
  constexpr uint32_t C = CUBE_EDGE;
  const uint32_t cnx = (uint32_t)(g.nx - 1 - GRID_PAD) / C + 1, cny = (uint32_t)(g.ny - 1 - GRID_PAD) / C + 1;
  const uint32_t bx = (uint32_t)cx / C, by = (uint32_t)cy / C, bz = (uint32_t)cz / C;
  const uint32_t cube = (bz * cny + by) * cnx + bx;
  const uint32_t local = (((uint32_t)cz - bz * C) * C + ((uint32_t)cy - by * C)) * C + ((uint32_t)cx - bx * C);
  return cube * (C * C * C) + local;
}

__global__ void k_cube_keys_tf(const float* __restrict__ xyz, uint32_t n, GridDev g, Tf T, uint32_t* keys, uint32_t* vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float qx, qy, qz;
    transform_point(T.m, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], qx, qy, qz);
    keys[i] = cube_key_of(g, qx, qy, qz);
    vals[i] = i;
  }
}

__global__ void k_shift_keys(const uint32_t* __restrict__ in, uint32_t n, uint32_t* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] / (uint32_t)(CUBE_EDGE * CUBE_EDGE * CUBE_EDGE);
}

// tiles of one cube: ceil(count / TILE_QUERIES)
__global__ void k_tile_counts(const uint32_t* __restrict__ cube_start, uint32_t ncubes, uint32_t* counts) {
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncubes; c += gridDim.x * blockDim.x)
    counts[c] = (cube_start[c + 1] - cube_start[c] + (TILE_QUERIES - 1)) / TILE_QUERIES;
}

// tile_center[t] = centre of the tile's cube mapped back to SOURCE space (Tinv = inverse of the sort transform):
// with the common half-axes (tile_axes) that is the oriented box k_search_tiled pushes through the current transform
__global__ void k_emit_tiles(const uint32_t* __restrict__ cube_start, const uint32_t* __restrict__ tile_off, uint32_t ncubes, GridDev g, Tf Tinv,
                             uint2* tiles, float4* tile_center) {
  constexpr uint32_t C = CUBE_EDGE;
  const uint32_t cnx = (uint32_t)(g.nx - 1 - GRID_PAD) / C + 1, cny = (uint32_t)(g.ny - 1 - GRID_PAD) / C + 1;
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncubes; c += gridDim.x * blockDim.x) {
    const uint32_t b = cube_start[c], e = cube_start[c + 1];
    if (b == e) continue;
    const uint32_t bx = c % cnx, by = (c / cnx) % cny, bz = c / (cnx * cny);
    // cube b spans the cells [GRID_PAD + b C, GRID_PAD + (b + 1) C)
    const float hx = g.ox + ((float)(bx * C) + (float)GRID_PAD + 0.5f * (float)C) * g.cell, hy = g.oy + ((float)(by * C) + (float)GRID_PAD + 0.5f * (float)C) * g.cell,
                hz = g.oz + ((float)(bz * C) + (float)GRID_PAD + 0.5f * (float)C) * g.cell;
    const float* m = Tinv.m;
    const float4 ctr = make_float4(m[0] * hx + m[4] * hy + m[8] * hz + m[12], m[1] * hx + m[5] * hy + m[9] * hz + m[13],
                                   m[2] * hx + m[6] * hy + m[10] * hz + m[14], 0.0f);
    uint32_t t = tile_off[c];
    for (uint32_t q = b; q < e; q += TILE_QUERIES) { tiles[t] = make_uint2(q, min(q + TILE_QUERIES, e)); tile_center[t] = ctr; ++t; }
  }
}

// ws (optional): scratch and tile-table storage kept between calls -- a registration loop re-sorts the same source for every new
// initial transform, and the dozen hipMalloc / hipFree of a stand-alone call cost more than its kernels (1.7 -> 0.8 ms at 10M).
// With ws the tile table lives in ws (valid until the next call with it); without, the caller frees *d_tiles_out / *d_tile_center_out.
hipError_t sort_source(const float* d_xyz, uint32_t n, const GridDev& g, const float T[16], float4* d_out, hipStream_t s,
                       uint2** d_tiles_out, float4** d_tile_center_out, float tile_axes_out[9], uint32_t* ntiles_out, SortWorkspace* ws) {
  *d_tiles_out = nullptr; *d_tile_center_out = nullptr; *ntiles_out = 0;
  // inverse of the (affine) sort transform, in double; a singular linear part leaves a zero box (every query
  // then takes the clean-up pass: slow, still exact)
  Tf tinv;
  {
    const double a00 = T[0], a01 = T[4], a02 = T[8], a10 = T[1], a11 = T[5], a12 = T[9], a20 = T[2], a21 = T[6], a22 = T[10];
    const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    const double id = (det != 0.0 && std::isfinite(det)) ? 1.0 / det : 0.0;
    const double i00 = (a11 * a22 - a12 * a21) * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
    const double i10 = (a12 * a20 - a10 * a22) * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
    const double i20 = (a10 * a21 - a11 * a20) * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
    const double t0 = T[12], t1 = T[13], t2 = T[14];
    const double inv[16] = {i00, i10, i20, 0, i01, i11, i21, 0, i02, i12, i22, 0,
                            -(i00 * t0 + i01 * t1 + i02 * t2), -(i10 * t0 + i11 * t1 + i12 * t2), -(i20 * t0 + i21 * t1 + i22 * t2), 1};
    for (int i = 0; i < 16; ++i) tinv.m[i] = (float)inv[i];
    const double h = 0.5 * (double)CUBE_EDGE * (double)g.cell;
    const double li[9] = {i00, i01, i02, i10, i11, i12, i20, i21, i22};
    for (int i = 0; i < 9; ++i) tile_axes_out[i] = (float)(h * li[i]);
  }
  if (n == 0) return hipSuccess;
  constexpr uint32_t CE = CUBE_EDGE;
  const uint32_t cnx = (uint32_t)(g.nx - 1 - GRID_PAD) / CE + 1, cny = (uint32_t)(g.ny - 1 - GRID_PAD) / CE + 1, cnz = (uint32_t)(g.nz - 1 - GRID_PAD) / CE + 1;
  const uint32_t ncubes = cnx * cny * cnz;
  const unsigned bits = std::min(32u, bits_for(ncubes * CE * CE * CE));
  // one slab of scratch: 4 key / value arrays, 3 per-cube arrays, the temporaries of the sort and of the scan
  uint32_t* nul = nullptr;
  size_t sort_tmp = 0, scan_tmp = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_tmp, nul, nul, nul, nul, (size_t)n, 0u, bits, s));
  HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, nul, nul, 0u, (size_t)ncubes + 1, rocprim::plus<uint32_t>(), s));
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t arr = up((size_t)n * 4), carr = up(((size_t)ncubes + 1) * 4);
  const size_t need = 4 * arr + 3 * carr + up(sort_tmp ? sort_tmp : 16) + up(scan_tmp ? scan_tmp : 16);
  SortWorkspace local;
  SortWorkspace* w = ws ? ws : &local;
  hipError_t e = hipSuccess;
  if (w->scratch_bytes < need) {
    if (w->scratch) (void)hipFree(w->scratch);
    w->scratch = nullptr; w->scratch_bytes = 0;
    if ((e = hipMalloc(&w->scratch, need)) != hipSuccess) return e;
    w->scratch_bytes = need;
  }
  char* base = static_cast<char*>(w->scratch);
  uint32_t *k_in = reinterpret_cast<uint32_t*>(base), *k_out = reinterpret_cast<uint32_t*>(base + arr), *v_in = reinterpret_cast<uint32_t*>(base + 2 * arr),
           *v_out = reinterpret_cast<uint32_t*>(base + 3 * arr);
  uint32_t *cube_start = reinterpret_cast<uint32_t*>(base + 4 * arr), *tcount = reinterpret_cast<uint32_t*>(base + 4 * arr + carr),
           *toff = reinterpret_cast<uint32_t*>(base + 4 * arr + 2 * carr);
  void* tmp_sort = base + 4 * arr + 3 * carr;
  void* tmp_scan = base + 4 * arr + 3 * carr + up(sort_tmp ? sort_tmp : 16);
  Tf tf;
  for (int i = 0; i < 16; ++i) tf.m[i] = T[i];
  uint2* tiles = nullptr;
  float4* centers = nullptr;
  do {
    hipLaunchKernelGGL(k_cube_keys_tf, dim3(grid_blocks(n)), dim3(256), 0, s, d_xyz, n, g, tf, k_in, v_in);
    if ((e = rocprim::radix_sort_pairs(tmp_sort, sort_tmp, k_in, k_out, v_in, v_out, (size_t)n, 0u, bits, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_gather, dim3(grid_blocks(n)), dim3(256), 0, s, d_xyz, (const float*)nullptr, v_out, n, d_out, (float4*)nullptr);
    // cube_start[] over the sorted cube ids, then the tile table
    hipLaunchKernelGGL(k_shift_keys, dim3(grid_blocks(n)), dim3(256), 0, s, k_out, n, k_in);
    hipLaunchKernelGGL(k_cell_start, dim3(grid_blocks(n + 1)), dim3(256), 0, s, k_in, n, ncubes, cube_start);
    if ((e = hipMemsetAsync(tcount, 0, ((size_t)ncubes + 1) * 4, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_tile_counts, dim3(grid_blocks(ncubes)), dim3(256), 0, s, cube_start, ncubes, tcount);
    if ((e = rocprim::exclusive_scan(tmp_scan, scan_tmp, tcount, toff, 0u, (size_t)ncubes + 1, rocprim::plus<uint32_t>(), s)) != hipSuccess) break;
    uint32_t ntiles = 0;
    if ((e = hipMemcpyAsync(&ntiles, toff + ncubes, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) break;
    if (ws) {
      if (ws->tile_cap < ntiles + 1) {
        if (ws->tiles) (void)hipFree(ws->tiles);
        if (ws->centers) (void)hipFree(ws->centers);
        ws->tiles = nullptr; ws->centers = nullptr; ws->tile_cap = 0;
        const uint32_t cap = ntiles + 1 + ntiles / 8;
        if ((e = hipMalloc(&ws->tiles, (size_t)cap * sizeof(uint2))) != hipSuccess) break;
        if ((e = hipMalloc(&ws->centers, (size_t)cap * sizeof(float4))) != hipSuccess) break;
        ws->tile_cap = cap;
      }
      tiles = ws->tiles; centers = ws->centers;
    } else {
      if ((e = hipMalloc(&tiles, ((size_t)ntiles + 1) * sizeof(uint2))) != hipSuccess) break;
      if ((e = hipMalloc(&centers, ((size_t)ntiles + 1) * sizeof(float4))) != hipSuccess) break;
    }
    hipLaunchKernelGGL(k_emit_tiles, dim3(grid_blocks(ncubes)), dim3(256), 0, s, cube_start, toff, ncubes, g, tinv, tiles, centers);
    e = ws ? hipGetLastError() : hipStreamSynchronize(s);      // (with ws everything stays on the stream)
    *d_tiles_out = tiles; *d_tile_center_out = centers; *ntiles_out = ntiles;
  } while (0);
  if (!ws) {
    (void)hipStreamSynchronize(s);
    if (local.scratch) (void)hipFree(local.scratch);
    if (e != hipSuccess) {
      if (tiles) (void)hipFree(tiles);
      if (centers) (void)hipFree(centers);
    }
  }
  if (e != hipSuccess) { *d_tiles_out = nullptr; *d_tile_center_out = nullptr; *ntiles_out = 0; }
  return e;
}

void free_sort_workspace(SortWorkspace& ws) {
  if (ws.scratch) (void)hipFree(ws.scratch);
  if (ws.tiles) (void)hipFree(ws.tiles);
  if (ws.centers) (void)hipFree(ws.centers);
  ws = SortWorkspace();
}

}  // namespace cilhip
