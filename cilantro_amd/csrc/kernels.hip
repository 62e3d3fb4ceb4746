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
// Files: this one has the tile kernels, the per-lane / cooperative searches and the streaming accumulation; warm.hip the warm-started
// iteration; epilogue.hip the reduction + solve; extract.hip result extraction, keys, residuals; search_device.hpp what they share.
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
#include "search_device.hpp"

namespace cilhip {

void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

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
static_assert(FUSED_WAVE_BYTES * TILE_WAVES <= TILE_BYTES, "the accumulation scratch reuses the tile's point buffer");
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
void debug_dump_warm_clocks();      // (warm.hip)
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

}  // namespace cilhip
