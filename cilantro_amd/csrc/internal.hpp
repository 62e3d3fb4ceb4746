// internal.hpp -- shared declarations of libcilantro_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <thread>

#include "solve.hpp"

struct cilhip_ctx;      // (c_api.h: the opaque context)

namespace cilhip {

constexpr uint32_t NONE_U32 = 0xFFFFFFFFu;
#ifndef CILHIP_CUBE_EDGE
#define CILHIP_CUBE_EDGE 12
#endif
// LDS-tiled search: queries are grouped by cubes of CUBE_EDGE^3 target-grid cells
constexpr int CUBE_EDGE = CILHIP_CUBE_EDGE;
// Layers of empty cells around the data's bounding box.  Two: queries up to one cell outside the data (source points
// that noise / the current transform pushed just past the target's bounding box) still have all 26 neighbour cells
// inside the grid and stay on the fast search path.
constexpr int GRID_PAD = 2;
#ifndef CILHIP_TILE_THREADS
#define CILHIP_TILE_THREADS 1024
#endif
#ifndef CILHIP_TILE_WAVES_PER_SIMD
#define CILHIP_TILE_WAVES_PER_SIMD 8   /* resident waves per SIMD the tiled kernel is compiled for (register budget) */
#endif
#ifndef CILHIP_TILE_BYTES
#define CILHIP_TILE_BYTES 61568   /* LDS bytes of a tile's staged target points (pairs of records {x0, x1, y0, y1, z0, z1}: 12 B per point, + pad records); >= 3584 B per wave (accumulation scratch) */
#endif
#ifndef CILHIP_TILE_MAXE
#define CILHIP_TILE_MAXE 4352  /* entries of the staged cell table: region rows x (region width + 1) */
#endif
constexpr int TILE_THREADS = CILHIP_TILE_THREADS;  // workgroup size of the tiled search kernel
constexpr int TILE_QUERIES = 2 * TILE_THREADS;     // max queries per tile (two per lane)
constexpr int TILE_BYTES = CILHIP_TILE_BYTES;
constexpr int TILE_MAXSPAN = CUBE_EDGE + 6;                  // region rows per axis (y, z) the row tables hold
constexpr int TILE_MAXROWS = TILE_MAXSPAN * TILE_MAXSPAN;    // RY*RZ
constexpr int TILE_CAP = TILE_BYTES / 12 - 12;     // target points one tile can stage in LDS (behind them: 10 pad records, the 3x3x3 pass's queue)

// Uniform grid over the target cloud (the structure that replaces the nanoflann kd-tree).
// Target points are stored sorted by linear cell id (x fastest) as 16-byte records
// {x, y, z, bitcast(original index)} so one global_load_dwordx4 fetches a candidate and a run of
// cells along x is one contiguous range.
struct GridDev {
  float ox, oy, oz;            // origin = bbox min
  float cell, inv_cell;        // cell edge and its reciprocal
  float margin;                // geometric safety margin for the pruning bounds (cell * 2^-9)
  int nx, ny, nz;
  uint32_t n;                  // number of target points
  const float4* pts;           // [n]  sorted {x,y,z,orig_idx}
  const float4* nrm;           // [n]  sorted {nx,ny,nz,0} or nullptr
  const float4* pn;            // [2n] or nullptr: point and normal of position j side by side ({pts[j], nrm[j]}, 32 B): what the streaming
                               //      accumulation gathers through the stored matches -- one line per match instead of two (built on first use)
  const uint32_t* cell_start;  // [nx*ny*nz + 1]
};

// Device-resident ICP loop state: lets every iteration be enqueued without a host round trip.
struct IcpState {
  float T[16];        // transform_ (col-major)
  float smt[3];       // transform_ * src_mean_   (icp_single_transform_combined_metric.hpp:196)
  float delta;        // last_delta_norm_
  float innerL[9];    // Gauss-Newton inner tform (row-major), identity at step 0
  float innert[3];
  int iterations;     // iterations_
  int done;           // last_delta_norm_ < convergence_tol_ reached
  int inner_done;     // inner GN loop converged (transform_estimation.hpp:360)
  int pad0;
  unsigned long long ncorr;
  double dLd[9];      // f64 copies of the inner tform (the f32 ones feed the kernels)
  double dtd[3];
  double sums[SUMS_MAX];  // reduced sums of the last accumulation
  int slab_violation;     // spatially sharded runs: some source point may have left its slab's halo since the partition (sticky)
  unsigned int unproven;  // tiled search of the last iteration: queries its first stage (the octant block) did not prove
  float prev_delta;       // delta of the iteration before the last one (INFINITY before there is one)
  unsigned int listed;    // warm-started iteration: queries the nearest-other-point table did not settle (searched from their lists)
  // slab-sharded runs: the loop state right after the update that raised slab_violation (that update is still exact -- its
  // search ran inside the halos; the flag is about the NEXT search), so the caller can keep every iteration up to and including it
  int violation_iter;     // iterations performed when the flag went up (0: never)
  float violation_delta;
  unsigned long long violation_ncorr;
  float violation_T[16];
  float Tprev[16];        // transform_ BEFORE the last update = the transform the last executed iteration searched under: what the
                          // engine's correspondence set refers to after estimate() (correspondence_search_kd_tree.hpp:231)
  // Query motion since the run began (the warm-started iteration's margin test): motion_acc >= the PATH LENGTH every
  // query q = T s has travelled over the run's updates (sum over updates of max over the source's bounding box of |T' s - T s|;
  // rounded up), so the distance any query moved between two iterations is at most the difference of the two values;
  // motion_eps >= the rounding error of one computed query under the current transform; motion_step = the last update's share.
  float motion_acc, motion_eps, motion_step;
  // the step the NEXT update is expected to make at most: the last one times the contraction it showed (0 before there are two) --
  // what the cold kernels' forecast of a warm-started iteration's searches is counted against
  float motion_pred;
  // TieDev::counters of the context's searches live HERE (the context's state only): the host reads them with the loop state, one copy
  unsigned int tie_counters[4];
};

enum IterMetric { IM_NONE = 0, IM_KABSCH = 1, IM_PLANE = 2, IM_POINT = 3, IM_BOTH = 4,
                  // moments of the 12-unknown affine normal equations (transform_estimation.hpp:369-476), three streaming passes:
                  IM_AFF0 = 5,    // [0] n, [1..6] sum s s^T (upper), [7..9] sum s, [10..18] sum s_a d_r, [19..21] sum d, [22..33] sum (n.d) n_j (s,1)_a
                  IM_AFF1 = 6,    // sum n_j n_k (s,1)_a (s,1)_b, (j,k) = (0,0),(0,1),(0,2) x the 10 pairs a <= b
                  IM_AFF2 = 7,    // the same for (j,k) = (1,1),(1,2),(2,2)
                  // ... and the same moments in ONE pass on the matrix cores (affine_device.hpp: rows of AFF_ROW doubles, layout AffSlot), the
                  // loops of the affine classes without per-pair weights: IM_AFFC with the target's normals, IM_AFFP without (point terms only)
                  IM_AFFC = 8, IM_AFFP = 9 };
inline bool im_affine_fused(int m) { return m == IM_AFFC || m == IM_AFFP; }
// One row of partial sums of the one-pass affine moments: 94 used of 128 (affine_device.hpp: what they are, where they sit).
constexpr int AFF_ROW = 128;
constexpr int AFF_SUMS = 94;

// Correspondence weight evaluators of the combined-metric classes (core/common_pair_evaluators.hpp:14-27 Identity, :30-43
// Unity, :46-80 RBF kernel over squared distances), selected per term type.  enabled: some evaluator is not Unity -- the
// accumulation then forms the reference's per-pair f32 weight metric_weight * evaluator(corr.value)
// (registration/transform_estimation.hpp:301-303, :330-332) and the solver is handed unit metric weights.
enum { CW_UNITY = 0, CW_IDENTITY = 1, CW_RBF = 2 };
struct CorrWeights {
  int enabled;
  int point_kind, plane_kind;
  float point_coeff, plane_coeff;   // RBF: -0.5 / sigma^2 (common_pair_evaluators.hpp:53)
  float w_p2p, w_p2pl;              // the metric weights, folded into the per-pair weight
  // a caller's own evaluators (cilhip_set_pair_weight_callback): the weights of every stored correspondence, evaluated on the host, by
  // position in the stream the accumulation pass walks (sorted source position, or pair index of a pair list); null: the kinds above
  const float* point_table;
  const float* plane_table;
};

struct F3 { float x, y, z; };      // 12-byte record (one global_load_dwordx3 per lane)

// Which of several EXACTLY equidistant nearest target points a correspondence names (option "tie_rule"; tie_build.hip has the
// why and the host side).  mode 0: the lowest target index -- what the packed keys give by themselves, nothing below is read.
// mode 1: the point the reference's kd-tree traversal meets first.  Every search notices when the smallest distance was met on a
// second point (an equality test beside the key compare); such a query is then looked at once more by tie_settle(): with the
// order tables of the reference's tree loaded, all points at exactly that distance are enumerated (the closed ball of that
// radius) and the first-met one is taken; without them the query is COUNTED (counters[0]) and keeps the lowest index -- the host
// builds the tables and runs the search / loop again (tie_rule "auto": a target whose searches never tie never pays for a tree).
struct TieDev {
  const uint2* leaf_slot;   // [grid.n] by sorted target position: {leaf node, slot in the reference's permutation}; null: no tables loaded
  const uint4* nodes;       // TieNode records: {parent, (depth << 3) | (split dimension << 1) | is-second-child, bits(divlow), bits(divhigh)}
  unsigned int* counters;   // [4]  0: tied queries met without tables, 1: tied queries resolved with them, 2: of those, matches that are not the lowest index,
                            //      3: (reverse searches of FIRST_TO_SECOND / BOTH) tied target points met without the transformed source's tables
  int mode;
};

// Six-dimensional feature search (correspondence_search/common_transformable_feature_adaptors.hpp): features = (point, w * v), v a
// per-point 3-vector (normal or colour).  mode: how the SOURCE's feature part follows the transform --
//   0: rigid -- L * (w v)                                                         PointNormalFeaturesAdaptor, Isometry   :104-111
//   1: affine -- nw * normalized(L^-T (w v)), M = L^-T, nw = |w v_0|              PointNormalFeaturesAdaptor, otherwise  :112-124
//   2: not at all -- w v                                                          PointColorFeaturesAdaptor              :236-243
struct FeatSpec {
  const float4* src;     // the source's feature vectors in the order of the searched source array
  const float4* dst;     // the target's, in sorted-target order
  float w;               // weight of the first attribute (may be 0 under the 9-D adaptor: then only the second one counts)
  int enabled;           // a feature adaptor is in force: candidates are compared by feature distance (0 = plain point features)
  int mode;
  float M[9];            // row-major L^-T (mode 1)
  float nw;
  // 9-D point + normal + colour features (PointNormalColorFeaturesAdaptor, common_transformable_feature_adaptors.hpp:255-343): a
  // SECOND attribute that does not follow the transform, weight w2; nullptr = 6-D
  const float4* src2;
  const float4* dst2;
  float w2;
};

struct IterArgs {
  GridDev grid;
  const float4* src;       // [ns] source sorted by target-grid cell {x,y,z,orig_idx}
  const float4* src_nrm;   // [ns] source normals in the same order, or nullptr (non-null => symmetric metric)
  uint32_t ns;
  float max_sq;            // engine max_distance_ (squared)
  float dst_mean[3];
  float tile_axes[9];      // half-axes of a tile cube in SOURCE space (row-major 3x3: component j of axis k), see k_search_tiled
  const IcpState* state;   // T, smt, inner tform, done flags
  uint32_t* nn_pos;        // [ns] (sorted-source order) sorted-target position or NONE
  float* nn_d2;            // [ns], or null: the squared distances are not stored (the ICP loop without post-filters never reads them)
  double* partials;        // [nblocks * SUMS_MAX]
  unsigned long long* defer_mask;  // [ntiles * 2 * (TILE_THREADS / 64)] tiled search: queries handed to its clean-up pass, one word per wave and query slot
  double* tile_partials;   // [ntiles * SUMS_MAX] tiled search with in-tile accumulation: one row of partial sums per tile (followed by the clean-up pass's rows)
  uint32_t* defer_flag;    // [1] set by a tile that defers a query: the clean-up pass has work
  uint32_t* unproven_cnt;  // [128] (spread by tile / block index) [0, 64): queries the octant block did not prove (how far the source is from
                           // alignment; the warm-started kernel: queries without a usable bound), [64, 128): queries the warm-started kernel listed
  int store_matches;       // tiled search: also write nn_pos / nn_d2 for the queries settled inside the tile (the pure ICP loop needs neither)
  int skip_if_inner_done;
  int no_centering;        // affine point-to-point class: moments of the raw coordinates (no means subtracted)
  FeatSpec feat;               // 6-D feature search (feat.w > 0): feature vectors, weight, how the source's part follows the transform
  CorrWeights cw;              // per-correspondence weights (enabled = 0: unity)
  const uint32_t* warm_pos;    // [ns] or null: matches of the previous iteration, the per-lane search's warm start (may alias nn_pos)
  float warm_far_sq;           // warm bounds at or above this (squared) are counted in unproven_cnt
  float warm_extra;            // (k_warm) a listed query's search reaches this fraction of a cell beyond its bound: the room of its fresh margin
  float4* warm_rec;            // [ns] or null: per query {matched point, its safe2 entry (< 0: no match)}
  F3* warm_rec_n;              // [ns]: the matched point's normal
  F3* warm_src3;               // [ns]: the sorted source points without their index (12 B instead of 16)
  const float* safe2;          // [grid.n] per sorted target point: lower bound on the squared distance to its nearest other target point (k_self_nn)
  // Margin keys (warm-started iterations).  A search that PROVES its result also knows a lower bound Lb on the
  // distance from the query to every OTHER target point (second smallest distance inside the searched block, capped by the gap to
  // the block's faces).  Stored per query as  B = +-(Lb - motion_eps + motion_acc)  (rounded down; sign: + match, - no match, then
  // Lb bounds the distance to EVERY target point; +0 / -FLT_MIN: no bound known): under a later transform the same bound reads
  // B - motion_acc' - motion_eps'.  nn_lb: written next to nn_pos by the search-only forms (null: not wanted); the accumulating
  // tile kernel writes it straight into its match records (warm_rec[i].w) when warm_rec is set.
  float* nn_lb;
  int lb_valid;                // (record-writing warm kernel) nn_lb holds the keys of the search that left warm_pos
  TieDev tie;                  // option "tie_rule"
};

// What k_tile_boxes needs to compute the tiles' regions for the state's transform.
struct BoxArgs {
  const float4* tile_center;   // null: nothing to do
  int* tile_box;
  uint32_t ntiles;
  uint32_t* defer_flag;
  float ox, oy, oz, inv_cell;
  int nx, ny, nz;
  float tile_axes[9];
  int trim;
};

// Loop state the device publishes to the host after every iteration (pinned, host-coherent memory the epilogue kernel
// writes directly: no copy engine, no event).  Iteration k goes to slot k % 4, then -- after a system-scope fence -- the
// `latest` word: the host is never more than two iterations ahead of the device, so the slot it reads (the latest published
// one) cannot be the one the device is writing -- a snapshot is never torn.
struct FeedbackSlot {
  unsigned int unproven;        // IcpState::unproven of that iteration
  unsigned int listed;          // IcpState::listed of that iteration
  float delta;                  // IcpState::delta (last_delta_norm_) of that iteration
  float prev_delta;             // ... and of the one before it
  float step;                   // IcpState::motion_step: how far a source point can have moved in that iteration's update
  float pad;
  unsigned long long commit;    // (run tag << 32) | iterations performed: sanity check of the slot
};
struct Feedback {
  FeedbackSlot slot[4];
  unsigned long long latest;    // (run tag << 32) | (converged ? 1u << 31 : 0) | iterations performed
};

struct SolveArgs {
  Feedback* feedback;      // device-visible address of the host's Feedback (or null)
  unsigned int run_tag;
  IcpState* state;
  const double* partials;
  int nblocks;             // 0: sums already reduced in `reduced`
  const double* reduced;   // [SUMS_MAX] (multi-GPU: all-reduced buffer)
  int metric;              // IterMetric
  float w_p2p, w_p2pl;
  int point_weighted;      // per-pair weights in the sums (CorrWeights::enabled): the point block's count is slot 43, sum of weights
  float conv_tol, opt_conv_tol;
  float dst_mean[3], src_mean[3];
  int gn_last_step;        // finalize the outer iteration after this GN step
  int gn_zero_steps;       // max_optimization_iterations == 0: the estimator's loop does not run, tform = t_dst * I * t_src (transform_estimation.hpp:281, :365)
  int has_normals;
  int affine_centered;     // (IM_AFFC / IM_AFFP) 1: the combined-metric class (means subtracted, tform = t_dst * tform * t_src), 0: the point-to-point class
  // spatially sharded runs (slab partition of target and source along one axis): after every update of the transform the
  // epilogue bounds how far ANY source point (global bounding box of the source, in source coordinates) can have moved
  // along the slab axis since the partition was made, and raises IcpState::slab_violation when that exceeds the slack the
  // halos were sized with -- every rank evaluates the same bound on the same values, so all take the same decision
  uint32_t* unproven_cnt;  // [128] counters of the search kernels, summed into IcpState::unproven / ::listed and zeroed by the epilogue (or null)
  int guard_axis;          // -1: off
  float guard_slack;
  float guard_center[3], guard_half[3];
  float guard_T[16];       // transform the partition was made under
  // bounding box of the source in SOURCE coordinates (centre, half extents): the epilogue bounds how far a query can move per update
  float src_center[3], src_half[3];
  uint4* trace;            // [RUN_TRACE_CAP] or null: per iteration {unproven, listed, bits(motion_step), bits(delta)} (cilhip_get_last_run_trace)
};
constexpr int RUN_TRACE_CAP = 256;

#if defined(__HIPCC__)
// ---- option "tie_rule": the reference's choice among exactly equidistant nearest points (TieDev above) --------------
// Does the reference's traversal for query q reach the target point at sorted position pa before the one at pb?  Same leaf: the lower
// slot of the reference's permutation.  Otherwise walk both leaves up to their lowest common ancestor (parents + depths); there
// nanoflann's searchLevel (nanoflann.hpp:1931-1947) descends first into the child on the query's side of the split:
// (val - divlow) + (val - divhigh) < 0 -> the first child.  (tests/cpp/tie_order_host.hpp: before(), the host restatement; pinned against the reference's own
// nanoflann by tests/test_tie_order_cpu.py.)
__device__ __forceinline__ bool tie_before(const TieDev& tt, float qx, float qy, float qz, uint32_t pa, uint32_t pb) {
  const uint2 la = tt.leaf_slot[pa], lb = tt.leaf_slot[pb];
  if (la.x == lb.x) return la.y < lb.y;
  uint32_t na = la.x, nb = lb.x;
  uint4 A = tt.nodes[na], B = tt.nodes[nb];
  uint32_t a_second = 0;      // is the node on a's path just below the common ancestor a SECOND child
  while ((A.y >> 3) > (B.y >> 3)) { a_second = A.y & 1u; na = A.x; A = tt.nodes[na]; }
  while ((B.y >> 3) > (A.y >> 3)) { nb = B.x; B = tt.nodes[nb]; }
  while (na != nb) { a_second = A.y & 1u; na = A.x; A = tt.nodes[na]; nb = B.x; B = tt.nodes[nb]; }
  const uint32_t feat = (A.y >> 1) & 3u;
  const float val = feat == 0u ? qx : (feat == 1u ? qy : qz);
  const float diff1 = __fsub_rn(val, __uint_as_float(A.z)), diff2 = __fsub_rn(val, __uint_as_float(A.w));
  const uint32_t first_is_second = __fadd_rn(diff1, diff2) < 0.0f ? 0u : 1u;
  return a_second == first_is_second;
}
// The transformed feature part of a source point (the three adaptor behaviours of FeatSpec::mode).  f32, the engine's pinned
// 3-term pairing r0*x + (r1*y + r2*z) for every matrix-vector product (Eigen's unrolled redux); the affine case's norm and
// divisions are the correctly rounded f32 ones (formed in f64: the device's f32 sqrt / divide instructions are not).
__device__ __forceinline__ void source_feature(const FeatSpec& a, const float* T /*col-major 4x4*/, const float4 sn, float& fx, float& fy, float& fz) {
  const float wx = __fmul_rn(a.w, sn.x), wy = __fmul_rn(a.w, sn.y), wz = __fmul_rn(a.w, sn.z);
  if (a.mode == 2) { fx = wx; fy = wy; fz = wz; return; }
  if (a.mode == 1) {
    const float* M = a.M;
    const float v0 = __fadd_rn(__fmul_rn(M[0], wx), __fadd_rn(__fmul_rn(M[1], wy), __fmul_rn(M[2], wz)));
    const float v1 = __fadd_rn(__fmul_rn(M[3], wx), __fadd_rn(__fmul_rn(M[4], wy), __fmul_rn(M[5], wz)));
    const float v2 = __fadd_rn(__fmul_rn(M[6], wx), __fadd_rn(__fmul_rn(M[7], wy), __fmul_rn(M[8], wz)));
    const float nrm = (float)sqrt((double)__fadd_rn(__fmul_rn(v0, v0), __fadd_rn(__fmul_rn(v1, v1), __fmul_rn(v2, v2))));
    // (Eigen's normalized() leaves a vector of norm 0 as it is: a zero normal, or a zero normal weight under the 9-D adaptor)
    if (!(nrm > 0.0f)) { fx = __fmul_rn(a.nw, v0); fy = __fmul_rn(a.nw, v1); fz = __fmul_rn(a.nw, v2); return; }
    fx = __fmul_rn(a.nw, (float)((double)v0 / (double)nrm));
    fy = __fmul_rn(a.nw, (float)((double)v1 / (double)nrm));
    fz = __fmul_rn(a.nw, (float)((double)v2 / (double)nrm));
    return;
  }
  fx = __fadd_rn(__fmul_rn(T[0], wx), __fadd_rn(__fmul_rn(T[4], wy), __fmul_rn(T[8], wz)));
  fy = __fadd_rn(__fmul_rn(T[1], wx), __fadd_rn(__fmul_rn(T[5], wy), __fmul_rn(T[9], wz)));
  fz = __fadd_rn(__fmul_rn(T[2], wx), __fadd_rn(__fmul_rn(T[6], wy), __fmul_rn(T[10], wz)));
}
// squared distance of two 6-D features exactly as nanoflann's L2_Adaptor::evalMetric forms it for DIM = 6 (nanoflann.hpp:570-604): one
// group of four -- ((d0*d0 + d1*d1) + d2*d2) + d3*d3 -- then the tail loop adds d4*d4 and d5*d5 one by one; every operation
// individually rounded.  (a - b) and (b - a) square to the same bits: the value does not depend on which side is the query.
__device__ __forceinline__ float d6_features(float ax, float ay, float az, float afx, float afy, float afz, float bx, float by, float bz, float bfx, float bfy,
                                             float bfz) {
  const float d0 = __fsub_rn(ax, bx), d1 = __fsub_rn(ay, by), d2 = __fsub_rn(az, bz);
  const float d3 = __fsub_rn(afx, bfx), d4 = __fsub_rn(afy, bfy), d5 = __fsub_rn(afz, bfz);
  float r = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
  r = __fadd_rn(r, __fmul_rn(d4, d4));
  return __fadd_rn(r, __fmul_rn(d5, d5));
}
// ... and for DIM = 9 (point, w v, w2 c): two groups of four, result = 0 + g1, result += g2, then the tail adds d8*d8.
__device__ __forceinline__ float d9_features(float ax, float ay, float az, float afx, float afy, float afz, float agx, float agy, float agz, float bx, float by, float bz,
                                             float bfx, float bfy, float bfz, float bgx, float bgy, float bgz) {
  const float d0 = __fsub_rn(ax, bx), d1 = __fsub_rn(ay, by), d2 = __fsub_rn(az, bz);
  const float d3 = __fsub_rn(afx, bfx), d4 = __fsub_rn(afy, bfy), d5 = __fsub_rn(afz, bfz);
  const float d6 = __fsub_rn(agx, bgx), d7 = __fsub_rn(agy, bgy), d8 = __fsub_rn(agz, bgz);
  const float g1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
  const float g2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d4, d4), __fmul_rn(d5, d5)), __fmul_rn(d6, d6)), __fmul_rn(d7, d7));
  return __fadd_rn(__fadd_rn(g1, g2), __fmul_rn(d8, d8));
}
#endif

// kernels.hip
void launch_iter(const IterArgs& a, int metric, bool search, bool store, int nblocks, hipStream_t s);
// warm-started search + accumulation (a.warm_pos / a.nn_pos: previous / new matches, may alias); nblocks: a multiple of 8
void launch_self_nn(const GridDev& g, float* safe2, hipStream_t s);
// rec: 1 = gather through warm_pos (+ the searches' margin keys in nn_lb) and write the match records, 2 = read the match records (a.warm_rec)
void launch_warm(const IterArgs& a, int metric, int rec, int nblocks, hipStream_t s);
// kernel timing without extra packets: the NEXT launch_warm / launch_search_tiled of this thread attaches these events to its own
// kernels' dispatch packets (start of its first kernel, stop of its last)
void set_launch_events(hipEvent_t start, hipEvent_t stop);
void launch_interleave_pn(const float4* pts, const float4* nrm, uint32_t n, float4* pn, hipStream_t s);
void launch_copy_src3(const float4* src_sorted, uint32_t ns, F3* out, hipStream_t s);   // a.warm_src3 of the record-reading form
int warm_num_blocks(uint32_t ns);      // blocks (= partial-sum rows) of launch_warm
void launch_solve(const SolveArgs& a, hipStream_t s);
// acc_metric IM_NONE: search only; IM_KABSCH / IM_PLANE / IM_POINT / IM_BOTH: search + accumulation inside the tile
// (first Gauss-Newton step), leaving tiled_partial_rows(ntiles) rows in a.partials
void launch_search_tiled(const IterArgs& a, int acc_metric, const uint2* tiles, const float4* tile_center, int* tile_box /*[8*ntiles] scratch*/,
                         uint32_t ntiles, hipStream_t s);
int tiled_partial_rows(uint32_t ntiles);
void launch_count_deferred(const unsigned long long* mask, uint32_t ntiles, uint32_t* out2, hipStream_t s);
void launch_search_group(const IterArgs& a, int lanes /*4, 8, 16*/, hipStream_t s);   // exact search, several lanes per query (small / far-from-alignment clouds)
void launch_search_feat6(const IterArgs& a, hipStream_t s);   // correspondence search over 6-D point+normal features (one lane per query, global memory: small clouds)
// feat_warm.hip: the same search warm-started from the matches in a.nn_pos (in / out), margin test against a.safe2 (k_self_nn's table); listed queries counted in a.unproven_cnt[64 ..)
// acc_metric != IM_NONE: the first Gauss-Newton step's sums in the same pass (feat_warm_blocks(ns) rows of SUMS_MAX doubles in a.partials)
int feat_warm_blocks(uint32_t ns);
void launch_feat_warm(const IterArgs& a, int acc_metric, hipStream_t s);
void launch_search_tiled_feat6(const IterArgs& a, const uint2* tiles, const float4* tile_center, int* tile_box, uint32_t ntiles, hipStream_t s);   // its LDS-tiled form
#ifdef CILHIP_EXP_PHASE_CLOCKS
void debug_dump_phase_clocks();   // dev experiment: per-phase clock sums of k_search_tiled -> stderr
#endif
void launch_reduce_partials(const double* partials, int nblocks, double* stage, double* out, hipStream_t s);
void launch_reduce_stage1_groups(const double* partials, int nblocks, double* stage, int groups, hipStream_t s);
int launch_reduce_stage1(const double* partials, int nblocks, double* stage, hipStream_t s);
// both stages of the reduction and the epilogue: one launch when there are more than 64 rows (ticket: one zeroed word per context; null: the two-kernel path)
void launch_reduce_and_solve(const double* partials, int nblocks, double* stage, unsigned int* ticket, SolveArgs a, hipStream_t s);
constexpr int REDUCE_STAGE_DOUBLES = 128 * SUMS_MAX;
// affine.hip -- the affine classes' loop on the device: one-pass moments over stored matches (IM_AFFC / IM_AFFP; a.partials: rows of
// AFF_ROW doubles), their fixed-order reduction and the 12-unknown solve + compose + loop state in one epilogue
int affine_acc_blocks(uint32_t ns);
void launch_acc_affine(const IterArgs& a, int metric, int nblocks, hipStream_t s);
void launch_reduce_and_solve_affine(const double* partials, int nrows, double* stage /*[32 * AFF_ROW]*/, const SolveArgs& a, hipStream_t s);
void launch_init_state(IcpState* st, const float T0[16], const float src_mean[3], hipStream_t s, Feedback* fb = nullptr, unsigned int run_tag = 0,
                       const float* src_center = nullptr, const float* src_half = nullptr, unsigned int* tie_counters = nullptr /*[4], zeroed*/);
void launch_scatter_nn(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos,
                       const float* nn_d2, uint32_t ns, uint32_t* out_idx, float* out_d2,
                       hipStream_t s);
void launch_gather_by_w(const float4* src_sorted, const float* in_xyz, uint32_t ns, float4* out, hipStream_t s);
// out[i] = in[original index of the source point at sorted position i]  (one float per point)
void launch_gather1_by_w(const float4* src_sorted, const float* in, uint32_t ns, float* out, hipStream_t s);
void launch_pack_keys(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos, const float* nn_d2,
                      uint32_t ns, uint32_t index_offset, unsigned long long* keys, hipStream_t s);
void launch_keys_to_pos(const float4* src_sorted, const unsigned long long* keys, const uint32_t* inv_perm, uint32_t ns,
                        uint32_t index_offset, uint32_t n_local, uint32_t* nn_pos, float* nn_d2, hipStream_t s, unsigned int* tie_counter = nullptr);
// the reference's tie order across target shards: traversal keys of the matches at the winning distance / the selection after their MIN
void launch_order_keys(const float4* src_sorted, const IcpState* state, const unsigned long long* win, const uint32_t* nn_pos, const float* nn_d2,
                       uint32_t ns, const TieDev& tt, unsigned long long* own, unsigned long long* out, hipStream_t s);
void launch_select_ordered(const float4* src_sorted, const unsigned long long* own, const unsigned long long* reduced, const unsigned long long* win,
                           uint32_t ns, uint32_t* nn_pos, float* nn_d2, hipStream_t s);
void launch_inv_perm(const float4* dst_sorted, uint32_t n, uint32_t* inv, hipStream_t s);
void launch_count_found(const uint32_t* nn_pos, uint32_t ns, unsigned long long* out, hipStream_t s);
// queries (sorted source under T) whose nearest target point within the radius is not unique in the pinned f32 distance
// the order tables by ORIGINAL target index -> by sorted position (TieDev::leaf_slot)
// q = fl(T s) for every source point (original order), the search kernels' pinned expression, T held by the host (bidir.hip)
void launch_transform_original_host_T(const float* d_src_xyz, uint32_t ns, const float T[16], float* d_out, hipStream_t s);
// One node of the order tables (TieDev::nodes reads it as a uint4).
struct TieNode {          // 16 bytes: one load on the device
  int32_t parent;         // -1: the root
  uint32_t info;          // (depth << 3) | (split dimension << 1) | (1: this node is its parent's SECOND child)
  float divlow, divhigh;  // internal nodes: the split; a leaf: divlow = the slot of its first point (as bits)
};
// tie_build.hip: the order tables of the index the reference builds over a cloud (nanoflann 1.7.1 divideTree / middleSplit_ /
// planeSplit, leaf size 10), level by level on the device.  Input: d_xyz -- the cloud in its ORIGINAL order, 3 floats per point -- or
// d_sorted -- {x, y, z, bits(original index)} records in any order (exactly one non-null).  Output by ORIGINAL index: the leaf node of
// every point and its slot in the reference's permutation (device arrays [n] the caller provides); *d_nodes_out: the TieNode
// records (hipMalloc'ed: the caller frees), breadth-first ids.
extern int g_knn_tie_rule;      // knn.hip: cilhip_knn_set_tie_rule (k-NN lists and the KMeans kd branch)
hipError_t tie_order_build_device(const float* d_xyz, const float4* d_sorted, uint32_t n, hipStream_t s, uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index,
                                  uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out);
// ... and of the tree a feature adaptor's search walks (DIM = 6 / 9: points + w1 * att1 [+ w2 * att2], attributes by sorted position);
// TieNode::info there = (depth << 5) | (split dimension << 1) | second child (tie_before_nd)
hipError_t tie_order_build_device_features(int dim, const float4* d_sorted, const float4* att1, float w1, const float4* att2, float w2, uint32_t n, hipStream_t s,
                                           uint32_t* d_leaf_by_index, uint32_t* d_slot_by_index, uint4** d_nodes_out, size_t* n_nodes_out, int* max_depth_out);
void launch_tie_tables_by_position(const float4* dst_sorted, uint32_t n, const uint32_t* leaf_by_index, const uint32_t* slot_by_index, uint2* leaf_slot, hipStream_t s);
void launch_count_ties(const GridDev& g, const float4* src_sorted, uint32_t ns, const float T[16], float max_sq, unsigned long long* out, hipStream_t s);
// squared distances of the stored matches under T, formed again with the search's pinned arithmetic (bit-identical to what the
// search compared): the ICP loop does not store them, a caller of getCorrespondences() after estimate() reads them
void launch_fill_d2(const float4* src_sorted, const float4* dst_sorted, const uint32_t* nn_pos, const float T[16], uint32_t ns, float* nn_d2, hipStream_t s);
void launch_residuals(const IterArgs& a, int metric, float w_p2p, float w_p2pl, float* out, hipStream_t s);
int iter_num_blocks(uint32_t ns);
// accumulation over reverse matches (FIRST_TO_SECOND / BOTH loops without post-filters): element i = target sorted position i,
// rev_pos[i] = position of its match in the source's own grid; mode 1: all, 2: not the reciprocal duplicates, 3: only those
void launch_acc_reverse(const IterArgs& a, int metric, const float4* sgrid_pts, const uint32_t* rev_pos, uint32_t nd, int mode, const uint32_t* fwd_pos,
                        const uint32_t* src_inv, int nblocks, hipStream_t s);

// filters.hip
void launch_filter_fraction(const float4* src_sorted, uint32_t* nn_pos, const float* nn_d2, uint32_t ns, double fraction,
                            unsigned long long* keys, void* state, hipStream_t s);
void launch_filter_one_to_one(const float4* src_sorted, uint32_t* nn_pos, const float* nn_d2, uint32_t ns,
                              unsigned long long* winner, uint32_t n_target, hipStream_t s);
size_t filter_state_bytes();
void launch_select_fraction(const float* d2, uint32_t n, double fraction, unsigned long long* keys, void* state, uint32_t* flags, hipStream_t s);

// bidir.hip -- search directions FIRST_TO_SECOND / BOTH: the correspondence set as a device pair list
struct PairSet {
  uint32_t *first = nullptr, *second = nullptr;   // ORIGINAL target / source indices, ascending (first, second)
  uint32_t *posd = nullptr, *poss = nullptr;      // sorted-target / sorted-source positions of the same pairs
  float* d2 = nullptr;
  uint32_t *first2 = nullptr, *second2 = nullptr, *posd2 = nullptr, *poss2 = nullptr;   // ping-pong buffers of the post-filters
  float* d2b = nullptr;
  float4 *src_view = nullptr, *nrm_view = nullptr;   // sorted-source records (and normals) gathered per pair: what the accumulation streams over
  size_t cap = 0;
  uint32_t count = 0;
  // workspace of find_pairs (reverse matches, sort keys / slots, flags, scan temporaries ...): allocated once per size
  static constexpr int WS_COUNT = 15;
  void* ws[WS_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t ws_cand = 0, ws_tmp_bytes = 0, ws_nd = 0, ws_ns = 0;      // what the workspace was sized for: candidates, scan scratch, target / source points
};
void free_pairs(PairSet& p);
// the reverse search of those directions alone (every target point against the source, through the inverse of the state's
// rigid transform computed on the device): rev_pos / rev_d2 [nd] by target sorted position
// feat (optional, w > 0): the reverse matches are the nearest 6-D FEATURES (feat->src in the source grid's order, feat->dst by target position)
// rev_tie (option "tie_rule" for these matches): the reference searches a kd-tree over the TRANSFORMED SOURCE, rebuilt every iteration
// (correspondence_search_kd_tree.hpp:185-222): leaf_slot = that tree's order tables by position in the source grid, valid for the state's
// transform only (built per search: c_api.hip build_rev_tie_tables); null = count the tied target points (counters[3]) and keep the lowest source index
// warm_src_safe2 (plain point features only): rev_pos holds the PREVIOUS iteration's reverse matches, the table is k_self_nn's over the source grid -- k_reverse_warm
// settles every target point whose old match passes the margin test without a search and searches the rest (the same exact result)
// fused (with warm_src_safe2): the first Gauss-Newton step's sums over the reverse matches accumulated in the same kernel (rows of SUMS_MAX doubles, one per block:
// reverse_warm_blocks(nd) of them) instead of by launch_acc_reverse afterwards; mode as launch_acc_reverse's
struct RevFused { int metric; int mode; const uint32_t* fwd_pos; const uint32_t* src_inv; const uint32_t* grid_to_sorted; double* partials; float dst_mean[3]; };
void launch_grid_to_sorted(const float4* sgrid_pts, uint32_t ns, const uint32_t* src_inv, uint32_t* out /*[ns] by source-grid position*/, hipStream_t s);
int reverse_warm_blocks(uint32_t nd);
void launch_reverse_search_rigid(const GridDev& g, const GridDev& src_grid, const IcpState* state, float max_sq, uint32_t* rev_pos, float* rev_d2, hipStream_t s,
                                 const FeatSpec* feat = nullptr, const TieDev* rev_tie = nullptr, const float* warm_src_safe2 = nullptr, const RevFused* fused = nullptr);
hipError_t find_pairs(const FeatSpec& feat, const GridDev& g, const GridDev& src_grid /*over the source, SOURCE coordinates*/, const float* d_src_xyz, const float* d_src_nrm,
                      const float4* src_sorted, uint32_t ns, const IcpState* state, const IcpState* id_state, const float T_host[16], float max_sq,
                      int direction, bool reciprocal, double inlier_fraction, bool one_to_one, const uint32_t* fwd_pos, const float* fwd_d2,
                      PairSet& out, hipStream_t s, const TieDev* rev_tie = nullptr);

// grid_build.hip
struct GridBuildResult {
  GridDev grid;
  double avg_occupancy;
  size_t n_cells;
};
// Builds the grid for n points (device xyz, optional device normals).  Allocates the sorted
// arrays and the cell table (freed by free_grid).  Returns hipSuccess or an error.
// refined_factor: a cloud whose density-based first guess leaves far too many points per cell (a surface, clusters) is refined until
// the expected own-cell population is at most 3 x target x refined_factor (1: as dense a grid as for a volumetric cloud).
hipError_t build_grid(const float* d_xyz, const float* d_nrm, uint32_t n, hipStream_t s,
                      GridBuildResult* out, double mean_out[3], double target_occupancy, double refined_factor = 1.0);
void free_grid(GridDev& g);
// Sorts the source by the target-grid cell of T*s; writes {x,y,z,orig} records.  d_out preallocated [n].
// Also emits the tile table of the LDS-tiled search kernel: tiles[t] = [begin,end) of <= TILE_QUERIES sorted
// queries that share one 4x4x4-cell cube (caller frees *d_tiles_out with hipFree).
struct SortWorkspace {      // scratch + tile-table storage a caller keeps between sort_source calls (free_sort_workspace)
  void* scratch = nullptr; size_t scratch_bytes = 0;
  uint2* tiles = nullptr; float4* centers = nullptr; uint32_t tile_cap = 0;
};
hipError_t sort_source(const float* d_xyz, uint32_t n, const GridDev& g, const float T[16], float4* d_out,
                       hipStream_t s, uint2** d_tiles_out, float4** d_tile_center_out, float tile_axes_out[9], uint32_t* ntiles_out,
                       SortWorkspace* ws = nullptr);
void free_sort_workspace(SortWorkspace& ws);
hipError_t mean3_device(const float* d_xyz, uint32_t n, hipStream_t s, double mean_out[3], float* lo_out = nullptr, float* hi_out = nullptr);

// c_api.hip <-> multi.hip
hipStream_t ctx_stream(const ::cilhip_ctx* c);
double ctx_wait_us(const ::cilhip_ctx* c);      // microseconds this context's host loop has spent waiting for published loop state
constexpr float kIdentity16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
// spin-wait courtesy (host)
inline void cpu_relax(unsigned spins) {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
  if ((spins & 255u) == 255u) std::this_thread::yield();      // (the device publishes within tens of microseconds: rarely reached)
}
}  // namespace cilhip
