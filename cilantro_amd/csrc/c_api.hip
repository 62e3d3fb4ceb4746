// c_api.hip -- the C ABI of libcilantro_hip.so (declared in include/cilantro_hip/c_api.h).
// Host-side orchestration only: owns device buffers + stream, enqueues the kernels of kernels.hip /
// grid_build.hip.  No CPU compute fallback exists: without a usable HIP device every call fails.
#include "../../include/cilantro_hip/c_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "internal.hpp"

using namespace cilhip;

// Device allocations of a target that SEVERAL contexts use (cilhip_share_target): freed when the last of them lets go.
struct TargetShare { int refs = 0; std::vector<void*> allocs; };

struct cilhip_ctx {
  int device = 0;
  TargetShare* tshare = nullptr;  // non-null: some of this context's target pointers belong to a share (target_ptr_free / release_target_share)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // target
  bool has_target = false;
  GridDev grid{};
  bool has_normals = false;
  double grid_occ = 0.0;
  size_t grid_cells = 0;
  double build_ms = 0.0;
  float dst_mean[3] = {0, 0, 0};
  uint32_t index_offset = 0;      // global index of this shard's first target point (target-sharded runs)
  bool partial_target = false;    // this context holds only PART of the cloud the reference would index (an index shard, a spatial slab: cilhip_set_shard_info
                                  // with an offset or the whole cloud's mean): the order tables are the WHOLE cloud's -- loaded (cilhip_load_tie_order), never built here
  uint32_t* d_inv_perm = nullptr; // [n_target] original local index -> sorted position (built on first use)

  // source
  bool has_source = false;
  uint32_t ns = 0;
  float* d_src_xyz = nullptr;     // original order (kept for re-sorting)
  float4* d_src_sorted = nullptr; // sorted cube-major by target-grid cell under sort_T
  SortWorkspace sort_ws;          // scratch + tile table of sort_source, kept between the sorts of a source (d_tiles / d_tile_center point into it)
  uint32_t tile_aux_cap = 0;      // tiles d_tile_box / d_defer_mask are sized for
  uint2* d_tiles = nullptr;       // [ntiles] query ranges of the LDS-tiled search kernel
  float4* d_tile_center = nullptr;  // [ntiles] cube centre of each tile in source space
  int* d_tile_box = nullptr;        // [8*ntiles] cell range of each tile's cube under the current transform (recomputed per search)
  float tile_axes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long* d_defer_mask = nullptr;  // [ntiles * 32] queries the tiles hand to the clean-up pass (bit masks, rewritten by every search)
  uint32_t* d_defer_flag = nullptr;            // [1] "some tile deferred a query" (reset before, set by, every tiled search)
  uint32_t* d_unproven = nullptr;              // [128] queries the tiles' first stage did not prove / the warm-started kernel listed (summed / zeroed by the epilogue)
  bool warm_banned = false;                    // the warm-started form was seen not to pay on this cloud pair (too few queries settled by the table)
  Feedback* h_feedback = nullptr;              // pinned, host-coherent: what the epilogue kernel publishes after every iteration (pacing, kernel form)
  Feedback* d_feedback = nullptr;              // the device's address of it
  unsigned int run_tag = 0;
  bool far_mode = true;                        // tiled ICP loop: the source is far from alignment (many unproven octant searches): search and
                                               // accumulate in two passes (the search's 3x3x3 pass settles them in LDS) instead of one
  int last_fused_iters = 0, last_two_pass_iters = 0, last_warm_iters = 0;
  int run_calls = 0;              // cilhip_icp_partial_sums calls since cilhip_icp_begin
  bool run_warm_on = false;       // sharded runs: the loop has been seen to (nearly) stand still
  unsigned int run_judged = 0;    // ... and the last published iteration whose count of searched queries has been judged
  std::vector<unsigned char> iter_form;   // form of every timed search / one-pass launch of the last run (FORM_*), in launch order
  std::vector<unsigned char> trace_form;  // form of every iteration enqueued by the last run, timed or not (cilhip_get_last_run_trace)
  double form_ms[5] = {0, 0, 0, 0, 0};    // ... and the kernel time summed per form
  int form_n[5] = {0, 0, 0, 0, 0};
  int warm_start = 1;             // option "warm_start": 0 = never, 1 = when the device reports the source near alignment, 2 = from the second iteration on
  uint32_t* d_dbg = nullptr;                   // [2] cilhip_debug_counters scratch
  uint4* d_trace = nullptr;                    // [RUN_TRACE_CAP] per-iteration loop state of the last run, written by the epilogue (cilhip_get_last_run_trace)
  uint32_t ntiles = 0;
                                  // (round 3 experiment, exact, measured 17 % slower than two workgroups per CU: off)
  int tiled = 1;                  // 0: per-lane global-memory search; 1: LDS-tiled search when the cloud is large enough; 2: always tiled
  bool src_sorted = false;
  float sort_T[16];
  float src_mean[3] = {0, 0, 0};
  float* d_src_nrm = nullptr;         // optional source normals, original order (4-cloud ctor => symmetric metric)
  float4* d_src_nrm_sorted = nullptr;
  uint32_t* d_nn_pos = nullptr;
  float* d_nn_d2 = nullptr;
  float4* d_warm_rec = nullptr;   // [ns] float4 + 2 x [ns] F3: match records {matched point, margin key} {normal} and the 12-byte source copy of the warm-started iterations
  bool rec_valid = false;         // the records describe the last executed iteration's matches (inside a run)
  bool src3_valid = false;        // the 12-byte source copy matches d_src_sorted (rewritten after a re-sort)
  float* d_nn_lb = nullptr;       // [ns] margin keys the search-only tile kernel leaves next to nn_pos (IterArgs::nn_lb)
  bool lb_fresh = false;          // ... and they belong to the search that left nn_pos (inside a run)
  bool warm_forecast = true;      // option "warm_forecast": the cold kernels' count of the queries a warm-started iteration would have to search gates the form
  // option "tie_rule": which of several EXACTLY equidistant nearest target points a correspondence names.  0 = the lowest target index;
  // 1 = the one the reference's kd-tree traversal meets first, order tables built before the first search; 2 (default) = the same
  // choice, the tables built when a search first MEETS a tie (that search / run is then executed again): a target that never ties never
  // pays for a tree.  The device resolves ties inside its search kernels (TieDev, kernels.hip: tie_settle).
  int tie_rule = 2;
  uint2* d_tief_leaf_slot = nullptr;             // [grid.n] the order tables of the FEATURE tree (6-D / 9-D adaptors: points + weighted normals / colours), for the
  uint4* d_tief_nodes = nullptr;                 // feature options they were built under (dropped with any of them); TieNode::info with four dimension bits
  int tief_builds = 0;
  uint2* d_tie_leaf_slot = nullptr;              // [grid.n] the order tables by sorted target position (null: not loaded)
  uint4* d_tie_nodes = nullptr;
  unsigned int* d_tie_counters = nullptr;        // [4] TieDev::counters
  unsigned int* d_ticket = nullptr;              // [1] k_reduce_solve's ticket (zero between launches)
  // option "group_search": the global-memory search with SEVERAL lanes per query (k_search_group: small clouds and sources far from
  // alignment, where one lane per query leaves the chip idle behind chains of dependent trips).  -1 (default) = the ICP loop decides per
  // iteration (cold iterations of clouds the tiles do not take: always for clouds below the warm-started form's floor, from the
  // kernels' own forecast above it); 0 = never; 4 .. 64 = that many lanes in every global-memory search.
  int group_lanes = -1;
  double wait_us = 0.0;                          // time spent waiting for the device to publish loop state (wait_published), accumulated: not enqueue work
  bool feat_warm = true;                         // option "feature_warm_start": the feature adaptors' forward search warm-started from the previous matches once the loop moves little (feat_warm.hip)
  bool affine_device_loop = true;                // option "affine_device_loop": the affine classes' loop device-resident (one-pass moments on the matrix cores, 12-unknown
                                                 // solve in the epilogue kernel) whenever nothing needs the stored set per iteration; 0 = the host-driven loop (A/B)
  bool fused_epilogue = false;                   // option "fused_epilogue": stage-1 reduction + epilogue in ONE launch (the last of the 32 stage-1 blocks runs the
                                                 // epilogue).  Bitwise the same results, measured SLOWER: 0.129 -> 0.136 ms per iteration at 10M, 0.037 -> 0.044 at 1M --
                                                 // a device-scope fence costs more on this eight-L2 part than the kernel boundary it removes (NOTEBOOK.md): off
  unsigned int tie_counters_host[4] = {0, 0, 0, 0};      // ... as read together with the loop state at the end of a run (read_state: one synchronisation for both)
  bool tie_counters_fresh = false;
  double tie_build_ms = 0.0;                     // host time of the last table build (tree + upload)
  int tie_builds = 0;                            // table builds on this context (diagnostics)
  // the reverse matches of FIRST_TO_SECOND / BOTH: the reference's tree is over the TRANSFORMED source, a new one per search -- once a
  // reverse search has met exactly equidistant source points (or under tie_rule 1) that tree's order tables are built (on the device) before
  // every reverse search (the loops then run host-driven, one search at a time)
  bool rev_tie_aware = false;
  uint2* d_rev_tie_leaf_slot = nullptr;          // [ns] by position in the source grid; valid for rev_tie_T only
  uint4* d_rev_tie_nodes = nullptr;
  size_t rev_tie_nodes_cap = 0;
  bool rev_tie_valid = false;
  float rev_tie_T[16];
  int rev_tie_builds = 0;
  float warm_extra = 0.0625f;     // option "warm_extra_fraction"
  bool pair_records = true;       // option "pair_records": the streaming accumulation gathers a match's point and normal from one 32-byte record (GridDev::pn)
  void* rank_comm = nullptr; int rank_comm_size = 0; double* d_rank_sums = nullptr;      // cilhip_rank_comm_*: this process' rank in an RCCL communicator
  bool tile_records = true;       // option "tile_records": the accumulating tile kernel writes the warm-started form's match records itself
  float warm_enter = 0.15f;       // option "warm_enter_fraction": the bar a run starts with, as a fraction of a grid cell
  float warm_thresh = 0.0f;       // a run's bar for (re-)entering the warm-started form: the last update moved no source point by more than this
  int warm_strikes = 0;           // warm iterations of the run that had to search a quarter of their queries
  float src_center[3] = {0, 0, 0}, src_half[3] = {0, 0, 0};   // bounding box of the source (source coordinates): the epilogue's bound on how far a query moves per update
  float* d_safe2 = nullptr;       // [grid.n] k_self_nn's table for the warm-started iteration; built with the target
  int cw_point_kind = 0, cw_plane_kind = 0;     // correspondence weight evaluators (CW_*), combined metric
  float cw_point_sigma = 1.0f, cw_plane_sigma = 1.0f;
  cilhip_pair_weight_fn weight_fn = nullptr;    // a caller's own evaluators (cilhip_set_pair_weight_callback): the estimates call them on the host
  void* weight_user = nullptr;
  float* d_wtab = nullptr;        // [2 * wtab_cap] point / plane weights by stream position (CorrWeights::point_table / plane_table)
  float* d_wtab_in = nullptr;     // [2 * wtab_cap] ... by original source index, as the host filled them
  size_t wtab_cap = 0;
  bool have_nn = false;           // nn_pos/nn_d2 hold the result of a search
  bool d2_stale = false;          // ... but nn_d2 has not been formed yet (matches left by a loop whose kernels keep no distances: ensure_d2)
  float nn_T[16];                 // transform used by that search
  // after cilhip_icp_run the engine's correspondence set is the last executed iteration's (correspondence_search_kd_tree.hpp:231 through
  // icp_base.hpp:32-38): either the loop's kernels left it in nn_pos (have_nn, origin 1) or it is searched again on demand under
  // nn_T = the transform that iteration searched under (pending_matches, origin 2) -- the search is exact, so it is the same set
  bool pending_matches = false;
  float pending_max_sq = 0.0f;
  int matches_origin = 0;         // cilhip_get_last_matches_origin

  // loop state / scratch
  IcpState* d_state = nullptr;
  double* d_partials = nullptr;
  int partial_blocks = 0;
  double* d_stage = nullptr;      // [REDUCE_STAGE_DOUBLES] stage-1 rows of the cross-block reduction
  double* d_sums = nullptr;       // [3 * SUMS_MAX] (the affine estimator reduces three passes before one copy to the host)
  bool tile_acc_adaptive = true;  // choose one pass / two passes per iteration from the device's feedback (option "tile_accumulation" = 2: always one pass)
  bool tile_acc = true;           // accumulate inside the LDS tiles of the search when the engine allows it (option "tile_accumulation", A/B)
  bool fused = false;             // true: search+accumulate in one kernel; false: search kernel + streaming accumulate kernel (faster: the search runs at 2x the occupancy)
  double cell_occupancy = 1.0;    // target points per grid cell (takes effect at the next set_target)
  double refined_occupancy = 3.0; // option "refined_occupancy_factor": how much denser than that a REFINED grid (surface-like / clustered target) may stay
  unsigned long long* d_count = nullptr;
  uint32_t* d_out_idx = nullptr;  // [ns] original-order results
  float* d_out_d2 = nullptr;

  // engine post-filters (correspondence_search_kd_tree.hpp:224-225)
  double inlier_fraction = 1.0;
  bool one_to_one = false;
  unsigned long long* d_keys = nullptr;    // [ns]
  unsigned long long* d_own_order = nullptr;   // [ns] this shard's traversal keys of the current iteration (cilhip_icp_order_keys)
  int tie_max_depth = 0;                   // depth of the loaded order tree (the traversal keys hold 58 levels)
  void* d_sel_state = nullptr;
  unsigned long long* d_winner = nullptr;  // [n_target]

  // other search directions (correspondence_search_kd_tree.hpp:185-222): the correspondence set is a pair list
  int search_dir = 0;             // 0 = SECOND_TO_FIRST (default), 1 = FIRST_TO_SECOND, 2 = BOTH
  bool reciprocal = false;        // require_reciprocality_ (BOTH only)
  int transform_mode = 0;         // 0 = rigid (Isometry), 1 = affine: which ICP instance family cilhip_icp_run mirrors
  float normal_weight = 0.0f;     // > 0: the correspondence search runs on 6-D features (point, weight * v)
  int feature_kind = 0;           // option "feature_kind": 0 = v = normals, following the transform (PointNormalFeaturesAdaptor);
                                  // 1 = v = colours, untouched by it (PointColorFeaturesAdaptor; cilhip_set_color_features)
  float *d_dst_rgb = nullptr, *d_src_rgb = nullptr;             // colour features, original order
  float4 *d_dst_rgb_sorted = nullptr, *d_src_rgb_sorted = nullptr;
  float4* d_src_rgb_grid = nullptr;      // the source's colours in the order of the source's own grid (9-D reverse search)
  float color_weight = 0.0f;             // option "feature_color_weight" (feature_kind 2: the 9-D adaptor's colour weight)
  bool dst_rgb_sorted_ok = false;
  float src_nrm0[3] = {0, 0, 0};  // the first source normal (the affine feature adaptor's normal weight is |w n_0|, adaptors.hpp:113-114)
  float feat_M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};              // L^-T of the transform being searched under (affine adaptor)
  bool symmetric = true;          // source normals, when set, also switch the combined metric to the symmetric objective
  PairSet pairs;
  GridDev src_grid{};             // grid over the source in SOURCE coordinates (built on the first FIRST_TO_SECOND / BOTH search of a source)
  bool has_src_grid = false;
  float* d_src_safe2 = nullptr;   // [ns] k_self_nn's table over the SOURCE grid: the margin test of the warm-started reverse search (k_reverse_warm)
  bool reverse_warm = true;       // option "reverse_warm_start": the device-resident FIRST_TO_SECOND / BOTH loops start every reverse search but the first from the previous matches
  uint32_t* d_grid_to_sorted = nullptr;   // [ns] source-grid position -> sorted source position (d_src_inv through the source grid's order): the fused reverse pass's duplicate test
  uint32_t *d_rev_pos = nullptr, *d_src_inv = nullptr;   // list-free loops of those directions: reverse matches by target position; original -> sorted source position
  float* d_rev_d2 = nullptr;
  bool have_pairs = false;        // `pairs` holds the result of the last find_correspondences
  IcpState* d_state_id = nullptr; // a state holding the identity transform (the reverse search transforms nothing)

  // sharded-run state
  cilhip_icp_params run_prm{};
  bool run_active = false;
  int guard_axis = -1;            // slab-sharded runs: see SolveArgs::guard_*
  float guard_slack = 0.0f, guard_center[3] = {0, 0, 0}, guard_half[3] = {0, 0, 0}, guard_T[16] = {0};
  float run_src_mean[3] = {0, 0, 0};

  // timing
  bool kernel_timing = false;
  int timing_stride = 1;          // option "kernel_timing_stride": with kernel timing on, iterations 0..2 and every stride-th one carry events
  std::vector<unsigned int> timed_iter;      // the iterations of the last run that did
  std::vector<float> timed_ms;               // ... and the kernel time of each (cilhip_get_last_iteration_timing)
  double last_loop_ms = 0.0, last_search_ms = 0.0, last_acc_ms = 0.0;
  int last_search_launches = 0;
  size_t run_nev = 0;             // sharded runs: hipEvents recorded by cilhip_icp_partial_sums since cilhip_icp_begin (3 per call)
  std::vector<hipEvent_t> ev, ev_acc;
  std::vector<hipEvent_t> ev_ar;  // ranked loop: event pairs around the sampled all-reduces since cilhip_icp_begin (cilhip_get_last_allreduce_timing)
  size_t run_nar = 0;             // ... how many of them are recorded
  double last_allreduce_ms = 0.0; int last_allreduce_n = 0;
  double run_enqueue_us = 0.0; int run_enqueue_iters = 0;      // ranked loop: host time of its enqueue calls (the paced waits for the device's feedback word excluded)

};

#define CK(ctx, call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
      return CILHIP_ERR_HIP;                                                                            \
    }                                                                                                   \
  } while (0)

static int fail(cilhip_ctx* c, int code, const char* msg) {
  if (c) c->err = msg;
  return code;
}

static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
// (multi.hip -- the C entry of the multi-device loops -- drives contexts through the public entry points; these two are all it reads of one)
namespace cilhip {
hipStream_t ctx_stream(const cilhip_ctx* c) { return c->stream; }
double ctx_wait_us(const cilhip_ctx* c) { return c->wait_us; }
}  // namespace cilhip

// the stored correspondence set (matches or pair list) no longer describes anything a caller may read
static void drop_src_grid(cilhip_ctx* c) {
  if (c->has_src_grid) { free_grid(c->src_grid); c->has_src_grid = false; }
  if (c->d_src_safe2) { (void)hipFree(c->d_src_safe2); c->d_src_safe2 = nullptr; }
  if (c->d_grid_to_sorted) { (void)hipFree(c->d_grid_to_sorted); c->d_grid_to_sorted = nullptr; }
  if (c->d_src_rgb_grid) { (void)hipFree(c->d_src_rgb_grid); c->d_src_rgb_grid = nullptr; }
}
static void drop_matches(cilhip_ctx* c) { c->have_nn = false; c->d2_stale = false; c->pending_matches = false; c->matches_origin = 0; }
static void drop_rev_tie_tables(cilhip_ctx* c) {  // (they describe ONE source under ONE transform)
  if (c->d_rev_tie_leaf_slot) { (void)hipFree(c->d_rev_tie_leaf_slot); c->d_rev_tie_leaf_slot = nullptr; }
  if (c->d_rev_tie_nodes) { (void)hipFree(c->d_rev_tie_nodes); c->d_rev_tie_nodes = nullptr; }
  c->rev_tie_nodes_cap = 0; c->rev_tie_valid = false; c->rev_tie_aware = false;
}
// a target-side allocation of this context: freed here unless it belongs to a share (then by whoever lets go of the share last)
static void target_ptr_free(cilhip_ctx* c, const void* p) {
  if (!p) return;
  if (c->tshare && std::find(c->tshare->allocs.begin(), c->tshare->allocs.end(), p) != c->tshare->allocs.end()) return;
  (void)hipFree(const_cast<void*>(p));
}
static void release_target_share(cilhip_ctx* c) {
  if (!c->tshare) return;
  if (--c->tshare->refs == 0) {
    for (void* p : c->tshare->allocs) (void)hipFree(p);
    delete c->tshare;
  }
  c->tshare = nullptr;
}
static void drop_feat_tie_tables(cilhip_ctx* c) {      // (one target under one set of feature options; never shared)
  if (c->d_tief_leaf_slot) { (void)hipFree(c->d_tief_leaf_slot); c->d_tief_leaf_slot = nullptr; }
  if (c->d_tief_nodes) { (void)hipFree(c->d_tief_nodes); c->d_tief_nodes = nullptr; }
}
static void drop_tie_tables(cilhip_ctx* c) {      // (they describe ONE target)
  target_ptr_free(c, c->d_tie_leaf_slot); c->d_tie_leaf_slot = nullptr;
  target_ptr_free(c, c->d_tie_nodes); c->d_tie_nodes = nullptr;
}
// everything a context holds of its target (its own allocations are freed, shared ones only let go of)
static void release_target(cilhip_ctx* c) {
  if (c->has_target) {
    target_ptr_free(c, c->grid.pts); target_ptr_free(c, c->grid.nrm); target_ptr_free(c, c->grid.pn); target_ptr_free(c, c->grid.cell_start);
    c->grid.pts = nullptr; c->grid.nrm = nullptr; c->grid.pn = nullptr; c->grid.cell_start = nullptr;
    c->has_target = false;
  }
  target_ptr_free(c, c->d_inv_perm); c->d_inv_perm = nullptr;
  target_ptr_free(c, c->d_safe2); c->d_safe2 = nullptr;
  drop_tie_tables(c);
  drop_feat_tie_tables(c);
  release_target_share(c);
}

extern "C" {

int cilhip_create(cilhip_ctx** out, int device) {
  if (!out) return CILHIP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CILHIP_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return CILHIP_ERR_INVALID;
  cilhip_ctx* c = new (std::nothrow) cilhip_ctx();
  if (!c) return CILHIP_ERR_HIP;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return CILHIP_ERR_HIP;
  }
  c->stream = c->own_stream;
  if (hipMalloc(&c->d_state, sizeof(IcpState)) != hipSuccess || hipMalloc(&c->d_count, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&c->d_defer_flag, sizeof(uint32_t)) != hipSuccess || hipMemset(c->d_defer_flag, 0, sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&c->d_unproven, 128 * sizeof(uint32_t)) != hipSuccess || hipMemset(c->d_unproven, 0, 128 * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&c->d_ticket, sizeof(unsigned int)) != hipSuccess || hipMemset(c->d_ticket, 0, sizeof(unsigned int)) != hipSuccess ||
      hipHostMalloc(&c->h_feedback, sizeof(Feedback), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_feedback), c->h_feedback, 0) != hipSuccess ||
      hipMalloc(&c->d_trace, RUN_TRACE_CAP * sizeof(uint4)) != hipSuccess || hipMemset(c->d_trace, 0, RUN_TRACE_CAP * sizeof(uint4)) != hipSuccess ||
      hipMalloc(&c->d_stage, REDUCE_STAGE_DOUBLES * sizeof(double)) != hipSuccess || hipMalloc(&c->d_sums, 3 * SUMS_MAX * sizeof(double)) != hipSuccess) {
    delete c;
    return CILHIP_ERR_HIP;
  }
  if (hipMemset(c->d_state, 0, sizeof(IcpState)) != hipSuccess) { delete c; return CILHIP_ERR_HIP; }
  c->d_tie_counters = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(c->d_state) + offsetof(IcpState, tie_counters));      // (read with the state: read_state)
  memcpy(c->sort_T, kIdentity, sizeof(kIdentity));
  memcpy(c->nn_T, kIdentity, sizeof(kIdentity));
  *out = c;
  return CILHIP_OK;
}

static void free_source(cilhip_ctx* c) {
  drop_rev_tie_tables(c);
  if (c->d_src_xyz) (void)hipFree(c->d_src_xyz);
  if (c->d_src_sorted) (void)hipFree(c->d_src_sorted);
  if (c->d_nn_pos) (void)hipFree(c->d_nn_pos);
  if (c->d_nn_d2) (void)hipFree(c->d_nn_d2);
  if (c->d_warm_rec) { (void)hipFree(c->d_warm_rec); c->d_warm_rec = nullptr; }
  if (c->d_nn_lb) { (void)hipFree(c->d_nn_lb); c->d_nn_lb = nullptr; }
  c->src3_valid = false; c->lb_fresh = false;
  c->rec_valid = false;
  if (c->d_out_idx) (void)hipFree(c->d_out_idx);
  if (c->d_out_d2) (void)hipFree(c->d_out_d2);
  free_sort_workspace(c->sort_ws);      // (owns d_tiles / d_tile_center)
  c->tile_aux_cap = 0;
  if (c->d_tile_box) (void)hipFree(c->d_tile_box);
  if (c->d_src_nrm) (void)hipFree(c->d_src_nrm);
  if (c->d_src_nrm_sorted) (void)hipFree(c->d_src_nrm_sorted);
  c->d_src_nrm = nullptr; c->d_src_nrm_sorted = nullptr;
  if (c->d_src_rgb) (void)hipFree(c->d_src_rgb);
  if (c->d_src_rgb_sorted) (void)hipFree(c->d_src_rgb_sorted);
  c->d_src_rgb = nullptr; c->d_src_rgb_sorted = nullptr;
  if (c->d_defer_mask) (void)hipFree(c->d_defer_mask);
  if (c->d_keys) (void)hipFree(c->d_keys);
  c->d_keys = nullptr;
  if (c->d_own_order) (void)hipFree(c->d_own_order);
  c->d_own_order = nullptr;
  c->d_tiles = nullptr; c->d_tile_center = nullptr; c->d_tile_box = nullptr; c->ntiles = 0; c->d_defer_mask = nullptr;
  c->d_src_xyz = nullptr; c->d_src_sorted = nullptr; c->d_nn_pos = nullptr; c->d_nn_d2 = nullptr;
  c->d_out_idx = nullptr; c->d_out_d2 = nullptr;
  drop_src_grid(c);
  if (c->d_src_inv) { (void)hipFree(c->d_src_inv); c->d_src_inv = nullptr; }
  if (c->d_grid_to_sorted) { (void)hipFree(c->d_grid_to_sorted); c->d_grid_to_sorted = nullptr; }
  c->has_source = false; c->src_sorted = false; drop_matches(c); c->ns = 0;
  c->have_pairs = false; c->pairs.count = 0;   // a pair list refers to the source / target it was found on
  c->far_mode = true;
  c->warm_banned = false;
}

void cilhip_destroy(cilhip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)cilhip_rank_comm_destroy(c);
  free_source(c);
  release_target(c);
  if (c->d_ticket) (void)hipFree(c->d_ticket);
  if (c->d_wtab) (void)hipFree(c->d_wtab);
  if (c->d_wtab_in) (void)hipFree(c->d_wtab_in);
  if (c->d_dst_rgb) (void)hipFree(c->d_dst_rgb);
  if (c->d_dst_rgb_sorted) (void)hipFree(c->d_dst_rgb_sorted);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_state_id) (void)hipFree(c->d_state_id);
  free_pairs(c->pairs);
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->d_sel_state) (void)hipFree(c->d_sel_state);
  if (c->d_winner) (void)hipFree(c->d_winner);
  if (c->d_count) (void)hipFree(c->d_count);
  if (c->d_dbg) (void)hipFree(c->d_dbg);
  if (c->d_trace) (void)hipFree(c->d_trace);
  if (c->d_defer_flag) (void)hipFree(c->d_defer_flag);
  if (c->d_unproven) (void)hipFree(c->d_unproven);
  if (c->d_rev_pos) (void)hipFree(c->d_rev_pos);
  if (c->d_rev_d2) (void)hipFree(c->d_rev_d2);
  if (c->h_feedback) (void)hipHostFree(c->h_feedback);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->d_sums) (void)hipFree(c->d_sums);
  for (auto e : c->ev) (void)hipEventDestroy(e);
  for (auto e : c->ev_acc) (void)hipEventDestroy(e);
  for (auto e : c->ev_ar) (void)hipEventDestroy(e);

  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* cilhip_last_error(const cilhip_ctx* c) { return c ? c->err.c_str() : "null context"; }

int cilhip_set_stream(cilhip_ctx* c, void* s) {
  if (!c) return CILHIP_ERR_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return CILHIP_OK;
}

int cilhip_synchronize(cilhip_ctx* c) {
  if (!c) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  CK(c, hipStreamSynchronize(c->stream));
  return CILHIP_OK;
}

// ---- the options of a context, as a table a C caller can enumerate and check at compile time (enum cilhip_option in c_api.h):
// id, key, default, admissible range, one line of documentation, how to read the current value back.  cilhip_set_option() below
// validates and applies; the table is what tests/test_capi_symbols.py walks (every option documented, accepted with its default,
// readable, exercised by a test).  Long-form documentation: c_api.h above cilhip_set_option.
namespace {
struct OptionRow { cilhip_option_info_t info; double (*get)(const cilhip_ctx*); };
#define OPT(ID, KEY, DEF, LO, HI, DOC, EXPR) {{ID, KEY, DEF, LO, HI, DOC}, [](const cilhip_ctx* c) -> double { return (double)(EXPR); }}
const OptionRow g_options[] = {
  OPT(CILHIP_OPT_FUSED, "fused", 0, 0, 1, "1 = one per-lane search+accumulate kernel per iteration; 0 = search kernel + streaming accumulation (or the LDS tiles)", c->fused),
  OPT(CILHIP_OPT_INLIER_FRACTION, "inlier_fraction", 1, 0, 1, "CorrespondenceSearchKDTree::setInlierFraction: keep that fraction of the correspondences, nearest first", c->inlier_fraction),
  OPT(CILHIP_OPT_ONE_TO_ONE, "one_to_one", 0, 0, 1, "setOneToOne: a target point keeps only its nearest source point", c->one_to_one),
  OPT(CILHIP_OPT_TILED, "tiled", 1, 0, 2, "LDS-tiled search: 0 = never, 1 = when the cloud fills the chip with full tiles, 2 = always", c->tiled),
  OPT(CILHIP_OPT_WARM_START, "warm_start", 1, 0, 2, "warm-started iterations (margin proof): 0 = never, 1 = when the loop is near alignment, 2 = from the second iteration on", c->warm_start),
  OPT(CILHIP_OPT_WARM_FORECAST, "warm_forecast", 1, 0, 1, "the cold kernels' forecast gates the warm-started form (0: the step alone; tests)", c->warm_forecast),
  OPT(CILHIP_OPT_FUSED_EPILOGUE, "fused_epilogue", 0, 0, 1, "stage-1 reduction + epilogue as one launch behind a device-scope fence (bitwise equal, measured slower; A/B)", c->fused_epilogue),
  OPT(CILHIP_OPT_GROUP_SEARCH, "group_search", -1, -1, 64, "lanes per query of the cooperative global-memory search: -1 = the loop decides, 0 = never, 4 / 8 / 16 / 32 / 64", c->group_lanes),
  OPT(CILHIP_OPT_TIE_RULE, "tie_rule", 2, 0, 2, "exactly equidistant nearest points: 0 = lowest index, 1 = the reference's kd-tree order (tables up front), 2 = the same, tables when a tie is first met", c->tie_rule),
  OPT(CILHIP_OPT_WARM_EXTRA_FRACTION, "warm_extra_fraction", 0.0625, 1e-9, 1, "room (fraction of a cell) of the ball a warm-started iteration searches a listed query in", c->warm_extra),
  OPT(CILHIP_OPT_PAIR_RECORDS, "pair_records", 1, 0, 1, "streaming accumulation gathers a match's point and normal from one 32-byte record (A/B)", c->pair_records),
  OPT(CILHIP_OPT_TILE_RECORDS, "tile_records", 1, 0, 1, "the accumulating tile kernel writes the warm-started form's match records itself (A/B)", c->tile_records),
  OPT(CILHIP_OPT_WARM_ENTER_FRACTION, "warm_enter_fraction", 0.15, 1e-9, 1e9, "the warm-started form is entered once an update moves no source point by more than this fraction of a cell", c->warm_enter),
  OPT(CILHIP_OPT_POINT_WEIGHT_EVALUATOR, "point_weight_evaluator", 0, 0, 2, "combined metric, point-to-point terms: 0 = UnityWeightEvaluator, 1 = DistanceEvaluator, 2 = RBFKernelWeightEvaluator", c->cw_point_kind),
  OPT(CILHIP_OPT_PLANE_WEIGHT_EVALUATOR, "plane_weight_evaluator", 0, 0, 2, "combined metric, point-to-plane terms: 0 = Unity, 1 = Identity (distance), 2 = RBF kernel", c->cw_plane_kind),
  OPT(CILHIP_OPT_POINT_WEIGHT_SIGMA, "point_weight_sigma", 1, 1e-30, 1e30, "sigma of the RBF evaluator of the point-to-point terms", c->cw_point_sigma),
  OPT(CILHIP_OPT_PLANE_WEIGHT_SIGMA, "plane_weight_sigma", 1, 1e-30, 1e30, "sigma of the RBF evaluator of the point-to-plane terms", c->cw_plane_sigma),
  OPT(CILHIP_OPT_TILE_ACCUMULATION, "tile_accumulation", 1, 0, 2, "first Gauss-Newton step accumulated inside the LDS tiles: 0 = never, 1 = unless the source is far from alignment, 2 = always", (c->tile_acc ? (c->tile_acc_adaptive ? 1 : 2) : 0)),
  OPT(CILHIP_OPT_SEARCH_DIRECTION, "search_direction", 0, 0, 2, "CorrespondenceSearchDirection: 0 = SECOND_TO_FIRST, 1 = FIRST_TO_SECOND, 2 = BOTH", c->search_dir),
  OPT(CILHIP_OPT_FEATURE_NORMAL_WEIGHT, "feature_normal_weight", 0, 0, 1e30, "PointNormalFeaturesAdaptor's normal weight (> 0: the search runs on 6-D features)", c->normal_weight),
  OPT(CILHIP_OPT_FEATURE_KIND, "feature_kind", 0, 0, 2, "second feature block: 0 = normals (follow the transform), 1 = colours (do not), 2 = normals + colours (9-D)", c->feature_kind),
  OPT(CILHIP_OPT_FEATURE_COLOR_WEIGHT, "feature_color_weight", 0, 0, 1e30, "PointNormalColorFeaturesAdaptor's colour weight (feature_kind 2)", c->color_weight),
  OPT(CILHIP_OPT_SYMMETRIC_METRIC, "symmetric_metric", 1, 0, 1, "source normals, when set, switch the combined metric to the symmetric objective", c->symmetric),
  OPT(CILHIP_OPT_TRANSFORM_MODE, "transform_mode", 0, 0, 1, "ICP instance family: 0 = rigid, 1 = affine", c->transform_mode),
  OPT(CILHIP_OPT_REQUIRE_RECIPROCALITY, "require_reciprocality", 0, 0, 1, "setRequireReciprocality (search_direction BOTH)", c->reciprocal),
  OPT(CILHIP_OPT_CELL_OCCUPANCY, "cell_occupancy", 1, 1e-3, 1e6, "target points per grid cell the next cilhip_set_target aims at", c->cell_occupancy),
  OPT(CILHIP_OPT_REFINED_OCCUPANCY_FACTOR, "refined_occupancy_factor", 3, 1, 64, "how much denser than that a grid that had to be refined (surface, clusters) may stay", c->refined_occupancy),
  OPT(CILHIP_OPT_KERNEL_TIMING, "kernel_timing", 0, 0, 1, "hipEvents around the search / accumulation kernels (cilhip_enable_kernel_timing)", c->kernel_timing),
  OPT(CILHIP_OPT_KERNEL_TIMING_STRIDE, "kernel_timing_stride", 1, 1, 4096, "with kernel timing on: iterations 0..2 and every stride-th one carry events", c->timing_stride),
  OPT(CILHIP_OPT_REVERSE_WARM_START, "reverse_warm_start", 1, 0, 1, "device-resident FIRST_TO_SECOND / BOTH loops: reverse searches after the first start from the previous reverse matches (margin test over the source; A/B)", c->reverse_warm),
  OPT(CILHIP_OPT_FEATURE_WARM_START, "feature_warm_start", 1, 0, 1, "feature adaptors (SECOND_TO_FIRST loops): searches warm-started from the previous matches once the loop moves little (margin test with the feature distance; A/B)", c->feat_warm),
  OPT(CILHIP_OPT_AFFINE_DEVICE_LOOP, "affine_device_loop", 1, 0, 1, "affine classes: 1 = device-resident loop (one-pass moments, solve in the epilogue kernel), 0 = host-driven loop (three passes + host solve; A/B)", c->affine_device_loop),
};
#undef OPT
constexpr int N_OPTIONS = (int)(sizeof(g_options) / sizeof(g_options[0]));
static_assert(N_OPTIONS == CILHIP_OPT_COUNT, "one table row per enum cilhip_option value, in the enum's order");
}  // namespace

int cilhip_option_count(void) { return N_OPTIONS; }
const cilhip_option_info_t* cilhip_option_info(int id) { return (id >= 0 && id < N_OPTIONS) ? &g_options[id].info : nullptr; }
int cilhip_set_option_id(cilhip_ctx* c, cilhip_option id, double value) {
  if (!c) return CILHIP_ERR_INVALID;
  if ((int)id < 0 || (int)id >= N_OPTIONS) return fail(c, CILHIP_ERR_INVALID, "set_option_id: unknown option");
  return cilhip_set_option(c, g_options[(int)id].info.key, value);
}
int cilhip_get_option(cilhip_ctx* c, const char* key, double* value) {
  if (!c || !key || !value) return CILHIP_ERR_INVALID;
  for (int i = 0; i < N_OPTIONS; ++i)
    if (!strcmp(key, g_options[i].info.key)) { *value = g_options[i].get(c); return CILHIP_OK; }
  return fail(c, CILHIP_ERR_INVALID, "get_option: unknown key");
}

int cilhip_set_option(cilhip_ctx* c, const char* key, double value) {
  if (!c || !key) return CILHIP_ERR_INVALID;
  if (value != value) return fail(c, CILHIP_ERR_INVALID, "set_option: the value is not a number");
  if (!strcmp(key, "fused")) { c->fused = value != 0.0; return CILHIP_OK; }
  // (a finished run's set that has not been searched again yet -- cilhip_get_last_matches_origin 2 -- would be filtered with the
  //  NEW values: a changed post-filter drops it; a set already in memory is what its search left, whatever is set afterwards)
  if (!strcmp(key, "inlier_fraction")) { if (c->inlier_fraction != value && c->pending_matches) drop_matches(c); c->inlier_fraction = value; return CILHIP_OK; }
  if (!strcmp(key, "one_to_one")) { if (c->one_to_one != (value != 0.0) && c->pending_matches) drop_matches(c); c->one_to_one = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "tiled")) { c->tiled = (int)value; return CILHIP_OK; }
  if (!strcmp(key, "warm_start")) { c->warm_start = (int)value; return CILHIP_OK; }
  if (!strcmp(key, "warm_forecast")) { c->warm_forecast = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "fused_epilogue")) { c->fused_epilogue = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "reverse_warm_start")) { c->reverse_warm = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "feature_warm_start")) { c->feat_warm = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "affine_device_loop")) { c->affine_device_loop = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "group_search")) {
    if (value != -1.0 && value != 0.0 && value != 4.0 && value != 8.0 && value != 16.0 && value != 32.0 && value != 64.0)
      return fail(c, CILHIP_ERR_INVALID, "group_search: -1 (the loop decides), 0 (never), or 4, 8, 16, 32, 64 lanes per query");
    c->group_lanes = (int)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "tie_rule")) {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return fail(c, CILHIP_ERR_INVALID, "tie_rule: 0 (lowest index), 1 (the reference's kd-tree order, tables built up front) or 2 (the same, tables built when a tie is first met)");
    if (((int)value != 0) != (c->tie_rule != 0)) drop_matches(c);
    c->tie_rule = (int)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "warm_extra_fraction")) {
    if (!(value > 0.0 && value <= 1.0)) return fail(c, CILHIP_ERR_INVALID, "warm_extra_fraction: in (0, 1]");
    c->warm_extra = (float)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "pair_records")) { c->pair_records = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "tile_records")) { c->tile_records = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "warm_enter_fraction")) {
    if (!(value > 0.0)) return fail(c, CILHIP_ERR_INVALID, "warm_enter_fraction: > 0 (fraction of a grid cell)");
    c->warm_enter = (float)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "point_weight_evaluator") || !strcmp(key, "plane_weight_evaluator")) {
    if (value != 0.0 && value != 1.0 && value != 2.0) return fail(c, CILHIP_ERR_INVALID, "weight evaluator: 0 = Unity, 1 = Identity, 2 = RBF kernel");
    (key[1] == 'o' ? c->cw_point_kind : c->cw_plane_kind) = (int)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "point_weight_sigma") || !strcmp(key, "plane_weight_sigma")) {
    if (!(value > 0.0)) return fail(c, CILHIP_ERR_INVALID, "weight evaluator sigma must be positive");
    (key[1] == 'o' ? c->cw_point_sigma : c->cw_plane_sigma) = (float)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "tile_accumulation")) { c->tile_acc = value != 0.0; c->tile_acc_adaptive = value != 2.0; return CILHIP_OK; }
  if (!strcmp(key, "search_direction")) {
    if (value != 0.0 && value != 1.0 && value != 2.0) return fail(c, CILHIP_ERR_INVALID, "search_direction: 0 = SECOND_TO_FIRST, 1 = FIRST_TO_SECOND, 2 = BOTH");
    c->search_dir = (int)value; drop_matches(c); c->have_pairs = false;
    return CILHIP_OK;
  }
  if (!strcmp(key, "feature_normal_weight")) {
    if (!(value >= 0.0)) return fail(c, CILHIP_ERR_INVALID, "feature_normal_weight: >= 0 (0 = plain point features)");
    if (c->normal_weight != (float)value) drop_feat_tie_tables(c);
    c->normal_weight = (float)value; drop_matches(c); c->have_pairs = false;
    return CILHIP_OK;
  }
  if (!strcmp(key, "feature_kind")) {
    if (value != 0.0 && value != 1.0 && value != 2.0)
      return fail(c, CILHIP_ERR_INVALID, "feature_kind: 0 = normals (follow the transform), 1 = colours (do not), 2 = normals + colours (9-D)");
    if ((int)value != c->feature_kind) { drop_src_grid(c); drop_feat_tie_tables(c); }      // (the source's grid carries the feature vectors of the reverse searches)
    c->feature_kind = (int)value; drop_matches(c); c->have_pairs = false;
    return CILHIP_OK;
  }
  if (!strcmp(key, "feature_color_weight")) {
    if (!(value >= 0.0)) return fail(c, CILHIP_ERR_INVALID, "feature_color_weight: >= 0");
    if (c->color_weight != (float)value) drop_feat_tie_tables(c);
    c->color_weight = (float)value; drop_matches(c); c->have_pairs = false;
    return CILHIP_OK;
  }
  if (!strcmp(key, "symmetric_metric")) { c->symmetric = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "transform_mode")) {
    if (value != 0.0 && value != 1.0) return fail(c, CILHIP_ERR_INVALID, "transform_mode: 0 = rigid, 1 = affine");
    c->transform_mode = (int)value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "require_reciprocality")) { c->reciprocal = value != 0.0; drop_matches(c); c->have_pairs = false; return CILHIP_OK; }
  if (!strcmp(key, "cell_occupancy")) { c->cell_occupancy = value; return CILHIP_OK; }
  if (!strcmp(key, "refined_occupancy_factor")) {
    if (!(value >= 1.0 && value <= 64.0)) return fail(c, CILHIP_ERR_INVALID, "refined_occupancy_factor: in [1, 64]");
    c->refined_occupancy = value;
    return CILHIP_OK;
  }
  if (!strcmp(key, "kernel_timing")) { c->kernel_timing = value != 0.0; return CILHIP_OK; }
  if (!strcmp(key, "kernel_timing_stride")) {
    if (!(value >= 1.0 && value <= 4096.0)) return fail(c, CILHIP_ERR_INVALID, "kernel_timing_stride: 1 .. 4096");
    c->timing_stride = (int)value;
    return CILHIP_OK;
  }
  return fail(c, CILHIP_ERR_INVALID, "set_option: unknown key");
}

int cilhip_debug_counters(cilhip_ctx* c, uint32_t out[2]) {
  if (!c || !out) return CILHIP_ERR_INVALID;
  out[0] = out[1] = 0;
  if (!c->d_defer_mask || !c->ntiles) return CILHIP_OK;
  CK(c, hipSetDevice(c->device));
  if (!c->d_dbg) CK(c, hipMalloc(&c->d_dbg, 2 * sizeof(uint32_t)));
  launch_count_deferred(c->d_defer_mask, c->ntiles, c->d_dbg, c->stream);
  CK(c, hipMemcpyAsync(out, c->d_dbg, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
#ifdef CILHIP_EXP_PHASE_CLOCKS
  cilhip::debug_dump_phase_clocks();
#endif
  return CILHIP_OK;
}

int cilhip_get_last_timing2(cilhip_ctx* c, double* search_ms, double* accumulate_ms) {
  if (!c) return CILHIP_ERR_INVALID;
  if (search_ms) *search_ms = c->last_search_ms;
  if (accumulate_ms) *accumulate_ms = c->last_acc_ms;
  return CILHIP_OK;
}

int cilhip_get_last_iteration_timing(cilhip_ctx* c, int cap, int* n, unsigned int* iteration, float* kernel_ms) {
  if (!c || !n || cap < 0) return CILHIP_ERR_INVALID;
  const size_t m = c->timed_ms.size() < c->timed_iter.size() ? c->timed_ms.size() : c->timed_iter.size();
  *n = (int)m;
  for (size_t k = 0; k < m && k < (size_t)cap; ++k) { if (iteration) iteration[k] = c->timed_iter[k]; if (kernel_ms) kernel_ms[k] = c->timed_ms[k]; }
  return CILHIP_OK;
}

int cilhip_get_last_form_timing(cilhip_ctx* c, int form, double* kernel_ms, int* launches) {
  if (!c || form < 0 || form > 4) return CILHIP_ERR_INVALID;
  if (kernel_ms) *kernel_ms = c->form_ms[form];
  if (launches) *launches = c->form_n[form];
  return CILHIP_OK;
}

int cilhip_get_last_run_trace(cilhip_ctx* c, int cap, int* n_out, unsigned int* unproven, unsigned int* listed, float* step, float* delta, int* form) {
  if (!c || !n_out || cap < 0) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  IcpState hs;
  CK(c, hipMemcpyAsync(&hs, c->d_state, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  uint4 tr[RUN_TRACE_CAP];
  CK(c, hipMemcpyAsync(tr, c->d_trace, sizeof(tr), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  const int n = std::min(std::min(hs.iterations, (int)RUN_TRACE_CAP), cap);
  for (int i = 0; i < n; ++i) {
    if (unproven) unproven[i] = tr[i].x;
    if (listed) listed[i] = tr[i].y;
    if (step) memcpy(&step[i], &tr[i].z, 4);
    if (delta) memcpy(&delta[i], &tr[i].w, 4);
    if (form) form[i] = (size_t)i < c->trace_form.size() ? (int)(c->trace_form[i] & 0x7f) : -1;
  }
  *n_out = n;
  return CILHIP_OK;
}

int cilhip_get_last_warm_iterations(cilhip_ctx* c, int* warm_iterations) {
  if (!c || !warm_iterations) return CILHIP_ERR_INVALID;
  *warm_iterations = c->last_warm_iters;
  return CILHIP_OK;
}

int cilhip_get_last_run_forms(cilhip_ctx* c, int* one_pass_iterations, int* two_pass_iterations) {
  if (!c) return CILHIP_ERR_INVALID;
  if (one_pass_iterations) *one_pass_iterations = c->last_fused_iters;
  if (two_pass_iterations) *two_pass_iterations = c->last_two_pass_iters;
  return CILHIP_OK;
}

int cilhip_enable_kernel_timing(cilhip_ctx* c, int on) {
  if (!c) return CILHIP_ERR_INVALID;
  c->kernel_timing = on != 0;
  return CILHIP_OK;
}

static int upload(cilhip_ctx* c, const float* src, size_t count, int mem, float** d_out) {
  *d_out = nullptr;
  CK(c, hipMalloc(d_out, (count ? count : 1) * sizeof(float)));
  if (count)
    CK(c, hipMemcpyAsync(*d_out, src, count * sizeof(float), mem == CILHIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  return CILHIP_OK;
}

int cilhip_set_target(cilhip_ctx* c, const float* xyz, const float* nrm, size_t n, int mem) {
  if (!c) return CILHIP_ERR_INVALID;
  if ((n && !xyz) || n >= 0xFFFFFFF0ull) return fail(c, CILHIP_ERR_INVALID, "set_target: bad cloud (null or >= 2^32-16 points)");
  CK(c, hipSetDevice(c->device));
  auto t0 = std::chrono::steady_clock::now();
  release_target(c);      // (incl. the order tables, the nearest-other-point table: they describe ONE target; a shared target is only let go of)
  if (c->d_winner) { (void)hipFree(c->d_winner); c->d_winner = nullptr; }
  if (c->d_rev_pos) { (void)hipFree(c->d_rev_pos); c->d_rev_pos = nullptr; }
  if (c->d_rev_d2) { (void)hipFree(c->d_rev_d2); c->d_rev_d2 = nullptr; }
  float *d_xyz = nullptr, *d_nrm = nullptr;
  int rc = upload(c, xyz, 3 * n, mem, &d_xyz);
  if (rc) return rc;
  if (nrm) { rc = upload(c, nrm, 3 * n, mem, &d_nrm); if (rc) { (void)hipFree(d_xyz); return rc; } }
  GridBuildResult r{};
  double mean[3];
  hipError_t e = build_grid(d_xyz, d_nrm, (uint32_t)n, c->stream, &r, mean, c->cell_occupancy, c->refined_occupancy);
  (void)hipFree(d_xyz);
  if (d_nrm) (void)hipFree(d_nrm);
  if (e != hipSuccess) { c->err = std::string("build_grid: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
  c->grid = r.grid; c->grid_occ = r.avg_occupancy; c->grid_cells = r.n_cells;
  c->warm_banned = false;
  c->dst_rgb_sorted_ok = false;
  if (c->d_dst_rgb) { (void)hipFree(c->d_dst_rgb); c->d_dst_rgb = nullptr; }      // (colours belong to the target they were set for)
  if (c->d_dst_rgb_sorted) { (void)hipFree(c->d_dst_rgb_sorted); c->d_dst_rgb_sorted = nullptr; }
  c->has_normals = (nrm != nullptr);
  for (int i = 0; i < 3; ++i) c->dst_mean[i] = (float)mean[i];
  c->partial_target = false;      // (a new target stands for itself until cilhip_set_shard_info says otherwise)
  c->has_target = true;
  c->src_sorted = false;  // source order is tied to the target grid
  drop_matches(c);
  c->have_pairs = false; c->pairs.count = 0;
  c->far_mode = true;
  c->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return CILHIP_OK;
}

// CorrespondenceSearchKDTree::getFirstSearchTree / setFirstSearchTree (correspondence_search_kd_tree.hpp:273-296): a second engine
// takes the index another one built instead of building its own.  Here: `c` takes `from`'s target as it stands -- the sorted
// points and normals, the cell table, and whatever has been built on top of them by now (the paired point+normal records, the
// nearest-other-point table, the order tables of the reference's tree, the index -> position map) -- without copying a byte.  The
// allocations move into a reference-counted share: either context may be destroyed or given another target first, the memory goes
// when the last user lets go.  What one of them builds LATER (tables a run finds it needs) is its own.
int cilhip_share_target(cilhip_ctx* c, cilhip_ctx* from) {
  if (!c || !from || c == from) return CILHIP_ERR_INVALID;
  if (!from->has_target) return fail(c, CILHIP_ERR_INVALID, "share_target: the other context has no target");
  if (c->device != from->device) return fail(c, CILHIP_ERR_INVALID, "share_target: the two contexts live on different devices");
  CK(c, hipSetDevice(c->device));
  CK(c, hipStreamSynchronize(from->stream));      // (whatever is still building the lender's tables)
  CK(c, hipStreamSynchronize(c->stream));
  release_target(c);
  if (c->d_winner) { (void)hipFree(c->d_winner); c->d_winner = nullptr; }
  if (c->d_rev_pos) { (void)hipFree(c->d_rev_pos); c->d_rev_pos = nullptr; }
  if (c->d_rev_d2) { (void)hipFree(c->d_rev_d2); c->d_rev_d2 = nullptr; }
  if (c->d_dst_rgb) { (void)hipFree(c->d_dst_rgb); c->d_dst_rgb = nullptr; }
  if (c->d_dst_rgb_sorted) { (void)hipFree(c->d_dst_rgb_sorted); c->d_dst_rgb_sorted = nullptr; }
  c->dst_rgb_sorted_ok = false;
  // the lender's own allocations become the share's (its earlier share, if it has one, already holds the rest)
  if (!from->tshare) { from->tshare = new (std::nothrow) TargetShare(); if (!from->tshare) return fail(c, CILHIP_ERR_HIP, "share_target: out of memory"); from->tshare->refs = 1; }
  TargetShare* sh = from->tshare;
  const void* ptrs[] = {from->grid.pts, from->grid.nrm, from->grid.pn, from->grid.cell_start, from->d_inv_perm, from->d_safe2, from->d_tie_leaf_slot, from->d_tie_nodes};
  for (const void* p : ptrs)
    if (p && std::find(sh->allocs.begin(), sh->allocs.end(), p) == sh->allocs.end()) sh->allocs.push_back(const_cast<void*>(p));
  ++sh->refs;
  c->tshare = sh;
  c->grid = from->grid; c->has_target = true; c->has_normals = from->has_normals;
  c->grid_occ = from->grid_occ; c->grid_cells = from->grid_cells; c->build_ms = 0.0;
  for (int i = 0; i < 3; ++i) c->dst_mean[i] = from->dst_mean[i];
  c->index_offset = from->index_offset; c->partial_target = from->partial_target;
  c->d_inv_perm = from->d_inv_perm; c->d_safe2 = from->d_safe2;
  c->d_tie_leaf_slot = from->d_tie_leaf_slot; c->d_tie_nodes = from->d_tie_nodes;
  c->warm_banned = false;
  c->src_sorted = false;  // source order is tied to the target grid
  drop_matches(c);
  c->have_pairs = false; c->pairs.count = 0;
  c->far_mode = true;
  return CILHIP_OK;
}

int cilhip_set_source(cilhip_ctx* c, const float* xyz, size_t n, int mem) {
  if (!c) return CILHIP_ERR_INVALID;
  if ((n && !xyz) || n >= 0xFFFFFFF0ull) return fail(c, CILHIP_ERR_INVALID, "set_source: bad cloud");
  CK(c, hipSetDevice(c->device));
  free_source(c);
  int rc = upload(c, xyz, 3 * n, mem, &c->d_src_xyz);
  if (rc) return rc;
  const size_t cap = n ? n : 1;
  CK(c, hipMalloc(&c->d_src_sorted, cap * sizeof(float4)));
  CK(c, hipMalloc(&c->d_nn_pos, cap * sizeof(uint32_t)));
  CK(c, hipMalloc(&c->d_nn_d2, cap * sizeof(float)));
  c->ns = (uint32_t)n;
  double mean[3];
  float lo[3], hi[3];
  hipError_t e = mean3_device(c->d_src_xyz, c->ns, c->stream, mean, lo, hi);
  if (e != hipSuccess) { c->err = std::string("mean3: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
  for (int i = 0; i < 3; ++i) {
    c->src_mean[i] = (float)mean[i];
    // (centre and half extent rounded so that the box holds every point: the half extent is taken from the rounded centre)
    c->src_center[i] = 0.5f * (lo[i] + hi[i]);
    c->src_half[i] = c->ns ? std::max(hi[i] - c->src_center[i], c->src_center[i] - lo[i]) * 1.000001f : 0.0f;
    if (!(c->src_half[i] >= 0.0f) || !std::isfinite(c->src_center[i])) { c->src_center[i] = 0.0f; c->src_half[i] = 1.0e30f; }   // (non-finite coordinates: no bound)
  }
  const int nb = std::max(iter_num_blocks(c->ns), warm_num_blocks(c->ns));      // rows of partial sums: the streaming and the warm-started kernels
  if (nb > c->partial_blocks) {
    if (c->d_partials) (void)hipFree(c->d_partials);
    CK(c, hipMalloc(&c->d_partials, (size_t)nb * SUMS_MAX * sizeof(double)));
    c->partial_blocks = nb;
  }
  c->has_source = true;
  return CILHIP_OK;
}

int cilhip_set_source_normals(cilhip_ctx* c, const float* nrm, int mem) {
  if (!c) return CILHIP_ERR_INVALID;
  if (!c->has_source) return fail(c, CILHIP_ERR_INVALID, "set_source_normals: set_source first");
  CK(c, hipSetDevice(c->device));
  if (c->d_src_nrm) { (void)hipFree(c->d_src_nrm); c->d_src_nrm = nullptr; }
  if (c->d_src_nrm_sorted) { (void)hipFree(c->d_src_nrm_sorted); c->d_src_nrm_sorted = nullptr; }
  c->have_pairs = false; c->pairs.count = 0;
  drop_src_grid(c);      // (it carries the feature vectors of the reverse searches)
  if (!nrm) return CILHIP_OK;                              // back to the 3-cloud (non-symmetric) form
  int rc = upload(c, nrm, 3 * (size_t)c->ns, mem, &c->d_src_nrm);
  if (rc) return rc;
  c->src_nrm0[0] = c->src_nrm0[1] = c->src_nrm0[2] = 0.0f;
  if (c->ns) {
    CK(c, hipMemcpyAsync(c->src_nrm0, c->d_src_nrm, 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
  }
  CK(c, hipMalloc(&c->d_src_nrm_sorted, (c->ns ? c->ns : 1) * sizeof(float4)));
  c->src_sorted = false;                                   // the sorted copy is (re)built with the next sort
  drop_matches(c);
  c->have_pairs = false; c->pairs.count = 0;
  return CILHIP_OK;
}

int cilhip_set_color_features(cilhip_ctx* c, const float* dst_rgb, const float* src_rgb, int mem) {
  if (!c) return CILHIP_ERR_INVALID;
  if (!c->has_target || !c->has_source) return fail(c, CILHIP_ERR_INVALID, "set_color_features: set_target and set_source first");
  if (!dst_rgb || !src_rgb) return fail(c, CILHIP_ERR_INVALID, "set_color_features: both clouds' colours are needed");
  CK(c, hipSetDevice(c->device));
  if (c->d_dst_rgb) { (void)hipFree(c->d_dst_rgb); c->d_dst_rgb = nullptr; }
  if (c->d_src_rgb) { (void)hipFree(c->d_src_rgb); c->d_src_rgb = nullptr; }
  if (c->d_src_rgb_sorted) { (void)hipFree(c->d_src_rgb_sorted); c->d_src_rgb_sorted = nullptr; }
  int rc = upload(c, dst_rgb, 3 * (size_t)c->grid.n, mem, &c->d_dst_rgb);
  if (rc) return rc;
  rc = upload(c, src_rgb, 3 * (size_t)c->ns, mem, &c->d_src_rgb);
  if (rc) return rc;
  CK(c, hipMalloc(&c->d_src_rgb_sorted, (c->ns ? c->ns : 1) * sizeof(float4)));
  c->dst_rgb_sorted_ok = false;
  c->src_sorted = false;                                   // the source's sorted copy is (re)built with the next sort
  drop_src_grid(c);      // (it carries the features of the reverse searches)
  drop_feat_tie_tables(c);      // (the colours are coordinates of the feature tree)
  drop_matches(c);
  c->have_pairs = false; c->pairs.count = 0;
  return CILHIP_OK;
}

int cilhip_get_means(cilhip_ctx* c, float dm[3], float sm[3]) {
  if (!c) return CILHIP_ERR_INVALID;
  if (dm) memcpy(dm, c->dst_mean, sizeof(c->dst_mean));
  if (sm) memcpy(sm, c->src_mean, sizeof(c->src_mean));
  return CILHIP_OK;
}

// Spatially sort the source under T (once; re-sorted only if the transform moved it by more than a few cells).
static int ensure_sorted(cilhip_ctx* c, const float T[16]) {
  if (!c->has_target || !c->has_source) return fail(c, CILHIP_ERR_INVALID, "set_target and set_source first");
  bool need = !c->src_sorted;
  if (!need) {
    // displacement of the source bbox centre proxy: compare transforms on the source mean
    float a[3], b[3];
    transform_point(T, c->src_mean[0], c->src_mean[1], c->src_mean[2], a[0], a[1], a[2]);
    transform_point(c->sort_T, c->src_mean[0], c->src_mean[1], c->src_mean[2], b[0], b[1], b[2]);
    float dl = 0.f;
    for (int i = 0; i < 3; ++i) dl = fmaxf(dl, fabsf(a[i] - b[i]));
    float dr = 0.f;
    for (int i = 0; i < 11; ++i) if (i % 4 != 3) dr = fmaxf(dr, fabsf(T[i] - c->sort_T[i]));
    const float ext = fmaxf(c->grid.nx, fmaxf(c->grid.ny, c->grid.nz)) * c->grid.cell;
    if (dl > 4.0f * c->grid.cell || dr * ext > 4.0f * c->grid.cell) need = true;
  }
  if (need) {
    // (scratch, tile table and the per-tile arrays are kept between the sorts of a source: a re-sort costs its kernels only)
    c->d_tiles = nullptr; c->d_tile_center = nullptr; c->ntiles = 0;
    hipError_t e = sort_source(c->d_src_xyz, c->ns, c->grid, T, c->d_src_sorted, c->stream, &c->d_tiles, &c->d_tile_center, c->tile_axes, &c->ntiles, &c->sort_ws);
    if (e != hipSuccess) { c->err = std::string("sort_source: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
    if (c->ntiles + 1 > c->tile_aux_cap) {
      if (c->d_tile_box) { (void)hipFree(c->d_tile_box); c->d_tile_box = nullptr; }
      if (c->d_defer_mask) { (void)hipFree(c->d_defer_mask); c->d_defer_mask = nullptr; }
      c->tile_aux_cap = 0;
      const uint32_t cap = c->ntiles + 1 + c->ntiles / 8;
      CK(c, hipMalloc(&c->d_defer_mask, (size_t)cap * 2 * (TILE_THREADS / 64) * sizeof(unsigned long long)));
      CK(c, hipMalloc(&c->d_tile_box, (size_t)cap * 8 * sizeof(int)));
      c->tile_aux_cap = cap;
    }
    CK(c, hipMemsetAsync(c->d_defer_mask, 0, ((size_t)c->ntiles + 1) * 2 * (TILE_THREADS / 64) * sizeof(unsigned long long), c->stream));
    {   // the tiled search with in-tile accumulation leaves one row of partial sums per tile and per block of its clean-up pass
      const int rows = std::max(std::max(iter_num_blocks(c->ns), warm_num_blocks(c->ns)), tiled_partial_rows(c->ntiles));
      if (rows > c->partial_blocks) {
        if (c->d_partials) (void)hipFree(c->d_partials);
        c->d_partials = nullptr; c->partial_blocks = 0;
        CK(c, hipMalloc(&c->d_partials, (size_t)rows * SUMS_MAX * sizeof(double)));
        c->partial_blocks = rows;
      }
    }
    if (c->d_src_nrm) launch_gather_by_w(c->d_src_sorted, c->d_src_nrm, c->ns, c->d_src_nrm_sorted, c->stream);
    if (c->d_src_rgb) launch_gather_by_w(c->d_src_sorted, c->d_src_rgb, c->ns, c->d_src_rgb_sorted, c->stream);
    if (c->d_src_inv) { (void)hipFree(c->d_src_inv); c->d_src_inv = nullptr; }
    if (c->d_grid_to_sorted) { (void)hipFree(c->d_grid_to_sorted); c->d_grid_to_sorted = nullptr; }
    memcpy(c->sort_T, T, sizeof(c->sort_T));
    c->src_sorted = true;
    c->src3_valid = false; c->rec_valid = false; c->lb_fresh = false;     // (per sorted order)
    drop_matches(c);
  }
  return CILHIP_OK;
}

int cilhip_prepare_source(cilhip_ctx* c, const float* T, int force, double* ms) {
  if (!c) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  CK(c, hipStreamSynchronize(c->stream));
  const auto t0 = std::chrono::steady_clock::now();
  if (force) c->src_sorted = false;
  const int rc = ensure_sorted(c, T ? T : kIdentity);
  if (rc) return rc;
  CK(c, hipStreamSynchronize(c->stream));
  if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return CILHIP_OK;
}

// The LDS-tiled kernel runs 1024-thread workgroups, two per CU: below ~2 full rounds of tiles on the
// 256 CUs the per-lane kernel (8x more, smaller workgroups) balances better (measured: per-lane wins at 1M
// points = 580 tiles, tiled wins from 2M = 1160 tiles on).
// It also needs tiles that are reasonably full (a source much sparser than the target leaves most lanes of
// a tile idle: 10M source points against an 80M-point target fill 14 % of the slots) and a target whose
// local density fits the LDS budget of a tile's region (cube + halo + one cell of drift per axis);
// otherwise every tile would be handed to the clean-up pass, which is the per-lane search done worse.
static bool use_tiled(const cilhip_ctx* c) {
  if (c->ns >= 0x80000000ull) return false;   // the clean-up list keeps a flag in bit 31 of a query index
  if (c->tiled >= 2) return true;
  if (c->tiled != 1 || c->ntiles < 600) return false;   // (measured: 729 tiles / 1M points already favour the tiles by 4 %, 2M by 27 %)
  const double fill = (double)c->ns / ((double)c->ntiles * (double)TILE_QUERIES);
  const double region_cells = (double)(CUBE_EDGE + 3) * (CUBE_EDGE + 3) * (CUBE_EDGE + 3);
  const double density = c->grid_occ > 1.0 ? c->grid_occ - 1.0 : c->grid_occ;   // sum(count^2)/n = lambda + 1 for a Poisson cloud
  return fill >= 0.45 && density * region_cells <= 0.92 * (double)TILE_CAP;
}

static bool filters_active(const cilhip_ctx* c) {
  return (c->inlier_fraction > 0.0 && c->inlier_fraction < 1.0) || c->one_to_one;
}

// The ICP loop's first Gauss-Newton step is accumulated inside the LDS tiles of the search (one pass instead of a search
// pass + a streaming accumulation pass) whenever the plain engine runs tiled: no post-filters (they act on the complete
// match set), point features, the three-cloud metric (the symmetric objective reads source normals per pair), and not
// the A/B option "fused" (per-lane kernel) or "tile_accumulation" = 0.
// kernel forms of an iteration's search (+ accumulation): cilhip_get_last_form_timing
enum { FORM_SEARCH = 0, FORM_TILE_ONE_PASS = 1, FORM_WARM_FIRST = 2, FORM_WARM = 3, FORM_LANE_FUSED = 4 };
// The warm-started form (k_warm) pays while the queries move little between iterations: a query is settled without any search as
// long as it has moved less than the MARGIN its last search left it (distance to the second nearest target point minus distance
// to the nearest, capped by the searched block's faces -- a good fraction of the target's point spacing, whatever the source is).
// The epilogue publishes how far any source point can have moved in the last update (IcpState::motion_step); a run enters the
// form when that falls below warm_thresh (a fraction of a cell), and the kernel's own count of the queries it had to search
// corrects the guess: a quarter of them searched = one iteration through the cold form (whose searches leave fresh margins)
// and half the bar; three such falls and the run stays cold.
static void warm_run_reset(cilhip_ctx* c) { c->warm_thresh = c->warm_enter * c->grid.cell; c->warm_strikes = 0; }
// a warm iteration was seen to search `listed` of its queries: keep going?
static bool warm_keeps_paying(cilhip_ctx* c, unsigned int listed) {
  if ((unsigned long long)listed * 4ull <= (unsigned long long)c->ns) return true;
  c->warm_thresh *= 0.5f;
  if (++c->warm_strikes >= 3) c->warm_banned = true;
  return false;
}
static bool warm_worthwhile(const cilhip_ctx* c, float step) { return step < c->warm_thresh; }
// the matches records / margin keys / 12-byte source copy of the warm-started iterations: allocated by the first run that can use them
static int ensure_warm_buffers(cilhip_ctx* c) {
  const size_t cap = c->ns ? c->ns : 1;
  if (!c->d_warm_rec) { CK(c, hipMalloc(&c->d_warm_rec, cap * (sizeof(float4) + 2 * sizeof(F3)))); c->src3_valid = false; }
  if (!c->d_nn_lb) CK(c, hipMalloc(&c->d_nn_lb, cap * sizeof(float)));
  if (!c->src3_valid) {
    launch_copy_src3(c->d_src_sorted, c->ns, reinterpret_cast<F3*>(c->d_warm_rec + cap) + cap, c->stream);
    c->src3_valid = true;
  }
  return CILHIP_OK;
}
// {point, normal} of every target position side by side (GridDev::pn), for the streaming accumulation's gathers: built by the first run
// that streams over stored matches with a metric that reads normals (32 B per target point; without room for it the two arrays serve)
static void ensure_pair_records(cilhip_ctx* c) {
  if (c->grid.pn || !c->pair_records || !c->grid.nrm || !c->grid.n) return;
  float4* pn = nullptr;
  if (hipMalloc(&pn, (size_t)c->grid.n * 2 * sizeof(float4)) != hipSuccess) { (void)hipGetLastError(); return; }
  launch_interleave_pn(c->grid.pts, c->grid.nrm, c->grid.n, pn, c->stream);
  c->grid.pn = pn;
}
static void set_warm_args(const cilhip_ctx* c, IterArgs& wa) {
  const size_t cap = c->ns ? c->ns : 1;
  wa.warm_extra = c->warm_extra;
  wa.warm_rec = c->d_warm_rec;
  wa.warm_rec_n = reinterpret_cast<F3*>(c->d_warm_rec + cap);
  wa.warm_src3 = wa.warm_rec_n + cap;
}
static bool weighted(const cilhip_ctx* c) { return c->weight_fn != nullptr || c->cw_point_kind != CW_UNITY || c->cw_plane_kind != CW_UNITY; }
// The per-pair weights of the combined-metric classes (PointToPoint/PointToPlaneCorrWeightEvaluatorT of
// icp_single_transform_combined_metric.hpp:11-14; the point-to-point class has none): evaluator(corr.value) times the
// metric weight, in f32.  RBF coefficient as common_pair_evaluators.hpp:53.
static CorrWeights corr_weights_of(const cilhip_ctx* c, bool combined_metric, float w_p2p, float w_p2pl) {
  CorrWeights w{};
  w.enabled = (combined_metric && weighted(c)) ? 1 : 0;
  w.point_kind = c->cw_point_kind; w.plane_kind = c->cw_plane_kind;
  w.point_coeff = -0.5f / (c->cw_point_sigma * c->cw_point_sigma);
  w.plane_coeff = -0.5f / (c->cw_plane_sigma * c->cw_plane_sigma);
  w.w_p2p = w_p2p; w.w_p2pl = w_p2pl;
  // (a caller's own evaluators: prepare_pair_weights() has put the weights of the stored correspondences into the tables)
  if (w.enabled && c->weight_fn) { w.point_table = c->d_wtab; w.plane_table = c->d_wtab + c->wtab_cap; }
  return w;
}
static CorrWeights corr_weights_of(const cilhip_ctx* c, const cilhip_icp_params* p) {
  return corr_weights_of(c, p->metric == CILHIP_METRIC_COMBINED, p->w_p2p, p->w_p2pl);
}
// The warm-started iteration (k_warm) needs stored matches, unit weights and the first Gauss-Newton step's plain terms -- the
// same engine conditions as the in-tile accumulation, but no tiles: it also serves clouds the tiles do not (a source much
// sparser than the target: BASELINE configs[3]).
// a feature adaptor is in force (6-D point+normal or point+colour, 9-D point+normal+colour): correspondences are compared by feature distance
static bool feat6(const cilhip_ctx* c) { return c->normal_weight > 0.0f || (c->feature_kind == 2 && c->color_weight > 0.0f); }
static bool warm_capable(const cilhip_ctx* c) {
  // (the symmetric objective -- source normals set, option symmetric_metric on -- runs warm-started too: k_warm<., ., SYM> streams the source normals)
  return c->warm_start && c->ns >= 65536 && !filters_active(c) && !weighted(c) && !feat6(c) && !c->fused;
}
// k_self_nn's nearest-other-point table (4 B per target point, 0.5 ms at 10M): built by the first warm-capable run on a target
static int ensure_safe2(cilhip_ctx* c) {
  if (c->d_safe2) return CILHIP_OK;
  CK(c, hipMalloc(&c->d_safe2, (c->grid.n ? c->grid.n : 1) * sizeof(float)));
  launch_self_nn(c->grid, c->d_safe2, c->stream);
  return CILHIP_OK;
}
static bool tile_accumulation(const cilhip_ctx* c) {
  return c->tile_acc && use_tiled(c) && !filters_active(c) && !weighted(c) && !feat6(c) && !(c->d_src_nrm && c->symmetric) && !c->fused;
}

// ---- option "tie_rule": the reference's order among exactly equidistant nearest points ------------------------------------------
// Is the option in force for this context's searches?  The order is the reference's kd-tree over the TARGET POINTS: it covers the
// SECOND_TO_FIRST matches (also the forward half of BOTH) under rigid and affine transforms.  Feature adaptors search another
// space (nanoflann's DIM = 6 / 9 tree: tie_feat_on below), the reverse matches of FIRST_TO_SECOND / BOTH a tree over the transformed
// SOURCE that the reference rebuilds every iteration (rev_tie_aware); the feature adaptors' reverse searches keep the lowest index
// (tie_rule 2) or are refused (tie_rule 1, the explicit request).  An
// index shard of a target (cilhip_set_shard_info) notices and counts ties like any context, but never builds tables from its own points:
// the order belongs to the WHOLE target's tree -- whoever owns the shards loads it (cilhip_load_tie_order with the global indices) and
// runs the two-key protocol between them (cilhip_icp_order_keys).
static bool tie_mode_on(const cilhip_ctx* c) { return c->tie_rule != 0 && !feat6(c); }
// ... and over 6-D / 9-D features: the forward (SECOND_TO_FIRST) search of a whole target follows the reference's DIM = 6 / 9 tree
// (its order tables: tie_order_build_device_features; tie_settle<true> / tie_before_nd on the device)
static bool tie_feat_on(const cilhip_ctx* c) { return c->tie_rule != 0 && feat6(c) && c->search_dir == 0 && !c->partial_target && !c->index_offset; }
static TieDev tie_dev_of(const cilhip_ctx* c) {
  TieDev t{};
  if (feat6(c)) {
    t.mode = tie_feat_on(c) ? 1 : 0;
    t.leaf_slot = t.mode ? c->d_tief_leaf_slot : nullptr;
    t.nodes = c->d_tief_nodes;
    t.counters = c->d_tie_counters;
    return t;
  }
  t.mode = tie_mode_on(c) ? 1 : 0;
  t.leaf_slot = t.mode ? c->d_tie_leaf_slot : nullptr;
  t.nodes = c->d_tie_nodes;
  t.counters = c->d_tie_counters;
  return t;
}
// Order tables by this target's ORIGINAL (local) index -> device, by sorted position.
static int load_tie_tables(cilhip_ctx* c, const uint32_t* leaf_by_index, const uint32_t* slot_by_index, const cilhip::TieNode* nodes, size_t n_nodes) {
  static_assert(sizeof(cilhip::TieNode) == sizeof(uint4), "TieNode is read as one 16-byte record");
  CK(c, hipSetDevice(c->device));
  drop_tie_tables(c);
  const size_t n = c->grid.n;
  uint32_t *d_leaf = nullptr, *d_slot = nullptr;
  // (d_tie_leaf_slot != null is the "tables loaded" flag: nothing may be left half set when an allocation fails)
  hipError_t e = hipMalloc(&c->d_tie_leaf_slot, (n ? n : 1) * sizeof(uint2));
  if (e == hipSuccess) e = hipMalloc(&c->d_tie_nodes, (n_nodes ? n_nodes : 1) * sizeof(uint4));
  if (e == hipSuccess) e = hipMalloc(&d_leaf, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&d_slot, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess && n) {
    e = hipMemcpyAsync(d_leaf, leaf_by_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_slot, slot_by_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && n_nodes) e = hipMemcpyAsync(c->d_tie_nodes, nodes, n_nodes * sizeof(uint4), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_tie_tables_by_position(c->grid.pts, c->grid.n, d_leaf, d_slot, c->d_tie_leaf_slot, c->stream); e = hipGetLastError(); }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // (the host arrays and the two staging buffers live on this frame)
  if (d_leaf) (void)hipFree(d_leaf);
  if (d_slot) (void)hipFree(d_slot);
  if (e != hipSuccess) { drop_tie_tables(c); c->err = std::string("tie_rule: loading the order tables: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
  c->tie_max_depth = 0;
  for (size_t k = 0; k < n_nodes; ++k) c->tie_max_depth = std::max(c->tie_max_depth, (int)(nodes[k].info >> 3));
  return CILHIP_OK;
}
// The tables of THIS context's target, built on the device from the grid's own records (tie_build.hip).
static int build_tie_tables(cilhip_ctx* c) {
  if (c->d_tie_leaf_slot || !c->has_target) return CILHIP_OK;
  const auto t0 = std::chrono::steady_clock::now();
  CK(c, hipSetDevice(c->device));
  const uint32_t n = c->grid.n;
  drop_tie_tables(c);
  uint32_t *d_leaf = nullptr, *d_slot = nullptr;
  uint4* d_nodes = nullptr;
  size_t n_nodes = 0;
  int depth = 0;
  hipError_t e = hipMalloc(&d_leaf, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&d_slot, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&c->d_tie_leaf_slot, (n ? n : 1) * sizeof(uint2));
  if (e == hipSuccess) e = tie_order_build_device(nullptr, c->grid.pts, n, c->stream, d_leaf, d_slot, &d_nodes, &n_nodes, &depth);
  if (e == hipSuccess && !d_nodes) e = hipMalloc(&d_nodes, sizeof(uint4));      // (an empty target)
  if (e == hipSuccess && n) { launch_tie_tables_by_position(c->grid.pts, n, d_leaf, d_slot, c->d_tie_leaf_slot, c->stream); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (d_leaf) (void)hipFree(d_leaf);
  if (d_slot) (void)hipFree(d_slot);
  if (e != hipSuccess) {
    if (d_nodes) (void)hipFree(d_nodes);
    drop_tie_tables(c);
    c->err = std::string("tie_rule: building the order tables: ") + hipGetErrorString(e);
    return CILHIP_ERR_HIP;
  }
  c->d_tie_nodes = d_nodes;
  c->tie_max_depth = depth;
  c->tie_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ++c->tie_builds;
  return CILHIP_OK;
}
// the counters of the searches since the last launch_init_state (a host round trip)
static int read_tie_counters(cilhip_ctx* c, unsigned int out[4]) {
  CK(c, hipMemcpyAsync(out, c->d_tie_counters, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  return CILHIP_OK;
}
// The reverse matches' order (FIRST_TO_SECOND / BOTH): what k_reverse_search is handed.  Without valid tables it counts the tied target
// points (counters[3]) and keeps the lowest source index.
static TieDev tie_dev_rev(const cilhip_ctx* c) {
  TieDev t{};
  t.mode = tie_mode_on(c) ? 1 : 0;
  t.leaf_slot = (t.mode && c->rev_tie_valid) ? c->d_rev_tie_leaf_slot : nullptr;
  t.nodes = c->d_rev_tie_nodes;
  t.counters = c->d_tie_counters;
  return t;
}
// The order tables of the tree the reference builds over the source transformed by T (src_points_trans = transform_ * src, the engine's
// pinned f32 expression; correspondence_search_kd_tree.hpp:185-222), by position in the source grid.  Device work per search: the
// transform of the source (its original order) and tie_order_build_device over it (2 ms for a 110k-point frame, 25 ms at 10M).
static int build_rev_tie_tables(cilhip_ctx* c, const float T[16]) {
  if (c->rev_tie_valid && memcmp(c->rev_tie_T, T, sizeof(c->rev_tie_T)) == 0) return CILHIP_OK;
  c->rev_tie_valid = false;
  const uint32_t n = c->ns;
  if (!c->has_src_grid || n == 0) return CILHIP_OK;
  CK(c, hipSetDevice(c->device));
  float* d_q = nullptr;
  uint32_t *d_leaf = nullptr, *d_slot = nullptr;
  uint4* d_nodes = nullptr;
  size_t n_nodes = 0;
  hipError_t e = hipMalloc(&d_q, (size_t)n * 3 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&d_leaf, (size_t)n * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&d_slot, (size_t)n * sizeof(uint32_t));
  if (e == hipSuccess && !c->d_rev_tie_leaf_slot) e = hipMalloc(&c->d_rev_tie_leaf_slot, (size_t)n * sizeof(uint2));
  if (e == hipSuccess) { launch_transform_original_host_T(c->d_src_xyz, n, T, d_q, c->stream); e = hipGetLastError(); }      // q = fl(T s), the engine's pinned expression
  if (e == hipSuccess) e = tie_order_build_device(d_q, nullptr, n, c->stream, d_leaf, d_slot, &d_nodes, &n_nodes, nullptr);
  if (e == hipSuccess) {
    if (c->d_rev_tie_nodes) (void)hipFree(c->d_rev_tie_nodes);
    c->d_rev_tie_nodes = d_nodes; c->rev_tie_nodes_cap = n_nodes; d_nodes = nullptr;
    launch_tie_tables_by_position(c->src_grid.pts, n, d_leaf, d_slot, c->d_rev_tie_leaf_slot, c->stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (d_q) (void)hipFree(d_q);
  if (d_leaf) (void)hipFree(d_leaf);
  if (d_slot) (void)hipFree(d_slot);
  if (d_nodes) (void)hipFree(d_nodes);
  if (e != hipSuccess) { c->err = std::string("tie_rule: the transformed source's order tables: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
  memcpy(c->rev_tie_T, T, sizeof(c->rev_tie_T));
  c->rev_tie_valid = true;
  ++c->rev_tie_builds;
  return CILHIP_OK;
}
// tie_rule 1: the tables before the first search; refusals of the explicit request (see tie_mode_on)
static int ensure_feature_arrays(cilhip_ctx* c);
static FeatSpec feat_spec_of(const cilhip_ctx* c);
// The order tables of the tree the reference's feature adaptor searches (DIM = 6: points + weighted normals or colours; 9: + colours), for
// this target under the CURRENT feature options, built on the device (tie_build.hip).
static int build_feat_tie_tables(cilhip_ctx* c) {
  if (c->d_tief_leaf_slot || !c->has_target) return CILHIP_OK;
  CK(c, hipSetDevice(c->device));
  { const int rc = ensure_feature_arrays(c); if (rc) return rc; }
  const FeatSpec f = feat_spec_of(c);
  const int dim = c->feature_kind == 2 ? 9 : 6;
  if (!f.dst || (dim == 9 && !f.dst2)) return fail(c, CILHIP_ERR_INVALID, "tie_rule: the target's feature attributes (normals / colours) are not set");
  const uint32_t n = c->grid.n;
  uint32_t *d_leaf = nullptr, *d_slot = nullptr;
  uint4* d_nodes = nullptr;
  size_t n_nodes = 0;
  hipError_t e = hipMalloc(&d_leaf, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&d_slot, (n ? n : 1) * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&c->d_tief_leaf_slot, (n ? n : 1) * sizeof(uint2));
  if (e == hipSuccess) e = tie_order_build_device_features(dim, c->grid.pts, f.dst, f.w, f.dst2, f.w2, n, c->stream, d_leaf, d_slot, &d_nodes, &n_nodes, nullptr);
  if (e == hipSuccess && !d_nodes) e = hipMalloc(&d_nodes, sizeof(uint4));
  if (e == hipSuccess && n) { launch_tie_tables_by_position(c->grid.pts, n, d_leaf, d_slot, c->d_tief_leaf_slot, c->stream); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (d_leaf) (void)hipFree(d_leaf);
  if (d_slot) (void)hipFree(d_slot);
  if (e != hipSuccess) {
    if (d_nodes) (void)hipFree(d_nodes);
    drop_feat_tie_tables(c);
    c->err = std::string("tie_rule: building the feature tree's order tables: ") + hipGetErrorString(e);
    return CILHIP_ERR_HIP;
  }
  c->d_tief_nodes = d_nodes;
  ++c->tief_builds;
  return CILHIP_OK;
}
static int tie_prepare(cilhip_ctx* c, const char* what) {
  c->tie_counters_fresh = false;      // (a new search / run: whatever the host holds of the counters is history)
  if (c->tie_rule == 1 && ((feat6(c) && !tie_feat_on(c)) || (!feat6(c) && c->partial_target && !c->d_tie_leaf_slot))) {
    c->err = std::string(what) + ": tie_rule = 1 covers the SECOND_TO_FIRST search (point features: every direction) on a whole target, or on shards of a point-feature target with the whole target's order loaded (tie_rule = 2 applies the reference's order where it is defined)";
    return CILHIP_ERR_UNSUPPORTED;
  }
  if (c->tie_rule == 1 && tie_feat_on(c) && c->ns && c->grid.n) return build_feat_tie_tables(c);
  if (c->tie_rule == 1 && c->search_dir != 0) c->rev_tie_aware = true;
  if (c->tie_rule == 0) c->rev_tie_aware = false;
  if (c->tie_rule == 1 && tie_mode_on(c) && !c->partial_target && c->search_dir != 1 && c->ns && c->grid.n) return build_tie_tables(c);
  return CILHIP_OK;
}
// After a search / run: did it meet ties without tables (tie_rule 2)?  Then the tables are built and *again says: run it once more.
static int tie_check_pending(cilhip_ctx* c, bool* again) {
  *again = false;
  if (tie_feat_on(c)) {      // a feature search: its forward matches counted tied queries while the feature tree's tables were not there
    if (c->d_tief_leaf_slot || !c->ns || !c->grid.n) return CILHIP_OK;
    unsigned int cnt[4];
    if (c->tie_counters_fresh) memcpy(cnt, c->tie_counters_host, sizeof(cnt));
    else { const int rc = read_tie_counters(c, cnt); if (rc) return rc; }
    c->tie_counters_fresh = false;
    if (cnt[0] == 0u) return CILHIP_OK;
    *again = true;
    CK(c, hipMemsetAsync(c->d_tie_counters, 0, 4 * sizeof(unsigned int), c->stream));
    return build_feat_tie_tables(c);
  }
  if (!tie_mode_on(c) || c->partial_target || !c->ns || !c->grid.n) return CILHIP_OK;      // (a part of a target: its caller loads the whole cloud's order)
  const bool fwd_open = !c->d_tie_leaf_slot && c->search_dir != 1;      // (forward matches: SECOND_TO_FIRST, the forward half of BOTH)
  const bool rev_open = !c->rev_tie_aware && c->search_dir != 0;
  if (!fwd_open && !rev_open) return CILHIP_OK;
  unsigned int cnt[4];
  if (c->tie_counters_fresh) {      // (a run's read_state has just brought them over with the loop state: no second round trip)
    memcpy(cnt, c->tie_counters_host, sizeof(cnt));
  } else {
    const int rc = read_tie_counters(c, cnt);
    if (rc) return rc;
  }
  c->tie_counters_fresh = false;
  // Forward matches: any tie, the tables are built once per target.  Reverse matches: the tables cost a host tree build PER SEARCH (0.25 s
  // at 10M points against an iteration of a millisecond), so the automatic rule pays it for clouds that tie systematically -- duplicated
  // points, lattices: at least 16 tied target points and one in 100 000 -- and not for the isolated coincidence of two f32 distances in a
  // large random cloud (about one target point in ten million): those keep the lowest source index and stay counted
  // (cilhip_get_tie_rule_stats); tie_rule 1 follows the reference for every one of them.
  const bool fwd = fwd_open && cnt[0] != 0u, rev = rev_open && cnt[3] >= 16u && (unsigned long long)cnt[3] * 100000ull >= (unsigned long long)c->grid.n;
  if (!fwd && !rev) return CILHIP_OK;
  *again = true;
  CK(c, hipMemsetAsync(c->d_tie_counters, 0, 4 * sizeof(unsigned int), c->stream));      // (the repeated search counts afresh)
  if (rev) c->rev_tie_aware = true;      // (the tables themselves: per search, under its transform -- run_pair_search)
  return fwd ? build_tie_tables(c) : CILHIP_OK;
}

// filterCorrespondencesFraction then filterCorrespondencesOneToOne on the stored matches
static int apply_filters(cilhip_ctx* c) {
  if (!filters_active(c) || c->ns == 0) return CILHIP_OK;
  if (c->inlier_fraction > 0.0 && c->inlier_fraction < 1.0) {
    if (!c->d_keys) CK(c, hipMalloc(&c->d_keys, (size_t)c->ns * sizeof(unsigned long long)));
    if (!c->d_sel_state) CK(c, hipMalloc(&c->d_sel_state, filter_state_bytes()));
    launch_filter_fraction(c->d_src_sorted, c->d_nn_pos, c->d_nn_d2, c->ns, c->inlier_fraction, c->d_keys, c->d_sel_state, c->stream);
  }
  if (c->one_to_one && c->grid.n) {
    if (!c->d_winner) CK(c, hipMalloc(&c->d_winner, (size_t)c->grid.n * sizeof(unsigned long long)));
    launch_filter_one_to_one(c->d_src_sorted, c->d_nn_pos, c->d_nn_d2, c->ns, c->d_winner, c->grid.n, c->stream);
  }
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

// The 6-D feature search's inputs: vectors (normals or colours), weight, and how the source's part follows the transform being
// searched under (FeatSpec::mode; M = L^-T of that transform is refreshed by cilhip_find_correspondences -- the affine loops are
// host-driven, the device-resident loops are rigid).
static FeatSpec feat_spec_of(const cilhip_ctx* c) {
  FeatSpec f{};
  f.w = c->normal_weight;
  f.enabled = feat6(c) ? 1 : 0;
  if (c->feature_kind == 1) { f.src = c->d_src_rgb_sorted; f.dst = c->d_dst_rgb_sorted; f.mode = 2; }
  else { f.src = c->d_src_nrm ? c->d_src_nrm_sorted : nullptr; f.dst = c->grid.nrm; f.mode = c->transform_mode == 1 ? 1 : 0; }
  if (c->feature_kind == 2) { f.src2 = c->d_src_rgb_sorted; f.dst2 = c->d_dst_rgb_sorted; f.w2 = c->color_weight; }
  for (int i = 0; i < 9; ++i) f.M[i] = c->feat_M[i];
  // normal_weight = the norm of the FIRST source feature's normal part (adaptors.hpp:113-114), f32
  const float x = c->normal_weight * c->src_nrm0[0], y = c->normal_weight * c->src_nrm0[1], z = c->normal_weight * c->src_nrm0[2];
  f.nw = std::sqrt(x * x + (y * y + z * z));
  return f;
}
// sorted copy of the target's colour features (gathered by the sorted records' original indices), built on first use
static int ensure_feature_arrays(cilhip_ctx* c) {
  if (c->feature_kind == 0) return CILHIP_OK;
  if (!c->d_dst_rgb || !c->d_src_rgb) return fail(c, CILHIP_ERR_INVALID, "colour features: cilhip_set_color_features first");
  if (!c->dst_rgb_sorted_ok) {
    if (!c->d_dst_rgb_sorted) CK(c, hipMalloc(&c->d_dst_rgb_sorted, (c->grid.n ? c->grid.n : 1) * sizeof(float4)));
    launch_gather_by_w(c->grid.pts, c->d_dst_rgb, c->grid.n, c->d_dst_rgb_sorted, c->stream);
    c->dst_rgb_sorted_ok = true;
  }
  return CILHIP_OK;
}

static IterArgs make_iter_args(cilhip_ctx* c, float max_sq) {
  IterArgs a{};
  a.grid = c->grid;
  a.src = c->d_src_sorted;
  a.src_nrm = (c->d_src_nrm && c->symmetric) ? c->d_src_nrm_sorted : nullptr;
  a.feat = feat_spec_of(c);
  a.ns = c->ns;
  a.max_sq = max_sq;
  for (int i = 0; i < 3; ++i) a.dst_mean[i] = c->dst_mean[i];
  for (int i = 0; i < 9; ++i) a.tile_axes[i] = c->tile_axes[i];
  a.state = c->d_state;
  a.nn_pos = c->d_nn_pos;
  a.nn_d2 = c->d_nn_d2;
  a.partials = c->d_partials;
  a.defer_mask = c->d_defer_mask;
  a.tile_partials = c->d_partials;            // (in-tile accumulation: tile rows first, then the clean-up pass's rows)
  a.defer_flag = c->d_defer_flag;
  a.unproven_cnt = c->d_unproven;
  a.store_matches = 1;
  a.skip_if_inner_done = 0;
  a.tie = tie_dev_of(c);
  return a;
}

// The SECOND_TO_FIRST search under the transform held by c->d_state: LDS-tiled or per-lane kernel for point features, the
// 6-D feature search when a normal weight is set.
static int launch_search(cilhip_ctx* c, const IterArgs& a, int lanes = -1 /* -1: the option's own value when it names a lane count */) {
  if (lanes < 0) lanes = c->group_lanes > 0 ? c->group_lanes : 0;
  if (feat6(c)) {
    if (c->feature_kind != 1 && (!c->has_normals || !c->d_src_nrm)) return fail(c, CILHIP_ERR_INVALID, "point+normal features need target and source normals");
    if (c->feature_kind == 1 && (!a.feat.src || !a.feat.dst)) return fail(c, CILHIP_ERR_INVALID, "colour features: cilhip_set_color_features first");
    if (c->feature_kind == 2 && (!a.feat.src2 || !a.feat.dst2)) return fail(c, CILHIP_ERR_INVALID, "point+normal+colour features: cilhip_set_color_features first");
    if (c->index_offset) return fail(c, CILHIP_ERR_UNSUPPORTED, "feature adaptors are not available on target shards");
    if (use_tiled(c)) launch_search_tiled_feat6(a, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);
    else launch_search_feat6(a, c->stream);
    return CILHIP_OK;
  }
  if (use_tiled(c)) launch_search_tiled(a, IM_NONE, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);   // LDS-tiled search kernel
  else if (lanes) launch_search_group(a, lanes, c->stream);                                                    // several lanes per query (a.warm_pos: the previous matches bound the search)
  else launch_iter(a, IM_NONE, true, true, iter_num_blocks(c->ns), c->stream);                                 // per-lane global-memory search
  return CILHIP_OK;
}

// Search directions FIRST_TO_SECOND / BOTH with the transform held by c->d_state: fills c->pairs (post-filters included).
// the source's own grid (source coordinates) and what else the list-free FIRST_TO_SECOND / BOTH loop needs
static int ensure_src_grid(cilhip_ctx* c) {
  if (!c->has_src_grid && c->ns) {
    // the source indexed once, in its own coordinates (the reference builds a kd-tree over T*src per search): the reverse
    // searches go through the inverse transform
    GridBuildResult r{};
    double mean[3];
    // (its per-point attribute = the feature vectors a 6-D reverse search compares by: normals or colours)
    const float* attr = c->feature_kind == 1 ? c->d_src_rgb : c->d_src_nrm;
    const hipError_t eg = build_grid(c->d_src_xyz, attr, c->ns, c->stream, &r, mean, 1.0);
    if (eg != hipSuccess) { c->err = std::string("build_grid (source): ") + hipGetErrorString(eg); return CILHIP_ERR_HIP; }
    c->src_grid = r.grid; c->has_src_grid = true;
  }
  if (c->has_src_grid && c->feature_kind == 2 && c->d_src_rgb && !c->d_src_rgb_grid) {      // 9-D: the colours in the same order
    CK(c, hipMalloc(&c->d_src_rgb_grid, (c->ns ? c->ns : 1) * sizeof(float4)));
    launch_gather_by_w(c->src_grid.pts, c->d_src_rgb, c->ns, c->d_src_rgb_grid, c->stream);
  }
  return CILHIP_OK;
}

static int ensure_reverse_buffers(cilhip_ctx* c) {
  const int rc = ensure_src_grid(c);
  if (rc) return rc;
  if (!c->d_rev_pos) {
    CK(c, hipMalloc(&c->d_rev_pos, (c->grid.n ? c->grid.n : 1) * sizeof(uint32_t)));
    CK(c, hipMalloc(&c->d_rev_d2, (c->grid.n ? c->grid.n : 1) * sizeof(float)));
  }
  if (!c->d_src_safe2 && c->has_src_grid && c->reverse_warm) {      // (lives and dies with the source grid: drop_src_grid)
    CK(c, hipMalloc(&c->d_src_safe2, (c->ns ? c->ns : 1) * sizeof(float)));
    launch_self_nn(c->src_grid, c->d_src_safe2, c->stream);
  }
  if (!c->d_src_inv) {      // original source index -> position in the cube-sorted source (the forward matches are stored by that)
    CK(c, hipMalloc(&c->d_src_inv, (c->ns ? c->ns : 1) * sizeof(uint32_t)));
    launch_inv_perm(c->d_src_sorted, c->ns, c->d_src_inv, c->stream);
  }
  if (!c->d_grid_to_sorted && c->has_src_grid) {
    CK(c, hipMalloc(&c->d_grid_to_sorted, (c->ns ? c->ns : 1) * sizeof(uint32_t)));
    launch_grid_to_sorted(c->src_grid.pts, c->ns, c->d_src_inv, c->d_grid_to_sorted, c->stream);
  }
  return CILHIP_OK;
}

static int run_pair_search(cilhip_ctx* c, const IterArgs& a, float max_sq, const float T_host[16]) {
  if (!c->d_state_id) {
    CK(c, hipMalloc(&c->d_state_id, sizeof(IcpState)));
    const float zero[3] = {0, 0, 0};
    launch_init_state(c->d_state_id, kIdentity, zero, c->stream);
  }
  if (c->search_dir == 2 && c->ns && c->grid.n) {   // forward half of BOTH: the usual search, no filters yet
    const int src_rc = launch_search(c, a);
    if (src_rc) return src_rc;
  }
  { const int grc = ensure_src_grid(c); if (grc) return grc; }
  FeatSpec rf = a.feat;                              // the reverse search reads the source's features in the source grid's order
  rf.src = c->has_src_grid ? c->src_grid.nrm : nullptr;
  if (rf.dst2) rf.src2 = c->d_src_rgb_grid;
  if (feat6(c) && (!rf.src || !rf.dst || (rf.dst2 && !rf.src2))) return fail(c, CILHIP_ERR_INVALID, "feature search: both clouds' feature vectors are needed");
  if (!feat6(c)) { rf.w = 0.0f; rf.enabled = 0; }
  if (c->rev_tie_aware && tie_mode_on(c)) { const int trc = build_rev_tie_tables(c, T_host); if (trc) return trc; }
  const TieDev rt = tie_dev_rev(c);
  const hipError_t e = find_pairs(rf, c->grid, c->src_grid, c->d_src_xyz, (c->d_src_nrm && c->symmetric) ? c->d_src_nrm : nullptr, c->d_src_sorted, c->ns, c->d_state,
                                  c->d_state_id, T_host, max_sq, c->search_dir, c->reciprocal, c->inlier_fraction, c->one_to_one, c->d_nn_pos, c->d_nn_d2,
                                  c->pairs, c->stream, &rt);
  if (e != hipSuccess) { c->err = std::string("find_pairs: ") + hipGetErrorString(e); return CILHIP_ERR_HIP; }
  return CILHIP_OK;
}

int cilhip_find_correspondences(cilhip_ctx* c, const float T[16], float max_sq, size_t* n_found) {
  if (!c || !T) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  int rc = ensure_sorted(c, T);
  if (rc) return rc;
  rc = tie_prepare(c, "find_correspondences");
  if (rc) return rc;
  launch_init_state(c->d_state, T, c->src_mean, c->stream, nullptr, 0, nullptr, nullptr, c->d_tie_counters);
  if (feat6(c)) {
    rc = ensure_feature_arrays(c);
    if (rc) return rc;
    linear_inverse_transpose_f32(T, c->feat_M);
  }
  IterArgs a = make_iter_args(c, max_sq);
  if (c->search_dir != 0) {
    if (c->index_offset) return fail(c, CILHIP_ERR_UNSUPPORTED, "search directions other than SECOND_TO_FIRST are not available on target shards");
    rc = run_pair_search(c, a, max_sq, T);
    if (rc) return rc;
    {      // (ties met without tables -- the forward half of BOTH: the target's; the reverse matches: the transformed source's --: once more with them)
      bool again = false;
      rc = tie_check_pending(c, &again);
      if (rc) return rc;
      if (again) { a = make_iter_args(c, max_sq); rc = run_pair_search(c, a, max_sq, T); if (rc) return rc; }
    }
    memcpy(c->nn_T, T, sizeof(c->nn_T));
    drop_matches(c);
    c->have_pairs = true;
    c->matches_origin = 3;
    if (n_found) *n_found = c->pairs.count;
    return CILHIP_OK;
  }
  c->have_pairs = false;
  if (c->ns) {
    rc = launch_search(c, a);
    if (rc) return rc;
    bool again = false;      // (tie_rule 2: the search met ties and there were no order tables yet -- they exist now: once more)
    rc = tie_check_pending(c, &again);
    if (rc) return rc;
    if (again) { a = make_iter_args(c, max_sq); rc = launch_search(c, a); if (rc) return rc; }
  }
  CK(c, hipGetLastError());
  rc = apply_filters(c);
  if (rc) return rc;
  memcpy(c->nn_T, T, sizeof(c->nn_T));
  c->pending_matches = false;
  c->have_nn = true; c->d2_stale = false;
  c->matches_origin = 3;
  if (n_found) {
    unsigned long long cnt = 0;
    launch_count_found(c->d_nn_pos, c->ns, c->d_count, c->stream);
    CK(c, hipMemcpyAsync(&cnt, c->d_count, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    *n_found = (size_t)cnt;
  }
  return CILHIP_OK;
}

// nn_d2 of the matches a loop left behind (finish_run_matches), with the search's pinned arithmetic under the transform they were found under
static int ensure_d2(cilhip_ctx* c) {
  if (!c->d2_stale || !c->have_nn) return CILHIP_OK;
  CK(c, hipSetDevice(c->device));
  if (c->ns) launch_fill_d2(c->d_src_sorted, c->grid.pts, c->d_nn_pos, c->nn_T, c->ns, c->d_nn_d2, c->stream);
  c->d2_stale = false;
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

static int scatter_to_original(cilhip_ctx* c) {
  { const int drc = ensure_d2(c); if (drc) return drc; }
  const size_t cap = c->ns ? c->ns : 1;
  if (!c->d_out_idx) CK(c, hipMalloc(&c->d_out_idx, cap * sizeof(uint32_t)));
  if (!c->d_out_d2) CK(c, hipMalloc(&c->d_out_d2, cap * sizeof(float)));
  launch_scatter_nn(c->d_src_sorted, c->grid.pts, c->d_nn_pos, c->d_nn_d2, c->ns, c->d_out_idx, c->d_out_d2, c->stream);
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

// The correspondence set of the last executed iteration of cilhip_icp_run, when the loop's kernels did not leave it in memory
// (post-filters, pair-list directions, feature search, forms that store no matches): searched again under the transform that
// iteration searched under.  The search is exact and deterministic: the same set.
static int ensure_d2(cilhip_ctx* c);
static int materialize_pending(cilhip_ctx* c) {
  if (!c->pending_matches) return ensure_d2(c);      // (matches a loop left in place: their distances are formed now, if not yet)
  float T[16];
  memcpy(T, c->nn_T, sizeof(T));
  const float r = c->pending_max_sq;
  c->pending_matches = false;
  // (the search runs through the context's loop state: the finished run's state -- what cilhip_icp_state and
  //  cilhip_get_slab_violation_state report -- is put back afterwards)
  IcpState* keep = nullptr;
  CK(c, hipMalloc(&keep, sizeof(IcpState)));
  hipError_t e = hipMemcpyAsync(keep, c->d_state, sizeof(IcpState), hipMemcpyDeviceToDevice, c->stream);
  int rc = e == hipSuccess ? cilhip_find_correspondences(c, T, r, nullptr) : CILHIP_ERR_HIP;
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_state, keep, sizeof(IcpState), hipMemcpyDeviceToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(keep);
  if (e != hipSuccess && rc == CILHIP_OK) { c->err = std::string("materialize_pending: ") + hipGetErrorString(e); rc = CILHIP_ERR_HIP; }
  if (rc == CILHIP_OK) c->matches_origin = 2;
  return rc;
}

int cilhip_get_last_matches_origin(cilhip_ctx* c, int* origin) {
  if (!c || !origin) return CILHIP_ERR_INVALID;
  *origin = c->matches_origin;
  return CILHIP_OK;
}

int cilhip_get_matches_transform(cilhip_ctx* c, float T[16]) {
  if (!c || !T) return CILHIP_ERR_INVALID;
  if (!c->have_nn && !c->have_pairs && !c->pending_matches) return fail(c, CILHIP_ERR_INVALID, "get_matches_transform: no search has been run");
  memcpy(T, c->nn_T, sizeof(c->nn_T));
  return CILHIP_OK;
}

int cilhip_get_tie_count(cilhip_ctx* c, const float T[16], float max_sq, size_t* n_ties) {
  if (!c || !T || !n_ties) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  const int rc = ensure_sorted(c, T);
  if (rc) return rc;
  launch_count_ties(c->grid, c->d_src_sorted, c->ns, T, max_sq, c->d_count, c->stream);
  unsigned long long v = 0;
  CK(c, hipMemcpyAsync(&v, c->d_count, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  CK(c, hipGetLastError());
  *n_ties = (size_t)v;
  return CILHIP_OK;
}

struct cilhip_tie_order { std::vector<uint32_t> leaf, slot; std::vector<cilhip::TieNode> nodes; uint32_t n = 0; int max_depth = 0; };
int cilhip_tie_order_create(const float* xyz, size_t n, cilhip_tie_order** out) {
  if (!out || (n && !xyz) || n >= 0xFFFFFFF0ull) return CILHIP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0, dev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipGetDevice(&dev) != hipSuccess) return CILHIP_ERR_NO_DEVICE;      // (built on the device: no CPU fallback)
  cilhip_tie_order* o = nullptr;
  float* d_xyz = nullptr;
  uint32_t *d_leaf = nullptr, *d_slot = nullptr;
  uint4* d_nodes = nullptr;
  size_t n_nodes = 0;
  hipStream_t s = nullptr;
  hipError_t e = hipSuccess;
  int rc = CILHIP_OK;
  try {
    o = new cilhip_tie_order();
    o->n = (uint32_t)n;
    o->leaf.assign(n, 0u); o->slot.assign(n, 0u);
    if (n) {
      e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipMalloc(&d_xyz, n * 3 * sizeof(float));
      if (e == hipSuccess) e = hipMalloc(&d_leaf, n * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMalloc(&d_slot, n * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = tie_order_build_device(d_xyz, nullptr, (uint32_t)n, s, d_leaf, d_slot, &d_nodes, &n_nodes, &o->max_depth);
      if (e == hipSuccess) { o->nodes.resize(n_nodes); e = hipMemcpyAsync(o->nodes.data(), d_nodes, n_nodes * sizeof(uint4), hipMemcpyDeviceToHost, s); }
      if (e == hipSuccess) e = hipMemcpyAsync(o->leaf.data(), d_leaf, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipMemcpyAsync(o->slot.data(), d_slot, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
  } catch (...) { rc = CILHIP_ERR_HIP; }      // (out of host memory: never across the C boundary)
  if (d_xyz) (void)hipFree(d_xyz);
  if (d_leaf) (void)hipFree(d_leaf);
  if (d_slot) (void)hipFree(d_slot);
  if (d_nodes) (void)hipFree(d_nodes);
  if (s) (void)hipStreamDestroy(s);
  if (e != hipSuccess) rc = CILHIP_ERR_HIP;
  if (rc != CILHIP_OK) { delete o; return rc; }
  *out = o;
  return CILHIP_OK;
}
void cilhip_tie_order_destroy(cilhip_tie_order* order) { delete order; }
int cilhip_tie_order_tables(const cilhip_tie_order* o, uint32_t* leaf_by_index, uint32_t* slot_by_index, void* nodes_out, size_t nodes_cap, size_t* n_nodes, int* max_depth) {
  if (!o) return CILHIP_ERR_INVALID;
  if (leaf_by_index && o->n) memcpy(leaf_by_index, o->leaf.data(), (size_t)o->n * sizeof(uint32_t));
  if (slot_by_index && o->n) memcpy(slot_by_index, o->slot.data(), (size_t)o->n * sizeof(uint32_t));
  if (nodes_out && nodes_cap) memcpy(nodes_out, o->nodes.data(), std::min(nodes_cap, o->nodes.size()) * sizeof(cilhip::TieNode));
  if (n_nodes) *n_nodes = o->nodes.size();
  if (max_depth) *max_depth = o->max_depth;
  return CILHIP_OK;
}
int cilhip_load_tie_order(cilhip_ctx* c, const cilhip_tie_order* order, const uint32_t* global_index) {
  if (!c || !order) return CILHIP_ERR_INVALID;
  if (!c->has_target) return fail(c, CILHIP_ERR_INVALID, "load_tie_order: set_target first");
  const uint32_t n = c->grid.n, N = order->n;
  if (n == 0) return load_tie_tables(c, nullptr, nullptr, order->nodes.data(), order->nodes.size());      // (a shard without target points)
  if (!global_index) {
    if (n != N) return fail(c, CILHIP_ERR_INVALID, "load_tie_order: the order was built for a cloud of another size (pass global_index for a part of it)");
    return load_tie_tables(c, order->leaf.data(), order->slot.data(), order->nodes.data(), order->nodes.size());
  }
  try {
    std::vector<uint32_t> leaf(n ? n : 1), slot(n ? n : 1);
    for (uint32_t i = 0; i < n; ++i) {
      if (global_index[i] >= N) return fail(c, CILHIP_ERR_INVALID, "load_tie_order: global index out of range");
      leaf[i] = order->leaf[global_index[i]]; slot[i] = order->slot[global_index[i]];
    }
    return load_tie_tables(c, leaf.data(), slot.data(), order->nodes.data(), order->nodes.size());
  } catch (...) { return fail(c, CILHIP_ERR_HIP, "load_tie_order: out of host memory"); }
}
int cilhip_build_tie_order(cilhip_ctx* c) {
  if (!c) return CILHIP_ERR_INVALID;
  if (!c->has_target) return fail(c, CILHIP_ERR_INVALID, "build_tie_order: set_target first");
  if (c->partial_target) return fail(c, CILHIP_ERR_INVALID, "build_tie_order: this context holds a PART of a target (cilhip_set_shard_info): the order is the whole cloud's -- cilhip_tie_order_create + cilhip_load_tie_order");
  return build_tie_tables(c);
}
int cilhip_get_tie_order_info(cilhip_ctx* c, cilhip_tie_order_info* out) {
  if (!c || !out) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  unsigned int cnt[4];
  { const int rc = read_tie_counters(c, cnt); if (rc) return rc; }
  out->loaded = (feat6(c) ? c->d_tief_leaf_slot != nullptr : c->d_tie_leaf_slot != nullptr) ? 1 : 0; out->builds = c->tie_builds + c->tief_builds; out->build_ms = c->tie_build_ms; out->pending = cnt[0];
  return CILHIP_OK;
}

int cilhip_get_tie_rule_stats(cilhip_ctx* c, size_t* tied_queries, size_t* repointed) {
  if (!c) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  unsigned int cnt[4];
  { const int rc = read_tie_counters(c, cnt); if (rc) return rc; }
  if (tied_queries) *tied_queries = (size_t)cnt[1] + (size_t)cnt[0] + (size_t)cnt[3];      // (settled from tables + met without them, forward and reverse)
  if (repointed) *repointed = (size_t)cnt[2];
  return CILHIP_OK;
}

int cilhip_get_nn(cilhip_ctx* c, uint32_t* nn_idx, float* nn_d2, int mem) {
  if (!c) return CILHIP_ERR_INVALID;
  { const int prc = materialize_pending(c); if (prc) return prc; }
  if (c->have_pairs) return fail(c, CILHIP_ERR_UNSUPPORTED, "get_nn: the last search ran in a direction whose result is a pair list; use get_correspondences");
  if (!c->have_nn) return fail(c, CILHIP_ERR_INVALID, "get_nn: no search has been run");
  CK(c, hipSetDevice(c->device));
  int rc = scatter_to_original(c);
  if (rc) return rc;
  const hipMemcpyKind k = mem == CILHIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (nn_idx && c->ns) CK(c, hipMemcpyAsync(nn_idx, c->d_out_idx, (size_t)c->ns * 4, k, c->stream));
  if (nn_d2 && c->ns) CK(c, hipMemcpyAsync(nn_d2, c->d_out_d2, (size_t)c->ns * 4, k, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  return CILHIP_OK;
}

static int get_correspondences_impl(cilhip_ctx* c, uint64_t* i1, uint64_t* i2, float* val, size_t cap, size_t* n_out);
int cilhip_get_correspondences(cilhip_ctx* c, uint64_t* i1, uint64_t* i2, float* val, size_t cap, size_t* n_out) {
  if (!c || !n_out) return CILHIP_ERR_INVALID;
  // (host vectors of the size of the source: an allocation failure must not cross the C boundary)
  try { return get_correspondences_impl(c, i1, i2, val, cap, n_out); }
  catch (const std::bad_alloc&) { return fail(c, CILHIP_ERR_HIP, "get_correspondences: out of host memory"); }
  catch (...) { return fail(c, CILHIP_ERR_HIP, "get_correspondences: unexpected exception"); }
}
static int get_correspondences_impl(cilhip_ctx* c, uint64_t* i1, uint64_t* i2, float* val, size_t cap, size_t* n_out) {
  { const int prc = materialize_pending(c); if (prc) return prc; }
  if (c->have_pairs) {
    // pair list of FIRST_TO_SECOND / BOTH: stored ascending (first, second); the reference leaves the set sorted by
    // value after the fraction filter (correspondence.hpp:61) and by (indexInSecond, value) after the FIRST_TO_SECOND
    // one-to-one filter (:74-82) -- reproduce that (ties keep the stored order)
    const size_t cnt = c->pairs.count;
    *n_out = cnt;
    if (cnt > cap) return fail(c, CILHIP_ERR_INVALID, "get_correspondences: capacity too small");
    if (cnt == 0) return CILHIP_OK;
    CK(c, hipSetDevice(c->device));
    std::vector<uint32_t> f(cnt), sc(cnt);
    std::vector<float> v(cnt);
    CK(c, hipMemcpyAsync(f.data(), c->pairs.first, cnt * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipMemcpyAsync(sc.data(), c->pairs.second, cnt * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipMemcpyAsync(v.data(), c->pairs.d2, cnt * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    std::vector<size_t> ord(cnt);
    for (size_t k = 0; k < cnt; ++k) ord[k] = k;
    const bool frac = c->inlier_fraction > 0.0 && c->inlier_fraction < 1.0;
    if (c->one_to_one && c->search_dir == 1)
      std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return sc[x] != sc[y] ? sc[x] < sc[y] : v[x] < v[y]; });
    else if (frac)
      std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return v[x] < v[y]; });
    for (size_t k = 0; k < cnt; ++k) {
      if (i1) i1[k] = f[ord[k]];
      if (i2) i2[k] = sc[ord[k]];
      if (val) val[k] = v[ord[k]];
    }
    return CILHIP_OK;
  }
  if (!c->have_nn) return fail(c, CILHIP_ERR_INVALID, "get_correspondences: no search has been run");
  std::vector<uint32_t> idx(c->ns ? c->ns : 1);
  std::vector<float> d2(c->ns ? c->ns : 1);
  int rc = cilhip_get_nn(c, idx.data(), d2.data(), CILHIP_MEM_HOST);
  if (rc) return rc;
  // order-preserving compaction in ascending source index (kd_tree_utilities.hpp:45-50)
  size_t cnt = 0;
  for (uint32_t i = 0; i < c->ns; ++i) {
    if (idx[i] == NONE_U32) continue;
    if (cnt < cap) {
      if (i1) i1[cnt] = idx[i];
      if (i2) i2[cnt] = i;
      if (val) val[cnt] = d2[i];
    }
    ++cnt;
  }
  *n_out = cnt;
  if (cnt > cap) return fail(c, CILHIP_ERR_INVALID, "get_correspondences: capacity too small");
  // the reference's filters leave the set sorted: by value after the fraction filter (correspondence.hpp:61),
  // by indexInFirst after the one-to-one filter (:86-94); reproduce that order (ties: ascending source index)
  if (filters_active(c) && cnt > 1 && i1 && i2 && val) {
    std::vector<size_t> ord(cnt);
    for (size_t k = 0; k < cnt; ++k) ord[k] = k;
    if (c->one_to_one) std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return i1[a] < i1[b]; });
    else std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return val[a] < val[b]; });
    std::vector<uint64_t> t1(cnt), t2(cnt); std::vector<float> tv(cnt);
    for (size_t k = 0; k < cnt; ++k) { t1[k] = i1[ord[k]]; t2[k] = i2[ord[k]]; tv[k] = val[ord[k]]; }
    memcpy(i1, t1.data(), cnt * sizeof(uint64_t)); memcpy(i2, t2.data(), cnt * sizeof(uint64_t)); memcpy(val, tv.data(), cnt * sizeof(float));
  }
  return CILHIP_OK;
}

static void pack_T(const double L[9], const double t[3], float T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  T[15] = 1.f;
  for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) T[cc * 4 + r] = (float)L[r * 3 + cc]; T[12 + r] = (float)t[r]; }
}

// ---- a caller's own weight evaluators -----------------------------------------------------------------------------------------
// The reference's combined-metric classes take their evaluators as template arguments (icp_single_transform_combined_metric.hpp:10-14)
// and the estimators call them per correspondence: evaluator(corr.indexInFirst, corr.indexInSecond, corr.value)
// (transform_estimation.hpp:303, :332, :432, :453).  A functor cannot cross a C boundary onto the device; with a callback set, every
// estimate brings the stored correspondence set to the host, lets the callback fill both weights of every pair, and the accumulation
// pass reads them from tables (CorrWeights::point_table) instead of evaluating a kind.  Stored order: ascending source index
// (SECOND_TO_FIRST), or the pair list's (first, second).
static int prepare_pair_weights_impl(cilhip_ctx* c);
static int prepare_pair_weights(cilhip_ctx* c) {
  if (!c->weight_fn) return CILHIP_OK;
  // (host vectors of the size of the correspondence set: an allocation failure must not cross the C boundary; neither may whatever a
  //  C++ callback lets escape)
  try {
    return prepare_pair_weights_impl(c);
  } catch (const std::bad_alloc&) {
    return fail(c, CILHIP_ERR_HIP, "pair-weight callback: out of host memory for the correspondence set");
  } catch (...) {
    return fail(c, CILHIP_ERR_INVALID, "pair-weight callback: an exception escaped the callback");
  }
}
static int prepare_pair_weights_impl(cilhip_ctx* c) {
  const bool pairs = c->have_pairs;
  const size_t slots = pairs ? c->pairs.count : c->ns;      // stream positions
  if (slots > c->wtab_cap || !c->d_wtab) {
    if (c->d_wtab) (void)hipFree(c->d_wtab);
    if (c->d_wtab_in) (void)hipFree(c->d_wtab_in);
    c->d_wtab = c->d_wtab_in = nullptr; c->wtab_cap = 0;
    const size_t cap = slots ? slots : 1;
    CK(c, hipMalloc(&c->d_wtab, 2 * cap * sizeof(float)));
    CK(c, hipMalloc(&c->d_wtab_in, 2 * cap * sizeof(float)));
    c->wtab_cap = cap;
  }
  if (slots == 0) return CILHIP_OK;
  std::vector<uint64_t> i1(slots), i2(slots);
  std::vector<float> val(slots), wq(slots, 0.0f), wl(slots, 0.0f);
  size_t cnt = 0;
  if (pairs) {
    std::vector<uint32_t> f(slots), sc(slots);
    CK(c, hipMemcpyAsync(f.data(), c->pairs.first, slots * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipMemcpyAsync(sc.data(), c->pairs.second, slots * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipMemcpyAsync(val.data(), c->pairs.d2, slots * 4, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < slots; ++k) { i1[k] = f[k]; i2[k] = sc[k]; }
    cnt = slots;
  } else {
    std::vector<uint32_t> idx(slots);
    std::vector<float> d2(slots);
    const int rc = cilhip_get_nn(c, idx.data(), d2.data(), CILHIP_MEM_HOST);
    if (rc) return rc;
    for (size_t i = 0; i < slots; ++i)
      if (idx[i] != NONE_U32) { i1[cnt] = idx[i]; i2[cnt] = i; val[cnt] = d2[i]; ++cnt; }
  }
  if (cnt) c->weight_fn(c->weight_user, i1.data(), i2.data(), val.data(), cnt, wq.data(), wl.data());
  if (pairs) {
    CK(c, hipMemcpyAsync(c->d_wtab, wq.data(), slots * 4, hipMemcpyHostToDevice, c->stream));
    CK(c, hipMemcpyAsync(c->d_wtab + c->wtab_cap, wl.data(), slots * 4, hipMemcpyHostToDevice, c->stream));
  } else {
    // by original source index (unmatched: 0, never read), then into the sorted order the pass streams over
    std::vector<float> oq(slots, 0.0f), ol(slots, 0.0f);
    for (size_t k = 0; k < cnt; ++k) { oq[i2[k]] = wq[k]; ol[i2[k]] = wl[k]; }
    CK(c, hipMemcpyAsync(c->d_wtab_in, oq.data(), slots * 4, hipMemcpyHostToDevice, c->stream));
    CK(c, hipMemcpyAsync(c->d_wtab_in + c->wtab_cap, ol.data(), slots * 4, hipMemcpyHostToDevice, c->stream));
    launch_gather1_by_w(c->d_src_sorted, c->d_wtab_in, (uint32_t)slots, c->d_wtab, c->stream);
    launch_gather1_by_w(c->d_src_sorted, c->d_wtab_in + c->wtab_cap, (uint32_t)slots, c->d_wtab + c->wtab_cap, c->stream);
  }
  CK(c, hipStreamSynchronize(c->stream));      // (the host vectors go out of scope)
  return CILHIP_OK;
}

// Accumulate over the stored matches (transform = nn_T) and bring the reduced sums to the host.
static int accumulate_stored(cilhip_ctx* c, int metric, const double innerL[9], const double innert[3], double sums[SUMS_MAX],
                             const CorrWeights* cw = nullptr) {
  IcpState hs;
  launch_init_state(c->d_state, c->nn_T, c->src_mean, c->stream);
  if (innerL) {
    CK(c, hipMemcpyAsync(&hs, c->d_state, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 9; ++i) { hs.innerL[i] = (float)innerL[i]; hs.dLd[i] = innerL[i]; }
    for (int i = 0; i < 3; ++i) { hs.innert[i] = (float)innert[i]; hs.dtd[i] = innert[i]; }
    CK(c, hipMemcpyAsync(c->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice, c->stream));
  }
  IterArgs a = make_iter_args(c, 0.0f);
  if (cw) a.cw = *cw;
  // (a pair list -- FIRST_TO_SECOND / BOTH -- carries its own view of the source, per pair)
  if (c->have_pairs) {
    a.src = c->pairs.src_view; a.ns = c->pairs.count; a.nn_pos = c->pairs.posd; a.nn_d2 = c->pairs.d2;
    a.src_nrm = (c->d_src_nrm && c->symmetric) ? c->pairs.nrm_view : nullptr;
  }
  const int nb = iter_num_blocks(a.ns);
  for (int i = 0; i < SUMS_MAX; ++i) sums[i] = 0.0;
  if (a.ns == 0) return CILHIP_OK;
  if (nb > c->partial_blocks) {
    if (c->d_partials) (void)hipFree(c->d_partials);
    c->d_partials = nullptr; c->partial_blocks = 0;
    CK(c, hipMalloc(&c->d_partials, (size_t)nb * SUMS_MAX * sizeof(double)));
    c->partial_blocks = nb;
  }
  a.partials = c->d_partials;
  launch_iter(a, metric, false, false, nb, c->stream);
  launch_reduce_partials(c->d_partials, nb, c->d_stage, c->d_sums, c->stream);
  CK(c, hipGetLastError());
  CK(c, hipMemcpyAsync(sums, c->d_sums, SUMS_MAX * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  return CILHIP_OK;
}

int cilhip_estimate_point_to_point(cilhip_ctx* c, float dT[16], double* sums_out, int* ok) {
  if (!c || !dT) return CILHIP_ERR_INVALID;
  { const int prc = materialize_pending(c); if (prc) return prc; }
  if (!c->have_nn) return fail(c, CILHIP_ERR_INVALID, "estimate: run find_correspondences first");
  CK(c, hipSetDevice(c->device));
  double sums[SUMS_MAX];
  int rc = accumulate_stored(c, IM_KABSCH, nullptr, nullptr, sums);
  if (rc) return rc;
  double L[9], t[3];
  kabsch_from_sums(sums, L, t);
  pack_T(L, t, dT);
  if (sums_out) memcpy(sums_out, sums, 16 * sizeof(double));
  if (ok) *ok = sums[0] >= 3.0;
  return CILHIP_OK;
}

int cilhip_estimate_combined(cilhip_ctx* c, float w_p2p, float w_p2pl, size_t max_iter, float conv_tol, float dT[16],
                             double* AtA_out, double* Atb_out, int* converged) {
  if (!c || !dT) return CILHIP_ERR_INVALID;
  { const int prc = materialize_pending(c); if (prc) return prc; }
  if (!c->have_nn && !c->have_pairs) return fail(c, CILHIP_ERR_INVALID, "estimate: run find_correspondences first");
  CK(c, hipSetDevice(c->device));
  { const int wrc = prepare_pair_weights(c); if (wrc) return wrc; }
  double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  memcpy(dT, kIdentity, sizeof(kIdentity));
  if (converged) *converged = 0;
  if (AtA_out) for (int i = 0; i < 36; ++i) AtA_out[i] = 0.0;
  if (Atb_out) for (int i = 0; i < 6; ++i) Atb_out[i] = 0.0;
  const bool wp = w_p2p > 0.0f, wl = w_p2pl > 0.0f;
  if (!wp && !wl) return CILHIP_OK;                      // transform_estimation.hpp:264-272
  if (wl && !c->has_normals) return CILHIP_OK;           // dst_p.cols() != dst_n.cols() -> identity, false
  const int metric = (wp && wl) ? IM_BOTH : (wl ? IM_PLANE : IM_POINT);
  float smt[3];
  transform_point(c->nn_T, c->src_mean[0], c->src_mean[1], c->src_mean[2], smt[0], smt[1], smt[2]);
  int conv = 0;
  const CorrWeights cw = corr_weights_of(c, true, w_p2p, w_p2pl);
  if (max_iter == 0) {      // the loop body never runs; "no usable terms" (no correspondences) still means identity (:264-272)
    double sums[SUMS_MAX];
    const int rc = accumulate_stored(c, metric, L, t, sums, &cw);
    if (rc) return rc;
    if (!(sums[0] > 0.0)) return CILHIP_OK;
  }
  for (size_t it = 0; it < max_iter; ++it) {
    double sums[SUMS_MAX];
    int rc = accumulate_stored(c, metric, L, t, sums, &cw);
    if (rc) return rc;
    if (!(sums[0] > 0.0)) return CILHIP_OK;              // no correspondences: identity
    double AtA[36], Atb[6], dth[6];
    if (cw.enabled) gn_normal_equations(sums, wp ? 1.0 : 0.0, wl ? 1.0 : 0.0, AtA, Atb, true);   // (metric weights inside the per-pair weights)
    else gn_normal_equations(sums, wp ? (double)w_p2p : 0.0, wl ? (double)w_p2pl : 0.0, AtA, Atb);
    if (it == 0) {
      if (AtA_out) memcpy(AtA_out, AtA, sizeof(AtA));
      if (Atb_out) memcpy(Atb_out, Atb, sizeof(Atb));
    }
    ldlt6_solve(AtA, Atb, dth);
    rigid_gn_update(dth, L, t);
    double nrm = 0.0;
    for (int i = 0; i < 6; ++i) nrm += dth[i] * dth[i];
    if (std::sqrt(nrm) < (double)conv_tol) { conv = 1; break; }
  }
  double tt[3];
  for (int r = 0; r < 3; ++r)
    tt[r] = t[r] - (L[r * 3] * (double)smt[0] + L[r * 3 + 1] * (double)smt[1] + L[r * 3 + 2] * (double)smt[2]) + (double)c->dst_mean[r];
  pack_T(L, tt, dT);
  if (converged) *converged = conv;
  return CILHIP_OK;
}

// ---- two correspondence sets in one combined-metric estimate: CorrespondenceSearchCombinedMetricCombiner -------------
// (registration/correspondence_search_combined_metric_combiner.hpp:8-81: the point-to-point terms read one engine's
//  correspondences, the point-to-plane terms another's -- other radius, features, filters -- over the same two clouds.)
// Each context accumulates its own block of the sums over its own stored matches (the point block, slots [28, 44), on
// c_point; the plane block, slots [0, 28), on c_plane); the normal equations are assembled from the two.
static int combined_two_sets_step(cilhip_ctx* cp, cilhip_ctx* cl, bool wp, bool wl, float w_p2p, float w_p2pl, const double L[9], const double t[3],
                                  double sums[SUMS_MAX], bool* weighted_out) {
  CorrWeights cwp = corr_weights_of(cp, true, w_p2p, w_p2pl), cwl = corr_weights_of(cl, true, w_p2p, w_p2pl);
  const bool any = cwp.enabled || cwl.enabled;      // (some evaluator is not Unity: both blocks then carry their metric weight per pair)
  cwp.enabled = cwl.enabled = any ? 1 : 0;
  *weighted_out = any;
  for (int i = 0; i < SUMS_MAX; ++i) sums[i] = 0.0;
  double s1[SUMS_MAX], s2[SUMS_MAX];
  if (wp) {
    CK(cp, hipSetDevice(cp->device));
    const int rc = accumulate_stored(cp, IM_POINT, L, t, s1, &cwp);
    if (rc) return rc;
    for (int i = 28; i < 44; ++i) sums[i] = s1[i];
    if (!(s1[0] > 0.0)) sums[43] = 0.0;
  }
  if (wl) {
    CK(cl, hipSetDevice(cl->device));
    const int rc = accumulate_stored(cl, IM_PLANE, L, t, s2, &cwl);
    if (rc) { if (cl != cp) cp->err = cl->err; return rc; }
    for (int i = 0; i < 28; ++i) sums[i] = s2[i];
  }
  // slot 0 = the plane set's count; the point set's count travels in slot 43 (sum of the unit weights) -- keep a copy where
  // the caller can tell "no point correspondences" from "no plane correspondences"
  sums[44] = wp ? s1[0] : 0.0;
  return CILHIP_OK;
}

int cilhip_estimate_combined_two_sets(cilhip_ctx* cp, cilhip_ctx* cl, float w_p2p, float w_p2pl, size_t max_iter, float conv_tol, float dT[16],
                                      int* converged) {
  if (!cp || !cl || !dT) return CILHIP_ERR_INVALID;
  static_assert(SUMS_MAX >= 45, "slot 44 carries the point set's count");
  { int prc = materialize_pending(cp); if (prc) return prc; prc = materialize_pending(cl); if (prc) { cp->err = cl->err; return prc; } }
  if (!cp->have_nn || !cl->have_nn)
    return fail(cp, CILHIP_ERR_INVALID, "estimate (two sets): both engines need stored SECOND_TO_FIRST correspondences (find_correspondences first)");
  if (memcmp(cp->nn_T, cl->nn_T, sizeof(cp->nn_T)) != 0) return fail(cp, CILHIP_ERR_INVALID, "estimate (two sets): the two engines searched under different transforms");
  if (cp->ns != cl->ns || cp->grid.n != cl->grid.n) return fail(cp, CILHIP_ERR_INVALID, "estimate (two sets): the two engines hold different clouds");
  { int wrc = prepare_pair_weights(cp); if (wrc) return wrc; if (cl != cp) { wrc = prepare_pair_weights(cl); if (wrc) { cp->err = cl->err; return wrc; } } }
  double L[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  memcpy(dT, kIdentity, sizeof(kIdentity));
  if (converged) *converged = 0;
  bool wp = w_p2p > 0.0f, wl = w_p2pl > 0.0f;
  if (!wp && !wl) return CILHIP_OK;                      // transform_estimation.hpp:264-272
  float smt[3];
  transform_point(cp->nn_T, cp->src_mean[0], cp->src_mean[1], cp->src_mean[2], smt[0], smt[1], smt[2]);
  int conv = 0;
  const size_t steps = max_iter ? max_iter : 1;          // (max_iter 0: one pass for the "no usable terms" test only)
  for (size_t it = 0; it < steps; ++it) {
    double sums[SUMS_MAX];
    bool weighted_sums = false;
    const int rc = combined_two_sets_step(cp, cl, wp, wl, w_p2p, w_p2pl, L, t, sums, &weighted_sums);
    if (rc) return rc;
    const bool has_p2p = wp && sums[44] > 0.0, has_p2pl = wl && sums[0] > 0.0;     // :264-267
    if ((!has_p2p && !has_p2pl) || (has_p2pl && !cl->has_normals)) return CILHIP_OK;   // :269-272: identity, false
    if (max_iter == 0) break;
    double AtA[36], Atb[6], dth[6];
    if (weighted_sums) gn_normal_equations(sums, has_p2p ? 1.0 : 0.0, has_p2pl ? 1.0 : 0.0, AtA, Atb, true);
    else gn_normal_equations(sums, has_p2p ? (double)w_p2p : 0.0, has_p2pl ? (double)w_p2pl : 0.0, AtA, Atb, true);
    ldlt6_solve(AtA, Atb, dth);
    rigid_gn_update(dth, L, t);
    double nrm = 0.0;
    for (int i = 0; i < 6; ++i) nrm += dth[i] * dth[i];
    if (std::sqrt(nrm) < (double)conv_tol) { conv = 1; break; }
  }
  double tt[3];
  for (int r = 0; r < 3; ++r)
    tt[r] = t[r] - (L[r * 3] * (double)smt[0] + L[r * 3 + 1] * (double)smt[1] + L[r * 3 + 2] * (double)smt[2]) + (double)cp->dst_mean[r];
  pack_T(L, tt, dT);
  if (converged) *converged = conv;
  return CILHIP_OK;
}

// CombinedMetricSingleTransformICP over a Combiner (icp_single_transform_combined_metric.hpp:169-217 with the engine of
// correspondence_search_combined_metric_combiner.hpp): per iteration both engines search under the current transform (each with
// its own radius and options), the estimator reads the two sets, the instance class composes and tests the update norm.
// Host-driven: a few synchronisations per iteration -- this is the reference's thin combination class, not the hot path.
int cilhip_icp_run_two_sets(cilhip_ctx* cp, float max_sq_point, cilhip_ctx* cl, float max_sq_plane, const cilhip_icp_params* p, const float* T0,
                            cilhip_icp_result* out) {
  if (!cp || !cl || !p || !out) return CILHIP_ERR_INVALID;
  if (p->metric != CILHIP_METRIC_COMBINED) return fail(cp, CILHIP_ERR_INVALID, "icp_run (two sets): the combined metric is what takes two correspondence sets");
  if (cp->transform_mode != 0 || cl->transform_mode != 0) return fail(cp, CILHIP_ERR_UNSUPPORTED, "icp_run (two sets): rigid transforms");
  if (cp->search_dir != 0 || cl->search_dir != 0) return fail(cp, CILHIP_ERR_UNSUPPORTED, "icp_run (two sets): SECOND_TO_FIRST engines");
  float T[16];
  memcpy(T, T0 ? T0 : kIdentity, sizeof(T));
  memcpy(out->T, T, sizeof(T));
  out->iterations = 0; out->last_delta_norm = INFINITY; out->last_ncorr = 0;
  for (size_t it = 0; it < p->max_iter; ++it) {
    size_t n1 = 0, n2 = 0;
    int rc = cilhip_find_correspondences(cp, T, max_sq_point, &n1);
    if (rc) return rc;
    if (cl != cp) { rc = cilhip_find_correspondences(cl, T, max_sq_plane, &n2); if (rc) { cp->err = cl->err; return rc; } } else n2 = n1;
    float dT[16];
    int conv = 0;
    rc = cilhip_estimate_combined_two_sets(cp, cl, p->w_p2p, p->w_p2pl, p->max_opt_iter, p->opt_conv_tol, dT, &conv);
    if (rc) return rc;
    double L[9], t[3];
    for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) L[r * 3 + k] = (double)dT[k * 4 + r]; t[r] = (double)dT[12 + r]; }
    float Tn[16];
    const float delta = compose_update(L, t, T, Tn);      // rotation() polish, transform_ = tform_iter * transform_, update norm (:207-216)
    memcpy(T, Tn, sizeof(T));
    out->iterations = it + 1; out->last_delta_norm = delta; out->last_ncorr = n1 > n2 ? n1 : n2;
    if (delta < p->conv_tol) break;                       // icp_base.hpp:83
  }
  memcpy(out->T, T, sizeof(T));
  return CILHIP_OK;
}

static hipEvent_t get_event(cilhip_ctx* c, size_t i);
static hipEvent_t get_acc_event(cilhip_ctx* c, size_t i);

// ---- affine variants: SimpleCombinedMetricAffineICP3f / SimplePointToPointMetricAffineICP3f ---------------------------
// Moments of the 12-unknown normal equations over the stored correspondences (matches or pair list), three streaming
// passes on the device, one copy to the host.
static int affine_accumulate(cilhip_ctx* c, bool centered, bool plane, double sums[3 * SUMS_MAX], const CorrWeights* cw = nullptr) {
  for (int i = 0; i < 3 * SUMS_MAX; ++i) sums[i] = 0.0;
  launch_init_state(c->d_state, c->nn_T, c->src_mean, c->stream);
  IterArgs a = make_iter_args(c, 0.0f);
  if (cw) a.cw = *cw;
  // (a pair list carries its own values, per pair)
  if (c->have_pairs) { a.src = c->pairs.src_view; a.ns = c->pairs.count; a.nn_pos = c->pairs.posd; a.nn_d2 = c->pairs.d2; }
  a.src_nrm = nullptr;   // (the symmetric metric exists for the rigid classes only)
  a.no_centering = centered ? 0 : 1;
  if (a.ns == 0) return CILHIP_OK;
  const int nb = iter_num_blocks(a.ns);
  if (nb > c->partial_blocks) {
    if (c->d_partials) (void)hipFree(c->d_partials);
    c->d_partials = nullptr; c->partial_blocks = 0;
    CK(c, hipMalloc(&c->d_partials, (size_t)nb * SUMS_MAX * sizeof(double)));
    c->partial_blocks = nb;
  }
  a.partials = c->d_partials;
  const int passes[3] = {IM_AFF0, IM_AFF1, IM_AFF2};
  const int np = plane ? 3 : 1;
  for (int k = 0; k < np; ++k) {
    launch_iter(a, passes[k], false, false, nb, c->stream);
    launch_reduce_partials(c->d_partials, nb, c->d_stage, c->d_sums + k * SUMS_MAX, c->stream);
  }
  CK(c, hipGetLastError());
  CK(c, hipMemcpyAsync(sums, c->d_sums, (size_t)np * SUMS_MAX * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  return CILHIP_OK;
}

int cilhip_estimate_affine(cilhip_ctx* c, float w_p2p, float w_p2pl, int centered, float dT[16], double* AtA_out,
                           double* Atb_out, size_t* n_corr, int* ok) {
  if (!c || !dT) return CILHIP_ERR_INVALID;
  { const int prc = materialize_pending(c); if (prc) return prc; }
  if (!c->have_nn && !c->have_pairs) return fail(c, CILHIP_ERR_INVALID, "estimate: run find_correspondences first");
  CK(c, hipSetDevice(c->device));
  memcpy(dT, kIdentity, sizeof(kIdentity));
  if (ok) *ok = 0;
  if (n_corr) *n_corr = 0;
  if (AtA_out) for (int i = 0; i < 144; ++i) AtA_out[i] = 0.0;
  if (Atb_out) for (int i = 0; i < 12; ++i) Atb_out[i] = 0.0;
  const bool wp = w_p2p > 0.0f, wl = w_p2pl > 0.0f;
  if (!wp && !wl) return CILHIP_OK;                      // transform_estimation.hpp:400-409
  if (wl && !c->has_normals) return CILHIP_OK;           // dst_p.cols() != dst_n.cols() -> identity, false
  double sums[3 * SUMS_MAX];
  // weight evaluators (the combined-metric class only: `centered` distinguishes it from the point-to-point class here)
  if (centered) { const int wrc = prepare_pair_weights(c); if (wrc) return wrc; }
  const CorrWeights cw = corr_weights_of(c, centered != 0, w_p2p, w_p2pl);
  const int rc = affine_accumulate(c, centered != 0, wl, sums, &cw);
  if (rc) return rc;
  const double n = sums[0];
  if (n_corr) *n_corr = (size_t)n;
  if (!(n > 0.0)) return CILHIP_OK;                      // no correspondences: identity, false
  double AtA[144], Atb[12], th[12];
  if (cw.enabled) affine_normal_equations(sums, sums + SUMS_MAX, sums + 2 * SUMS_MAX, wp ? 1.0 : 0.0, wl ? 1.0 : 0.0, AtA, Atb, true);      // (metric weights inside the per-pair weights)
  else affine_normal_equations(sums, sums + SUMS_MAX, sums + 2 * SUMS_MAX, wp ? (double)w_p2p : 0.0, wl ? (double)w_p2pl : 0.0, AtA, Atb);
  if (AtA_out) memcpy(AtA_out, AtA, sizeof(AtA));
  if (Atb_out) memcpy(Atb_out, Atb, sizeof(Atb));
  ldlt_solve_n(12, AtA, Atb, th);                        // :468 AtA.ldlt().solve(Atb)
  double L[9], t[3];
  for (int i = 0; i < 9; ++i) L[i] = th[i];              // :470-472 row-major linear part, then the translation
  for (int i = 0; i < 3; ++i) t[i] = th[9 + i];
  if (centered) {                                        // :473 tform = t_dst * tform * t_src
    float smt[3];
    transform_point(c->nn_T, c->src_mean[0], c->src_mean[1], c->src_mean[2], smt[0], smt[1], smt[2]);
    for (int r = 0; r < 3; ++r)
      t[r] = t[r] - (L[r * 3] * (double)smt[0] + L[r * 3 + 1] * (double)smt[1] + L[r * 3 + 2] * (double)smt[2]) + (double)c->dst_mean[r];
  }
  pack_T(L, t, dT);
  if (ok) *ok = ((wp ? 1.0 : 0.0) + (wl ? 1.0 : 0.0)) * n >= 4.0;
  return CILHIP_OK;
}

// icp_base.hpp:68-87 with the affine updateEstimate() (icp_single_transform_point_to_point_metric.hpp:46-65,
// icp_single_transform_combined_metric.hpp:173-217 without the rotation() polish): host-driven, one 12x12 solve per iteration.
static int icp_run_affine(cilhip_ctx* c, const cilhip_icp_params* p, const float* T0, cilhip_icp_result* out) {
  float T[16];
  memcpy(T, T0 ? T0 : kIdentity, sizeof(T));
  float delta = INFINITY;
  size_t it = 0, ncorr = 0;
  hipEvent_t e_beg = get_event(c, 0), e_end = get_event(c, 1);
  CK(c, hipEventRecord(e_beg, c->stream));
  while (it < p->max_iter) {
    int rc = cilhip_find_correspondences(c, T, p->max_sq_dist, nullptr);
    if (rc) return rc;
    float dT[16];
    if (p->metric == CILHIP_METRIC_POINT_TO_POINT) rc = cilhip_estimate_affine(c, 1.0f, 0.0f, 0, dT, nullptr, nullptr, &ncorr, nullptr);
    else rc = cilhip_estimate_affine(c, p->w_p2p, p->w_p2pl, 1, dT, nullptr, nullptr, &ncorr, nullptr);
    if (rc) return rc;
    float Tn[16] = {0};
    Tn[15] = 1.0f;                                       // transform_ = tform_iter * transform_ (f32, Eigen affine product)
    for (int r = 0; r < 3; ++r) {
      for (int cc = 0; cc < 3; ++cc) Tn[cc * 4 + r] = dT[0 * 4 + r] * T[cc * 4 + 0] + dT[1 * 4 + r] * T[cc * 4 + 1] + dT[2 * 4 + r] * T[cc * 4 + 2];
      Tn[12 + r] = (dT[0 * 4 + r] * T[12] + dT[1 * 4 + r] * T[13] + dT[2 * 4 + r] * T[14]) + dT[12 + r];
    }
    memcpy(T, Tn, sizeof(T));
    float dn = 0.0f;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) { const float v = dT[cc * 4 + r] - (r == cc ? 1.0f : 0.0f); dn += v * v; }
    for (int r = 0; r < 3; ++r) dn += dT[12 + r] * dT[12 + r];
    delta = std::sqrt(dn);
    ++it;
    if (delta < p->conv_tol) break;
  }
  CK(c, hipEventRecord(e_end, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  memcpy(out->T, T, sizeof(T));
  out->iterations = it;
  out->last_delta_norm = delta;
  out->last_ncorr = ncorr;
  // (the last iteration's cilhip_find_correspondences left the set it estimated from: what getCorrespondences() returns)
  if (it == 0) { drop_matches(c); c->have_pairs = false; } else c->matches_origin = 1;
  float ms = 0.f;
  CK(c, hipEventElapsedTime(&ms, e_beg, e_end));
  c->last_loop_ms = ms; c->last_search_ms = 0.0; c->last_acc_ms = 0.0; c->last_search_launches = 0;
  return CILHIP_OK;
}

int cilhip_set_pair_weight_callback(cilhip_ctx* c, cilhip_pair_weight_fn fn, void* user) {
  if (!c) return CILHIP_ERR_INVALID;
  c->weight_fn = fn; c->weight_user = fn ? user : nullptr;
  return CILHIP_OK;
}

void cilhip_icp_default_params(cilhip_icp_params* p) {
  if (!p) return;
  p->metric = CILHIP_METRIC_COMBINED;
  p->w_p2p = 0.0f; p->w_p2pl = 1.0f;
  p->max_iter = 15; p->conv_tol = 1e-5f;
  p->max_opt_iter = 1; p->opt_conv_tol = 1e-5f;
  p->max_sq_dist = 0.01f * 0.01f;
}

// What the accumulation kernels sum for one ICP instance.  A plane term without target normals is the reference's
// "dst_p.cols() != dst_n.cols()" case (transform_estimation.hpp:264-272: identity, false): the kernels must then never
// touch grid.nrm (it is null) -- they count the correspondences only (IM_POINT's slot 0) and the epilogue's identity
// branch (k_solve: has_p2pl && !has_normals) does the rest.
static int iter_metric_of(const cilhip_ctx* c, const cilhip_icp_params* p) {
  if (p->metric == CILHIP_METRIC_POINT_TO_POINT) return IM_KABSCH;
  const bool wp = p->w_p2p > 0.0f, wl = p->w_p2pl > 0.0f;
  if (wl && !c->has_normals) return IM_POINT;
  if (wp && wl) return IM_BOTH;
  if (wl) return IM_PLANE;
  if (wp) return IM_POINT;
  return IM_PLANE;  // no terms: sums unused, the epilogue takes the identity branch
}

static SolveArgs make_solve_args(cilhip_ctx* c, const cilhip_icp_params* p, int im, const float src_mean[3]) {
  SolveArgs sa{};
  sa.state = c->d_state;
  sa.partials = c->d_partials;
  sa.nblocks = iter_num_blocks(c->ns);
  sa.reduced = nullptr;
  sa.metric = im;
  sa.w_p2p = p->w_p2p; sa.w_p2pl = p->w_p2pl;
  if (p->metric == CILHIP_METRIC_COMBINED && weighted(c)) {   // the metric weights are inside the per-pair weights already
    sa.w_p2p = p->w_p2p > 0.0f ? 1.0f : 0.0f; sa.w_p2pl = p->w_p2pl > 0.0f ? 1.0f : 0.0f;
    sa.point_weighted = 1;
  }
  sa.conv_tol = p->conv_tol; sa.opt_conv_tol = p->opt_conv_tol;
  for (int i = 0; i < 3; ++i) { sa.dst_mean[i] = c->dst_mean[i]; sa.src_mean[i] = src_mean[i]; }
  sa.gn_last_step = 1;
  sa.has_normals = c->has_normals ? 1 : 0;
  sa.unproven_cnt = c->d_unproven;
  sa.guard_axis = c->guard_axis; sa.guard_slack = c->guard_slack;
  for (int i = 0; i < 3; ++i) { sa.guard_center[i] = c->guard_center[i]; sa.guard_half[i] = c->guard_half[i]; }
  for (int i = 0; i < 16; ++i) sa.guard_T[i] = c->guard_T[i];
  for (int i = 0; i < 3; ++i) { sa.src_center[i] = c->src_center[i]; sa.src_half[i] = c->src_half[i]; }
  sa.trace = c->d_trace;
  return sa;
}

static hipEvent_t get_acc_event(cilhip_ctx* c, size_t i) {
  while (c->ev_acc.size() <= i) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_acc.push_back(e); }
  return c->ev_acc[i];
}

static hipEvent_t get_event(cilhip_ctx* c, size_t i) {
  while (c->ev.size() <= i) { hipEvent_t e; (void)hipEventCreate(&e); c->ev.push_back(e); }
  return c->ev[i];
}

// What the run's epilogues have published (Feedback): a consistent snapshot of the LATEST published iteration.
struct FbView { bool done; unsigned int iterations, unproven, listed; float delta, prev_delta, step; };
// Waits until iteration `need` of the current run (or its convergence) has been published.  patience_s: how long to spin;
// returns 0 and fills *v, or 1 when nothing came in that time.
static int wait_published(cilhip_ctx* c, unsigned int need, double patience_s, FbView* v) {
  const volatile Feedback* fb = c->h_feedback;
  const auto t0 = std::chrono::steady_clock::now();
  struct Acc { cilhip_ctx* c; std::chrono::steady_clock::time_point t; ~Acc() { c->wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); } } acc{c, t0};
  for (unsigned spins = 0;; ++spins) {
    const unsigned long long lt = fb->latest;
    if ((unsigned int)(lt >> 32) == c->run_tag) {
      const bool done = (lt & 0x80000000ull) != 0ull;
      const unsigned int iters = (unsigned int)lt & 0x7fffffffu;
      if (done || iters >= need) {
        // the slot of the latest published iteration: the device's next write goes to another slot (the host is at most two
        // iterations ahead), so this read cannot be torn; its commit word is checked all the same
        const volatile FeedbackSlot* sl = &fb->slot[iters & 3u];
        v->done = done; v->iterations = iters;
        v->unproven = sl->unproven; v->listed = sl->listed; v->delta = sl->delta; v->prev_delta = sl->prev_delta; v->step = iters ? sl->step : INFINITY;
        if (iters == 0u || sl->commit == (((unsigned long long)c->run_tag << 32) | iters)) return 0;
      }
    }
    cpu_relax(spins);
    if ((spins & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > patience_s) return 1;
  }
}
// ... with the long-stall handling of cilhip_icp_run: nothing for 30 s -- a caller-owned stream may have long work of its own
// queued ahead of this run -- wait for the stream (that also surfaces a device fault); everything enqueued has then run and
// must have been published
static int wait_published_or_sync(cilhip_ctx* c, unsigned int need, FbView* v) {
  if (wait_published(c, need, 30.0, v) == 0) return CILHIP_OK;
  CK(c, hipStreamSynchronize(c->stream));
  if (wait_published(c, need, 0.01, v) == 0) return CILHIP_OK;
  return fail(c, CILHIP_ERR_HIP, "icp_run: the device stopped publishing its loop state");
}

static int read_state(cilhip_ctx* c, cilhip_icp_result* out, float* Tprev = nullptr) {
  IcpState hs;
  CK(c, hipMemcpyAsync(&hs, c->d_state, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  memcpy(c->tie_counters_host, hs.tie_counters, sizeof(c->tie_counters_host));
  c->tie_counters_fresh = true;
  memcpy(out->T, hs.T, sizeof(hs.T));
  if (Tprev) memcpy(Tprev, hs.Tprev, sizeof(hs.Tprev));
  out->iterations = (size_t)hs.iterations;
  out->last_delta_norm = hs.delta;
  out->last_ncorr = (size_t)hs.ncorr;
  return CILHIP_OK;
}

// What the engine's getCorrespondences() refers to after a run: the set of the last executed iteration, found under Tprev
// (correspondence_search_kd_tree.hpp:231 keeps it; icp_base.hpp:32-38 hands the engine out).  stored: the loop's kernels left
// it in nn_pos (the squared distances are formed again with the search's pinned arithmetic); pairs: c->pairs holds it;
// otherwise it is searched again when somebody asks (materialize_pending).
static void finish_run_matches(cilhip_ctx* c, const cilhip_icp_params* p, size_t iterations, const float Tprev[16], bool stored, bool pairs) {
  drop_matches(c); c->have_pairs = false;
  if (iterations == 0) return;
  memcpy(c->nn_T, Tprev, sizeof(c->nn_T));
  if (pairs) { c->have_pairs = true; c->matches_origin = 1; return; }
  if (stored && c->ns) {
    // (the squared distances of the stored matches are formed when somebody asks for them -- ensure_d2: a pass over the source that
    //  a caller who only wants the transform does not pay, 80 us at 10M)
    c->have_nn = true; c->d2_stale = true; c->matches_origin = 1;
  } else {
    c->pending_matches = true; c->pending_max_sq = p->max_sq_dist; c->matches_origin = 2;
  }
}

static int icp_run_once(cilhip_ctx* c, const cilhip_icp_params* p, const float* T0, cilhip_icp_result* out);
int cilhip_icp_run(cilhip_ctx* c, const cilhip_icp_params* p, const float* T0, cilhip_icp_result* out) {
  if (!c || !p || !out) return CILHIP_ERR_INVALID;
  if (p->metric != CILHIP_METRIC_POINT_TO_POINT && p->metric != CILHIP_METRIC_COMBINED) return fail(c, CILHIP_ERR_INVALID, "icp_run: bad metric");
  CK(c, hipSetDevice(c->device));
  int rc = tie_prepare(c, "icp_run");
  if (rc) return rc;
  rc = icp_run_once(c, p, T0, out);
  if (rc) return rc;
  // tie_rule 2: some search of the run met exactly equidistant nearest points and the reference's order tables were not there: they
  // are now (built once per target) -- the run is executed again, from T0, with the ties resolved inside its kernels
  bool again = false;
  rc = tie_check_pending(c, &again);
  if (rc) return rc;
  return again ? icp_run_once(c, p, T0, out) : CILHIP_OK;
}
static int icp_run_once(cilhip_ctx* c, const cilhip_icp_params* p, const float* T0, cilhip_icp_result* out) {
  if (c->weight_fn && p->metric == CILHIP_METRIC_COMBINED && c->transform_mode == 0) {
    // a caller's own weight evaluators run on the host: the reference's loop step by step (search, estimate over the stored set with
    // the callback's weights, rotation() polish + compose), the combiner's loop with one engine in both roles
    if (c->search_dir != 0) return fail(c, CILHIP_ERR_UNSUPPORTED, "a pair-weight callback runs with SECOND_TO_FIRST searches (rigid loop); estimate from pair lists through cilhip_estimate_combined");
    c->last_loop_ms = 0.0; c->last_search_ms = 0.0; c->last_acc_ms = 0.0; c->last_search_launches = 0;
    return cilhip_icp_run_two_sets(c, p->max_sq_dist, c, p->max_sq_dist, p, T0, out);
  }
  // The affine classes: their loop runs device-resident like the rigid one -- search-only kernels + one streaming pass of moments
  // (k_acc_affine) while the source is far from alignment, search + moments in the warm-started kernel afterwards, the 12-unknown solve
  // and the f32 compose in k_solve_affine -- unless something asks for the stored set per iteration (post-filters, per-pair weights,
  // other directions, feature adaptors): those keep the host-driven loop (icp_run_affine: three passes + a host solve per iteration).
  const bool affine = c->transform_mode == 1;
  if (affine) {
    if (c->index_offset) return fail(c, CILHIP_ERR_UNSUPPORTED, "the affine variants are not available on target shards");
    const bool device_loop = c->affine_device_loop && c->ns != 0 && c->grid.n != 0 && c->search_dir == 0 && !filters_active(c) && !weighted(c) && !feat6(c) && !c->fused &&
                             !(c->d_src_nrm && c->symmetric) && c->guard_axis < 0;
    if (!device_loop) return icp_run_affine(c, p, T0, out);
  }
  const float* Ti = T0 ? T0 : kIdentity;
  int rc = ensure_sorted(c, Ti);
  if (rc) return rc;
  const bool affine_combined = affine && p->metric == CILHIP_METRIC_COMBINED;
  const int im = !affine ? iter_metric_of(c, p) : (affine_combined && p->w_p2pl > 0.0f && c->has_normals) ? IM_AFFC : IM_AFFP;
  const bool gn = (im != IM_KABSCH) && !affine;
  // max_optimization_iterations == 0 (combined metric): the estimator's loop body never runs -- one accumulation pass still counts
  // the correspondences (the "no usable terms" test, transform_estimation.hpp:264-272), the epilogue skips the solve
  const bool zero_steps = gn && p->max_opt_iter == 0;
  const size_t opt_steps = gn ? (p->max_opt_iter ? p->max_opt_iter : 1) : 1;
  ++c->run_tag;
  launch_init_state(c->d_state, Ti, c->src_mean, c->stream, c->d_feedback, c->run_tag, c->src_center, c->src_half, c->d_tie_counters);
  if (gn && c->ns >= 65536) ensure_pair_records(c);
  IterArgs a = make_iter_args(c, p->max_sq_dist);
  if (!c->pair_records) a.grid.pn = nullptr;
  a.cw = corr_weights_of(c, p);
  SolveArgs sa = make_solve_args(c, p, im, c->src_mean);
  sa.feedback = c->d_feedback; sa.run_tag = c->run_tag;
  sa.gn_zero_steps = zero_steps ? 1 : 0;
  const int nb = sa.nblocks;
  const int nb_aff = affine ? affine_acc_blocks(c->ns) : 0;
  if (affine) {
    a.no_centering = affine_combined ? 0 : 1;
    sa.affine_centered = affine_combined ? 1 : 0;
    if (!affine_combined) { sa.w_p2p = 1.0f; sa.w_p2pl = 0.0f; }      // the point-to-point class: unit point terms of the raw coordinates
    // rows of AFF_ROW doubles: the streaming pass's or the warm-started kernel's
    const size_t rows = (size_t)std::max(nb_aff, warm_num_blocks(c->ns));
    const size_t need = (rows * AFF_ROW + SUMS_MAX - 1) / SUMS_MAX;
    if (need > (size_t)c->partial_blocks) {
      if (c->d_partials) (void)hipFree(c->d_partials);
      c->d_partials = nullptr; c->partial_blocks = 0;
      CK(c, hipMalloc(&c->d_partials, need * SUMS_MAX * sizeof(double)));
      c->partial_blocks = (int)need;
      a.partials = c->d_partials; a.tile_partials = c->d_partials;
      sa.partials = c->d_partials;
    }
  }
  if (c->ns == 0) {  // no source points: the epilogue runs on all-zero sums (identity step)
    CK(c, hipMemsetAsync(c->d_sums, 0, SUMS_MAX * sizeof(double), c->stream));
    sa.nblocks = 0;
    sa.reduced = c->d_sums;
  }
  if (c->search_dir != 0) {
    // FIRST_TO_SECOND / BOTH: the correspondence set is a pair list rebuilt every iteration (a grid over the transformed
    // source, like the reference's per-iteration kd-tree); host-driven loop, the accumulation kernels stream over the pairs
    if (c->index_offset) return fail(c, CILHIP_ERR_UNSUPPORTED, "search directions other than SECOND_TO_FIRST are not available on target shards");
    if (feat6(c)) { rc = ensure_feature_arrays(c); if (rc) return rc; a.feat = feat_spec_of(c); }
    // (a weight evaluator over FEATURE distances reads them per pair: those loops go through the pair list below)
    const bool feat_weights = feat6(c) && a.cw.enabled;
    hipEvent_t e_beg = get_event(c, 0), e_end = get_event(c, 1);
    CK(c, hipEventRecord(e_beg, c->stream));
    // Without post-filters the loop needs the SUMS over the correspondence set, not the sorted list: the reverse matches are
    // found through the inverse of the (rigid) transform against a grid over the source built once, and accumulated where
    // they are found (BOTH: forward pass + the reverse matches that are not reciprocal duplicates; reciprocal: the
    // duplicates alone) -- no per-iteration index, no sort, no host round trip: every iteration is enqueued back to back.
    bool t0_rigid = true;
    for (int i = 0; i < 3 && t0_rigid; ++i)
      for (int j = 0; j < 3; ++j) {
        const double dot = (double)Ti[i * 4] * Ti[j * 4] + (double)Ti[i * 4 + 1] * Ti[j * 4 + 1] + (double)Ti[i * 4 + 2] * Ti[j * 4 + 2];
        if (std::fabs(dot - (i == j ? 1.0 : 0.0)) > 1e-5) t0_rigid = false;
      }
    if (!filters_active(c) && !(c->d_src_nrm && c->symmetric) && t0_rigid && c->ns && c->grid.n && !feat_weights && !(c->rev_tie_aware && tie_mode_on(c))) {
      rc = ensure_reverse_buffers(c);
      if (rc) return rc;
      FeatSpec rf = a.feat;
      rf.src = c->src_grid.nrm;
      if (rf.dst2) rf.src2 = c->d_src_rgb_grid;
      if (feat6(c) && (!rf.src || !rf.dst || (rf.dst2 && !rf.src2))) return fail(c, CILHIP_ERR_INVALID, "feature search: both clouds' feature vectors are needed");
      const int nb_f = iter_num_blocks(c->ns), nb_r = iter_num_blocks(c->grid.n);
      // BOTH: the forward half runs warm-started from its third iteration on (search + accumulation in k_warm, like the plain loop's
      // steady state: exact whatever the source's distance, and these loops have no cheaper forward form to go back to)
      const bool fwd_wcap = c->search_dir == 2 && warm_capable(c);
      if (fwd_wcap) { rc = ensure_safe2(c); if (rc) return rc; rc = ensure_warm_buffers(c); if (rc) return rc; }
      const int nb_w = fwd_wcap ? warm_num_blocks(c->ns) : 0;
      const int nb_fmax = std::max(nb_f, nb_w);
      // the warm-started reverse search accumulates the first step's sums itself (one pass over the target; per-pair weights keep the separate pass)
      const bool rev_fusable = c->reverse_warm && !feat6(c) && !a.cw.enabled && c->d_src_safe2 != nullptr;
      const int nb_rw = rev_fusable ? reverse_warm_blocks(c->grid.n) : 0;
      const int nb_rmax = std::max(nb_r, nb_rw);
      if (nb_fmax + nb_rmax > c->partial_blocks) {
        if (c->d_partials) (void)hipFree(c->d_partials);
        c->d_partials = nullptr; c->partial_blocks = 0;
        CK(c, hipMalloc(&c->d_partials, (size_t)(nb_fmax + nb_rmax) * SUMS_MAX * sizeof(double)));
        c->partial_blocks = nb_fmax + nb_rmax;
      }
      const bool both_union = c->search_dir == 2 && !c->reciprocal;
      const int rmode = c->search_dir == 1 ? 1 : (c->reciprocal ? 3 : 2);
      // rows: the reverse matches' first, the forward half's (streaming pass or warm-started kernel) right behind them
      IterArgs ar = a;
      ar.partials = c->d_partials;
      a.nn_d2 = nullptr;
      c->rec_valid = false; c->lb_fresh = false;
      warm_run_reset(c);
      c->last_fused_iters = c->last_two_pass_iters = c->last_warm_iters = 0;
      for (size_t it = 0; it < p->max_iter; ++it) {
        bool fwd_warm = false;
        const bool rev_fused = rev_fusable && it >= 1;      // (this iteration's reverse search starts from the previous matches and accumulates)
        for (size_t st = 0; st < opt_steps; ++st) {
          a.skip_if_inner_done = ar.skip_if_inner_done = (st > 0);
          const int rev_rows = (rev_fused && st == 0) ? nb_rw : nb_r;
          a.partials = c->d_partials + (size_t)rev_rows * SUMS_MAX; a.tile_partials = a.partials;
          if (st == 0) {
            if (c->search_dir == 2) {
              fwd_warm = fwd_wcap && it >= 2;
              if (fwd_warm) {
                IterArgs wa = a;
                wa.warm_pos = c->d_nn_pos;
                wa.safe2 = c->d_safe2;
                wa.warm_far_sq = 0.25f * c->grid.cell * c->grid.cell;
                set_warm_args(c, wa);
                wa.nn_lb = c->d_nn_lb; wa.lb_valid = c->lb_fresh ? 1 : 0;
                launch_warm(wa, im, c->rec_valid ? 2 : 1, nb_w, c->stream);
                c->rec_valid = true; c->lb_fresh = false;
                ++c->last_warm_iters;
              } else {
                IterArgs sa2 = a;
                if (fwd_wcap) { sa2.nn_lb = c->d_nn_lb; c->lb_fresh = true; }      // (the margin keys the first warm-started iteration starts from)
                c->rec_valid = false;
                const int src_rc = launch_search(c, sa2);
                if (src_rc) return src_rc;
              }
            }
            // (from the second iteration on d_rev_pos holds the previous reverse matches: the search starts from them)
            const float* warm_tab = (it >= 1 && c->reverse_warm && !feat6(c)) ? c->d_src_safe2 : nullptr;
            RevFused rfu{};
            rfu.metric = im; rfu.mode = rmode; rfu.fwd_pos = c->d_nn_pos; rfu.src_inv = c->d_src_inv; rfu.grid_to_sorted = c->d_grid_to_sorted; rfu.partials = c->d_partials;
            for (int k = 0; k < 3; ++k) rfu.dst_mean[k] = a.dst_mean[k];
            { const TieDev rt = tie_dev_rev(c); launch_reverse_search_rigid(c->grid, c->src_grid, c->d_state, p->max_sq_dist, c->d_rev_pos, c->d_rev_d2, c->stream, feat6(c) ? &rf : nullptr, &rt, warm_tab, rev_fused ? &rfu : nullptr); }
          }
          const bool fwd_in_kernel = fwd_warm && st == 0;      // (the warm-started kernel accumulated the first step's terms itself)
          if (both_union && !fwd_in_kernel) launch_iter(a, im, false, false, nb_f, c->stream);
          if (!(rev_fused && st == 0)) launch_acc_reverse(ar, im, c->src_grid.pts, c->d_rev_pos, c->grid.n, rmode, c->d_nn_pos, c->d_src_inv, nb_r, c->stream);
          sa.gn_last_step = (st + 1 == opt_steps);
          const int rows_total = rev_rows + (both_union ? (fwd_in_kernel ? nb_w : nb_f) : 0);
          const int rows = launch_reduce_stage1(c->d_partials, rows_total, c->d_stage, c->stream);
          sa.partials = rows ? c->d_stage : c->d_partials;
          sa.nblocks = rows ? rows : rows_total;
          sa.reduced = nullptr;
          launch_solve(sa, c->stream);
        }
        if (p->max_iter > 64 && (it + 1) % 32 == 0 && it + 1 < p->max_iter) {     // (long runs: stop enqueueing once converged)
          int done = 0;
          CK(c, hipMemcpyAsync(&done, reinterpret_cast<const char*>(c->d_state) + offsetof(IcpState, done), sizeof(int), hipMemcpyDeviceToHost, c->stream));
          CK(c, hipStreamSynchronize(c->stream));
          if (done) break;
        }
      }
      CK(c, hipEventRecord(e_end, c->stream));
      CK(c, hipGetLastError());
      float Tprev[16];
      rc = read_state(c, out, Tprev);
      if (rc) return rc;
      finish_run_matches(c, p, out->iterations, Tprev, false, false);      // (nothing was listed: searched again on demand)
      float ms = 0.f;
      CK(c, hipEventElapsedTime(&ms, e_beg, e_end));
      c->last_loop_ms = ms; c->last_search_ms = 0.0; c->last_acc_ms = 0.0; c->last_search_launches = 0;
      return CILHIP_OK;
    }
    for (size_t it = 0; it < p->max_iter; ++it) {
      rc = run_pair_search(c, a, p->max_sq_dist, it == 0 ? Ti : out->T);
      if (rc) return rc;
      IterArgs pa = a;
      pa.nn_d2 = c->pairs.d2;  // (per pair: corr.value -- the 6-D distance under a feature adaptor -- for the weight evaluators)
      pa.src = c->pairs.src_view; pa.src_nrm = (c->d_src_nrm && c->symmetric) ? c->pairs.nrm_view : nullptr; pa.ns = c->pairs.count; pa.nn_pos = c->pairs.posd;
      const int pnb = iter_num_blocks(pa.ns);
      if (pnb > c->partial_blocks) {
        if (c->d_partials) (void)hipFree(c->d_partials);
        c->d_partials = nullptr; c->partial_blocks = 0;
        CK(c, hipMalloc(&c->d_partials, (size_t)pnb * SUMS_MAX * sizeof(double)));
        c->partial_blocks = pnb;
      }
      pa.partials = c->d_partials;
      for (size_t st = 0; st < opt_steps; ++st) {
        pa.skip_if_inner_done = (st > 0);
        sa.gn_last_step = (st + 1 == opt_steps);
        if (pa.ns) {
          launch_iter(pa, im, false, false, pnb, c->stream);
          const int rows = launch_reduce_stage1(c->d_partials, pnb, c->d_stage, c->stream);
          sa.partials = rows ? c->d_stage : c->d_partials;
          sa.nblocks = rows ? rows : pnb;
          sa.reduced = nullptr;
        } else {
          CK(c, hipMemsetAsync(c->d_sums, 0, SUMS_MAX * sizeof(double), c->stream));
          sa.nblocks = 0;
          sa.reduced = c->d_sums;
        }
        launch_solve(sa, c->stream);
      }
      rc = read_state(c, out);
      if (rc) return rc;
      if (out->last_delta_norm < p->conv_tol) break;   // the device sets `done` by the same test (icp_base.hpp:83)
    }
    CK(c, hipEventRecord(e_end, c->stream));
    CK(c, hipGetLastError());
    float Tprev[16];
    rc = read_state(c, out, Tprev);
    if (rc) return rc;
    finish_run_matches(c, p, out->iterations, Tprev, false, out->iterations > 0);      // c->pairs: the last iteration's list
    float ms = 0.f;
    CK(c, hipEventElapsedTime(&ms, e_beg, e_end));
    c->last_loop_ms = ms; c->last_search_ms = 0.0; c->last_acc_ms = 0.0; c->last_search_launches = 0;
    return CILHIP_OK;
  }
  if (feat6(c)) { rc = ensure_feature_arrays(c); if (rc) return rc; a.feat = feat_spec_of(c); }
  if (!filters_active(c) && !(a.cw.enabled && feat6(c))) a.nn_d2 = nullptr;   // nobody reads the distances inside the loop: 4 B per query less to write
                                                                              // (a weight evaluator over the 6-D feature distance does)
  const bool tile_acc = tile_accumulation(c) && !affine;      // (the tiles accumulate the rigid classes' terms only)
  const bool timing = c->kernel_timing && p->max_iter <= 4096;
  hipEvent_t e_beg = get_event(c, 0), e_end = get_event(c, 1);
  CK(c, hipEventRecord(e_beg, c->stream));
  size_t nev = 2, nacc = 0;
  int launches = 0;
  // Tiled runs are PACED: the host stays at most two iterations ahead of the device and looks at the loop state of
  // iteration it - 2 before it enqueues iteration it (a pinned copy + an event per iteration; the device never waits: an
  // iteration takes hundreds of microseconds, the look a few).  That buys (1) no launches after convergence and (2) the
  // choice of the kernel FORM per iteration: while the octant stage leaves many queries unproven (source far from
  // alignment: first iterations of a registration) the search runs with its in-LDS 3x3x3 second pass and a separate
  // streaming accumulation; once nearly all are proven, search + accumulation run as one pass inside the tiles.
  const bool wcap = warm_capable(c);
  if (wcap) { rc = ensure_safe2(c); if (rc) return rc; rc = ensure_warm_buffers(c); if (rc) return rc; }
  // the feature adaptors' searches warm-started from the previous matches (feat_warm.hip): once the published step is within reach,
  // while the kernel's own count of the queries it had to search says that it pays
  // (from 400 000 source points up: below, the look at the published state before every enqueue costs what the form saves -- measured 200k: +5 %, 1M: -23 %)
  const bool fwcap = feat6(c) && c->feat_warm && c->warm_start != 0 && c->ns >= 400000 && !filters_active(c) && !c->fused && !affine;
  if (fwcap) {
    rc = ensure_safe2(c); if (rc) return rc;
    const int rows = feat_warm_blocks(c->ns);
    if (rows > c->partial_blocks) {
      if (c->d_partials) (void)hipFree(c->d_partials);
      c->d_partials = nullptr; c->partial_blocks = 0;
      CK(c, hipMalloc(&c->d_partials, (size_t)rows * SUMS_MAX * sizeof(double)));
      c->partial_blocks = rows;
      a.partials = c->d_partials; a.tile_partials = c->d_partials; sa.partials = c->d_partials;
    }
  }
  bool feat_warm_now = false;
  unsigned int feat_judged = 0;
  const bool paced = ((tile_acc || wcap) && c->ns && p->max_iter > 2 && c->tile_acc_adaptive) || (fwcap && p->max_iter > 2);
  if (tile_acc && !c->tile_acc_adaptive) c->far_mode = false;
  c->last_fused_iters = c->last_two_pass_iters = c->last_warm_iters = 0;
  c->rec_valid = false; c->lb_fresh = false;
  warm_run_reset(c);
  c->iter_form.clear(); c->trace_form.clear(); c->timed_iter.clear();
  for (int k = 0; k < 5; ++k) { c->form_ms[k] = 0.0; c->form_n[k] = 0; }
  // The cooperative search (several lanes per query) for the cold iterations of clouds the tiles do not take: lanes so that the
  // queries fill the machine; below the warm-started form's floor always (nothing is lost: no margin keys are wanted there), above it
  // while the cold kernels' forecast says that most queries are far from settled (their margins would not survive the next step) and
  // the loop is not yet within reach of the warm-started form -- whose entry needs the keys only the one-lane search leaves.
  const int glanes = c->group_lanes > 0 ? c->group_lanes
                     : (c->group_lanes < 0 && !use_tiled(c) && !feat6(c) && !c->fused) ? (c->ns <= 400000u ? 16 : c->ns <= 1500000u ? 8 : 0) : 0;
  bool group_now = glanes != 0 && (c->group_lanes > 0 || !wcap);
  size_t next_probe = 0, probe_gap = 8;
  bool warm_on = false;       // the loop has been seen to move little: iterations run warm-started until one of them has to search too many of its queries
  unsigned int judged = 0;    // the last published iteration whose listed count has been judged
  bool all_stored = true;     // every iteration enqueued left its matches in nn_pos (finish_run_matches)
  bool prev_stored = false;   // ... the previous one did
  for (size_t it = 0; it < p->max_iter; ++it) {
    if (paced && it == 1 && wcap && c->warm_start == 1 && !c->warm_banned && !c->trace_form.empty() && (c->trace_form[0] & 0x80)) {
      // The SECOND iteration can already run warm-started when the first one moved the source by a small fraction of a cell (a
      // source that starts aligned: tracking, a refinement pass) and its kernels' own forecast agrees: worth one look at the
      // first iteration's result before the second is enqueued (the device idles for the host's reaction once per run; a cold
      // iteration costs three times a warm one).
      FbView fv;
      rc = wait_published_or_sync(c, 1u, &fv);
      if (rc) return rc;
      if (fv.done) break;
      if ((c->trace_form[0] & 0x7f) <= FORM_TILE_ONE_PASS) c->far_mode = (unsigned long long)fv.unproven * 16ull > (unsigned long long)c->ns;
      const bool forecast_ok = !c->warm_forecast || (unsigned long long)fv.listed * 8ull <= (unsigned long long)c->ns;
      if (glanes && c->group_lanes < 0 && fv.iterations == 1u) group_now = (unsigned long long)fv.listed * 2ull > (unsigned long long)c->ns;
      warm_on = fv.iterations == 1u && forecast_ok && !group_now && warm_worthwhile(c, fv.step);
    }
    if (paced && it >= 2) {
      // wait (briefly, if at all) until iteration it - 2 has been published
      FbView fv;
      rc = wait_published_or_sync(c, (unsigned int)(it - 1), &fv);
      if (rc) return rc;
      if (fv.done) break;
      if (fwcap && !c->warm_banned) {
        // (a warm-started feature search reports the queries it had to search in full: more than a quarter of them = a cold tile search's price)
        const bool was_fw = fv.iterations >= 1 && fv.iterations <= c->trace_form.size() && (c->trace_form[fv.iterations - 1] & 0x7f) == FORM_WARM;
        if (was_fw && fv.iterations > feat_judged) { feat_judged = fv.iterations; if (!warm_keeps_paying(c, fv.listed)) feat_warm_now = false; }
        else if (!feat_warm_now) feat_warm_now = warm_worthwhile(c, fv.step);
      }
      // the form of the COLD iterations (one pass / two passes), from the last cold iteration's count of queries its octant stage
      // left open (a warm-started iteration counts something else there: the queries its own search took to the shells)
      if (fv.iterations >= 1 && fv.iterations <= c->trace_form.size() && (c->trace_form[fv.iterations - 1] & 0x7f) <= FORM_TILE_ONE_PASS)
        c->far_mode = (unsigned long long)fv.unproven * 16ull > (unsigned long long)c->ns;
      // what the published iteration's `listed` count means: a warm-started iteration reports the queries it had to search, a
      // cold iteration whose kernels leave margins (bit 7 of its form) the queries a warm-started iteration after it would have to
      auto form_of = [&](const FbView& f) -> int { return (f.iterations >= 1 && f.iterations <= c->trace_form.size()) ? (int)c->trace_form[f.iterations - 1] : -1; };
      auto is_warm = [&](const FbView& f) { const int fo = form_of(f); return fo >= 0 && ((fo & 0x7f) == FORM_WARM || (fo & 0x7f) == FORM_WARM_FIRST); };
      if (glanes && c->group_lanes < 0 && wcap) {
        // a cold iteration that counted (bit 7 of its form): its forecast decides (every eighth iteration of a stretch of cooperative
        // searches is such a one: below)
        const int fo = form_of(fv);
        if (fo >= 0 && (fo & 0x80)) group_now = (unsigned long long)fv.listed * 2ull > (unsigned long long)c->ns;
      }
      // (a published iteration is judged once: the same one can be the latest at two consecutive looks)
      bool fell = false;
      if (wcap && warm_on && c->warm_start == 1 && fv.iterations > judged && is_warm(fv)) {
        judged = fv.iterations;
        if (!warm_keeps_paying(c, fv.listed)) { warm_on = false; fell = true; }
      }
      if (wcap && c->warm_start == 1 && !warm_on && !fell && !c->warm_banned && fv.step < 8.0f * c->warm_thresh) {
        // Candidate for the warm-started form (below).  Decided on the step the loop made LAST -- it is the distance between
        // the queries the margins were left for and the queries about to be searched -- so wait for iteration it - 1 itself
        // (a bubble of some tens of microseconds, only while this decision is pending and the loop is within reach of it).
        rc = wait_published_or_sync(c, (unsigned int)it, &fv);
        if (rc) return rc;
        if (fv.done) break;
        if (is_warm(fv)) {
          if (fv.iterations > judged && fv.listed != 0u) { judged = fv.iterations; fell = !warm_keeps_paying(c, fv.listed); }
          if (!fell && !c->warm_banned) warm_on = warm_worthwhile(c, fv.step);
        } else {
          // the cold iteration's own forecast: enter only if at most an eighth of the queries would have to be searched
          // (never out of a stretch of cooperative searches: they leave no keys; its next one-lane iteration's forecast ends the stretch first)
          const int fo = form_of(fv);
          const bool forecast_ok = !c->warm_forecast || !(fo >= 0 && (fo & 0x80)) || (unsigned long long)fv.listed * 8ull <= (unsigned long long)c->ns;
          warm_on = forecast_ok && !(glanes && c->group_lanes < 0 && group_now) && warm_worthwhile(c, fv.step);
        }
      }
    }
    const bool one_pass = tile_acc && !c->far_mode;
    // Third form, from the second iteration on: search + accumulation WARM-STARTED from the previous iteration's matches and
    // the margins their searches left (kept by the forms above) -- no tile to stage at all.  Same matches, same sums up to
    // the order of the f64 additions.
    const bool warm = wcap && it >= 1 && (c->warm_start == 2 || (paced && warm_on));
    const bool single = one_pass || warm;        // search + accumulation in one kernel
    bool warm_first = false;
    bool stored_now = true;    // this iteration leaves its matches in nn_pos
    bool counted = false;      // a cold iteration whose kernels count the queries a warm-started iteration after it would have to search
    bool feat_warm_it = false; // this iteration's feature search ran warm-started
    bool feat_fused_it = false; // ... and accumulated the first step's sums itself
    // (kernel timing on: does THIS iteration carry events?  Every event between dependent kernels idles the device for ~6 us --
    //  two per iteration are a tenth of a warm-started iteration at 10M -- so a caller may ask for a sample: option kernel_timing_stride)
    const bool timing_it = timing && (c->timing_stride <= 1 || it < 3 || it % (size_t)c->timing_stride == 0);
    for (size_t st = 0; st < opt_steps; ++st) {
      a.skip_if_inner_done = (st > 0);
      // (the one-kernel forms are timed through their own dispatch packets: no event packets between dependent kernels)
      const bool lane_fused = c->fused && !filters_active(c) && !feat6(c);
      const bool ext_ev = timing_it && st == 0 && c->ns && !lane_fused && (warm || one_pass);
      if (timing_it && st == 0 && !ext_ev) CK(c, hipEventRecord(get_event(c, nev++), c->stream));
      if (ext_ev) { hipEvent_t e0 = get_event(c, nev), e1 = get_event(c, nev + 1); set_launch_events(e0, e1); nev += 2; }
      if (c->ns) {
        if (st == 0 && c->fused && !filters_active(c) && !feat6(c)) {
          launch_iter(a, im, true, gn && opt_steps > 1, nb, c->stream);
          all_stored = all_stored && gn && opt_steps > 1;
          stored_now = gn && opt_steps > 1;
        } else if (st == 0 && warm) {
          IterArgs wa = a;
          wa.warm_pos = c->d_nn_pos;
          wa.safe2 = c->d_safe2;
          wa.warm_far_sq = 0.25f * c->grid.cell * c->grid.cell;
          // the first warm iteration after the search-only forms gathers through the stored positions, takes the margin keys
          // those searches left (nn_lb) and writes a match record per query; after a tile iteration with the accumulation
          // inside -- which writes the records itself -- and from then on, the records are streamed instead
          set_warm_args(c, wa);
          wa.nn_lb = c->d_nn_lb; wa.lb_valid = c->lb_fresh ? 1 : 0;
          warm_first = !c->rec_valid;
          launch_warm(wa, im, c->rec_valid ? 2 : 1, warm_num_blocks(c->ns), c->stream);
          c->rec_valid = true; c->lb_fresh = false;
        } else if (st == 0 && one_pass) {
          // search + accumulation of the first Gauss-Newton step inside the LDS tiles (one pass; the matches are only
          // stored when further Gauss-Newton steps will stream over them or the next iteration may start from them)
          IterArgs fa = a;
          fa.store_matches = (opt_steps > 1 || c->warm_start) ? 1 : 0;
          fa.partials = c->d_partials + (size_t)c->ntiles * SUMS_MAX;
          // ... and from the second iteration on the tile leaves the match records of the warm-started form (not the first: a
          // registration's first step is its largest, its margins would be spent at once)
          const bool recs = wcap && it >= 1 && c->tile_records && fa.store_matches;
          if (recs) set_warm_args(c, fa);
          launch_search_tiled(fa, im, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);
          c->rec_valid = recs; c->lb_fresh = false;
          counted = recs;
          all_stored = all_stored && fa.store_matches != 0;
          stored_now = fa.store_matches != 0;
        } else if (st == 0) {
          c->rec_valid = false;
          // (search-only form of the tiles: the margin keys of its searches next to the matches)
          IterArgs sa2 = a;
          // (above the warm-started form's floor a stretch of cooperative searches is interrupted by a one-lane search now and then -- after
          //  8 iterations, then 16, 32 ...: it leaves the margin keys and the forecast the loop's decisions, this form or that, the
          //  warm-started one, are taken from)
          const bool probe = c->group_lanes < 0 && wcap && it >= next_probe;
          if (probe) { next_probe = it + probe_gap; probe_gap *= 2; }
          const int lanes_it = (group_now && !probe && !use_tiled(c) && !feat6(c)) ? glanes : 0;
          const bool keys = wcap && !feat6(c) && !lanes_it;
          if (keys) sa2.nn_lb = c->d_nn_lb;
          c->lb_fresh = keys;
          counted = keys;
          // (the cooperative form: the previous iteration's matches, when it left them in nn_pos, bound every query's search)
          if (lanes_it && it >= 1 && prev_stored) sa2.warm_pos = c->d_nn_pos;
          feat_warm_it = fwcap && feat_warm_now && it >= 1 && prev_stored && !c->warm_banned;
          // (... with the sums in the same pass when the terms are the three-cloud metric's own: no source normals in the objective, no per-pair weights)
          feat_fused_it = feat_warm_it && !(c->d_src_nrm && c->symmetric) && !a.cw.enabled;
          if (feat_warm_it) { sa2.safe2 = c->d_safe2; sa2.partials = c->d_partials; launch_feat_warm(sa2, feat_fused_it ? im : (int)IM_NONE, c->stream); ++c->last_warm_iters; }
          else { const int src_rc = launch_search(c, sa2, lanes_it); if (src_rc) return src_rc; }
          { const int frc = apply_filters(c); if (frc) return frc; }
          if (timing_it) { CK(c, hipEventRecord(get_event(c, nev++), c->stream)); CK(c, hipEventRecord(get_acc_event(c, nacc++), c->stream)); }
          if (affine) launch_acc_affine(a, im, nb_aff, c->stream);            // streaming accumulation kernel
          else if (!feat_fused_it) launch_iter(a, im, false, false, nb, c->stream);
        } else {
          launch_iter(a, im, false, false, nb, c->stream);
        }
      }
      if (timing_it && st == 0) {
        // (two events per iteration around the search / one-pass kernels; a two-pass iteration adds a pair around its
        //  streaming accumulation, kept in a list of its own)
        if (ext_ev) {}
        else if (single || (c->fused && !filters_active(c) && !feat6(c)) || !c->ns) CK(c, hipEventRecord(get_event(c, nev++), c->stream));
        else CK(c, hipEventRecord(get_acc_event(c, nacc++), c->stream));
        ++launches; c->timed_iter.push_back((unsigned int)it);
      }
      if (st == 0) { if (single) ++c->last_fused_iters; else ++c->last_two_pass_iters; if (warm) ++c->last_warm_iters; }
      if (st == 0) {
        const unsigned char form = (unsigned char)(warm ? (warm_first ? FORM_WARM_FIRST : FORM_WARM) : feat_warm_it ? FORM_WARM : one_pass ? FORM_TILE_ONE_PASS
                                                   : (c->fused && !filters_active(c) && !feat6(c)) ? FORM_LANE_FUSED : FORM_SEARCH);
        if (timing_it) c->iter_form.push_back(form);
        c->trace_form.push_back((unsigned char)(form | (counted ? 0x80 : 0)));
      }
      sa.gn_last_step = (st + 1 == opt_steps);
      if (c->ns) {
        const int prows = (st == 0 && warm) ? warm_num_blocks(c->ns) : (st == 0 && one_pass) ? tiled_partial_rows(c->ntiles) : (st == 0 && feat_fused_it) ? feat_warm_blocks(c->ns)
                          : affine ? nb_aff : nb;
        if (affine) launch_reduce_and_solve_affine(c->d_partials, prows, c->d_stage, sa, c->stream);
        else launch_reduce_and_solve(c->d_partials, prows, c->d_stage, c->fused_epilogue ? c->d_ticket : nullptr, sa, c->stream);
      } else {
        launch_solve(sa, c->stream);
      }
    }
    prev_stored = stored_now && c->ns != 0;
    // Long runs ("iterate until converged" with a large max_iter): the kernels of a converged run return at once, but
    // the post-filter / reduction launches do not look at the flag, so look at it from the host now and then and stop
    // enqueueing.  Short runs (the reference's default is 15) stay free of host round trips.
    if (!paced && p->max_iter > 64 && (it + 1) % 32 == 0 && it + 1 < p->max_iter) {
      int done = 0;
      CK(c, hipMemcpyAsync(&done, reinterpret_cast<const char*>(c->d_state) + offsetof(IcpState, done), sizeof(int), hipMemcpyDeviceToHost, c->stream));
      CK(c, hipStreamSynchronize(c->stream));
      if (done) break;
    }
  }
  CK(c, hipEventRecord(e_end, c->stream));
  CK(c, hipGetLastError());
  float Tprev[16];
  rc = read_state(c, out, Tprev);
  if (rc) return rc;
  finish_run_matches(c, p, out->iterations, Tprev, all_stored && !filters_active(c) && !feat6(c), false);
  float ms = 0.f;
  CK(c, hipEventElapsedTime(&ms, e_beg, e_end));
#ifdef CILHIP_EXP_PHASE_CLOCKS
  cilhip::debug_dump_phase_clocks();
#endif
  c->last_loop_ms = ms;
  c->last_search_ms = 0.0; c->last_search_launches = 0;
  if (timing) {
    // only iterations that actually executed (not the early-exit launches after convergence)
    size_t executed = 0;
    while (executed < c->timed_iter.size() && executed < (size_t)launches && (size_t)c->timed_iter[executed] < out->iterations) ++executed;
    c->last_acc_ms = 0.0;
    c->timed_ms.assign(executed, 0.0f);
    for (size_t k = 0; k < executed; ++k) {
      float m = 0.f;
      CK(c, hipEventElapsedTime(&m, c->ev[2 + 2 * k], c->ev[3 + 2 * k]));
      c->timed_ms[k] = m;
      c->last_search_ms += m;
      if (k < c->iter_form.size()) { c->form_ms[c->iter_form[k]] += m; ++c->form_n[c->iter_form[k]]; }
    }
    for (size_t k = 0; k + 1 < nacc; k += 2) {      // (two-pass iterations; those enqueued past convergence measure ~0)
      float m = 0.f;
      CK(c, hipEventElapsedTime(&m, c->ev_acc[k], c->ev_acc[k + 1]));
      c->last_acc_ms += m;
    }
    c->last_search_launches = (int)executed;
  }
  return CILHIP_OK;
}

int cilhip_icp_begin(cilhip_ctx* c, const cilhip_icp_params* p, const float* T0, const float* gmean) {
  if (!c || !p) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  if (p->metric == CILHIP_METRIC_COMBINED && p->max_opt_iter != 1) return fail(c, CILHIP_ERR_UNSUPPORTED, "sharded runs support max_opt_iter == 1");
  if (filters_active(c)) return fail(c, CILHIP_ERR_UNSUPPORTED, "inlier_fraction / one_to_one are global filters: not available in sharded runs");
  if (c->weight_fn && p->metric == CILHIP_METRIC_COMBINED)
    return fail(c, CILHIP_ERR_UNSUPPORTED, "a pair-weight callback is evaluated on the host, per estimate: not available in sharded runs (the stock evaluators are)");
  { const int trc = tie_prepare(c, "icp_begin"); if (trc) return trc; }
  if (c->search_dir != 0) return fail(c, CILHIP_ERR_UNSUPPORTED, "search directions other than SECOND_TO_FIRST are not available in sharded runs");
  if (feat6(c) || c->transform_mode != 0) return fail(c, CILHIP_ERR_UNSUPPORTED, "point+normal features and the affine variants are not available in sharded runs");
  const float* Ti = T0 ? T0 : kIdentity;
  int rc = ensure_sorted(c, Ti);
  if (rc) return rc;
  c->run_prm = *p;
  for (int i = 0; i < 3; ++i) c->run_src_mean[i] = gmean ? gmean[i] : c->src_mean[i];
  ++c->run_tag;
  launch_init_state(c->d_state, Ti, c->run_src_mean, c->stream, c->d_feedback, c->run_tag, c->src_center, c->src_half, c->d_tie_counters);     // (the epilogue publishes the loop state: see cilhip_icp_partial_sums)
  CK(c, hipGetLastError());
  c->run_active = true;
  c->run_nev = 0; c->run_nar = 0; c->last_allreduce_ms = 0.0; c->last_allreduce_n = 0;
  c->run_enqueue_us = 0.0; c->run_enqueue_iters = 0;
  c->run_calls = 0;
  c->run_warm_on = false; c->run_judged = 0;
  c->rec_valid = false; c->lb_fresh = false;
  warm_run_reset(c);
  if (warm_capable(c) && !(c->d_src_nrm && c->symmetric)) { rc = ensure_safe2(c); if (rc) return rc; rc = ensure_warm_buffers(c); if (rc) return rc; }
  c->iter_form.clear(); c->trace_form.clear();
  for (int k = 0; k < 5; ++k) { c->form_ms[k] = 0.0; c->form_n[k] = 0; }
  c->last_fused_iters = c->last_two_pass_iters = c->last_warm_iters = 0;    // counted per cilhip_icp_partial_sums call (cilhip_get_last_run_forms)
  return CILHIP_OK;
}

// sums_dev != null: the 48 sums of this iteration's search + accumulation (cilhip_icp_partial_sums).  rows_dev != null instead: RANK_ROWS
// rows that still have to be folded -- the stage-1 reduction with a FIXED number of groups, whatever form the iteration took and
// however many blocks this rank has -- for the ranked loop, which all-reduces those (12 KB instead of 384 B: both latency-bound) and
// lets the epilogue fold them as it does in cilhip_icp_run: one kernel and one gap less per iteration.
constexpr int RANK_ROWS = 32;
static int partial_sums_core(cilhip_ctx* c, double* sums_dev, double* rows_dev) {
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  CK(c, hipSetDevice(c->device));
  const int im = iter_metric_of(c, &c->run_prm);
  IterArgs a = make_iter_args(c, c->run_prm.max_sq_dist);
  a.cw = corr_weights_of(c, &c->run_prm);
  const int nb = iter_num_blocks(c->ns);
  int prows = nb;
  unsigned char form_now = FORM_LANE_FUSED;      // (the form this iteration takes: what its published counts will mean)
  if (c->ns && c->grid.n) {      // (a shard without target points -- a slab beyond the target's extent -- has nothing to match: zero sums)
    if (c->fused) {
      launch_iter(a, im, true, false, nb, c->stream);
    } else {
      a.nn_d2 = nullptr;   // no post-filters in sharded runs: nobody reads the squared distances (as in cilhip_icp_run)
      const bool timing = c->kernel_timing && c->run_nev + 3 <= 3 * 4096 &&
                          (c->timing_stride <= 1 || c->run_calls < 3 || c->run_calls % c->timing_stride == 0);      // (a sample of the iterations: option kernel_timing_stride)
      const size_t e = 2 + c->run_nev;
      if (timing) CK(c, hipEventRecord(get_event(c, e), c->stream));
      // Warm-started form (see cilhip_icp_run): from the second call on, when the latest loop state this run's epilogues have
      // published (a bounded wait for iteration run_calls - 2) says the source is near alignment.  Ranks may differ in their choice: the sums are
      // the same up to the order of the f64 additions.
      bool warm = false;
      const bool wcap = warm_capable(c) && !(c->d_src_nrm && c->symmetric);      // (the sharded building blocks: the symmetric objective stays with the streaming pass)
      if (wcap && c->run_calls >= 1) {
        warm = c->warm_start == 2;
        if (!warm && c->run_calls >= 2) {
          // paced like cilhip_icp_run: at most two iterations ahead of the device (which never waits: an iteration takes
          // hundreds of microseconds), so that the loop state looked at is at least that of iteration run_calls - 2; a brief
          // wait at most (5 s without news: the cold form)
          FbView fv;
          if (wait_published(c, (unsigned int)(c->run_calls - 1), 5.0, &fv) == 0) {
            // (what a published iteration's counts mean depends on the form it ran in: as in cilhip_icp_run)
            auto form_of = [&](const FbView& f) -> int { return (f.iterations >= 1 && f.iterations <= c->trace_form.size()) ? (int)c->trace_form[f.iterations - 1] : -1; };
            auto is_warm = [&](const FbView& f) { const int fo = form_of(f); return fo >= 0 && ((fo & 0x7f) == FORM_WARM || (fo & 0x7f) == FORM_WARM_FIRST); };
            if (form_of(fv) >= 0 && (form_of(fv) & 0x7f) <= FORM_TILE_ONE_PASS) c->far_mode = (unsigned long long)fv.unproven * 16ull > (unsigned long long)c->ns;
            bool fell = false;
            if (c->run_warm_on && fv.iterations > c->run_judged && is_warm(fv)) {
              c->run_judged = fv.iterations;
              if (!warm_keeps_paying(c, fv.listed)) { c->run_warm_on = false; fell = true; }
            }
            if (!c->run_warm_on && !fell && !c->warm_banned && fv.step < 8.0f * c->warm_thresh) {
              // candidate for the warm-started form: decided on the step the loop made LAST -- wait for iteration run_calls - 1
              // itself (its epilogue has been enqueued by the caller's previous apply; a bubble of some tens of microseconds, only
              // while this decision is pending and the loop is within reach of it), then as cilhip_icp_run decides
              FbView f2;
              if (wait_published(c, (unsigned int)c->run_calls, 5.0, &f2) == 0) {
                fv = f2;
                if (is_warm(fv)) {
                  if (fv.iterations > c->run_judged && fv.listed != 0u) { c->run_judged = fv.iterations; fell = !warm_keeps_paying(c, fv.listed); }
                  if (!fell && !c->warm_banned) c->run_warm_on = warm_worthwhile(c, fv.step);
                } else {
                  const int fo = form_of(fv);
                  const bool forecast_ok = !c->warm_forecast || !(fo >= 0 && (fo & 0x80)) || (unsigned long long)fv.listed * 8ull <= (unsigned long long)c->ns;
                  c->run_warm_on = forecast_ok && warm_worthwhile(c, fv.step);
                }
              }
            }
            warm = c->run_warm_on;
          }
        }
      }
      if (warm) {
        IterArgs wa = a;
        wa.nn_pos = c->d_nn_pos;
        wa.warm_pos = c->d_nn_pos;
        wa.safe2 = c->d_safe2;
        wa.warm_far_sq = 0.25f * c->grid.cell * c->grid.cell;
        set_warm_args(c, wa);
        wa.nn_lb = c->d_nn_lb; wa.lb_valid = c->lb_fresh ? 1 : 0;
        if (timing) c->iter_form.push_back((unsigned char)(c->rec_valid ? FORM_WARM : FORM_WARM_FIRST));
        form_now = (unsigned char)(c->rec_valid ? FORM_WARM : FORM_WARM_FIRST);
        launch_warm(wa, im, c->rec_valid ? 2 : 1, warm_num_blocks(c->ns), c->stream);
        prows = warm_num_blocks(c->ns);
        c->rec_valid = true; c->lb_fresh = false;
        if (timing) CK(c, hipEventRecord(get_event(c, e + 1), c->stream));
        ++c->last_fused_iters; ++c->last_warm_iters;
      } else if (tile_accumulation(c)) {
        if (timing) c->iter_form.push_back((unsigned char)FORM_TILE_ONE_PASS);
        IterArgs fa = a;
        fa.store_matches = c->warm_start ? 1 : 0;
        fa.partials = c->d_partials + (size_t)c->ntiles * SUMS_MAX;
        const bool recs = wcap && c->run_calls >= 1 && c->tile_records && fa.store_matches;
        if (recs) set_warm_args(c, fa);
        launch_search_tiled(fa, im, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);
        c->rec_valid = recs; c->lb_fresh = false;
        form_now = (unsigned char)(FORM_TILE_ONE_PASS | (recs ? 0x80 : 0));
        if (timing) CK(c, hipEventRecord(get_event(c, e + 1), c->stream));
        prows = tiled_partial_rows(c->ntiles);
        ++c->last_fused_iters;
      } else {
        c->rec_valid = false;
        if (timing) c->iter_form.push_back((unsigned char)FORM_SEARCH);
        ++c->last_two_pass_iters;
        IterArgs sa2 = a;
        const bool keys = wcap;
        if (keys) sa2.nn_lb = c->d_nn_lb;
        c->lb_fresh = keys;
        form_now = (unsigned char)(FORM_SEARCH | (keys ? 0x80 : 0));
        if (use_tiled(c)) launch_search_tiled(sa2, IM_NONE, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);
        else launch_iter(sa2, IM_NONE, true, true, nb, c->stream);
        if (timing) CK(c, hipEventRecord(get_event(c, e + 1), c->stream));
        launch_iter(a, im, false, false, nb, c->stream);
      }
      if (timing) { CK(c, hipEventRecord(get_event(c, e + 2), c->stream)); c->run_nev += 3; }
    }
    if (sums_dev) launch_reduce_partials(c->d_partials, prows, c->d_stage, sums_dev, c->stream);
    else launch_reduce_stage1_groups(c->d_partials, prows, rows_dev, RANK_ROWS, c->stream);
  } else if (sums_dev) {
    CK(c, hipMemsetAsync(sums_dev, 0, SUMS_MAX * sizeof(double), c->stream));
  } else {
    CK(c, hipMemsetAsync(rows_dev, 0, (size_t)RANK_ROWS * SUMS_MAX * sizeof(double), c->stream));
  }
  c->trace_form.push_back(form_now);
  ++c->run_calls;
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

int cilhip_icp_partial_sums(cilhip_ctx* c, double* sums_dev) {
  if (!c || !sums_dev) return CILHIP_ERR_INVALID;
  return partial_sums_core(c, sums_dev, nullptr);
}

int cilhip_icp_apply_sums(cilhip_ctx* c, const double* sums_dev) {
  if (!c || !sums_dev) return CILHIP_ERR_INVALID;
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  CK(c, hipSetDevice(c->device));
  const int im = iter_metric_of(c, &c->run_prm);
  SolveArgs sa = make_solve_args(c, &c->run_prm, im, c->run_src_mean);
  sa.feedback = c->d_feedback; sa.run_tag = c->run_tag;
  sa.nblocks = 0;
  sa.reduced = sums_dev;
  launch_solve(sa, c->stream);
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

int cilhip_set_shard_info(cilhip_ctx* c, uint64_t target_index_offset, const float* dst_mean, const float* src_mean) {
  if (!c) return CILHIP_ERR_INVALID;
  if (target_index_offset + (c->has_target ? c->grid.n : 0) > 0xFFFFFFFFull) return fail(c, CILHIP_ERR_INVALID, "global target indices must fit 32 bits");
  c->index_offset = (uint32_t)target_index_offset;
  c->partial_target = target_index_offset != 0 || dst_mean != nullptr;      // (the whole cloud's mean handed in: this target is a part of it)
  if (dst_mean) memcpy(c->dst_mean, dst_mean, sizeof(c->dst_mean));
  if (src_mean) memcpy(c->src_mean, src_mean, sizeof(c->src_mean));
  return CILHIP_OK;
}

int cilhip_set_slab_guard(cilhip_ctx* c, int axis, float slack, const float center[3], const float half_extent[3], const float T_part[16]) {
  if (!c) return CILHIP_ERR_INVALID;
  if (axis < 0) { c->guard_axis = -1; return CILHIP_OK; }
  if (axis > 2 || !center || !half_extent || !T_part || !(slack >= 0.0f)) return fail(c, CILHIP_ERR_INVALID, "set_slab_guard: axis 0..2, slack >= 0, box and transform required");
  c->guard_axis = axis; c->guard_slack = slack;
  memcpy(c->guard_center, center, sizeof(c->guard_center)); memcpy(c->guard_half, half_extent, sizeof(c->guard_half));
  memcpy(c->guard_T, T_part, sizeof(c->guard_T));
  return CILHIP_OK;
}

int cilhip_get_slab_violation(cilhip_ctx* c, int* out) {
  if (!c || !out) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  int v = 0;
  CK(c, hipMemcpyAsync(&v, reinterpret_cast<const char*>(c->d_state) + offsetof(IcpState, slab_violation), sizeof(int), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  *out = v;
  return CILHIP_OK;
}

int cilhip_get_slab_violation_state(cilhip_ctx* c, int* violated, cilhip_icp_result* at) {
  if (!c || !violated) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  IcpState hs;
  CK(c, hipMemcpyAsync(&hs, c->d_state, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  CK(c, hipStreamSynchronize(c->stream));
  *violated = hs.slab_violation;
  if (at) {
    memcpy(at->T, hs.slab_violation ? hs.violation_T : hs.T, sizeof(hs.T));
    at->iterations = (size_t)(hs.slab_violation ? hs.violation_iter : hs.iterations);
    at->last_delta_norm = hs.slab_violation ? hs.violation_delta : hs.delta;
    at->last_ncorr = (size_t)(hs.slab_violation ? hs.violation_ncorr : hs.ncorr);
  }
  return CILHIP_OK;
}

int cilhip_icp_partial_keys(cilhip_ctx* c, uint64_t* keys_dev) {
  if (!c || !keys_dev) return CILHIP_ERR_INVALID;
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  CK(c, hipSetDevice(c->device));
  IterArgs a = make_iter_args(c, c->run_prm.max_sq_dist);
  if (c->ns) {
    if (c->grid.n == 0) CK(c, hipMemsetAsync(c->d_nn_pos, 0xFF, (size_t)c->ns * sizeof(uint32_t), c->stream));      // (a shard without target points: every key "none")
    else if (use_tiled(c)) launch_search_tiled(a, IM_NONE, c->d_tiles, c->d_tile_center, c->d_tile_box, c->ntiles, c->stream);
    else launch_iter(a, IM_NONE, true, true, iter_num_blocks(c->ns), c->stream);
    launch_pack_keys(c->d_src_sorted, c->grid.pts, c->d_nn_pos, c->d_nn_d2, c->ns, c->index_offset,
                     reinterpret_cast<unsigned long long*>(keys_dev), c->stream);
  }
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

int cilhip_icp_sums_from_keys(cilhip_ctx* c, const uint64_t* keys_dev, double* sums_dev) {
  if (!c || !keys_dev || !sums_dev) return CILHIP_ERR_INVALID;
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  CK(c, hipSetDevice(c->device));
  if (!c->d_inv_perm) {
    CK(c, hipMalloc(&c->d_inv_perm, (c->grid.n ? c->grid.n : 1) * sizeof(uint32_t)));
    launch_inv_perm(c->grid.pts, c->grid.n, c->d_inv_perm, c->stream);
  }
  const int im = iter_metric_of(c, &c->run_prm);
  IterArgs a = make_iter_args(c, c->run_prm.max_sq_dist);
  a.cw = corr_weights_of(c, &c->run_prm);
  const int nb = iter_num_blocks(c->ns);
  if (c->ns && c->grid.n) {
    launch_keys_to_pos(c->d_src_sorted, reinterpret_cast<const unsigned long long*>(keys_dev), c->d_inv_perm, c->ns,
                       c->index_offset, c->grid.n, c->d_nn_pos, c->d_nn_d2, c->stream,
                       (tie_mode_on(c) && !c->d_tie_leaf_slot) ? c->d_tie_counters : nullptr);
    launch_iter(a, im, false, false, nb, c->stream);
    launch_reduce_partials(c->d_partials, nb, c->d_stage, sums_dev, c->stream);
  } else {
    CK(c, hipMemsetAsync(sums_dev, 0, SUMS_MAX * sizeof(double), c->stream));
  }
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

// The reference's tie order across target shards (extract.hip: tie_rank): after the MIN all-reduce of cilhip_icp_partial_keys' keys,
//   cilhip_icp_order_keys(ctx, win_keys_dev, order_keys_dev)   order_keys_dev[i] = where this shard's match of source point i comes in
//                                                              the query's traversal of the WHOLE target's tree, if it is at the
//                                                              winning distance; 0x7fff...f otherwise
//   all-reduce(MIN, 64-bit) of order_keys_dev                  -> the first-met point of the whole target
//   cilhip_icp_sums_from_ordered_keys(ctx, win_keys_dev, order_keys_dev, sums_dev)   accumulates the pairs whose key came back
// Needs the whole target's order tables on every shard (cilhip_load_tie_order with the shard's global indices).
int cilhip_icp_order_keys(cilhip_ctx* c, const uint64_t* win_keys_dev, uint64_t* order_keys_dev) {
  if (!c || !win_keys_dev || !order_keys_dev) return CILHIP_ERR_INVALID;
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  if (!c->d_tie_leaf_slot) return fail(c, CILHIP_ERR_INVALID, "icp_order_keys: load the whole target's tie order first (cilhip_load_tie_order)");
  if (c->tie_max_depth > 58) return fail(c, CILHIP_ERR_UNSUPPORTED, "icp_order_keys: the order tree is deeper than the 58 levels a traversal key holds");
  CK(c, hipSetDevice(c->device));
  if (!c->d_own_order) CK(c, hipMalloc(&c->d_own_order, (c->ns ? c->ns : 1) * sizeof(unsigned long long)));
  TieDev t = tie_dev_of(c);
  launch_order_keys(c->d_src_sorted, c->d_state, reinterpret_cast<const unsigned long long*>(win_keys_dev), c->d_nn_pos, c->d_nn_d2, c->ns, t,
                    c->d_own_order, reinterpret_cast<unsigned long long*>(order_keys_dev), c->stream);
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

int cilhip_icp_sums_from_ordered_keys(cilhip_ctx* c, const uint64_t* win_keys_dev, const uint64_t* order_keys_dev, double* sums_dev) {
  if (!c || !win_keys_dev || !order_keys_dev || !sums_dev) return CILHIP_ERR_INVALID;
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  if (!c->d_own_order) return fail(c, CILHIP_ERR_INVALID, "icp_sums_from_ordered_keys: cilhip_icp_order_keys first");
  CK(c, hipSetDevice(c->device));
  const int im = iter_metric_of(c, &c->run_prm);
  IterArgs a = make_iter_args(c, c->run_prm.max_sq_dist);
  a.cw = corr_weights_of(c, &c->run_prm);
  const int nb = iter_num_blocks(c->ns);
  if (c->ns && c->grid.n) {
    launch_select_ordered(c->d_src_sorted, c->d_own_order, reinterpret_cast<const unsigned long long*>(order_keys_dev),
                          reinterpret_cast<const unsigned long long*>(win_keys_dev), c->ns, c->d_nn_pos, c->d_nn_d2, c->stream);
    launch_iter(a, im, false, false, nb, c->stream);
    launch_reduce_partials(c->d_partials, nb, c->d_stage, sums_dev, c->stream);
  } else {
    CK(c, hipMemsetAsync(sums_dev, 0, SUMS_MAX * sizeof(double), c->stream));
  }
  CK(c, hipGetLastError());
  return CILHIP_OK;
}

int cilhip_icp_state(cilhip_ctx* c, cilhip_icp_result* out) {
  if (!c || !out) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  const int rc = read_state(c, out);   // (synchronises the stream)
  if (rc == CILHIP_OK && c->run_nar) {
    double ms = 0.0;
    for (size_t k = 0; k + 2 <= c->run_nar; k += 2) { float a = 0.f; CK(c, hipEventElapsedTime(&a, c->ev_ar[k], c->ev_ar[k + 1])); ms += a; }
    c->last_allreduce_ms = ms; c->last_allreduce_n = (int)(c->run_nar / 2);
    c->run_nar = 0;
  }
  if (rc == CILHIP_OK && c->run_nev) {
    // kernel timing of a sharded run: search / accumulation time summed over the cilhip_icp_partial_sums calls since
    // cilhip_icp_begin (read with cilhip_get_last_timing / cilhip_get_last_timing2)
    double sm = 0.0, am = 0.0;
    for (size_t k = 0; k + 3 <= c->run_nev; k += 3) {
      float a = 0.f, b = 0.f;
      CK(c, hipEventElapsedTime(&a, get_event(c, 2 + k), get_event(c, 2 + k + 1)));
      CK(c, hipEventElapsedTime(&b, get_event(c, 2 + k + 1), get_event(c, 2 + k + 2)));
      sm += a; am += b;
      if (k / 3 < c->iter_form.size()) { c->form_ms[c->iter_form[k / 3]] += a; ++c->form_n[c->iter_form[k / 3]]; }
    }
    c->last_search_ms = sm; c->last_acc_ms = am; c->last_search_launches = (int)(c->run_nev / 3);
    c->last_loop_ms = 0.0;
    c->run_nev = 0;
  }
  return rc;
}

int cilhip_compute_residuals(cilhip_ctx* c, int metric, float w_p2p, float w_p2pl, const float T[16], float* out, int mem) {
  if (!c || !T || !out) return CILHIP_ERR_INVALID;
  CK(c, hipSetDevice(c->device));
  if (metric != 0 && !c->has_normals) return fail(c, CILHIP_ERR_INVALID, "compute_residuals: combined metric needs target normals");
  int rc = ensure_sorted(c, T);
  if (rc) return rc;
  c->tie_counters_fresh = false;
  rc = (c->tie_rule == 1 && tie_mode_on(c) && c->ns && c->grid.n) ? build_tie_tables(c) : CILHIP_OK;
  if (rc) return rc;
  launch_init_state(c->d_state, T, c->src_mean, c->stream, nullptr, 0, nullptr, nullptr, c->d_tie_counters);
  IterArgs a = make_iter_args(c, 3.402823466e+38f);
  float* d_out = out;
  if (mem != CILHIP_MEM_DEVICE) CK(c, hipMalloc(&d_out, (c->ns ? c->ns : 1) * sizeof(float)));
  launch_residuals(a, metric, w_p2p, w_p2pl, d_out, c->stream);
  CK(c, hipGetLastError());
  if (metric != 0) {      // (the point-to-plane term reads the matched point's normal: which of two equidistant points matters)
    bool again = false;
    rc = tie_check_pending(c, &again);
    if (rc) { if (mem != CILHIP_MEM_DEVICE) (void)hipFree(d_out); return rc; }
    if (again) { a = make_iter_args(c, 3.402823466e+38f); launch_residuals(a, metric, w_p2p, w_p2pl, d_out, c->stream); CK(c, hipGetLastError()); }
  }
  if (mem != CILHIP_MEM_DEVICE) {
    if (c->ns) CK(c, hipMemcpyAsync(out, d_out, (size_t)c->ns * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_out);
  }
  return CILHIP_OK;
}

int cilhip_get_grid_info(cilhip_ctx* c, cilhip_grid_info* o) {
  if (!c || !o) return CILHIP_ERR_INVALID;
  if (!c->has_target) return fail(c, CILHIP_ERR_INVALID, "no target");
  o->nx = c->grid.nx; o->ny = c->grid.ny; o->nz = c->grid.nz;
  o->cell = c->grid.cell;
  o->origin[0] = c->grid.ox; o->origin[1] = c->grid.oy; o->origin[2] = c->grid.oz;
  o->n_cells = c->grid_cells;
  o->avg_occupancy = c->grid_occ;
  o->build_ms = c->build_ms;
  return CILHIP_OK;
}

int cilhip_get_last_timing(cilhip_ctx* c, double* loop_ms, double* search_ms, int* launches) {
  if (!c) return CILHIP_ERR_INVALID;
  if (loop_ms) *loop_ms = c->last_loop_ms;
  if (search_ms) *search_ms = c->last_search_ms;
  if (launches) *launches = c->last_search_launches;
  return CILHIP_OK;
}

}  // extern "C"

#include "rccl_api.hpp"

// =====================================================================================================================
// One process PER device (torchrun, MPI): this process' context as one rank of an RCCL communicator, and the sharded loop's
// inner triple -- partial sums, all-reduce of the 48 f64, epilogue -- run for a number of iterations inside ONE call: per
// iteration the host enqueues a handful of launches and one ncclAllReduce on the context's stream instead of going through three
// foreign-function calls and a framework collective (measured with one rank: 0.169 -> see DESIGN.md section 8).  The id travels
// by whatever the launcher already has (torch.distributed broadcast, MPI_Bcast, a file).
namespace { RcclApi g_rank_rccl; }

extern "C" {

int cilhip_rank_comm_unique_id(unsigned char id_out[128]) {
  if (!id_out) return CILHIP_ERR_INVALID;
  if (!g_rank_rccl.load()) return CILHIP_ERR_UNSUPPORTED;
  RcclApi::UniqueId u;
  if (g_rank_rccl.GetUniqueId(&u) != 0) return CILHIP_ERR_HIP;
  memcpy(id_out, u.internal, sizeof(u.internal));
  return CILHIP_OK;
}

// Everything of cilhip_rank_comm_init that can fail on ONE rank alone -- opening librccl, the buffer of the rows -- done beforehand, so
// that the ranks can agree (one MIN over whatever channel the launcher has) to enter the collective ncclCommInitRank only when every
// one of them will get through: a rank that bailed out before the collective would leave its peers waiting inside it.
int cilhip_rank_comm_prepare(cilhip_ctx* c) {
  if (!c) return CILHIP_ERR_INVALID;
  if (!g_rank_rccl.load()) return fail(c, CILHIP_ERR_UNSUPPORTED, "rank_comm_prepare: librccl.so.1 could not be opened");
  CK(c, hipSetDevice(c->device));
  if (!c->d_rank_sums && hipMalloc(&c->d_rank_sums, (size_t)RANK_ROWS * SUMS_MAX * sizeof(double)) != hipSuccess)
    return fail(c, CILHIP_ERR_HIP, "rank_comm_prepare: out of device memory");
  return CILHIP_OK;
}

int cilhip_rank_comm_init(cilhip_ctx* c, const unsigned char id[128], int nranks, int rank) {
  if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return CILHIP_ERR_INVALID;
  if (c->rank_comm) return fail(c, CILHIP_ERR_INVALID, "rank_comm_init: the context already holds a communicator");
  if (!g_rank_rccl.load()) return fail(c, CILHIP_ERR_UNSUPPORTED, "rank_comm_init: librccl.so.1 could not be opened");
  CK(c, hipSetDevice(c->device));
  RcclApi::UniqueId u;
  memcpy(u.internal, id, sizeof(u.internal));
  rccl_comm_t comm = nullptr;
  if (g_rank_rccl.CommInitRank(&comm, nranks, u, rank) != 0 || !comm) return fail(c, CILHIP_ERR_HIP, "ncclCommInitRank failed");
  if (!c->d_rank_sums && hipMalloc(&c->d_rank_sums, (size_t)RANK_ROWS * SUMS_MAX * sizeof(double)) != hipSuccess) {
    (void)g_rank_rccl.CommDestroy(comm);
    return fail(c, CILHIP_ERR_HIP, "rank_comm_init: out of device memory");
  }
  c->rank_comm = comm; c->rank_comm_size = nranks;
  return CILHIP_OK;
}

int cilhip_rank_comm_destroy(cilhip_ctx* c) {
  if (!c) return CILHIP_ERR_INVALID;
  if (c->rank_comm) { (void)hipStreamSynchronize(c->stream); (void)g_rank_rccl.CommDestroy(c->rank_comm); c->rank_comm = nullptr; c->rank_comm_size = 0; }
  if (c->d_rank_sums) { (void)hipFree(c->d_rank_sums); c->d_rank_sums = nullptr; }
  return CILHIP_OK;
}

int cilhip_get_last_allreduce_timing(cilhip_ctx* c, double* total_ms, int* timed) {
  if (!c) return CILHIP_ERR_INVALID;
  if (total_ms) *total_ms = c->last_allreduce_ms;
  if (timed) *timed = c->last_allreduce_n;
  return CILHIP_OK;
}

int cilhip_get_last_host_enqueue_time(cilhip_ctx* c, double* us_per_iteration) {
  if (!c || !us_per_iteration) return CILHIP_ERR_INVALID;
  *us_per_iteration = c->run_enqueue_iters ? c->run_enqueue_us / c->run_enqueue_iters : 0.0;
  return CILHIP_OK;
}

int cilhip_icp_iterate_ranked(cilhip_ctx* c, int iterations) {
  if (!c || iterations < 0) return CILHIP_ERR_INVALID;
  if (!c->rank_comm) return fail(c, CILHIP_ERR_INVALID, "icp_iterate_ranked: cilhip_rank_comm_init first");
  if (!c->run_active) return fail(c, CILHIP_ERR_INVALID, "icp_begin first");
  CK(c, hipSetDevice(c->device));
  const int im = iter_metric_of(c, &c->run_prm);
  const auto t_call = std::chrono::steady_clock::now();
  const double wait0 = c->wait_us;
  struct Acc { cilhip_ctx* c; std::chrono::steady_clock::time_point t; double w0; int n;
               ~Acc() { c->run_enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count() - (c->wait_us - w0); c->run_enqueue_iters += n; } } acc{c, t_call, wait0, iterations};
  for (int k = 0; k < iterations; ++k) {
    // this rank's RANK_ROWS rows of partial sums -> summed over the ranks, row by row -> folded by the epilogue (the same values on
    // every rank: identical transforms and decisions everywhere)
    // (with kernel timing on, the iterations that carry kernel events also time their collective: what the all-reduce costs per
    //  iteration ON THE STREAM -- launch of RCCL's kernel, the exchange over xGMI, the wait for the slowest rank -- is the figure a
    //  scaling curve has to be read against; cilhip_get_last_allreduce_timing)
    const bool time_ar = c->kernel_timing && c->run_nar + 2 <= 2 * 4096 &&
                         (c->timing_stride <= 1 || c->run_calls < 3 || c->run_calls % c->timing_stride == 0);
    const int rc = partial_sums_core(c, nullptr, c->d_rank_sums);
    if (rc) return rc;
    if (time_ar) {
      while (c->ev_ar.size() < c->run_nar + 2) { hipEvent_t e; CK(c, hipEventCreate(&e)); c->ev_ar.push_back(e); }
      CK(c, hipEventRecord(c->ev_ar[c->run_nar], c->stream));
    }
    if (g_rank_rccl.AllReduce(c->d_rank_sums, c->d_rank_sums, (size_t)RANK_ROWS * SUMS_MAX, RCCL_DOUBLE, RCCL_SUM, c->rank_comm, c->stream) != 0)
      return fail(c, CILHIP_ERR_HIP, "ncclAllReduce failed");
    if (time_ar) { CK(c, hipEventRecord(c->ev_ar[c->run_nar + 1], c->stream)); c->run_nar += 2; }
    SolveArgs sa = make_solve_args(c, &c->run_prm, im, c->run_src_mean);
    sa.feedback = c->d_feedback; sa.run_tag = c->run_tag;
    sa.partials = c->d_rank_sums; sa.nblocks = RANK_ROWS; sa.reduced = nullptr;
    launch_solve(sa, c->stream);
    CK(c, hipGetLastError());
  }
  return CILHIP_OK;
}

}  // extern "C"
