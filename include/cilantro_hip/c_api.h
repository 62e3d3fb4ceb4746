/*
 * cilantro_hip/c_api.h -- C ABI of the MI355X-native rigid ICP engine (libcilantro_hip.so).
 *
 * cilantro (the reference) has NO ABI for this path: it is C++ template duck typing
 * (registration/icp_base.hpp:8-10,114).  This header is the drop-in
 * boundary a maintainer binds instead: plain pointers and sizes, no Eigen / torch types.
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/include/cilantro/).  The C++ mirror of the reference classes that sits on
 * top of it is include/cilantro_hip/icp.hpp; INTEGRATION.md shows the binding.
 *
 * Conventions (identical to the reference):
 *   - clouds: const float* xyz, point i at xyz[3i..3i+2]  (Eigen::Matrix<float,3,Dynamic>
 *     column-major, core/data_containers.hpp:155-156); "first" = dst/target, "second" = src.
 *   - transforms: float[16], 4x4 COLUMN-major (Eigen::Transform<float,3,Isometry>::data()).
 *   - all distances are SQUARED L2 (core/kd_tree.hpp:46-47).
 *   - every call returns 0 on success or a negative cilhip_status; cilhip_last_error() has text.
 *     Nothing throws across the boundary.  A context is single-caller (like one ICP object).
 *   - `mem` arguments: CILHIP_MEM_HOST = pointer is host memory (copied), CILHIP_MEM_DEVICE =
 *     pointer is device memory on the context's GPU (copied device-to-device; caller keeps
 *     ownership either way).
 */
#ifndef CILANTRO_HIP_C_API_H
#define CILANTRO_HIP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cilhip_ctx cilhip_ctx;

typedef enum {
  CILHIP_OK = 0,
  CILHIP_ERR_INVALID = -1,      /* bad argument / call order */
  CILHIP_ERR_HIP = -2,          /* HIP runtime failure (no device, OOM, launch failure) */
  CILHIP_ERR_UNSUPPORTED = -3,  /* option combination the GPU path does not implement */
  CILHIP_ERR_NO_DEVICE = -4
} cilhip_status;

enum { CILHIP_MEM_HOST = 0, CILHIP_MEM_DEVICE = 1 };

/* metric selector: which reference ICP class is being replaced */
enum {
  CILHIP_METRIC_POINT_TO_POINT = 0, /* PointToPointMetricSingleTransformICP (closed-form SVD),
                                       registration/icp_single_transform_point_to_point_metric.hpp */
  CILHIP_METRIC_COMBINED = 1        /* CombinedMetricSingleTransformICP (Gauss-Newton; defaults
                                       w_p2p=0, w_p2pl=1 = point-to-plane),
                                       registration/icp_single_transform_combined_metric.hpp */
};

/* ---- lifetime -------------------------------------------------------------------------------- */
/* device: HIP device ordinal.  Owns one stream, all device buffers, the target grid index. */
int cilhip_create(cilhip_ctx** out, int device);
void cilhip_destroy(cilhip_ctx* ctx);
const char* cilhip_last_error(const cilhip_ctx* ctx);
/* Run all work on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * context's own; pass NULL to go back.  The caller keeps the stream alive.  NULL never means the legacy default
 * stream here: a caller whose work runs there (torch's default stream reports handle 0) passes hipStreamLegacy --
 * otherwise the context's own NON-BLOCKING stream is not ordered with that work at all. */
int cilhip_set_stream(cilhip_ctx* ctx, void* hip_stream);
int cilhip_synchronize(cilhip_ctx* ctx);

/* ---- clouds ---------------------------------------------------------------------------------- */
/* Target ("first"/dst) cloud + optional normals.  Builds the uniform-grid index on the GPU.
 * Replaces: KDTree ctor core/kd_tree.hpp:162-170 -> nanoflann buildIndex
 * (3rd_party/nanoflann/nanoflann.hpp:1661-1687), lazily triggered at
 * correspondence_search/correspondence_search_kd_tree.hpp:202-203; and the dst_mean_ of
 * registration/icp_single_transform_combined_metric.hpp:51-54. */
int cilhip_set_target(cilhip_ctx* ctx, const float* xyz, const float* normals_or_null, size_t n,
                      int mem);
/* Source ("second"/src) cloud.  Replaces the PointFeaturesAdaptor src_feat_
 * (correspondence_search/common_transformable_feature_adaptors.hpp:14-17) and src_mean_
 * (icp_single_transform_combined_metric.hpp:55-58). */
int cilhip_set_source(cilhip_ctx* ctx, const float* xyz, size_t n, int mem);
/* Optional source normals (call after cilhip_set_source; NULL removes them).  With them the combined
 * metric becomes the SYMMETRIC objective, exactly as constructing CombinedMetricSingleTransformICP with four
 * clouds does (icp_single_transform_combined_metric.hpp:63-93,182-189 -> estimateTransformSymmetricMetric,
 * transform_estimation.hpp:604-739: n = n_dst + R*n_src), and computeResiduals adds them (:237). */
int cilhip_set_source_normals(cilhip_ctx* ctx, const float* normals_or_null, int mem);
/* The per-registration set-up of THIS design (the reference has no counterpart: its queries are searched in their
 * given order, correspondence_search_kd_tree_utilities.hpp:26): spatial pre-sort of the source by the target-grid cube
 * of T*s and the tile table of the LDS-tiled search.  It runs lazily inside the first search / cilhip_icp_run of a
 * (target, source) pair and is re-run when the transform has moved the source by more than a few cells; this entry
 * runs it explicitly so that its cost can be put on the bench line.  T NULL = identity; force != 0 re-sorts even if a
 * valid order exists; ms_or_null = host wall time of the step (allocations and the two host round trips of the
 * radix sort included), stream-synchronised. */
int cilhip_prepare_source(cilhip_ctx* ctx, const float* T_or_null, int force, double* ms_or_null);
/* dst_mean_ / src_mean_ as the ICP classes hold them (f64 sum, rounded to f32). */
int cilhip_get_means(cilhip_ctx* ctx, float dst_mean[3], float src_mean[3]);
/* Per-point colours of both clouds (packed rgb, n_target / n_source triples) for the point + colour feature adaptor
 * (PointColorFeaturesAdaptor3f(points, colors, color_weight), common_transformable_feature_adaptors.hpp:164-252; options
 * "feature_kind" = 1, "feature_normal_weight" = the colour weight) and for the 9-D point + normal + colour adaptor (:255-343;
 * "feature_kind" = 2, "feature_color_weight").  After cilhip_set_target and cilhip_set_source. */
int cilhip_set_color_features(cilhip_ctx* ctx, const float* dst_rgb, const float* src_rgb, int mem);
/* CorrespondenceSearchKDTree::getFirstSearchTree() / setFirstSearchTree(tree) (correspondence_search/correspondence_search_kd_tree.hpp:273-296;
 * the shared_ptr<SearchTree> of :299-300): a second engine searches the index another engine built instead of building its own --
 * many sources registered against one model (examples/fusion.cpp style) pay for the index once.  `ctx` takes `from`'s target as it
 * stands: the cell-sorted points and normals, the cell table, and everything built on top of them by now (point+normal records,
 * the nearest-other-point table of the warm-started iterations, the order tables of the reference's tree, the index -> position
 * map), without copying: the allocations become a reference-counted share, so either context may be destroyed or given another
 * target first.  What a context builds later is its own.  Same device; neither context may have work in flight on another host
 * thread during the call.  Colour features are per context (cilhip_set_color_features again). */
int cilhip_share_target(cilhip_ctx* ctx, cilhip_ctx* from);

/* ---- correspondence search (engine concept) -------------------------------------------------- */
/* CorrespondenceSearchKDTree::findCorrespondences(tform), SECOND_TO_FIRST, L2, identity
 * evaluator, inlier_fraction 1, no one-to-one/reciprocity
 * (correspondence_search/correspondence_search_kd_tree.hpp:107-229 live code :185-228):
 *   q_i = T*s_i (common_transformable_feature_adaptors.hpp:28-33), exact 1-NN of q_i among dst with
 *   d2 < max_sq_dist, strict (correspondence_search_kd_tree_utilities.hpp:26-33; nanoflann.hpp:1901).
 * Ties on d2 (several target points at EXACTLY the smallest distance) name the point the reference's kd-tree traversal meets first
 * (option "tie_rule" = 2, the default; 0: the lowest dst index).
 * Results stay on the device; n_found (optional) forces a sync and returns the count. */
int cilhip_find_correspondences(cilhip_ctx* ctx, const float T[16], float max_sq_dist,
                                size_t* n_found_or_null);
/* Per-source raw result of the last search, in ORIGINAL source order: nn_idx[i] = dst index or
 * 0xFFFFFFFF (none), nn_d2[i] = squared distance (undefined when none).  Either may be NULL.
 * "The last search" includes the one of the last executed iteration of cilhip_icp_run: like the reference's engine, which
 * keeps `correspondences_` of its last findCorrespondences call (correspondence_search_kd_tree.hpp:231; the ICP object hands
 * the engine out, registration/icp_base.hpp:32-38), the context answers cilhip_get_nn / cilhip_get_correspondences /
 * cilhip_estimate_* after a run with that iteration's set (found under the transform BEFORE the last update). */
int cilhip_get_nn(cilhip_ctx* ctx, uint32_t* nn_idx, float* nn_d2, int mem);
/* Where the set those calls return comes from: 0 = there is none, 3 = a cilhip_find_correspondences call, 1 = left in device
 * memory by the kernels of the last cilhip_icp_run iteration (the squared distances formed again with the search's pinned
 * arithmetic), 2 = searched again on demand under that iteration's transform (loops whose kernels keep no per-query matches:
 * post-filters, pair-list directions, the feature search, options "fused" / "warm_start" = 0).  The search is exact, so 1 and
 * 2 are the same set; tests use the value to know WHICH kernel's matches they are comparing. */
int cilhip_get_last_matches_origin(cilhip_ctx* ctx, int* origin);
/* The transform (col-major 4x4) the current correspondence set was found under: the argument of the last
 * cilhip_find_correspondences, or -- after cilhip_icp_run -- transform_ as it was BEFORE the last iteration's update, the
 * tform the reference's loop handed its last engine.findCorrespondences(transform_) call
 * (registration/icp_single_transform_combined_metric.hpp:170). */
int cilhip_get_matches_transform(cilhip_ctx* ctx, float T[16]);
/* The reference's SearchResult: CorrespondenceSet<float,size_t> compacted in ascending source
 * index order (correspondence_search_kd_tree_utilities.hpp:45-50; core/correspondence.hpp:9-55),
 * as three host arrays of capacity `cap` (>= n_found): indexInFirst, indexInSecond, value. */
int cilhip_get_correspondences(cilhip_ctx* ctx, uint64_t* index_in_first, uint64_t* index_in_second,
                               float* value, size_t cap, size_t* n_out);

/* ---- estimators on the correspondences of the last search ------------------------------------ */
/* estimateTransformPointToPointMetric (rigid, closed form), registration/transform_estimation.hpp
 * :11-48 via :104-113.  dT_out: the step transform (tform_iter before the rotation() polish).
 * sums_or_null (16 doubles): n, sum p(3), sum q(3), sum p q^T(9, row-major) -- raw moments.
 * Returns CILHIP_OK; *ok_or_null = (n >= 3) as the reference's bool. */
int cilhip_estimate_point_to_point(cilhip_ctx* ctx, float dT_out[16], double* sums_or_null,
                                   int* ok_or_null);
/* estimateTransformCombinedMetric (rigid 3D, unity weights), transform_estimation.hpp:237-367,
 * called as icp_single_transform_combined_metric.hpp:191-196 does (dst_mean, T*src_mean).
 * AtA_or_null (36, row-major) / Atb_or_null (6): first Gauss-Newton step's normal equations.
 * *converged_or_null: the reference's bool return (d_theta.norm() < conv_tol inside max_iter).
 * Reads the stored set of the last search whatever its direction: the per-source matches of SECOND_TO_FIRST or the pair list of
 * FIRST_TO_SECOND / BOTH (one term per pair); with the context's weight evaluators (options) or pair-weight callback. */
int cilhip_estimate_combined(cilhip_ctx* ctx, float w_p2p, float w_p2pl, size_t max_iter,
                             float conv_tol, float dT_out[16], double* AtA_or_null,
                             double* Atb_or_null, int* converged_or_null);

/* Affine closed form: estimateTransformCombinedMetric for Affine / AffineCompact transforms
 * (transform_estimation.hpp:369-476) called as icp_single_transform_combined_metric.hpp:199-204 does (centered != 0:
 * dst_mean, T*src_mean), or estimateTransformPointToPointMetric (affine, :50-102 via :104-113) with
 * w_p2p = 1, w_p2pl = 0, centered = 0 (that overload works on the raw coordinates).  Works on the correspondences of
 * the last cilhip_find_correspondences in every search direction.  AtA_or_null (144, row-major) / Atb_or_null (12):
 * the normal equations in the reference's unknown order (row-major linear part, translation).
 * *ok_or_null: the reference's bool return (terms >= Dim + 1). */
int cilhip_estimate_affine(cilhip_ctx* ctx, float w_p2p, float w_p2pl, int centered, float dT_out[16],
                           double* AtA_or_null, double* Atb_or_null, size_t* n_corr_or_null, int* ok_or_null);

/* ---- the whole ICP loop, fused on the device ------------------------------------------------- */
typedef struct {
  int metric;            /* CILHIP_METRIC_* */
  float w_p2p, w_p2pl;   /* combined metric weights (reference defaults 0 / 1) */
  size_t max_iter;       /* icp_base.hpp:24  (default 15) */
  float conv_tol;        /* icp_base.hpp:25  (default 1e-5) */
  size_t max_opt_iter;   /* max_optimization_iterations_ (default 1) */
  float opt_conv_tol;    /* optimization_convergence_tol_ (default 1e-5) */
  float max_sq_dist;     /* engine max_distance_, SQUARED (default 0.01*0.01) */
} cilhip_icp_params;

typedef struct {
  float T[16];           /* transform_ */
  size_t iterations;     /* iterations_ */
  float last_delta_norm; /* last_delta_norm_ */
  size_t last_ncorr;     /* correspondences used by the last executed iteration */
} cilhip_icp_result;

void cilhip_icp_default_params(cilhip_icp_params* p);
/* IterativeClosestPointBase::estimate() (registration/icp_base.hpp:68-87) with
 * updateCorrespondences()+updateEstimate() of the selected class, all iterations enqueued
 * back-to-back on the stream (convergence is decided on the device); one sync at the end. */
int cilhip_icp_run(cilhip_ctx* ctx, const cilhip_icp_params* prm, const float T0_or_null[16],
                   cilhip_icp_result* out);

/* A caller's OWN correspondence weight evaluators.  The reference takes them as template arguments of the combined-metric classes
 * (registration/icp_single_transform_combined_metric.hpp:10-14, icp_common_instances.hpp:74-97) and its estimators call them once per
 * correspondence: point_corr_evaluator(corr.indexInFirst, corr.indexInSecond, corr.value) and the plane one alike
 * (registration/transform_estimation.hpp:303, :332 rigid; :432, :453 affine; core/common_pair_evaluators.hpp:14-80 are the three stock
 * classes, which the options "point_weight_evaluator" / "plane_weight_evaluator" evaluate on the device).  A functor cannot cross a C
 * boundary onto the device: with a callback set, every combined-metric estimate brings the stored correspondence set to the host, calls
 * fn ONCE with all n correspondences -- index_in_first (target), index_in_second (source), value (the search's squared distance; the
 * feature distance under a feature adaptor), in stored order: ascending source index, or ascending (first, second) for the pair
 * lists of FIRST_TO_SECOND / BOTH -- and fn writes both weights of every pair (the metric weights w_p2p / w_p2pl multiply them as in
 * the reference).  The accumulation then reads the weights from tables; cilhip_icp_run becomes the reference's loop step by step on
 * the host (search, estimate, rotation() polish, compose: cilhip_icp_run_two_sets with this engine in both roles; SECOND_TO_FIRST).
 * Applies to cilhip_estimate_combined, _estimate_combined_two_sets (each context its own callback), _estimate_affine (combined class)
 * and the loops over them; the point-to-point classes have no evaluators.  fn = NULL: back to the option-selected stock evaluators.
 * fn runs on the calling thread, between device passes.  The sharded building blocks (cilhip_icp_begin ...) refuse a context with a
 * callback (CILHIP_ERR_UNSUPPORTED: no host in their loop); the stock evaluators work there. */
typedef void (*cilhip_pair_weight_fn)(void* user, const uint64_t* index_in_first, const uint64_t* index_in_second, const float* value,
                                      size_t n, float* point_weight_out, float* plane_weight_out);
int cilhip_set_pair_weight_callback(cilhip_ctx* ctx, cilhip_pair_weight_fn fn, void* user);

/* ---- building blocks for source-sharded multi-GPU runs (one process per GPU) ----------------- */
/* Each rank holds the full target and a shard of the source.  Per iteration:
 *   cilhip_icp_begin (once)  ->  { cilhip_icp_partial_sums -> all-reduce(sum) of
 *   CILHIP_SUMS_LEN doubles over RCCL (caller, e.g. torch.distributed) -> cilhip_icp_apply_sums }
 * sums_dev: DEVICE pointer to CILHIP_SUMS_LEN doubles owned by the caller.  No host sync. */
#define CILHIP_SUMS_LEN 48
int cilhip_icp_begin(cilhip_ctx* ctx, const cilhip_icp_params* prm, const float T0_or_null[16],
                     const float global_src_mean_or_null[3]);
int cilhip_icp_partial_sums(cilhip_ctx* ctx, double* sums_dev);
int cilhip_icp_apply_sums(cilhip_ctx* ctx, const double* sums_dev);
int cilhip_icp_state(cilhip_ctx* ctx, cilhip_icp_result* out); /* syncs */

/* Spatially sharded runs (SURVEY.md 8(e) partitioning B; no counterpart in the reference, which has one address space):
 * target and source are cut into slabs along one axis -- a context holds the target points of its slab plus a halo of
 * sqrt(max distance) + slack and the source points whose image under T_part falls into the slab -- so every nearest
 * neighbour is local and the only exchange per iteration is cilhip_icp_partial_sums' 48 doubles.  The guard keeps that
 * exact: after every transform update the device bounds how far any point of the source's bounding box (centre,
 * half-extents, SOURCE coordinates, the same box on every rank) can have moved along the axis since the partition; once
 * that exceeds the slack the sticky flag cilhip_get_slab_violation reads is raised (identically on all ranks): the
 * caller re-partitions under the current transform and repeats from its last checked state.  axis < 0 disarms. */
int cilhip_set_slab_guard(cilhip_ctx* ctx, int axis, float slack, const float center[3], const float half_extent[3],
                          const float T_part[16]);
int cilhip_get_slab_violation(cilhip_ctx* ctx, int* violated_out);
/* ... and the loop state right AFTER the update that raised the flag (iterations performed, transform, update norm,
 * correspondence count): that update is still exact -- its search ran inside the halos, the flag is about the next search --
 * so a caller re-partitions under THAT transform and keeps every iteration up to and including it.  Not violated: the
 * current state. */
int cilhip_get_slab_violation_state(cilhip_ctx* ctx, int* violated, cilhip_icp_result* at_violation);

/* ---- target-sharded runs (BASELINE configs[3]: one target too large / sharded over the GPUs of a node) ----
 * Every rank holds ALL source points and ONE shard of the target (set with cilhip_set_target) plus
 *   cilhip_set_shard_info(ctx, global index of the shard's first point, GLOBAL dst mean, GLOBAL src mean).
 * Per iteration (after cilhip_icp_begin):
 *   cilhip_icp_partial_keys(ctx, keys_dev)      keys_dev[i] = (bits(d2) << 32) | global target index of source
 *                                               point i's nearest neighbour in THIS shard (0x7fff...f = none)
 *   all-reduce(MIN, int64) of keys_dev over RCCL  -> the globally nearest target per source point
 *   cilhip_icp_sums_from_keys(ctx, keys_dev, sums_dev)  accumulates only the pairs won by this shard
 *   all-reduce(SUM, f64) of sums_dev;  cilhip_icp_apply_sums(ctx, sums_dev)
 * keys_dev: DEVICE pointer to n_source uint64 (original source order). */
int cilhip_set_shard_info(cilhip_ctx* ctx, uint64_t target_index_offset, const float* dst_mean_or_null,
                          const float* src_mean_or_null);
int cilhip_icp_partial_keys(cilhip_ctx* ctx, uint64_t* keys_dev);
int cilhip_icp_sums_from_keys(cilhip_ctx* ctx, const uint64_t* keys_dev, double* sums_dev);
/* Exactly equidistant nearest points across shards: MIN of (d2, global index) keeps the lowest global index, the reference's kd-tree
 * the first one its traversal meets (core/kd_tree.hpp:82-90, nanoflann.hpp:1885-1961).  A shard notices such a tie in
 * cilhip_icp_sums_from_keys (its own nearest point is exactly as far as the winner's) and counts it like the ties its searches notice
 * (cilhip_get_tie_order_info: pending).  With the WHOLE target's order loaded on every shard (cilhip_tie_order_create over the whole
 * cloud, cilhip_load_tie_order with the shard's global indices) the searches settle the ties inside a shard, and a second key settles
 * them between shards -- per iteration, after the MIN of the first keys:
 *   cilhip_icp_order_keys(ctx, win_keys_dev, order_keys_dev)    order_keys_dev[i] = position of this shard's match of source point i in
 *                                                                 that query's traversal of the whole tree (one bit per level: 0 = the
 *                                                                 child nanoflann's searchLevel descends into first; then the slot
 *                                                                 in the leaf) if the match is at the winning distance, else 0x7fff...f
 *   all-reduce(MIN, 64-bit) of order_keys_dev
 *   cilhip_icp_sums_from_ordered_keys(ctx, win_keys_dev, order_keys_dev, sums_dev)   accumulates the pairs whose key came back
 * Trees deeper than 58 levels: CILHIP_ERR_UNSUPPORTED.  cilhip_multi_set_clouds(partition = 2) runs this protocol by itself. */
int cilhip_icp_order_keys(cilhip_ctx* ctx, const uint64_t* win_keys_dev, uint64_t* order_keys_dev);
int cilhip_icp_sums_from_ordered_keys(cilhip_ctx* ctx, const uint64_t* win_keys_dev, const uint64_t* order_keys_dev, double* sums_dev);

/* ---- residuals ------------------------------------------------------------------------------- */
/* computeResiduals() of both classes (icp_single_transform_combined_metric.hpp:220-243,
 * icp_single_transform_point_to_point_metric.hpp:68-85): unbounded 1-NN of T*s_i, then
 * w_p2p*|p-q|^2 + w_p2pl*(n.(p-q))^2  (metric 0: |p-q|^2).  out: ns floats, original order. */
int cilhip_compute_residuals(cilhip_ctx* ctx, int metric, float w_p2p, float w_p2pl,
                             const float T[16], float* out, int mem);

/* ---- next tier: KMeans3f --------------------------------------------------------- */
/* KMeans<float,3>::cluster(centroids, max_iter, tol, use_kd_tree = false)  (clustering/kmeans.hpp:24-30 ->
 * cluster_ :67-194): brute-force assignment (strict '<' over ascending cluster index, :95-119), centroid
 * update, empty-cluster repair (:134-176, including the reference's quirk that the moved point is not added
 * to the re-seeded cluster's sum), convergence on assignments (:122) or on the centroid shift (:186-188).
 * centroids: HOST array of 3*k floats, in = initial centroids, out = final.  labels_out: HOST array of n
 * (point_to_cluster_index_map_) or NULL.  k <= 2048.  Cluster sums are exact (fixed point) instead of the
 * reference's serial f32 sums; labels are bit-identical to the reference given identical centroids. */
int cilhip_kmeans3f(int device, const float* xyz, size_t n, int mem, float* centroids, size_t k, size_t max_iter,
                    float tol, uint32_t* labels_out, size_t* iterations_out);
/* The brute-force branch's assignment (kmeans.hpp:95-119: argmin over all k centroids, strict '<' over ascending index) is computed
 * EXACTLY with pruning by default: the centroids are binned into a grid (rebuilt every Lloyd iteration), a point looks at the 3x3x3
 * (then 5x5x5) block of cells around its own with the branch's own distance expression and a proof that nothing outside can win or tie,
 * and falls back to all k centroids otherwise -- labels bit-identical to the exhaustive pass (50M x 1024: 8.05 -> 1.47 ms per Lloyd
 * iteration).  on = 0 keeps the exhaustive pass (process-wide; tests, A/B runs); both branches (use_kd_tree or not) are pruned. */
int cilhip_kmeans_set_pruning(int on);
/* one assignment pass only (kmeans.hpp:95-119) */
int cilhip_kmeans3f_assign(int device, const float* xyz, size_t n, int mem, const float* centroids, size_t k,
                           uint32_t* labels_out);
/* ... with the reference's `use_kd_tree` argument (clustering/kmeans.hpp:24-30, branch :86-94 -- the mode examples/kmeans.cpp
 * uses): a kd-tree over the centroids is only a way of finding the same nearest centroid, so the device runs the same pass (pruned
 * through the centroid grid like the brute-force branch's: 50M x 1024 in 1.9 ms per Lloyd iteration, 2.2 - 2.7 ms when exact ties are met);
 * what the flag changes is the ROUNDING of the compared distance -- nanoflann's L2 metric ((dx*dx)+(dy*dy))+(dz*dz)
 * instead of Eigen's squaredNorm pairing of the brute-force branch -- so that labels equal the reference's kd-tree branch
 * wherever its nearest centroid is unique; among EXACTLY equidistant centroids the one the reference's traversal meets first: the
 * order tables of the tree over the iteration's centroids are built on the device (one workgroup for up to 2048 points: a fraction of a
 * millisecond; the reference builds its KDTree at :87) in the iterations whose pass met such points -- it lists them, k_fix_ties settles them with tie_before afterwards
 * (cilhip_knn_set_tie_rule(0): the lowest index instead).  tests/test_gpu_tie_rule.py: lattice centroids, label for label. */
int cilhip_kmeans3f_ex(int device, const float* xyz, size_t n, int mem, float* centroids, size_t k, size_t max_iter, float tol, int use_kd_tree,
                       uint32_t* labels_out, size_t* iterations_out);
int cilhip_kmeans3f_assign_ex(int device, const float* xyz, size_t n, int mem, const float* centroids, size_t k, int use_kd_tree,
                              uint32_t* labels_out);

/* KMeans over SHARDS of the points (last row: "points sharded, centroids replicated; all-reduce of k x (3 sums +
 * count)"): one shard object per device / rank holds its points and their labels; a Lloyd iteration is cilhip_kmeans_shard_assign on
 * every shard under the same centroids, the SUM of the shards' fixed-point sums {x, y, z, count} per cluster (int64: exact, order-free --
 * MPI / RCCL all-reduce of 4k integers), then clustering/kmeans.hpp:122-188 on the summed values, identically on every rank.  All shards
 * of a run use ONE scale exponent: cilhip_kmeans_scale_exponent(largest |coordinate| over all shards, total point count).  The
 * empty-cluster repair (:134-176) needs the farthest member of a cluster over all shards: cilhip_kmeans_shard_farthest returns this shard's
 * candidate as a key (bits(distance^2) << 32 | 0xFFFFFFFF - GLOBAL index; 0 = no member) whose MAXIMUM over the shards names the
 * reference's point (ties: lowest index); its owner moves it (cilhip_kmeans_shard_move_point: new label, coordinates out -- every rank
 * subtracts them from the donor cluster's sums).  With one shard holding all points this is cilhip_kmeans3f_ex itself (it runs on these
 * calls); with several the centroids, labels and iteration counts are the single-device run's bit for bit (the sums are integers).
 * cilantro_amd/distributed_models.py: ShardedKMeans3f is the loop over torch.distributed; tests/test_distributed_cpu.py (gloo, world 2)
 * and tests/test_gpu_distributed.py (two processes, HIP shards). */
typedef struct cilhip_kmeans_shard cilhip_kmeans_shard;
int cilhip_kmeans_shard_create(int device, const float* xyz, size_t n, int mem, size_t k, uint64_t index_offset, cilhip_kmeans_shard** out);
void cilhip_kmeans_shard_destroy(cilhip_kmeans_shard* shard);
int cilhip_kmeans_shard_maxabs(cilhip_kmeans_shard* shard, float* maxabs_out);
int cilhip_kmeans_scale_exponent(double maxabs_all, size_t n_all);
int cilhip_kmeans_shard_assign(cilhip_kmeans_shard* shard, const float* centroids, int scale_exponent, int use_kd_tree, int64_t* sums_out /* [4k] */,
                               uint64_t* changed_out);
int cilhip_kmeans_shard_farthest(cilhip_kmeans_shard* shard, uint32_t cluster, const float center[3], uint64_t* key_out);
int cilhip_kmeans_shard_move_point(cilhip_kmeans_shard* shard, uint64_t global_index, uint32_t to_cluster, float xyz_out[3]);
int cilhip_kmeans_shard_labels(cilhip_kmeans_shard* shard, uint32_t* labels_out);

/* ---- next tier: PlaneRANSACEstimator3f ----------------------------------- */
typedef struct {
  float normal[3];     /* Eigen::Hyperplane<float,3>::normal()  (unit; sign as PCA leaves it)               */
  float offset;        /* ::offset(): the plane is normal . x + offset = 0                                  */
  size_t iterations;   /* getNumberOfPerformedIterations()   (ransac_base.hpp:103)                          */
  size_t n_inliers;    /* getModelInliers().size()                                                          */
  int target_reached;  /* targetInlierCountAchieved()        (ransac_base.hpp:172)                          */
  double device_ms;    /* kernels only (hypothesis fit + scoring + re-estimation + outputs), HIP events     */
} cilhip_plane_model;
/* HyperplaneRANSACEstimator<float,3>::estimate() (model_estimation/ransac_base.hpp:64-131 with
 * ransac_hyperplane_estimator.hpp:47-55 computeResiduals and :78-85 estimate_params_).
 * samples: HOST array of 3*max_iter point indices, the random sample of every iteration in order
 *   (ransac_base.hpp:83-91), or NULL to draw them here (uniform distinct triples from `seed`; the
 *   reference seeds std::mt19937 from std::random_device, so its sequence is not reproducible either).
 * All max_iter hypotheses may be scored, but the result is the one the reference's sequential loop reaches
 * with the same samples: first strictly-better model wins, stop at the first iteration whose best inlier
 * count reaches target_inliers (clamped to n, :68).  re_estimate: PCA over the best model's inliers, then
 * residuals / inliers of the re-estimated plane (:118-128).
 * residuals_out: HOST, n floats or NULL.  inliers_out: HOST, capacity n, ascending indices, or NULL.
 * No accepted model => NaN plane, no inliers. */
int cilhip_plane_ransac3f(int device, const float* xyz, size_t n, int mem, const uint32_t* samples, uint64_t seed,
                          float max_residual, size_t target_inliers, size_t max_iter, int re_estimate,
                          cilhip_plane_model* out, float* residuals_out, uint32_t* inliers_out);
/* inlier counts (#points with absDistance <= max_residual, ransac_base.hpp:98-100) of m given planes
 * (HOST, 4*m floats: normal, offset) in one pass over the points per 128 planes.  counts_out: HOST, m. */
int cilhip_plane_score3f(int device, const float* xyz, size_t n, int mem, const float* planes, size_t m,
                         float max_residual, uint32_t* counts_out);
/* estimateModel() over ALL points (ransac_hyperplane_estimator.hpp:22-25, :70-76): PCA plane fit. */
int cilhip_plane_fit3f(int device, const float* xyz, size_t n, int mem, float plane_out[4]);

/* ---- RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp, SURVEY.md section 2 "next tier") ---- */
typedef struct cilhip_transform_model {
  float T[16];         /* getModel(): col-major 4x4 rigid transform mapping src onto dst (identity when nothing was estimated) */
  size_t iterations;   /* getNumberOfPerformedIterations()                                                   */
  size_t n_inliers;    /* getModelInliers().size()                                                           */
  int have_model;      /* bit 0: some hypothesis was accepted (>= sample_size inliers), bit 1: re-estimated  */
  int target_reached;  /* targetInlierCountAchieved()        (ransac_base.hpp:172)                           */
  double device_ms;    /* kernels only, HIP events                                                           */
} cilhip_transform_model;
/* TransformRANSACEstimator<RigidTransform<float,3>>::estimate() over n point PAIRS (dst_xyz[i], src_xyz[i]) -- the
 * reference's constructors gather them from a correspondence set or two index lists (ransac_transform_estimator.hpp:34-59);
 * estimateModel = estimateTransformPointToPointMetric of the sampled pairs (:75-83; registration/transform_estimation.hpp:11-48),
 * residual_i = |T * src_i - dst_i| (:90-98), loop / replay / re-estimation as ransac_base.hpp:64-131.
 * samples: HOST, 3 * max_iter pair indices (the sample of every iteration, in order) or NULL to draw them here from `seed`.
 * Defaults of the reference (:27-30): sample size 3, target ceil(n / 2), 100 iterations, max residual 0.01, re-estimate.
 * residuals_out: HOST, n floats or NULL.  inliers_out: HOST, capacity n, ascending pair indices, or NULL.
 * No accepted model and no re-estimation => identity, no inliers (the reference's model is then uninitialised). */
int cilhip_transform_ransac3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, const uint32_t* samples, uint64_t seed,
                              float max_residual, size_t target_inliers, size_t max_iter, int re_estimate, cilhip_transform_model* out,
                              float* residuals_out, uint32_t* inliers_out);
/* inlier counts of m given transforms (HOST, 16 * m floats, col-major 4x4 each), one pass over the pairs per 64 transforms */
int cilhip_transform_score3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, const float* transforms, size_t m,
                             float max_residual, uint32_t* counts_out);
/* estimateModel() over ALL pairs (ransac_transform_estimator.hpp:61-64): the closed-form rigid fit */
int cilhip_transform_fit3f(int device, const float* dst_xyz, const float* src_xyz, size_t n, int mem, float T_out[16]);

/* ---- next tier: k-NN (k > 1) and NormalEstimation ------------------------- */
/* KDTree<float,3,L2>::kNNSearch / kNNInRadiusSearch for a set of queries (core/kd_tree.hpp:216-256, :286-318;
 * result adaptor :63-109): the k nearest reference points of every query with d2 < max_sq_dist (strict; pass
 * INFINITY for a plain k-NN), ascending by (d2, index).  query_xyz == NULL: the reference points are the queries
 * (every point then finds itself first).  1 <= k <= 32.  All outputs are HOST arrays:
 * idx_out [n_query*k] (row per query, padded with 0xFFFFFFFF), d2_out [n_query*k] or NULL (padded with +inf),
 * counts_out [n_query] or NULL (neighbours found).  Neighbour sets, their order and the distances are the reference's bit for
 * bit -- among EXACTLY equal distances (inside a list and at its k-th place) the candidates the reference's kd-tree traversal meets
 * first, in that order (core/kd_tree.hpp:80-99 over nanoflann searchLevel): the search notices lists that hold equal distances or
 * whose k-th distance was met on a further point, builds the order tables of the reference's tree over the searched cloud
 * (csrc/tie_build.hip, on the device) the first time a call needs them, and searches again with them -- a cloud without exact ties never pays.
 * cilhip_knn_set_tie_rule (process-wide; also cilhip_normals_knn3f): 2 = that (default), 1 = tables built up front, 0 = lowest
 * index among equal distances (a brute-force argsort's order).  tests/test_gpu_parity.py: the reference's sensor frames and
 * lattices, index for index against the reference's own nanoflann knnSearch. */
int cilhip_knn_set_tie_rule(int rule);
int cilhip_knn3f(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem, size_t k,
                 float max_sq_dist, uint32_t* idx_out, float* d2_out, uint32_t* counts_out);
/* KDTree<float,3>::radiusSearch (core/kd_tree.hpp:251-282; RadiusSearchResultAdaptor :111-142): per query, EVERY target
 * point with squared distance < radius_sq (strict), ascending by distance; equal distances are ordered by index (the
 * reference leaves them as std::sort does).  query_xyz == NULL: the target points are the queries.
 * offsets_out: HOST n_query + 1 -- list of query i = [offsets[i], offsets[i+1]); *total_out_or_null = offsets[n_query].
 * idx_out / d2_out: HOST `capacity` entries (d2_out may be NULL); when capacity < total (or idx_out == NULL) only the
 * offsets and the total are produced -- call again with enough room.  radius_sq must be finite. */
int cilhip_radius_search3f(int device, const float* ref_xyz, size_t n_ref, const float* query_xyz, size_t n_query, int mem,
                           float radius_sq, uint64_t* offsets_out, uint32_t* idx_out, float* d2_out, size_t capacity,
                           size_t* total_out_or_null);
/* NormalEstimation<float,3>::estimateNormalsAndCurvatureKNN / ...KNNInRadius (core/normal_estimation.hpp:72-90,
 * :166-187 -> :362-420): per point the k-NN neighbourhood (including the point itself), mean and covariance of it
 * (core/covariance.hpp:140-170), normal = eigenvector of the smallest eigenvalue, curvature = l_min / (l0+l1+l2);
 * fewer than 3 neighbours => NaN.  view_point: 3 floats; if all finite the normal is flipped to point towards it
 * (:326-330), NULL / non-finite: sign left as the eigen-solver gives it (as in the reference).
 * normals_out: HOST 3*n, curvature_out: HOST n or NULL.  f32 per-term arithmetic, f64 accumulation and f64 Jacobi
 * eigen-solve in place of Eigen's f32 SelfAdjointEigenSolver. */
int cilhip_normals_knn3f(int device, const float* xyz, size_t n, int mem, size_t k, float max_sq_dist, const float* view_point,
                         float* normals_out, float* curvature_out);
/* ...Radius (core/normal_estimation.hpp:120-162, RadiusNeighborhoodSpecification): every point with d2 < radius_sq
 * (strict) takes part; the neighbourhood is unbounded, its moments are accumulated without listing it. */
int cilhip_normals_radius3f(int device, const float* xyz, size_t n, int mem, float radius_sq, const float* view_point,
                            float* normals_out, float* curvature_out);

/* ---- introspection (bench / tests) ----------------------------------------------------------- */
typedef struct {
  int nx, ny, nz;        /* grid dims */
  float cell;            /* cell edge */
  float origin[3];
  size_t n_cells;
  double avg_occupancy;  /* sum(count^2)/n: expected own-cell candidates of a target point */
  double build_ms;       /* last set_target wall time (upload + grid build) */
} cilhip_grid_info;
int cilhip_get_grid_info(cilhip_ctx* ctx, cilhip_grid_info* out);
/* How many source points have, under T and within max_sq_dist, a nearest target point that is NOT unique in the pinned f32 squared
 * distance (exactly equidistant candidates: duplicated points, a depth sensor's lattice).  Only there do a brute-force argmin
 * (lowest target index) and the reference differ: nanoflann keeps the candidate its traversal meets first (core/kd_tree.hpp:82-90) --
 * both exact nearest neighbours; option "tie_rule" says which one the engine names.  Diagnostic (one more exact search of every query). */
int cilhip_get_tie_count(cilhip_ctx* ctx, const float T[16], float max_sq_dist, size_t* n_ties);
/* Option "tie_rule" != 0: of the last cilhip_find_correspondences (or summed over the searches of the last cilhip_icp_run / since
 * the last cilhip_icp_begin), the queries that had several exactly equidistant nearest target points, and how many of their
 * matches are NOT the lowest index (the reference's traversal met another one first). */
int cilhip_get_tie_rule_stats(cilhip_ctx* ctx, size_t* tied_queries, size_t* repointed);
/* The order tables behind "tie_rule" (kd_tree.hpp:162-170 -> nanoflann.hpp:1150-1212, :1321-1428: the permutation and the splits of
 * the index the reference builds over the target), built ON THE DEVICE (csrc/tie_build.hip: all nodes of a level are independent
 * segments -- a segmented min / max, two flagged prefix sums and two scatters per level reproduce planeSplit's two-pointer sweeps slot
 * for slot).  loaded: this context's target has them on the device; builds: how often this context built them (tie_rule 2 builds
 * when a search first meets a tie; never, on clouds that do not tie); build_ms: wall time of the last build; pending: tied queries the searches since the last
 * cilhip_icp_begin / find_correspondences met WITHOUT tables (cilhip_icp_run and cilhip_find_correspondences deal with those
 * themselves; a caller driving cilhip_icp_begin / _partial_sums / _apply_sums reads it after its loop, calls
 * cilhip_build_tie_order and runs the loop again -- cilantro_amd/distributed.py does). */
typedef struct cilhip_tie_order_info { int loaded; int builds; double build_ms; size_t pending; } cilhip_tie_order_info;
int cilhip_get_tie_order_info(cilhip_ctx* ctx, cilhip_tie_order_info* out);
/* Builds and loads the tables for this context's own target now (no-op when loaded). */
int cilhip_build_tie_order(cilhip_ctx* ctx);
/* The same tables for a target that is only PART of the cloud the reference would index (a spatial slab of a sharded run): the
 * order is a property of the WHOLE cloud.  cilhip_tie_order_create builds it once from the whole cloud (host memory, original
 * order; built on the calling thread's current HIP device -- CILHIP_ERR_NO_DEVICE without one -- and kept on the host);
 * cilhip_load_tie_order hands a context the entries of its own points: global_index[i] = index in the whole cloud of the
 * context's target point i (null: the context holds the whole cloud).  cilhip_tie_order_tables copies the tables out (tests,
 * tools: leaf and slot by original index, 16-byte node records {parent, (depth << 3) | (dimension << 1) | second child, divlow,
 * divhigh}; any pointer may be null). */
typedef struct cilhip_tie_order cilhip_tie_order;
int cilhip_tie_order_create(const float* xyz, size_t n, cilhip_tie_order** out);
void cilhip_tie_order_destroy(cilhip_tie_order* order);
int cilhip_tie_order_tables(const cilhip_tie_order* order, uint32_t* leaf_by_index, uint32_t* slot_by_index, void* nodes_out, size_t nodes_cap,
                            size_t* n_nodes, int* max_depth);
int cilhip_load_tie_order(cilhip_ctx* ctx, const cilhip_tie_order* order, const uint32_t* global_index);

/* CorrespondenceSearchCombinedMetricCombiner (registration/correspondence_search_combined_metric_combiner.hpp:8-81): the combined
 * metric's point-to-point terms read ONE engine's correspondence set, its point-to-plane terms ANOTHER's (own radius, feature
 * adaptors, post-filters) over the same two clouds.  Both contexts hold the same target / source and SECOND_TO_FIRST matches
 * found under the same transform (cilhip_find_correspondences on each); ctx_point == ctx_plane is the single-engine case.
 * cilhip_estimate_combined_two_sets = estimateTransformCombinedMetric with its two set arguments
 * (registration/transform_estimation.hpp:237-367); cilhip_icp_run_two_sets = the ICP loop of
 * icp_single_transform_combined_metric.hpp:169-217 with that engine (host-driven: both searches, the estimate, compose, test). */
int cilhip_estimate_combined_two_sets(cilhip_ctx* ctx_point, cilhip_ctx* ctx_plane, float w_p2p, float w_p2pl, size_t max_iter, float conv_tol,
                                      float dT[16], int* converged);
int cilhip_icp_run_two_sets(cilhip_ctx* ctx_point, float max_sq_point, cilhip_ctx* ctx_plane, float max_sq_plane, const cilhip_icp_params* p,
                            const float* T0, cilhip_icp_result* out);

/* ---- one process, several devices (devices[], one stream per device) ---------------------------------------
 * The sharded ICP loop driven from C: one context per entry of devices[], per iteration every context enqueues its partial
 * sums, the 48 f64 are all-reduced on the devices' streams by RCCL (ncclAllReduce; librccl is opened at run time, only when
 * ndev > 1) and every context applies them -- the reference has no counterpart; the loop is
 * IterativeClosestPointBase::estimate (registration/icp_base.hpp:68-87).  devices[] may repeat ONE ordinal (several shards on one
 * GPU, reduced by a kernel: what a single-GPU box can test).
 * cilhip_multi_set_clouds copies the host clouds (they are cut again when a slab guard fires) and uploads every shard:
 * partition 0 = the source in contiguous shards, the whole target on every device; 1 = spatial slabs along the longest axis of the
 * target's bounding box, target slabs with a halo of sqrt(max_sq_dist) + slack (slack = 2 sqrt(max_sq_dist)), source points by the
 * slab their image under T_part (null: identity; cilhip_multi_icp_run re-cuts under its T0) falls into, the device-side guard
 * armed (cilhip_set_slab_guard) and handled inside cilhip_multi_icp_run: all shards re-partitioned under the last exact
 * transform, no exact iteration discarded; 2 = the TARGET in contiguous index shards, the whole source on every device (SURVEY.md 8(e)
 * partitioning A: for a target that does not fit one device): per iteration an all-reduce(MIN) of one packed (d2, global index) key
 * per source point (ncclUint64 / ncclMin over RCCL), every shard accumulates the pairs it won, then the all-reduce of the sums; when
 * the searches meet exactly equidistant nearest points (inside a shard or across shards) the whole target's tie order is built once
 * and the run repeated with a second MIN per iteration (cilhip_icp_order_keys): the reference's matches, index for index.
 * Engine options are set per shard through cilhip_multi_context(rank). */
typedef struct cilhip_multi cilhip_multi;
int cilhip_multi_create(cilhip_multi** out, const int* devices, int ndev);
void cilhip_multi_destroy(cilhip_multi* m);
const char* cilhip_multi_last_error(const cilhip_multi* m);
cilhip_ctx* cilhip_multi_context(cilhip_multi* m, int rank);
int cilhip_multi_set_clouds(cilhip_multi* m, const float* dst_xyz, const float* dst_nrm_or_null, size_t n_dst, const float* src_xyz, size_t n_src,
                            float max_sq_dist, int partition, const float* T_part_or_null);
int cilhip_multi_icp_run(cilhip_multi* m, const cilhip_icp_params* p, const float* T0, int check_every, cilhip_icp_result* out);
/* Host time the shards' enqueue calls took per iteration of the last cilhip_multi_icp_run (the slowest shard's thread: the run has one
 * host thread per shard; CILHIP_MULTI_THREADS=0 in the environment: one thread walks the shards, and the figure is the mean per shard). */
int cilhip_multi_last_host_time(const cilhip_multi* m, double* us_per_iteration_per_shard);
int cilhip_multi_repartitions(const cilhip_multi* m);
/* how far a source point may move along the slab axis before the slabs are cut again (the halos are the search radius + this);
 * < 0: the default, twice the search radius.  Takes effect at the next cilhip_multi_set_clouds. */
int cilhip_multi_set_slab_slack(cilhip_multi* m, float slack);
int cilhip_multi_shard_sizes(const cilhip_multi* m, int rank, size_t* n_target, size_t* n_source);

/* ---- one process PER device (torchrun, MPI): this context as one rank of an RCCL communicator -------------------------------
 * The sharded loop's inner triple (cilhip_icp_partial_sums, all-reduce of the 48 f64, cilhip_icp_apply_sums) for `iterations`
 * iterations inside one call, the all-reduce = ncclAllReduce on the context's stream (librccl opened at run time): per iteration
 * the host enqueues a handful of launches instead of three foreign-function calls and a framework collective.  Rank 0 creates
 * the id (cilhip_rank_comm_unique_id), the launcher carries its 128 bytes to every rank (torch.distributed.broadcast, MPI_Bcast,
 * a file), every rank calls cilhip_rank_comm_init(ctx, id, nranks, rank) -- collective, as ncclCommInitRank is.  Between
 * cilhip_icp_begin and cilhip_icp_state the caller alternates cilhip_icp_iterate_ranked(k) with whatever it checks every k
 * iterations (convergence, cilhip_get_slab_violation_state).  Every rank must ask for the same number of iterations.  The
 * reference has no counterpart; the loop is IterativeClosestPointBase::estimate (registration/icp_base.hpp:68-87). */
int cilhip_rank_comm_unique_id(unsigned char id_out[128]);
/* What cilhip_rank_comm_init can fail at on one rank alone (opening librccl, its buffer), beforehand: the ranks agree -- one MIN over the
 * launcher's channel -- before any of them enters the collective init (cilantro_amd/distributed.py init_rank_comm). */
int cilhip_rank_comm_prepare(cilhip_ctx* ctx);
int cilhip_rank_comm_init(cilhip_ctx* ctx, const unsigned char id[128], int nranks, int rank);
int cilhip_rank_comm_destroy(cilhip_ctx* ctx);
int cilhip_icp_iterate_ranked(cilhip_ctx* ctx, int iterations);
/* With kernel timing on (cilhip_enable_kernel_timing; sampled by option "kernel_timing_stride"): the time of the ranked loop's
 * ncclAllReduce ON THE STREAM -- hipEvents around the collective: launch of RCCL's kernel, the exchange over xGMI, the wait for the
 * slowest rank -- summed over the `timed` iterations since cilhip_icp_begin; valid after cilhip_icp_state.  (No reference
 * counterpart: what bench.py --gpus N reports as allreduce_us_per_iteration.) */
int cilhip_get_last_allreduce_timing(cilhip_ctx* ctx, double* total_ms, int* timed);
/* Host time the ranked loop's enqueue calls took per iteration since cilhip_icp_begin -- launches and the collective's enqueue, the
 * paced waits for the device's feedback word (the host stays at most two iterations ahead) excluded: what must stay below a rank's
 * iteration time for the host to be off the critical path (bench.py: host_enqueue_us_per_iteration_per_rank). */
int cilhip_get_last_host_enqueue_time(cilhip_ctx* ctx, double* us_per_iteration);

/* How the iterations of the last cilhip_icp_run were executed: as ONE pass (search with the accumulation inside the LDS
 * tiles) or as TWO (search with its in-tile 3x3x3 second pass, then the streaming accumulation).  Large clouds choose per
 * iteration from the device's count of queries the first search stage left unproven (source far from alignment: two passes).
 * Sharded runs: the same two counts over the cilhip_icp_partial_sums calls since cilhip_icp_begin. */
int cilhip_get_last_run_forms(cilhip_ctx* ctx, int* one_pass_iterations, int* two_pass_iterations);
/* ... and how many of the one-pass iterations ran as the WARM-STARTED per-lane kernel (option "warm_start"): from the second
 * iteration on, near alignment, the search starts from the previous iteration's match -- a real target point, so its
 * distance from the new query bounds the search -- and usually ends inside the query's own cell; same matches. */
int cilhip_get_last_warm_iterations(cilhip_ctx* ctx, int* warm_iterations);
/* The last cilhip_icp_run iteration by iteration (at most the first 256, at most `cap`; written by the device's epilogue, read
 * back here): queries the search's first stage left unproven, queries a warm-started iteration had to search, the bound on
 * how far any source point moved in the iteration's update (what the warm-started form's margins are spent on), the update
 * norm (last_delta_norm_ of icp_base.hpp:83) and the kernel form (the codes of cilhip_get_last_form_timing).  Any output
 * array may be null.  Diagnostics: nothing in the loop depends on it. */
int cilhip_get_last_run_trace(cilhip_ctx* ctx, int cap, int* n, unsigned int* unproven, unsigned int* listed, float* step, float* delta, int* form);
/* With kernel timing on: kernel time (hipEvents on the ctx stream) and launch count of the last run's iterations per FORM of
 * their search kernel -- 0: search alone (a streaming accumulation follows: cilhip_get_last_timing2), 1: LDS-tiled search with
 * the accumulation inside the tile, 2: the first warm-started iteration of a stretch (gathers through the stored matches and
 * writes the match records), 3: warm-started iterations reading the records, 4: the per-lane fused kernel (option "fused"). */
int cilhip_get_last_form_timing(cilhip_ctx* ctx, int form, double* kernel_ms, int* launches);
/* ... and iteration by iteration: which iterations of the last cilhip_icp_run carried events (option "kernel_timing_stride") and the time
 * of the search (+ accumulation) kernel(s) of each.  *n = how many (may exceed cap: the first cap are written). */
int cilhip_get_last_iteration_timing(cilhip_ctx* ctx, int cap, int* n, unsigned int* iteration, float* kernel_ms);

/* ms of the kernels of the last cilhip_icp_run, measured with hipEvents on the ctx stream:
 * total loop, and the fused search+accumulate kernel alone (sum over executed iterations). */
int cilhip_get_last_timing(cilhip_ctx* ctx, double* loop_ms, double* search_kernel_ms,
                           int* search_kernel_launches);
/* Record a hipEvent pair around every fused search+accumulate launch of cilhip_icp_run (off by
 * default: the extra event records perturb a back-to-back loop slightly). */
int cilhip_enable_kernel_timing(cilhip_ctx* ctx, int on);
/* Tuning knobs (never change results beyond f64 summation order):
 *   "fused" (default 0): 1 = one fused search+accumulate kernel per iteration,
 *                        0 = search kernel (stores the matches) + streaming accumulation kernel.
 *   "tiled" (default 1): LDS-tiled search kernel (one workgroup stages the cell region around an
 *                        12x12x12-cell cube of queries in LDS; tiles that do not fit fall back per tile):
 *                        0 = never (per-lane global-memory search), 1 = when the cloud is large enough
 *                        to fill the chip with tiles (>= 600 tiles, ~1M points), full enough tiles and a
 *                        target density that fits a tile's LDS budget, 2 = always.
 *   "tile_accumulation" (default 1): where the first Gauss-Newton step of an iteration is accumulated when the tiled search runs
 *                        without post-filters: 0 = always in a separate streaming pass, 1 = inside the LDS tiles (one pass per
 *                        iteration) unless the device reports the source far from alignment, 2 = always inside the tiles.
 *   "warm_start" (default 1): the warm-started iteration kernel (cilhip_get_last_warm_iterations): 0 = never, 1 = once the last
 *                        update moved no source point by more than "warm_enter_fraction" of a grid cell (and for as long as the
 *                        kernel settles most queries from their margins: it reports how many it had to search), 2 = from the
 *                        second iteration on.
 *   "warm_forecast" (default 1): the cold iterations of a run count the queries whose margin the next update is expected to spend
 *                        (or that leave without one); the warm-started form is entered only when that is at most an eighth of the
 *                        queries -- a pair whose matches lie far beyond the target's point spacing never pays for a try.  0 = enter
 *                        on the step alone (tests).
 *   "tie_rule" (default 2): which of several EXACTLY equidistant nearest target points a correspondence names.  The reference's
 *                        kd-tree search returns the candidate its traversal meets FIRST (core/kd_tree.hpp:82-90 over nanoflann
 *                        1.7.1 searchLevel), which depends on the tree it built (leaf size 10, core/kd_tree.hpp:162-170).
 *                        2 = that point, on the device, in every kernel form: each search notices when its smallest distance was
 *                        met on a second point and settles such a query from the order tables of the reference's tree (per point:
 *                        leaf + slot of the reference's permutation; per node: parent, depth, split -- csrc/tie_build.hip builds
 *                        them on the device, level by level, csrc/search_device.hpp tie_settle reads them).  The tables are built when a search first
 *                        MEETS a tie (that search / run is then executed once more): a target whose searches never tie never pays
 *                        for a tree, one that does pays once.  1 = the same choice, tables built before the first search.
 *                        0 = the lowest index (what a brute-force argmin gives).  Covers every search over point features:
 *                        SECOND_TO_FIRST, rigid and affine, sharded runs included (cilhip_load_tie_order; index shards of a target:
 *                        cilhip_icp_order_keys), and the reverse matches of FIRST_TO_SECOND / BOTH -- there the reference's tree is
 *                        over the TRANSFORMED SOURCE, a new one per search (correspondence_search_kd_tree.hpp:185-222): a reverse
 *                        search counts the target points with several exactly equidistant source points; when they are
 *                        systematic (at least 16 and one target point in 100 000: duplicated points, lattices; under 1: always,
 *                        from the start) that tree's tables are built (on the device) before every reverse search (2 ms for a
 *                        110k-point cloud, 25 ms at 10M points) and the loops run one search at a time -- for this source, until it is set
 *                        again (duplicated points and lattices tie under every transform / systematically).  Below that (the coincidence of
 *                        two f32 distances in a large random cloud: about one target point in ten million) rule 2 keeps the
 *                        lowest source index for those and reports them (cilhip_get_tie_rule_stats).  The 6-D / 9-D feature adaptors
 *                        (SECOND_TO_FIRST) follow the tree the reference builds over the target's FEATURES (DIM = 6 / 9: other
 *                        splits, other leaves; tables per feature weights, rebuilt when those change; tie_before_nd); their
 *                        reverse searches keep the lowest index (refused under 1).  tests/test_gpu_tie_rule.py: every
 *                        index of the reference's sensor frames, of clouds with doubled and tripled points and of lattices with
 *                        8-way ties equals nanoflann's, in every direction.
 *   "group_search" (default -1): the global-memory search with SEVERAL lanes per query (small clouds, sources far from alignment: one
 *                        lane per query leaves the chip idle behind chains of dependent trips -- the reference's 120k-point sensor
 *                        frames: 0.34 -> 0.12 ms per iteration).  G adjacent lanes share a query: the rows of the block around its
 *                        cell are dealt to them, each row clipped to the cells the ball of the best distance so far reaches, the
 *                        minimum key goes round the group, the previous iteration's match bounds the search.  Same keys, same tie
 *                        rule, same results.  -1 = the ICP loop decides per iteration (clouds the tiles do not take: always below
 *                        the warm-started form's floor of 65 536 points; above it while the cold kernels' forecast says most
 *                        queries are far from settled); 0 = never; 4, 8, 16, 32, 64 = that many lanes in every such search.
 *   "fused_epilogue" (default 0): 1 = the stage-1 reduction of the partial sums and the epilogue run as ONE launch (the block that takes
 *                        the last ticket of the 32 stage-1 blocks runs the epilogue behind a device-scope fence).  Bitwise the same
 *                        results; measured slower than the two launches on this eight-L2 part (0.129 -> 0.136 ms per iteration at
 *                        10M, 0.037 -> 0.044 at 1M: NOTEBOOK.md) -- kept for A/B runs.
 *   "warm_extra_fraction" (default 0.0625): a query the warm-started form has to search is searched inside the ball of its bound
 *                        plus this fraction of a grid cell -- the room its fresh margin can have.  Larger: more cells per
 *                        search, margins that last longer; measured best at 10M (independent source: 0.25 -> 0.216 ms per
 *                        warm-started iteration, 0.0625 -> 0.197, 0.03 -> 0.199).  Never changes a result.
 *   "kernel_timing_stride" (default 1): with kernel timing on (cilhip_enable_kernel_timing), iterations 0, 1, 2 and every stride-th one
 *                        carry events; the others run as they do without timing.  An event between two dependent kernels idles the
 *                        device for ~6 us: two per iteration are a tenth of a warm-started iteration at 10M.  The per-form averages
 *                        of cilhip_get_last_form_timing are then over the timed launches (its `launches` = how many).
 *   "pair_records" (default 1): the streaming accumulation (second pass of a two-pass iteration, later Gauss-Newton steps) gathers a
 *                        match's point and normal from ONE 32-byte record instead of two arrays (a copy of the target in that layout,
 *                        32 B per point, built by the first run that needs it; without room for it the two arrays serve): 123.5 ->
 *                        113.5 us at 10M.  0 = the two arrays (A/B).  Same values, same sums.
 *   "tile_records" (default 1): the accumulating tile kernel writes the match records of the warm-started form itself (from a
 *                        run's second iteration on), so that the next iteration can read them; 0 = the first warm-started
 *                        iteration of a stretch gathers through the stored matches and writes them (A/B).
 *   "warm_enter_fraction" (default 0.15): that bar, as a fraction of a grid cell (halved each time a warm-started iteration of
 *                        the run had to search more than a quarter of its queries).
 *                        Neither option changes a result beyond the order of f64 additions.
 *   "refined_occupancy_factor" (default 3): a target whose density-based first guess of the cell size leaves far too many points per
 *                        cell (a surface, clusters: most cells of its bounding box are empty) has its grid refined until the expected
 *                        own-cell population is at most 3 x cell_occupancy x this factor.  1 = as fine a grid as a volumetric cloud
 *                        gets.  A search that has to leave the first block of cells walks shells of mostly empty cells on such a cloud,
 *                        and pays per cell: measured on the reference's sensor frames (120k points), frame_1 vs frame_2 (residuals of
 *                        several cells) 0.67 ms per iteration at factor 1, 0.30 at 3, 0.25 at 5; the near-aligned pair 0.033 / 0.036 /
 *                        0.045.  Used by the next cilhip_set_target; never changes a result.
 *   "cell_occupancy" (default 1): target points per grid cell, used by the next cilhip_set_target.
 *   "kernel_timing": same as cilhip_enable_kernel_timing.
 * Engine post-filters (correspondence_search_kd_tree.hpp:224-225, setInlierFraction / setOneToOne :253-271):
 *   "inlier_fraction" (default 1): in (0,1) keeps the llround(f*n) correspondences of smallest value
 *                        (core/correspondence.hpp:57-66; equal values: lowest source index first),
 *   "one_to_one" (default 0): 1 keeps, per target point, the correspondence of smallest value (:84-95).
 *   Applied to cilhip_find_correspondences and inside cilhip_icp_run; not available in sharded runs.
 * Search direction (correspondence_search_kd_tree.hpp:185-222, setSearchDirection / setRequireReciprocality :239-263):
 *   "search_direction" (default 0): 0 = SECOND_TO_FIRST (source points look for their nearest target point),
 *                        1 = FIRST_TO_SECOND (every target point looks for its nearest TRANSFORMED source point; like the
 *                        reference, an index over the transformed source is rebuilt for every search), 2 = BOTH (the
 *                        union of the two sets in (indexInFirst, indexInSecond) order, kd_tree_utilities.hpp:65-101),
 *   "require_reciprocality" (default 0): with BOTH, the intersection instead of the union.
 *   With 1 / 2 the correspondence set is a pair list (up to n_target + n_source entries): read it with
 *   cilhip_get_correspondences (cilhip_get_nn does not apply); the post-filters follow the reference's branches for
 *   those directions (one-to-one: per source point for FIRST_TO_SECOND, a no-op for BOTH, correspondence.hpp:72-98);
 *   cilhip_icp_run accumulates over the pair list.  Not available in sharded runs.
 *   "reverse_warm_start" (default 1): the device-resident loops of those directions (rigid start transform, no post-filters, plain
 *                        point features) start every reverse search but the first from the previous iteration's reverse matches: a
 *                        target point whose old match T s_i is nearer than half the distance from T s_i to its nearest other
 *                        transformed source point (a table over the source, built once) keeps it without looking at a cell; the rest
 *                        is searched as before -- the exact argmin either way (bidir.hip k_reverse_warm).  0 = every search from
 *                        scratch, for A/B runs.
 * Feature adaptor of the engine (correspondence_search/common_transformable_feature_adaptors.hpp):
 *   "feature_normal_weight" (default 0 = PointFeaturesAdaptor3f, :8-57): w > 0 = PointNormalFeaturesAdaptor3f (:60-161)
 *                        on both clouds -- features (p, w n), transformed as (T p, L (w n)), matched by the 6-D squared
 *                        distance (which is then also what max_sq_dist, the filters and the returned values refer to).
 *                        Needs target normals and source normals (cilhip_set_source_normals).  Every search direction; under
 *                        "transform_mode" = 1 the normal part follows the adaptor's non-rigid branch (:112-124):
 *                        normal_weight * (L^-T (w n)).normalized(), normal_weight = |w n_0| of the first source point.
 *                        Unsharded runs.
 *   "feature_kind" (default 0): 1 = the 6-D features are point + COLOUR, PointColorFeaturesAdaptor (:164-252): (p, w c) with
 *                        the colour part untouched by the transform; colours through cilhip_set_color_features, weight through
 *                        "feature_normal_weight".  2 = the 9-D point + NORMAL + COLOUR features, PointNormalColorFeaturesAdaptor
 *                        (:255-343): (p, wn n, wc c), matched by the 9-D squared distance (nanoflann's DIM = 9 summation order);
 *                        the normal part follows the transform as for kind 0, the colour part does not move; wn through
 *                        "feature_normal_weight", wc through "feature_color_weight"; needs both clouds' normals and colours.
 *   "feature_color_weight" (default 0): the colour weight of feature_kind 2.
 *   "feature_warm_start" (default 1): SECOND_TO_FIRST loops over features (rigid classes, no post-filters, >= 400 000 source points):
 *                        once an update moves no source point by more than the warm-started form's entry fraction of a cell, the
 *                        search starts from the previous matches -- a query whose old match p has 4 d_feat(q, p) < nnd(p)^2 (nnd: the
 *                        distance from p to its nearest other target point; d_feat >= the squared point distance) keeps it without looking
 *                        at a cell, the rest is searched in full (feat_warm.hip) -- and goes back to the tile search when more than a
 *                        quarter of the queries had to be searched.  With the three-cloud metric and unity evaluators the same pass also
 *                        accumulates the step's sums.  Same correspondences either way; 0 = every search from scratch (A/B).
 *   "symmetric_metric" (default 1): 0 = source normals feed the feature adaptor only and the combined metric stays the
 *                        three-cloud one (the reference decides this by the ICP constructor used,
 *                        icp_common_instances.hpp:74-97).  The symmetric objective runs the plain loop's kernel forms, the
 *                        warm-started one included (the queries' source normals streamed with them); the accumulating tiles and the
 *                        sharded building blocks keep the streaming pass for it.
 * Transform family (registration/icp_common_instances.hpp:253-267):
 *   "transform_mode" (default 0): 0 = rigid -- cilhip_icp_run is Simple{PointToPoint,Combined}MetricRigidICP3f;
 *                        1 = affine -- Simple{PointToPoint,Combined}MetricAffineICP3f: same loop and correspondence engine,
 *                        the step is cilhip_estimate_affine's closed form and there is no rotation() polish
 *                        (icp_single_transform_combined_metric.hpp:207-216); max_opt_iter / opt_conv_tol are unused,
 *                        as in the reference's affine overload.  Not available in sharded runs.
 *   "affine_device_loop" (default 1): the affine classes' loop runs device-resident like the rigid one whenever nothing needs the
 *                        stored correspondence set per iteration (SECOND_TO_FIRST, no post-filters, unity evaluators, point features):
 *                        search-only kernels + ONE streaming pass of the 112 moments on the matrix cores while the source is far from
 *                        alignment, search + moments in the warm-started kernel afterwards, the pivoted 12x12 LDL^T, the un-centring and
 *                        the f32 compose in the epilogue kernel -- no host round trip per iteration.  0 = the host-driven loop (three
 *                        moment passes + a host solve per iteration: what the other configurations run), for A/B runs.
 * Correspondence weight evaluators of the combined-metric classes (the PointToPoint/PointToPlaneCorrWeightEvaluatorT
 * template arguments of registration/icp_single_transform_combined_metric.hpp:11-14, core/common_pair_evaluators.hpp):
 *   "point_weight_evaluator", "plane_weight_evaluator" (default 0): 0 = UnityWeightEvaluator (:30-43),
 *                        1 = IdentityWeightEvaluator (:14-27: the weight is the correspondence's value, i.e. the squared
 *                        search distance -- 6-D with the point+normal features), 2 = RBFKernelWeightEvaluator over squared
 *                        distances (:46-80): exp(-0.5 / sigma^2 * value).
 *   "point_weight_sigma", "plane_weight_sigma" (default 1): the RBF evaluators' sigma (setSigma, :55-58).
 *                        The per-pair f32 weight is metric weight * evaluator(value) as transform_estimation.hpp:301-303,
 *                        :330-332 form it; exp() is a pinned f32 sequence (within 1 ulp of the correctly rounded value; the
 *                        reference's std::exp depends on its libm).  Rigid classes only (cilhip_icp_run, the sharded
 *                        building blocks, cilhip_estimate_combined); the accumulation then always runs as its own
 *                        streaming pass. */
int cilhip_set_option(cilhip_ctx* ctx, const char* key, double value);
/* The same options as a typed table: an enum a C caller can check at compile time, and per option its key, default, admissible
 * range and one line of documentation (the long form is the comment above).  cilhip_option_info(id) is valid for
 * 0 <= id < cilhip_option_count() == CILHIP_OPT_COUNT, in the enum's order (NULL otherwise); cilhip_set_option_id is
 * cilhip_set_option by id; cilhip_get_option reads the current value back.  tests/test_capi_symbols.py walks the table: every
 * option is documented, accepted with its default, readable, and named by at least one test. */
typedef enum cilhip_option {
  CILHIP_OPT_FUSED = 0, CILHIP_OPT_INLIER_FRACTION, CILHIP_OPT_ONE_TO_ONE, CILHIP_OPT_TILED, CILHIP_OPT_WARM_START, CILHIP_OPT_WARM_FORECAST,
  CILHIP_OPT_FUSED_EPILOGUE, CILHIP_OPT_GROUP_SEARCH, CILHIP_OPT_TIE_RULE, CILHIP_OPT_WARM_EXTRA_FRACTION, CILHIP_OPT_PAIR_RECORDS,
  CILHIP_OPT_TILE_RECORDS, CILHIP_OPT_WARM_ENTER_FRACTION, CILHIP_OPT_POINT_WEIGHT_EVALUATOR, CILHIP_OPT_PLANE_WEIGHT_EVALUATOR,
  CILHIP_OPT_POINT_WEIGHT_SIGMA, CILHIP_OPT_PLANE_WEIGHT_SIGMA, CILHIP_OPT_TILE_ACCUMULATION, CILHIP_OPT_SEARCH_DIRECTION,
  CILHIP_OPT_FEATURE_NORMAL_WEIGHT, CILHIP_OPT_FEATURE_KIND, CILHIP_OPT_FEATURE_COLOR_WEIGHT, CILHIP_OPT_SYMMETRIC_METRIC,
  CILHIP_OPT_TRANSFORM_MODE, CILHIP_OPT_REQUIRE_RECIPROCALITY, CILHIP_OPT_CELL_OCCUPANCY, CILHIP_OPT_REFINED_OCCUPANCY_FACTOR,
  CILHIP_OPT_KERNEL_TIMING, CILHIP_OPT_KERNEL_TIMING_STRIDE, CILHIP_OPT_REVERSE_WARM_START, CILHIP_OPT_FEATURE_WARM_START, CILHIP_OPT_AFFINE_DEVICE_LOOP,
  CILHIP_OPT_COUNT
} cilhip_option;
typedef struct cilhip_option_info_t {
  int id;                 /* enum cilhip_option */
  const char* key;        /* the name cilhip_set_option takes */
  double default_value, min_value, max_value;
  const char* doc;
} cilhip_option_info_t;
int cilhip_option_count(void);
const cilhip_option_info_t* cilhip_option_info(int id);
int cilhip_set_option_id(cilhip_ctx* ctx, cilhip_option id, double value);
int cilhip_get_option(cilhip_ctx* ctx, const char* key, double* value);
/* With "fused"=0 and kernel timing on: ms spent in the search kernels and in the accumulation
 * kernels of the last cilhip_icp_run (sum over executed iterations).  Sharded runs: the same two sums over the
 * cilhip_icp_partial_sums calls since cilhip_icp_begin, available after cilhip_icp_state (which synchronises);
 * cilhip_get_last_timing then reports the number of those calls as the launch count. */
int cilhip_get_last_timing2(cilhip_ctx* ctx, double* search_ms, double* accumulate_ms);
/* Tiled search bookkeeping of the most recent search launch: out[0] = queries, out[1] = whole tiles that were
 * handed to the global-memory clean-up pass (syncs). */
int cilhip_debug_counters(cilhip_ctx* ctx, uint32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
