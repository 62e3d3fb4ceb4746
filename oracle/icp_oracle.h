/*
 * icp_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the rigid-ICP hot path of
 * kzampog/cilantro.  Plain C restatement of the reference algorithm; every function cites the
 * reference file:line it follows (paths relative to /root/reference/include/cilantro/).
 *
 * THIS IS NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it -- and only as the checker.  The shipped path is the HIP library
 * (cilantro_amd/csrc -> libcilantro_hip.so); it never links or calls anything in oracle/.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - kNN half: PINNED against the reference's own vendored nanoflann 1.7.1 compiled from
 *     /root/reference into oracle/_ref/ (tests/test_oracle_cpu.py) and against the
 *     examples/kd_tree.cpp known answer.
 *   - estimator / solver half: the arithmetic lives in Eigen3 (>=3.3, unpinned, NOT vendored,
 *     absent from this container) => "parity unpinned" for JacobiSVD/LDLT/AngleAxis round-off;
 *     anchored analytically (numpy SVD/solve, convergence to ground truth on synthetic data).
 *
 * Numeric contract pinned here (and reproduced by the HIP kernels):
 *   d2(q,p)   = ((dx*dx) + (dy*dy)) + (dz*dz), dx = q.x - p.x, each op rounded to f32
 *               (3rd_party/nanoflann/nanoflann.hpp:570-604, DIM=3 takes the tail loop only)
 *   q = T*s   : q.x = (L00*x + (L01*y + L02*z)) + t0, each op rounded to f32
 *               (correspondence_search/common_transformable_feature_adaptors.hpp:28-33; pairing is
 *               Eigen's unrolled 3-term redux; build with -ffp-contract=off)
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kd-tree (restates nanoflann 1.7.1 KDTreeSingleIndexAdaptor, L2, DIM=3) ------------------ */
typedef struct orc_kdtree orc_kdtree;

/* core/kd_tree.hpp:162-170 (leaf_max=10, 1 build thread); nanoflann.hpp:1661-1687,1150-1212 */
orc_kdtree* orc_kdtree_build(const float* pts_xyz, size_t n, size_t leaf_max);
void orc_kdtree_free(orc_kdtree* t);

/* core/kd_tree.hpp:283-291 kNNInRadiusSearch (KNNSearchResultAdaptor :63-109).
 * radius is SQUARED.  Returns the number of neighbours found (<=k), sorted ascending,
 * first-encountered wins ties (strict '<' insert). */
size_t orc_kdtree_knn_in_radius(const orc_kdtree* t, const float q[3], size_t k, float radius_sq,
                                size_t* out_idx, float* out_d2);

/* ---- transform (common_transformable_feature_adaptors.hpp:28-33) ----------------------------- */
/* T: 4x4 column-major (Eigen::Transform<float,3,Isometry>::matrix().data()). */
void orc_transform_points(const float T[16], const float* src_xyz, size_t n, float* out_xyz);
/* core/space_transformations.hpp:374-390 (rigid): n' = L*n */
void orc_transform_normals(const float T[16], const float* nrm, size_t n, float* out);

/* ---- correspondence search (correspondence_search_kd_tree_utilities.hpp:7-51, ref_is_first) -- */
/* q: already transformed query points.  Outputs (capacity nq): ascending src index order.
 * Returns the number of correspondences kept (strict d2 < max_sq_dist). */
size_t orc_find_correspondences(const orc_kdtree* t, const float* q_xyz, size_t nq,
                                float max_sq_dist, int64_t* dst_idx, int64_t* src_idx, float* d2,
                                int num_threads);

/* Engine post-filters, applied in place in this order (correspondence_search_kd_tree.hpp:224-225).
 * filterCorrespondencesFraction (core/correspondence.hpp:57-66): keep the llround(f*n) smallest values; the
 * reference uses an unstable std::sort, so WHICH of several equal values survives is unspecified there --
 * pinned here (and in the HIP path) to: lowest source index first.  Result is ordered by (value, src).
 * filterCorrespondencesOneToOne, SECOND_TO_FIRST (:84-95): per target index keep the smallest value (ties:
 * lowest source index); result ordered by target index.  Both return the new count. */
size_t orc_filter_fraction(int64_t* dst_idx, int64_t* src_idx, float* d2, size_t n, double fraction);
size_t orc_filter_one_to_one(int64_t* dst_idx, int64_t* src_idx, float* d2, size_t n);

/* Exhaustive exact 1-NN-in-radius (validation of both the kd-tree restatement and the GPU grid):
 * argmin over all dst of the pinned d2 expression, strict '<' vs radius, LOWEST index on ties.
 * nn_idx[i] = -1 if none. */
/* queries whose nearest target point within the radius is not unique in the pinned f32 distance (exhaustive) */
size_t orc_count_ties_brute(const float* dst_xyz, size_t nd, const float* q_xyz, size_t nq, float max_sq_dist, int num_threads);
void orc_nn_brute(const float* dst_xyz, size_t nd, const float* q_xyz, size_t nq, float max_sq_dist,
                  int64_t* nn_idx, float* nn_d2, int num_threads);

/* ---- small dense solvers (Eigen-free restatements; double and float variants) ---------------- */
/* Two-sided Jacobi SVD of a 3x3 (row-major in/out), A = U diag(S) V^T, S >= 0 sorted descending.
 * Stands in for Eigen::JacobiSVD<Matrix3> (registration/transform_estimation.hpp:36-37). */
void orc_svd3_f64(const double A[9], double U[9], double S[3], double V[9]);
void orc_svd3_f32(const float A[9], float U[9], float S[3], float V[9]);
/* LDLT with diagonal pivoting + Eigen's pseudo-inverse-of-D solve (transform_estimation.hpp:346). */
void orc_ldlt6_solve_f64(const double A[36], const double b[6], double x[6]);
void orc_ldlt6_solve_f32(const float A[36], const float b[6], float x[6]);
/* core/space_transformations.hpp:43-51: nearest rotation, flips U.col(0) on det<0. Row-major. */
void orc_nearest_rotation_f64(const double L[9], double R[9]);
void orc_nearest_rotation_f32(const float L[9], float R[9]);

/* ---- correspondence weight evaluators (core/common_pair_evaluators.hpp:14-27, :30-43, :46-80) ---- */
enum { ORC_W_UNITY = 0, ORC_W_IDENTITY = 1, ORC_W_RBF = 2 };
typedef struct { int point_kind, plane_kind; float point_sigma, plane_sigma; } orc_weights;
/* exp() as ONE pinned sequence of f32 operations (the HIP path runs the same one): nearest-integer reduction by ln 2 in two
 * parts, degree-6 polynomial, exact scaling; 0 below -80.  tests/ hold it within 1 ulp of the correctly rounded value. */
float orc_pinned_expf(float x);

/* ---- estimators ------------------------------------------------------------------------------ */
/* mode: 0 = all-f32 serial ("reference-like": ENABLE_NON_DETERMINISTIC_PARALLELISM off),
 *       1 = per-term f32, accumulate/solve f64 (what the HIP path mirrors, deterministic),
 *       2 = all f64. */
enum { ORC_MODE_F32 = 0, ORC_MODE_MIXED = 1, ORC_MODE_F64 = 2 };

/* registration/transform_estimation.hpp:11-48 + :104-113 (gather by correspondences).
 * src_trans = already transformed source points.  T_out: 4x4 col-major float.
 * sums_out (optional, 16 doubles): n, sum_d(3), sum_s(3), sum_{d s^T}(9, row-major) --
 * raw (uncentred) moments, for checking the GPU accumulation kernel.
 * Returns 1 if n >= 3, 0 otherwise (identity on n == 0). */
int orc_estimate_p2p(const float* dst_xyz, const float* src_trans_xyz, const int64_t* dst_idx,
                     const int64_t* src_idx, size_t ncorr, int mode, float T_out[16],
                     double* sums_out);

/* registration/transform_estimation.hpp:237-367 (rigid, combined metric, 3D).
 * Unity weight evaluators.  AtA_out(36,row-major)/Atb_out(6): first Gauss-Newton step's normal
 * equations (optional).  Returns 1 if converged inside max_iter (d_theta.norm() < tol). */
/* src_nrm_trans (optional): transformed source normals => estimateTransformSymmetricMetric (:604-739). */
int orc_estimate_combined(const float* dst_xyz, const float* dst_nrm, const float* src_trans_xyz,
                          const float* src_nrm_trans, const int64_t* dst_idx, const int64_t* src_idx, size_t ncorr,
                          float w_p2p, float w_p2pl, size_t max_iter, float conv_tol,
                          const float dst_mean[3], const float src_mean[3], int mode,
                          float T_out[16], double* AtA_out, double* Atb_out);

/* OpenMP threads of the combined-metric estimators' accumulation loops: 1 (default) = serial; > 1 = the reference's default build
 * (ENABLE_NON_DETERMINISTIC_PARALLELISM, transform_estimation.hpp:284-344).  Used by the CPU baseline of bench.py only. */
void orc_set_estimator_threads(int n);

/* Two correspondence sets, as a CorrespondenceSearchCombinedMetricCombiner hands them over
 * (registration/correspondence_search_combined_metric_combiner.hpp:8-81): point terms from (dst_idx, src_idx, ncorr), plane terms
 * from (dst_idx_pl, src_idx_pl, ncorr_pl); unity evaluators. */
int orc_estimate_combined_two_sets(const float* dst_xyz, const float* dst_nrm, const float* src_trans_xyz, const int64_t* dst_idx,
                                   const int64_t* src_idx, size_t ncorr, const int64_t* dst_idx_pl, const int64_t* src_idx_pl, size_t ncorr_pl,
                                   float w_p2p, float w_p2pl, size_t max_iter, float conv_tol, const float dst_mean[3], const float src_mean[3],
                                   int mode, float T_out[16]);

/* The same with weight evaluators: val[k] = value of correspondence k (its search distance), wt = the evaluators. */
int orc_estimate_combined_w(const float* dst_xyz, const float* dst_nrm, const float* src_trans_xyz,
                            const float* src_nrm_trans, const int64_t* dst_idx, const int64_t* src_idx, size_t ncorr,
                            float w_p2p, float w_p2pl, size_t max_iter, float conv_tol,
                            const float dst_mean[3], const float src_mean[3], int mode,
                            float T_out[16], double* AtA_out, double* Atb_out, const float* val, const orc_weights* wt);

/* ---- whole ICP loop (registration/icp_base.hpp:68-87 + the two instance classes) ------------- */
typedef struct {
  int metric;            /* 0 = point-to-point (icp_single_transform_point_to_point_metric.hpp),
                            1 = combined (icp_single_transform_combined_metric.hpp) */
  float w_p2p, w_p2pl;   /* combined metric weights (defaults 0 / 1, :46-47) */
  size_t max_iter;       /* icp_base.hpp:24 default 15 */
  float conv_tol;        /* icp_base.hpp:25 default 1e-5 */
  size_t max_opt_iter;   /* combined: max_optimization_iterations_ (default 1) */
  float opt_conv_tol;    /* combined: optimization_convergence_tol_ (default 1e-5) */
  float max_sq_dist;     /* engine max_distance_ (squared; default 0.01*0.01) */
  int mode;              /* ORC_MODE_* */
  int num_threads;       /* OpenMP threads for the kNN loop */
  double inlier_fraction; /* engine inlier_fraction_ (filter active iff 0 < f < 1; correspondence.hpp:57-66) */
  int one_to_one;        /* engine one_to_one_ (correspondence.hpp:68-100) */
  int direction;         /* search_dir_: 0 = SECOND_TO_FIRST (the default), 1 = FIRST_TO_SECOND, 2 = BOTH */
  int reciprocal;        /* require_reciprocality_ (only read for BOTH) */
  int transform_mode;    /* 0 = rigid instances, 1 = affine (icp_common_instances.hpp:253-267) */
  float normal_weight;   /* > 0: the engine runs on PointNormalFeaturesAdaptor features (needs both clouds' normals); direction 0 */
  int three_cloud_metric; /* source normals feed the feature adaptor only (three-cloud ICP constructor): no symmetric metric */
  int point_weight_kind, plane_weight_kind;      /* ORC_W_*: the combined-metric classes' correspondence weight evaluators */
  float point_weight_sigma, plane_weight_sigma;  /* RBF evaluators' sigma */
} orc_icp_params;

typedef struct {
  float T[16];           /* final transform_, col-major */
  size_t iterations;     /* iterations_ */
  float last_delta_norm; /* last_delta_norm_ */
  size_t last_ncorr;     /* correspondences in the last iteration */
  double t_build_s, t_knn_s, t_est_s; /* wall-clock split (tree build once / kNN / estimate) */
} orc_icp_result;

/* KDTree::radiusSearch (core/kd_tree.hpp:251-282), exhaustive; see normals_oracle.c */
size_t orc_radius_search(const float* pts, size_t n, const float* q, size_t nq, float radius_sq, uint64_t* offsets, int64_t* idx,
                         float* d2, size_t cap);

/* 6-D point+normal features (common_transformable_feature_adaptors.hpp:60-161), row-major n x 6 */
void orc_point_normal_features(const float* pts, const float* nrm, size_t n, float normal_weight, float* out6);
void orc_transform_features6(const float T[16], const float* in6, size_t n, float* out6);
/* transformFeatures(tform) of the 6-D adaptors: mode 0 rigid point+normal, 1 affine point+normal, 2 point+colour */
void orc_transform_features6_mode(const float T[16], const float* in6, size_t n, int mode, float* out6);
size_t orc_find_correspondences_feat6_dir(const float* dst6, size_t nd, const float* q6, size_t ns, float max_d, int direction, int reciprocal,
                                          int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads);
/* 9-D point + normal + colour features (PointNormalColorFeaturesAdaptor, :255-343), row-major n x 9 */
void orc_point_normal_color_features(const float* pts, const float* nrm, const float* rgb, size_t n, float wn, float wc, float* out9);
void orc_transform_features9_mode(const float T[16], const float* in9, size_t n, int mode, float* out9);
size_t orc_find_correspondences_feat9(const float* dst9, size_t nd, const float* q9, size_t nq, float max_sq_dist,
                                      int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads);
size_t orc_find_correspondences_feat9_dir(const float* dst9, size_t nd, const float* q9, size_t ns, float max_d, int direction, int reciprocal,
                                          int64_t* di, int64_t* si, float* d2, int num_threads);
size_t orc_find_correspondences_feat6(const float* dst6, size_t nd, const float* q6, size_t nq, float max_sq_dist,
                                      int64_t* dst_idx, int64_t* src_idx, float* d2, int num_threads);

/* Affine closed form, transform_estimation.hpp:369-476 (and :50-102 with w_p2p = 1, w_p2pl = 0, zero means).
 * AtA_out (144) / Atb_out (12) optional.  Returns the reference's bool. */
int orc_estimate_affine(const float* dst_xyz, const float* dst_nrm_or_null, const float* src_xyz, const int64_t* dst_idx,
                        const int64_t* src_idx, size_t n, float w_p2p, float w_p2pl, const float dst_mean[3],
                        const float src_mean[3], int mode, float T_out[16], double* AtA_out, double* Atb_out);
/* ... with the correspondence weight evaluators of the combined-metric class (val = corr.value per pair; :432-434, :453-455) */
int orc_estimate_affine_w(const float* dst_xyz, const float* dst_nrm_or_null, const float* src_xyz, const int64_t* dst_idx,
                          const int64_t* src_idx, size_t n, float w_p2p, float w_p2pl, const float dst_mean[3],
                          const float src_mean[3], int mode, float T_out[16], double* AtA_out, double* Atb_out, const float* val,
                          const orc_weights* wt);

/* dst_nrm may be NULL for metric 0.  T0: initial transform (col-major) or NULL = identity.
 * tree: optional prebuilt kd-tree on dst (NULL = build here, as the engine does lazily). */
int orc_icp_run(const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz,
                const float* src_nrm_or_null /* 4-cloud ctor: symmetric metric */, size_t ns, const float* T0, const orc_icp_params* prm, const orc_kdtree* tree,
                orc_icp_result* out);

/* One ICP outer iteration given correspondences (used by tests to step GPU vs oracle in lockstep).
 * T_cur -> T_new, returns delta norm. */
float orc_icp_update(const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz,
                     const float* src_nrm_or_null, size_t ns, const float T_cur[16], const int64_t* dst_idx,
                     const int64_t* src_idx, size_t ncorr, const orc_icp_params* prm,
                     float T_new[16]);

/* ... over a Combiner's two sets (point terms / plane terms) */
float orc_icp_update_two_sets(const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz, size_t ns, const float T_cur[16],
                              const int64_t* dst_idx, const int64_t* src_idx, size_t ncorr, const int64_t* dst_idx_pl, const int64_t* src_idx_pl,
                              size_t ncorr_pl, const orc_icp_params* prm, float T_new[16]);
/* orc_icp_update with the correspondences' values (read by prm's weight evaluators; NULL = unity). */
float orc_icp_update_w(const float* dst_xyz, const float* dst_nrm, size_t nd, const float* src_xyz,
                       const float* src_nrm_or_null, size_t ns, const float T_cur[16], const int64_t* dst_idx,
                       const int64_t* src_idx, const float* val, size_t ncorr, const orc_icp_params* prm,
                       float T_new[16]);

/* rowwise().mean() as the reference ctor does (icp_single_transform_combined_metric.hpp:51-58):
 * f32 serial sum / n. */
void orc_mean3(const float* xyz, size_t n, int mode, float mean[3]);

/* other search directions (correspondence_search_kd_tree.hpp:185-222); output capacity nd + ns */
size_t orc_find_correspondences_dir(const float* dst, size_t nd, const orc_kdtree* dst_tree, const float* q, size_t ns, float max_d,
                                    int direction, int reciprocal, int64_t* di, int64_t* si, float* d2, int num_threads);
size_t orc_filter_fraction_lex(int64_t* di, int64_t* si, float* d2, size_t n, double f);
size_t orc_filter_one_to_one_f2s(int64_t* di, int64_t* si, float* d2, size_t n);

/* ---- kmeans_oracle.c: KMeans<float,3> brute-force path (clustering/kmeans.hpp:67-194) ---- */
size_t orc_kmeans_assign(const float* x, size_t n, const float* c, size_t k, int64_t* labels);
size_t orc_kmeans(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int64_t* labels);
/* the use_kd_tree = true branch (kmeans.hpp:86-94) */
size_t orc_kmeans_assign_kd(const float* x, size_t n, const float* c, size_t k, int64_t* labels);
size_t orc_kmeans_kd(const float* x, size_t n, float* c, size_t k, size_t max_iter, float tol, int mode, int64_t* labels);

/* ---- ransac_oracle.c: PlaneRANSACEstimator3f (model_estimation/ransac_base.hpp:64-131) ---- */
void orc_sym_eig3(const double A[9], double w[3], double V[9]);
void orc_plane_residuals(const float* pts, size_t n, const float plane[4], float* res);
size_t orc_plane_count_inliers(const float* pts, size_t n, const float plane[4], float thresh);
size_t orc_plane_count_inliers_mt(const float* pts, size_t n, const float plane[4], float thresh);   /* all host cores (OpenMP) */
void orc_plane_fit(const float* pts, const uint32_t* idx, size_t m, int mode, float plane[4]);
size_t orc_plane_ransac(const float* pts, size_t n, const uint32_t* samples, size_t max_iter, float thresh,
                        size_t target_inliers, int re_estimate, int mode, float plane[4], float* residuals,
                        uint32_t* inliers, size_t* n_inliers);

/* RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp) over point pairs; T col-major 4x4 */
void orc_transform_residuals(const float* dst, const float* src, size_t n, const float T[16], float* res);
size_t orc_transform_count_inliers(const float* dst, const float* src, size_t n, const float T[16], float thresh);
void orc_transform_fit(const float* dst, const float* src, const uint32_t* idx, size_t m, int mode, float T[16]);
size_t orc_transform_ransac(const float* dst, const float* src, size_t n, const uint32_t* samples, size_t max_iter, float thresh,
                            size_t target_inliers, int re_estimate, int mode, float T[16], float* residuals, uint32_t* inliers,
                            size_t* n_inliers, int* have_model);

/* ---- normals_oracle.c: k-NN batch + NormalEstimation (core/normal_estimation.hpp:294-420) ---- */
void orc_knn_batch(const orc_kdtree* t, const float* q, size_t nq, size_t k, float radius_sq, int64_t* idx, float* d2, uint32_t* cnt);
void orc_normals_radius(const float* pts, size_t n, float radius_sq, const float* view_point, int mode, float* normals, float* curvature);
void orc_normals_knn(const float* pts, size_t n, size_t k, float radius_sq, const float* view_point, int mode, float* normals,
                     float* curvature);

#ifdef __cplusplus
}
#endif
#endif
