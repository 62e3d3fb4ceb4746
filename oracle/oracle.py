"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  Nothing under ``cilantro_amd/`` imports it.

Two libraries:
  * ``oracle/liboracle.so``            -- plain-C restatement (``icp_oracle.c``)
  * ``oracle/_ref/libref_nanoflann.so`` -- the reference's own nanoflann 1.7.1, compiled from
    /root/reference by ``oracle/Makefile`` (prebuilt file travels to the GPU box)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_nanoflann.so")

MODE_F32, MODE_MIXED, MODE_F64 = 0, 1, 2
METRIC_P2P, METRIC_COMBINED = 0, 1


def build(force=False):
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    # make is incremental: liboracle.so rebuilds only when its sources changed; the `ref` target
    # rebuilds oracle/_ref only when /root/reference is present and keeps the prebuilt file otherwise.
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class IcpParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int), ("w_p2p", C.c_float), ("w_p2pl", C.c_float),
        ("max_iter", C.c_size_t), ("conv_tol", C.c_float), ("max_opt_iter", C.c_size_t),
        ("opt_conv_tol", C.c_float), ("max_sq_dist", C.c_float), ("mode", C.c_int),
        ("num_threads", C.c_int), ("inlier_fraction", C.c_double), ("one_to_one", C.c_int),
        ("direction", C.c_int), ("reciprocal", C.c_int), ("transform_mode", C.c_int),
        ("normal_weight", C.c_float), ("three_cloud_metric", C.c_int),
        ("point_weight_kind", C.c_int), ("plane_weight_kind", C.c_int),
        ("point_weight_sigma", C.c_float), ("plane_weight_sigma", C.c_float),
    ]


class Weights(C.Structure):
    _fields_ = [("point_kind", C.c_int), ("plane_kind", C.c_int), ("point_sigma", C.c_float), ("plane_sigma", C.c_float)]


W_UNITY, W_IDENTITY, W_RBF = 0, 1, 2


class IcpResult(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16), ("iterations", C.c_size_t), ("last_delta_norm", C.c_float),
        ("last_ncorr", C.c_size_t), ("t_build_s", C.c_double), ("t_knn_s", C.c_double),
        ("t_est_s", C.c_double),
    ]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [_f32p, C.c_size_t, C.c_size_t]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_knn_in_radius.restype = C.c_size_t
        L.orc_kdtree_knn_in_radius.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_float, _u64p, _f32p]
        L.orc_transform_points.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        L.orc_find_correspondences.restype = C.c_size_t
        L.orc_find_correspondences.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        L.orc_filter_fraction.restype = C.c_size_t
        L.orc_filter_fraction.argtypes = [_i64p, _i64p, _f32p, C.c_size_t, C.c_double]
        L.orc_filter_one_to_one.restype = C.c_size_t
        L.orc_filter_one_to_one.argtypes = [_i64p, _i64p, _f32p, C.c_size_t]
        L.orc_count_ties_brute.restype = C.c_size_t
        L.orc_count_ties_brute.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_int]
        L.orc_nn_brute.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _i64p, _f32p, C.c_int]
        L.orc_svd3_f64.argtypes = [_f64p] * 4
        L.orc_svd3_f32.argtypes = [_f32p] * 4
        L.orc_ldlt6_solve_f64.argtypes = [_f64p] * 3
        L.orc_ldlt6_solve_f32.argtypes = [_f32p] * 3
        L.orc_nearest_rotation_f64.argtypes = [_f64p] * 2
        L.orc_nearest_rotation_f32.argtypes = [_f32p] * 2
        L.orc_estimate_p2p.restype = C.c_int
        L.orc_estimate_p2p.argtypes = [_f32p, _f32p, _i64p, _i64p, C.c_size_t, C.c_int, _f32p, C.c_void_p]
        L.orc_radius_search.restype = C.c_size_t
        L.orc_radius_search.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _u64p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_point_normal_features.restype = None
        L.orc_point_normal_features.argtypes = [_f32p, _f32p, C.c_size_t, C.c_float, _f32p]
        L.orc_transform_features6.restype = None
        L.orc_transform_features6.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        L.orc_find_correspondences_feat6.restype = C.c_size_t
        L.orc_find_correspondences_feat6.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        L.orc_estimate_affine.restype = C.c_int
        L.orc_estimate_affine.argtypes = [_f32p, C.c_void_p, _f32p, _i64p, _i64p, C.c_size_t, C.c_float, C.c_float,
                                          _f32p, _f32p, C.c_int, _f32p, C.c_void_p, C.c_void_p]
        L.orc_estimate_combined.restype = C.c_int
        L.orc_transform_normals.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        L.orc_estimate_combined.argtypes = [_f32p, _f32p, _f32p, C.c_void_p, _i64p, _i64p, C.c_size_t, C.c_float,
                                            C.c_float, C.c_size_t, C.c_float, _f32p, _f32p, C.c_int,
                                            _f32p, C.c_void_p, C.c_void_p]
        L.orc_estimate_combined_w.restype = C.c_int
        L.orc_estimate_combined_w.argtypes = [_f32p, _f32p, _f32p, C.c_void_p, _i64p, _i64p, C.c_size_t, C.c_float,
                                              C.c_float, C.c_size_t, C.c_float, _f32p, _f32p, C.c_int,
                                              _f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_icp_update_w.restype = C.c_float
        L.orc_icp_update_w.argtypes = [_f32p, C.c_void_p, C.c_size_t, _f32p, C.c_void_p, C.c_size_t, _f32p, _i64p,
                                       _i64p, C.c_void_p, C.c_size_t, C.POINTER(IcpParams), _f32p]
        L.orc_icp_update_two_sets.restype = C.c_float
        L.orc_icp_update_two_sets.argtypes = [_f32p, _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p, _i64p, _i64p, C.c_size_t, _i64p, _i64p, C.c_size_t,
                                              C.POINTER(IcpParams), _f32p]
        L.orc_pinned_expf.restype = C.c_float
        L.orc_pinned_expf.argtypes = [C.c_float]
        L.orc_icp_run.restype = C.c_int
        L.orc_icp_run.argtypes = [_f32p, C.c_void_p, C.c_size_t, _f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.POINTER(IcpParams), C.c_void_p, C.POINTER(IcpResult)]
        L.orc_icp_update.restype = C.c_float
        L.orc_icp_update.argtypes = [_f32p, C.c_void_p, C.c_size_t, _f32p, C.c_void_p, C.c_size_t, _f32p, _i64p,
                                     _i64p, C.c_size_t, C.POINTER(IcpParams), _f32p]
        L.orc_mean3.argtypes = [_f32p, C.c_size_t, C.c_int, _f32p]
        L.orc_kmeans_assign.restype = C.c_size_t
        L.orc_kmeans_assign.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _i64p]
        L.orc_kmeans.restype = C.c_size_t
        L.orc_kmeans.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_int, _i64p]
        L.orc_kmeans_assign_kd.restype = C.c_size_t
        L.orc_kmeans_assign_kd.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _i64p]
        L.orc_kmeans_kd.restype = C.c_size_t
        L.orc_kmeans_kd.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_int, _i64p]
        L.orc_sym_eig3.argtypes = [_f64p, _f64p, _f64p]
        L.orc_plane_residuals.argtypes = [_f32p, C.c_size_t, _f32p, _f32p]
        L.orc_plane_count_inliers.restype = C.c_size_t
        L.orc_plane_count_inliers.argtypes = [_f32p, C.c_size_t, _f32p, C.c_float]
        L.orc_plane_count_inliers_mt.restype = C.c_size_t
        L.orc_plane_count_inliers_mt.argtypes = [_f32p, C.c_size_t, _f32p, C.c_float]
        L.orc_plane_fit.argtypes = [_f32p, C.c_void_p, C.c_size_t, C.c_int, _f32p]
        L.orc_plane_ransac.restype = C.c_size_t
        L.orc_plane_ransac.argtypes = [_f32p, C.c_size_t, _u32p, C.c_size_t, C.c_float, C.c_size_t, C.c_int, C.c_int,
                                       _f32p, _f32p, _u32p, C.POINTER(C.c_size_t)]
        L.orc_transform_residuals.argtypes = [_f32p, _f32p, C.c_size_t, _f32p, _f32p]
        L.orc_transform_count_inliers.restype = C.c_size_t
        L.orc_transform_count_inliers.argtypes = [_f32p, _f32p, C.c_size_t, _f32p, C.c_float]
        L.orc_transform_fit.argtypes = [_f32p, _f32p, C.c_void_p, C.c_size_t, C.c_int, _f32p]
        L.orc_transform_ransac.restype = C.c_size_t
        L.orc_transform_ransac.argtypes = [_f32p, _f32p, C.c_size_t, _u32p, C.c_size_t, C.c_float, C.c_size_t, C.c_int, C.c_int,
                                           _f32p, _f32p, _u32p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.orc_transform_features6_mode.argtypes = [_f32p, _f32p, C.c_size_t, C.c_int, _f32p]
        L.orc_point_normal_color_features.restype = None
        L.orc_point_normal_color_features.argtypes = [_f32p, _f32p, _f32p, C.c_size_t, C.c_float, C.c_float, _f32p]
        L.orc_transform_features9_mode.restype = None
        L.orc_transform_features9_mode.argtypes = [_f32p, _f32p, C.c_size_t, C.c_int, _f32p]
        L.orc_find_correspondences_feat9.restype = C.c_size_t
        L.orc_find_correspondences_feat9.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        L.orc_find_correspondences_feat9_dir.restype = C.c_size_t
        L.orc_find_correspondences_feat9_dir.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_int, C.c_int, _i64p, _i64p, _f32p, C.c_int]
        L.orc_find_correspondences_feat6_dir.restype = C.c_size_t
        L.orc_find_correspondences_feat6_dir.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_int, C.c_int, _i64p, _i64p, _f32p, C.c_int]
        L.orc_find_correspondences_dir.restype = C.c_size_t
        L.orc_find_correspondences_dir.argtypes = [_f32p, C.c_size_t, C.c_void_p, _f32p, C.c_size_t, C.c_float, C.c_int, C.c_int,
                                                   _i64p, _i64p, _f32p, C.c_int]
        L.orc_filter_fraction_lex.restype = C.c_size_t
        L.orc_filter_fraction_lex.argtypes = [_i64p, _i64p, _f32p, C.c_size_t, C.c_double]
        L.orc_filter_one_to_one_f2s.restype = C.c_size_t
        L.orc_filter_one_to_one_f2s.argtypes = [_i64p, _i64p, _f32p, C.c_size_t]
        L.orc_knn_batch.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, C.c_float, _i64p, _f32p, _u32p]
        L.orc_normals_radius.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_void_p, C.c_int, _f32p, _f32p]
        L.orc_normals_knn.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_int, _f32p, _f32p]
        _lib = L
    return _lib


def ref_available():
    return os.path.exists(_REF)


def ref():
    """The reference's own nanoflann (oracle/_ref). Raises if it was never built."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF):
            raise RuntimeError("oracle/_ref/libref_nanoflann.so missing (needs /root/reference at build time)")
        R = C.CDLL(_REF)
        R.ref_kdtree_build.restype = C.c_void_p
        R.ref_kdtree_build.argtypes = [_f32p, C.c_size_t]
        R.ref_kdtree_free.argtypes = [C.c_void_p]
        R.ref_kdtree_knn_in_radius.restype = C.c_size_t
        R.ref_kdtree_knn_in_radius.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_float, _u64p, _f32p]
        R.ref_kdtree_knn_batch.restype = None
        R.ref_kdtree_knn_batch.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, C.c_float, _i64p, _f32p, _u32p, C.c_int]
        R.ref_kdtree_radius_search.restype = C.c_size_t
        R.ref_kdtree_radius_search.argtypes = [C.c_void_p, _f32p, C.c_float, _u64p, _f32p, C.c_size_t]
        R.ref_find_correspondences6.restype = C.c_size_t
        R.ref_find_correspondences6.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        R.ref_find_correspondences9.restype = C.c_size_t
        R.ref_find_correspondences9.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        R.ref_find_correspondences.restype = C.c_size_t
        R.ref_find_correspondences.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_float, _i64p, _i64p, _f32p, C.c_int]
        _ref = R
    return _ref


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


class KDTree:
    """Oracle kd-tree (restatement) or the reference's nanoflann (``use_ref=True``)."""

    def __init__(self, pts, use_ref=False, leaf_max=10):
        self.pts = _c(pts).reshape(-1, 3)
        self.use_ref = use_ref
        if use_ref:
            self.h = ref().ref_kdtree_build(self.pts, len(self.pts))
        else:
            self.h = lib().orc_kdtree_build(self.pts, len(self.pts), leaf_max)

    def __del__(self):
        try:
            if self.h:
                (ref().ref_kdtree_free if self.use_ref else lib().orc_kdtree_free)(self.h)
                self.h = None
        except Exception:
            pass

    def knn_in_radius(self, q, k, radius_sq):
        q = _c(q).reshape(3)
        idx = np.zeros(k, np.uint64)
        d2 = np.zeros(k, np.float32)
        fn = ref().ref_kdtree_knn_in_radius if self.use_ref else lib().orc_kdtree_knn_in_radius
        n = fn(self.h, q, k, np.float32(radius_sq), idx, d2)
        return idx[:n].astype(np.int64), d2[:n]

    def find_correspondences(self, q, max_sq_dist, num_threads=0):
        """-> (dst_idx, src_idx, d2) in ascending src order (kd_tree_utilities.hpp:45-50)."""
        q = _c(q).reshape(-1, 3)
        n = len(q)
        di = np.zeros(max(n, 1), np.int64)
        si = np.zeros(max(n, 1), np.int64)
        d2 = np.zeros(max(n, 1), np.float32)
        fn = ref().ref_find_correspondences if self.use_ref else lib().orc_find_correspondences
        if self.use_ref and num_threads <= 0:
            num_threads = os.cpu_count() or 1
        c = fn(self.h, q, n, np.float32(max_sq_dist), di, si, d2, num_threads)
        return di[:c].copy(), si[:c].copy(), d2[:c].copy()


def transform_points(T, pts):
    """T: 4x4 (numpy, math layout). Pinned f32 expression."""
    pts = _c(pts).reshape(-1, 3)
    out = np.empty_like(pts)
    lib().orc_transform_points(T_to_colmajor(T), pts, len(pts), out)
    return out


def T_to_colmajor(T):
    return _c(np.asarray(T, np.float32).reshape(4, 4).T).reshape(16)


def T_from_colmajor(t16):
    return np.asarray(t16, np.float32).reshape(4, 4).T.copy()


def count_ties_brute(dst, q, max_sq_dist, num_threads=0):
    """queries whose nearest target point within the radius is not unique in the pinned f32 distance (exhaustive)"""
    dst = _c(dst).reshape(-1, 3); q = _c(q).reshape(-1, 3)
    return int(lib().orc_count_ties_brute(dst, len(dst), q, len(q), np.float32(max_sq_dist), num_threads))


def set_estimator_threads(n):
    """OpenMP threads of the combined-metric estimator's accumulation (1 = serial, the tests' order; > 1 = the reference's default reduction)"""
    lib().orc_set_estimator_threads(int(n))


def nn_brute(dst, q, max_sq_dist, num_threads=0):
    dst = _c(dst).reshape(-1, 3)
    q = _c(q).reshape(-1, 3)
    idx = np.zeros(len(q), np.int64)
    d2 = np.zeros(len(q), np.float32)
    lib().orc_nn_brute(dst, len(dst), q, len(q), np.float32(max_sq_dist), idx, d2, num_threads)
    return idx, d2


def svd3(A, dtype=np.float64):
    A = _c(A, dtype).reshape(9)
    U = np.zeros(9, dtype); S = np.zeros(3, dtype); V = np.zeros(9, dtype)
    (lib().orc_svd3_f64 if dtype == np.float64 else lib().orc_svd3_f32)(A, U, S, V)
    return U.reshape(3, 3), S, V.reshape(3, 3)


def ldlt6_solve(A, b, dtype=np.float64):
    A = _c(A, dtype).reshape(36); b = _c(b, dtype).reshape(6)
    x = np.zeros(6, dtype)
    (lib().orc_ldlt6_solve_f64 if dtype == np.float64 else lib().orc_ldlt6_solve_f32)(A, b, x)
    return x


def nearest_rotation(L, dtype=np.float64):
    L = _c(L, dtype).reshape(9)
    R = np.zeros(9, dtype)
    (lib().orc_nearest_rotation_f64 if dtype == np.float64 else lib().orc_nearest_rotation_f32)(L, R)
    return R.reshape(3, 3)


def estimate_p2p(dst, src_trans, dst_idx, src_idx, mode=MODE_MIXED):
    dst = _c(dst).reshape(-1, 3); src_trans = _c(src_trans).reshape(-1, 3)
    di = _c(dst_idx, np.int64); si = _c(src_idx, np.int64)
    T = np.zeros(16, np.float32); sums = np.zeros(16, np.float64)
    ok = lib().orc_estimate_p2p(dst, src_trans, di, si, len(di), mode, T, sums.ctypes.data)
    return T_from_colmajor(T), sums, bool(ok)


def transform_normals(T, nrm):
    nrm = _c(nrm).reshape(-1, 3)
    out = np.empty_like(nrm)
    lib().orc_transform_normals(T_to_colmajor(T), nrm, len(nrm), out)
    return out


def estimate_combined(dst, dst_n, src_trans, dst_idx, src_idx, w_p2p, w_p2pl, dst_mean, src_mean,
                      max_iter=1, conv_tol=1e-5, mode=MODE_MIXED, src_n_trans=None, values=None, weights=None):
    """values / weights=(point kind, plane kind, point sigma, plane sigma): the correspondences' values and the weight
    evaluators applied to them"""
    dst = _c(dst).reshape(-1, 3); dst_n = _c(dst_n).reshape(-1, 3)
    src_trans = _c(src_trans).reshape(-1, 3)
    di = _c(dst_idx, np.int64); si = _c(src_idx, np.int64)
    T = np.zeros(16, np.float32); AtA = np.zeros(36, np.float64); Atb = np.zeros(6, np.float64)
    sn = _c(src_n_trans).reshape(-1, 3) if src_n_trans is not None else None
    val = _c(values) if values is not None else None
    wt = Weights(int(weights[0]), int(weights[1]), float(weights[2]), float(weights[3])) if weights is not None else None
    ok = lib().orc_estimate_combined_w(dst, dst_n, src_trans, sn.ctypes.data if sn is not None else None, di, si, len(di), w_p2p, w_p2pl, max_iter,
                                       conv_tol, _c(dst_mean).reshape(3), _c(src_mean).reshape(3), mode,
                                       T, AtA.ctypes.data, Atb.ctypes.data, val.ctypes.data if val is not None else None,
                                       C.byref(wt) if wt is not None else None)
    return T_from_colmajor(T), AtA.reshape(6, 6), Atb, bool(ok)


def radius_search(pts, queries, radius_sq):
    """KDTree::radiusSearch, exhaustive -> (offsets int64 [nq+1], idx int64 [total], d2 f32 [total]); (d2, index) order."""
    pts = _c(pts).reshape(-1, 3); q = _c(queries).reshape(-1, 3)
    off = np.zeros(len(q) + 1, np.uint64)
    total = lib().orc_radius_search(pts, len(pts), q, len(q), float(radius_sq), off, None, None, 0)
    idx = np.zeros(max(total, 1), np.int64); d2 = np.zeros(max(total, 1), np.float32)
    lib().orc_radius_search(pts, len(pts), q, len(q), float(radius_sq), off, idx.ctypes.data, d2.ctypes.data, total)
    return off.astype(np.int64), idx[:total], d2[:total]


def ref_radius_search(tree, query, radius_sq, cap=100000):
    """One query through the reference's nanoflann (tree: KDTree(..., use_ref=True)) -> (idx, d2) sorted by distance."""
    idx = np.zeros(cap, np.uint64); d2 = np.zeros(cap, np.float32)
    n = ref().ref_kdtree_radius_search(tree.h, _c(query).reshape(3), float(radius_sq), idx, d2, cap)
    return idx[:n].astype(np.int64), d2[:n]


def point_normal_features(pts, nrm, w):
    """PointNormalFeaturesAdaptor ctor: [points; w * normals] -> (n, 6)"""
    pts = _c(pts).reshape(-1, 3); nrm = _c(nrm).reshape(-1, 3)
    out = np.empty((len(pts), 6), np.float32)
    lib().orc_point_normal_features(pts, nrm, len(pts), float(w), out.reshape(-1))
    return out


def transform_features6(T, feat6, mode=0):
    """transformFeatures(tform) of the 6-D adaptors: mode 0 rigid point+normal, 1 affine point+normal, 2 point+colour"""
    feat6 = _c(feat6).reshape(-1, 6)
    out = np.empty_like(feat6)
    lib().orc_transform_features6_mode(T_to_colmajor(T), feat6.reshape(-1), len(feat6), int(mode), out.reshape(-1))
    return out


def find_correspondences_feat6_dir(dst6, q6, max_sq_dist, direction, reciprocal=False, num_threads=0):
    """6-D feature correspondences in any search direction (exhaustive) -> (dst_idx, src_idx, d2)"""
    dst6 = _c(dst6).reshape(-1, 6); q6 = _c(q6).reshape(-1, 6)
    cap = len(dst6) + len(q6) + 1
    di = np.zeros(cap, np.int64); si = np.zeros(cap, np.int64); d2 = np.zeros(cap, np.float32)
    n = lib().orc_find_correspondences_feat6_dir(dst6.reshape(-1), len(dst6), q6.reshape(-1), len(q6), np.float32(max_sq_dist), int(direction),
                                                 1 if reciprocal else 0, di, si, d2, num_threads)
    return di[:n].copy(), si[:n].copy(), d2[:n].copy()


def point_normal_color_features(pts, nrm, rgb, normal_weight, color_weight):
    """PointNormalColorFeaturesAdaptor's data (common_transformable_feature_adaptors.hpp:251-258): n x 9 rows (p, wn n, wc c)"""
    pts = _c(pts).reshape(-1, 3); nrm = _c(nrm).reshape(-1, 3); rgb = _c(rgb).reshape(-1, 3)
    out = np.empty((len(pts), 9), np.float32)
    lib().orc_point_normal_color_features(pts.reshape(-1), nrm.reshape(-1), rgb.reshape(-1), len(pts), np.float32(normal_weight), np.float32(color_weight), out.reshape(-1))
    return out


def transform_features9(T, feat9, mode=0):
    """transformFeatures(tform) of the 9-D adaptor: mode 0 rigid, 1 otherwise (normals through L^-T, renormalised); colours copied"""
    feat9 = _c(feat9).reshape(-1, 9)
    out = np.empty_like(feat9)
    lib().orc_transform_features9_mode(T_to_colmajor(T), feat9.reshape(-1), len(feat9), int(mode), out.reshape(-1))
    return out


def find_correspondences_feat9_dir(dst9, q9, max_sq_dist, direction, reciprocal=False, num_threads=0):
    """9-D feature correspondences in any search direction (exhaustive) -> (dst_idx, src_idx, d2)"""
    dst9 = _c(dst9).reshape(-1, 9); q9 = _c(q9).reshape(-1, 9)
    cap = len(dst9) + len(q9) + 1
    di = np.zeros(cap, np.int64); si = np.zeros(cap, np.int64); d2 = np.zeros(cap, np.float32)
    n = lib().orc_find_correspondences_feat9_dir(dst9.reshape(-1), len(dst9), q9.reshape(-1), len(q9), np.float32(max_sq_dist), int(direction),
                                                 1 if reciprocal else 0, di, si, d2, num_threads)
    return di[:n].copy(), si[:n].copy(), d2[:n].copy()


def find_correspondences_feat9(dst9, q9, max_sq_dist, use_ref=False, num_threads=0):
    """Nearest 9-D feature per query (exhaustive restatement, or the reference's nanoflann for DIM = 9 with use_ref)."""
    dst9 = _c(dst9).reshape(-1, 9); q9 = _c(q9).reshape(-1, 9)
    nq = len(q9)
    di = np.empty(max(nq, 1), np.int64); si = np.empty(max(nq, 1), np.int64); dv = np.empty(max(nq, 1), np.float32)
    if use_ref:
        n = ref().ref_find_correspondences9(dst9.reshape(-1), len(dst9), q9.reshape(-1), nq, float(max_sq_dist), di, si, dv,
                                            num_threads if num_threads > 0 else (os.cpu_count() or 1))
    else:
        n = lib().orc_find_correspondences_feat9(dst9.reshape(-1), len(dst9), q9.reshape(-1), nq, float(max_sq_dist), di, si, dv, num_threads)
    return di[:n].copy(), si[:n].copy(), dv[:n].copy()


def find_correspondences_feat6(dst6, q6, max_sq_dist, use_ref=False, num_threads=0):
    """Nearest 6-D feature per query (exhaustive restatement, or the reference's nanoflann for DIM = 6 with use_ref)."""
    dst6 = _c(dst6).reshape(-1, 6); q6 = _c(q6).reshape(-1, 6)
    nq = len(q6)
    di = np.empty(max(nq, 1), np.int64); si = np.empty(max(nq, 1), np.int64); dv = np.empty(max(nq, 1), np.float32)
    if use_ref:
        n = ref().ref_find_correspondences6(dst6.reshape(-1), len(dst6), q6.reshape(-1), nq, float(max_sq_dist), di, si, dv,
                                            num_threads if num_threads > 0 else (os.cpu_count() or 1))
    else:
        n = lib().orc_find_correspondences_feat6(dst6.reshape(-1), len(dst6), q6.reshape(-1), nq, float(max_sq_dist), di, si, dv, num_threads)
    return di[:n].copy(), si[:n].copy(), dv[:n].copy()


def estimate_affine(dst, dst_n, src_trans, dst_idx, src_idx, w_p2p, w_p2pl, dst_mean, src_mean, mode=MODE_MIXED):
    """transform_estimation.hpp:369-476; the point-to-point overload :50-102 with w_p2p=1, w_p2pl=0 and zero means."""
    dst = _c(dst).reshape(-1, 3); src_trans = _c(src_trans).reshape(-1, 3)
    dn = _c(dst_n).reshape(-1, 3) if dst_n is not None else None
    di = _c(dst_idx, np.int64); si = _c(src_idx, np.int64)
    T = np.zeros(16, np.float32); AtA = np.zeros(144, np.float64); Atb = np.zeros(12, np.float64)
    ok = lib().orc_estimate_affine(dst, dn.ctypes.data if dn is not None else None, src_trans, di, si, len(di), w_p2p, w_p2pl,
                                   _c(dst_mean).reshape(3), _c(src_mean).reshape(3), mode, T, AtA.ctypes.data, Atb.ctypes.data)
    return T_from_colmajor(T), AtA.reshape(12, 12), Atb, bool(ok)


def mean3(pts, mode=MODE_MIXED):
    pts = _c(pts).reshape(-1, 3)
    m = np.zeros(3, np.float32)
    lib().orc_mean3(pts, len(pts), mode, m)
    return m


def make_params(metric=METRIC_COMBINED, w_p2p=0.0, w_p2pl=1.0, max_iter=15, conv_tol=1e-5,
                max_opt_iter=1, opt_conv_tol=1e-5, max_sq_dist=1e-4, mode=MODE_MIXED, num_threads=0,
                inlier_fraction=1.0, one_to_one=False, direction=0, reciprocal=False, affine=False,
                normal_weight=0.0, three_cloud_metric=False, point_weight=W_UNITY, plane_weight=W_UNITY,
                point_sigma=1.0, plane_sigma=1.0):
    """direction: 0 = SECOND_TO_FIRST (default), 1 = FIRST_TO_SECOND, 2 = BOTH; affine: the Affine ICP instances;
    point_weight / plane_weight: W_* correspondence weight evaluators of the combined-metric classes"""
    return IcpParams(metric, w_p2p, w_p2pl, max_iter, conv_tol, max_opt_iter, opt_conv_tol,
                     max_sq_dist, mode, num_threads, inlier_fraction, 1 if one_to_one else 0, int(direction), 1 if reciprocal else 0,
                     1 if affine else 0, float(normal_weight), 1 if three_cloud_metric else 0,
                     int(point_weight), int(plane_weight), float(point_sigma), float(plane_sigma))


def filter_fraction(dst_idx, src_idx, d2, fraction):
    di = _c(dst_idx, np.int64).copy(); si = _c(src_idx, np.int64).copy(); dv = _c(d2).copy()
    n = lib().orc_filter_fraction(di, si, dv, len(di), float(fraction))
    return di[:n], si[:n], dv[:n]


def filter_one_to_one(dst_idx, src_idx, d2):
    di = _c(dst_idx, np.int64).copy(); si = _c(src_idx, np.int64).copy(); dv = _c(d2).copy()
    n = lib().orc_filter_one_to_one(di, si, dv, len(di))
    return di[:n], si[:n], dv[:n]


def icp_run(dst, dst_n, src, params, T0=None, tree=None, src_n=None):
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3)
    dn = None
    if dst_n is not None:
        dn = _c(dst_n).reshape(-1, 3)
    res = IcpResult()
    t0 = T_to_colmajor(T0) if T0 is not None else None
    sn = _c(src_n).reshape(-1, 3) if src_n is not None else None
    lib().orc_icp_run(dst, dn.ctypes.data if dn is not None else None, len(dst), src,
                      sn.ctypes.data if sn is not None else None, len(src),
                      t0.ctypes.data if t0 is not None else None, C.byref(params),
                      tree.h if tree is not None else None, C.byref(res))
    return {
        "T": T_from_colmajor(np.array(res.T[:], np.float32)),
        "iterations": int(res.iterations),
        "last_delta_norm": float(res.last_delta_norm),
        "last_ncorr": int(res.last_ncorr),
        "t_build_s": res.t_build_s, "t_knn_s": res.t_knn_s, "t_est_s": res.t_est_s,
    }


def pinned_expf(x):
    x = _c(x).reshape(-1)
    f = lib().orc_pinned_expf
    return np.array([f(float(v)) for v in x], np.float32)


def icp_update(dst, dst_n, src, T_cur, dst_idx, src_idx, params, src_n=None, values=None):
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3)
    dn = _c(dst_n).reshape(-1, 3) if dst_n is not None else None
    di = _c(dst_idx, np.int64); si = _c(src_idx, np.int64)
    Tn = np.zeros(16, np.float32)
    sn = _c(src_n).reshape(-1, 3) if src_n is not None else None
    val = _c(values) if values is not None else None
    d = lib().orc_icp_update_w(dst, dn.ctypes.data if dn is not None else None, len(dst), src,
                               sn.ctypes.data if sn is not None else None, len(src),
                               T_to_colmajor(T_cur), di, si, val.ctypes.data if val is not None else None, len(di), C.byref(params), Tn)
    return T_from_colmajor(Tn), float(d)


def icp_update_two_sets(dst, dst_n, src, T_cur, dst_idx_pt, src_idx_pt, dst_idx_pl, src_idx_pl, params):
    """one ICP iteration over a Combiner's two correspondence sets (point terms from the first, plane terms from the second)"""
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3); dn = _c(dst_n).reshape(-1, 3)
    d1 = _c(dst_idx_pt, np.int64); s1 = _c(src_idx_pt, np.int64); d2 = _c(dst_idx_pl, np.int64); s2 = _c(src_idx_pl, np.int64)
    Tn = np.zeros(16, np.float32)
    d = lib().orc_icp_update_two_sets(dst, dn, len(dst), src, len(src), T_to_colmajor(T_cur), d1, s1, len(d1), d2, s2, len(d2), C.byref(params), Tn)
    return T_from_colmajor(Tn), float(d)


def kmeans_assign(x, centroids, labels=None, use_kd_tree=False):
    """clustering/kmeans.hpp:95-119 (use_kd_tree: :86-94, a kd-tree over the centroids) -> (labels int64, number changed)"""
    x = _c(x).reshape(-1, 3); c = _c(centroids).reshape(-1, 3)
    lab = np.zeros(len(x), np.int64) if labels is None else _c(labels, np.int64).copy()
    ch = (lib().orc_kmeans_assign_kd if use_kd_tree else lib().orc_kmeans_assign)(x, len(x), c, len(c), lab)
    return lab, int(ch)


def kmeans(x, centroids, max_iter=100, tol=np.finfo(np.float32).eps, mode=1, use_kd_tree=False):
    """KMeans<float,3>::cluster(centroids, max_iter, tol, use_kd_tree) -> (centroids, labels, iterations)"""
    x = _c(x).reshape(-1, 3); c = _c(centroids).reshape(-1, 3).copy()
    lab = np.zeros(len(x), np.int64)
    it = (lib().orc_kmeans_kd if use_kd_tree else lib().orc_kmeans)(x, len(x), c, len(c), max_iter, np.float32(tol), mode, lab)
    return c, lab, int(it)


# ---- plane RANSAC (model_estimation/ransac_hyperplane_estimator.hpp, ransac_base.hpp) ----------------
def sym_eig3(A):
    """symmetric 3x3 eigen-decomposition, PCA convention (descending, det fix) -> (w, V)"""
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    w = np.zeros(3); V = np.zeros(9)
    lib().orc_sym_eig3(A, w, V)
    return w, V.reshape(3, 3)


def plane_residuals(pts, plane):
    pts = _c(pts).reshape(-1, 3); r = np.zeros(len(pts), np.float32)
    lib().orc_plane_residuals(pts, len(pts), _c(plane).reshape(4), r)
    return r


def plane_count_inliers(pts, plane, thresh):
    pts = _c(pts).reshape(-1, 3)
    return int(lib().orc_plane_count_inliers(pts, len(pts), _c(plane).reshape(4), np.float32(thresh)))


def plane_count_inliers_mt(pts, plane, thresh):
    """the same count on all host cores (OpenMP)"""
    pts = _c(pts).reshape(-1, 3)
    return int(lib().orc_plane_count_inliers_mt(pts, len(pts), _c(plane).reshape(4), np.float32(thresh)))


def plane_fit(pts, idx=None, mode=1):
    """estimate_params_ (ransac_hyperplane_estimator.hpp:70-85) -> plane (nx, ny, nz, offset) f32"""
    pts = _c(pts).reshape(-1, 3); pl = np.zeros(4, np.float32)
    if idx is None:
        lib().orc_plane_fit(pts, None, len(pts), mode, pl)
    else:
        idx = np.ascontiguousarray(idx, np.uint32)
        lib().orc_plane_fit(pts, idx.ctypes.data, len(idx), mode, pl)
    return pl


def plane_ransac(pts, samples, thresh, target_inliers, max_iter=None, re_estimate=True, mode=1):
    """RandomSampleConsensusBase::estimate() -> (plane, residuals, inliers, iterations)"""
    pts = _c(pts).reshape(-1, 3); n = len(pts)
    samples = np.ascontiguousarray(samples, np.uint32).reshape(-1, 3)
    max_iter = len(samples) if max_iter is None else max_iter
    pl = np.zeros(4, np.float32); res = np.zeros(n, np.float32); inl = np.zeros(max(n, 1), np.uint32)
    k = C.c_size_t(0)
    it = lib().orc_plane_ransac(pts, n, samples, max_iter, np.float32(thresh), target_inliers, int(re_estimate), mode,
                                pl, res, inl, C.byref(k))
    return pl, res, inl[:k.value].copy(), int(it)


# ---- RigidTransformRANSACEstimator3f over point pairs (model_estimation/ransac_transform_estimator.hpp) ----------------------
def transform_residuals(dst, src, T):
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3); res = np.zeros(len(dst), np.float32)
    lib().orc_transform_residuals(dst, src, len(dst), T_to_colmajor(T), res)
    return res


def transform_count_inliers(dst, src, T, thresh):
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3)
    return int(lib().orc_transform_count_inliers(dst, src, len(dst), T_to_colmajor(T), np.float32(thresh)))


def transform_fit(dst, src, idx=None, mode=1):
    """estimateModel (ransac_transform_estimator.hpp:61-83) -> 4x4 rigid transform mapping src onto dst"""
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3); T = np.zeros(16, np.float32)
    if idx is None:
        lib().orc_transform_fit(dst, src, None, len(dst), mode, T)
    else:
        idx = np.ascontiguousarray(idx, np.uint32)
        lib().orc_transform_fit(dst, src, idx.ctypes.data, len(idx), mode, T)
    return T_from_colmajor(T)


def transform_ransac(dst, src, samples, thresh, target_inliers, max_iter=None, re_estimate=True, mode=1):
    """RandomSampleConsensusBase::estimate() -> (T 4x4, residuals, inliers, iterations, have_model bits)"""
    dst = _c(dst).reshape(-1, 3); src = _c(src).reshape(-1, 3); n = len(dst)
    samples = np.ascontiguousarray(samples, np.uint32).reshape(-1, 3)
    max_iter = len(samples) if max_iter is None else max_iter
    T = np.zeros(16, np.float32); res = np.zeros(max(n, 1), np.float32); inl = np.zeros(max(n, 1), np.uint32)
    k = C.c_size_t(0); have = C.c_int(0)
    it = lib().orc_transform_ransac(dst, src, n, samples if len(samples) else np.zeros((1, 3), np.uint32), max_iter, np.float32(thresh), target_inliers,
                                    int(re_estimate), mode, T, res, inl, C.byref(k), C.byref(have))
    return T_from_colmajor(T), res[:n], inl[:k.value].copy(), int(it), have.value


# ---- k-NN batch + NormalEstimation (core/kd_tree.hpp kNNSearch, core/normal_estimation.hpp) -------------
def knn_batch(tree, queries, k, radius_sq=np.inf):
    """tree: oracle.KDTree (restatement, not the _ref one) -> (idx int64 [nq,k] -1 padded, d2 [nq,k], counts)"""
    q = _c(queries).reshape(-1, 3)
    idx = np.zeros((len(q), k), np.int64); d2 = np.zeros((len(q), k), np.float32); cnt = np.zeros(len(q), np.uint32)
    lib().orc_knn_batch(tree.h, q, len(q), k, np.float32(radius_sq), idx.reshape(-1), d2.reshape(-1), cnt)
    return idx, d2, cnt


def ref_knn_batch(tree, queries, k, radius_sq=np.float32(np.finfo(np.float32).max), num_threads=0):
    """tree: oracle.KDTree(use_ref=True) -- the REFERENCE's own nanoflann knnSearch through cilantro's result adaptor
    -> (idx int64 [nq,k] -1 padded, d2 [nq,k], counts)"""
    assert tree.use_ref
    q = _c(queries).reshape(-1, 3)
    idx = np.zeros((len(q), k), np.int64); d2 = np.zeros((len(q), k), np.float32); cnt = np.zeros(len(q), np.uint32)
    ref().ref_kdtree_knn_batch(tree.h, q, len(q), k, np.float32(radius_sq), idx.reshape(-1), d2.reshape(-1), cnt, int(num_threads) if num_threads > 0 else (os.cpu_count() or 1))
    return idx, d2, cnt


def normals_knn(pts, k, radius_sq=np.inf, view_point=None, mode=1):
    pts = _c(pts).reshape(-1, 3)
    nrm = np.zeros((len(pts), 3), np.float32); cur = np.zeros(len(pts), np.float32)
    vp = None if view_point is None else np.ascontiguousarray(view_point, np.float32)
    lib().orc_normals_knn(pts, len(pts), k, np.float32(radius_sq), None if vp is None else vp.ctypes.data, mode, nrm.reshape(-1), cur)
    return nrm, cur


# ---- other search directions (correspondence_search_kd_tree.hpp:185-222) ---------------------------------
def find_correspondences_dir(dst, q, max_sq_dist, direction, reciprocal=False, inlier_fraction=1.0, one_to_one=False):
    """direction 0 = SECOND_TO_FIRST, 1 = FIRST_TO_SECOND, 2 = BOTH; q = already transformed source.
    -> (dst_idx, src_idx, d2) after the engine's post-filters, in the reference's order"""
    dst = _c(dst).reshape(-1, 3); q = _c(q).reshape(-1, 3)
    cap = len(dst) + len(q) + 1
    di = np.zeros(cap, np.int64); si = np.zeros(cap, np.int64); dv = np.zeros(cap, np.float32)
    tree = KDTree(dst) if direction != 1 else None
    n = lib().orc_find_correspondences_dir(dst, len(dst), tree.h if tree is not None else None, q, len(q), np.float32(max_sq_dist),
                                           int(direction), 1 if reciprocal else 0, di, si, dv, 0)
    if direction == 0:
        n = lib().orc_filter_fraction(di, si, dv, n, float(inlier_fraction))
        if one_to_one:
            n = lib().orc_filter_one_to_one(di, si, dv, n)
    else:
        n = lib().orc_filter_fraction_lex(di, si, dv, n, float(inlier_fraction))
        if one_to_one and direction == 1:
            n = lib().orc_filter_one_to_one_f2s(di, si, dv, n)
    return di[:n].copy(), si[:n].copy(), dv[:n].copy()


def normals_radius(pts, radius_sq, view_point=None, mode=1):
    pts = _c(pts).reshape(-1, 3)
    nrm = np.zeros((len(pts), 3), np.float32); cur = np.zeros(len(pts), np.float32)
    vp = None if view_point is None else np.ascontiguousarray(view_point, np.float32)
    lib().orc_normals_radius(pts, len(pts), np.float32(radius_sq), None if vp is None else vp.ctypes.data, mode, nrm.reshape(-1), cur)
    return nrm, cur
