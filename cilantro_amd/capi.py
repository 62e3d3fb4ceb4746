"""ctypes binding of libcilantro_hip.so (include/cilantro_hip/c_api.h) -- 1:1, no logic.

The library is the product; this module only loads it.  There is NO CPU fallback anywhere in
``cilantro_amd``: if the shared library is missing, ``load()`` raises; if no HIP device is usable,
``cilhip_create`` returns CILHIP_ERR_NO_DEVICE and :class:`Context` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CILHIP_LIB_PATH") or os.path.join(_HERE, "lib", "libcilantro_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = 0, -1, -2, -3, -4
MEM_HOST, MEM_DEVICE = 0, 1
METRIC_POINT_TO_POINT, METRIC_COMBINED = 0, 1
SUMS_LEN = 48
NONE_IDX = 0xFFFFFFFF

# every symbol include/cilantro_hip/c_api.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "cilhip_create", "cilhip_destroy", "cilhip_last_error", "cilhip_set_stream", "cilhip_synchronize",
    "cilhip_set_target", "cilhip_set_source", "cilhip_set_source_normals", "cilhip_get_means", "cilhip_set_color_features", "cilhip_share_target", "cilhip_find_correspondences",
    "cilhip_get_nn", "cilhip_get_correspondences", "cilhip_estimate_point_to_point",
    "cilhip_estimate_combined", "cilhip_estimate_affine", "cilhip_icp_default_params", "cilhip_icp_run", "cilhip_icp_begin",
    "cilhip_icp_partial_sums", "cilhip_icp_apply_sums", "cilhip_icp_state", "cilhip_compute_residuals",
    "cilhip_get_grid_info", "cilhip_get_last_timing", "cilhip_enable_kernel_timing", "cilhip_prepare_source", "cilhip_get_last_run_forms", "cilhip_get_last_warm_iterations", "cilhip_get_last_matches_origin", "cilhip_get_matches_transform", "cilhip_get_last_form_timing", "cilhip_get_last_iteration_timing", "cilhip_get_last_run_trace", "cilhip_estimate_combined_two_sets", "cilhip_icp_run_two_sets", "cilhip_set_pair_weight_callback", "cilhip_kmeans_set_pruning", "cilhip_get_tie_count", "cilhip_get_tie_rule_stats", "cilhip_get_tie_order_info", "cilhip_build_tie_order", "cilhip_tie_order_create", "cilhip_tie_order_destroy", "cilhip_tie_order_tables", "cilhip_load_tie_order", "cilhip_multi_create", "cilhip_multi_destroy", "cilhip_multi_last_error", "cilhip_multi_context", "cilhip_multi_set_clouds", "cilhip_multi_icp_run", "cilhip_multi_repartitions", "cilhip_multi_last_host_time", "cilhip_multi_set_slab_slack", "cilhip_multi_shard_sizes", "cilhip_rank_comm_unique_id", "cilhip_rank_comm_prepare", "cilhip_rank_comm_init", "cilhip_rank_comm_destroy", "cilhip_icp_iterate_ranked", "cilhip_get_last_allreduce_timing", "cilhip_get_last_host_enqueue_time", "cilhip_set_slab_guard", "cilhip_get_slab_violation", "cilhip_get_slab_violation_state",
    "cilhip_set_option", "cilhip_option_count", "cilhip_option_info", "cilhip_set_option_id", "cilhip_get_option", "cilhip_get_last_timing2", "cilhip_set_shard_info", "cilhip_icp_partial_keys",
    "cilhip_icp_sums_from_keys", "cilhip_icp_order_keys", "cilhip_icp_sums_from_ordered_keys", "cilhip_debug_counters", "cilhip_kmeans3f", "cilhip_kmeans3f_assign", "cilhip_kmeans3f_ex", "cilhip_kmeans3f_assign_ex", "cilhip_kmeans_shard_create", "cilhip_kmeans_shard_destroy", "cilhip_kmeans_shard_maxabs", "cilhip_kmeans_scale_exponent", "cilhip_kmeans_shard_assign", "cilhip_kmeans_shard_farthest", "cilhip_kmeans_shard_move_point", "cilhip_kmeans_shard_labels",
    "cilhip_plane_ransac3f", "cilhip_plane_score3f", "cilhip_plane_fit3f", "cilhip_transform_ransac3f", "cilhip_transform_score3f", "cilhip_transform_fit3f", "cilhip_knn3f", "cilhip_knn_set_tie_rule", "cilhip_normals_knn3f", "cilhip_normals_radius3f", "cilhip_radius_search3f",
]


# cilhip_pair_weight_fn: (user, index_in_first, index_in_second, value, n, point_weight_out, plane_weight_out)
PAIR_WEIGHT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_size_t,
                             C.POINTER(C.c_float), C.POINTER(C.c_float))


class IcpParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int), ("w_p2p", C.c_float), ("w_p2pl", C.c_float), ("max_iter", C.c_size_t),
        ("conv_tol", C.c_float), ("max_opt_iter", C.c_size_t), ("opt_conv_tol", C.c_float),
        ("max_sq_dist", C.c_float),
    ]


class IcpResult(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16), ("iterations", C.c_size_t), ("last_delta_norm", C.c_float),
        ("last_ncorr", C.c_size_t),
    ]


class GridInfo(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("cell", C.c_float),
        ("origin", C.c_float * 3), ("n_cells", C.c_size_t), ("avg_occupancy", C.c_double),
        ("build_ms", C.c_double),
    ]


class OptionInfo(C.Structure):
    _fields_ = [("id", C.c_int), ("key", C.c_char_p), ("default_value", C.c_double), ("min_value", C.c_double), ("max_value", C.c_double),
                ("doc", C.c_char_p)]


class TieOrderInfo(C.Structure):
    _fields_ = [("loaded", C.c_int), ("builds", C.c_int), ("build_ms", C.c_double), ("pending", C.c_size_t)]


class PlaneModel(C.Structure):
    _fields_ = [
        ("normal", C.c_float * 3), ("offset", C.c_float), ("iterations", C.c_size_t), ("n_inliers", C.c_size_t),
        ("target_reached", C.c_int), ("device_ms", C.c_double),
    ]


class TransformModel(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16), ("iterations", C.c_size_t), ("n_inliers", C.c_size_t), ("have_model", C.c_int),
        ("target_reached", C.c_int), ("device_ms", C.c_double),
    ]


_lib = None


def load():
    """Load libcilantro_hip.so.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m cilantro_amd.build` / `__graft_entry__.build()`). There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64 (same
    # SONAME as /opt/rocm's).  If torch is imported AFTER this library, two runtimes end up loaded and
    # the second one finds no device; importing torch first makes the dynamic loader bind this
    # library to the runtime torch already loaded (measured: tools/diag_runtime.py).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, f32p, f64p = C.c_void_p, C.c_void_p, C.c_void_p
    L.cilhip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.cilhip_destroy.argtypes = [vp]
    L.cilhip_destroy.restype = None
    L.cilhip_last_error.argtypes = [vp]
    L.cilhip_last_error.restype = C.c_char_p
    L.cilhip_set_stream.argtypes = [vp, vp]
    L.cilhip_synchronize.argtypes = [vp]
    L.cilhip_set_target.argtypes = [vp, f32p, f32p, C.c_size_t, C.c_int]
    L.cilhip_set_source.argtypes = [vp, f32p, C.c_size_t, C.c_int]
    L.cilhip_set_source_normals.argtypes = [vp, f32p, C.c_int]
    L.cilhip_get_means.argtypes = [vp, f32p, f32p]
    L.cilhip_find_correspondences.argtypes = [vp, f32p, C.c_float, C.POINTER(C.c_size_t)]
    L.cilhip_get_nn.argtypes = [vp, vp, f32p, C.c_int]
    L.cilhip_get_correspondences.argtypes = [vp, vp, vp, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cilhip_estimate_point_to_point.argtypes = [vp, f32p, f64p, C.POINTER(C.c_int)]
    L.cilhip_estimate_combined.argtypes = [vp, C.c_float, C.c_float, C.c_size_t, C.c_float, f32p, f64p,
                                           f64p, C.POINTER(C.c_int)]
    L.cilhip_estimate_affine.argtypes = [vp, C.c_float, C.c_float, C.c_int, f32p, f64p, f64p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.cilhip_icp_default_params.argtypes = [C.POINTER(IcpParams)]
    L.cilhip_icp_default_params.restype = None
    L.cilhip_icp_run.argtypes = [vp, C.POINTER(IcpParams), f32p, C.POINTER(IcpResult)]
    L.cilhip_icp_begin.argtypes = [vp, C.POINTER(IcpParams), f32p, f32p]
    L.cilhip_icp_partial_sums.argtypes = [vp, f64p]
    L.cilhip_icp_apply_sums.argtypes = [vp, f64p]
    L.cilhip_icp_state.argtypes = [vp, C.POINTER(IcpResult)]
    L.cilhip_compute_residuals.argtypes = [vp, C.c_int, C.c_float, C.c_float, f32p, f32p, C.c_int]
    L.cilhip_get_grid_info.argtypes = [vp, C.POINTER(GridInfo)]
    L.cilhip_get_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.cilhip_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.cilhip_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.cilhip_share_target.argtypes = [vp, vp]
    L.cilhip_option_count.argtypes = []
    L.cilhip_option_info.argtypes = [C.c_int]
    L.cilhip_option_info.restype = C.POINTER(OptionInfo)
    L.cilhip_set_option_id.argtypes = [vp, C.c_int, C.c_double]
    L.cilhip_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    L.cilhip_set_shard_info.argtypes = [vp, C.c_uint64, f32p, f32p]
    L.cilhip_icp_partial_keys.argtypes = [vp, vp]
    L.cilhip_debug_counters.argtypes = [vp, vp]
    L.cilhip_kmeans3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_float, vp, C.POINTER(C.c_size_t)]
    L.cilhip_kmeans3f_assign.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, vp]
    L.cilhip_kmeans3f_ex.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_int, vp, C.POINTER(C.c_size_t)]
    L.cilhip_kmeans3f_assign_ex.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, C.c_int, vp]
    L.cilhip_kmeans_shard_create.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_uint64, C.POINTER(C.c_void_p)]
    L.cilhip_kmeans_shard_destroy.argtypes = [vp]
    L.cilhip_kmeans_shard_destroy.restype = None
    L.cilhip_kmeans_shard_maxabs.argtypes = [vp, C.POINTER(C.c_float)]
    L.cilhip_kmeans_scale_exponent.argtypes = [C.c_double, C.c_size_t]
    L.cilhip_kmeans_shard_assign.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_uint64)]
    L.cilhip_kmeans_shard_farthest.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_uint64)]
    L.cilhip_kmeans_shard_move_point.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.cilhip_kmeans_shard_labels.argtypes = [vp, vp]
    L.cilhip_plane_ransac3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, vp, C.c_uint64, C.c_float, C.c_size_t, C.c_size_t,
                                        C.c_int, C.POINTER(PlaneModel), vp, vp]
    L.cilhip_plane_score3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, C.c_float, vp]
    L.cilhip_plane_fit3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, f32p]
    L.cilhip_transform_ransac3f.argtypes = [C.c_int, f32p, f32p, C.c_size_t, C.c_int, vp, C.c_uint64, C.c_float, C.c_size_t, C.c_size_t,
                                            C.c_int, C.POINTER(TransformModel), vp, vp]
    L.cilhip_transform_score3f.argtypes = [C.c_int, f32p, f32p, C.c_size_t, C.c_int, f32p, C.c_size_t, C.c_float, vp]
    L.cilhip_transform_fit3f.argtypes = [C.c_int, f32p, f32p, C.c_size_t, C.c_int, f32p]
    L.cilhip_knn3f.argtypes = [C.c_int, f32p, C.c_size_t, f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_float, vp, vp, vp]
    L.cilhip_normals_radius3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, C.c_float, f32p, f32p, f32p]
    L.cilhip_radius_search3f.argtypes = [C.c_int, f32p, C.c_size_t, f32p, C.c_size_t, C.c_int, C.c_float, vp, vp, vp, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
    L.cilhip_normals_knn3f.argtypes = [C.c_int, f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_float, f32p, f32p, f32p]
    L.cilhip_icp_sums_from_keys.argtypes = [vp, vp, f64p]
    L.cilhip_icp_order_keys.argtypes = [vp, vp, vp]
    L.cilhip_icp_sums_from_ordered_keys.argtypes = [vp, vp, vp, f64p]
    L.cilhip_get_last_timing2.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cilhip_prepare_source.argtypes = [vp, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.cilhip_get_last_run_forms.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cilhip_get_last_warm_iterations.argtypes = [vp, C.POINTER(C.c_int)]
    L.cilhip_set_color_features.argtypes = [vp, vp, vp, C.c_int]
    L.cilhip_get_last_matches_origin.argtypes = [vp, C.POINTER(C.c_int)]
    L.cilhip_get_matches_transform.argtypes = [vp, vp]
    L.cilhip_get_last_form_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.cilhip_estimate_combined_two_sets.argtypes = [vp, vp, C.c_float, C.c_float, C.c_size_t, C.c_float, vp, C.POINTER(C.c_int)]
    L.cilhip_icp_run_two_sets.argtypes = [vp, C.c_float, vp, C.c_float, C.POINTER(IcpParams), vp, C.POINTER(IcpResult)]
    L.cilhip_set_pair_weight_callback.argtypes = [vp, PAIR_WEIGHT_FN, vp]
    L.cilhip_get_tie_count.argtypes = [vp, vp, C.c_float, C.POINTER(C.c_size_t)]
    L.cilhip_get_last_iteration_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp, vp]
    L.cilhip_get_tie_rule_stats.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cilhip_get_tie_order_info.argtypes = [vp, C.POINTER(TieOrderInfo)]
    L.cilhip_build_tie_order.argtypes = [vp]
    L.cilhip_tie_order_create.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]
    L.cilhip_tie_order_tables.argtypes = [vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.cilhip_tie_order_destroy.argtypes = [vp]
    L.cilhip_tie_order_destroy.restype = None
    L.cilhip_load_tie_order.argtypes = [vp, vp, vp]
    L.cilhip_multi_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.cilhip_multi_destroy.argtypes = [vp]; L.cilhip_multi_destroy.restype = None
    L.cilhip_multi_last_error.argtypes = [vp]; L.cilhip_multi_last_error.restype = C.c_char_p
    L.cilhip_multi_context.argtypes = [vp, C.c_int]; L.cilhip_multi_context.restype = vp
    L.cilhip_multi_set_clouds.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_size_t, C.c_float, C.c_int, vp]
    L.cilhip_multi_icp_run.argtypes = [vp, C.POINTER(IcpParams), vp, C.c_int, C.POINTER(IcpResult)]
    L.cilhip_multi_repartitions.argtypes = [vp]
    L.cilhip_multi_set_slab_slack.argtypes = [vp, C.c_float]
    L.cilhip_multi_shard_sizes.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cilhip_rank_comm_unique_id.argtypes = [vp]
    L.cilhip_rank_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.cilhip_rank_comm_prepare.argtypes = [vp]
    L.cilhip_rank_comm_destroy.argtypes = [vp]
    L.cilhip_icp_iterate_ranked.argtypes = [vp, C.c_int]
    L.cilhip_get_last_allreduce_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.cilhip_get_last_host_enqueue_time.argtypes = [vp, C.POINTER(C.c_double)]
    L.cilhip_get_last_run_trace.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.cilhip_set_slab_guard.argtypes = [vp, C.c_int, C.c_float, f32p, f32p, f32p]
    L.cilhip_get_slab_violation.argtypes = [vp, C.POINTER(C.c_int)]
    L.cilhip_get_slab_violation_state.argtypes = [vp, C.POINTER(C.c_int), vp]
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the library does not export it
        if name not in ("cilhip_destroy", "cilhip_last_error", "cilhip_icp_default_params", "cilhip_option_info"):
            fn.restype = C.c_int
    _lib = L
    return L


class CilhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cilhip error {code}: {msg}")
        self.code = code


def options():
    """the option table of the library (cilhip_option_info): [{id, key, default, min, max, doc}]; needs no device"""
    L = load()
    out = []
    for i in range(L.cilhip_option_count()):
        o = L.cilhip_option_info(i).contents
        out.append({"id": o.id, "key": o.key.decode(), "default": o.default_value, "min": o.min_value, "max": o.max_value, "doc": o.doc.decode()})
    return out
