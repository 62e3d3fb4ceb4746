"""Python host-side mirror of cilantro's rigid ICP interface for this path, on top of the C ABI.

Same names, argument meaning and defaults as the reference classes so the parity tests read like
the reference's own usage (examples/rigid_icp.cpp:116-125):

    icp = SimpleCombinedMetricRigidICP3f(dst_points, dst_normals, src_points)
    icp.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0) \
       .setPointToPlaneMetricWeight(1.0)
    icp.correspondenceSearchEngine().setMaxDistance(0.1 * 0.1)
    icp.setConvergenceTolerance(1e-4).setMaxNumberOfIterations(30)
    T = icp.estimate().getTransform()

Mirrors: registration/icp_base.hpp (IterativeClosestPointBase), icp_common_instances.hpp:250,261
(SimplePointToPointMetricRigidICP3f / SimpleCombinedMetricRigidICP3f),
correspondence_search/correspondence_search_kd_tree.hpp (engine knobs :239-271).
Clouds: (N,3) float32 C-contiguous numpy arrays (== Eigen 3xN column-major) or CUDA torch tensors.
Transforms: 4x4 numpy float32 in math layout (row, col); converted to Eigen column-major at the ABI.

All compute happens in libcilantro_hip.so; nothing here falls back to the CPU.
"""
import ctypes as C
from enum import Enum

import numpy as np

from . import capi


class CorrespondenceSearchDirection(Enum):  # core/correspondence.hpp:7
    FIRST_TO_SECOND = 0
    SECOND_TO_FIRST = 1
    BOTH = 2


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _as_cloud(x):
    """-> (pointer, n, mem, keepalive)"""
    if x is None:
        return None, 0, capi.MEM_HOST, None
    if _is_torch(x):
        import torch

        t = x
        if t.dtype != torch.float32:
            raise TypeError("clouds must be float32")
        t = t.contiguous()
        if t.dim() != 2 or t.shape[1] != 3:
            raise ValueError("clouds must have shape (N, 3)")
        if t.is_cuda:
            return t.data_ptr(), t.shape[0], capi.MEM_DEVICE, t
        a = t.numpy()
        return a.ctypes.data, a.shape[0], capi.MEM_HOST, a
    a = np.ascontiguousarray(x, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("clouds must have shape (N, 3)")
    return a.ctypes.data, a.shape[0], capi.MEM_HOST, a


def _T_to_abi(T):
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T).reshape(16)
    return a


def _T_from_abi(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


class Context:
    """Thin RAII wrapper of cilhip_ctx."""

    def __init__(self, device=0, stream=None):
        self._L = capi.load()
        h = C.c_void_p()
        rc = self._L.cilhip_create(C.byref(h), int(device))
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_create failed (no usable HIP device? there is no CPU fallback)")
        self._h = h
        self._keep = []
        self._on_caller_stream = False
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cilhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        exc = getattr(self, "_weight_exc", None)
        if exc is not None:      # (a pair-weight callback raised inside the library call: ctypes cannot carry it through the C frames)
            self._weight_exc = None
            raise exc
        if rc != capi.OK:
            raise capi.CilhipError(rc, self._L.cilhip_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        """Run all work of this context on a caller-owned stream (e.g. ``torch.cuda.current_stream().cuda_stream``), so
        that it is ordered with the caller's own work (torch ops, RCCL collectives).  ``None``: back to the context's own
        stream.  Handle 0 -- what torch reports for its default stream -- means the LEGACY DEFAULT stream and is passed on
        as hipStreamLegacy (the C ABI reserves NULL for "the context's own stream")."""
        HIP_STREAM_LEGACY = 1   # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)
        if stream_ptr is None:
            h = 0
        else:
            h = int(stream_ptr) or HIP_STREAM_LEGACY
        self._on_caller_stream = stream_ptr is not None
        self._ck(self._L.cilhip_set_stream(self._h, C.c_void_p(h)))

    def synchronize(self):
        self._ck(self._L.cilhip_synchronize(self._h))

    def _settle_device_inputs(self, mem):
        """Device-resident inputs produced on the caller's (torch) stream: when this context runs on its own stream it is
        not ordered with that work, so wait for it once before the library reads the arrays."""
        if mem == capi.MEM_DEVICE and not self._on_caller_stream:
            import torch

            torch.cuda.current_stream().synchronize()

    def set_target(self, points, normals=None):
        p, n, mem, k1 = _as_cloud(points)
        q, nn, mem2, k2 = _as_cloud(normals)
        self._settle_device_inputs(mem)
        if normals is not None and (nn != n or mem2 != mem):
            raise ValueError("normals must match points (count and memory space)")
        self._ck(self._L.cilhip_set_target(self._h, p, q, n, mem))
        self.n_target = n

    def set_color_features(self, dst_colors, src_colors):
        """per-point colours of both clouds (cilhip_set_color_features): the attribute of feature_kind 1 / the second attribute of feature_kind 2"""
        d, nd, mem, _k1 = _as_cloud(dst_colors)
        s, ns, mem2, _k2 = _as_cloud(src_colors)
        if mem != mem2:
            raise ValueError("colours of both clouds must live in the same memory space")
        self._settle_device_inputs(mem)
        self._ck(self._L.cilhip_set_color_features(self._h, d, s, mem))

    def share_target(self, other):
        """CorrespondenceSearchKDTree::setFirstSearchTree(other.getFirstSearchTree()) (correspondence_search_kd_tree.hpp:273-296): search the
        index `other` built -- and everything built on top of it so far -- instead of building one; nothing is copied, either context may go first"""
        self._ck(self._L.cilhip_share_target(self._h, other._h))
        self.n_target = other.n_target

    def set_source(self, points, normals=None):
        p, n, mem, k = _as_cloud(points)
        self._settle_device_inputs(mem)
        self._ck(self._L.cilhip_set_source(self._h, p, n, mem))
        self.n_source = n
        if normals is not None:
            q, nn, mem2, k2 = _as_cloud(normals)
            if nn != n:
                raise ValueError("source normals must match source points")
            self._ck(self._L.cilhip_set_source_normals(self._h, q, mem2))

    def means(self):
        dm = np.zeros(3, np.float32); sm = np.zeros(3, np.float32)
        self._ck(self._L.cilhip_get_means(self._h, dm.ctypes.data, sm.ctypes.data))
        return dm, sm

    def find_correspondences(self, T, max_sq_dist, count=True):
        t = _T_to_abi(T)
        n = C.c_size_t(0)
        self._ck(self._L.cilhip_find_correspondences(self._h, t.ctypes.data, C.c_float(max_sq_dist),
                                                     C.byref(n) if count else None))
        return n.value if count else None

    def tie_count(self, T, max_sq_dist):
        """queries whose nearest target point (under T, within the radius) is not unique in the pinned f32 distance: where the
        engine's lowest-index rule and the reference's first-met rule may name different correspondences (cilhip_get_tie_count)"""
        t = _T_to_abi(T)
        n = C.c_size_t(0)
        self._ck(self._L.cilhip_get_tie_count(self._h, t.ctypes.data, C.c_float(max_sq_dist), C.byref(n)))
        return n.value

    def tie_rule_stats(self):
        """option tie_rule != 0: (tied queries, matches that are not the lowest index: the reference's traversal met another point first)
        of the last search / run"""
        a = C.c_size_t(0); b = C.c_size_t(0)
        self._ck(self._L.cilhip_get_tie_rule_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def tie_order_info(self):
        """the order tables behind option tie_rule (cilhip_get_tie_order_info): dict(loaded, builds, build_ms, pending)"""
        o = capi.TieOrderInfo()
        self._ck(self._L.cilhip_get_tie_order_info(self._h, C.byref(o)))
        return {"loaded": bool(o.loaded), "builds": int(o.builds), "build_ms": float(o.build_ms), "pending": int(o.pending)}

    def build_tie_order(self):
        self._ck(self._L.cilhip_build_tie_order(self._h))

    def load_tie_order(self, order_handle, global_index=None):
        """order_handle: from capi cilhip_tie_order_create over the WHOLE target cloud; global_index: uint32 index in that cloud of each
        of this context's target points (None: the context holds the whole cloud)"""
        gi = None if global_index is None else np.ascontiguousarray(global_index, np.uint32)
        self._ck(self._L.cilhip_load_tie_order(self._h, order_handle, None if gi is None else gi.ctypes.data))

    def get_nn(self):
        idx = np.zeros(max(self.n_source, 1), np.uint32); d2 = np.zeros(max(self.n_source, 1), np.float32)
        self._ck(self._L.cilhip_get_nn(self._h, idx.ctypes.data, d2.ctypes.data, capi.MEM_HOST))
        return idx[: self.n_source], d2[: self.n_source]

    def get_correspondences(self):
        cap = max(self.n_source + self.n_target, 1)   # search direction BOTH: up to one match per point of either cloud
        i1 = np.zeros(cap, np.uint64); i2 = np.zeros(cap, np.uint64); v = np.zeros(cap, np.float32)
        n = C.c_size_t(0)
        self._ck(self._L.cilhip_get_correspondences(self._h, i1.ctypes.data, i2.ctypes.data, v.ctypes.data, cap, C.byref(n)))
        return i1[: n.value].astype(np.int64), i2[: n.value].astype(np.int64), v[: n.value].copy()

    def estimate_point_to_point(self):
        T = np.zeros(16, np.float32); sums = np.zeros(16, np.float64); ok = C.c_int(0)
        self._ck(self._L.cilhip_estimate_point_to_point(self._h, T.ctypes.data, sums.ctypes.data, C.byref(ok)))
        return _T_from_abi(T), sums, bool(ok.value)

    def estimate_combined(self, w_p2p, w_p2pl, max_iter=1, conv_tol=1e-5):
        T = np.zeros(16, np.float32); AtA = np.zeros(36, np.float64); Atb = np.zeros(6, np.float64)
        cv = C.c_int(0)
        self._ck(self._L.cilhip_estimate_combined(self._h, w_p2p, w_p2pl, max_iter, conv_tol, T.ctypes.data,
                                                  AtA.ctypes.data, Atb.ctypes.data, C.byref(cv)))
        return _T_from_abi(T), AtA.reshape(6, 6), Atb, bool(cv.value)

    def estimate_affine(self, w_p2p, w_p2pl, centered=True):
        """Affine closed form over the last correspondences (transform_estimation.hpp:369-476; the point-to-point
        overload :50-102 is w_p2p=1, w_p2pl=0, centered=False)."""
        T = np.zeros(16, np.float32); AtA = np.zeros(144, np.float64); Atb = np.zeros(12, np.float64)
        ok = C.c_int(0); n = C.c_size_t(0)
        self._ck(self._L.cilhip_estimate_affine(self._h, w_p2p, w_p2pl, 1 if centered else 0, T.ctypes.data, AtA.ctypes.data,
                                                Atb.ctypes.data, C.byref(n), C.byref(ok)))
        return _T_from_abi(T), AtA.reshape(12, 12), Atb, bool(ok.value)

    def icp_run(self, params, T0=None):
        res = capi.IcpResult()
        t0 = _T_to_abi(T0) if T0 is not None else None
        self._ck(self._L.cilhip_icp_run(self._h, C.byref(params), t0.ctypes.data if t0 is not None else None, C.byref(res)))
        return res

    def icp_begin(self, params, T0=None, global_src_mean=None):
        t0 = _T_to_abi(T0) if T0 is not None else None
        gm = np.ascontiguousarray(global_src_mean, np.float32) if global_src_mean is not None else None
        self._ck(self._L.cilhip_icp_begin(self._h, C.byref(params), t0.ctypes.data if t0 is not None else None,
                                          gm.ctypes.data if gm is not None else None))

    def icp_partial_sums(self, sums_dev_ptr):
        self._ck(self._L.cilhip_icp_partial_sums(self._h, C.c_void_p(sums_dev_ptr)))

    def icp_apply_sums(self, sums_dev_ptr):
        self._ck(self._L.cilhip_icp_apply_sums(self._h, C.c_void_p(sums_dev_ptr)))

    @staticmethod
    def rank_comm_unique_id():
        """128 bytes that make the ranks of one RCCL communicator find each other: created on ONE rank, carried to all (cilhip_rank_comm_unique_id)"""
        buf = np.zeros(128, np.uint8)
        rc = capi.load().cilhip_rank_comm_unique_id(buf.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_rank_comm_unique_id failed (librccl not loadable?)")
        return buf

    def rank_comm_prepare(self):
        """what rank_comm_init can fail at on this rank alone (librccl, its buffer), done beforehand (cilhip_rank_comm_prepare)"""
        self._ck(self._L.cilhip_rank_comm_prepare(self._h))

    def rank_comm_init(self, unique_id, nranks, rank):
        """this context as rank `rank` of `nranks` (collective: every rank calls it with the same id)"""
        uid = np.ascontiguousarray(unique_id, np.uint8)
        assert uid.size == 128
        self._ck(self._L.cilhip_rank_comm_init(self._h, uid.ctypes.data, int(nranks), int(rank)))

    def rank_comm_destroy(self):
        self._ck(self._L.cilhip_rank_comm_destroy(self._h))

    def icp_iterate_ranked(self, iterations):
        """`iterations` x {partial sums -> ncclAllReduce of the 48 f64 on the context's stream -> epilogue}, between icp_begin and icp_state"""
        self._ck(self._L.cilhip_icp_iterate_ranked(self._h, int(iterations)))

    def last_host_enqueue_time(self):
        """us of host time per iteration the ranked loop's enqueue calls took since icp_begin (paced waits excluded)"""
        v = C.c_double(0)
        self._ck(self._L.cilhip_get_last_host_enqueue_time(self._h, C.byref(v)))
        return v.value

    def get_option(self, key):
        v = C.c_double(0)
        self._ck(self._L.cilhip_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def last_allreduce_timing(self):
        """-> (ms summed over the timed iterations, how many were timed) of the ranked loop's ncclAllReduce since icp_begin (after icp_state)"""
        a = C.c_double(0); n = C.c_int(0)
        self._ck(self._L.cilhip_get_last_allreduce_timing(self._h, C.byref(a), C.byref(n)))
        return a.value, n.value

    def set_shard_info(self, target_index_offset, dst_mean=None, src_mean=None):
        dm = np.ascontiguousarray(dst_mean, np.float32) if dst_mean is not None else None
        sm = np.ascontiguousarray(src_mean, np.float32) if src_mean is not None else None
        self._ck(self._L.cilhip_set_shard_info(self._h, int(target_index_offset), dm.ctypes.data if dm is not None else None,
                                               sm.ctypes.data if sm is not None else None))

    def icp_partial_keys(self, keys_dev_ptr):
        self._ck(self._L.cilhip_icp_partial_keys(self._h, C.c_void_p(keys_dev_ptr)))

    def icp_sums_from_keys(self, keys_dev_ptr, sums_dev_ptr):
        self._ck(self._L.cilhip_icp_sums_from_keys(self._h, C.c_void_p(keys_dev_ptr), C.c_void_p(sums_dev_ptr)))

    def icp_order_keys(self, win_keys_dev_ptr, order_keys_dev_ptr):
        """the reference's tie order across target shards (c_api.h: cilhip_icp_order_keys): the second key of the two-key protocol"""
        self._ck(self._L.cilhip_icp_order_keys(self._h, C.c_void_p(win_keys_dev_ptr), C.c_void_p(order_keys_dev_ptr)))

    def icp_sums_from_ordered_keys(self, win_keys_dev_ptr, order_keys_dev_ptr, sums_dev_ptr):
        self._ck(self._L.cilhip_icp_sums_from_ordered_keys(self._h, C.c_void_p(win_keys_dev_ptr), C.c_void_p(order_keys_dev_ptr), C.c_void_p(sums_dev_ptr)))

    def set_slab_guard(self, axis, slack=0.0, center=None, half_extent=None, T_part=None):
        if axis is None or axis < 0:
            self._ck(self._L.cilhip_set_slab_guard(self._h, -1, C.c_float(0.0), None, None, None))
            return
        c = np.ascontiguousarray(center, np.float32); h = np.ascontiguousarray(half_extent, np.float32); t = _T_to_abi(T_part)
        self._ck(self._L.cilhip_set_slab_guard(self._h, int(axis), C.c_float(slack), c.ctypes.data, h.ctypes.data, t.ctypes.data))

    def slab_violation(self):
        v = C.c_int(0)
        self._ck(self._L.cilhip_get_slab_violation(self._h, C.byref(v)))
        return bool(v.value)

    def slab_violation_state(self):
        """-> (violated, IcpResult): the loop state right after the update that raised the guard (still exact: the flag is about
        the NEXT search); the current state when the guard has not fired"""
        v = C.c_int(0); res = capi.IcpResult()
        self._ck(self._L.cilhip_get_slab_violation_state(self._h, C.byref(v), C.byref(res)))
        return bool(v.value), res

    def icp_state(self):
        res = capi.IcpResult()
        self._ck(self._L.cilhip_icp_state(self._h, C.byref(res)))
        return res

    def compute_residuals(self, metric, w_p2p, w_p2pl, T):
        out = np.zeros(max(self.n_source, 1), np.float32)
        t = _T_to_abi(T)
        self._ck(self._L.cilhip_compute_residuals(self._h, metric, w_p2p, w_p2pl, t.ctypes.data, out.ctypes.data, capi.MEM_HOST))
        return out[: self.n_source]

    def prepare_source(self, T=None, force=False):
        """Run the spatial pre-sort of the source explicitly (otherwise lazy); returns its wall time in ms."""
        ms = C.c_double(0)
        t = None if T is None else _T_to_abi(T)
        self._ck(self._L.cilhip_prepare_source(self._h, None if t is None else t.ctypes.data, 1 if force else 0, C.byref(ms)))
        return ms.value

    def grid_info(self):
        gi = capi.GridInfo()
        self._ck(self._L.cilhip_get_grid_info(self._h, C.byref(gi)))
        return gi

    def enable_kernel_timing(self, on=True):
        self._ck(self._L.cilhip_enable_kernel_timing(self._h, 1 if on else 0))

    def debug_counters(self):
        out = (C.c_uint32 * 2)()
        self._ck(self._L.cilhip_debug_counters(self._h, out))
        return int(out[0]), int(out[1])

    def set_option(self, key, value):
        self._ck(self._L.cilhip_set_option(self._h, key.encode(), float(value)))

    def set_pair_weight_callback(self, fn):
        """a caller's own correspondence weight evaluators (cilhip_set_pair_weight_callback): fn(index_in_first, index_in_second,
        value) -> (point weights, plane weights), called once per estimate with the whole stored correspondence set as numpy arrays
        (uint64, uint64, float32); None = back to the option-selected stock evaluators"""
        if fn is None:
            self._weight_cb = None
            self._ck(self._L.cilhip_set_pair_weight_callback(self._h, capi.PAIR_WEIGHT_FN(), None))
            return

        def tramp(_user, i1, i2, val, n, wq, wl):
            try:
                if getattr(self, "_weight_exc", None) is not None:
                    return      # (an earlier call raised: the weights stay 0, the exception surfaces when the library call returns)
                n = int(n)
                a1 = np.ctypeslib.as_array(i1, shape=(n,)); a2 = np.ctypeslib.as_array(i2, shape=(n,)); v = np.ctypeslib.as_array(val, shape=(n,))
                q, l = fn(a1, a2, v)
                np.ctypeslib.as_array(wq, shape=(n,))[:] = np.asarray(q, np.float32)
                np.ctypeslib.as_array(wl, shape=(n,))[:] = np.asarray(l, np.float32)
            except BaseException as e:      # noqa: BLE001  (kept for _ck)
                self._weight_exc = e

        self._weight_cb = capi.PAIR_WEIGHT_FN(tramp)      # (kept alive with the context)
        self._ck(self._L.cilhip_set_pair_weight_callback(self._h, self._weight_cb, None))

    def last_run_forms(self):
        """(iterations run as one pass, iterations run as search + streaming accumulation) of the last icp_run"""
        a = C.c_int(0); b = C.c_int(0)
        self._ck(self._L.cilhip_get_last_run_forms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_form_timing(self):
        """{form: (kernel ms, launches)} of the last timed run; forms: 0 search alone, 1 tiled one pass, 2 first warm-started
        iteration, 3 warm-started iterations, 4 per-lane fused kernel"""
        out = {}
        for f in range(5):
            ms = C.c_double(0); n = C.c_int(0)
            self._ck(self._L.cilhip_get_last_form_timing(self._h, f, C.byref(ms), C.byref(n)))
            out[f] = (ms.value, n.value)
        return out

    def last_run_trace(self, cap=256):
        """the last icp_run iteration by iteration: list of dicts {form, unproven, listed, step, delta} (cilhip_get_last_run_trace)"""
        n = C.c_int(0)
        un = (C.c_uint * cap)(); li = (C.c_uint * cap)(); st = (C.c_float * cap)(); de = (C.c_float * cap)(); fo = (C.c_int * cap)()
        self._ck(self._L.cilhip_get_last_run_trace(self._h, cap, C.byref(n), un, li, st, de, fo))
        return [{"form": fo[i], "unproven": un[i], "listed": li[i], "step": st[i], "delta": de[i]} for i in range(n.value)]

    def last_iteration_timing(self, cap=4096):
        """[(iteration, kernel ms)] of the iterations of the last icp_run that carried events (kernel timing on; option kernel_timing_stride)"""
        n = C.c_int(0)
        it = np.zeros(cap, np.uint32); ms = np.zeros(cap, np.float32)
        self._ck(self._L.cilhip_get_last_iteration_timing(self._h, cap, C.byref(n), it.ctypes.data, ms.ctypes.data))
        k = min(n.value, cap)
        return [(int(it[i]), float(ms[i])) for i in range(k)]

    def last_warm_iterations(self):
        """how many of the one-pass iterations of the last icp_run ran as the warm-started per-lane kernel"""
        a = C.c_int(0)
        self._ck(self._L.cilhip_get_last_warm_iterations(self._h, C.byref(a)))
        return a.value

    def last_matches_origin(self):
        """where get_nn / get_correspondences read from: 0 none, 3 find_correspondences, 1 kept by the last icp_run's kernels,
        2 searched again on demand under the last iteration's transform"""
        a = C.c_int(0)
        self._ck(self._L.cilhip_get_last_matches_origin(self._h, C.byref(a)))
        return a.value

    def matches_transform(self):
        """the 4x4 transform the current correspondence set was found under (after icp_run: transform_ before the last update)"""
        T = np.zeros(16, np.float32)
        self._ck(self._L.cilhip_get_matches_transform(self._h, T.ctypes.data))
        return _T_from_abi(T)

    def last_timing2(self):
        a = C.c_double(0); b = C.c_double(0)
        self._ck(self._L.cilhip_get_last_timing2(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_timing(self):
        a = C.c_double(0); b = C.c_double(0); n = C.c_int(0)
        self._ck(self._L.cilhip_get_last_timing(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value


class CorrespondenceSearchHIP:
    """Models the reference's correspondence-search engine concept
    (CorrespondenceSearchKDTree, correspondence_search/correspondence_search_kd_tree.hpp) on the GPU.
    Option combinations the GPU path does not implement raise -- they never silently differ."""

    def __init__(self, ctx):
        self._ctx = ctx
        self.search_dir_ = CorrespondenceSearchDirection.SECOND_TO_FIRST   # :47
        self.max_distance_ = np.float32(0.01 * 0.01)                        # :48 (squared)
        self.inlier_fraction_ = 1.0
        self.require_reciprocality_ = False
        self.one_to_one_ = False
        self._corr = None

    def findCorrespondences(self, tform=None):
        T = np.eye(4, dtype=np.float32) if tform is None else tform
        self._ctx.find_correspondences(T, float(self.max_distance_), count=False)
        self._corr = None
        return self

    def setPointNormalFeatureAdaptors(self, src_normals, normal_weight, keep_metric=True):
        """Feature adaptors of the engine (common_transformable_feature_adaptors.hpp): the default is PointFeaturesAdaptor3f
        on both clouds; this switches both to PointNormalFeaturesAdaptor3f(points, normals, normal_weight) (:60-161): the
        search runs on the 6-D features (p, w n).  The target's normals are the ones the ICP object was built with;
        src_normals: the source's (None: reuse the normals given to a four-cloud ICP constructor).  keep_metric: the
        combined metric stays the three-cloud one.  normal_weight = 0 switches back to point features."""
        if src_normals is not None:
            q, nn, mem, _ = _as_cloud(src_normals)
            if nn != self._ctx.n_source:
                raise ValueError("source normals must match source points")
            self._ctx._ck(self._ctx._L.cilhip_set_source_normals(self._ctx._h, q, mem))
            self._ctx.set_option("symmetric_metric", 0 if keep_metric else 1)
        self._ctx.set_option("feature_kind", 0)
        self._ctx.set_option("feature_normal_weight", float(normal_weight))
        self._corr = None
        return self

    def setPointColorFeatureAdaptors(self, dst_colors, src_colors, color_weight):
        """both clouds' adaptors become PointColorFeaturesAdaptor3f(points, colors, color_weight)
        (common_transformable_feature_adaptors.hpp:164-252): the search runs on the 6-D features (p, w c); the colour part does
        not move with the transform.  color_weight = 0 switches back to point features."""
        d, nd, mem, _k1 = _as_cloud(dst_colors)
        s, ns, mem2, _k2 = _as_cloud(src_colors)
        if nd != self._ctx.n_target or ns != self._ctx.n_source or mem != mem2:
            raise ValueError("colours must match the clouds (and live in the same memory space)")
        self._ctx._ck(self._ctx._L.cilhip_set_color_features(self._ctx._h, d, s, mem))
        self._ctx.set_option("feature_kind", 1)
        self._ctx.set_option("feature_normal_weight", float(color_weight))
        self._corr = None
        return self

    def setPointNormalColorFeatureAdaptors(self, src_normals, dst_colors, src_colors, normal_weight, color_weight, keep_metric=True):
        """both clouds' adaptors become PointNormalColorFeaturesAdaptor3f(points, normals, colors, normal_weight, color_weight)
        (common_transformable_feature_adaptors.hpp:255-343): the search runs on the 9-D features (p, wn n, wc c); the normal part
        follows the transform as the point+normal adaptor's does, the colour part does not move.  Target normals: the ones the
        ICP object was built with; src_normals None: reuse the normals given to a four-cloud constructor."""
        if src_normals is not None:
            q, nn, mem, _ = _as_cloud(src_normals)
            if nn != self._ctx.n_source:
                raise ValueError("source normals must match source points")
            self._ctx._ck(self._ctx._L.cilhip_set_source_normals(self._ctx._h, q, mem))
            self._ctx.set_option("symmetric_metric", 0 if keep_metric else 1)
        d, nd, mem, _k1 = _as_cloud(dst_colors)
        s, ns, mem2, _k2 = _as_cloud(src_colors)
        if nd != self._ctx.n_target or ns != self._ctx.n_source or mem != mem2:
            raise ValueError("colours must match the clouds (and live in the same memory space)")
        self._ctx._ck(self._ctx._L.cilhip_set_color_features(self._ctx._h, d, s, mem))
        self._ctx.set_option("feature_kind", 2)
        self._ctx.set_option("feature_normal_weight", float(normal_weight))
        self._ctx.set_option("feature_color_weight", float(color_weight))
        self._corr = None
        return self

    def getCorrespondences(self):
        """-> structured view of the reference's CorrespondenceSet: (indexInFirst, indexInSecond, value)."""
        if self._corr is None:
            self._corr = self._ctx.get_correspondences()
        return self._corr

    def getSearchDirection(self):
        return self.search_dir_

    def setSearchDirection(self, d):
        """correspondence_search_kd_tree.hpp:239-247.  FIRST_TO_SECOND / BOTH rebuild an index over the transformed
        source every search (as the reference rebuilds its kd-tree) and hand the estimators a pair list."""
        d = CorrespondenceSearchDirection(d)
        self.search_dir_ = d
        if self._ctx is not None:
            self._ctx.set_option("search_direction", {CorrespondenceSearchDirection.SECOND_TO_FIRST: 0,
                                                      CorrespondenceSearchDirection.FIRST_TO_SECOND: 1,
                                                      CorrespondenceSearchDirection.BOTH: 2}[d])
        self._corr = None
        return self

    def getMaxDistance(self):
        return self.max_distance_

    def setMaxDistance(self, dist_thresh):
        self.max_distance_ = np.float32(dist_thresh)
        return self

    def getInlierFraction(self):
        return self.inlier_fraction_

    def setInlierFraction(self, f):
        self.inlier_fraction_ = float(f)
        if self._ctx is not None:
            self._ctx.set_option("inlier_fraction", self.inlier_fraction_)
        return self

    def getRequireReciprocality(self):
        return self.require_reciprocality_

    def setRequireReciprocality(self, b):
        """:257-263: only read when the search direction is BOTH (set_intersection instead of set_union)"""
        self.require_reciprocality_ = bool(b)
        if self._ctx is not None:
            self._ctx.set_option("require_reciprocality", 1.0 if b else 0.0)
        self._corr = None
        return self

    def getOneToOne(self):
        return self.one_to_one_

    def setOneToOne(self, b):
        self.one_to_one_ = bool(b)
        if self._ctx is not None:
            self._ctx.set_option("one_to_one", 1.0 if b else 0.0)
        return self


class _IterativeClosestPointBase:
    """registration/icp_base.hpp"""

    def __init__(self, device=0, stream=None):
        self._ctx = Context(device, stream)
        self._engine = CorrespondenceSearchHIP(self._ctx)
        self.max_iterations_ = 15            # :24
        self.convergence_tol_ = np.float32(1e-5)  # :25
        self.iterations_ = 0
        self.last_delta_norm_ = np.float32(np.inf)
        self.transform_init_ = np.eye(4, dtype=np.float32)
        self.transform_ = np.eye(4, dtype=np.float32)
        self.last_ncorr_ = 0

    def correspondenceSearchEngine(self):
        return self._engine

    def getMaxNumberOfIterations(self):
        return self.max_iterations_

    def setMaxNumberOfIterations(self, n):
        self.max_iterations_ = int(n)
        return self

    def getNumberOfPerformedIterations(self):
        return self.iterations_

    def getConvergenceTolerance(self):
        return self.convergence_tol_

    def setConvergenceTolerance(self, tol):
        self.convergence_tol_ = np.float32(tol)
        return self

    def getInitialTransform(self):
        return self.transform_init_

    def setInitialTransform(self, T):
        self.transform_init_ = np.asarray(T, np.float32).reshape(4, 4).copy()
        return self

    def getLastUpdateNorm(self):
        return self.last_delta_norm_

    def hasConverged(self):
        return bool(self.last_delta_norm_ < self.convergence_tol_)

    def getTransform(self):
        return self.transform_

    def _params(self):
        raise NotImplementedError

    def estimate(self, max_iter=None, conv_tol=None):
        if max_iter is not None:
            self.max_iterations_ = int(max_iter)
        if conv_tol is not None:
            self.convergence_tol_ = np.float32(conv_tol)
        res = self._ctx.icp_run(self._params(), self.transform_init_)
        self._engine._corr = None      # the engine now holds the last iteration's set (correspondence_search_kd_tree.hpp:231)
        self.transform_ = _T_from_abi(res.T[:])
        self.iterations_ = int(res.iterations)
        self.last_delta_norm_ = np.float32(res.last_delta_norm)
        self.last_ncorr_ = int(res.last_ncorr)
        return self


class SimplePointToPointMetricRigidICP3f(_IterativeClosestPointBase):
    """registration/icp_common_instances.hpp:250 (wrapper :34-45) over
    PointToPointMetricSingleTransformICP (icp_single_transform_point_to_point_metric.hpp)."""

    def __init__(self, dst_points, src_points, device=0, stream=None):
        super().__init__(device, stream)
        self._ctx.set_target(dst_points, None)
        self._ctx.set_source(src_points)

    def _params(self):
        p = capi.IcpParams()
        self._ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric = capi.METRIC_POINT_TO_POINT
        p.max_iter = self.max_iterations_
        p.conv_tol = float(self.convergence_tol_)
        p.max_sq_dist = float(self._engine.max_distance_)
        return p

    def getResiduals(self):
        return self._ctx.compute_residuals(0, 0.0, 0.0, self.transform_)


class UnityWeightEvaluator:
    """core/common_pair_evaluators.hpp:30-43: every correspondence weighs 1 (the instances' default)."""
    kind = 0
    sigma = 1.0


class IdentityWeightEvaluator:
    """core/common_pair_evaluators.hpp:14-27: the weight is the correspondence's value (its squared search distance)."""
    kind = 1
    sigma = 1.0


class RBFKernelWeightEvaluator:
    """core/common_pair_evaluators.hpp:46-80 over squared distances: exp(-0.5 / sigma^2 * value)."""
    kind = 2

    def __init__(self, sigma=1.0):
        self.sigma = float(sigma)

    def setSigma(self, sigma):
        self.sigma = float(sigma)
        return self


def _as_host_evaluator(ev):
    """a stock evaluator object as the host function the reference's class computes (core/common_pair_evaluators.hpp), any other
    callable as it is"""
    kind = getattr(ev, "kind", None)
    if kind is None:
        return ev
    if kind == 0:
        return lambda i1, i2, v: np.ones(len(v), np.float32)
    if kind == 1:
        return lambda i1, i2, v: np.asarray(v, np.float32)
    coeff = np.float32(-0.5) / (np.float32(ev.sigma) * np.float32(ev.sigma))
    return lambda i1, i2, v: np.exp(coeff * np.asarray(v, np.float32)).astype(np.float32)


class SimpleCombinedMetricRigidICP3f(_IterativeClosestPointBase):
    """registration/icp_common_instances.hpp:261 (wrapper :74-97) over
    CombinedMetricSingleTransformICP (icp_single_transform_combined_metric.hpp); defaults :44-47."""

    def __init__(self, dst_points, dst_normals, src_points, src_normals=None, device=0, stream=None):
        """Three clouds: combined (point-to-point + point-to-plane) metric.  Four clouds (src_normals):
        the symmetric objective (icp_common_instances.hpp:88-97 -> transform_estimation.hpp:604-739)."""
        super().__init__(device, stream)
        self._ctx.set_target(dst_points, dst_normals)
        self._ctx.set_source(src_points, src_normals)
        self.max_optimization_iterations_ = 1
        self.optimization_convergence_tol_ = np.float32(1e-5)
        self.point_to_point_weight_ = np.float32(0.0)
        self.point_to_plane_weight_ = np.float32(1.0)
        # the class template's PointToPoint/PointToPlaneCorrWeightEvaluatorT (icp_single_transform_combined_metric.hpp:11-14)
        self.point_corr_eval_ = UnityWeightEvaluator()
        self.plane_corr_eval_ = UnityWeightEvaluator()

    def pointToPointCorrespondenceWeightEvaluator(self):
        """:95-97"""
        return self.point_corr_eval_

    def pointToPlaneCorrespondenceWeightEvaluator(self):
        """:99-101"""
        return self.plane_corr_eval_

    def setCorrespondenceWeightEvaluators(self, point_eval=None, plane_eval=None):
        """The reference fixes the evaluator TYPES at compile time (template arguments); here they are objects."""
        if point_eval is not None:
            self.point_corr_eval_ = point_eval
        if plane_eval is not None:
            self.plane_corr_eval_ = plane_eval
        return self

    def _push_weight_evaluators(self):
        evs = (("point", self.point_corr_eval_), ("plane", self.plane_corr_eval_))
        if all(hasattr(ev, "kind") for _, ev in evs):      # the three stock classes: evaluated on the device
            self._ctx.set_pair_weight_callback(None)
            for name, ev in evs:
                self._ctx.set_option(name + "_weight_evaluator", ev.kind)
                self._ctx.set_option(name + "_weight_sigma", ev.sigma)
            return
        # any other callable evaluator(index_in_first, index_in_second, value) -> weights (arrays in, array out): the reference's
        # template argument (icp_single_transform_combined_metric.hpp:10-14); evaluated on the host, per estimate
        fns = [_as_host_evaluator(ev) for _, ev in evs]
        self._ctx.set_pair_weight_callback(lambda i1, i2, v: (fns[0](i1, i2, v), fns[1](i1, i2, v)))

    def getPointToPointMetricWeight(self):
        return self.point_to_point_weight_

    def setPointToPointMetricWeight(self, w):
        self.point_to_point_weight_ = np.float32(w)
        return self

    def getPointToPlaneMetricWeight(self):
        return self.point_to_plane_weight_

    def setPointToPlaneMetricWeight(self, w):
        self.point_to_plane_weight_ = np.float32(w)
        return self

    def getMaxNumberOfOptimizationStepIterations(self):
        return self.max_optimization_iterations_

    def setMaxNumberOfOptimizationStepIterations(self, n):
        self.max_optimization_iterations_ = int(n)
        return self

    def getOptimizationStepConvergenceTolerance(self):
        return self.optimization_convergence_tol_

    def setOptimizationStepConvergenceTolerance(self, tol):
        self.optimization_convergence_tol_ = np.float32(tol)
        return self

    def _params(self):
        p = capi.IcpParams()
        self._ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric = capi.METRIC_COMBINED
        p.w_p2p = float(self.point_to_point_weight_)
        p.w_p2pl = float(self.point_to_plane_weight_)
        p.max_iter = self.max_iterations_
        p.conv_tol = float(self.convergence_tol_)
        p.max_opt_iter = self.max_optimization_iterations_
        p.opt_conv_tol = float(self.optimization_convergence_tol_)
        p.max_sq_dist = float(self._engine.max_distance_)
        self._push_weight_evaluators()
        return p

    def getResiduals(self):
        return self._ctx.compute_residuals(1, float(self.point_to_point_weight_), float(self.point_to_plane_weight_),
                                           self.transform_)


class CorrespondenceSearchCombinedMetricCombiner:
    """registration/correspondence_search_combined_metric_combiner.hpp:8-81: one engine finds the correspondences of the
    combined metric's point-to-point terms, another those of its point-to-plane terms (own radius, post-filters, feature
    adaptors).  Both engines hold the same two clouds (make_engine builds one more over them); the same object twice is the
    single-engine case (:33-43)."""

    def __init__(self, point_to_point_corr_search, point_to_plane_corr_search):
        self.point_to_point_corr_search_ = point_to_point_corr_search
        self.point_to_plane_corr_search_ = point_to_plane_corr_search

    @staticmethod
    def make_engine(dst_points, dst_normals, src_points, device=0, stream=None):
        """a CorrespondenceSearchHIP over its own context holding the two clouds"""
        ctx = Context(device, stream)
        ctx.set_target(dst_points, dst_normals)
        ctx.set_source(src_points)
        return CorrespondenceSearchHIP(ctx)

    def findCorrespondences(self, tform=None):
        self.point_to_point_corr_search_.findCorrespondences(tform)
        if self.point_to_plane_corr_search_ is not self.point_to_point_corr_search_:
            self.point_to_plane_corr_search_.findCorrespondences(tform)
        return self

    def getPointToPointCorrespondences(self):
        return self.point_to_point_corr_search_.getCorrespondences()

    def getPointToPlaneCorrespondences(self):
        return self.point_to_plane_corr_search_.getCorrespondences()

    def pointToPointCorrespondenceSearchEngine(self):
        return self.point_to_point_corr_search_

    def pointToPlaneCorrespondenceSearchEngine(self):
        return self.point_to_plane_corr_search_


class CombinedMetricRigidICP3f(_IterativeClosestPointBase):
    """CombinedMetricSingleTransformICP (registration/icp_single_transform_combined_metric.hpp:8-243) handed its correspondence
    search engine instead of owning one -- here a CorrespondenceSearchCombinedMetricCombiner over two engines.  Defaults :44-47."""

    def __init__(self, combiner):
        # (no context of its own: the two engines' contexts hold the clouds)
        self._combiner = combiner
        self._engine = combiner
        self._ctx = combiner.point_to_point_corr_search_._ctx
        self.max_iterations_ = 15
        self.convergence_tol_ = np.float32(1e-5)
        self.iterations_ = 0
        self.last_delta_norm_ = np.float32(np.inf)
        self.transform_init_ = np.eye(4, dtype=np.float32)
        self.transform_ = np.eye(4, dtype=np.float32)
        self.last_ncorr_ = 0
        self.max_optimization_iterations_ = 1
        self.optimization_convergence_tol_ = np.float32(1e-5)
        self.point_to_point_weight_ = np.float32(0.0)
        self.point_to_plane_weight_ = np.float32(1.0)

    def setPointToPointMetricWeight(self, w):
        self.point_to_point_weight_ = np.float32(w)
        return self

    def setPointToPlaneMetricWeight(self, w):
        self.point_to_plane_weight_ = np.float32(w)
        return self

    def setMaxNumberOfOptimizationStepIterations(self, n):
        self.max_optimization_iterations_ = int(n)
        return self

    def setOptimizationStepConvergenceTolerance(self, tol):
        self.optimization_convergence_tol_ = np.float32(tol)
        return self

    def estimate(self, max_iter=None, conv_tol=None):
        if max_iter is not None:
            self.max_iterations_ = int(max_iter)
        if conv_tol is not None:
            self.convergence_tol_ = np.float32(conv_tol)
        e1, e2 = self._combiner.point_to_point_corr_search_, self._combiner.point_to_plane_corr_search_
        p = capi.IcpParams()
        self._ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric = capi.METRIC_COMBINED
        p.w_p2p, p.w_p2pl = float(self.point_to_point_weight_), float(self.point_to_plane_weight_)
        p.max_iter, p.conv_tol = self.max_iterations_, float(self.convergence_tol_)
        p.max_opt_iter, p.opt_conv_tol = self.max_optimization_iterations_, float(self.optimization_convergence_tol_)
        res = capi.IcpResult()
        T0 = _T_to_abi(self.transform_init_)
        self._ctx._ck(self._ctx._L.cilhip_icp_run_two_sets(e1._ctx._h, float(e1.max_distance_), e2._ctx._h, float(e2.max_distance_), C.byref(p),
                                                          T0.ctypes.data_as(C.c_void_p), C.byref(res)))
        e1._corr = None
        e2._corr = None
        self.transform_ = _T_from_abi(res.T[:])
        self.iterations_ = int(res.iterations)
        self.last_delta_norm_ = np.float32(res.last_delta_norm)
        self.last_ncorr_ = int(res.last_ncorr)
        return self


class SimplePointToPointMetricAffineICP3f(SimplePointToPointMetricRigidICP3f):
    """registration/icp_common_instances.hpp:255: the same loop and correspondence engine with an AffineTransform --
    the step is the affine closed form on the raw coordinates (transform_estimation.hpp:50-102), no rotation() polish."""

    def __init__(self, dst_points, src_points, device=0, stream=None):
        super().__init__(dst_points, src_points, device, stream)
        self._ctx.set_option("transform_mode", 1)


class SimpleCombinedMetricAffineICP3f(SimpleCombinedMetricRigidICP3f):
    """registration/icp_common_instances.hpp:266: combined metric with an AffineTransform -- 12-unknown closed form
    (transform_estimation.hpp:369-476).  Source normals, if given, are not used: the symmetric objective exists for the
    rigid instances only (icp_single_transform_combined_metric.hpp:180-204)."""

    def __init__(self, dst_points, dst_normals, src_points, src_normals=None, device=0, stream=None):
        super().__init__(dst_points, dst_normals, src_points, None, device, stream)
        self._ctx.set_option("transform_mode", 1)
