"""Python mirrors of the k-NN side of cilantro's KDTree3f and of NormalEstimation3f on top of the C ABI
(cilhip_knn3f, cilhip_normals_knn3f).

    tree = KDTree3f(points)                                   # core/kd_tree.hpp:144-388
    idx, d2, cnt = tree.kNNSearch(queries, k)                  # :233-240   (rows padded with -1 / +inf)
    idx, d2, cnt = tree.kNNInRadiusSearch(queries, k, radius)  # :303-318   (radius is a SQUARED distance, as everywhere)

    ne = NormalEstimation3f(points).setViewPoint([0, 0, 0])   # core/normal_estimation.hpp
    normals, curvature = ne.getNormalsAndCurvatureKNN(k)       # :72-80
    normals = ne.getNormalsKNNInRadius(k, radius)              # :189-196   (NormalEstimation squares its radius itself, :174)

Radius-only neighbourhoods are supported for the normals (moments accumulated without listing the neighbours).
"""
import ctypes as C

import numpy as np

from . import capi
from .icp import _as_cloud


def set_knn_tie_rule(rule):
    """which of several EXACTLY equidistant points a k-NN list names, and in which order (cilhip_knn_set_tie_rule; process-wide):
    2 (default) = the reference's -- the ones its kd-tree traversal meets first, in that order (core/kd_tree.hpp:80-99) --, its order
    tables built the first time a call meets equal distances; 1 = the same, tables built up front; 0 = lowest index"""
    rc = capi.load().cilhip_knn_set_tie_rule(int(rule))
    if rc != capi.OK:
        raise capi.CilhipError(rc, "cilhip_knn_set_tie_rule: 0, 1 or 2")


class KDTree3f:
    def __init__(self, points, device=0):
        self._L = capi.load()
        self._points = points
        self._device = device

    def _search(self, queries, k, radius_sq):
        p, n, mem, keep = _as_cloud(self._points)
        if queries is None:
            qp, nq = None, n
        else:
            qp, nq, qmem, qkeep = _as_cloud(queries)
            if qmem != mem:
                raise ValueError("reference points and queries must live in the same memory space")
        k = int(k)
        idx = np.zeros((nq, k), np.uint32); d2 = np.zeros((nq, k), np.float32); cnt = np.zeros(nq, np.uint32)
        rc = self._L.cilhip_knn3f(self._device, p, n, qp, nq, mem, k, C.c_float(radius_sq), idx.ctypes.data, d2.ctypes.data, cnt.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_knn3f failed (no HIP device, k > 32, or bad arguments)")
        out = idx.astype(np.int64)
        out[idx == capi.NONE_IDX] = -1
        return out, d2, cnt.astype(np.int64)

    def kNNSearch(self, queries, k):
        """queries None: the tree's own points"""
        return self._search(queries, k, np.inf)

    def kNNInRadiusSearch(self, queries, k, radius):
        return self._search(queries, k, float(radius))

    def radiusSearch(self, queries, radius):
        """core/kd_tree.hpp:251-282: every neighbour with squared distance < radius, ascending by distance (ties by index).
        -> (offsets int64 [nq+1], indices int64 [total], squared distances f32 [total]); query i owns
        indices[offsets[i]:offsets[i+1]].  queries None: the tree's own points."""
        p, n, mem, keep = _as_cloud(self._points)
        if queries is None:
            qp, nq = None, n
        else:
            qp, nq, qmem, qkeep = _as_cloud(queries)
            if qmem != mem:
                raise ValueError("reference points and queries must live in the same memory space")
        off = np.zeros(nq + 1, np.uint64)
        total = C.c_size_t(0)
        rc = self._L.cilhip_radius_search3f(self._device, p, n, qp, nq, mem, C.c_float(radius), off.ctypes.data, None, None, 0, C.byref(total))
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_radius_search3f failed (no HIP device or bad arguments)")
        idx = np.zeros(max(total.value, 1), np.uint32); d2 = np.zeros(max(total.value, 1), np.float32)
        if total.value:
            rc = self._L.cilhip_radius_search3f(self._device, p, n, qp, nq, mem, C.c_float(radius), off.ctypes.data, idx.ctypes.data,
                                                d2.ctypes.data, total.value, C.byref(total))
            if rc != capi.OK:
                raise capi.CilhipError(rc, "cilhip_radius_search3f failed")
        return off.astype(np.int64), idx[: total.value].astype(np.int64), d2[: total.value]

    def nearestNeighborSearch(self, queries):
        idx, d2, _ = self._search(queries, 1, np.inf)
        return idx[:, 0], d2[:, 0]


class NormalEstimation3f:
    def __init__(self, points, device=0):
        self._L = capi.load()
        self._points = points
        self._device = device
        self._vp = None          # normal_estimation.hpp:24: NaN view point = no orientation step

    def setViewPoint(self, vp):
        self._vp = None if vp is None else np.ascontiguousarray(vp, np.float32).reshape(3)
        return self

    def getViewPoint(self):
        return np.full(3, np.nan, np.float32) if self._vp is None else self._vp

    def _run(self, k, radius_sq, want_curvature):
        p, n, mem, keep = _as_cloud(self._points)
        nrm = np.zeros((n, 3), np.float32)
        cur = np.zeros(n, np.float32) if want_curvature else None
        rc = self._L.cilhip_normals_knn3f(self._device, p, n, mem, int(k), C.c_float(radius_sq),
                                          None if self._vp is None else self._vp.ctypes.data, nrm.ctypes.data,
                                          None if cur is None else cur.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_normals_knn3f failed (no HIP device, k > 32, or bad arguments)")
        return nrm, cur

    def getNormalsAndCurvatureKNN(self, k):
        return self._run(k, np.inf, True)

    def getNormalsKNN(self, k):
        return self._run(k, np.inf, False)[0]

    def getCurvatureKNN(self, k):
        return self._run(k, np.inf, True)[1]

    @staticmethod
    def _sq(radius):
        """NormalEstimation takes a plain radius and squares it in f32 (normal_estimation.hpp:126, :174)"""
        r = np.float32(radius)
        return float(r * r)

    def getNormalsAndCurvatureKNNInRadius(self, k, radius):
        return self._run(k, self._sq(radius), True)

    def getNormalsKNNInRadius(self, k, radius):
        return self._run(k, self._sq(radius), False)[0]

    def getCurvatureKNNInRadius(self, k, radius):
        return self._run(k, self._sq(radius), True)[1]

    def _run_radius(self, radius_sq, want_curvature):
        p, n, mem, keep = _as_cloud(self._points)
        nrm = np.zeros((n, 3), np.float32)
        cur = np.zeros(n, np.float32) if want_curvature else None
        rc = self._L.cilhip_normals_radius3f(self._device, p, n, mem, C.c_float(radius_sq), None if self._vp is None else self._vp.ctypes.data,
                                             nrm.ctypes.data, None if cur is None else cur.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_normals_radius3f failed (no HIP device or bad arguments)")
        return nrm, cur

    def getNormalsAndCurvatureRadius(self, radius):
        return self._run_radius(self._sq(radius), True)

    def getNormalsRadius(self, radius):
        return self._run_radius(self._sq(radius), False)[0]

    def getCurvatureRadius(self, radius):
        return self._run_radius(self._sq(radius), True)[1]
