"""Python mirror of cilantro's PlaneRANSACEstimator3f (model_estimation/ransac_hyperplane_estimator.hpp,
model_estimation/ransac_base.hpp) on top of the C ABI (cilhip_plane_ransac3f).

    pe = PlaneRANSACEstimator3f(points)
    pe.setMaxInlierResidual(0.01).setTargetInlierCount(n // 2).setMaxNumberOfIterations(250).setReEstimationStep(True)
    plane = pe.estimate().getModel()          # (nx, ny, nz, offset): n . x + offset = 0
    pe.getModelInliers(); pe.getModelResiduals(); pe.getNumberOfPerformedIterations(); pe.targetInlierCountAchieved()

Defaults are the reference's (ransac_hyperplane_estimator.hpp:17-19): sample size 3, target inlier count
ceil(n / 2), 100 iterations, max residual 0.1, re-estimation on.
"""
import ctypes as C

import numpy as np

from . import capi
from .icp import _as_cloud


class PlaneRANSACEstimator3f:
    def __init__(self, points, device=0):
        self._L = capi.load()
        self._points = points
        self._device = device
        _, n, _, _ = _as_cloud(points)
        self._n = n
        self.inlier_count_thresh_ = n // 2 + n % 2
        self.max_iter_ = 100
        self.inlier_dist_thresh_ = 0.1
        self.re_estimate_ = True
        self._samples = None
        self._seed = 0
        self._model = None
        self._residuals = None
        self._inliers = None
        self._raw = None

    # ---- ransac_base.hpp:29-61 --------------------------------------------------------------------
    def getTargetInlierCount(self):
        return self.inlier_count_thresh_

    def setTargetInlierCount(self, v):
        self.inlier_count_thresh_ = int(v)
        return self

    def getMaxNumberOfIterations(self):
        return self.max_iter_

    def setMaxNumberOfIterations(self, v):
        self.max_iter_ = int(v)
        return self

    def getMaxInlierResidual(self):
        return self.inlier_dist_thresh_

    def setMaxInlierResidual(self, v):
        self.inlier_dist_thresh_ = float(v)
        return self

    def getReEstimationStep(self):
        return self.re_estimate_

    def setReEstimationStep(self, v):
        self.re_estimate_ = bool(v)
        return self

    # ---- not in the reference: reproducible sampling ------------------------------------------------
    def setSamples(self, samples):
        """explicit random samples, (max_iter, 3) point indices (ransac_base.hpp:83-91 draws them from
        std::random_device); None -> drawn by the library from `setSeed`"""
        self._samples = None if samples is None else np.ascontiguousarray(samples, np.uint32).reshape(-1, 3)
        return self

    def setSeed(self, seed):
        self._seed = int(seed)
        return self

    # ---- ransac_base.hpp:64-131 ---------------------------------------------------------------------
    def estimate(self, max_residual=None, target_inlier_count=None, max_iter=None):
        if max_residual is not None:   # estimate(max_residual, target_inlier_count, max_iter): :133-139
            self.inlier_dist_thresh_, self.inlier_count_thresh_, self.max_iter_ = float(max_residual), int(target_inlier_count), int(max_iter)
        p, n, mem, keep = _as_cloud(self._points)
        max_iter = self.max_iter_
        sp = None
        if self._samples is not None:
            if len(self._samples) < max_iter:
                raise ValueError("setSamples: need one sample triple per iteration")
            sp = self._samples.ctypes.data
        out = capi.PlaneModel()
        res = np.zeros(n, np.float32)
        inl = np.zeros(max(n, 1), np.uint32)
        rc = self._L.cilhip_plane_ransac3f(self._device, p, n, mem, sp, self._seed, C.c_float(self.inlier_dist_thresh_),
                                           self.inlier_count_thresh_, max_iter, int(self.re_estimate_), C.byref(out),
                                           res.ctypes.data, inl.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_plane_ransac3f failed (no HIP device or bad arguments)")
        self._raw = out
        self._model = np.array([out.normal[0], out.normal[1], out.normal[2], out.offset], np.float32)
        self._residuals = res
        self._inliers = inl[: out.n_inliers].astype(np.int64)
        return self

    def _ensure(self):
        if self._raw is None:
            self.estimate()

    def getModel(self):
        self._ensure()
        return self._model

    def getModelResiduals(self):
        self._ensure()
        return self._residuals

    def getModelInliers(self):
        self._ensure()
        return self._inliers

    def getNumberOfPerformedIterations(self):
        return 0 if self._raw is None else int(self._raw.iterations)

    def getNumberOfInliers(self):
        return 0 if self._raw is None else int(self._raw.n_inliers)

    def targetInlierCountAchieved(self):
        return self._raw is not None and bool(self._raw.target_reached)

    def getDeviceMilliseconds(self):
        return 0.0 if self._raw is None else float(self._raw.device_ms)

    # ---- ransac_hyperplane_estimator.hpp:22-62 ------------------------------------------------------
    def estimateModel(self):
        """PCA plane through ALL points"""
        p, n, mem, keep = _as_cloud(self._points)
        pl = np.zeros(4, np.float32)
        rc = self._L.cilhip_plane_fit3f(self._device, p, n, mem, pl.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_plane_fit3f failed")
        return pl

    def countInliers(self, planes, max_residual=None):
        """inlier counts of given planes (m, 4) -- the scoring half of the loop, in one pass"""
        p, n, mem, keep = _as_cloud(self._points)
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
        cnt = np.zeros(len(planes), np.uint32)
        thr = self.inlier_dist_thresh_ if max_residual is None else max_residual
        rc = self._L.cilhip_plane_score3f(self._device, p, n, mem, planes.ctypes.data, len(planes), C.c_float(thr), cnt.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_plane_score3f failed")
        return cnt.astype(np.int64)


class RigidTransformRANSACEstimator3f:
    """Python mirror of cilantro's RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp,
    loop model_estimation/ransac_base.hpp:64-131) on top of cilhip_transform_ransac3f.

        te = RigidTransformRANSACEstimator3f(dst_points, src_points, correspondences=(index_in_first, index_in_second))
        te.setMaxInlierResidual(0.01).setTargetInlierCount(n // 2).setMaxNumberOfIterations(100).setReEstimationStep(True)
        T = te.estimate().getModel()              # 4x4, maps src onto dst
        te.getModelInliers(); te.getModelResiduals(); te.getNumberOfPerformedIterations(); te.targetInlierCountAchieved()

    Constructors of the reference (:25-59): paired clouds, clouds + a correspondence set, clouds + two index lists -- the last two
    gather the pairs (here: `correspondences=(dst_indices, src_indices)`).  Defaults (:27-30): sample size 3, target inlier
    count ceil(n / 2), 100 iterations, max residual 0.01, re-estimation on."""

    def __init__(self, dst_points, src_points, correspondences=None, device=0):
        self._L = capi.load()
        self._device = device
        d = np.ascontiguousarray(np.asarray(dst_points.cpu() if hasattr(dst_points, "cpu") else dst_points, np.float32)).reshape(-1, 3)
        s = np.ascontiguousarray(np.asarray(src_points.cpu() if hasattr(src_points, "cpu") else src_points, np.float32)).reshape(-1, 3)
        if correspondences is not None:      # :34-59: dst_points_tmp_.col(i) = dst_points.col(corr[i].indexInFirst) ...
            i1 = np.asarray(correspondences[0], np.int64); i2 = np.asarray(correspondences[1], np.int64)
            d, s = np.ascontiguousarray(d[i1]), np.ascontiguousarray(s[i2])
        if len(d) != len(s):
            raise ValueError("dst / src pairs must have the same length")
        self._dst, self._src = d, s
        n = len(d)
        self._n = n
        self.inlier_count_thresh_ = n // 2 + n % 2
        self.max_iter_ = 100
        self.inlier_dist_thresh_ = 0.01
        self.re_estimate_ = True
        self._samples = None
        self._seed = 0
        self._raw = None
        self._model = np.eye(4, dtype=np.float32)
        self._residuals = np.zeros(0, np.float32)
        self._inliers = np.zeros(0, np.int64)

    def getTargetInlierCount(self):
        return self.inlier_count_thresh_

    def setTargetInlierCount(self, v):
        self.inlier_count_thresh_ = int(v)
        return self

    def getMaxNumberOfIterations(self):
        return self.max_iter_

    def setMaxNumberOfIterations(self, v):
        self.max_iter_ = int(v)
        return self

    def getMaxInlierResidual(self):
        return self.inlier_dist_thresh_

    def setMaxInlierResidual(self, v):
        self.inlier_dist_thresh_ = float(v)
        return self

    def getReEstimationStep(self):
        return self.re_estimate_

    def setReEstimationStep(self, v):
        self.re_estimate_ = bool(v)
        return self

    def getDataPointsCount(self):
        return self._n

    # ---- not in the reference: reproducible sampling (ransac_base.hpp:73 seeds from std::random_device) ----
    def setSamples(self, samples):
        self._samples = None if samples is None else np.ascontiguousarray(samples, np.uint32).reshape(-1, 3)
        return self

    def setSeed(self, seed):
        self._seed = int(seed)
        return self

    def estimate(self, max_residual=None, target_inlier_count=None, max_iter=None):
        if max_residual is not None:   # :133-139
            self.inlier_dist_thresh_, self.inlier_count_thresh_, self.max_iter_ = float(max_residual), int(target_inlier_count), int(max_iter)
        n, max_iter = self._n, self.max_iter_
        sp = None
        if self._samples is not None:
            if len(self._samples) < max_iter:
                raise ValueError("setSamples: need one sample triple per iteration")
            sp = self._samples.ctypes.data
        out = capi.TransformModel()
        res = np.zeros(max(n, 1), np.float32)
        inl = np.zeros(max(n, 1), np.uint32)
        rc = self._L.cilhip_transform_ransac3f(self._device, self._dst.ctypes.data, self._src.ctypes.data, n, capi.MEM_HOST, sp, self._seed,
                                               C.c_float(self.inlier_dist_thresh_), self.inlier_count_thresh_, max_iter, int(self.re_estimate_),
                                               C.byref(out), res.ctypes.data, inl.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_transform_ransac3f failed (no HIP device or bad arguments)")
        self._raw = out
        self._model = np.array(out.T[:], np.float32).reshape(4, 4).T.copy()
        # no accepted hypothesis and no re-estimation: the reference's residuals / inliers stay empty
        self._residuals = res[:n] if out.have_model else np.zeros(0, np.float32)
        self._inliers = inl[: out.n_inliers].astype(np.int64)
        return self

    def _ensure(self):
        if self._raw is None:
            self.estimate()

    def getModel(self):
        self._ensure()
        return self._model

    def getModelResiduals(self):
        self._ensure()
        return self._residuals

    def getModelInliers(self):
        self._ensure()
        return self._inliers

    def getNumberOfPerformedIterations(self):
        return 0 if self._raw is None else int(self._raw.iterations)

    def getNumberOfInliers(self):
        return 0 if self._raw is None else int(self._raw.n_inliers)

    def targetInlierCountAchieved(self):
        return self._raw is not None and bool(self._raw.target_reached)

    def getDeviceMilliseconds(self):
        return 0.0 if self._raw is None else float(self._raw.device_ms)

    # ---- ransac_transform_estimator.hpp:61-104 -------------------------------------------------------
    def estimateModel(self):
        """estimateTransformPointToPointMetric over ALL pairs"""
        T = np.zeros(16, np.float32)
        rc = self._L.cilhip_transform_fit3f(self._device, self._dst.ctypes.data, self._src.ctypes.data, self._n, capi.MEM_HOST, T.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_transform_fit3f failed")
        return T.reshape(4, 4).T.copy()

    def countInliers(self, transforms, max_residual=None):
        """inlier counts of given transforms (m, 4, 4) -- the scoring half of the loop, one pass per 64 transforms"""
        Ts = np.ascontiguousarray(np.asarray(transforms, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1, 16)
        cnt = np.zeros(len(Ts), np.uint32)
        thr = self.inlier_dist_thresh_ if max_residual is None else max_residual
        rc = self._L.cilhip_transform_score3f(self._device, self._dst.ctypes.data, self._src.ctypes.data, self._n, capi.MEM_HOST, Ts.ctypes.data,
                                              len(Ts), C.c_float(thr), cnt.ctypes.data)
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_transform_score3f failed")
        return cnt.astype(np.int64)
