"""Python mirror of cilantro's KMeans3f (clustering/kmeans.hpp) on top of the C ABI (cilhip_kmeans3f).

    km = KMeans3f(points)
    km.cluster(initial_centroids, max_iter=100, tol=eps)      # kmeans.hpp:24-30
    km.cluster(num_clusters, ...)                               # :32-53 (random initial centroids)
    km.getClusterCentroids(); km.getPointToClusterIndexMap(); km.getClusterToPointIndicesMap()

`use_kd_tree=True` (kmeans.hpp:86-94, the mode examples/kmeans.cpp uses): a kd-tree over the centroids only accelerates the same
nearest-centroid search, so the device runs the same exhaustive pass -- with the distance rounded as nanoflann's L2 metric rounds
it (((dx*dx)+(dy*dy))+(dz*dz)), so labels equal that branch's wherever the nearest centroid is unique.
"""
import ctypes as C

import numpy as np

from . import capi
from .icp import _as_cloud


class KMeans3f:
    def __init__(self, data, device=0):
        self._L = capi.load()
        self._data = data
        self._device = device
        self.cluster_centroids_ = None
        self.point_to_cluster_index_map_ = None
        self.iteration_count_ = 0

    def cluster(self, centroids_or_k, max_iter=100, tol=float(np.finfo(np.float32).eps), use_kd_tree=False, seed=None):
        p, n, mem, keep = _as_cloud(self._data)
        if np.isscalar(centroids_or_k):
            # kmeans.hpp:32-53 draws distinct random points (std::random_device): same law, numpy generator
            k = max(1, min(int(centroids_or_k), n))
            idx = np.random.default_rng(seed).choice(n, size=k, replace=False)
            host = keep.cpu().numpy() if hasattr(keep, "cpu") else np.asarray(keep)
            cent = np.ascontiguousarray(host[idx], np.float32)
        else:
            cent = np.ascontiguousarray(centroids_or_k, np.float32).reshape(-1, 3).copy()
        labels = np.zeros(n, np.uint32)
        iters = C.c_size_t(0)
        rc = self._L.cilhip_kmeans3f_ex(self._device, p, n, mem, cent.ctypes.data, len(cent), int(max_iter), C.c_float(tol), int(bool(use_kd_tree)),
                                        labels.ctypes.data, C.byref(iters))
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_kmeans3f failed (no HIP device, k > 2048, or bad arguments)")
        self.cluster_centroids_ = cent
        self.point_to_cluster_index_map_ = labels.astype(np.int64)
        self.iteration_count_ = int(iters.value)
        return self

    def getClusterCentroids(self):
        return self.cluster_centroids_

    def getNumberOfPerformedIterations(self):
        return self.iteration_count_

    def getPointToClusterIndexMap(self):
        return self.point_to_cluster_index_map_

    def getNumberOfClusters(self):
        return 0 if self.cluster_centroids_ is None else len(self.cluster_centroids_)

    def getClusterToPointIndicesMap(self):
        """clustering_base.hpp:22-33: per cluster, ascending point indices"""
        order = np.argsort(self.point_to_cluster_index_map_, kind="stable")
        counts = np.bincount(self.point_to_cluster_index_map_, minlength=self.getNumberOfClusters())
        return np.split(order, np.cumsum(counts)[:-1])


def set_pruning(on=True):
    """the brute-force branch's assignment computed exactly with pruning (default) or exhaustively (cilhip_kmeans_set_pruning; process-wide)"""
    capi.load().cilhip_kmeans_set_pruning(1 if on else 0)


def kmeans_assign(data, centroids, device=0, use_kd_tree=False):
    L = capi.load()
    p, n, mem, keep = _as_cloud(data)
    cent = np.ascontiguousarray(centroids, np.float32).reshape(-1, 3)
    labels = np.zeros(n, np.uint32)
    rc = L.cilhip_kmeans3f_assign_ex(device, p, n, mem, cent.ctypes.data, len(cent), int(bool(use_kd_tree)), labels.ctypes.data)
    if rc != capi.OK:
        raise capi.CilhipError(rc, "cilhip_kmeans3f_assign failed")
    return labels.astype(np.int64)
