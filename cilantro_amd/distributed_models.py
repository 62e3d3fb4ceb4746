"""The next-tier consumers across ranks (last row): KMeans3f and the plane-RANSAC scoring pass with the
POINTS sharded -- one process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm), centroids / hypotheses replicated.

KMeans (clustering/kmeans.hpp:67-194).  One Lloyd iteration has exactly one exchange step:

    every rank:  a shard of the points (+ their labels), all k centroids
    per iteration:  local assignment + local cluster sums            (HIP: cilhip_kmeans_shard_assign, no communication)
                    all-reduce(sum) of 4k + 1 int64                   (RCCL; k = 1024: 32 KB)
                    kmeans.hpp:122-188 on the summed values           (host, identical on every rank)

The cluster sums are exact fixed-point integers (one scale for all shards), so the summed values do not depend on how the points
were cut: centroids, labels and the iteration count are the single-device run's bit for bit.  The empty-cluster repair (:134-176)
adds two small collectives per empty cluster (MAX of a packed key, SUM of three coordinates); it is rare.

RANSAC (model_estimation/ransac_base.hpp:81-115, ransac_hyperplane_estimator.hpp:47-55): the scoring pass -- H hypotheses
against all points -- is an inlier COUNT per hypothesis: local counts, all-reduce(sum) of H integers.

The loops are engine-agnostic (like cilantro_amd.distributed): they drive any object with the methods of HipKMeansShard / a
`count_inliers(planes)` callable; the CPU (gloo, world_size 2) tests plug in test-only engines.
"""
import ctypes as C

import numpy as np

from . import capi
from .icp import _as_cloud


def scale_exponent(maxabs_all, n_all):
    """cilhip_kmeans_scale_exponent: |x| * 2^S < 2^(62 - ceil(log2 n)) (a whole cluster's sum cannot overflow int64)"""
    e = int(np.frexp(float(maxabs_all) if maxabs_all > 0.0 else 1.0)[1])
    nbits = 0
    while (1 << nbits) < int(n_all):
        nbits += 1
    return 62 - nbits - e


class HipKMeansShard:
    """This rank's points on its device (cilhip_kmeans_shard_*): the product engine of ShardedKMeans3f."""

    def __init__(self, points, k, index_offset=0, device=0):
        self._L = capi.load()
        p, n, mem, self._keep = _as_cloud(points)
        self.n, self.k, self.index_offset = int(n), int(k), int(index_offset)
        h = C.c_void_p()
        rc = self._L.cilhip_kmeans_shard_create(device, p, n, mem, self.k, C.c_uint64(self.index_offset), C.byref(h))
        if rc != capi.OK:
            raise capi.CilhipError(rc, "cilhip_kmeans_shard_create failed (no HIP device, k > 2048, or bad arguments)")
        self._h = h

    def _ck(self, rc, what):
        if rc != capi.OK:
            raise capi.CilhipError(rc, what + " failed")

    def maxabs(self):
        v = C.c_float(0.0)
        self._ck(self._L.cilhip_kmeans_shard_maxabs(self._h, C.byref(v)), "cilhip_kmeans_shard_maxabs")
        return float(v.value)

    def assign(self, centroids, scale_exp, use_kd_tree=False):
        """-> (sums int64 [k, 4] = {x, y, z (fixed point, 2^scale_exp), count}, labels changed)"""
        cent = np.ascontiguousarray(centroids, np.float32).reshape(-1, 3)
        sums = np.zeros((self.k, 4), np.int64)
        ch = C.c_uint64(0)
        self._ck(self._L.cilhip_kmeans_shard_assign(self._h, cent.ctypes.data, int(scale_exp), int(bool(use_kd_tree)), sums.ctypes.data, C.byref(ch)),
                 "cilhip_kmeans_shard_assign")
        return sums, int(ch.value)

    def farthest(self, cluster, center):
        c = np.ascontiguousarray(center, np.float32).reshape(3)
        key = C.c_uint64(0)
        self._ck(self._L.cilhip_kmeans_shard_farthest(self._h, int(cluster), c.ctypes.data, C.byref(key)), "cilhip_kmeans_shard_farthest")
        return int(key.value)

    def move_point(self, global_index, to_cluster):
        p = np.zeros(3, np.float32)
        self._ck(self._L.cilhip_kmeans_shard_move_point(self._h, C.c_uint64(int(global_index)), int(to_cluster), p.ctypes.data), "cilhip_kmeans_shard_move_point")
        return p

    def labels(self):
        out = np.zeros(max(self.n, 1), np.uint32)
        self._ck(self._L.cilhip_kmeans_shard_labels(self._h, out.ctypes.data), "cilhip_kmeans_shard_labels")
        return out[:self.n].astype(np.int64)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cilhip_kmeans_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedKMeans3f:
    """KMeans<float,3>::cluster(centroids, max_iter, tol, use_kd_tree) (clustering/kmeans.hpp:24-30 -> cluster_ :67-194) with the
    points cut over the ranks of ``dist`` (torch.distributed, initialised; None: one process).  ``engine``: this rank's shard
    (HipKMeansShard).  ``device``: where the small tensors of the collectives live ("cuda" under nccl, "cpu" under gloo).
    After cluster(): getClusterCentroids() (the same on every rank), getPointToClusterIndexMap() (THIS rank's points),
    getNumberOfPerformedIterations()."""

    def __init__(self, engine, dist=None, group=None, device="cpu"):
        self.engine, self.dist, self.group, self.device = engine, dist, group, device
        self.cluster_centroids_ = None
        self.point_to_cluster_index_map_ = None
        self.iteration_count_ = 0

    def _reduce(self, arr, op):
        """all-reduce of a small int64 / float64 numpy array (returned as numpy)"""
        if self.dist is None or self.dist.get_world_size(self.group) <= 1:
            return arr
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op), group=self.group)
        return t.cpu().numpy()

    def cluster(self, centroids, max_iter=100, tol=float(np.finfo(np.float32).eps), use_kd_tree=False, fetch_labels=True):
        """fetch_labels=False leaves the labels on the device (getPointToClusterIndexMap() then fetches them on demand)"""
        eng = self.engine
        cent = np.ascontiguousarray(centroids, np.float32).reshape(-1, 3).copy()
        k = len(cent)
        n_all = int(self._reduce(np.array([eng.n], np.int64), "SUM")[0])
        maxabs_all = float(self._reduce(np.array([eng.maxabs()], np.float64), "MAX")[0])
        S = scale_exponent(maxabs_all, n_all)
        scale = float(np.ldexp(1.0, S))
        tol = np.float32(tol)
        tol_sq = np.float32(tol * tol)
        it = 0
        while it < int(max_iter):
            sums, changed = eng.assign(cent, S, use_kd_tree)
            red = self._reduce(np.concatenate([sums.reshape(-1), np.array([changed], np.int64)]), "SUM")
            hs, changed = red[:-1].reshape(k, 4).copy(), int(red[-1])
            if changed == 0 and it > 0:                                           # kmeans.hpp:122
                break
            c_old = cent.copy()                                                   # :123
            for i in range(k):                                                    # empty clusters (:134-176), ascending like the reference
                if hs[i, 3] != 0:
                    continue
                mx = int(np.argmax(hs[:, 3]))                                     # (first maximum: strict '>' over ascending j)
                cm = np.float64(hs[mx, 3])
                oc = (hs[mx, :3].astype(np.float64) / scale / cm).astype(np.float32)
                key = int(self._reduce(np.array([eng.farthest(mx, oc)], np.int64), "MAX")[0])      # (bits(d) < 2^31: the key is a positive int64)
                gidx = 0xFFFFFFFF - (key & 0xFFFFFFFF)
                mine = key != 0 and eng.index_offset <= gidx < eng.index_offset + eng.n
                p = eng.move_point(gidx, i).astype(np.float64) if mine else np.zeros(3)
                p = self._reduce(p, "SUM").astype(np.float32)                     # (one owner: the sum is its value, exactly)
                hs[mx, :3] -= np.rint(p.astype(np.float64) * scale).astype(np.int64)
                hs[mx, 3] -= 1
                hs[i, 3] += 1                                                     # the reference does not add the point to cluster i's sum (:171-175)
            with np.errstate(divide="ignore", invalid="ignore"):
                cent = (hs[:, :3].astype(np.float64) / scale / hs[:, 3:4].astype(np.float64)).astype(np.float32)      # :179-181
            it += 1
            if tol > 0:                                                           # :186-188
                d = cent - c_old
                sq = d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
                mxs = np.float32(0.0)
                for v in sq:                                                      # (a NaN never exceeds: the C loop's `if (sq > mxs)`)
                    if v > mxs:
                        mxs = v
                if mxs < tol_sq:
                    break
        self.cluster_centroids_ = cent
        self.point_to_cluster_index_map_ = eng.labels() if fetch_labels else None
        self.iteration_count_ = it
        return self

    def getClusterCentroids(self):
        return self.cluster_centroids_

    def getPointToClusterIndexMap(self):
        if self.point_to_cluster_index_map_ is None and self.cluster_centroids_ is not None:
            self.point_to_cluster_index_map_ = self.engine.labels()
        return self.point_to_cluster_index_map_

    def getNumberOfPerformedIterations(self):
        return self.iteration_count_


def sharded_plane_inlier_counts(count_inliers, planes, dist=None, group=None, device="cpu"):
    """The RANSAC scoring pass over sharded points: ``count_inliers(planes) -> counts`` scores this rank's shard (the product:
    ``PlaneRANSACEstimator3f(shard).setMaxInlierResidual(r).countInliers``), the counts are summed over the ranks -- every rank
    then picks the same best hypothesis (ransac_base.hpp:100-113)."""
    counts = np.ascontiguousarray(count_inliers(planes), np.int64)
    if dist is None or dist.get_world_size(group) <= 1:
        return counts
    import torch

    t = torch.from_numpy(counts).to(device)
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()
