"""Worker of tests/test_gpu_distributed.py: the product's sharded loops (cilantro_amd/distributed.py) with the PRODUCT's per-rank
engines (HipShardEngine / HipSlabEngine / HipTargetShardEngine over libcilantro_hip.so), one process per rank, all ranks on cuda:0 (one GPU per box), the
48-double exchange over gloo.  Prints rank 0's result as JSON."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, distributed, synthetic as syn  # noqa: E402


def clouds(kind, n):
    d = syn.make_pair(n, perturb=0.5)
    if kind == "dup":      # doubled and tripled target points: exact ties in every search (option tie_rule: the reference's order)
        rng = np.random.default_rng(3)
        dup = rng.choice(len(d["dst"]), n // 30, replace=False)
        d["dst"] = np.ascontiguousarray(np.concatenate([d["dst"], d["dst"][dup], d["dst"][dup[: len(dup) // 6]]]))
        d["dst_n"] = np.ascontiguousarray(np.concatenate([d["dst_n"], d["dst_n"][dup], d["dst_n"][dup[: len(dup) // 6]]]))
    return d


def main_models(mode, kind, n, iters):
    """SURVEY.md 8(e), last row: KMeans / RANSAC scoring with the points sharded over the ranks (cilantro_amd/distributed_models.py),
    HIP shards on every rank, against the single-device entry points on rank 0"""
    from cilantro_amd import distributed_models as dm
    from cilantro_amd.clustering import KMeans3f
    from cilantro_amd.model_estimation import PlaneRANSACEstimator3f

    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(11)
    x = rng.random((n, 3), dtype=np.float32)
    out = {"rank": rank}
    cut = [0, n // 3, n] if world == 2 else [round(i * n / world) for i in range(world + 1)]
    lo, hi = cut[rank], cut[rank + 1]
    if mode == "kmeans":
        k, tol, kd = (1024 if kind == "big" else 64), (2e-2 if kind == "tol" else 0.0), kind == "kd"
        c0 = x[:k].copy()
        if kind == "empty":
            c0[5] = [50.0, 50.0, 50.0]; c0[40] = [-40.0, 3.0, 2.0]
        if kind == "kd":                # lattice centroids + half-lattice points: exact ties, the reference's tree order on every rank
            g = 4
            c0 = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)[rng.permutation(g ** 3)]
            x = (rng.integers(-1, 2 * g + 1, size=(n, 3)) * 0.5).astype(np.float32)
        eng = dm.HipKMeansShard(x[lo:hi], len(c0), lo, 0)
        km = dm.ShardedKMeans3f(eng, dist).cluster(c0, max_iter=iters, tol=tol, use_kd_tree=kd)
        out.update(cent=km.getClusterCentroids().astype(np.float64).tolist(), lab=km.getPointToClusterIndexMap().tolist(), it=km.getNumberOfPerformedIterations())
        eng.close()
        if rank == 0:
            one = KMeans3f(x).cluster(c0, max_iter=iters, tol=tol, use_kd_tree=kd)
            out.update(one_cent=one.getClusterCentroids().astype(np.float64).tolist(), one_lab=one.getPointToClusterIndexMap().tolist(), one_it=one.getNumberOfPerformedIterations())
    else:
        x = x * 2 - 1
        x[: n // 2, 2] = 0.25 * x[: n // 2, 0] + 0.1
        nrm = rng.normal(size=(256, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        planes = np.concatenate([nrm, rng.uniform(-0.5, 0.5, (256, 1))], axis=1).astype(np.float32)
        pe = PlaneRANSACEstimator3f(x[lo:hi]).setMaxInlierResidual(0.02)
        tot = dm.sharded_plane_inlier_counts(pe.countInliers, planes, dist)
        out.update(counts=np.asarray(tot).tolist())
        if rank == 0:
            out.update(one_counts=np.asarray(PlaneRANSACEstimator3f(x).setMaxInlierResidual(0.02).countInliers(planes)).tolist())
    rows = [None] * world
    dist.all_gather_object(rows, out)
    if rank == 0:
        print("RESULT " + json.dumps({"world": world, "rows": rows}))
    dist.destroy_process_group()


def main():
    import signal
    signal.alarm(600)
    mode, kind, n, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    if mode in ("kmeans", "ransac"):
        return main_models(mode, kind, n, iters)
    d = clouds(kind, n)
    p = distributed.default_params(capi.METRIC_COMBINED, max_sq_dist=float(d["max_sq_dist"]), max_iter=iters, conv_tol=0.0)
    T0 = np.eye(4, dtype=np.float32)
    extra = {}
    if mode == "source":
        lo, hi = distributed.shard_bounds(len(d["src"]), rank, world)
        eng = distributed.HipShardEngine(d["dst"], d["dst_n"], d["src"][lo:hi], 0)
        T, it, delta, nc = distributed.ShardedRigidICP(eng, dist).estimate(p, T0, check_every=0)
        extra["tables_loaded"] = eng.ctx.tie_order_info()["loaded"]
    elif mode == "tshard":      # partitioning A: index shards of the target, MIN of packed keys (+ the traversal keys once ties were met)
        lo, hi = distributed.shard_bounds(len(d["dst"]), rank, world)
        dm = d["dst"].astype(np.float64).mean(axis=0).astype(np.float32)
        eng = distributed.HipTargetShardEngine(d["dst"][lo:hi], d["dst_n"][lo:hi], d["src"], lo, dm, 0, whole_target=d["dst"])
        T, it, delta, nc = distributed.TargetShardedRigidICP(eng, dist).estimate(p, T0)
        extra["tables_loaded"] = eng.ctx.tie_order_info()["loaded"]
    else:
        slack = None if mode == "slab" else float(mode[4:]) * d["h"]
        part = distributed.SlabPartition.plan(d["dst"], d["src"], T0, float(d["max_sq_dist"]), world, slack=slack)
        eng = distributed.HipSlabEngine(part, rank, d["dst"], d["dst_n"], d["src"], 0)

        def repartition(T):
            pt = distributed.SlabPartition.plan(d["dst"], d["src"], T, float(d["max_sq_dist"]), world, slack=slack)
            return distributed.HipSlabEngine(pt, rank, d["dst"], d["dst_n"], d["src"], 0)

        icp = distributed.SlabShardedRigidICP(eng, dist, repartition=repartition)
        T, it, delta, nc = icp.estimate(p, T0, check_every=2)
        extra["repartitions"] = icp.repartitions
        extra["tables_loaded"] = icp.engine.ctx.tie_order_info()["loaded"]
        extra["n_target_local"] = int(icp.engine.ctx.n_target) if hasattr(icp.engine.ctx, "n_target") else -1
    rows = [None] * world
    dist.all_gather_object(rows, {"rank": rank, "T": np.asarray(T, np.float64).tolist(), "it": int(it), "nc": int(nc), **extra})
    if rank == 0:
        print("RESULT " + json.dumps({"world": world, "rows": rows}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
