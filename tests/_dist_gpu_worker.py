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


def main():
    import signal
    signal.alarm(600)
    mode, kind, n, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
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
