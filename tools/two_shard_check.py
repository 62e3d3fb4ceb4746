"""Dev check: two target shards on one GPU (element-wise MIN of keys, sum of partial sums) against the unsharded run."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from cilantro_amd import distributed, synthetic as syn
from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

for nd in [int(float(a)) for a in sys.argv[1:]] or [200_000]:
    d = syn.make_pair(nd, nd // 8, with_normals=True, src_stride=8)
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.setPointToPointMetricWeight(0.1).setPointToPlaneMetricWeight(1.0)
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    T1 = icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).estimate().getTransform()
    nc1 = icp.last_ncorr_
    dm, _ = icp._ctx.means()
    del icp
    half = nd // 2
    engs = [distributed.HipTargetShardEngine(d["dst"][lo:hi], d["dst_n"][lo:hi], d["src"], lo, dm, 0) for lo, hi in ((0, half), (half, nd))]
    p = distributed.default_params(max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    p.w_p2p = 0.1; p.w_p2pl = 1.0
    for e in engs:
        e.begin(p, np.eye(4, dtype=np.float32))
    for it in range(6):
        keys = torch.minimum(engs[0].partial_keys(), engs[1].partial_keys())
        sums = engs[0].sums_from_keys(keys).clone() + engs[1].sums_from_keys(keys)
        for e in engs:
            e.apply_sums(sums)
    Ta, ita, _, nca = engs[0].state()
    print(f"nd={nd}: |T_sharded - T_unsharded|max={np.abs(Ta.astype(np.float64)-T1).max():.3e} nc {nca} vs {nc1}; |T1-T_true|={np.linalg.norm(T1-d['T_true']):.3e} |Ta-T_true|={np.linalg.norm(Ta-d['T_true']):.3e}")
