import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi
from cilantro_amd.icp import Context
from oracle import oracle as orc
f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
keep = p1[:, 0] > -0.4
D, N, S = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep]), np.ascontiguousarray(p2)
r2 = float(np.float32(0.02 * 0.02))
T = np.eye(4, dtype=np.float32)
ctx = Context(); ctx.set_target(D, N); ctx.set_source(S)
ctx.find_correspondences(T, r2, count=False)
gi, gd = ctx.get_nn(); gi = gi.astype(np.int64); gi[gi == capi.NONE_IDX] = -1
tree = orc.KDTree(D, use_ref=orc.ref_available())
o1, o2, ov = tree.find_correspondences(orc.transform_points(T, S), r2)
oi = np.full(len(S), -1, np.int64); od = np.zeros(len(S), np.float32); oi[o2] = o1; od[o2] = ov
bad = np.nonzero(gi != oi)[0]
print("mismatches", len(bad), "ties", int(np.sum(gd[bad] == od[bad])), "found gpu", (gi >= 0).sum(), "oracle", (oi >= 0).sum())
for i in bad[:5]: print(i, gi[i], oi[i], gd[i], od[i], D[gi[i]], D[oi[i]], N[gi[i]], N[oi[i]])
p = orc.make_params(metric=1, max_iter=1, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED)
m = gi >= 0
Tg_or, _ = orc.icp_update(D, N, S, T, gi[m], np.nonzero(m)[0], p)
To_or, _ = orc.icp_update(D, N, S, T, o1, o2, p)
pp = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(pp)); pp.max_sq_dist, pp.max_iter, pp.conv_tol = r2, 1, 0.0
res = ctx.icp_run(pp); Tg = np.array(res.T[:], np.float32).reshape(4, 4).T.astype(np.float64)
print("oracle update from gpu matches vs from nanoflann matches: %.3e" % np.linalg.norm(Tg_or.astype(np.float64) - To_or.astype(np.float64)))
print("gpu run vs oracle update from gpu matches: %.3e ; vs from nanoflann matches %.3e" % (np.linalg.norm(Tg - Tg_or), np.linalg.norm(Tg - To_or)))
