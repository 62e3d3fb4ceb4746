"""dev: frame_1 vs frame_2 at full resolution -- per-iteration distance of the GPU loop from the oracle's, next to the
distance between the oracle's own arithmetic modes (how ill-conditioned is the registration itself?)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402
from oracle import oracle as orc  # noqa: E402

f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
keep = p1[:, 0] > -0.4
D, N, S = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep]), np.ascontiguousarray(p2)
r2 = float(np.float32(0.02 * 0.02))
for iters in (1, 2, 3, 4, 6, 8, 12, 20, 40):
    Ts = {}
    for name, mode in (("mixed", orc.MODE_MIXED), ("f64", orc.MODE_F64), ("f32", orc.MODE_F32)):
        r = orc.icp_run(D, N, S, orc.make_params(metric=1, max_iter=iters, conv_tol=0.0, max_sq_dist=r2, mode=mode))
        Ts[name] = (r["T"].astype(np.float64), r["last_ncorr"], r.get("last_delta_norm"))
    ctx = Context()
    ctx.set_target(D, N); ctx.set_source(S)
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.max_sq_dist, p.max_iter, p.conv_tol = r2, iters, 0.0
    res = ctx.icp_run(p)
    Tg = np.array(res.T[:], np.float32).reshape(4, 4).T.astype(np.float64)
    ctx.close()
    print(iters, "gpu-mixed %.2e (ncorr %d vs %d)" % (np.linalg.norm(Tg - Ts["mixed"][0]), res.last_ncorr, Ts["mixed"][1]),
          "f64-mixed %.2e" % np.linalg.norm(Ts["f64"][0] - Ts["mixed"][0]), "f32-mixed %.2e" % np.linalg.norm(Ts["f32"][0] - Ts["mixed"][0]),
          "delta", res.last_delta_norm, flush=True)
