"""BASELINE configs[3] shape on ONE GPU: 10M source points (every 8th target point + noise) against an 80M-point
target, combined metric (w_p2p = 0.1, w_p2pl = 1).  Capability / timing check only (tools, not the bench):
the 8-GPU form shards the target (cilantro_amd.distributed.TargetShardedRigidICP)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import capi, synthetic as syn
from cilantro_amd.icp import Context
import ctypes as C

nd = int(float(sys.argv[1])) if len(sys.argv) > 1 else 80_000_000
t0 = time.perf_counter()
d = syn.make_pair(nd, nd // 8, with_normals=True, src_stride=8)
print(f"generated {nd} target / {nd // 8} source points in {time.perf_counter() - t0:.1f} s", flush=True)
ctx = Context(0)
t0 = time.perf_counter(); ctx.set_target(d["dst"], d["dst_n"]); ctx.synchronize(); t1 = time.perf_counter()
ctx.set_source(d["src"]); ctx.synchronize()
gi = ctx.grid_info()
print(f"set_target (upload + grid build) {1e3*(t1-t0):.0f} ms; grid {gi.nx}x{gi.ny}x{gi.nz} cells, occupancy {gi.avg_occupancy:.2f}", flush=True)
p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
p.metric = capi.METRIC_COMBINED; p.w_p2p = 0.1; p.w_p2pl = 1.0; p.conv_tol = 0.0; p.max_sq_dist = float(d["max_sq_dist"]); p.max_iter = 3
ctx.icp_run(p)
p.max_iter = 20
t0 = time.perf_counter(); r = ctx.icp_run(p); dt = time.perf_counter() - t0
T = np.array(r.T[:], np.float32).reshape(4, 4).T
dq, dtl = ctx.debug_counters()
print(f"10M<->{nd/1e6:g}M combined (0.1 / 1.0): {1e3*dt/20:.3f} ms/iteration = {20/dt:.0f} it/s, ncorr={r.last_ncorr}, |T-T_true|={np.linalg.norm(T-d['T_true']):.2e}, deferred queries {dq} tiles {dtl}")
