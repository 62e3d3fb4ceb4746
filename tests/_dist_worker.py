"""Worker of tests/test_distributed_cpu.py: runs ShardedRigidICP over gloo with a TEST-ONLY per-rank
engine (oracle kNN + numpy estimator) and prints the final transform as JSON on rank 0."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, distributed, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (tests may use the oracle)


class OracleShardEngine:
    """Same interface as cilantro_amd.distributed.HipShardEngine, computed on the CPU by the checker."""

    def __init__(self, dst, dst_n, src_shard):
        self.dst, self.dst_n, self.src = dst, dst_n, np.ascontiguousarray(src_shard)
        self.tree = orc.KDTree(dst)
        self.sums = torch.zeros(distributed.SUMS_LEN, dtype=torch.float64)
        self.dst_mean = orc.mean3(dst)

    def local_source_sum(self):
        return self.src.astype(np.float64).sum(0), len(self.src)

    def begin(self, params, T0, gmean):
        self.p, self.T, self.gmean = params, np.asarray(T0, np.float32).copy(), np.asarray(gmean, np.float32)
        self.iters, self.delta, self.nc, self.done = 0, np.inf, 0, False

    def partial_sums(self):
        self.sums.zero_()
        if self.done:
            return self.sums
        q = orc.transform_points(self.T, self.src)
        di, si, _ = self.tree.find_correspondences(q, self.p.max_sq_dist, num_threads=1)
        s = np.zeros(distributed.SUMS_LEN)
        if self.p.metric == capi.METRIC_POINT_TO_POINT:
            _, s16, _ = orc.estimate_p2p(self.dst, q, di, si, orc.MODE_MIXED)
            s[:16] = s16
        else:
            smt = orc.transform_points(self.T, self.gmean.reshape(1, 3))[0]
            _, AtA, Atb, _ = orc.estimate_combined(self.dst, self.dst_n, q, di, si, 0.0, 1.0, self.dst_mean, smt, 1, 1e-5, orc.MODE_MIXED)
            s[0] = len(di)
            s[1:22] = AtA[np.triu_indices(6)]
            s[22:28] = Atb
        self.sums.copy_(torch.from_numpy(s))
        return self.sums

    def apply_sums(self, sums):
        if self.done:
            return
        s = sums.numpy()
        n = s[0]
        L, t = np.eye(3), np.zeros(3)
        if self.p.metric == capi.METRIC_POINT_TO_POINT:
            if n > 0:
                mp, mq = s[1:4] / n, s[4:7] / n
                S = s[7:16].reshape(3, 3) / n - np.outer(mp, mq)
                U, _, Vt = np.linalg.svd(S)
                if np.linalg.det(U @ Vt) < 0:
                    U[:, 2] *= -1
                L = U @ Vt
                t = mp - L @ mq
        elif n > 0:
            AtA = np.zeros((6, 6)); AtA[np.triu_indices(6)] = s[1:22]; AtA = AtA + np.triu(AtA, 1).T
            x = np.linalg.solve(AtA, s[22:28])
            na = np.linalg.norm(x[:3]); th = np.arctan(na); u = x[:3] / na if na > 0 else np.zeros(3)
            K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
            Ra = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
            L = Ra @ Ra
            smt = orc.transform_points(self.T, self.gmean.reshape(1, 3))[0].astype(np.float64)
            t = Ra @ (np.cos(th) * x[3:]) - L @ smt + self.dst_mean.astype(np.float64)
        R = orc.nearest_rotation(L)
        Tn = np.eye(4)
        Tn[:3, :3] = R @ self.T[:3, :3].astype(np.float64)
        Tn[:3, 3] = R @ self.T[:3, 3].astype(np.float64) + t
        self.T = Tn.astype(np.float32)
        self.delta = float(np.sqrt(((R - np.eye(3)) ** 2).sum() + (t ** 2).sum()))
        self.iters += 1
        self.nc = int(round(n))
        self.done = self.delta < self.p.conv_tol

    def state(self):
        return self.T, self.iters, self.delta, self.nc


class NativeLoopMixin:
    """What cilantro_amd.distributed.HipShardEngine.enable_native_allreduce / iterate give the loops: blocks of iterations run by the
    engine itself with its own all-reduce (here: gloo) -- the control flow of ShardedRigidICP / SlabShardedRigidICP around such an engine."""
    native = False

    def enable_native_allreduce(self, dist_, group=None):
        self._dist, self._group, self.native = dist_, group, True
        return True

    def iterate(self, k):
        for _ in range(int(k)):
            sums = self.partial_sums()
            self._dist.all_reduce(sums, group=self._group)
            self.apply_sums(sums)


class OracleTargetShardEngine(OracleShardEngine):
    """Test-only counterpart of HipTargetShardEngine: kd-tree over dst[lo:hi), keys with global indices."""

    def __init__(self, dst_full, dst_n_full, src, lo, hi):
        super().__init__(np.ascontiguousarray(dst_full[lo:hi]), np.ascontiguousarray(dst_n_full[lo:hi]), src)
        self.lo, self.hi = lo, hi
        self.dst_mean = orc.mean3(dst_full)                       # GLOBAL target mean
        self.keys = torch.full((len(src),), distributed.KEY_NONE, dtype=torch.int64)

    def begin(self, params, T0):
        super().begin(params, T0, orc.mean3(self.src))

    def partial_keys(self):
        self.keys.fill_(distributed.KEY_NONE)
        if not self.done:
            self.q = orc.transform_points(self.T, self.src)
            di, si, d2 = self.tree.find_correspondences(self.q, self.p.max_sq_dist, num_threads=1)
            k = (d2.view(np.uint32).astype(np.int64) << 32) | (di + self.lo)
            self.keys[torch.from_numpy(si)] = torch.from_numpy(k)
        return self.keys

    def sums_from_keys(self, keys):
        self.sums.zero_()
        if self.done:
            return self.sums
        k = keys.numpy()
        gidx = k & 0xFFFFFFFF
        mine = (k != distributed.KEY_NONE) & (gidx >= self.lo) & (gidx < self.hi)
        si = np.nonzero(mine)[0].astype(np.int64)
        di = (gidx[mine] - self.lo).astype(np.int64)
        s = np.zeros(distributed.SUMS_LEN)
        if self.p.metric == capi.METRIC_POINT_TO_POINT:
            _, s16, _ = orc.estimate_p2p(self.dst, self.q, di, si, orc.MODE_MIXED)
            s[:16] = s16
        else:
            smt = orc.transform_points(self.T, self.gmean.reshape(1, 3))[0]
            _, AtA, Atb, _ = orc.estimate_combined(self.dst, self.dst_n, self.q, di, si, 0.0, 1.0, self.dst_mean, smt, 1, 1e-5, orc.MODE_MIXED)
            s[0] = len(di); s[1:22] = AtA[np.triu_indices(6)]; s[22:28] = Atb
        self.sums.copy_(torch.from_numpy(s))
        return self.sums


class OrderedTargetShardEngine(OracleTargetShardEngine):
    """Test-only counterpart of HipTargetShardEngine WITH the whole target at hand (whole_target=...): the shard's exact nearest
    points per query, all of them when several are equidistant (scipy neighbours + the pinned distance expression); without the order
    the lowest index of the shard and a count of what was tied (inside the shard, and against the winner of the MIN), with it the
    first-met one of the reference's tree over the WHOLE target and the traversal key of that match -- the host restatement
    (csrc/tie_order.hpp behind tests/cpp/tie_order_shim.cpp), which tests/test_tie_order_cpu.py pins against the reference's nanoflann."""
    K = 12

    def __init__(self, dst_full, dst_n_full, src, lo, hi):
        import ctypes as C

        from scipy.spatial import cKDTree

        super().__init__(dst_full, dst_n_full, src, lo, hi)
        self.dst_full = np.ascontiguousarray(dst_full, np.float32)
        self.ck = cKDTree(self.dst.astype(np.float64))
        self.L = C.CDLL(os.path.join(ROOT, "tests", "cpp", "bin", "libtie_order_shim.so"))
        self.L.tie_shim_build.restype = C.c_void_p
        self.L.tie_shim_build.argtypes = [C.c_void_p, C.c_uint32]
        self.L.tie_shim_first_met.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_void_p]
        self.L.tie_shim_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        self.h, self.ordered, self.pending = None, False, 0
        self.okeys = torch.full((len(src),), distributed.KEY_NONE, dtype=torch.int64)

    @staticmethod
    def _pinned_d2(a, b):
        d = (a - b).astype(np.float32)
        return ((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])).astype(np.float32) + (d[..., 2] * d[..., 2]).astype(np.float32)

    def begin(self, params, T0):
        super().begin(params, T0)
        self.pending = 0

    def partial_keys(self):
        self.keys.fill_(distributed.KEY_NONE)
        self.m_si = np.zeros(0, np.int64)
        if self.done:
            return self.keys
        self.q = orc.transform_points(self.T, self.src)
        _, si, d2 = self.tree.find_correspondences(self.q, self.p.max_sq_dist, num_threads=1)      # which queries match inside the radius, at what distance
        qm = np.ascontiguousarray(self.q[si])
        _, nb = self.ck.query(qm.astype(np.float64), k=self.K)
        dd = self._pinned_d2(qm[:, None, :], self.dst[nb])
        tied = dd == d2[:, None]
        cnt = tied.sum(axis=1)
        assert cnt.min() >= 1 and cnt.max() < self.K
        cand = np.zeros((len(si), self.K), np.uint32)
        order = np.argsort(np.where(tied, nb, np.iinfo(np.int64).max), axis=1)      # tied ones first, ascending index
        cand[:] = (np.take_along_axis(nb, order, axis=1) + self.lo).astype(np.uint32)
        if self.ordered:
            pick = np.zeros(len(si), np.uint32)
            cnt32 = np.ascontiguousarray(cnt, np.int32)      # (kept alive across the call)
            self.L.tie_shim_first_met(self.h, qm.ctypes.data, cand.ctypes.data, cnt32.ctypes.data, self.K, len(si), pick.ctypes.data)
        else:
            pick = cand[:, 0].copy()                      # the lowest index
            self.pending += int(np.count_nonzero(cnt >= 2))
        self.m_si, self.m_gi, self.m_d2 = si.astype(np.int64), pick.astype(np.int64), d2
        k = (d2.view(np.uint32).astype(np.int64) << 32) | self.m_gi
        self.keys[torch.from_numpy(self.m_si)] = torch.from_numpy(k)
        return self.keys

    def _accumulate(self, si, gi):
        self.sums.zero_()
        self.last_won = (si, gi)
        di = gi - self.lo
        s = np.zeros(distributed.SUMS_LEN)
        if self.p.metric == capi.METRIC_POINT_TO_POINT:
            _, s16, _ = orc.estimate_p2p(self.dst, self.q, di, si, orc.MODE_MIXED)
            s[:16] = s16
        else:
            smt = orc.transform_points(self.T, self.gmean.reshape(1, 3))[0]
            _, AtA, Atb, _ = orc.estimate_combined(self.dst, self.dst_n, self.q, di, si, 0.0, 1.0, self.dst_mean, smt, 1, 1e-5, orc.MODE_MIXED)
            s[0] = len(di); s[1:22] = AtA[np.triu_indices(6)]; s[22:28] = Atb
        self.sums.copy_(torch.from_numpy(s))
        return self.sums

    def sums_from_keys(self, keys):
        if self.done:
            self.sums.zero_()
            return self.sums
        k = keys.numpy()[self.m_si]
        won = (k & 0xFFFFFFFF) == self.m_gi
        self.pending += int(np.count_nonzero(~won & ((k >> 32) == (self.m_d2.view(np.uint32).astype(np.int64)))))      # as far as the winner, not the winner
        return self._accumulate(self.m_si[won], self.m_gi[won])

    def ties_pending(self):
        return (not self.ordered) and self.pending > 0

    def load_tie_order(self):
        self.h = self.L.tie_shim_build(self.dst_full.ctypes.data, len(self.dst_full))
        self.ordered = True

    def order_keys(self, keys):
        self.okeys.fill_(distributed.KEY_NONE)
        if self.done or len(self.m_si) == 0:
            return self.okeys
        k = keys.numpy()[self.m_si]
        at = (k >> 32) == self.m_d2.view(np.uint32).astype(np.int64)
        qa = np.ascontiguousarray(self.q[self.m_si[at]]); ia = np.ascontiguousarray(self.m_gi[at].astype(np.uint32))
        out = np.zeros(len(ia), np.uint64)
        self.L.tie_shim_keys(self.h, qa.ctypes.data, ia.ctypes.data, len(ia), out.ctypes.data)
        self.own = np.full(len(self.src), distributed.KEY_NONE, np.int64)
        self.own[self.m_si[at]] = out.astype(np.int64)
        self.okeys.copy_(torch.from_numpy(self.own))
        return self.okeys

    def sums_from_ordered_keys(self, keys, okeys):
        if self.done:
            self.sums.zero_()
            return self.sums
        o = okeys.numpy()[self.m_si]
        won = (self.own[self.m_si] != distributed.KEY_NONE) & (self.own[self.m_si] == o)
        return self._accumulate(self.m_si[won], self.m_gi[won])


def main_target_sharded_ties(metric, n):
    """index shards of a target with doubled and tripled points, shuffled (ties inside the shards and across them): the protocol of
    TargetShardedRigidICP -- a first run that notices, the order, a second run with the third collective -- over gloo; the last
    iteration's pairs, gathered from the ranks that won them, against the reference's nanoflann over the whole target"""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d = syn.make_pair(n, perturb=0.5)
    rng = np.random.default_rng(17)
    dup = rng.choice(n, n // 10, replace=False)
    dst = np.concatenate([d["dst"], d["dst"][dup], d["dst"][dup[: n // 40]]]); dst_n = np.concatenate([d["dst_n"], d["dst_n"][dup], d["dst_n"][dup[: n // 40]]])
    perm = rng.permutation(len(dst))
    dst, dst_n = np.ascontiguousarray(dst[perm]), np.ascontiguousarray(dst_n[perm])
    lo, hi = distributed.shard_bounds(len(dst), rank, world)
    eng = OrderedTargetShardEngine(dst, dst_n, d["src"], lo, hi)
    p = distributed.default_params(metric=metric, max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    T, iters, delta, nc = distributed.TargetShardedRigidICP(eng, dist).estimate(p)
    rows = [None] * world
    dist.all_gather_object(rows, {"T": T.tolist(), "ordered": bool(eng.ordered), "si": eng.last_won[0].tolist(), "gi": eng.last_won[1].tolist()})
    if rank == 0:
        got = np.full(len(d["src"]), -1, np.int64); times = np.zeros(len(d["src"]), np.int64)
        for r in rows:
            got[np.array(r["si"], np.int64)] = np.array(r["gi"], np.int64); times[np.array(r["si"], np.int64)] += 1
        tree = orc.KDTree(dst, use_ref=orc.ref_available())
        o1, o2, _ = tree.find_correspondences(eng.q, float(d["max_sq_dist"]))
        want = np.full(len(d["src"]), -1, np.int64); want[o2] = o1
        low = np.full(len(d["src"]), -1, np.int64)      # what one key alone would have named: the lowest index among the equidistant ones
        from scipy.spatial import cKDTree
        _, nb = cKDTree(dst.astype(np.float64)).query(eng.q[o2].astype(np.float64), k=6)
        dd = OrderedTargetShardEngine._pinned_d2(eng.q[o2][:, None, :], dst[nb])
        low[o2] = np.where(dd == dd.min(axis=1, keepdims=True), nb, np.iinfo(np.int64).max).min(axis=1)
        if os.environ.get("DIST_DEBUG"):
            for i in np.nonzero(got != want)[0][:5]:
                dq = OrderedTargetShardEngine._pinned_d2(eng.q[i][None, :], dst[[got[i], want[i]]]) if got[i] >= 0 and want[i] >= 0 else None
                print("DEBUG", int(i), int(got[i]), int(want[i]), None if dq is None else dq.view(np.uint32).tolist(), dst[got[i]].tolist() if got[i] >= 0 else None, dst[want[i]].tolist() if want[i] >= 0 else None, eng.q[i].tolist(), file=sys.stderr)
        po = orc.make_params(metric=metric, max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]), mode=orc.MODE_MIXED)
        ro = orc.icp_run(dst, dst_n, d["src"], po)
        print("RESULT " + json.dumps({"world": world, "identical": all(r["T"] == rows[0]["T"] for r in rows), "ordered": [r["ordered"] for r in rows],
                                       "iters": iters, "ncorr": nc, "oracle_ncorr": int(ro["last_ncorr"]), "T_err": float(np.linalg.norm(np.array(T, np.float64) - ro["T"])),
                                       "mismatches": int(np.count_nonzero(got != want)), "won_once": bool(np.array_equal(times, (want >= 0).astype(np.int64))),
                                       "lowest_index_would_differ": int(np.count_nonzero(low != want))}))
    dist.destroy_process_group()


def main_target_sharded(metric, n):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d = syn.make_pair(n, perturb=0.5)
    lo, hi = distributed.shard_bounds(n, rank, world)
    eng = OracleTargetShardEngine(d["dst"], d["dst_n"], d["src"], lo, hi)
    p = distributed.default_params(metric=metric, max_iter=8, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    T, iters, delta, nc = distributed.TargetShardedRigidICP(eng, dist).estimate(p)
    allT = [None] * world
    dist.all_gather_object(allT, T.tolist())
    if rank == 0:
        print("RESULT " + json.dumps({"T": T.tolist(), "iters": iters, "delta": delta, "ncorr": nc, "world": world,
                                       "identical": all(a == allT[0] for a in allT)}))
    dist.destroy_process_group()


class OracleSlabEngine(OracleShardEngine):
    """Test-only counterpart of HipSlabEngine: kd-tree over the rank's slab + halo, its owned source points, the GLOBAL
    means, and the guard's bound restated on the host (cilhip_set_slab_guard / k_solve)."""

    def __init__(self, part, rank, d):
        dst, dst_n, src = part.select(rank, d["dst"], d["dst_n"], d["src"])
        super().__init__(dst, dst_n, src)
        self.part = part
        gdm, self.gsm = part.global_means(d["dst"], d["src"])
        self.dst_mean = gdm
        self.viol = False

    def begin(self, params, T0, gmean):
        super().begin(params, T0, self.gsm)
        self.viol = False
        self.viol_state = None

    def apply_sums(self, sums):
        super().apply_sums(sums)
        pt, ax = self.part, self.part.axis
        dl = self.T[ax, :3].astype(np.float32) - pt.T_part[ax, :3]
        bound = abs(float(self.T[ax, 3] - pt.T_part[ax, 3] + dl @ pt.src_center)) + float(np.abs(dl) @ pt.src_half)
        if not (bound <= pt.slack) and not self.viol:
            self.viol = True
            self.viol_state = self.state()          # (cilhip_get_slab_violation_state: the loop state at the update that raised the flag)

    def violated(self):
        return self.viol

    def violation_state(self):
        return self.viol, (self.viol_state if self.viol else self.state())


def main_slab(metric, n, slack_cells, native=False):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d = syn.make_pair(n, perturb=0.5)
    slack = None if slack_cells < 0 else slack_cells * d["h"]
    plans = []

    def engine_for(T):
        plans.append(distributed.SlabPartition.plan(d["dst"], d["src"], T, float(d["max_sq_dist"]), world, slack=slack))
        return (type("NativeSlabEngine", (NativeLoopMixin, OracleSlabEngine), {}) if native else OracleSlabEngine)(plans[-1], rank, d)

    first = engine_for(np.eye(4, dtype=np.float32))
    if native:
        first.enable_native_allreduce(dist)
    icp = distributed.SlabShardedRigidICP(first, dist, repartition=engine_for)
    p = distributed.default_params(metric=metric, max_iter=12, conv_tol=1e-6, max_sq_dist=float(d["max_sq_dist"]))
    T, iters, delta, nc = icp.estimate(p, check_every=2)
    rep1 = icp.repartitions
    # the SAME object asked again from the identity (what the re-run after loading the tie order does): an engine a re-partition left
    # behind was cut under a later transform and has to be cut again under the start -- same result, never a search outside the halos
    T_again, iters_again, _, nc_again = icp.estimate(p, check_every=2)
    rerun = {"same": bool(np.array_equal(T, T_again) and iters == iters_again and nc == nc_again), "recut": icp.repartitions - rep1,
             "partition_was_stale": bool(rep1 > 0)}
    icp.repartitions = rep1
    counts = [None] * world
    dist.all_gather_object(counts, (len(icp.engine.src), len(icp.engine.dst)))
    allT = [None] * world
    dist.all_gather_object(allT, T.tolist())
    if rank == 0:
        print("RESULT " + json.dumps({"T": T.tolist(), "iters": iters, "delta": delta, "ncorr": nc, "world": world, "repartitions": icp.repartitions,
                                       "n_src": [c[0] for c in counts], "n_dst": [c[1] for c in counts],
                                       "identical": all(a == allT[0] for a in allT), "rerun": rerun}))
    dist.destroy_process_group()


def main_rank_comm():
    """distributed.init_rank_comm's collective logic over gloo with a stand-in context: the id created on rank 0 reaches every rank,
    every rank joins with its own rank / the world size, and a rank that cannot join makes ALL ranks give up (and leave)."""
    from cilantro_amd import icp

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    class FakeCtx:
        def __init__(self, fail):
            self.fail, self.got, self.destroyed = fail, None, 0

        def rank_comm_init(self, uid, nranks, r):
            if self.fail:
                raise RuntimeError("no librccl here")
            self.got = (bytes(np.asarray(uid, np.uint8)), int(nranks), int(r))

        def rank_comm_destroy(self):
            self.destroyed += 1

    made = []
    real = icp.Context.rank_comm_unique_id
    icp.Context.rank_comm_unique_id = staticmethod(lambda: (made.append(1), (np.arange(128) * 7 % 251).astype(np.uint8))[1])
    try:
        ok_ctx = FakeCtx(False)
        r1 = distributed.init_rank_comm(ok_ctx, dist, None, "cpu")
        bad_ctx = FakeCtx(rank == world - 1)
        r2 = distributed.init_rank_comm(bad_ctx, dist, None, "cpu")
        icp.Context.rank_comm_unique_id = staticmethod(lambda: (_ for _ in ()).throw(RuntimeError("rank 0 has no librccl")))
        none_ctx = FakeCtx(False)
        r3 = distributed.init_rank_comm(none_ctx, dist, None, "cpu")
        icp.Context.rank_comm_unique_id = staticmethod(lambda: (np.arange(128) * 7 % 251).astype(np.uint8))

        class PrepCtx(FakeCtx):      # an init that IS a collective (as ncclCommInitRank): a rank entering it alone would hang
            def __init__(self, prep_fail):
                super().__init__(False)
                self.prep_fail, self.entered = prep_fail, 0

            def rank_comm_prepare(self):
                if self.prep_fail:
                    raise RuntimeError("librccl cannot be opened on this rank")

            def rank_comm_init(self, uid, nranks, r):
                self.entered += 1
                dist.barrier()
                super().rank_comm_init(uid, nranks, r)

        prep_bad = PrepCtx(rank == world - 1)
        r4 = distributed.init_rank_comm(prep_bad, dist, None, "cpu")
        prep_ok = PrepCtx(False)
        r5 = distributed.init_rank_comm(prep_ok, dist, None, "cpu")
    finally:
        icp.Context.rank_comm_unique_id = real
    rows = [None] * world
    dist.all_gather_object(rows, {"rank": rank, "r1": r1, "got": [ok_ctx.got[0].hex(), ok_ctx.got[1], ok_ctx.got[2]] if ok_ctx.got else None, "made": len(made),
                                  "r2": r2, "bad_destroyed": bad_ctx.destroyed, "r3": r3, "none_got": none_ctx.got,
                                  "r4": r4, "r4_entered": prep_bad.entered, "r5": r5, "r5_entered": prep_ok.entered})
    if rank == 0:
        print("RESULT " + json.dumps({"world": world, "rows": rows}))
    dist.destroy_process_group()


class OracleKMeansShard:
    """Same interface as cilantro_amd.distributed_models.HipKMeansShard, computed on the CPU by the checker: the oracle's assignment
    (kmeans.hpp:95-119 / :86-94) and numpy for the fixed-point sums."""

    def __init__(self, points, k, index_offset=0):
        self.x = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        self.n, self.k, self.index_offset = len(self.x), int(k), int(index_offset)
        self.lab = np.zeros(self.n, np.int64)

    def maxabs(self):
        return float(np.abs(self.x).max()) if self.n else 0.0

    def assign(self, centroids, scale_exp, use_kd_tree=False):
        sums = np.zeros((self.k, 4), np.int64)
        if self.n == 0:
            return sums, 0
        new, _ = orc.kmeans_assign(self.x, centroids, use_kd_tree=use_kd_tree)
        changed = int((new != self.lab).sum())
        self.lab = new.astype(np.int64)
        fx = np.rint(self.x.astype(np.float64) * np.ldexp(1.0, scale_exp)).astype(np.int64)
        for d in range(3):
            np.add.at(sums[:, d], self.lab, fx[:, d])
        np.add.at(sums[:, 3], self.lab, 1)
        return sums, changed

    def farthest(self, cluster, center):
        m = np.nonzero(self.lab == cluster)[0]
        if len(m) == 0:
            return 0
        c = np.asarray(center, np.float32)
        d = c[None, :] - self.x[m]
        e = d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        keys = (e.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - (m + self.index_offset).astype(np.uint64))
        return int(keys.max())

    def move_point(self, global_index, to_cluster):
        li = int(global_index) - self.index_offset
        self.lab[li] = to_cluster
        return self.x[li].copy()

    def labels(self):
        return self.lab.copy()


def main_kmeans(n, case):
    """ShardedKMeans3f over gloo against the SAME loop over one shard that holds all points (bit for bit: the sums are integers) and
    against the oracle's KMeans (the reference's loop: f32 serial sums)"""
    from cilantro_amd import distributed_models as dm

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(7)
    x = rng.random((n, 3), dtype=np.float32)
    k, iters, tol, kd = 24, 12, 0.0, False
    c0 = x[:k].copy()
    if case == "empty":                 # a far-away initial centroid attracts nothing: the repair (kmeans.hpp:134-176) across ranks
        c0[5] = [50.0, 50.0, 50.0]; c0[11] = [-40.0, 3.0, 2.0]; iters = 4
    elif case == "tol":
        iters, tol = 100, 1e-3
    elif case == "kd":
        kd = True
    # uneven shards, one of them possibly holding the moved point
    cut = [0, n // 3, n] if world == 2 else [round(i * n / world) for i in range(world + 1)]
    lo, hi = cut[rank], cut[rank + 1]
    km = dm.ShardedKMeans3f(OracleKMeansShard(x[lo:hi], k, lo), dist).cluster(c0, max_iter=iters, tol=tol, use_kd_tree=kd)
    rows = [None] * world
    dist.all_gather_object(rows, {"rank": rank, "cent": km.getClusterCentroids().tolist(), "lab": km.getPointToClusterIndexMap().tolist(), "it": km.getNumberOfPerformedIterations()})
    if rank == 0:
        one = dm.ShardedKMeans3f(OracleKMeansShard(x, k, 0), None).cluster(c0, max_iter=iters, tol=tol, use_kd_tree=kd)
        lab = np.concatenate([np.array(r["lab"], np.int64) for r in sorted(rows, key=lambda r: r["rank"])])
        co, lo_, ito = orc.kmeans(x, c0, max_iter=iters, tol=tol, mode=1, use_kd_tree=kd)
        print("RESULT " + json.dumps({"world": world, "same_centroids_on_all_ranks": all(r["cent"] == rows[0]["cent"] for r in rows),
                                       "equals_one_shard": bool(np.array_equal(np.array(rows[0]["cent"], np.float32), one.getClusterCentroids())
                                                                and np.array_equal(lab, one.getPointToClusterIndexMap()) and rows[0]["it"] == one.getNumberOfPerformedIterations()),
                                       "it": rows[0]["it"], "oracle_it": int(ito), "centroid_err_vs_oracle": float(np.abs(np.array(rows[0]["cent"]) - co).max()),
                                       "label_mismatches_vs_oracle": int((lab != lo_).sum())}))
    dist.destroy_process_group()


def main_ransac_counts(n):
    from cilantro_amd import distributed_models as dm

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(9)
    x = (rng.random((n, 3), dtype=np.float32) * 2 - 1)
    x[: n // 2, 2] = 0.25 * x[: n // 2, 0] + 0.1
    nrm = rng.normal(size=(32, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    planes = np.concatenate([nrm, rng.uniform(-0.5, 0.5, (32, 1))], axis=1).astype(np.float32)
    planes[0] = np.array([0.25, 0.0, -1.0, 0.1], np.float32) / np.float32(np.sqrt(1.0625))
    thr = np.float32(0.02)

    def count(shard):
        return lambda pl: np.array([int((np.abs((shard * p[None, :3]).sum(1) + p[3]) <= thr).sum()) for p in pl], np.int64)

    lo, hi = distributed.shard_bounds(n, rank, world)
    tot = dm.sharded_plane_inlier_counts(count(x[lo:hi]), planes, dist)
    if rank == 0:
        print("RESULT " + json.dumps({"world": world, "equal": bool(np.array_equal(tot, count(x)(planes))), "best": int(np.argmax(tot))}))
    dist.destroy_process_group()


def main():
    import signal
    signal.alarm(300)      # a worker never outlives its test (SIGALRM's default action terminates the process)
    metric = int(sys.argv[1]); n = int(sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[3].startswith("kmeans"):
        return main_kmeans(n, sys.argv[3][6:])
    if len(sys.argv) > 3 and sys.argv[3] == "ransac":
        return main_ransac_counts(n)
    if len(sys.argv) > 3 and sys.argv[3] == "rankcomm":
        return main_rank_comm()
    if len(sys.argv) > 3 and sys.argv[3] == "tshard":
        return main_target_sharded(metric, n)
    if len(sys.argv) > 3 and sys.argv[3] == "tshardties":
        return main_target_sharded_ties(metric, n)
    if len(sys.argv) > 3 and sys.argv[3].startswith("nslab"):
        return main_slab(metric, n, float(sys.argv[3][5:] or -1), native=True)
    if len(sys.argv) > 3 and sys.argv[3].startswith("slab"):
        return main_slab(metric, n, float(sys.argv[3][4:] or -1))
    native = len(sys.argv) > 3 and sys.argv[3] == "native"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d = syn.make_pair(n, perturb=0.5)
    lo, hi = distributed.shard_bounds(n, rank, world)
    eng = (type("NativeShardEngine", (NativeLoopMixin, OracleShardEngine), {}) if native else OracleShardEngine)(d["dst"], d["dst_n"], d["src"][lo:hi])
    if native:
        eng.enable_native_allreduce(dist)
    p = distributed.default_params(metric=metric, max_iter=12, conv_tol=1e-6, max_sq_dist=float(d["max_sq_dist"]))
    T, iters, delta, nc = distributed.ShardedRigidICP(eng, dist).estimate(p, check_every=3 if native else 1)
    allT = [None] * world
    dist.all_gather_object(allT, T.tolist())
    if rank == 0:
        print("RESULT " + json.dumps({"T": T.tolist(), "iters": iters, "delta": delta, "ncorr": nc, "world": world,
                                       "identical": all(a == allT[0] for a in allT)}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
