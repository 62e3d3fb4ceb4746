"""Source-sharded rigid ICP over torch.distributed (one process per GPU; backend "nccl" == RCCL on ROCm).

SURVEY.md section 8(e): source points are independent work units and one ICP iteration's only
global quantities are the 16/28/43 accumulated sums, so the path shards with exactly ONE exchange
step per iteration:

    every rank: full target + grid index, a shard of the source
    per iteration:  local  kNN search + accumulation          (HIP, no communication)
                    all-reduce(sum) of CILHIP_SUMS_LEN f64    (RCCL over xGMI; 384 bytes, latency-bound)
                    identical 6x6 / 3x3 solve on every rank   (device epilogue, replicated)

Every rank all-reduces the same values in the same order, so all ranks hold bit-identical
transforms and take the same convergence decision without any extra broadcast.

The protocol is engine-agnostic: ``ShardedRigidICP`` drives any object with the three methods of
:class:`HipShardEngine` (the product engine, backed by the C ABI ``cilhip_icp_begin /
cilhip_icp_partial_sums / cilhip_icp_apply_sums``).  The CPU (gloo, world_size 2) tests plug a
test-only engine in to check the sharding / reduction logic without a GPU.
"""
import ctypes as C

import numpy as np

from . import capi

SUMS_LEN = capi.SUMS_LEN


def shard_bounds(n, rank, world):
    """Contiguous, balanced shard [lo, hi) of n source points for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_rank_comm(ctx, dist, group=None, device="cuda"):
    """This rank's context as a rank of a dedicated RCCL communicator (cilhip_rank_comm_init): rank 0 creates the id, ``dist``
    (torch.distributed, initialised) carries its 128 bytes to every rank.  After it, ``ctx.icp_iterate_ranked(k)`` runs k
    iterations of {partial sums, ncclAllReduce on the context's stream, epilogue} inside the library -- per iteration a handful of
    launches instead of three calls and a framework collective.  Collective; returns whether it worked on EVERY rank (a rank whose
    librccl cannot be opened makes all ranks keep the framework's all-reduce: both protocols give the same sums)."""
    import torch

    from .icp import Context

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok = torch.ones(1, dtype=torch.int32, device=device)
    uid = torch.zeros(128, dtype=torch.uint8, device=device)
    # (1) what can fail on ONE rank alone -- opening librccl, this rank's buffer; rank 0: creating the id -- happens BEFORE the
    #     collective init, and the ranks agree on the outcome: a rank that bailed out early would leave its peers waiting inside
    #     ncclCommInitRank (a deadlock, not a fallback)
    try:
        prep = getattr(ctx, "rank_comm_prepare", None)
        if prep is not None:
            prep()
        if rank == 0:
            uid.copy_(torch.from_numpy(Context.rank_comm_unique_id()).to(device))
    except Exception:
        ok.zero_()
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        return False
    if world > 1:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(uid, src=src, group=group)
    # (2) the collective init: every rank enters it
    try:
        ctx.rank_comm_init(uid.cpu().numpy(), world, rank)
    except Exception:
        ok.zero_()
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        ctx.rank_comm_destroy()
        return False
    return True


# Option "tie_rule" in sharded runs.  Which of several exactly equidistant nearest points the reference names is a property of the
# tree it builds over the WHOLE target (csrc/tie_build.hip builds its order tables on the device).  The engines' kernels resolve ties from that tree's order tables when
# they are loaded and COUNT the tied queries when they are not (tie_rule 2); the loops below look at the count after a run --
# one MAX over the ranks, so that all take the same decision --, have every engine load the tables (built once per target and
# process: _TIE_ORDERS) and run again.  Engines that know nothing of this (the CPU test engine) are never asked.
_TIE_ORDERS = {}      # content key of a target cloud -> order handle; at most _TIE_ORDERS_MAX of them (least recently used goes, with its handle)
_TIE_ORDERS_MAX = 2


def _tie_order_key(dst):
    """a key of the cloud's CONTENT (a buffer reused in place for another cloud of the same size must not find the old tables)"""
    try:
        import xxhash

        return (len(dst), xxhash.xxh64(memoryview(dst).cast("B")).intdigest())
    except ImportError:
        import zlib

        return (len(dst), zlib.crc32(memoryview(dst).cast("B")))


def _tie_order_known(dst):
    dst = np.ascontiguousarray(np.asarray(dst, np.float32).reshape(-1, 3))
    return _tie_order_key(dst) in _TIE_ORDERS


def release_tie_orders():
    """free every cached order handle (cilhip_tie_order_destroy)"""
    L = capi.load()
    for h in _TIE_ORDERS.values():
        L.cilhip_tie_order_destroy(h)
    _TIE_ORDERS.clear()


def _tie_order_of(dst):
    """handle of the order tables of the whole target cloud `dst` (host array), built on first use (on the current HIP device)"""
    import ctypes as C

    dst = np.ascontiguousarray(np.asarray(dst, np.float32).reshape(-1, 3))
    key = _tie_order_key(dst)
    if key in _TIE_ORDERS:
        _TIE_ORDERS[key] = _TIE_ORDERS.pop(key)      # (most recently used last)
        return _TIE_ORDERS[key]
    h = C.c_void_p()
    rc = capi.load().cilhip_tie_order_create(dst.ctypes.data, len(dst), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"cilhip_tie_order_create failed ({rc})")
    while len(_TIE_ORDERS) >= _TIE_ORDERS_MAX:      # (the handle keeps its own copy of the tables: nothing refers to the evicted cloud)
        capi.load().cilhip_tie_order_destroy(_TIE_ORDERS.pop(next(iter(_TIE_ORDERS))))
    _TIE_ORDERS[key] = h
    return h


def _ties_pending_anywhere(engine, dist, group, seen_before=False):
    fn = getattr(engine, "ties_pending", None)
    if fn is None:
        return False
    mine = 1 if (seen_before or fn()) else 0
    if dist is not None and dist.get_world_size(group) > 1:
        import torch

        dev = engine.sums.device if hasattr(engine, "sums") else "cpu"
        t = torch.tensor([mine], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        mine = int(t.item())
    return mine != 0


class HipShardEngine:
    """Per-rank engine backed by libcilantro_hip.so.  All work is enqueued on torch's current stream
    so the RCCL all-reduce is ordered after the partial-sum kernels without host synchronisation."""

    def __init__(self, dst_points, dst_normals, src_shard, device):
        import torch

        from .icp import Context

        self.torch = torch
        self.ctx = Context(device, torch.cuda.current_stream().cuda_stream)
        self.ctx.set_target(dst_points, dst_normals)
        self.ctx.set_source(src_shard)
        self.n_local = self.ctx.n_source
        self.sums = torch.zeros(SUMS_LEN, dtype=torch.float64, device=f"cuda:{device}")

    def local_source_sum(self):
        """sum of the local source points (f64, 3) and their count -> for the global src_mean_."""
        _, sm = self.ctx.means()
        return sm.astype(np.float64) * self.n_local, self.n_local

    def begin(self, params, T0, global_src_mean):
        self.ctx.icp_begin(params, T0, global_src_mean)

    def partial_sums(self):
        """Enqueue search + accumulation; returns the tensor the caller all-reduces in place."""
        self.ctx.icp_partial_sums(self.sums.data_ptr())
        return self.sums

    def apply_sums(self, sums):
        self.ctx.icp_apply_sums(sums.data_ptr())

    def enable_native_allreduce(self, dist, group=None):
        """Make this rank's context a rank of its own RCCL communicator (init_rank_comm below), so that the loops can run a block
        of iterations inside ONE library call.  Collective.  Returns whether it worked on EVERY rank."""
        self.native = init_rank_comm(self.ctx, dist, group, self.sums.device)
        return self.native

    def iterate(self, k):
        """k iterations of the inner triple inside the library (needs enable_native_allreduce)"""
        self.ctx.icp_iterate_ranked(k)

    def state(self):
        r = self.ctx.icp_state()
        T = np.array(r.T[:], np.float32).reshape(4, 4).T.copy()
        return T, int(r.iterations), float(r.last_delta_norm), int(r.last_ncorr)

    def ties_pending(self):
        """this rank's searches since begin() met exactly equidistant nearest points and had no order tables (option tie_rule 2)"""
        info = self.ctx.tie_order_info()
        return (not info["loaded"]) and info["pending"] != 0

    def load_tie_order(self):
        self.ctx.build_tie_order()      # (this rank holds the whole target)


class ShardedRigidICP:
    """IterativeClosestPointBase::estimate() (registration/icp_base.hpp:68-87) across ranks.

    ``dist`` is ``torch.distributed`` (already initialised) or None for a single process.
    """

    def __init__(self, engine, dist=None, group=None):
        self.engine = engine
        self.dist = dist
        self.group = group

    def _allreduce(self, t):
        if self.dist is not None and self.dist.get_world_size(self.group) > 1:
            self.dist.all_reduce(t, group=self.group)
        return t

    def global_source_mean(self):
        import torch

        s, n = self.engine.local_source_sum()
        dev = self.engine.sums.device if hasattr(self.engine, "sums") else "cpu"
        t = torch.tensor([s[0], s[1], s[2], float(n)], dtype=torch.float64, device=dev)
        self._allreduce(t)
        t = t.cpu().numpy()
        return (t[:3] / max(t[3], 1.0)).astype(np.float32)

    def estimate(self, params, T0=None, check_every=0):
        out = self._estimate_once(params, T0, check_every)
        if _ties_pending_anywhere(self.engine, self.dist, self.group):      # (collective: every rank runs again or none)
            self.engine.load_tie_order()
            out = self._estimate_once(params, T0, check_every)
        return out

    def _estimate_once(self, params, T0=None, check_every=0):
        """Runs up to params.max_iter iterations.  check_every=0: enqueue everything and read the
        state once at the end (the device-side `done` flag turns the remaining launches into
        no-ops once converged); check_every=k: poll the state every k iterations and stop early."""
        T0 = np.eye(4, dtype=np.float32) if T0 is None else np.asarray(T0, np.float32)
        gmean = self.global_source_mean()
        self.engine.begin(params, T0, gmean)
        if getattr(self.engine, "native", False):
            # blocks of iterations inside the library, its own all-reduce on the engine's stream
            done, total = 0, int(params.max_iter)
            while done < total:
                k = min(check_every, total - done) if check_every else total
                self.engine.iterate(k)
                done += k
                if check_every and done < total:
                    T, iters, delta, nc = self.engine.state()
                    if delta < params.conv_tol:
                        break
            return self.engine.state()
        for it in range(int(params.max_iter)):
            sums = self.engine.partial_sums()
            self._allreduce(sums)
            self.engine.apply_sums(sums)
            if check_every and (it + 1) % check_every == 0:
                T, iters, delta, nc = self.engine.state()
                if delta < params.conv_tol:
                    break
        return self.engine.state()


def default_params(metric=capi.METRIC_COMBINED, **kw):
    p = capi.IcpParams()
    p.metric = metric
    p.w_p2p, p.w_p2pl = 0.0, 1.0
    p.max_iter, p.conv_tol = 15, 1e-5
    p.max_opt_iter, p.opt_conv_tol = 1, 1e-5
    p.max_sq_dist = 0.01 * 0.01
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ---- target-sharded runs (BASELINE configs[3]; SURVEY.md 8(e) partitioning A) ---------------------------
KEY_NONE = 0x7FFFFFFFFFFFFFFF  # packed (d2, global index) key meaning "no neighbour in this shard"


class HipTargetShardEngine:
    """Per-rank engine for a target cloud sharded by index over the ranks: this rank indexes
    dst[lo:hi) only, holds ALL source points, and tags its matches with GLOBAL target indices.

    whole_target (optional, HOST array of the whole cloud's points -- host memory holds what one device does not): lets the loop follow
    the reference's order among exactly equidistant nearest points, inside a shard and across shards (c_api.h: cilhip_icp_order_keys);
    without it such ties go to the lowest global index."""

    def __init__(self, dst_shard, dst_normals_shard, src_points, index_offset, global_dst_mean, device, whole_target=None):
        import torch

        from .icp import Context

        self.torch = torch
        self.ctx = Context(device, torch.cuda.current_stream().cuda_stream)
        self.ctx.set_target(dst_shard, dst_normals_shard)
        self.ctx.set_source(src_points)
        self.ctx.set_shard_info(index_offset, dst_mean=global_dst_mean)
        dev = f"cuda:{device}"
        self.keys = torch.full((max(self.ctx.n_source, 1),), KEY_NONE, dtype=torch.int64, device=dev)
        self.okeys = None
        self.sums = torch.zeros(SUMS_LEN, dtype=torch.float64, device=dev)
        self._whole, self._offset, self._n_local = whole_target, int(index_offset), len(np.asarray(dst_shard).reshape(-1, 3))
        self.ordered = False

    def begin(self, params, T0):
        self.ctx.icp_begin(params, T0, None)

    def partial_keys(self):
        self.ctx.icp_partial_keys(self.keys.data_ptr())
        return self.keys

    def sums_from_keys(self, keys):
        self.ctx.icp_sums_from_keys(keys.data_ptr(), self.sums.data_ptr())
        return self.sums

    # -- the reference's tie order (two keys per iteration once some search has met a tie)
    def ties_pending(self):
        if self._whole is None or self.ordered:
            return False
        return self.ctx.tie_order_info()["pending"] > 0

    def load_tie_order(self):
        if self._whole is None:
            return
        self.ctx.load_tie_order(_tie_order_of(self._whole), np.arange(self._offset, self._offset + self._n_local, dtype=np.uint32) if self._n_local else None)
        self.okeys = self.torch.full_like(self.keys, KEY_NONE)
        self.ordered = True

    def order_keys(self, keys):
        self.ctx.icp_order_keys(keys.data_ptr(), self.okeys.data_ptr())
        return self.okeys

    def sums_from_ordered_keys(self, keys, okeys):
        self.ctx.icp_sums_from_ordered_keys(keys.data_ptr(), okeys.data_ptr(), self.sums.data_ptr())
        return self.sums

    def apply_sums(self, sums):
        self.ctx.icp_apply_sums(sums.data_ptr())

    def state(self):
        r = self.ctx.icp_state()
        T = np.array(r.T[:], np.float32).reshape(4, 4).T.copy()
        return T, int(r.iterations), float(r.last_delta_norm), int(r.last_ncorr)


class TargetShardedRigidICP:
    """ICP with the TARGET sharded over the ranks: two collectives per iteration --
    all-reduce(MIN) of one packed int64 key per source point (the global nearest neighbour; ring cost
    2(G-1)/G x 8 B x Ns per GPU over xGMI), then all-reduce(SUM) of the 48 accumulated f64.  When some rank's searches meet exactly
    equidistant nearest points (inside its shard or against another rank's) and the engines can load the whole target's tie order, the
    run is repeated with a third collective per iteration: the MIN of the traversal keys (cilhip_icp_order_keys) -- the reference's
    matches, index for index."""

    def __init__(self, engine, dist=None, group=None):
        self.engine, self.dist, self.group = engine, dist, group

    def _multi(self):
        return self.dist is not None and self.dist.get_world_size(self.group) > 1

    def estimate(self, params, T0=None):
        out = self._estimate_once(params, T0)
        if not getattr(self.engine, "ordered", True) and _ties_pending_anywhere(self.engine, self.dist, self.group):
            self.engine.load_tie_order()
            out = self._estimate_once(params, T0)
        return out

    def _estimate_once(self, params, T0=None):
        T0 = np.eye(4, dtype=np.float32) if T0 is None else np.asarray(T0, np.float32)
        self.engine.begin(params, T0)
        ordered = getattr(self.engine, "ordered", False)
        for _ in range(int(params.max_iter)):
            keys = self.engine.partial_keys()
            if self._multi():
                self.dist.all_reduce(keys, op=self.dist.ReduceOp.MIN, group=self.group)
            if ordered:
                okeys = self.engine.order_keys(keys)
                if self._multi():
                    self.dist.all_reduce(okeys, op=self.dist.ReduceOp.MIN, group=self.group)
                sums = self.engine.sums_from_ordered_keys(keys, okeys)
            else:
                sums = self.engine.sums_from_keys(keys)
            if self._multi():
                self.dist.all_reduce(sums, group=self.group)
            self.engine.apply_sums(sums)
        return self.engine.state()


# ---- spatially sharded runs (SURVEY.md 8(e) partitioning B) --------------------------------------------------------
class SlabPartition:
    """Slabs along the longest axis of the target's bounding box.

    Rank r owns the source points whose image under ``T_part`` has its axis coordinate in [b_r, b_{r+1}) and holds the
    target points with coordinate in [b_r - halo, b_{r+1} + halo), halo = sqrt(max_sq_dist) + slack: the nearest neighbour
    within the search radius of every owned query is then among the rank's own target points, for as long as no source
    point has moved by more than ``slack`` along the axis since the partition (the device-side guard,
    ``cilhip_set_slab_guard``).  Boundaries are quantiles of the source coordinates: the queries are the work.
    No data-path collective besides the 48-double all-reduce of the partial sums.
    """

    def __init__(self, axis, bounds, halo, slack, T_part, src_center, src_half):
        self.axis, self.bounds, self.halo, self.slack = int(axis), np.asarray(bounds, np.float64), float(halo), float(slack)
        self.T_part = np.asarray(T_part, np.float32).reshape(4, 4).copy()
        self.src_center, self.src_half = np.asarray(src_center, np.float32), np.asarray(src_half, np.float32)

    @staticmethod
    def _image_coord(T, pts, axis):
        T = np.asarray(T, np.float64)
        return pts.astype(np.float64) @ T[axis, :3] + T[axis, 3]

    @classmethod
    def plan(cls, dst, src, T_part, max_sq_dist, world, slack=None):
        dst = np.asarray(dst, np.float32).reshape(-1, 3); src = np.asarray(src, np.float32).reshape(-1, 3)
        if len(dst) == 0:      # an empty target: nothing to cut -- one unbounded slab per rank boundary, no correspondences anywhere
            lo = hi = np.zeros(3)
        else:
            lo, hi = dst.min(axis=0).astype(np.float64), dst.max(axis=0).astype(np.float64)
        axis = int(np.argmax(hi - lo))
        r = float(np.sqrt(max_sq_dist)) if np.isfinite(max_sq_dist) else float((hi - lo)[axis])
        slack = 2.0 * r if slack is None else float(slack)
        q = cls._image_coord(T_part, src, axis) if len(src) else np.zeros(0)
        qs = np.quantile(q, np.linspace(0.0, 1.0, world + 1)[1:-1]) if len(q) and world > 1 else np.zeros(0)
        bounds = np.concatenate([[-np.inf], qs, [np.inf]])
        slo, shi = (src.min(axis=0), src.max(axis=0)) if len(src) else (np.zeros(3), np.zeros(3))
        return cls(axis, bounds, r + slack, slack, T_part, 0.5 * (slo + shi), 0.5 * (shi - slo))

    def select(self, rank, dst, dst_n, src):
        """-> (target points of the slab + halo, their normals or None, owned source points) for ``rank``; also keeps the
        index arrays (``self.dst_index``, ``self.src_index``) of the last call for callers that map results back."""
        dst = np.asarray(dst, np.float32).reshape(-1, 3); src = np.asarray(src, np.float32).reshape(-1, 3)
        b0, b1 = self.bounds[rank], self.bounds[rank + 1]
        x = dst[:, self.axis].astype(np.float64)
        dm = (x >= b0 - self.halo) & (x < b1 + self.halo)
        q = self._image_coord(self.T_part, src, self.axis)
        sm = (q >= b0) & (q < b1)
        self.dst_index, self.src_index = np.nonzero(dm)[0], np.nonzero(sm)[0]
        return (np.ascontiguousarray(dst[dm]), None if dst_n is None else np.ascontiguousarray(np.asarray(dst_n, np.float32).reshape(-1, 3)[dm]),
                np.ascontiguousarray(src[sm]))

    @staticmethod
    def global_means(dst, src):
        """dst_mean_ / src_mean_ of the WHOLE clouds as the ICP classes hold them (f64 sums rounded to f32); ranks that
        only hold their part obtain the same values from an all-reduce of (sum, count)."""
        return (np.asarray(dst, np.float64).reshape(-1, 3).mean(axis=0).astype(np.float32),
                np.asarray(src, np.float64).reshape(-1, 3).mean(axis=0).astype(np.float32))

    def arm_guard(self, ctx, T_part=None):
        ctx.set_slab_guard(self.axis, self.slack, self.src_center, self.src_half, self.T_part if T_part is None else T_part)

    def describe(self, rank, n_dst_local, n_src_local):
        return {"axis": self.axis, "halo": self.halo, "slack": self.slack, "rank0_slab": [float(self.bounds[rank]), float(self.bounds[rank + 1])],
                "n_target_local": int(n_dst_local), "n_source_local": int(n_src_local)}


class HipSlabEngine(HipShardEngine):
    """Per-rank engine of a slab-partitioned run: HipShardEngine over this rank's part, with the GLOBAL means and the guard."""

    def __init__(self, part, rank, dst, dst_n, src, device):
        d, n, s = part.select(rank, dst, dst_n, src)
        super().__init__(d, n, s, device)
        self.part = part
        self.gdm, self.gsm = part.global_means(dst, src)
        self.ctx.set_shard_info(0, dst_mean=self.gdm)
        part.arm_guard(self.ctx)
        self._dst_all = np.ascontiguousarray(np.asarray(dst, np.float32).reshape(-1, 3))
        self._dst_index = np.ascontiguousarray(part.dst_index, np.uint32)
        if _tie_order_known(self._dst_all):      # (a re-partitioned engine of a target whose order is known: loaded at once)
            self.load_tie_order()

    def load_tie_order(self):
        # the order of the WHOLE target; this rank's slab takes the entries of its own points
        if len(self._dst_index):
            self.ctx.load_tie_order(_tie_order_of(self._dst_all), self._dst_index)

    def begin(self, params, T0, global_src_mean):
        self.ctx.icp_begin(params, T0, self.gsm)

    def violated(self):
        return self.ctx.slab_violation()

    def violation_state(self):
        """-> (violated, (T, iterations, delta, ncorr) right after the update that raised the guard)"""
        bad, r = self.ctx.slab_violation_state()
        return bad, (np.array(r.T[:], np.float32).reshape(4, 4).T.copy(), int(r.iterations), float(r.last_delta_norm), int(r.last_ncorr))


class SlabShardedRigidICP:
    """ICP over a slab partition: ShardedRigidICP's loop (one all-reduce of the 48 partial sums per iteration) plus the
    guard: every ``check_every`` iterations the (identical on all ranks) violation flag is read; if a source point may
    have left its halo, ``repartition(T)`` -- a caller-supplied function that returns a new engine partitioned under T --
    is called with the last checked transform and the run continues from there."""

    def __init__(self, engine, dist=None, group=None, repartition=None):
        self.engine, self.dist, self.group, self.repartition = engine, dist, group, repartition
        self.repartitions = 0

    def estimate(self, params, T0=None, check_every=5):
        self._ties_seen = False      # (an engine replaced by a re-partition takes its counters with it: looked at before it goes)
        out = self._estimate_once(params, T0, check_every)
        if _ties_pending_anywhere(self.engine, self.dist, self.group, self._ties_seen):      # (collective; engines made by later re-partitions load the tables themselves)
            self.engine.load_tie_order()
            out = self._estimate_once(params, T0, check_every)
        return out

    def _swap_engine(self, T):
        """replace the engine by one partitioned under ``T`` (collective: every rank calls it at the same iteration)"""
        was_native = getattr(self.engine, "native", False)
        tp = getattr(self.engine, "ties_pending", None)
        self._ties_seen = getattr(self, "_ties_seen", False) or (tp is not None and tp())
        self.engine = self.repartition(T)
        if was_native and self.dist is not None and hasattr(self.engine, "enable_native_allreduce"):
            self.engine.enable_native_allreduce(self.dist, self.group)      # (a new context: a new communicator; collective like the re-partition itself)
        self.repartitions += 1

    def _estimate_once(self, params, T0=None, check_every=5):
        T_ck = np.eye(4, dtype=np.float32) if T0 is None else np.asarray(T0, np.float32).copy()
        total, base, since, begin_base = int(params.max_iter), 0, 0, 0   # base: iterations up to the last checked state; begin_base: up to the last begin()
        # The slabs (and the guard's reference transform) of the engine at hand may have been cut under ANOTHER transform: an engine
        # left by a re-partition of an earlier run -- the re-run after the tie order was loaded starts from T0 again.  The guard only
        # looks at transforms AFTER an update, so a first search under T_ck against slabs cut under something else would be accepted
        # unchecked: cut again under T_ck first (the C loop does the same: cilhip_multi_icp_run re-uploads when T_part != T0).
        part = getattr(self.engine, "part", None)
        Tp = None if part is None else np.asarray(part.T_part, np.float32).reshape(4, 4)
        if Tp is not None and not np.array_equal(Tp, T_ck.reshape(4, 4)) and self.repartition is not None:
            self._swap_engine(T_ck)
            Tp = T_ck.reshape(4, 4)
        inner = ShardedRigidICP(self.engine, self.dist, self.group)
        self.engine.begin(params, T_ck, None)
        fresh = Tp is None or np.array_equal(Tp, T_ck.reshape(4, 4))      # the engine's partition was made under exactly T_ck
        every = max(check_every, 1)
        while base + since < total:
            if getattr(self.engine, "native", False):
                # up to the next check inside the library (its own all-reduce on the engine's stream)
                k = min(every - since % every, total - base - since)
                self.engine.iterate(k)
                since += k
            else:
                sums = self.engine.partial_sums()
                inner._allreduce(sums)
                self.engine.apply_sums(sums)
                since += 1
            if since % every == 0 or base + since == total:
                T, iters, delta, nc = self.engine.state()
                # (the same answers on every rank: same transform, same global box)
                vs = getattr(self.engine, "violation_state", None)
                if vs is not None:
                    bad, (Tv, iv, dv, ncv) = vs()
                else:
                    bad, Tv, iv, dv, ncv = self.engine.violated(), None, 0, 0.0, 0
                # The flag is about the NEXT search: the update that raised it is still exact (its search ran inside the
                # halos).  An engine that reports the loop state at that update lets every iteration up to and including it
                # be kept; otherwise only a window whose single iteration ran under the partition's own transform is (that
                # alone already guarantees progress when every update trips the guard).
                if bad and Tv is not None and iv > 0:
                    T_ck, base, since, fresh = Tv, begin_base + iv, 0, False
                    if dv < params.conv_tol or base >= total:
                        return Tv, base, dv, ncv
                elif not bad or (fresh and since == 1):
                    T_ck, base, since, fresh = T, base + since, 0, False
                    if delta < params.conv_tol or base >= total:
                        return T, begin_base + iters, delta, nc
                    every = max(check_every, 1) if not bad else 1
                if bad:
                    if self.repartition is None:
                        raise RuntimeError("a source point may have left its slab's halo and no repartition function was given")
                    self._swap_engine(T_ck)
                    inner = ShardedRigidICP(self.engine, self.dist, self.group)
                    self.engine.begin(params, T_ck, None)
                    since, begin_base, fresh, every = 0, base, True, 1
        T, iters, delta, nc = self.engine.state()
        return T, begin_base + iters, delta, nc
