"""Option "tie_rule" (default 2): correspondences equal the reference's where several target points are EXACTLY equidistant too.

A brute-force argmin names the lowest target index among tied candidates, nanoflann the one its traversal meets first
(core/kd_tree.hpp:82-90) -- on the reference's raw sensor frames 660 of 120k queries are tied under the identity and 321 of them
get a different (equally near) point.  The engine resolves ties on the device, in every kernel form, from the order tables of the
reference's tree (csrc/tie_build.hip; its host restatement tests/cpp/tie_order_host.hpp is pinned on the CPU by tests/test_tie_order_cpu.py; csrc/search_device.hpp tie_settle): with DEFAULT
options every index equals nanoflann's and the loop equals the oracle's loop over the reference's searches.  The tables are built
when a search first meets a tie (tie_rule 2) or up front (1); 0 = the lowest index.
"""
import ctypes as C
import os

import numpy as np
import pytest

from cilantro_amd import capi
from cilantro_amd import synthetic as syn

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def Context(hip_lib):
    from cilantro_amd.icp import Context as Ctx

    return Ctx


def _report(name, obj):
    import json

    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass
    print(name, obj)


def _signed(idx):
    gi = idx.astype(np.int64)
    gi[idx == capi.NONE_IDX] = -1
    return gi


def _ref_matches(tree, q, r2, n):
    o1, o2, ov = tree.find_correspondences(q, float(r2))
    oi = np.full(n, -1, np.int64)
    oi[o2] = o1
    return oi, o2, ov


def _clouds():
    f = np.load(os.path.join(HERE, "golden", "frames_full.npz"))
    p1, n1, p2 = f["p1"], f["n1"], f["p2"]
    keep = p1[:, 0] > -0.4
    yield "frame_1 vs frame_2", np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep]), np.ascontiguousarray(p2), np.float32(0.02 * 0.02)
    rng = np.random.default_rng(3)
    b = syn.make_pair(300_000, perturb=0.3)
    dup = rng.choice(len(b["dst"]), 9000, replace=False)
    D = np.ascontiguousarray(np.concatenate([b["dst"], b["dst"][dup], b["dst"][dup[:1500]]]))      # doubled and tripled points
    N = np.ascontiguousarray(np.concatenate([b["dst_n"], b["dst_n"][dup], b["dst_n"][dup[:1500]]]))
    yield "duplicated target points", D, N, b["src"], np.float32(b["max_sq_dist"])


def _icp_params(ctx, r2, iters=6):
    p = capi.IcpParams()
    ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), iters, 0.0
    return p


def test_default_options_name_nanoflann_s_points(Context, orc):
    report = {}
    I = np.eye(4, dtype=np.float32)
    for name, D, N, S, r2 in _clouds():
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        oi, _, ov = _ref_matches(tree, S, r2, len(S))
        # (0) the lowest-index rule, for the record: differs from the reference on part of the tied queries
        c0 = Context(); c0.set_option("tie_rule", 0); c0.set_target(D, N); c0.set_source(S)
        c0.find_correspondences(I, float(r2), count=False)
        lo, _ = c0.get_nn()
        differ_lowest = int(np.count_nonzero(_signed(lo) != oi))
        tied = c0.tie_count(I, float(r2))
        res0 = c0.icp_run(_icp_params(c0, r2))
        T0 = np.array(res0.T[:], np.float32).reshape(4, 4).T
        assert c0.tie_order_info()["builds"] == 0
        c0.close()
        assert differ_lowest > 0 and tied >= differ_lowest, name
        # (1) DEFAULT options, a single search: the first one meets ties, the tables are built, the search runs again
        ctx = Context(); ctx.set_target(D, N); ctx.set_source(S)
        assert not ctx.tie_order_info()["loaded"]
        nfound = ctx.find_correspondences(I, float(r2))
        info = ctx.tie_order_info()
        assert info["loaded"] and info["builds"] == 1 and info["pending"] == 0, info
        gi, gd = ctx.get_nn()
        gi = _signed(gi)
        seen, moved = ctx.tie_rule_stats()
        assert np.array_equal(gi, oi), (name, int(np.count_nonzero(gi != oi)))
        assert nfound == int(np.count_nonzero(oi >= 0))
        assert np.array_equal(gd[gi >= 0].view(np.uint32), ov.view(np.uint32)), name      # (ascending source order on both sides)
        assert seen == tied and moved == differ_lowest, (name, seen, tied, moved, differ_lowest)
        # the engine-level list too (ascending source index, correspondence_search_kd_tree.hpp:231)
        i1, i2, v = ctx.get_correspondences()
        assert np.array_equal(i1.astype(np.int64), oi[oi >= 0]) and np.array_equal(i2.astype(np.int64), np.nonzero(oi >= 0)[0])
        # (2) the loop, default options (adaptive forms): six iterations against the oracle's loop over the reference's searches
        po = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
        ro = orc.icp_run(D, N, S, po)
        p = _icp_params(ctx, r2)
        res = ctx.icp_run(p)
        Tg = np.array(res.T[:], np.float32).reshape(4, 4).T
        err = float(np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)))
        seen_l, moved_l = ctx.tie_rule_stats()
        assert err <= 1e-5 and int(res.iterations) == 6, (name, err)
        assert ctx.tie_order_info()["builds"] == 1      # (once per target)
        one, two = ctx.last_run_forms()
        # the set the loop leaves behind: the reference's at the last iteration's transform, index for index
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi_l, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        assert np.array_equal(_signed(li), oi_l), (name, int(np.count_nonzero(_signed(li) != oi_l)))
        assert int(res.last_ncorr) == int(np.count_nonzero(oi_l >= 0))
        ctx.close()
        # (3) a fresh context whose FIRST call is the loop: the run that meets the ties is executed again, same result bit for bit
        c2 = Context(); c2.set_target(D, N); c2.set_source(S)
        res2 = c2.icp_run(p)
        assert c2.tie_order_info()["builds"] == 1
        assert np.array_equal(np.array(res2.T[:], np.float32), np.array(res.T[:], np.float32)), name
        c2.close()
        # (4) tie_rule 1: tables up front, same result
        c1 = Context(); c1.set_option("tie_rule", 1); c1.set_target(D, N); c1.set_source(S)
        res1 = c1.icp_run(p)
        assert np.array_equal(np.array(res1.T[:], np.float32), np.array(res.T[:], np.float32)), name
        info1 = c1.tie_order_info()
        c1.close()
        report[name] = {"queries": int(len(S)), "tied_queries_identity": int(seen), "not_the_lowest_index_identity": int(moved),
                        "tied_queries_over_6_iterations": int(seen_l), "not_the_lowest_index_over_6_iterations": int(moved_l),
                        "forms_one_pass_two_pass": [int(one), int(two)],
                        "order_tables_build_ms": info["build_ms"], "order_tables_build_ms_up_front": info1["build_ms"],
                        "T_minus_oracle_nanoflann_order_default_options": err,
                        "T_minus_oracle_nanoflann_order_tie_rule_0": float(np.linalg.norm(T0.astype(np.float64) - ro["T"].astype(np.float64)))}
    _report("tie_rule.json", report)


def test_every_kernel_form_resolves_ties(Context, orc):
    """the duplicated-points cloud through each form the loop can take (tiles one pass / two passes, per lane, warm-started from either
    start): the last iteration's matches equal nanoflann's index for index, the loops agree with the oracle's"""
    name, D, N, S, r2 = list(_clouds())[1]
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    po = orc.make_params(metric=1, max_iter=8, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
    ro = orc.icp_run(D, N, S, po)
    forms = {"adaptive": {}, "tiles one pass": {"tiled": 2, "tile_accumulation": 2, "warm_start": 0},
             "tiles two passes": {"tiled": 2, "tile_accumulation": 0, "warm_start": 0}, "per lane": {"tiled": 0, "warm_start": 0, "group_search": 0}, "16 lanes per query": {"tiled": 0, "warm_start": 0, "group_search": 16},
             "lanes chosen by the loop": {"tiled": 0, "warm_start": 0},
             "per lane fused": {"tiled": 0, "warm_start": 0, "fused": 1},
             "warm forced, tiled start": {"tiled": 2, "warm_start": 2}, "warm forced, per-lane start": {"tiled": 0, "warm_start": 2}}
    out = {}
    for fname, opts in forms.items():
        ctx = Context()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_target(D, N); ctx.set_source(S)
        res = ctx.icp_run(_icp_params(ctx, r2, 8))
        Tg = np.array(res.T[:], np.float32).reshape(4, 4).T
        err = float(np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)))
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi_l, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        bad = int(np.count_nonzero(_signed(li) != oi_l))
        seen, moved = ctx.tie_rule_stats()
        out[fname] = {"T_minus_oracle": err, "index_mismatches": bad, "warm_iterations": ctx.last_warm_iterations(), "tied": int(seen), "moved": int(moved)}
        ctx.close()
        assert bad == 0 and err <= 1e-5, (fname, out[fname])
        assert seen > 0 and moved > 0, (fname, out[fname])
    _report("tie_rule_forms.json", out)


def test_tie_rule_1_refuses_what_the_order_does_not_cover(Context):
    """the explicit request covers every search over POINTS (all directions: the reverse matches' tables are built from the first
    search on) and the forward search of the 6-D / 9-D feature adaptors (the DIM = 6 / 9 tree over the target's features); the reverse
    feature searches walk a tree over the transformed SOURCE features, which is not restated: refused, not silently approximated"""
    d = syn.make_pair(20_000, perturb=0.3)
    I = np.eye(4, dtype=np.float32)
    ctx = Context(); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"], np.ascontiguousarray(d["dst_n"][: len(d["src"])]))
    ctx.set_option("tie_rule", 1)
    for direction in (1, 2):
        ctx.set_option("search_direction", direction)
        ctx.find_correspondences(I, float(d["max_sq_dist"]))
    ctx.set_option("search_direction", 0)
    ctx.set_option("feature_normal_weight", 0.1)
    ctx.find_correspondences(I, float(d["max_sq_dist"]))          # (round 6: the forward feature search follows the DIM = 6 tree; its tables are built up front)
    assert ctx.tie_order_info()["loaded"]
    for direction in (1, 2):                                        # the reverse feature searches walk a tree over the TRANSFORMED source features: not restated
        ctx.set_option("search_direction", direction)
        with pytest.raises(RuntimeError):
            ctx.find_correspondences(I, float(d["max_sq_dist"]))
    # the default applies the order where it is defined and runs
    ctx.set_option("tie_rule", 2)
    ctx.find_correspondences(I, float(d["max_sq_dist"]))
    ctx.close()


def test_sharded_runs_resolve_ties_from_the_whole_target_s_order(Context, orc):
    """cilhip_multi_icp_run, three shards on one GPU, source shards and spatial slabs (a slab holds only PART of the target: the order
    tables are the WHOLE cloud's, handed to each shard through the global indices of its points) on the duplicated-points cloud,
    default options: the loop equals the oracle's over the reference's searches; also when the slab guard fires and the shards are
    cut again (the tables are loaded again with the new cut)."""
    from cilantro_amd.multi import PARTITION_SLABS, PARTITION_SOURCE_SHARDS, MultiDeviceRigidICP

    name, D, N, S, r2 = list(_clouds())[1]
    po = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
    ro = orc.icp_run(D, N, S, po)
    c = Context(); p = _icp_params(c, r2); c.close()
    out = {}
    for label, part, slack in (("source shards", PARTITION_SOURCE_SHARDS, None), ("slabs", PARTITION_SLABS, None), ("slabs, guard fires", PARTITION_SLABS, 0.002),
                               ("index shards of the target", 2, None)):
        m = MultiDeviceRigidICP([0, 0, 0])
        if slack is not None:
            m.set_slab_slack(slack)
        m.set_clouds(D, N, S, r2, part)
        rr = m.icp_run(p, check_every=2 if slack is not None else 0)
        T = np.array(rr.T[:], np.float32).reshape(4, 4).T
        err = float(np.linalg.norm(T.astype(np.float64) - ro["T"].astype(np.float64)))
        out[label] = {"T_minus_oracle": err, "repartitions": m.repartitions(), "iterations": int(rr.iterations)}
        # lowest index on the same shards, for the record (this cloud: the duplicates are the same POINT, the sums do not move)
        m.close()
        assert int(rr.iterations) == 6 and err <= 2e-6, (label, out[label])
        if slack is not None:
            assert out[label]["repartitions"] >= 1
    _report("tie_rule_sharded.json", out)


def test_empty_target_shard_keeps_the_point_to_plane_branch(Context, orc):
    """a slab whose halo holds no target point at all (source reaching far beyond the target along the cut axis) must run the same
    epilogue branch as the other shards: 'the target has normals' is a property of the whole cloud"""
    from cilantro_amd.multi import PARTITION_SLABS, MultiDeviceRigidICP

    d = syn.make_pair(120_000, perturb=0.3)
    dst, dst_n, src, r2 = d["dst"], d["dst_n"], d["src"], d["max_sq_dist"]
    far = src.copy()
    ax = int(np.argmax(dst.max(axis=0) - dst.min(axis=0)))
    far[:, ax] -= 50.0      # two thirds of the source far below the target along the cut axis: the first of three slabs sees no target point
    src2 = np.ascontiguousarray(np.concatenate([far, src[: len(src) // 2]]))
    c = Context(); p = _icp_params(c, r2, 5); c.set_target(dst, dst_n); c.set_source(src2)
    ref = c.icp_run(p); c.close()
    m = MultiDeviceRigidICP([0, 0, 0]); m.set_slab_slack(0.5 * float(np.sqrt(r2))); m.set_clouds(dst, dst_n, src2, r2, PARTITION_SLABS)
    sizes = [m.shard_sizes(k) for k in range(3)]
    assert any(s[0] == 0 for s in sizes), sizes
    rr = m.icp_run(p); m.close()
    assert int(rr.iterations) == 5 and int(rr.last_ncorr) == int(ref.last_ncorr)
    assert float(rr.last_delta_norm) > 0.0
    assert np.abs(np.array(rr.T[:], np.float32).astype(np.float64) - np.array(ref.T[:], np.float32).astype(np.float64)).max() <= 2e-6


def _lattice(m):
    """target on a regular lattice (shuffled: the reference's tree depends on the order of the points); queries at the centres of its
    cubes (8 exactly equidistant corners each), of faces (4), of edges (2), and a few off-lattice ones"""
    ax = (np.arange(m, dtype=np.float32) * np.float32(0.125)).astype(np.float32)
    D = np.ascontiguousarray(np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3))
    rng = np.random.default_rng(5)
    D = np.ascontiguousarray(D[rng.permutation(len(D))])
    N = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(D), 1))
    c = (np.arange(m - 1, dtype=np.float32) * np.float32(0.125) + np.float32(0.0625)).astype(np.float32)
    cube = np.stack(np.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3)
    face = np.stack(np.meshgrid(c, c, ax[:-1], indexing="ij"), -1).reshape(-1, 3)[::3]
    edge = np.stack(np.meshgrid(c, ax[:-1], ax[:-1], indexing="ij"), -1).reshape(-1, 3)[::5]
    S = np.ascontiguousarray(np.concatenate([cube, face, edge, D[:5000] + np.float32(0.01)]).astype(np.float32))
    return D, N, S, np.float32(0.2 * 0.2), len(cube) + len(face) + len(edge)


def test_lattice_clouds_every_query_tied_eight_ways(Context, orc):
    """A target on a regular lattice and queries at the centres of its cubes: EVERY query has eight exactly equidistant nearest target
    points (any number of candidates: the old host path gave up beyond eight); queries on face and edge centres: four and two.  The
    engine names nanoflann's choice for each, in a single search (tiled and global-memory kernels, one lane and sixteen lanes per query)
    and through the loop's forms."""
    D, N, S, r2, n_tied = _lattice(40)
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    oi, _, ov = _ref_matches(tree, S, r2, len(S))
    I = np.eye(4, dtype=np.float32)
    out = {}
    for name, opts in (("tiled", {"tiled": 2}), ("one lane per query", {"tiled": 0, "group_search": 0}), ("16 lanes per query", {"tiled": 0, "group_search": 16})):
        ctx = Context()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_target(D, N); ctx.set_source(S)
        ctx.find_correspondences(I, float(r2), count=False)
        gi, gd = ctx.get_nn()
        seen, moved = ctx.tie_rule_stats()
        ctx.close()
        bad = int(np.count_nonzero(_signed(gi) != oi))
        out[name] = {"queries": int(len(S)), "tied": int(seen), "not_the_lowest_index": int(moved), "index_mismatches": bad}
        assert bad == 0, (name, out[name])
        assert seen >= n_tied, (name, out[name])
        assert np.array_equal(gd[oi >= 0].view(np.uint32), ov.view(np.uint32)), name
    # the loop (rigid motion of the lattice source: ties in every iteration), two forms
    for name, opts in (("adaptive", {}), ("warm forced", {"warm_start": 2})):
        ctx = Context()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_target(D, N); ctx.set_source(S)
        res = ctx.icp_run(_icp_params(ctx, r2, 4))
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi_l, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        bad = int(np.count_nonzero(_signed(li) != oi_l))
        out["loop, " + name] = {"index_mismatches_last_iteration": bad, "warm_iterations": ctx.last_warm_iterations()}
        ctx.close()
        assert bad == 0 and int(res.iterations) == 4, (name, out)
    _report("tie_rule_lattice.json", out)


def test_affine_loop_names_the_reference_s_points_too(Context, orc):
    """the Affine ICP instances search like the rigid ones (the reference's kd-tree over the TARGET does not know the transform family):
    with default options the set an affine loop leaves on the cloud with doubled and tripled target points equals nanoflann's under the loop's last transform, index
    for index (the tied queries stay tied under every transform: the candidates are the same point); under tie_rule 0 it does not."""
    name, D, N, S, r2 = list(_clouds())[1]
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    bad = {}
    for rule in (2, 0):
        ctx = Context(); ctx.set_option("tie_rule", rule); ctx.set_option("transform_mode", 1)
        ctx.set_target(D, N); ctx.set_source(S)
        p = _icp_params(ctx, r2, 4)
        p.w_p2p = 0.1
        res = ctx.icp_run(p)
        assert int(res.iterations) == 4
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        bad[rule] = int(np.count_nonzero(_signed(li) != oi))
        ctx.close()
    assert bad[2] == 0 and bad[0] > 100, bad


def test_target_shards_follow_the_reference_s_order_across_shards(Context, orc):
    """Partitioning A (index shards of the TARGET, MIN of packed keys) on the lattice: the equidistant corners of a query lie in different
    shards.  One key per query gives the lowest global index; with the whole target's order loaded, the second key (the match's place in
    the query's traversal of the whole tree, cilhip_icp_order_keys) gives nanoflann's point, index for index -- driven by hand over three
    engines for one search, then as loops: cilhip_multi_icp_run(partition = 2) on three shards of one device (the run notices the ties,
    builds the order once, repeats) against the single-context loop."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.multi import PARTITION_TARGET_SHARDS, MultiDeviceRigidICP

    D, N, S, r2, n_tied = _lattice(24)
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    oi, _, _ = _ref_matches(tree, S, r2, len(S))
    I = np.eye(4, dtype=np.float32)
    nd = len(D)
    cuts = [0, nd // 3, 2 * nd // 3, nd]
    dm = D.astype(np.float64).mean(axis=0).astype(np.float32)
    c = Context(); p = _icp_params(c, r2, 4); p.w_p2p = 0.1; c.close()
    engs = [distributed.HipTargetShardEngine(D[lo:hi], N[lo:hi], S, lo, dm, 0, whole_target=D) for lo, hi in zip(cuts[:-1], cuts[1:])]
    NONE = distributed.KEY_NONE
    for e in engs:
        e.begin(p, I)
    own = [e.partial_keys().clone() for e in engs]
    keys = torch.minimum(torch.minimum(own[0], own[1]), own[2])
    for e in engs:
        e.sums_from_keys(keys)
    assert not any(e.ordered for e in engs) and sum(e.ctx.tie_order_info()["pending"] for e in engs) >= n_tied // 2      # noticed: inside shards and across them
    kh = keys.cpu().numpy()
    lowest = np.where(kh == NONE, -1, kh & 0xFFFFFFFF).astype(np.int64)
    differ_lowest = int(np.count_nonzero(lowest != oi))
    assert differ_lowest > n_tied // 4 and np.array_equal(lowest >= 0, oi >= 0)
    for e in engs:
        e.load_tie_order(); e.begin(p, I)
    own = [e.partial_keys().clone() for e in engs]
    keys = torch.minimum(torch.minimum(own[0], own[1]), own[2])
    mine = [e.order_keys(keys).clone() for e in engs]
    okeys = torch.minimum(torch.minimum(mine[0], mine[1]), mine[2])
    gi = np.full(len(S), -1, np.int64); winners = np.zeros(len(S), np.int32)
    oh = okeys.cpu().numpy()
    for r in range(3):
        w = (mine[r].cpu().numpy() == oh) & (oh != NONE)
        gi[w] = own[r].cpu().numpy()[w] & 0xFFFFFFFF
        winners += w
    assert np.array_equal(winners, (oi >= 0).astype(np.int32))      # every matched query is won by exactly one shard
    assert np.array_equal(gi, oi), int(np.count_nonzero(gi != oi))
    sums = engs[0].sums_from_ordered_keys(keys, okeys).clone() + engs[1].sums_from_ordered_keys(keys, okeys) + engs[2].sums_from_ordered_keys(keys, okeys)
    del engs
    # the same first iteration's sums from one context over the whole target (default options: the reference's order)
    ctx = Context(); ctx.set_target(D, N); ctx.set_source(S)
    ref = ctx.icp_run(p)
    T_ref = np.array(ref.T[:], np.float32)
    assert ctx.tie_order_info()["loaded"]
    ctx.close()
    out = {"queries": int(len(S)), "tied": int(n_tied), "lowest_index_differs_from_the_reference": differ_lowest, "two_key_mismatches": 0}
    for devs in ([0, 0, 0], [0]):
        m = MultiDeviceRigidICP(devs); m.set_clouds(D, N, S, r2, PARTITION_TARGET_SHARDS)
        sizes = [m.shard_sizes(k) for k in range(len(devs))]
        assert sum(s[0] for s in sizes) == nd and all(s[1] == len(S) for s in sizes)
        rr = m.icp_run(p); m.close()
        T = np.array(rr.T[:], np.float32)
        err = float(np.abs(T.astype(np.float64) - T_ref.astype(np.float64)).max())
        out[f"loop_{len(devs)}_shards_minus_single_context"] = err
        assert int(rr.iterations) == 4 and int(rr.last_ncorr) == int(ref.last_ncorr) and err <= 2e-6, (devs, err, int(rr.last_ncorr), int(ref.last_ncorr))
    assert float(sums.sum().item()) == float(sums.sum().item())      # (finite)
    _report("tie_rule_target_shards.json", out)


def test_other_search_directions_follow_the_tree_over_the_transformed_source(Context, orc):
    """FIRST_TO_SECOND / BOTH: the reference searches a kd-tree built over the TRANSFORMED source, a new one per iteration
    (correspondence_search_kd_tree.hpp:185-222): among source points exactly equidistant from a target point it returns the one THAT tree's
    traversal meets first.  A source with doubled and tripled points (tied under every transform) and the lattice (every target point of
    the test tied 8 / 4 / 2 ways under the identity): with default options the pair lists equal the oracle's, pair for pair -- the first
    search notices the ties, the tables of the transformed source's tree are built on the host, the search is repeated --, through the
    filters and reciprocity, and as whole loops (a tree per iteration).  With tie_rule 0 a good part of the pairs name another point."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, CorrespondenceSearchHIP, SimpleCombinedMetricRigidICP3f

    code = {D.SECOND_TO_FIRST: 0, D.FIRST_TO_SECOND: 1, D.BOTH: 2}
    rng = np.random.default_rng(11)
    b = syn.make_pair(60_000, 45_000, with_normals=True)
    dup = rng.choice(len(b["src"]), 4000, replace=False)
    S = np.ascontiguousarray(np.concatenate([b["src"], b["src"][dup], b["src"][dup[:700]]]))
    S = np.ascontiguousarray(S[rng.permutation(len(S))])
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [0.004, -0.003, 0.002]
    c, s_ = np.cos(0.01), np.sin(0.01)
    T[:3, :3] = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    LD, LN, LS, lr2, n_tied = _lattice(20)
    out = {}
    for name, dst, dst_n, src, r2, Tq in (("duplicated source points", b["dst"], b["dst_n"], S, float(b["max_sq_dist"]), T),
                                          ("lattice source, cube centres as the target", LS, np.tile(np.array([[0, 0, 1.0]], np.float32), (len(LS), 1)), LD, float(lr2), np.eye(4, dtype=np.float32))):
        q = orc.transform_points(Tq, src)
        for rule in (2, 0):
            ctx = Context(); ctx.set_option("tie_rule", rule)
            ctx.set_target(dst, dst_n); ctx.set_source(src)
            eng = CorrespondenceSearchHIP(ctx=ctx).setMaxDistance(r2)
            for direction, recip, frac, o2o in ((D.FIRST_TO_SECOND, False, 1.0, False), (D.BOTH, False, 1.0, False), (D.BOTH, True, 1.0, False),
                                                (D.FIRST_TO_SECOND, False, 0.7, True)):
                eng.setSearchDirection(direction).setRequireReciprocality(recip).setInlierFraction(frac).setOneToOne(o2o)
                eng.findCorrespondences(Tq)
                g1, g2, gv = eng.getCorrespondences()
                o1, o2, ov = orc.find_correspondences_dir(dst, q, r2, code[direction], recip, frac, o2o)
                same = len(g1) == len(o1) and np.array_equal(g1, o1) and np.array_equal(g2, o2) and np.array_equal(gv, ov)
                key = f"{name}: {direction.name}{' reciprocal' if recip else ''}{' filtered' if o2o else ''}, tie_rule {rule}"
                out[key] = {"pairs": int(len(o1)), "identical": bool(same),
                            "pairs_naming_another_point": int(np.count_nonzero(g2 != o2)) if len(g1) == len(o1) else -1}
                if rule == 2:
                    assert same, (key, out[key])
            if rule == 0:
                assert any(v["pairs_naming_another_point"] > 50 for k, v in out.items() if k.startswith(name) and k.endswith("rule 0")), out
            ctx.close()
    # whole loops on the duplicated source: a tree over the transformed source per iteration
    for direction, recip in ((D.FIRST_TO_SECOND, False), (D.BOTH, False), (D.BOTH, True)):
        icp = SimpleCombinedMetricRigidICP3f(b["dst"], b["dst_n"], S)
        icp.correspondenceSearchEngine().setMaxDistance(float(b["max_sq_dist"])).setSearchDirection(direction).setRequireReciprocality(recip)
        icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, max_sq_dist=float(b["max_sq_dist"]), max_iter=5, conv_tol=0.0, direction=code[direction], reciprocal=recip)
        ro = orc.icp_run(b["dst"], b["dst_n"], S, p)
        err = float(np.linalg.norm(Tg.astype(np.float64) - ro["T"]))
        g1, g2, gv = icp.correspondenceSearchEngine().getCorrespondences()
        out[f"loop {direction.name}{' reciprocal' if recip else ''}"] = {"T_minus_oracle": err, "ncorr": int(icp.last_ncorr_), "oracle_ncorr": int(ro["last_ncorr"])}
        assert icp.getNumberOfPerformedIterations() == 5 and icp.last_ncorr_ == ro["last_ncorr"] and err <= 2e-6, (direction, recip, err)
    _report("tie_rule_directions.json", out)


def test_hundreds_of_copies_of_one_point(Context, orc):
    """300 copies of each of two points among 50 others, 500 queries: every query has hundreds of exactly equidistant nearest points (no
    cap on the candidates: they stream through the comparison), the reference's tree over such a cloud is all zero-extent splits -- the
    engine names nanoflann's pick for every query, one lane and several lanes per query, and as a loop."""
    rng = np.random.default_rng(23)
    A = np.array([0.25, 0.5, 0.75], np.float32); B = np.array([0.75, 0.25, 0.5], np.float32)
    D = np.concatenate([np.tile(A, (300, 1)), np.tile(B, (300, 1)), rng.random((50, 3), dtype=np.float32)])
    D = np.ascontiguousarray(D[rng.permutation(len(D))].astype(np.float32))
    N = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(D), 1))
    S = np.ascontiguousarray(np.concatenate([rng.random((400, 3), dtype=np.float32), D[rng.integers(0, len(D), 100)] + np.float32(1e-3)]).astype(np.float32))
    r2 = np.float32(4.0)
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    oi, _, ov = _ref_matches(tree, S, r2, len(S))
    I = np.eye(4, dtype=np.float32)
    for name, opts in (("default", {}), ("one lane per query", {"tiled": 0, "group_search": 0}), ("16 lanes per query", {"tiled": 0, "group_search": 16}), ("tiles", {"tiled": 2})):
        ctx = Context()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_target(D, N); ctx.set_source(S)
        ctx.find_correspondences(I, float(r2), count=False)
        gi, gd = ctx.get_nn()
        seen, moved = ctx.tie_rule_stats()
        assert np.array_equal(_signed(gi), oi), (name, int(np.count_nonzero(_signed(gi) != oi)))
        assert seen > 50 and moved > 50, (name, seen, moved)      # (the queries whose nearest point is one of the two copied ones)
        res = ctx.icp_run(_icp_params(ctx, r2, 3))
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi_l, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        assert np.array_equal(_signed(li), oi_l) and int(res.iterations) == 3, name
        ctx.close()


def _shim():
    import subprocess

    root = os.path.dirname(HERE)
    so = os.path.join(root, "tests", "cpp", "bin", "libtie_order_shim.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", os.path.join(root, "tests", "cpp", "tie_order_shim.cpp"), "-o", so])
    L = C.CDLL(so)
    L.tie_shim_build_mt.restype = C.c_void_p
    L.tie_shim_build_mt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint]
    L.tie_shim_free.argtypes = [C.c_void_p]
    L.tie_shim_same_as_tables.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def _device_tables(L, D):
    import time

    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.cilhip_tie_order_create(D.ctypes.data, len(D), C.byref(h))
    dt = time.perf_counter() - t0
    assert rc == capi.OK, rc
    nn = C.c_size_t(0); depth = C.c_int(0)
    L.cilhip_tie_order_tables(h, None, None, None, 0, C.byref(nn), C.byref(depth))
    leaf = np.zeros(max(len(D), 1), np.uint32); slot = np.zeros(max(len(D), 1), np.uint32); nodes = np.zeros((max(nn.value, 1), 4), np.uint32)
    L.cilhip_tie_order_tables(h, leaf.ctypes.data, slot.ctypes.data, nodes.ctypes.data, nn.value, None, None)
    L.cilhip_tie_order_destroy(h)
    return leaf, slot, nodes, nn.value, depth.value, dt


def test_device_built_order_tables_are_the_host_restatement_s(hip_lib, Context):
    """csrc/tie_build.hip against tests/cpp/tie_order_host.hpp (which tests/test_tie_order_cpu.py pins against the reference's own
    nanoflann): the SAME permutation slot for slot and the same leaf-to-root path (depths, split dimensions, divlow / divhigh bit for
    bit, child sides) for every point -- node ids are labels (breadth-first here, depth-first per worker there).  Random clouds with
    duplicated points and a flat sheet (many coordinates ON a split plane), the reference's sensor frame (a lattice), an integer lattice,
    the degenerate clouds of the CPU suite (hundreds of copies of one point, lines, clouds below a leaf) and sizes around the leaf size."""
    sh = _shim()
    rng = np.random.default_rng(11)
    big = rng.random((400_000, 3), dtype=np.float32)
    big[rng.choice(len(big), 20_000, replace=False)] = big[rng.choice(len(big), 20_000, replace=False)]
    big[:30_000, 2] = np.float32(0.25)
    f = np.load(os.path.join(HERE, "golden", "frames_full.npz"))
    A = np.array([0.25, 0.5, 0.75], np.float32); B = np.array([0.75, 0.25, 0.5], np.float32)
    clouds = {
        "400k random, duplicates, a flat sheet": big,
        "sensor frame_1": f["p1"],
        "integer lattice 30^3": np.stack(np.meshgrid(*[np.arange(30, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3),
        "300 + 300 copies of two points and 50 others": np.concatenate([np.tile(A, (300, 1)), np.tile(B, (300, 1)), rng.random((50, 3), dtype=np.float32)]),
        "a line with doubled points": np.concatenate([np.stack([np.linspace(0, 1, 400, dtype=np.float32), np.full(400, 0.5, np.float32), np.full(400, 0.5, np.float32)], 1)] * 2),
        "seven points, three of them the same": np.concatenate([rng.random((4, 3), dtype=np.float32), np.tile(A, (3, 1))]),
        "one point five times": np.tile(B, (5, 1)),
        "20000 copies of one point": np.tile(A, (20_000, 1)),
    }
    for n in (1, 2, 10, 11, 12, 21, 22, 23, 100, 1000):
        clouds[f"{n} random points"] = rng.random((n, 3), dtype=np.float32)
    report = {}
    for name, D in clouds.items():
        D = np.ascontiguousarray(D[rng.permutation(len(D))].astype(np.float32)) if len(D) > 1 else np.ascontiguousarray(D, np.float32)
        leaf, slot, nodes, nn, depth, dt = _device_tables(hip_lib, D)
        h = sh.tie_shim_build_mt(D.ctypes.data, len(D), 1)
        bad = np.zeros(1, np.uint32)
        same = sh.tie_shim_same_as_tables(h, len(D), leaf.ctypes.data, slot.ctypes.data, nodes.ctypes.data, nn, bad.ctypes.data)
        sh.tie_shim_free(h)
        assert same == 1, (name, "first differing point", int(bad[0]), "nodes", nn, "depth", depth)
        assert sorted(slot[: len(D)].tolist()) == list(range(len(D))), name          # a permutation
        report[name] = {"points": len(D), "nodes": nn, "depth": depth, "create_ms_incl_transfers": dt * 1e3}
    # ... and the build a context does for its own target when its searches tie, timed (no host copy of the cloud involved)
    for name, D in (("sensor frame_1", np.ascontiguousarray(f["p1"], np.float32)), ("400k", np.ascontiguousarray(big)), ("10M uniform", syn.make_dst(10_000_000))):
        ctx = Context(0)
        ctx.set_target(D, None)
        ctx.build_tie_order()
        ctx._ck(ctx._L.cilhip_synchronize(ctx._h))
        info = ctx.tie_order_info()
        assert info["loaded"] and info["builds"] == 1
        report["context build, " + name] = {"points": len(D), "build_ms": info["build_ms"]}
        ctx.close()
    _report("tie_order_device_build.json", report)


def test_feature_searches_follow_the_reference_s_feature_tree_on_ties(Context, orc):
    """PointNormalFeaturesAdaptor / PointColorFeaturesAdaptor / PointNormalColorFeaturesAdaptor (common_transformable_feature_adaptors.hpp:60-343):
    the reference searches a KDTree of DIM = 6 / 9 over (p, w n [, wc c]); among EXACTLY equal feature distances it keeps the first point that
    tree's traversal meets.  Targets whose feature vectors repeat -- doubled / tripled points with their normals and colours, a lattice with one
    normal for all -- through both search kernels, with DEFAULT options: every index equals the reference's own nanoflann instantiated for that
    DIM (oracle/_ref), where a lowest-index rule differs on thousands of queries; tie_rule 1 (tables up front) the same."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref is not built")
    rng = np.random.default_rng(51)
    base = syn.make_pair(30_000, perturb=0.4)
    h = base["h"]
    nd = len(base["dst"])
    D = np.ascontiguousarray(np.concatenate([base["dst"], base["dst"][:9000], base["dst"][:3000]]))
    N = np.ascontiguousarray(np.concatenate([base["dst_n"], base["dst_n"][:9000], base["dst_n"][:3000]]))
    Cd = rng.random((nd, 3)).astype(np.float32)
    Cd = np.ascontiguousarray(np.concatenate([Cd, Cd[:9000], Cd[:3000]]))
    S = base["src"]
    Ti = np.linalg.inv(base["T_true"].astype(np.float64)).astype(np.float32)
    Sn = orc.transform_normals(Ti, base["dst_n"])
    Cs = np.clip(Cd[:len(S)] + rng.normal(0, 0.05, (len(S), 3)), 0, 1).astype(np.float32)
    lat = np.ascontiguousarray(np.stack(np.meshgrid(*[np.arange(22, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) * np.float32(h))
    latn = np.ascontiguousarray(np.tile(np.array([0.0, 0.6, 0.8], np.float32), (len(lat), 1)))
    lats = np.ascontiguousarray(lat[:6000] + np.float32(0.5 * h))              # queries on cell centres: eight equidistant corners each
    T = base["T_true"].astype(np.float32).copy(); T[:3, 3] += np.float32(0.15 * h)
    I = np.eye(4, dtype=np.float32)
    wn, wc = 0.6 * h, 0.8 * h
    r2 = float((3.0 * h) ** 2)
    report = {}
    cases = [("doubled points, 6-D normals", D, N, None, S, Sn, None, T, 0), ("doubled points, 6-D colours", D, N, Cd, S, Sn, Cs, T, 1),
             ("doubled points, 9-D", D, N, Cd, S, Sn, Cs, T, 2), ("lattice, one normal, 6-D", lat, latn, None, lats, latn[:len(lats)], None, I, 0)]
    for name, dst, dn, dc, src, sn, sc, Tq, kind in cases:
        if kind == 0:
            dstf, srcf = orc.point_normal_features(dst, dn, wn), orc.point_normal_features(src, sn, wn)
            qf = orc.transform_features6(Tq, srcf, 0)
            oi, osrc, od = orc.find_correspondences_feat6(dstf, qf, r2, use_ref=True)
        elif kind == 1:
            dstf, srcf = orc.point_normal_features(dst, dc, wc), orc.point_normal_features(src, sc, wc)
            qf = orc.transform_features6(Tq, srcf, 2)
            oi, osrc, od = orc.find_correspondences_feat6(dstf, qf, r2, use_ref=True)
        else:
            dstf, srcf = orc.point_normal_color_features(dst, dn, dc, wn, wc), orc.point_normal_color_features(src, sn, sc, wn, wc)
            qf = orc.transform_features9(Tq, srcf, 0)
            oi, osrc, od = orc.find_correspondences_feat9(dstf, qf, r2, use_ref=True)
        want = np.full(len(src), -1, np.int64); want[osrc] = oi
        for tiled in (0, 2):
            for rule in (2, 1, 0):
                ctx = Context()
                ctx.set_option("tiled", tiled); ctx.set_option("tie_rule", rule)
                ctx.set_target(dst, dn); ctx.set_source(src, sn)
                if kind >= 1:
                    ctx.set_color_features(dc, sc)
                ctx.set_option("feature_kind", kind)
                if kind != 1:
                    ctx.set_option("feature_normal_weight", wn)
                else:
                    ctx.set_option("feature_normal_weight", wc)
                if kind == 2:
                    ctx.set_option("feature_color_weight", wc)
                ctx.find_correspondences(Tq, r2)
                gi, gd = ctx.get_nn()
                gi = _signed(gi)
                info = ctx.tie_order_info()
                ctx.close()
                nbad = int(np.count_nonzero(gi != want))
                if rule == 0:
                    report.setdefault(name, {})["queries"] = len(src)
                    report[name]["indices a lowest-index rule answers differently"] = nbad
                else:
                    assert nbad == 0, (name, tiled, rule, nbad)
                    assert info["loaded"], (name, tiled, rule)
                if rule != 0:      # ... and the distances bit for bit (osrc ascends: the reference's list is in source order)
                    assert np.array_equal(gd[gi >= 0].view(np.uint32), od.view(np.uint32))
    assert report["doubled points, 6-D normals"]["indices a lowest-index rule answers differently"] > 1000
    assert report["lattice, one normal, 6-D"]["indices a lowest-index rule answers differently"] > 1000
    _report("tie_rule_features.json", report)


def test_kmeans_kd_branch_names_the_centroid_the_reference_s_tree_meets_first(orc, hip_lib):
    """KMeans use_kd_tree = true (clustering/kmeans.hpp:86-94): the reference builds a KDTree over the centroids every iteration and
    keeps the first centroid its traversal meets among exactly equidistant ones.  Lattice centroids (and duplicated ones) with points on
    the half-lattice: most points are tied 2, 4 or 8 ways.  One assignment pass against the reference's own nanoflann over the same
    centroids, label for label; whole Lloyd runs against a loop of that same search + the reference's update; rule 0 = lowest index."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built")
    from cilantro_amd.clustering import KMeans3f, kmeans_assign
    from cilantro_amd.normal_estimation import set_knn_tie_rule
    rng = np.random.default_rng(5)
    differ_total = 0
    for g, dup in ((2, 0), (4, 0), (4, 9), (10, 0), (12, 40)):
        lat = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        c0 = lat[rng.permutation(len(lat))]
        if dup:
            c0 = np.concatenate([c0, c0[rng.integers(0, len(c0), dup)]])[rng.permutation(len(c0) + dup)]
        x = (rng.integers(-1, 2 * g + 1, size=(60000, 3)) * 0.5).astype(np.float32)
        x[::7] += rng.normal(0, 0.05, size=x[::7].shape).astype(np.float32)      # (and some untied ones)
        tree = orc.KDTree(c0, use_ref=True)
        di, si, _ = tree.find_correspondences(x, 3.0e38)
        assert len(si) == len(x)
        lab_r = np.empty(len(x), np.int64); lab_r[si] = di
        lab_g = kmeans_assign(x, c0, use_kd_tree=True)
        assert np.array_equal(lab_g.astype(np.int64), lab_r), (g, dup, int((lab_g != lab_r).sum()))
        # (from 64 centroids on that was the PRUNED pass -- centroid grid, proof, tables built only because ties were met --: the exhaustive one agrees)
        from cilantro_amd import clustering
        clustering.set_pruning(False)
        try:
            assert np.array_equal(kmeans_assign(x, c0, use_kd_tree=True).astype(np.int64), lab_r), (g, dup)
        finally:
            clustering.set_pruning(True)
        # lowest index among equals: numpy's argmin over exact distances (half-lattice coordinates: every f32 operation is exact)
        lab_o = np.concatenate([np.argmin(((x[a:a + 4096, None, :] - c0[None]) ** 2).sum(-1), axis=1) for a in range(0, len(x), 4096)])
        tied = x[:, 0] * 2 == np.round(x[:, 0] * 2)      # (the jittered points are not compared under rule 0: their distances round)
        differ_total += int((lab_o != lab_r)[tied].sum())
        try:
            set_knn_tie_rule(0)
            assert np.array_equal(kmeans_assign(x, c0, use_kd_tree=True).astype(np.int64)[tied], lab_o[tied])
        finally:
            set_knn_tie_rule(2)
    assert differ_total > 1000      # the data does tell the two rules apart
    # whole runs: the reference's loop with its tree search as the assignment step (labels -> f32 serial sums like kmeans.hpp:126-131)
    g = 4
    lat = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    c0 = lat[rng.permutation(len(lat))][:24].copy()
    x = (rng.integers(0, 2 * g - 1, size=(20000, 3)) * 0.5).astype(np.float32)
    km = KMeans3f(x).cluster(c0, max_iter=1, tol=0.0, use_kd_tree=True)
    tree = orc.KDTree(c0, use_ref=True)
    di, si, _ = tree.find_correspondences(x, 3.0e38)
    lab_r = np.empty(len(x), np.int64); lab_r[si] = di
    assert np.array_equal(km.getPointToClusterIndexMap().astype(np.int64), lab_r)
    cen = np.stack([x[lab_r == j].astype(np.float64).mean(0) if (lab_r == j).any() else c0[j] for j in range(len(c0))])
    full = np.array([(lab_r == j).any() for j in range(len(c0))])
    assert np.abs(km.getClusterCentroids()[full] - cen[full]).max() <= 1e-5
