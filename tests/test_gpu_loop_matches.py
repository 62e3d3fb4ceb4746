"""The correspondence set the ICP loop's OWN kernels leave behind, index for index.

north_star bar #1 is "correspondence indices bit-exact".  cilhip_find_correspondences is compared with the oracle / the
reference's nanoflann all over tests/test_gpu_parity.py; the kernels that run INSIDE cilhip_icp_run -- the LDS tiles with the
accumulation inside, the warm-started kernel k_warm<ACC, REC> the headline number is measured on -- used to be compared
through counts and transforms only.  Here their matches are pulled out after the run (the engine keeps the last iteration's
set like the reference's, correspondence_search_kd_tree.hpp:231) and every index and every d2 BIT is compared with
  * a fresh search under the same transform by the search-only kernels (another code path altogether), and
  * the reference's own nanoflann (oracle/_ref) over a >= 100k-query sample at that transform, mismatches classified
    exact / tie / nearer / worse -- worse must be 0 and, on clouds without exact duplicates, everything exact.
Clouds: the synthetic recipe, a target with holes and exact duplicate points, the reference's real sensor frames at full
resolution (tests/golden/frames_full.npz: frame_1.ply / frame_2.ply of examples/test_clouds, 120k points each), and
BASELINE configs[2] at its full 10M <-> 10M size.
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
    """measured figures, kept next to the run (gpurun_out/ travels back from the GPU box)"""
    import json

    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass
    print(name, obj)


def _params(ctx, metric, w_p2p, r2, iters):
    p = capi.IcpParams()
    ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = metric, w_p2p, float(r2), iters, 0.0
    return p


def _signed(idx):
    gi = idx.astype(np.int64)
    gi[idx == capi.NONE_IDX] = -1
    return gi


def _classify(gi, gd, oi, od):
    """mismatching queries: equal-distance tie / GPU strictly nearer / GPU worse"""
    bad = np.nonzero(gi != oi)[0]
    ties = nearer = worse = 0
    for i in bad:
        if gi[i] >= 0 and oi[i] >= 0 and gd[i] == od[i]:
            ties += 1
        elif gi[i] >= 0 and (oi[i] < 0 or gd[i] < od[i]):
            nearer += 1
        else:
            worse += 1
    return len(bad), ties, nearer, worse


def _loop_matches(Context, D, N, S, r2, iters, metric=capi.METRIC_COMBINED, w_p2p=0.0, options=(), expect_origin=1):
    """run `iters` iterations (tolerance 0) and return what the loop's last iteration left: (idx, d2, T it searched under, ctx)"""
    ctx = Context()
    for k, v in options:
        ctx.set_option(k, v)
    ctx.set_target(D, N)
    ctx.set_source(S)
    res = ctx.icp_run(_params(ctx, metric, w_p2p, r2, iters))
    assert int(res.iterations) == iters
    origin = ctx.last_matches_origin()
    assert origin == expect_origin, (origin, options)
    T = ctx.matches_transform()
    idx, d2 = ctx.get_nn()
    return _signed(idx), d2, T, ctx, int(res.last_ncorr)


def _check_against_fresh_search_and_reference(Context, orc, name, D, N, S, r2, gi, gd, T, ncorr, sample_n, rng, allow_ties):
    n = len(S)
    assert int(np.count_nonzero(gi >= 0)) == ncorr, name                  # last_ncorr counts exactly this set
    # (1) a fresh search by the search-only kernels under the same transform: every index, every d2 bit
    for tiled in (2, 0):
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(D, N)
        ctx.set_source(S)
        ctx.find_correspondences(T, float(r2), count=False)
        assert ctx.last_matches_origin() == 3
        fi, fd = ctx.get_nn()
        fi = _signed(fi)
        ctx.close()
        assert np.array_equal(gi, fi), (name, tiled, np.nonzero(gi != fi)[0][:10])
        m = gi >= 0
        assert np.array_equal(gd[m].view(np.uint32), fd[m].view(np.uint32)), (name, tiled)
    # (2) the reference's nanoflann over a sample of the queries at that transform
    sample = np.sort(rng.choice(n, min(sample_n, n), replace=False))
    q = orc.transform_points(T, S[sample])
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    o1, o2, ov = tree.find_correspondences(q, float(r2))
    oi = np.full(len(sample), -1, np.int64)
    od = np.zeros(len(sample), np.float32)
    oi[o2] = o1
    od[o2] = ov
    nbad, ties, nearer, worse = _classify(gi[sample], gd[sample], oi, od)
    assert worse == 0 and nearer == 0, (name, nbad, ties, nearer, worse)
    if not allow_ties:
        assert nbad == 0, (name, nbad, ties)
    m = (gi[sample] >= 0) & (gi[sample] == oi)
    assert np.array_equal(gd[sample][m].view(np.uint32), od[m].view(np.uint32)), name
    return {"queries": int(n), "matched": int(ncorr), "sample": int(len(sample)), "mismatches": int(nbad), "ties": int(ties), "nearer": int(nearer),
            "worse": int(worse), "knn": "reference nanoflann" if tree.use_ref else "oracle kd-tree"}


def _holes_and_duplicates(rng):
    base = syn.make_pair(1_200_000, perturb=0.5)
    dst, dst_n = base["dst"], base["dst_n"]
    keep = ~(((dst[:, 0] > 0.3) & (dst[:, 0] < 0.36)) | ((dst[:, 2] > 0.7) & (dst[:, 2] < 0.73)))
    dup = rng.choice(np.nonzero(keep)[0], 5000, replace=False)
    D = np.ascontiguousarray(np.concatenate([dst[keep], dst[dup]]))
    N = np.ascontiguousarray(np.concatenate([dst_n[keep], dst_n[dup]]))
    return base, D, N


def test_warm_kernel_matches_index_for_index(Context, orc):
    """k_warm<ACC, REC>: the first warm iteration of a stretch (REC = 1: gathers through the stored positions, writes the
    match records) and the record-reading ones (REC = 2), all four accumulation variants, on the uniform recipe and on a
    target with holes and exact duplicates (nearest-other distance 0: never settled by the shortcut)."""
    rng = np.random.default_rng(11)
    base, Dh, Nh = _holes_and_duplicates(rng)
    report = {}
    for name, D, N, allow_ties in (("uniform", base["dst"], base["dst_n"], False), ("holes+duplicates", Dh, Nh, False)):      # (default options: ties take the reference's order -- every index equal)
        S, r2 = base["src"], base["max_sq_dist"]
        for metric, w_p2p, mname in ((capi.METRIC_COMBINED, 0.0, "plane"), (capi.METRIC_COMBINED, 0.1, "both"), (capi.METRIC_POINT_TO_POINT, 0.0, "kabsch")):
            for iters in (2, 5):         # the last iteration is REC = 1 / REC = 2
                if mname != "plane" and iters == 2 and name == "uniform":
                    continue
                gi, gd, T, ctx, nc = _loop_matches(Context, D, N, S, r2, iters, metric, w_p2p, (("warm_start", 2), ("tiled", 2)))
                assert ctx.last_warm_iterations() == iters - 1
                ctx.close()
                report[f"{name}/{mname}/{iters}"] = _check_against_fresh_search_and_reference(
                    Context, orc, (name, mname, iters), D, N, S, r2, gi, gd, T, nc, 100_000, rng, allow_ties)
    # point metric alone (w_p2pl = 0): the fourth ACC variant
    ctx = Context()
    ctx.set_option("warm_start", 2); ctx.set_option("tiled", 2)
    ctx.set_target(base["dst"], base["dst_n"]); ctx.set_source(base["src"])
    p = _params(ctx, capi.METRIC_COMBINED, 1.0, base["max_sq_dist"], 4)
    p.w_p2pl = 0.0
    res = ctx.icp_run(p)
    assert ctx.last_matches_origin() == 1 and ctx.last_warm_iterations() == 3
    T = ctx.matches_transform()
    idx, gd = ctx.get_nn()
    ctx.close()
    report["uniform/point/4"] = _check_against_fresh_search_and_reference(Context, orc, "point", base["dst"], base["dst_n"], base["src"], base["max_sq_dist"],
                                                                         _signed(idx), gd, T, int(res.last_ncorr), 100_000, rng, False)
    _report("warm_matches_1m.json", report)


def _independent_source(base, n, seed):
    """an independent uniform sample of the target's volume, moved by the inverse of the recipe's motion: its matches sit at about
    half the target's point spacing (the regime of any two separately sampled clouds)"""
    rng = np.random.default_rng(seed)
    Ti = np.linalg.inv(base["T_true"].astype(np.float64))
    return (rng.random((n, 3), dtype=np.float32).astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)


def test_margin_records_every_route(Context, orc):
    """The warm-started form's MARGIN proof on every route into it, index for index: the match records written by
    the accumulating tile kernel itself (tile one pass -> record-reading warm kernel), the margin keys the search-only tile
    kernel leaves next to its matches (two passes -> record-writing warm kernel), and the per-lane start (no keys: the
    nearest-other-point table and the warm kernel's own full search).  Clouds: the recipe, an INDEPENDENTLY sampled source
    (matches at half the point spacing: nothing the old half-the-target's-spacing rule could settle), a target with holes
    and exact duplicates (margin 0: never settled without a search), and a source whose first warm iteration comes after
    a large step (margins spent: the listed queries' own search)."""
    rng = np.random.default_rng(21)
    base, Dh, Nh = _holes_and_duplicates(rng)
    n = len(base["src"])
    far = syn.make_pair(n, perturb=0.9)
    clouds = (("uniform", base["dst"], base["dst_n"], base["src"], False),
              ("independent", base["dst"], base["dst_n"], _independent_source(base, n, 5), False),
              ("holes+duplicates", Dh, Nh, base["src"], False),
              ("far start", far["dst"], far["dst_n"], far["src"], False))
    routes = (("tile records", (("tiled", 2), ("warm_enter_fraction", 1.0e9), ("warm_forecast", 0))),
              ("search keys", (("tiled", 2), ("tile_accumulation", 0), ("warm_enter_fraction", 1.0e9), ("warm_forecast", 0))),
              ("per lane", (("tiled", 0), ("warm_enter_fraction", 1.0e9), ("warm_forecast", 0))),
              ("adaptive", ()))
    r2 = base["max_sq_dist"]
    report = {}
    for cname, D, N, S, allow_ties in clouds:
        for rname, opts in routes:
            for metric, w_p2p, mname in ((capi.METRIC_COMBINED, 0.0, "plane"), (capi.METRIC_POINT_TO_POINT, 0.0, "kabsch")):
                if mname == "kabsch" and rname in ("per lane", "adaptive"):
                    continue
                gi, gd, T, ctx, nc = _loop_matches(Context, D, N, S, r2, 7, metric, w_p2p, opts)
                warm = ctx.last_warm_iterations()
                trace = ctx.last_run_trace()
                ctx.close()
                if rname != "adaptive":
                    # (bar out of the way: every iteration from the third on is warm-started unless one of them had to search a
                    #  quarter of its queries -- then one cold iteration follows)
                    assert warm >= 3, (cname, rname, mname, warm, trace)
                chk = _check_against_fresh_search_and_reference(Context, orc, (cname, rname, mname), D, N, S, r2, gi, gd, T, nc, 100_000, rng, allow_ties)
                report[f"{cname}/{rname}/{mname}"] = {"warm_iterations": warm, "forms": [t["form"] for t in trace], "listed": [t["listed"] for t in trace],
                                                      "step_over_cell": None, **chk}
    _report("margin_routes.json", report)


def test_kernel_timing_by_sample_changes_nothing_but_the_events(Context):
    """Option kernel_timing_stride: with kernel timing on, iterations 0-2 and every stride-th one carry events (attached to the kernels' own
    dispatch packets); the run's result is bitwise the untimed run's, the timed iterations are the announced ones, every one of them
    reports a positive kernel time, and the per-form sums are over exactly those."""
    d = syn.make_pair(700_000, perturb=0.3)
    res = {}
    for timing, stride in ((0, 1), (1, 1), (1, 4)):
        ctx = Context()
        ctx.set_option("kernel_timing_stride", stride)
        ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
        ctx.enable_kernel_timing(bool(timing))
        r = ctx.icp_run(_params(ctx, capi.METRIC_COMBINED, 0.0, d["max_sq_dist"], 20))
        its = ctx.last_iteration_timing()
        ft = ctx.last_form_timing()
        ctx.close()
        res[(timing, stride)] = bytes(np.array(r.T[:], np.float32))
        if not timing:
            assert its == []
        else:
            want = list(range(20)) if stride == 1 else [0, 1, 2, 4, 8, 12, 16]
            assert [i for i, _ in its] == want and all(ms > 0.0 for _, ms in its), its
            assert sum(n for _, n in ft.values()) == len(want)
            assert abs(sum(ms for ms, _ in ft.values()) - sum(ms for _, ms in its)) <= 1e-3
    assert res[(0, 1)] == res[(1, 1)] == res[(1, 4)]


def test_ab_switches_never_change_a_result(Context):
    """The options that exist for A/B runs -- "pair_records", "tile_records", "warm_extra_fraction" and the "kernel_timing" switch --
    choose HOW a loop gets to its matches and sums, never which: the last iteration's matches index for index, the distances bit for bit,
    the transform to the order of the f64 additions (<= 1e-6), on the recipe and on an independent source (the regime where a
    warm-started iteration searches queries again)."""
    rng = np.random.default_rng(21)
    d = syn.make_pair(1_200_000, perturb=0.3)
    cases = {"recipe": d["src"], "independent": _independent_source(d, 1_200_000, 5)}
    for name, S in cases.items():
        base = None
        for opts in ((), (("pair_records", 0),), (("tile_records", 0),), (("warm_extra_fraction", 0.25),), (("warm_extra_fraction", 0.01),), (("kernel_timing", 1),),
                     (("pair_records", 0), ("tile_records", 0), ("kernel_timing", 1))):
            gi, gd, T, ctx, nc = _loop_matches(Context, d["dst"], d["dst_n"], S, d["max_sq_dist"], 8, options=opts)
            for k, v in opts:
                assert ctx.get_option(k) == pytest.approx(v)
            ctx.close()
            if base is None:
                base = (gi, gd, T, nc)
                continue
            assert nc == base[3] and np.array_equal(gi, base[0]), (name, opts)
            m = gi >= 0
            assert np.array_equal(gd[m].view(np.uint32), base[1][m].view(np.uint32)), (name, opts)
            assert np.abs(T - base[2]).max() <= 1e-6, (name, opts, np.abs(T - base[2]).max())


def test_tile_and_lane_loop_kernels_matches_index_for_index(Context, orc):
    """The other forms an iteration can take, same check: the LDS tiles with the accumulation inside (one pass), the two-pass
    form (tiled search with its 3x3x3 pass + streaming accumulation), the per-lane search; and loops whose kernels keep no
    per-query matches answer through a search repeated on demand (origin 2) -- the same set."""
    rng = np.random.default_rng(12)
    d = syn.make_pair(1_200_000, perturb=0.9)          # starts far: the first iterations leave many octant proofs open
    D, N, S, r2 = d["dst"], d["dst_n"], d["src"], d["max_sq_dist"]
    forms = (("tile one pass", (("warm_start", 0), ("tiled", 2), ("tile_accumulation", 2)), 2),         # stores nothing with warm_start = 0: searched again
             ("tile one pass, stored", (("warm_start", 1), ("tiled", 2), ("tile_accumulation", 2)), 1),
             ("two pass tiled", (("warm_start", 0), ("tiled", 2), ("tile_accumulation", 0)), 1),
             ("two pass per lane", (("warm_start", 0), ("tiled", 0)), 1),
             ("per lane fused", (("warm_start", 0), ("tiled", 0), ("fused", 1)), 2))
    for name, opts, origin in forms:
        for iters in (1, 3):
            gi, gd, T, ctx, nc = _loop_matches(Context, D, N, S, r2, iters, options=opts, expect_origin=origin)
            ctx.close()
            _check_against_fresh_search_and_reference(Context, orc, (name, iters), D, N, S, r2, gi, gd, T, nc, 100_000, rng, False)
    # post-filters: the set of the last iteration is the FILTERED one (searched again, filtered again)
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f
    icp = SimpleCombinedMetricRigidICP3f(D, N, S)
    icp.correspondenceSearchEngine().setMaxDistance(float(r2)).setInlierFraction(0.7)
    icp.setMaxNumberOfIterations(3).setConvergenceTolerance(0.0).estimate()
    i1, i2, v = icp.correspondenceSearchEngine().getCorrespondences()
    assert icp._ctx.last_matches_origin() == 2 and len(i1) == icp.last_ncorr_ and np.all(np.diff(v) >= 0)
    Tl = icp._ctx.matches_transform()
    eng = icp.correspondenceSearchEngine()
    eng.findCorrespondences(Tl)
    j1, j2, w = eng.getCorrespondences()
    assert np.array_equal(i1, j1) and np.array_equal(i2, j2) and np.array_equal(v, w)


def test_real_sensor_frames_every_form(Context, orc):
    """examples/test_clouds at full resolution (120k points; no voxel grid), two registrations: frame_1 against a jittered,
    moved copy of itself (the recipe of examples/rigid_icp.cpp:25-65 at the sensor's resolution) and frame_1 against frame_2
    (an independent sampling of the scene).  Every kernel form forced in turn -- tiles (one pass / two passes), per-lane, warm
    -- and the adaptive default: the last iteration's matches against a fresh search and the reference's nanoflann; the loops
    against the oracle's.  Which form the adaptive loop picks is recorded (profiles/r03_real_cloud.txt)."""
    f = np.load(os.path.join(HERE, "golden", "frames_full.npz"))
    p1, n1, p2 = f["p1"], f["n1"], f["p2"]
    rng = np.random.default_rng(13)
    # rigid_icp.cpp:33-62 -- src = dst + jitter, dst keeps x > -0.4, src moved by tf_ref (a smaller motion: the raw cloud's
    # spacing is ~1.5 mm, the example's 0.1 rad / 0.2 m start is far outside any nearest-neighbour basin at that resolution)
    jit = (np.float32(0.0005) * rng.uniform(-1, 1, p1.shape)).astype(np.float32)
    Tm = np.eye(4)
    Tm[:3, :3] = syn.rot_xyz(-0.004, 0.004, -0.004)
    Tm[:3, 3] = [-0.003, -0.001, 0.002]
    src_self = ((p1 + jit).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
    keep = p1[:, 0] > -0.4
    D, N = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep])
    report = {}
    tree_ref = orc.KDTree(D, use_ref=orc.ref_available())
    cases = (("frame_1 vs moved+jittered frame_1", src_self, np.float32(0.01 * 0.01)), ("frame_1 vs frame_2", np.ascontiguousarray(p2), np.float32(0.02 * 0.02)))
    forms = (("adaptive", (), None), ("warm forced, per-lane start", (("warm_start", 2), ("tiled", 0), ("group_search", 0)), 1), ("warm forced, tiled start", (("warm_start", 2), ("tiled", 2)), 1),
             ("tiles one pass", (("warm_start", 1), ("tiled", 2), ("tile_accumulation", 2)), 1), ("tiles two passes", (("warm_start", 0), ("tiled", 2), ("tile_accumulation", 0)), 1),
             ("per lane", (("warm_start", 0), ("tiled", 0), ("group_search", 0)), 1), ("16 lanes per query", (("warm_start", 0), ("tiled", 0), ("group_search", 16)), 1),
             ("8 lanes per query, warm forced", (("warm_start", 2), ("tiled", 0), ("group_search", 8)), 1))
    for cname, S, r2 in cases:
        po = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
        # Two oracle loops.  (1) the reference's order of ties: nanoflann keeps the first candidate met in ITS tree traversal
        # (core/kd_tree.hpp:82-90); (2) the same loop with the lowest target index on ties (exhaustive argmin).  They only differ
        # on exactly equal f32 distances -- which the depth sensor's lattice (232 distinct z values, a regular pixel grid) does
        # produce between two raw frames under the identity: 321 ties in the first search of frame_1 vs frame_2, a 4.5e-6
        # difference in that iteration's update (tools/real_cloud_diag2.py).  With DEFAULT options (tie_rule 2) the HIP loop must
        # equal (1) in every kernel form; under tie_rule 0 it equals (2) (one form checked, the distance between the two recorded).
        ro = orc.icp_run(D, N, S, po)
        T_low = np.eye(4, dtype=np.float32)
        ties_seen = 0
        for _ in range(6):
            q = orc.transform_points(T_low, S)
            bi, _bd = orc.nn_brute(D, q, float(r2))
            o1, o2, ov = tree_ref.find_correspondences(q, float(r2))
            oi = np.full(len(S), -1, np.int64); oi[o2] = o1
            ties_seen += int(np.count_nonzero(bi != oi))
            si = np.nonzero(bi >= 0)[0]
            T_low, _ = orc.icp_update(D, N, S, T_low, bi[si], si, po)
            nc_low = len(si)
        for fname, opts, origin in forms:
            ctx = Context()
            for k, v in opts:
                ctx.set_option(k, v)
            ctx.set_target(D, N)
            ctx.set_source(S)
            res = ctx.icp_run(_params(ctx, capi.METRIC_COMBINED, 0.0, r2, 6))
            Tg = np.array(res.T[:], np.float32).reshape(4, 4).T
            err = float(np.linalg.norm(Tg.astype(np.float64) - T_low.astype(np.float64)))
            err_ref = float(np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)))
            assert err_ref <= 2e-6 and int(res.last_ncorr) == ro["last_ncorr"], (cname, fname, err_ref, int(res.last_ncorr), ro["last_ncorr"])      # (the order of the f64 additions differs by form)
            assert err <= (1e-5 if ties_seen == 0 else 1e-4), (cname, fname, err, ties_seen)
            if fname == "per lane":      # tie_rule 0: the lowest-index loop
                c0 = Context()
                for k, v in opts:
                    c0.set_option(k, v)
                c0.set_option("tie_rule", 0)
                c0.set_target(D, N); c0.set_source(S)
                r0 = c0.icp_run(_params(c0, capi.METRIC_COMBINED, 0.0, r2, 6))
                c0.close()
                T0g = np.array(r0.T[:], np.float32).reshape(4, 4).T
                assert float(np.linalg.norm(T0g.astype(np.float64) - T_low.astype(np.float64))) <= 2e-6 and int(r0.last_ncorr) == nc_low, (cname, fname)
            one, two = ctx.last_run_forms()
            warm = ctx.last_warm_iterations()
            if origin is not None:
                assert ctx.last_matches_origin() == origin, (cname, fname)
            T = ctx.matches_transform()
            idx, gd = ctx.get_nn()
            ctx.close()
            chk = _check_against_fresh_search_and_reference(Context, orc, (cname, fname), D, N, S, r2, _signed(idx), gd, T, int(res.last_ncorr), 120_000, rng, False)
            report[f"{cname} / {fname}"] = {"one_pass_iterations": one, "two_pass_iterations": two, "warm_iterations": warm, "T_minus_oracle_lowest_index_ties": err,
                                            "T_minus_oracle_nanoflann_tie_order": err_ref, "tie_queries_over_6_searches": ties_seen, **chk}
        assert report[f"{cname} / warm forced, per-lane start"]["warm_iterations"] == 5
    _report("real_cloud_forms.json", report)


def test_tie_count_tells_when_index_parity_is_not_guaranteed(Context, orc):
    """cilhip_get_tie_count: queries with two or more target points at exactly the smallest f32 distance -- the only place where the
    engine (lowest index) and the reference (first met in its traversal, core/kd_tree.hpp:82-90) may differ.  Against the oracle's
    exhaustive count: 0 on the uniform recipe, the duplicated points' queries on a target with exact duplicates, and on the
    reference's raw frame_1 -> frame_2 under the identity at least the 321 queries where the two rules do pick different points."""
    rng = np.random.default_rng(31)
    base = syn.make_pair(60_000, perturb=0.3)
    T = np.eye(4, dtype=np.float32)
    ctx = Context(); ctx.set_target(base["dst"], base["dst_n"]); ctx.set_source(base["src"])
    assert ctx.tie_count(T, float(base["max_sq_dist"])) == 0 == orc.count_ties_brute(base["dst"], orc.transform_points(T, base["src"]), float(base["max_sq_dist"]))
    ctx.close()
    dup = rng.choice(len(base["dst"]), 700, replace=False)
    D = np.ascontiguousarray(np.concatenate([base["dst"], base["dst"][dup]]))
    ctx = Context(); ctx.set_target(D, None); ctx.set_source(base["src"])
    for Tq in (T, base["T_true"].astype(np.float32)):
        want = orc.count_ties_brute(D, orc.transform_points(Tq, base["src"]), float(base["max_sq_dist"]))
        got = ctx.tie_count(Tq, float(base["max_sq_dist"]))
        assert got == want and want > 0, (got, want)
    ctx.close()
    f = np.load(os.path.join(HERE, "golden", "frames_full.npz"))
    p1, p2 = f["p1"], f["p2"]
    Df = np.ascontiguousarray(p1[p1[:, 0] > -0.4]); S = np.ascontiguousarray(p2)
    r2 = float(np.float32(0.02 * 0.02))
    ctx = Context(); ctx.set_target(Df, None); ctx.set_source(S)
    got = ctx.tie_count(T, r2)
    ctx.close()
    want = orc.count_ties_brute(Df, S, r2)
    bi, _ = orc.nn_brute(Df, S, r2)
    tree = orc.KDTree(Df, use_ref=orc.ref_available())
    o1, o2, _ov = tree.find_correspondences(S, r2)
    oi = np.full(len(S), -1, np.int64); oi[o2] = o1
    differ = int(np.count_nonzero(bi != oi))
    assert got == want and got >= differ > 0, (got, want, differ)
    _report("tie_count.json", {"frame_1 vs frame_2, identity": {"tied_queries": got, "queries_where_the_two_tie_rules_differ": differ, "queries": int(len(S))}})


def test_warm_kernel_matches_index_for_index_10m(Context, orc):
    """BASELINE configs[2] at full size: the adaptive loop of the bench (tiles, then warm-started iterations) and the forced
    warm form; the last iteration's 10M matches against a fresh tiled search (every index, every d2 bit) and a 200k-query
    sample of the reference's nanoflann over the 10M-point target."""
    rng = np.random.default_rng(14)
    n = 10_000_000
    d = syn.make_pair(n, perturb=0.3)
    D, N, S, r2 = d["dst"], d["dst_n"], d["src"], d["max_sq_dist"]
    tree = orc.KDTree(D, use_ref=orc.ref_available())
    report = {}
    for name, opts, iters in (("adaptive, 6 iterations", (), 6), ("warm forced, 3 iterations", (("warm_start", 2),), 3)):
        gi, gd, T, ctx, nc = _loop_matches(Context, D, N, S, r2, iters, options=opts)
        warm = ctx.last_warm_iterations()
        assert warm >= 2, (name, warm)
        # fresh search under the same transform, same context (the tiled search-only kernel)
        ctx.find_correspondences(T, float(r2), count=False)
        fi, fd = ctx.get_nn()
        ctx.close()
        fi = _signed(fi)
        assert np.array_equal(gi, fi), (name, np.nonzero(gi != fi)[0][:10])
        assert np.array_equal(gd.view(np.uint32)[gi >= 0], fd.view(np.uint32)[gi >= 0]), name
        sample = np.sort(rng.choice(n, 200_000, replace=False))
        o1, o2, ov = tree.find_correspondences(orc.transform_points(T, S[sample]), float(r2))
        oi = np.full(len(sample), -1, np.int64); od = np.zeros(len(sample), np.float32)
        oi[o2] = o1; od[o2] = ov
        nbad, ties, nearer, worse = _classify(gi[sample], gd[sample], oi, od)
        assert nbad == 0, (name, nbad, ties, nearer, worse)
        assert np.array_equal(gd[sample].view(np.uint32)[oi >= 0], od.view(np.uint32)[oi >= 0]), name
        report[name] = {"queries": n, "matched": nc, "warm_iterations": warm, "sample": len(sample), "mismatches": nbad, "ties": ties, "nearer": nearer, "worse": worse,
                        "knn": "reference nanoflann" if tree.use_ref else "oracle kd-tree"}
    _report("warm_matches_10m.json", report)


