"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer / index work must be bit-exact; transforms within 1e-5 Frobenius
(BASELINE.json north_star)."""
import ctypes as C
import os

import numpy as np
import pytest

from cilantro_amd import capi
from cilantro_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL_T = 1e-5  # north_star: final rotation/translation within 1e-5 Frobenius


@pytest.fixture(scope="module")
def Context(hip_lib):
    from cilantro_amd.icp import Context as C

    return C


def classify_mismatches(orc, dst, q, gi, gd, oi, od):
    """exact / equal-distance tie / gpu strictly nearer / gpu worse (must be 0)"""
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


def _report(name, obj):
    """Measured figures of the full-size tests, kept next to the run (gpurun_out/ travels back from the GPU box)."""
    import json
    import os

    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass
    print(name, obj)


def _oracle_icp_loop(orc, tree, d, p, iters):
    """icp_base.hpp:68-87 driven from here so that the kNN half can be the reference's own nanoflann (oracle/_ref) over a
    tree built once: transformFeatures -> findCorrespondences -> updateEstimate (the oracle's estimator), `iters` times."""
    T = np.eye(4, dtype=np.float32)
    nc = 0
    for _ in range(iters):
        di, si, _ = tree.find_correspondences(orc.transform_points(T, d["src"]), float(d["max_sq_dist"]))
        T, _ = orc.icp_update(d["dst"], d["dst_n"], d["src"], T, di, si, p)
        nc = len(di)
    return T, nc


def gpu_nn(ctx, T, max_sq):
    ctx.find_correspondences(T, max_sq, count=False)
    idx, d2 = ctx.get_nn()
    gi = idx.astype(np.int64)
    gi[idx == capi.NONE_IDX] = -1
    return gi, d2


def test_kd_tree_example_known_answer(Context, orc):
    # examples/kd_tree.cpp:6-19 (k=2 there; the engine is k=1: nearest of the two is index 0, d2=0.18)
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1], [1, 1, 0], [1, 1, 1]], np.float32)
    ctx = Context()
    ctx.set_target(pts)
    ctx.set_source(np.array([[0.1, 0.1, 0.4]], np.float32))
    n = ctx.find_correspondences(np.eye(4), 1.001)
    i1, i2, v = ctx.get_correspondences()
    assert n == 1 and list(i1) == [0] and list(i2) == [0]
    oi, od = orc.KDTree(pts).knn_in_radius([0.1, 0.1, 0.4], 2, 1.001)
    assert v[0] == od[0] and oi[0] == 0 and oi[1] == 3


@pytest.mark.parametrize("n", [1, 7, 1000, 20000])
def test_nn_vs_bruteforce_exact(Context, orc, n):
    rng = np.random.default_rng(n)
    dst = rng.random((n, 3), dtype=np.float32)
    if n >= 1000:
        dst[n // 2: n // 2 + 50] = dst[:50]          # exact duplicates -> ties: lowest index under tie_rule 0, the reference's first-met point by default
    q = rng.random((max(n, 500), 3), dtype=np.float32) * 1.2 - 0.1   # some queries outside the bbox
    q[:20] = dst[:20] if n >= 20 else q[:20]          # zero-distance matches
    h = n ** (-1.0 / 3.0)
    tree = orc.KDTree(dst, use_ref=orc.ref_available())
    for max_sq in (np.float32((0.7 * h) ** 2), np.float32(0.05), np.float32(3.4e38)):
        # tie_rule 0: the brute-force argmin with the lowest index on ties
        ctx = Context()
        ctx.set_option("tie_rule", 0)
        ctx.set_target(dst)
        ctx.set_source(q)
        gi, gd = gpu_nn(ctx, np.eye(4), max_sq)
        ctx.close()
        bi, bd = orc.nn_brute(dst, q, max_sq)
        assert np.array_equal(gi, bi), (n, max_sq, np.nonzero(gi != bi)[0][:10])
        m = bi >= 0
        assert np.array_equal(gd[m], bd[m])
        # default options: the reference's kd-tree search, index for index (ties included)
        ctx = Context()
        ctx.set_target(dst)
        ctx.set_source(q)
        gi, gd = gpu_nn(ctx, np.eye(4), max_sq)
        ctx.close()
        o1, o2, ov = tree.find_correspondences(q, float(max_sq))
        oi = np.full(len(q), -1, np.int64); oi[o2] = o1
        assert np.array_equal(gi, oi), (n, max_sq, np.nonzero(gi != oi)[0][:10])
        assert np.array_equal(gd[oi >= 0].view(np.uint32), ov.view(np.uint32))


def test_radius_edge_is_strict(Context, orc):
    # d2 == r2 must be rejected (strict '<' at nanoflann.hpp:1901 and kd_tree_utilities.hpp:29)
    dst = np.array([[0, 0, 0], [4, 0, 0]], np.float32)
    q = np.array([[0.5, 0, 0], [2.0, 0, 0]], np.float32)
    ctx = Context()
    ctx.set_target(dst); ctx.set_source(q)
    gi, gd = gpu_nn(ctx, np.eye(4), np.float32(0.25))
    assert list(gi) == [-1, -1]
    gi, gd = gpu_nn(ctx, np.eye(4), np.nextafter(np.float32(0.25), np.float32(1)))
    assert list(gi) == [0, -1]
    gi, gd = gpu_nn(ctx, np.eye(4), np.float32(4.0))
    assert list(gi) == [0, -1]                         # q1 is at d2 == 4.0 from both -> rejected
    gi, gd = gpu_nn(ctx, np.eye(4), np.float32(4.5))
    assert list(gi) == [0, 0]                          # tie -> lowest index


def test_empty_inputs(Context):
    ctx = Context()
    ctx.set_target(np.zeros((0, 3), np.float32))
    ctx.set_source(np.random.default_rng(0).random((100, 3), dtype=np.float32))
    assert ctx.find_correspondences(np.eye(4), 1.0) == 0        # kd_tree_utilities.hpp:16-19
    ctx2 = Context()
    ctx2.set_target(np.random.default_rng(0).random((100, 3), dtype=np.float32))
    ctx2.set_source(np.zeros((0, 3), np.float32))
    assert ctx2.find_correspondences(np.eye(4), 1.0) == 0
    i1, i2, v = ctx2.get_correspondences()
    assert len(i1) == 0


@pytest.mark.parametrize("n", [200000, 1000000])
def test_nn_vs_kdtree_oracle_and_reference(Context, orc, n):
    d = syn.make_pair(n)
    T = np.eye(4, dtype=np.float32)
    q = orc.transform_points(T, d["src"])
    ctx = Context()
    ctx.set_target(d["dst"]); ctx.set_source(d["src"])
    ng = ctx.find_correspondences(T, d["max_sq_dist"])
    g1, g2, gv = ctx.get_correspondences()
    for use_ref in ([False, True] if orc.ref_available() else [False]):
        o1, o2, ov = orc.KDTree(d["dst"], use_ref=use_ref).find_correspondences(q, d["max_sq_dist"])
        assert ng == len(o1)
        assert np.array_equal(g2, o2)                  # same kept set, ascending source order
        nbad, ties, nearer, worse = classify_mismatches(orc, d["dst"], q, g1, gv, o1, ov)
        # uniform random data: no duplicates, so indices must match exactly; any deviation is classified
        assert worse == 0, (nbad, ties, nearer, worse)
        assert nbad == 0, (nbad, ties, nearer, worse)
        assert np.array_equal(gv, ov)


def test_accumulation_sums_vs_oracle(Context, orc):
    d = syn.make_pair(300000)
    T = syn.true_transform(d["h"], 0.1).astype(np.float32)
    ctx = Context()
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    ctx.find_correspondences(T, d["max_sq_dist"])
    g1, g2, gv = ctx.get_correspondences()
    q = orc.transform_points(T, d["src"])
    dm, sm = ctx.means()
    smt = orc.transform_points(T, sm.reshape(1, 3))[0]
    # point-to-point raw moments
    Tg, sums_g, ok = ctx.estimate_point_to_point()
    To, sums_o, ok2 = orc.estimate_p2p(d["dst"], q, g1, g2, orc.MODE_MIXED)
    assert ok == ok2
    np.testing.assert_allclose(sums_g, sums_o, rtol=1e-11, atol=1e-9)
    assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6
    # Gauss-Newton normal equations, all three weightings
    for w_p2p, w_p2pl in ((0.0, 1.0), (1.0, 0.0), (0.1, 1.0)):
        Tg, AtA, Atb, cv = ctx.estimate_combined(w_p2p, w_p2pl, 1, 1e-5)
        To, AtAo, Atbo, cvo = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, w_p2p, w_p2pl, dm, smt,
                                                    1, 1e-5, orc.MODE_MIXED)
        scale = np.abs(AtAo).max()
        assert np.abs(AtA - AtAo).max() <= 1e-9 * scale, (w_p2p, w_p2pl, np.abs(AtA - AtAo).max() / scale)
        assert np.abs(Atb - Atbo).max() <= 1e-9 * max(np.abs(Atbo).max(), 1e-30) + 1e-12 * scale
        assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6, (w_p2p, w_p2pl)
        assert cv == cvo
        # reference-like all-f32 arithmetic agrees loosely (f32 accumulation of 3e5 terms)
        Tf, _, _, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, w_p2p, w_p2pl, dm, smt, 1, 1e-5, orc.MODE_F32)
        assert np.linalg.norm(Tg - Tf) < 1e-4
    # multi-step Gauss-Newton (max_optimization_iterations_ = 3)
    Tg, _, _, cv = ctx.estimate_combined(0.0, 1.0, 3, 1e-7)
    To, _, _, cvo = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, 0.0, 1.0, dm, smt, 3, 1e-7, orc.MODE_MIXED)
    assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6 and cv == cvo


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n", [20000, 500000])
def test_icp_end_to_end_vs_oracle(orc, hip_lib, metric, n):
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    d = syn.make_pair(n, perturb=0.6)
    for max_iter, tol in ((6, 0.0), (50, 1e-5)):       # fixed iteration count AND convergence-gated
        if metric == 1:
            icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        else:
            icp = SimplePointToPointMetricRigidICP3f(d["dst"], d["src"])
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        icp.setMaxNumberOfIterations(max_iter).setConvergenceTolerance(tol)
        Tg = icp.estimate().getTransform()
        for mode in (orc.MODE_MIXED, orc.MODE_F32):
            p = orc.make_params(metric=metric, max_iter=max_iter, conv_tol=tol, max_sq_dist=d["max_sq_dist"], mode=mode)
            r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
            err = np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64))
            assert err <= TOL_T, (metric, n, max_iter, mode, err)
            if mode == orc.MODE_MIXED:
                assert abs(icp.getNumberOfPerformedIterations() - r["iterations"]) <= (0 if tol == 0.0 else 1)
                if tol == 0.0:
                    assert icp.last_ncorr_ == r["last_ncorr"]
        assert np.linalg.norm(Tg - d["T_true"]) < 2e-3 * (20000 / n) ** 0.5 + 2e-4
        if tol > 0:
            assert icp.hasConverged()


def test_every_accumulating_form_with_half_of_the_matches_cut_by_the_radius(orc, hip_lib):
    """Every kernel that forms the estimators' terms itself (k_warm, the accumulating tiles, the streaming passes, k_warm<IM_AFFC / IM_AFFP>,
    k_acc_affine) with a radius that leaves about half of the queries WITHOUT a correspondence -- lanes that contribute zeros sit next to
    lanes that contribute terms in every wave, in every accumulation form (Kabsch moments, plane terms, point terms, both; the affine
    moments).  Transforms and correspondence counts against the oracle's loops.  (Round 6: a back-end miscompile dropped one component of
    the plane-terms vector on a path with such lanes -- NOTEBOOK; this is the test that would have seen it in any of these kernels.)"""
    from cilantro_amd.icp import (SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f, SimpleCombinedMetricAffineICP3f,
                                  SimplePointToPointMetricAffineICP3f)

    n = 200_000
    d = syn.make_pair(n, perturb=0.1, noise=0.3)
    # nearest distances of this pair are spread over (0, ~0.5 h): a radius at their median cuts half of them
    ctx_probe = SimplePointToPointMetricRigidICP3f(d["dst"], d["src"])
    ctx_probe.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
    ctx_probe._ctx.find_correspondences(d["T_true"].astype(np.float32), float(d["max_sq_dist"]), count=False)
    dv = ctx_probe._ctx.get_correspondences()[2]
    r2 = float(np.median(dv))
    forms = (("adaptive", {}), ("warm from the second iteration", {"warm_start": 2}), ("tiles, accumulating", {"tiled": 2, "tile_accumulation": 2, "warm_start": 0}),
             ("streaming passes", {"tiled": 0, "warm_start": 0, "tile_accumulation": 0}))
    T0 = d["T_true"].astype(np.float32).copy()
    T0[:3, 3] += np.float32(0.05 * d["h"])
    for affine in (False, True):
        for metric, wts in ((0, None), (1, (0.0, 1.0)), (1, (1.0, 0.0)), (1, (0.1, 1.0))):
            p = orc.make_params(metric=metric, w_p2p=wts[0] if wts else 0.0, w_p2pl=wts[1] if wts else 1.0, max_iter=6, conv_tol=0.0, max_sq_dist=r2,
                                mode=orc.MODE_MIXED, affine=affine)
            ro = orc.icp_run(d["dst"], d["dst_n"], d["src"], p, T0=T0)
            assert 0.25 * n < ro["last_ncorr"] < 0.75 * n, ro["last_ncorr"]
            for name, opts in forms:
                if metric == 1:
                    icp = (SimpleCombinedMetricAffineICP3f if affine else SimpleCombinedMetricRigidICP3f)(d["dst"], d["dst_n"], d["src"])
                    icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1])
                else:
                    icp = (SimplePointToPointMetricAffineICP3f if affine else SimplePointToPointMetricRigidICP3f)(d["dst"], d["src"])
                for k, v in opts.items():
                    icp._ctx.set_option(k, v)
                icp.correspondenceSearchEngine().setMaxDistance(r2)
                icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).setInitialTransform(T0)
                Tg = icp.estimate().getTransform()
                err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
                assert err <= (3e-5 if affine else TOL_T), (affine, metric, wts, name, err)
                # (a query whose distance sits within rounding of the radius may fall on either side under transforms that differ in their last bits)
                assert abs(icp.last_ncorr_ - ro["last_ncorr"]) <= 5, (affine, metric, wts, name, icp.last_ncorr_, ro["last_ncorr"])


def test_symmetric_metric_warm_started_vs_streaming_pass_and_oracle(orc, hip_lib):
    """The symmetric objective (four-cloud constructor; transform_estimation.hpp:479-744 through accumulate_pair's sym branch) in the
    warm-started form (k_warm<., ., SYM>: the source normals streamed with the queries, n = n_dst + R n_src formed as the streaming pass
    forms it) against the loop without warm start (search + streaming pass every iteration) and against the oracle's symmetric loop:
    same iterations and counts, transforms to the order of the f64 additions; with both terms and plane terms only; a radius that cuts
    matches; and the warm-started form really ran."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 200_000
    d = syn.make_pair(n, n, with_normals=True, perturb=0.4)
    rng = np.random.default_rng(9)
    src_n = d["dst_n"] + 0.05 * rng.normal(size=d["dst_n"].shape).astype(np.float32)
    src_n = np.ascontiguousarray((src_n / np.linalg.norm(src_n, axis=1, keepdims=True)).astype(np.float32))
    for wts in ((0.0, 1.0), (0.2, 1.0)):
        for r2 in (float(d["max_sq_dist"]), float((0.12 * d["h"]) ** 2)):
            p = orc.make_params(metric=1, w_p2p=wts[0], w_p2pl=wts[1], max_iter=10, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED)
            ro = orc.icp_run(d["dst"], d["dst_n"], d["src"], p, src_n=src_n)
            got = []
            for warm in (1, 2, 0):
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"], src_n)
                icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1])
                icp._ctx.set_option("warm_start", warm)
                icp.correspondenceSearchEngine().setMaxDistance(r2)
                icp.setMaxNumberOfIterations(10).setConvergenceTolerance(0.0)
                T = icp.estimate().getTransform().astype(np.float64)
                got.append((T, icp.getNumberOfPerformedIterations(), icp.last_ncorr_, icp._ctx.last_warm_iterations()))
                assert np.linalg.norm(T - ro["T"].astype(np.float64)) <= TOL_T, (wts, r2, warm, np.linalg.norm(T - ro["T"]))
                assert abs(icp.last_ncorr_ - ro["last_ncorr"]) <= 3, (wts, r2, warm, icp.last_ncorr_, ro["last_ncorr"])
            assert got[1][3] >= 8 and got[2][3] == 0, [g[3] for g in got]      # warm_start 2: from the second iteration on; 0: never
            for g in got[:2]:
                assert g[1] == got[2][1] and abs(g[2] - got[2][2]) <= 3 and np.abs(g[0] - got[2][0]).max() < 2e-6, (wts, r2)


def test_icp_combined_weights_and_gn_steps(orc, hip_lib):
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(100000, perturb=0.5)
    for w_p2p, w_p2pl, steps in ((0.1, 1.0, 1), (1.0, 0.0, 1), (0.0, 1.0, 3), (0.5, 0.5, 2)):
        icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        icp.setPointToPointMetricWeight(w_p2p).setPointToPlaneMetricWeight(w_p2pl)
        icp.setMaxNumberOfOptimizationStepIterations(steps).setOptimizationStepConvergenceTolerance(1e-6)
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=w_p2p, w_p2pl=w_p2pl, max_iter=8, conv_tol=0.0, max_opt_iter=steps,
                            opt_conv_tol=1e-6, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
        r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
        err = np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64))
        assert err <= TOL_T, (w_p2p, w_p2pl, steps, err)
    # max_optimization_iterations = 0: the estimator's loop body never runs, tform = t_dst * I * t_src (transform_estimation.hpp:281,
    # :365) -- every outer iteration translates the source's mean onto the target's; tiled and per-lane forms, the pair-list directions
    from cilantro_amd.icp import CorrespondenceSearchDirection as D
    dz = syn.make_pair(1_200_000, perturb=0.5)
    for dd, opts, direction in ((d, (), D.SECOND_TO_FIRST), (dz, (("tiled", 2),), D.SECOND_TO_FIRST), (d, (), D.BOTH), (d, (), D.FIRST_TO_SECOND)):
        icp = SimpleCombinedMetricRigidICP3f(dd["dst"], dd["dst_n"], dd["src"])
        for k, v in opts:
            icp._ctx.set_option(k, v)
        icp.setMaxNumberOfOptimizationStepIterations(0)
        icp.correspondenceSearchEngine().setMaxDistance(dd["max_sq_dist"]).setSearchDirection(direction)
        icp.setMaxNumberOfIterations(3).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, max_iter=3, conv_tol=0.0, max_opt_iter=0, max_sq_dist=dd["max_sq_dist"], mode=orc.MODE_MIXED,
                            direction={D.SECOND_TO_FIRST: 0, D.FIRST_TO_SECOND: 1, D.BOTH: 2}[direction])
        r = orc.icp_run(dd["dst"], dd["dst_n"], dd["src"], p)
        err = np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64))
        assert err <= TOL_T and icp.getNumberOfPerformedIterations() == 3 and icp.last_ncorr_ == r["last_ncorr"], (direction, err)
        assert np.array_equal(Tg[:3, :3], np.eye(3, dtype=np.float32))            # a pure translation


def test_icp_degenerate_inputs(hip_lib):
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    rng = np.random.default_rng(1)
    dst = rng.random((1000, 3), dtype=np.float32)
    far = dst + np.float32(100.0)                       # no correspondences at all
    icp = SimplePointToPointMetricRigidICP3f(dst, far)
    T = icp.estimate().getTransform()
    # transform_estimation.hpp:20-23: identity step, delta 0 -> "converged" after one iteration
    assert np.array_equal(T, np.eye(4, dtype=np.float32)) and icp.getNumberOfPerformedIterations() == 1
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (1000, 1))
    icp = SimpleCombinedMetricRigidICP3f(dst, nrm, far)
    T = icp.estimate().getTransform()
    assert np.array_equal(T, np.eye(4, dtype=np.float32)) and icp.getNumberOfPerformedIterations() == 1
    # a plane term without target normals: dst_p.cols() != dst_n.cols() in the reference (transform_estimation.hpp:264-272)
    # -> identity step, "converged" after one iteration -- with correspondences present, in the loop, through the raw
    # sharded entry points, with the per-lane and the LDS-tiled search, and with the combined weights
    d = syn.make_pair(30000)
    for tiled in (0, 2):
        for w_p2p in (0.0, 0.3):
            icp = SimpleCombinedMetricRigidICP3f(d["dst"], None, d["src"])
            icp._ctx.set_option("tiled", tiled)
            icp.setPointToPointMetricWeight(w_p2p).setPointToPlaneMetricWeight(1.0)
            icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
            T = icp.setMaxNumberOfIterations(5).estimate().getTransform()
            assert np.array_equal(T, np.eye(4, dtype=np.float32)) and icp.getNumberOfPerformedIterations() == 1
            assert icp.last_ncorr_ > 0
    import torch

    from cilantro_amd import distributed
    eng = distributed.HipShardEngine(d["dst"], None, d["src"], 0)
    r = distributed.ShardedRigidICP(eng).estimate(distributed.default_params(max_iter=3, max_sq_dist=float(d["max_sq_dist"])))
    assert np.array_equal(r[0], np.eye(4, dtype=np.float32)) and r[1] == 1
    torch.cuda.synchronize()


def test_run_to_run_bitwise_reproducible(hip_lib):
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(300000)
    outs = []
    for _ in range(3):
        icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
        outs.append(icp.estimate().getTransform().tobytes())
    assert outs[0] == outs[1] == outs[2]


def test_residuals_vs_oracle(orc, hip_lib):
    # computeResiduals(): icp_single_transform_combined_metric.hpp:220-243 / point_to_point_metric.hpp:68-85
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    d = syn.make_pair(50000, perturb=0.4)
    f = np.float32
    for metric in (0, 1):
        icp = (SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"]) if metric else
               SimplePointToPointMetricRigidICP3f(d["dst"], d["src"]))
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        if metric:
            icp.setPointToPointMetricWeight(0.25)
        T = icp.estimate().getTransform()
        res = icp.getResiduals()
        q = orc.transform_points(T, d["src"])
        bi, _ = orc.nn_brute(d["dst"], q, np.float32(3.4e38))
        p = d["dst"][bi]
        dx, dy, dz = p[:, 0] - q[:, 0], p[:, 1] - q[:, 1], p[:, 2] - q[:, 2]
        sq = dx * dx + (dy * dy + dz * dz)
        if metric == 0:
            exp = sq
        else:
            n = d["dst_n"][bi]
            pd = n[:, 0] * dx + (n[:, 1] * dy + n[:, 2] * dz)
            exp = f(0.25) * sq + (f(1.0) * pd) * pd
        assert np.array_equal(res, exp.astype(np.float32))


def test_torch_device_inputs_and_shard_engine(orc, hip_lib):
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(200000, perturb=0.5)
    dst, nrm, src = (torch.from_numpy(d[k]).cuda() for k in ("dst", "dst_n", "src"))
    icp = SimpleCombinedMetricRigidICP3f(dst, nrm, src, stream=torch.cuda.current_stream().cuda_stream)
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    T1 = icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).estimate().getTransform()
    # the sharded protocol with one rank must reproduce icp_run bit for bit (same kernels, same order)
    eng = distributed.HipShardEngine(dst, nrm, src, 0)
    p = distributed.default_params(max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    T2, iters, delta, nc = distributed.ShardedRigidICP(eng, None).estimate(p)
    assert iters == 6 and np.array_equal(T1, T2) and nc == icp.last_ncorr_
    # fused and unfused kernels agree to f64 summation-order accuracy
    icp._ctx.set_option("fused", 1)
    T3 = icp.estimate().getTransform()
    assert np.abs(T3 - T1).max() <= 1e-7


@pytest.mark.parametrize("n", [20000, 1000000])
def test_tiled_search_kernel_is_exact_too(Context, orc, n):
    """The optional LDS-tiled search kernel ("tiled"=1) must give bit-identical matches."""
    d = syn.make_pair(n, perturb=0.8)
    T = syn.true_transform(d["h"], 0.5).astype(np.float32)
    outs = []
    for tiled in (0, 2):                                # 2 = force the tiled kernel whatever the cloud size
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(d["dst"]); ctx.set_source(d["src"])
        # sort under identity, search under T: queries have drifted from their sort-time cells
        ctx.find_correspondences(np.eye(4), d["max_sq_dist"], count=False)
        for max_sq in (d["max_sq_dist"], np.float32(3.0e38), np.float32((0.3 * d["h"]) ** 2)):
            outs.append((tiled, gpu_nn(ctx, T, max_sq)))
    half = len(outs) // 2
    for (t0, (i0, d0)), (t1, (i1, d1)) in zip(outs[:half], outs[half:]):
        assert np.array_equal(i0, i1)
        m = i0 >= 0
        assert np.array_equal(d0[m], d1[m])
    if n <= 20000:
        bi, bd = orc.nn_brute(d["dst"], orc.transform_points(T, d["src"]), d["max_sq_dist"])
        assert np.array_equal(outs[0][1][0], bi)


def _bumpy_surface(n, seed=7):
    """Non-uniform (2-D manifold) cloud: a bumpy height field + its analytic normals; well conditioned for ICP."""
    rng = np.random.default_rng(seed)
    x = rng.random(n) * 2 - 1
    y = rng.random(n) * 2 - 1
    z = 0.15 * np.sin(3 * x) * np.cos(4 * y) + 0.05 * np.sin(11 * x + 1.0) * np.sin(9 * y)
    dzdx = 0.45 * np.cos(3 * x) * np.cos(4 * y) + 0.55 * np.cos(11 * x + 1.0) * np.sin(9 * y)
    dzdy = -0.6 * np.sin(3 * x) * np.sin(4 * y) + 0.45 * np.sin(11 * x + 1.0) * np.cos(9 * y)
    nrm = np.stack([-dzdx, -dzdy, np.ones(n)], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.stack([x, y, z], 1).astype(np.float32), nrm.astype(np.float32)


def test_surface_cloud_adaptive_grid_parity(Context, orc, hip_lib):
    """Surface-like data: most grid cells are empty, the adaptive cell sizing and the generic shell search
    (radius >> point spacing, as examples/rigid_icp.cpp:122 uses) are exercised."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    n = 150000
    dst, nrm = _bumpy_surface(n)
    T_true = np.eye(4)
    T_true[:3, :3] = syn.rot_xyz(0.02, -0.015, 0.01)
    T_true[:3, 3] = [0.01, -0.008, 0.012]
    Ti = np.linalg.inv(T_true)
    rng = np.random.default_rng(11)
    src = ((dst.astype(np.float64) + rng.normal(0, 3e-4, dst.shape)) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    src = src[rng.random(n) < 0.7]                              # partial overlap in count
    max_sq = np.float32(0.1 * 0.1)                               # radius ~ 25x the point spacing
    # kNN parity with mismatch classification (ties are legal, "worse" is not)
    ctx = Context()
    ctx.set_target(dst, nrm); ctx.set_source(src)
    gi = ctx.grid_info()
    assert gi.avg_occupancy < 40, gi.avg_occupancy               # adaptive sizing kept cells small
    g_idx, g_d2 = gpu_nn(ctx, np.eye(4), max_sq)
    q = orc.transform_points(np.eye(4), src)
    di, si, dv = orc.KDTree(dst).find_correspondences(q, max_sq)
    o_idx = np.full(len(src), -1, np.int64); o_idx[si] = di
    o_d2 = np.zeros(len(src), np.float32); o_d2[si] = dv
    nbad, ties, nearer, worse = classify_mismatches(orc, dst, q, g_idx, g_d2, o_idx, o_d2)
    assert worse == 0 and nearer == 0 and nbad == ties and ties <= 5, (nbad, ties, nearer, worse)
    bi, bd = orc.nn_brute(dst, q[:20000], max_sq)
    assert np.array_equal(g_idx[:20000], bi)                     # == brute force incl. lowest-index ties
    # end-to-end
    for metric in (1, 0):
        icp = (SimpleCombinedMetricRigidICP3f(dst, nrm, src) if metric else SimplePointToPointMetricRigidICP3f(dst, src))
        icp.correspondenceSearchEngine().setMaxDistance(max_sq)
        icp.setMaxNumberOfIterations(40).setConvergenceTolerance(1e-6)
        T = icp.estimate().getTransform()
        p = orc.make_params(metric=metric, max_iter=40, conv_tol=1e-6, max_sq_dist=max_sq, mode=orc.MODE_MIXED)
        r = orc.icp_run(dst, nrm, src, p)
        assert np.linalg.norm(T.astype(np.float64) - r["T"]) <= TOL_T, (metric, np.linalg.norm(T - r["T"]))
        assert np.linalg.norm(T - T_true) < 2e-3


def test_symmetric_metric_vs_oracle(orc, hip_lib):
    """Four-cloud constructor => estimateTransformSymmetricMetric (transform_estimation.hpp:604-739) and the
    residual with the added source normal (icp_single_transform_combined_metric.hpp:237)."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(120000, perturb=0.5)
    Ri = np.linalg.inv(d["T_true"])[:3, :3]
    src_n = (d["dst_n"].astype(np.float64) @ Ri.T).astype(np.float32)     # normals of the source points
    for w_p2p, steps in ((0.0, 1), (0.2, 2)):
        icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"], src_n)
        icp.setPointToPointMetricWeight(w_p2p).setMaxNumberOfOptimizationStepIterations(steps)
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=w_p2p, w_p2pl=1.0, max_iter=8, conv_tol=0.0, max_opt_iter=steps,
                            max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
        r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p, src_n=src_n)
        assert np.linalg.norm(Tg.astype(np.float64) - r["T"]) <= TOL_T, (w_p2p, steps)
        # differs from the non-symmetric result (so the source normals are really used) yet converges to the truth
        r3 = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
        assert np.abs(r3["T"] - r["T"]).max() > 0
        assert np.linalg.norm(Tg - d["T_true"]) < 1e-3
    res = icp.getResiduals()
    q = orc.transform_points(Tg, d["src"])
    bi, _ = orc.nn_brute(d["dst"], q[:5000], np.float32(3.4e38))
    p_ = d["dst"][bi]
    dx, dy, dz = p_[:, 0] - q[:5000, 0], p_[:, 1] - q[:5000, 1], p_[:, 2] - q[:5000, 2]
    sq = dx * dx + (dy * dy + dz * dz)
    n = d["dst_n"][bi] + src_n[:5000]
    pd = n[:, 0] * dx + (n[:, 1] * dy + n[:, 2] * dz)
    exp = np.float32(0.2) * sq + (np.float32(1.0) * pd) * pd
    assert np.array_equal(res[:5000], exp.astype(np.float32))


def test_target_shard_protocol_single_rank(orc, hip_lib):
    """Target-sharded building blocks with one rank == the plain loop; and with the shard given a global
    index offset the published keys carry global indices."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(150000, perturb=0.5)
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    T1 = icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).estimate().getTransform()
    dm, _ = icp._ctx.means()
    off = 1000000
    eng = distributed.HipTargetShardEngine(d["dst"], d["dst_n"], d["src"], off, dm, 0)
    p = distributed.default_params(max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    T2, iters, delta, nc = distributed.TargetShardedRigidICP(eng, None).estimate(p)
    assert iters == 6 and nc == icp.last_ncorr_
    assert np.array_equal(T1, T2)
    # keys of a fresh search under T2: (d2 bits << 32) | (local index + offset), original source order
    eng.begin(p, T2)
    keys = eng.partial_keys().cpu().numpy()
    q = orc.transform_points(T2, d["src"])
    bi, bd = orc.nn_brute(d["dst"], q[:20000], d["max_sq_dist"])
    exp = np.where(bi >= 0, (bd.view(np.uint32).astype(np.int64) << 32) | (bi + off), distributed.KEY_NONE)
    assert np.array_equal(keys[:20000], exp)


def test_engine_post_filters_vs_oracle(Context, orc, hip_lib):
    """setInlierFraction / setOneToOne (correspondence_search_kd_tree.hpp:224-225, core/correspondence.hpp:57-100):
    identical surviving set AND identical order as the reference leaves it (ties pinned to lowest source index)."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(60000, perturb=0.6)
    dst = d["dst"][::2].copy(); nrm = d["dst_n"][::2].copy()       # 2 sources per target -> one-to-one really drops
    src = d["src"].copy()
    src[1000:1100] = src[:100]                                       # duplicated sources -> exact value ties
    T = np.eye(4, dtype=np.float32)
    q = orc.transform_points(T, src)
    base = orc.KDTree(dst).find_correspondences(q, d["max_sq_dist"])
    for frac, o2o in ((0.6, False), (1.0, True), (0.35, True), (0.999999, False), (1e-9, False)):
        ctx = Context()
        ctx.set_target(dst, nrm); ctx.set_source(src)
        ctx.set_option("inlier_fraction", frac); ctx.set_option("one_to_one", 1 if o2o else 0)
        n = ctx.find_correspondences(T, d["max_sq_dist"])
        g1, g2, gv = ctx.get_correspondences()
        o1, o2, ov = orc.filter_fraction(*base, frac)
        if o2o:
            o1, o2, ov = orc.filter_one_to_one(o1, o2, ov)
        assert n == len(o1), (frac, o2o, n, len(o1))
        assert np.array_equal(g1, o1) and np.array_equal(g2, o2) and np.array_equal(gv, ov), (frac, o2o)
    # inside the ICP loop
    for frac, o2o in ((0.8, False), (0.9, True)):
        icp = SimpleCombinedMetricRigidICP3f(dst, nrm, src)
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"]).setInlierFraction(frac).setOneToOne(o2o)
        Tg = icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).estimate().getTransform()
        p = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED,
                            inlier_fraction=frac, one_to_one=o2o)
        r = orc.icp_run(dst, nrm, src, p)
        assert icp.last_ncorr_ == r["last_ncorr"]
        assert np.linalg.norm(Tg.astype(np.float64) - r["T"]) <= TOL_T, (frac, o2o)


def _kmeans_label_mismatches_are_near_ties(x, lab_g, lab_o, centroids, k):
    """After several Lloyd steps the two sides' centroids differ in their last bits (the GPU sums exactly in fixed point,
    the reference serially in f32), so a point that sits on a cell boundary of the Voronoi diagram can flip.  Every
    mismatch must be such a point -- its two candidate centroids equidistant to 1e-5 relative -- and there are few."""
    bad = np.nonzero(lab_g != lab_o)[0]
    assert len(bad) <= max(1, int(1e-5 * len(x))), (k, len(bad))
    for i in bad:
        dg = float(((x[i].astype(np.float64) - centroids[lab_g[i]]) ** 2).sum())
        do = float(((x[i].astype(np.float64) - centroids[lab_o[i]]) ** 2).sum())
        assert abs(dg - do) <= 1e-5 * max(dg, do), (k, int(i), dg, do)


def test_kmeans3f_vs_oracle(orc, hip_lib):
    """KMeans<float,3> brute-force path (clustering/kmeans.hpp:67-194)."""
    from cilantro_amd.clustering import KMeans3f, kmeans_assign

    rng = np.random.default_rng(5)
    # clustered data (mixture) so that k-means has structure; explicit initial centroids (no random_device)
    centres = rng.random((40, 3)).astype(np.float32)
    x = (centres[rng.integers(0, 40, 300000)] + rng.normal(0, 0.03, (300000, 3))).astype(np.float32)
    for k in (1, 7, 64, 1024):
        c0 = x[:k].copy()
        # one assignment pass: labels bit-exact given identical centroids
        lab_g = kmeans_assign(x, c0)
        lab_o, _ = orc.kmeans_assign(x, c0)
        assert np.array_equal(lab_g, lab_o), k
    for k, iters, tol in ((64, 12, 0.0), (257, 6, 0.0), (64, 100, 1e-4)):
        c0 = x[:k].copy()
        km = KMeans3f(x).cluster(c0, max_iter=iters, tol=tol)
        co, lo, ito = orc.kmeans(x, c0, max_iter=iters, tol=tol, mode=1)
        assert km.getNumberOfPerformedIterations() == ito, (k, km.getNumberOfPerformedIterations(), ito)
        assert np.abs(km.getClusterCentroids() - co).max() <= 1e-6, k
        _kmeans_label_mismatches_are_near_ties(x, km.getPointToClusterIndexMap(), lo, co, k)
        groups = km.getClusterToPointIndicesMap()
        assert len(groups) == k and sum(len(g) for g in groups) == len(x)
    # empty-cluster repair (kmeans.hpp:134-176): a far-away initial centroid attracts nothing
    c0 = x[:8].copy(); c0[5] = [50.0, 50.0, 50.0]
    km = KMeans3f(x).cluster(c0, max_iter=3, tol=0.0)
    co, lo, ito = orc.kmeans(x, c0, max_iter=3, tol=0.0, mode=1)
    assert np.abs(km.getClusterCentroids() - co).max() <= 1e-6
    _kmeans_label_mismatches_are_near_ties(x, km.getPointToClusterIndexMap(), lo, co, 8)
    # use_kd_tree = true (kmeans.hpp:86-94, the mode examples/kmeans.cpp runs): a kd-tree over the centroids finds the same nearest
    # centroid, compared by nanoflann's rounding of the distance -- labels of one pass bit-exact against the oracle's kd-tree
    # restatement (pinned on the reference's nanoflann), whole runs iteration for iteration
    for k in (1, 7, 64, 1024):
        c0 = x[:k].copy()
        lab_g = kmeans_assign(x, c0, use_kd_tree=True)
        lab_o, _ = orc.kmeans_assign(x, c0, use_kd_tree=True)
        bad = np.nonzero(lab_g != lab_o)[0]
        for i in bad:      # only exactly equidistant centroids may differ (lowest index here, first met in the tree there)
            dg = orc.nn_brute(c0[[lab_g[i]]], x[[i]], 3.0e38)[1][0]; do = orc.nn_brute(c0[[lab_o[i]]], x[[i]], 3.0e38)[1][0]
            assert dg == do, (k, int(i), dg, do)
        assert len(bad) <= 2, (k, len(bad))
    for k, iters, tol in ((64, 12, 0.0), (64, 100, 1e-4)):
        c0 = x[:k].copy()
        km = KMeans3f(x).cluster(c0, max_iter=iters, tol=tol, use_kd_tree=True)
        co, lo, ito = orc.kmeans(x, c0, max_iter=iters, tol=tol, mode=1, use_kd_tree=True)
        assert km.getNumberOfPerformedIterations() == ito, (k, km.getNumberOfPerformedIterations(), ito)
        assert np.abs(km.getClusterCentroids() - co).max() <= 1e-6, k
        _kmeans_label_mismatches_are_near_ties(x, km.getPointToClusterIndexMap(), lo, co, k)


def test_kmeans_pruned_assignment_is_the_exhaustive_one(orc, hip_lib):
    """The pruned assignment pass (centroid grid in LDS, 3x3x3 / 5x5x5 block with a proof, all k centroids otherwise) against the
    exhaustive pass of the same library and against the oracle: labels identical, element for element -- uniform and clustered
    centroids (the grid's cells mostly empty: the fallbacks), duplicated centroids (equal distances: the lowest index wins), points far
    outside the centroids' box, non-finite points, k from 64 (the grid's floor) to 2048; whole Lloyd runs bit for bit."""
    from cilantro_amd import clustering
    from cilantro_amd.clustering import KMeans3f, kmeans_assign

    rng = np.random.default_rng(17)
    n = 400_000
    uni = rng.random((n, 3), dtype=np.float32)
    centres = rng.random((25, 3)).astype(np.float32)
    clu = (centres[rng.integers(0, 25, n)] + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
    far = uni.copy(); far[: n // 10] = (far[: n // 10] - 0.5) * 40.0      # a tenth of the points far outside
    bad = uni.copy(); bad[5] = [np.nan, 0.1, 0.2]; bad[77] = [np.inf, 0.5, 0.5]; bad[1234] = [0.3, -np.inf, 0.9]
    cases = []
    for k in (64, 257, 1024, 2048):
        cases.append((f"uniform k={k}", uni, uni[:k].copy()))
    cases.append(("clustered data, centroids from it", clu, clu[:1024].copy()))
    cases.append(("uniform data, clustered centroids", uni, clu[:1024].copy()))
    dupc = uni[:512].copy(); dupc[256:] = dupc[:256]      # every centroid twice: exact ties between an index and index + 256
    cases.append(("duplicated centroids", uni, dupc))
    cases.append(("points far outside the centroids' box", far, uni[:1024].copy()))
    cases.append(("non-finite points", bad, uni[:300].copy()))
    flat = uni[:700].copy(); flat[:, 2] = 0.25      # centroids on a plane: a degenerate extent along z
    cases.append(("coplanar centroids", uni, flat))
    try:
        for name, x, c0 in cases:
            clustering.set_pruning(True)
            lp = kmeans_assign(x, c0)
            clustering.set_pruning(False)
            le = kmeans_assign(x, c0)
            assert np.array_equal(lp, le), (name, int(np.count_nonzero(lp != le)))
            if "non-finite" not in name:
                lo, _ = orc.kmeans_assign(x, c0)
                assert np.array_equal(lp, lo), (name, int(np.count_nonzero(lp != lo)))
            # the kd branch (use_kd_tree: nanoflann's rounding of the distance, the tree's choice among equal distances) through the
            # same grid: pruned == exhaustive, label for label -- also with duplicated centroids (exact ties: tables on demand)
            clustering.set_pruning(True)
            kp = kmeans_assign(x, c0, use_kd_tree=True)
            clustering.set_pruning(False)
            ke = kmeans_assign(x, c0, use_kd_tree=True)
            assert np.array_equal(kp, ke), (name, "kd", int(np.count_nonzero(kp != ke)))
        for name, x, c0 in (cases[2], cases[4], cases[7]):
            clustering.set_pruning(True)
            kp = KMeans3f(x).cluster(c0.copy(), max_iter=8, tol=0.0)
            clustering.set_pruning(False)
            ke = KMeans3f(x).cluster(c0.copy(), max_iter=8, tol=0.0)
            assert kp.getNumberOfPerformedIterations() == ke.getNumberOfPerformedIterations(), name
            assert np.array_equal(kp.getClusterCentroids().view(np.uint32), ke.getClusterCentroids().view(np.uint32)), name
            assert np.array_equal(kp.getPointToClusterIndexMap(), ke.getPointToClusterIndexMap()), name
            clustering.set_pruning(True)
            kp = KMeans3f(x).cluster(c0.copy(), max_iter=8, tol=0.0, use_kd_tree=True)
            clustering.set_pruning(False)
            ke = KMeans3f(x).cluster(c0.copy(), max_iter=8, tol=0.0, use_kd_tree=True)
            assert kp.getNumberOfPerformedIterations() == ke.getNumberOfPerformedIterations(), (name, "kd")
            assert np.array_equal(kp.getClusterCentroids().view(np.uint32), ke.getClusterCentroids().view(np.uint32)), (name, "kd")
            assert np.array_equal(kp.getPointToClusterIndexMap(), ke.getPointToClusterIndexMap()), (name, "kd")
    finally:
        clustering.set_pruning(True)


def _plane_cloud(n, seed, inlier_frac=0.6, noise=0.004):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    k = int(n * inlier_frac)
    sel = rng.permutation(n)[:k]
    x[sel, 2] = (0.3 * x[sel, 0] - 0.2 * x[sel, 1] + 0.1 + rng.normal(0, noise, k)).astype(np.float32)
    return x, rng


def test_plane_ransac3f_vs_oracle(orc, hip_lib):
    """PlaneRANSACEstimator3f (model_estimation/ransac_base.hpp:64-131,
    ransac_hyperplane_estimator.hpp).  Scoring half bit-exact; model half within f32 round-off."""
    from cilantro_amd.model_estimation import PlaneRANSACEstimator3f

    n = 250_003   # not a multiple of the tile: exercises the ragged tail
    x, rng = _plane_cloud(n, 21)
    thr = 0.01
    # (1) scoring: counts of arbitrary planes are exact (same f32 expression, no contraction)
    planes = rng.standard_normal((301, 4)).astype(np.float32)
    planes[:, :3] /= np.linalg.norm(planes[:, :3], axis=1, keepdims=True)
    planes[:, 3] *= 0.2
    planes[7] = [np.nan, 0, 0, 0]
    pe = PlaneRANSACEstimator3f(x)
    got = pe.countInliers(planes, thr)
    want = np.array([orc.plane_count_inliers(x, p, thr) for p in planes])
    assert np.array_equal(got, want)
    # (2) whole-cloud PCA fit vs the oracle (f64 accumulation both sides; order differs -> last-bit only)
    pg = pe.estimateModel()
    po = orc.plane_fit(x, None, mode=1)
    s = 1.0 if np.dot(pg[:3], po[:3]) > 0 else -1.0
    assert np.abs(pg - s * po).max() <= 2e-6, (pg, po)
    # (3) full runs with explicit samples: same iteration count, same winner, same inlier set
    for max_iter, target, re_est in ((100, n // 2, True), (100, n // 2, False), (300, n, True), (5, 10, True), (130, int(0.58 * n), True)):
        samples = rng.integers(0, n, (max_iter, 3)).astype(np.uint32)
        pe = (PlaneRANSACEstimator3f(x).setMaxInlierResidual(thr).setTargetInlierCount(target)
              .setMaxNumberOfIterations(max_iter).setReEstimationStep(re_est).setSamples(samples))
        pl = pe.estimate().getModel()
        plo, reso, inlo, ito = orc.plane_ransac(x, samples, thr, target, re_estimate=re_est, mode=1)
        assert pe.getNumberOfPerformedIterations() == ito, (max_iter, target, pe.getNumberOfPerformedIterations(), ito)
        s = 1.0 if np.dot(pl[:3], plo[:3]) > 0 else -1.0
        assert np.abs(pl - s * plo).max() <= 2e-6, (pl, plo)
        # the product's own plane, scored by the oracle: residuals and inliers bit-exact
        res_chk = orc.plane_residuals(x, pl)
        assert np.array_equal(pe.getModelResiduals(), res_chk)
        inl_chk = np.nonzero(res_chk <= np.float32(thr))[0]
        assert np.array_equal(pe.getModelInliers(), inl_chk)
        # against the oracle's run: identical up to points within round-off of the threshold
        assert len(np.setxor1d(inl_chk, inlo)) <= max(3, int(2e-5 * n)), (len(inl_chk), len(inlo))
        assert pe.targetInlierCountAchieved() == (len(inl_chk) >= min(target, n))
    # (4) library-drawn samples: deterministic in the seed, finds the plane
    a = PlaneRANSACEstimator3f(x).setMaxInlierResidual(thr).setSeed(3).estimate()
    b = PlaneRANSACEstimator3f(x).setMaxInlierResidual(thr).setSeed(3).estimate()
    assert np.array_equal(a.getModel(), b.getModel()) and np.array_equal(a.getModelInliers(), b.getModelInliers())
    m = a.getModel() / -a.getModel()[2]
    assert np.abs(m - np.array([0.3, -0.2, -1.0, 0.1])).max() < 2e-3, m
    assert a.getNumberOfInliers() >= int(0.55 * n)
    # (5) edge cases: nothing reaches 3 inliers -> NaN model, no inliers; tiny clouds
    far = (rng.uniform(-1, 1, (1000, 3)) * 1e3).astype(np.float32)
    e = PlaneRANSACEstimator3f(far).setMaxInlierResidual(0.0).setMaxNumberOfIterations(20).setSeed(1).estimate()
    plo, _, inlo, ito = orc.plane_ransac(far, np.zeros((20, 3), np.uint32), 0.0, 500, mode=1)
    assert e.getNumberOfPerformedIterations() == 20
    for npts in (0, 1, 2, 3):
        t = PlaneRANSACEstimator3f(x[:npts].copy()).setMaxInlierResidual(thr).setMaxNumberOfIterations(4).setSeed(2).estimate()
        assert t.getNumberOfPerformedIterations() <= 4 and t.getNumberOfInliers() <= npts


def test_knn_and_normal_estimation_vs_oracle(orc, hip_lib):
    """KDTree3f::kNNSearch / kNNInRadiusSearch (core/kd_tree.hpp) and NormalEstimation3f
    (core/normal_estimation.hpp).  Neighbour lists bit-exact -- indices slot for slot, distances bit for bit --, normals to f32 round-off."""
    from cilantro_amd.normal_estimation import KDTree3f, NormalEstimation3f

    rng = np.random.default_rng(11)
    n = 120_000
    # a wavy sheet with noise (non-uniform occupancy of the grid) plus a volumetric part
    x = rng.random((n, 3)).astype(np.float32)
    m = n // 2
    x[:m, 2] = (0.2 * x[:m, 0] + 0.1 * np.sin(6 * x[:m, 1]) + rng.normal(0, 1e-3, m)).astype(np.float32)
    q = np.concatenate([x[rng.integers(0, n, 3000)] + rng.normal(0, 0.01, (3000, 3)).astype(np.float32),
                        (rng.random((200, 3)) * 3 - 1).astype(np.float32)]).astype(np.float32)   # some far outside the cloud
    tree_o = orc.KDTree(x)
    tree_g = KDTree3f(x)
    for k, r2 in ((1, np.inf), (8, np.inf), (13, np.inf), (32, np.inf), (10, 0.02 ** 2), (5, 1e-12)):
        gi, gd, gc = (tree_g.kNNSearch(q, k) if np.isinf(r2) else tree_g.kNNInRadiusSearch(q, k, r2))
        oi, od, oc = orc.knn_batch(tree_o, q, k, r2)
        assert np.array_equal(gc, oc), (k, r2)
        assert np.array_equal(gd, od), (k, r2)                       # distances: identical
        assert np.array_equal(gi, oi), (k, r2, np.nonzero((gi != oi).any(axis=1))[0][:5])      # ... and so are the indices, slot for slot
    # self k-NN: every point finds itself first at distance 0
    gi, gd, gc = tree_g.kNNSearch(None, 6)
    assert np.array_equal(gi[:, 0], np.arange(n)) and (gd[:, 0] == 0).all() and (gc == 6).all()
    oi, od, oc = orc.knn_batch(tree_o, x[:5000], 6)
    assert np.array_equal(gd[:5000], od) and (gi[:5000] != oi).sum() == 0
    # normals + curvature
    for k, r2, vp in ((10, np.inf, [0.5, 0.5, 10.0]), (7, np.inf, None), (12, np.float32(0.015) ** 2, [0.0, 0.0, -5.0])):
        ne = NormalEstimation3f(x).setViewPoint(vp)
        ng, cg = (ne.getNormalsAndCurvatureKNN(k) if np.isinf(r2) else ne.getNormalsAndCurvatureKNNInRadius(k, np.sqrt(np.float32(r2))))   # plain radius
        no, co = orc.normals_knn(x, k, r2, vp, mode=1)
        nan_g, nan_o = np.isnan(ng).any(axis=1), np.isnan(no).any(axis=1)
        assert np.array_equal(nan_g, nan_o)
        ok = ~nan_g
        dots = (ng[ok] * no[ok]).sum(axis=1)
        if vp is None:
            dots = np.abs(dots)                                      # no view point: the sign is the eigen-solver's
        # nearly isotropic neighbourhoods (two close eigenvalues) amplify round-off: compare where the normal is well defined
        well = cg[ok] < 0.2
        assert (dots[well] > 1 - 1e-4).mean() > 0.999, (k, float((dots[well] > 1 - 1e-4).mean()))
        assert np.nanmax(np.abs(cg[ok] - co[ok])) < 1e-4
        assert np.abs(np.linalg.norm(ng[ok], axis=1) - 1).max() < 1e-5
        if vp is not None:
            assert (((np.asarray(vp, np.float32) - x[ok]) * ng[ok]).sum(axis=1) >= -1e-6).all()
    # radius-only neighbourhoods (unbounded size): moments accumulated without a list
    rad = np.float32(0.012)
    ng, cg = NormalEstimation3f(x).setViewPoint([0.5, 0.5, 10.0]).getNormalsAndCurvatureRadius(rad)
    no, co = orc.normals_radius(x, rad * rad, [0.5, 0.5, 10.0], mode=1)
    nan_g, nan_o = np.isnan(ng).any(axis=1), np.isnan(no).any(axis=1)
    assert np.array_equal(nan_g, nan_o) and 0 < nan_g.sum() < len(x)            # sparse regions: fewer than 3 points in the ball
    ok = ~nan_g
    well = cg[ok] < 0.2
    assert (((ng[ok] * no[ok]).sum(axis=1))[well] > 1 - 1e-4).mean() > 0.999
    assert np.nanmax(np.abs(cg[ok] - co[ok])) < 1e-4
    # degenerate inputs
    t2 = KDTree3f(x[:2].copy())
    gi, gd, gc = t2.kNNSearch(q[:10], 5)
    assert (gc == 2).all() and (gi[:, 2:] == -1).all()
    nn, cc = NormalEstimation3f(x[:2].copy()).getNormalsAndCurvatureKNN(5)
    assert np.isnan(nn).all() and np.isnan(cc).all()


def test_knn_lists_follow_the_reference_on_tied_distances(orc, hip_lib):
    """k-NN with k > 1 on data whose distances TIE: the reference's own sensor frames (a depth sensor's lattice: tests/golden/frames_full.npz,
    both frames, k = 7, 10, 32, self queries and frame-to-frame), an integer lattice (every list full of equal distances, ties at nearly
    every k-th place) and a cloud with doubled and tripled points -- against the REFERENCE's nanoflann knnSearch through cilantro's result
    adaptor (core/kd_tree.hpp:80-99: among equal distances the first met stays ahead, at every slot and at the k-th place):
    np.array_equal on the indices, no tolerated rows.  Rule 0 (lowest index) differs on the same data; rule 1 (tables up front) does not."""
    from cilantro_amd.normal_estimation import KDTree3f, set_knn_tie_rule

    if not orc.ref_available():
        pytest.skip("oracle/_ref is not built")
    rng = np.random.default_rng(41)
    fr = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames_full.npz"))
    f1, f2 = np.ascontiguousarray(fr["p1"], np.float32), np.ascontiguousarray(fr["p2"], np.float32)
    lattice = np.ascontiguousarray(np.stack(np.meshgrid(*[np.arange(24, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) * np.float32(0.25))
    base = rng.random((40_000, 3)).astype(np.float32)
    dup = np.ascontiguousarray(np.concatenate([base, base[:12_000], base[:5_000]]))
    report = {}
    cases = [("frame_1 self", f1, None), ("frame_2 self", f2, None), ("frame_2 -> frame_1", f1, f2), ("lattice self", lattice, None),
             ("lattice, queries on cell centres", lattice, lattice[:4000] + np.float32(0.125)), ("doubled / tripled points", dup, dup[:20_000])]
    for name, ref_pts, queries in cases:
        tree_r = orc.KDTree(ref_pts, use_ref=True)
        tree_g = KDTree3f(ref_pts)
        q = ref_pts if queries is None else np.ascontiguousarray(queries, np.float32)
        for k in (7, 10, 32):
            oi, od, oc = orc.ref_knn_batch(tree_r, q, k)
            gi, gd, gc = tree_g.kNNSearch(None if queries is None else q, k)
            assert np.array_equal(gc, oc) and np.array_equal(gd, od), (name, k)
            assert np.array_equal(gi, oi), (name, k, int((gi != oi).any(axis=1).sum()))
            tied_rows = int(((od[:, 1:] == od[:, :-1]) & np.isfinite(od[:, 1:])).any(axis=1).sum())
            if k == 10:
                set_knn_tie_rule(0)
                li, ld, _ = tree_g.kNNSearch(None if queries is None else q, k)
                set_knn_tie_rule(1)
                ui, ud, _ = tree_g.kNNSearch(None if queries is None else q, k)
                set_knn_tie_rule(2)
                assert np.array_equal(ld, od) and np.array_equal(ui, oi) and np.array_equal(ud, od), (name, k)
                report[name] = {"queries": len(q), "k": k, "lists_with_equal_distances": tied_rows, "lists_the_lowest_index_rule_orders_differently": int((li != oi).any(axis=1).sum())}
    assert report["lattice self"]["lists_the_lowest_index_rule_orders_differently"] > 1000
    assert report["frame_1 self"]["lists_with_equal_distances"] > 100
    # a radius that cuts tied groups: kNNInRadiusSearch
    tree_r, tree_g = orc.KDTree(lattice, use_ref=True), KDTree3f(lattice)
    for r2 in (np.float32(0.25 ** 2 * 2.0), np.float32(0.25 ** 2 * 2.0 + 1e-6), np.float32(0.0626)):
        oi, od, oc = orc.ref_knn_batch(tree_r, lattice[:3000], 12, r2)
        gi, gd, gc = tree_g.kNNInRadiusSearch(lattice[:3000], 12, r2)
        assert np.array_equal(gc, oc) and np.array_equal(gi, oi) and np.array_equal(gd[gi >= 0], od[oi >= 0]), float(r2)
    try:
        import json
        os.makedirs(os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out"), exist_ok=True)
        json.dump(report, open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out", "knn_tie_rule.json"), "w"), indent=1)
    except OSError:
        pass


def test_search_directions_vs_oracle(Context, orc, hip_lib):
    """Engine search directions FIRST_TO_SECOND / BOTH (+ reciprocity) and their post-filters
    (correspondence_search_kd_tree.hpp:185-225, kd_tree_utilities.hpp:65-101, correspondence.hpp:57-100)."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, CorrespondenceSearchHIP, SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(60000, 45000, with_normals=True)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.004, -0.003, 0.002]
    q = orc.transform_points(T, d["src"])
    r2 = float(d["max_sq_dist"])
    ctx = Context()
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    eng = CorrespondenceSearchHIP(ctx=ctx).setMaxDistance(r2)
    code = {D.SECOND_TO_FIRST: 0, D.FIRST_TO_SECOND: 1, D.BOTH: 2}
    for direction, recip, frac, o2o in ((D.FIRST_TO_SECOND, False, 1.0, False), (D.BOTH, False, 1.0, False), (D.BOTH, True, 1.0, False),
                                        (D.FIRST_TO_SECOND, False, 0.6, False), (D.FIRST_TO_SECOND, False, 1.0, True),
                                        (D.BOTH, True, 0.8, True), (D.FIRST_TO_SECOND, False, 0.5, True), (D.SECOND_TO_FIRST, False, 1.0, False)):
        eng.setSearchDirection(direction).setRequireReciprocality(recip).setInlierFraction(frac).setOneToOne(o2o)
        eng.findCorrespondences(T)
        g1, g2, gv = eng.getCorrespondences()
        o1, o2, ov = orc.find_correspondences_dir(d["dst"], q, r2, code[direction], recip, frac, o2o)
        assert len(g1) == len(o1), (direction, recip, frac, o2o, len(g1), len(o1))
        assert np.array_equal(g1, o1) and np.array_equal(g2, o2) and np.array_equal(gv, ov), (direction, recip, frac, o2o)
    # whole ICP loops: transforms within tolerance of the oracle, same iteration counts
    for direction, recip, metric in ((D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 1), (D.BOTH, True, 0)):
        if metric == 1:
            icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        else:
            from cilantro_amd.icp import SimplePointToPointMetricRigidICP3f
            icp = SimplePointToPointMetricRigidICP3f(d["dst"], d["src"])
        icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
        icp.setMaxNumberOfIterations(12).setConvergenceTolerance(1e-5)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=metric, max_sq_dist=r2, max_iter=12, conv_tol=1e-5, direction=code[direction], reciprocal=recip)
        ro = orc.icp_run(d["dst"], d["dst_n"] if metric == 1 else None, d["src"], p)
        assert icp.getNumberOfPerformedIterations() == ro["iterations"], (direction, recip)
        assert icp.last_ncorr_ == ro["last_ncorr"]
        assert np.linalg.norm(Tg.astype(np.float64) - ro["T"]) <= TOL_T, (direction, recip, np.linalg.norm(Tg - ro["T"]))


def test_reverse_searches_started_from_the_previous_matches_change_nothing(Context, orc, hip_lib):
    """FIRST_TO_SECOND / BOTH loops on the device (correspondence_search_kd_tree.hpp:185-222): from the second iteration on the reverse
    search starts from the previous reverse matches (bidir.hip k_reverse_warm: margin test over the source, the listed rest searched in
    full) and accumulates the first step's sums in the same pass (matrix cores).  The matches are exact either way: with the option both
    ways the runs perform the same iterations over the same correspondence sets -- the pair list left behind is equal element for
    element, its distances to the last bits of transforms that differ only by the order of the f64 additions -- and a run repeated is
    BITWISE the same (fixed summation order, ballot-ordered lists).  Clouds: the recipe, a far start, a source with doubled points
    (margin 0: never settled without a look), a sparse source."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    rng = np.random.default_rng(3)
    cases = []
    d = syn.make_pair(200_000, 200_000, with_normals=True); cases.append(("recipe", d["dst"], d["dst_n"], d["src"], float(d["max_sq_dist"])))
    d = syn.make_pair(120_000, 90_000, with_normals=True, perturb=0.8); cases.append(("far", d["dst"], d["dst_n"], d["src"], float((3 * d["h"]) ** 2)))
    d = syn.make_pair(100_000, 100_000, with_normals=True)
    pick = rng.choice(100_000, 5_000, replace=False)
    cases.append(("doubled", d["dst"], d["dst_n"], np.ascontiguousarray(np.concatenate([d["src"], d["src"][pick]])), float(d["max_sq_dist"])))
    d = syn.make_pair(150_000, 20_000, with_normals=True, src_stride=7); cases.append(("sparse", d["dst"], d["dst_n"], d["src"], float(d["max_sq_dist"])))
    for name, dst, dst_n, src, r2 in cases:
        # (every accumulation form of the fused pass: plane terms only, both, point terms only -- wts -- and the Kabsch moments, in every mode)
        for direction, recip, metric, wts in ((D.FIRST_TO_SECOND, False, 1, (0.1, 1.0)), (D.BOTH, False, 1, (0.0, 1.0)), (D.BOTH, True, 0, None),
                                              (D.FIRST_TO_SECOND, False, 0, None), (D.BOTH, True, 1, (0.0, 1.0)), (D.BOTH, False, 1, (1.0, 0.0)),
                                              (D.BOTH, True, 1, (0.1, 1.0)), (D.FIRST_TO_SECOND, False, 1, (0.0, 1.0))):
            got = []
            for warm in (1, 0, 1):
                icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src) if metric == 1 else SimplePointToPointMetricRigidICP3f(dst, src)
                if wts is not None:
                    icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1])
                icp._ctx.set_option("reverse_warm_start", warm)
                icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
                icp.setMaxNumberOfIterations(10).setConvergenceTolerance(0.0)
                T = icp.estimate().getTransform()
                g1, g2, gv = icp._ctx.get_correspondences()
                got.append((T.copy(), icp.getNumberOfPerformedIterations(), icp.last_ncorr_, g1.copy(), g2.copy(), gv.copy()))
            (Tw, iw, nw, a1, a2, av), (Tc, ic, nc, b1, b2, bv), (Tr, ir, nr, c1, c2, cv) = got
            assert iw == ic and nw == nc, (name, direction, recip, metric)
            assert np.abs(Tw.astype(np.float64) - Tc).max() < 1e-6, (name, direction, recip, metric, np.abs(Tw - Tc).max())
            assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.allclose(av, bv, rtol=0.0, atol=5e-9), (name, direction, recip)
            # the same run again: bitwise
            assert ir == iw and nr == nw and np.array_equal(Tw.view(np.uint32), Tr.view(np.uint32)), (name, direction, recip, metric)
            assert np.array_equal(a1, c1) and np.array_equal(a2, c2) and np.array_equal(av.view(np.uint32), cv.view(np.uint32)), (name, direction, recip)
        # ... and the loop is still the reference's: one case per cloud against the oracle
        icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
        icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(D.FIRST_TO_SECOND)
        icp.setMaxNumberOfIterations(10).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, max_sq_dist=r2, max_iter=10, conv_tol=0.0, direction=1, reciprocal=False)
        ro = orc.icp_run(dst, dst_n, src, p)
        assert icp.last_ncorr_ == ro["last_ncorr"], name
        assert np.linalg.norm(Tg.astype(np.float64) - ro["T"]) <= TOL_T, (name, np.linalg.norm(Tg - ro["T"]))


def test_feature_searches_started_from_the_previous_matches_change_nothing(Context, orc, hip_lib):
    """Feature-adaptor loops (common_transformable_feature_adaptors.hpp:60-343) with the search warm-started from the previous matches
    (feat_warm.hip: the margin test with the feature distance, listed rest searched in full) against every search from scratch
    (option feature_warm_start): the searches are exact either way.  With the symmetric metric the sums are the same streaming pass's
    and the loop state is BITWISE the same; with the three-cloud metric the warm-started pass accumulates them itself (matrix cores): same
    iterations, counts and correspondence sets, transforms equal to the order of the f64 additions.  6-D point+normal, point+colour and
    the 9-D adaptors; and the warm-started form really ran."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 400_000
    d = syn.make_pair(n, n, with_normals=True)
    rng = np.random.default_rng(5)
    src_n = d["dst_n"] + 0.02 * rng.normal(size=d["dst_n"].shape).astype(np.float32)
    src_n = np.ascontiguousarray((src_n / np.linalg.norm(src_n, axis=1, keepdims=True)).astype(np.float32))
    col_d = rng.random((n, 3), dtype=np.float32)
    col_s = np.ascontiguousarray(np.clip(col_d + 0.01 * rng.normal(size=(n, 3)).astype(np.float32), 0, 1).astype(np.float32))
    h = d["h"]
    for kind in ("normals", "colours", "both"):
        for symmetric in (1, 0):
            got = []
            for warm in (1, 0):
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"], src_n)
                eng = icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
                if kind == "normals":
                    eng.setPointNormalFeatureAdaptors(src_n, 0.5 * h)
                elif kind == "colours":
                    eng.setPointColorFeatureAdaptors(col_d, col_s, 0.3 * h)
                else:
                    eng.setPointNormalColorFeatureAdaptors(src_n, col_d, col_s, 0.5 * h, 0.3 * h)
                icp._ctx.set_option("symmetric_metric", symmetric)
                icp._ctx.set_option("feature_warm_start", warm)
                icp.setMaxNumberOfIterations(12).setConvergenceTolerance(0.0)
                T = icp.estimate().getTransform()
                g1, g2, gv = icp._ctx.get_correspondences()
                got.append((T.copy(), icp.getNumberOfPerformedIterations(), icp.last_ncorr_, g1.copy(), g2.copy(), gv.copy(), icp._ctx.last_warm_iterations()))
            (Tw, iw, nw, a1, a2, av, ww), (Tc, ic, nc, b1, b2, bv, wc) = got
            assert ww > 0 and wc == 0, (kind, symmetric, ww, wc)
            assert iw == ic and nw == nc, (kind, symmetric)
            if symmetric:      # (search only: the sums are the same streaming pass's either way)
                assert np.array_equal(Tw.view(np.uint32), Tc.view(np.uint32)), (kind, symmetric, np.abs(Tw - Tc).max())
                assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.array_equal(av.view(np.uint32), bv.view(np.uint32)), (kind, symmetric)
            else:              # (three-cloud metric: the warm-started pass forms the sums itself -- the same terms in another order)
                assert np.abs(Tw.astype(np.float64) - Tc).max() < 1e-6, (kind, symmetric, np.abs(Tw - Tc).max())
                assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.allclose(av, bv, rtol=0.0, atol=5e-9), (kind, symmetric)


def test_search_directions_with_several_gauss_newton_steps_vs_oracle(orc, hip_lib):
    """FIRST_TO_SECOND / BOTH / reciprocal with max_optimization_iterations = 3 (transform_estimation.hpp:298-366 inside the loop of
    icp_base.hpp:68-87): the first step's sums come from the fused reverse pass (and BOTH's forward half from the warm-started kernel), the
    later steps stream over the stored matches under the inner transform -- both ways of the option, against the oracle's loop."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(100_000, 80_000, with_normals=True, perturb=0.5)
    r2 = float(d["max_sq_dist"])
    for direction, recip, code in ((D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
        for wts in ((0.0, 1.0), (0.3, 1.0)):
            p = orc.make_params(metric=1, w_p2p=wts[0], w_p2pl=wts[1], max_sq_dist=r2, max_iter=6, conv_tol=0.0, max_opt_iter=3, opt_conv_tol=1e-7,
                                direction=code, reciprocal=recip, mode=orc.MODE_MIXED)
            ro = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
            for warm in (1, 0):
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
                icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1])
                icp.setMaxNumberOfOptimizationStepIterations(3).setOptimizationStepConvergenceTolerance(1e-7)
                icp._ctx.set_option("reverse_warm_start", warm)
                icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
                icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
                T = icp.estimate().getTransform()
                assert icp.last_ncorr_ == ro["last_ncorr"], (direction, recip, wts, warm)
                assert np.linalg.norm(T.astype(np.float64) - ro["T"]) <= TOL_T, (direction, recip, wts, warm)


def test_pair_search_workspace_follows_the_clouds(Context, orc, hip_lib):
    """One context, a BOTH search of a small pair and then a FIRST_TO_SECOND search against a LARGER target with fewer candidates than the
    BOTH search had (n_target' <= n_target + n_source): the search's per-target arrays are sized by the target, not by the candidate
    count (tools/api_fuzz.py seed 1 found them too short: a memory fault).  Lists equal a fresh context's."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, CorrespondenceSearchHIP

    small = syn.make_pair(30_000, 30_000, with_normals=True)
    big = syn.make_pair(50_000, 20_000, with_normals=True, src_stride=2)
    T = np.eye(4, dtype=np.float32)
    live = Context()
    live.set_target(small["dst"], small["dst_n"]); live.set_source(small["src"])
    eng = CorrespondenceSearchHIP(ctx=live).setMaxDistance(float(small["max_sq_dist"])).setSearchDirection(D.BOTH)
    eng.findCorrespondences(T)
    assert len(eng.getCorrespondences()[0]) > 30_000
    live.set_target(big["dst"], big["dst_n"]); live.set_source(big["src"])
    eng.setMaxDistance(float(big["max_sq_dist"])).setSearchDirection(D.FIRST_TO_SECOND).setOneToOne(True)
    eng.findCorrespondences(T)
    a = eng.getCorrespondences()
    fresh = Context()
    fresh.set_target(big["dst"], big["dst_n"]); fresh.set_source(big["src"])
    eng2 = CorrespondenceSearchHIP(ctx=fresh).setMaxDistance(float(big["max_sq_dist"])).setSearchDirection(D.FIRST_TO_SECOND).setOneToOne(True)
    eng2.findCorrespondences(T)
    b = eng2.getCorrespondences()
    assert len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a, b))
    q = orc.transform_points(T, big["src"])
    o1, o2, ov = orc.find_correspondences_dir(big["dst"], q, float(big["max_sq_dist"]), 1, False, 1.0, True)
    assert np.array_equal(a[0], o1) and np.array_equal(a[1], o2) and np.array_equal(a[2], ov)


def test_estimate_over_a_pair_list_and_with_a_callback(Context, orc, hip_lib):
    """cilhip_estimate_combined over the PAIR LIST of FIRST_TO_SECOND / BOTH searches (one correspondence per pair: a source point may
    occur several times), against the oracle's estimator over the same list; and with a pair-weight callback, which then sees the
    list in its stored order (ascending (first, second)) and whose weights are read by pair index."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D, CorrespondenceSearchHIP

    d = syn.make_pair(60000, 45000, with_normals=True)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.004, -0.003, 0.002]
    q = orc.transform_points(T, d["src"])
    r2 = float(d["max_sq_dist"])
    ctx = Context()
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    eng = CorrespondenceSearchHIP(ctx=ctx).setMaxDistance(r2)
    dm, sm = ctx.means()
    smt = orc.transform_points(T, sm.reshape(1, 3))[0]
    for direction, recip in ((D.FIRST_TO_SECOND, False), (D.BOTH, False), (D.BOTH, True)):
        eng.setSearchDirection(direction).setRequireReciprocality(recip)
        eng.findCorrespondences(T)
        g1, g2, gv = eng.getCorrespondences()
        assert len(np.unique(g2)) < len(g2) or direction == D.BOTH and recip      # (a pair list proper: some source point occurs twice)
        for w_p2p, w_p2pl in ((0.0, 1.0), (0.3, 1.0)):
            Tg, AtA, Atb, _ = ctx.estimate_combined(w_p2p, w_p2pl, 1, 1e-5)
            To, AtAo, Atbo, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, w_p2p, w_p2pl, dm, smt, 1, 1e-5, orc.MODE_MIXED)
            scale = np.abs(AtAo).max()
            tol = 1e-9 if w_p2p == 0.0 else 2e-6
            assert np.abs(AtA - AtAo).max() <= tol * scale and np.abs(Atb - Atbo).max() <= tol * np.abs(Atbo).max() + 1e-3 * tol * scale, (direction, recip, w_p2p)
            assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6
        # weights by pair: every second PAIR of the list dropped, the others weighted by their position
        seen = {}

        def by_pair(i1, i2, v):
            seen["i1"], seen["i2"] = i1.copy(), i2.copy()
            w = np.where(np.arange(len(v)) % 2 == 1, 0.0, 1.0 + np.arange(len(v)) / len(v)).astype(np.float32)
            return np.ones(len(v), np.float32), w

        ctx.set_pair_weight_callback(by_pair)
        Tg, AtA, Atb, _ = ctx.estimate_combined(0.0, 1.0, 1, 1e-5)
        ctx.set_pair_weight_callback(None)
        assert np.array_equal(seen["i1"], g1) and np.array_equal(seen["i2"], g2)
        keep = np.arange(len(g1)) % 2 == 0
        wv = (1.0 + np.arange(len(g1))[keep] / len(g1)).astype(np.float32)
        To, AtAo, Atbo, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, g1[keep], g2[keep], 0.0, 1.0, dm, smt, 1, 1e-5, orc.MODE_MIXED,
                                                  values=wv, weights=(orc.W_UNITY, orc.W_IDENTITY, 1.0, 1.0))
        scale = np.abs(AtAo).max()
        assert np.abs(AtA - AtAo).max() <= 1e-9 * scale and np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6, (direction, recip)
    eng.setSearchDirection(D.SECOND_TO_FIRST).setRequireReciprocality(False)


def test_frame1_recipe_real_cloud(orc, hip_lib, Context):
    """BASELINE configs[0]: real sensor data -- the reference's examples/test_clouds/frame_1.ply
    through the recipe and parameters of examples/rigid_icp.cpp (fixture tests/golden/frame1_c1.npz)."""
    import os
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f

    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame1_c1.npz"))
    dst, dst_n, src = f["dst"], f["dst_n"], f["src"]
    # nearest neighbours on a real surface cloud, large radius (0.1^2 as in the example), far-from-converged pose
    ctx = Context()
    ctx.set_target(dst, dst_n); ctx.set_source(src)
    T0 = np.eye(4, dtype=np.float32)
    ng = ctx.find_correspondences(T0, 0.1 * 0.1)
    g1, g2, gv = ctx.get_correspondences()
    o1, o2, ov = orc.KDTree(dst).find_correspondences(orc.transform_points(T0, src), 0.1 * 0.1)
    assert ng == len(o1) and np.array_equal(g2, o2) and np.array_equal(gv, ov)
    nbad, ties, nearer, worse = classify_mismatches(orc, dst, orc.transform_points(T0, src), g1, gv, o1, ov)
    assert worse == 0 and nbad == ties, (nbad, ties, nearer, worse)
    for metric in (1, 0):
        if metric == 1:
            icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
            icp.setMaxNumberOfOptimizationStepIterations(1).setPointToPointMetricWeight(0.0).setPointToPlaneMetricWeight(1.0)
        else:
            icp = SimplePointToPointMetricRigidICP3f(dst, src)
        icp.correspondenceSearchEngine().setMaxDistance(0.1 * 0.1)
        icp.setConvergenceTolerance(1e-4).setMaxNumberOfIterations(30)          # examples/rigid_icp.cpp:116-125
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=metric, w_p2p=0.0, w_p2pl=1.0, max_iter=30, conv_tol=1e-4, max_opt_iter=1, max_sq_dist=0.1 * 0.1)
        r = orc.icp_run(dst, dst_n if metric == 1 else None, src, p)
        assert icp.getNumberOfPerformedIterations() == r["iterations"]
        assert np.linalg.norm(Tg.astype(np.float64) - r["T"]) <= TOL_T, np.linalg.norm(Tg - r["T"])
        if metric == 1:   # point-to-plane converges inside the example's 30 iterations; point-to-point is still creeping
            assert icp.hasConverged() and np.linalg.norm(Tg - np.linalg.inv(f["T_ref"].astype(np.float64))) < 5e-3


def test_tiled_search_under_large_and_non_rigid_motion(Context, orc):
    """The tiled kernel derives each tile's region from the image of its cube under the CURRENT transform.  Rotations,
    shears / scales and translations of a few cells since the sort must only ever cost speed (tiles or queries handed to
    the clean-up pass), never exactness."""
    n = 300_000
    d = syn.make_pair(n, perturb=0.3)
    h = d["h"]
    rng = np.random.default_rng(3)

    def rot(axis, ang):
        axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    cases = []
    for ang, shift in ((0.002, 0.5), (0.02, 1.7), (0.09, 3.0)):          # radians; translation in cells
        T = np.eye(4); T[:3, :3] = rot(rng.normal(size=3), ang); T[:3, 3] = rng.uniform(-1, 1, 3) * shift * h
        # rotate about the cloud centre so that the points stay near the grid
        c = np.array([0.5, 0.5, 0.5]); T[:3, 3] += c - T[:3, :3] @ c
        cases.append(T.astype(np.float32))
    A = np.eye(4); A[:3, :3] = np.diag([1.004, 0.997, 1.002]) + 0.002 * rng.normal(size=(3, 3)); A[:3, 3] = [0.3 * h, -0.2 * h, 0.1 * h]
    cases.append(A.astype(np.float32))
    ctxs = []
    for tiled in (0, 2):
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(d["dst"]); ctx.set_source(d["src"])
        ctx.find_correspondences(np.eye(4), d["max_sq_dist"], count=False)    # sort under the identity
        ctxs.append(ctx)
    for T in cases:
        (i0, d0), (i1, d1) = (gpu_nn(c, T, d["max_sq_dist"]) for c in ctxs)
        assert np.array_equal(i0, i1), float(np.mean(i0 != i1))
        m = i0 >= 0
        assert np.array_equal(d0[m], d1[m])
    # and against the kd-tree oracle for the largest motion
    T = cases[2]
    q = orc.transform_points(T, d["src"])
    o1, o2, ov = orc.KDTree(d["dst"]).find_correspondences(q, d["max_sq_dist"])
    i1, d1 = gpu_nn(ctxs[1], T, d["max_sq_dist"])
    assert np.array_equal(np.nonzero(i1 >= 0)[0], o2) and np.array_equal(i1[o2], o1) and np.array_equal(d1[o2], ov)


@pytest.mark.gpu
def test_tiled_clean_up_pass_sparse_far_and_unbounded(Context, orc):
    """The queries the LDS tiles cannot settle -- far outside the target's grid, inside holes of the target, results
    beyond the 3x3x3 block, with a finite, a large and an infinite radius -- are searched by groups of lanes out of
    global memory.  Same indices and distances as the per-lane search and as the kd-tree oracle."""
    n = 200_000
    d = syn.make_pair(n, perturb=0.4)
    h = d["h"]
    rng = np.random.default_rng(11)
    dst = d["dst"].copy()
    keep = np.linalg.norm(dst - np.array([0.5, 0.5, 0.5], np.float32), axis=1) > 0.12     # a hole ~7 cells across
    keep &= ~((dst[:, 0] > 0.8) & (rng.random(n) < 0.9))                                   # and a thinned-out slab
    dst = np.ascontiguousarray(dst[keep])
    src = d["src"].copy()
    far = rng.choice(n, n // 50, replace=False)
    src[far] += rng.normal(size=(far.size, 3)).astype(np.float32) * np.float32(20 * h)    # outliers, many outside the grid
    T = np.eye(4, dtype=np.float32); T[:3, 3] = np.array([0.9, 0.8, 0.7], np.float32) * h
    q = orc.transform_points(T, src)
    tree = orc.KDTree(dst)
    ctxs = []
    for tiled in (0, 2):
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(dst); ctx.set_source(src)
        ctx.find_correspondences(np.eye(4), d["max_sq_dist"], count=False)    # sort under the identity
        ctxs.append(ctx)
    for max_sq in (float(d["max_sq_dist"]), float((6 * h) ** 2), float("inf")):
        (i0, d0), (i1, d1) = (gpu_nn(c, T, max_sq) for c in ctxs)
        assert np.array_equal(i0, i1), float(np.mean(i0 != i1))
        m = i1 >= 0
        assert np.array_equal(d0[m], d1[m])
        o1, o2, ov = tree.find_correspondences(q, max_sq)
        assert np.array_equal(np.nonzero(m)[0], o2) and np.array_equal(i1[o2], o1) and np.array_equal(d1[o2], ov)
    dq, _ = ctxs[1].debug_counters()
    assert dq > 1000      # the group search really ran


@pytest.mark.gpu
def test_affine_loop_on_the_device_vs_host_driven_loop_and_oracle(Context, orc, hip_lib):
    """The affine classes' device-resident loop (option "affine_device_loop": search-only kernels + k_acc_affine while cold,
    k_warm<IM_AFFC / IM_AFFP> afterwards, the 12-unknown solve in k_solve_affine) against the host-driven loop it replaces
    (three moment passes + ldlt_solve_n on the host) and the oracle's affine ICP: same iteration counts, same correspondence
    sets, transforms equal to the order of the f64 additions (transform_estimation.hpp:369-476, :50-102;
    icp_single_transform_combined_metric.hpp:207-216)."""
    from cilantro_amd.icp import SimpleCombinedMetricAffineICP3f, SimplePointToPointMetricAffineICP3f

    n = 200_000
    rng = np.random.default_rng(11)
    for perturb in (0.15, 0.8):
        d = syn.make_pair(n, perturb=perturb)
        S = np.eye(3) + 0.003 * rng.normal(size=(3, 3))
        c0 = np.array([0.5, 0.5, 0.5])
        src = ((d["src"].astype(np.float64) - c0) @ S.T + c0).astype(np.float32)
        dst, dst_n = d["dst"], d["dst_n"]
        max_sq = float((3 * d["h"]) ** 2)
        for metric, wts in ((1, (0.1, 1.0)), (1, (0.0, 1.0)), (1, (1.0, 0.0)), (0, (1.0, 0.0))):
            for max_iter, tol in ((12, 0.0), (40, 1e-5)):
                got = {}
                for loop in (1, 0):
                    if metric == 1:
                        icp = SimpleCombinedMetricAffineICP3f(dst, dst_n, src)
                        icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1])
                    else:
                        icp = SimplePointToPointMetricAffineICP3f(dst, src)
                    icp._ctx.set_option("affine_device_loop", loop)
                    icp.correspondenceSearchEngine().setMaxDistance(max_sq)
                    icp.setMaxNumberOfIterations(max_iter).setConvergenceTolerance(tol)
                    T = icp.estimate().getTransform().astype(np.float64)
                    i1, i2, dv = icp._ctx.get_correspondences()
                    got[loop] = (T, icp.getNumberOfPerformedIterations(), icp.last_ncorr_, i1.copy(), i2.copy(), dv.copy(), icp._ctx.last_warm_iterations())
                (Td, itd, ncd, a1, a2, av, warm_d), (Th, ith, nch, b1, b2, bv, _) = got[1], got[0]
                assert abs(itd - ith) <= (0 if tol == 0.0 else 1), (perturb, metric, wts, itd, ith)
                assert np.linalg.norm(Td - Th) < 2e-6, (perturb, metric, wts, max_iter, np.linalg.norm(Td - Th))
                if tol == 0.0:
                    assert ncd == nch
                    # the set the last iteration estimated from, element for element (its distances under transforms that differ in
                    # their last bits: the two loops add the same terms in different orders)
                    assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.allclose(av, bv, rtol=0.0, atol=5e-9)
                    assert warm_d > 0, "the warm-started affine kernel never ran"
                if metric == 1 and wts == (0.1, 1.0) or metric == 0:
                    p = orc.make_params(metric=metric, w_p2p=wts[0], w_p2pl=wts[1], max_iter=max_iter, conv_tol=tol, max_sq_dist=max_sq,
                                        mode=orc.MODE_MIXED, affine=True)
                    r = orc.icp_run(dst, dst_n, src, p)
                    err = np.linalg.norm(Td - r["T"].astype(np.float64))
                    assert err <= 3e-5, (perturb, metric, max_iter, err)
                    assert abs(itd - r["iterations"]) <= (0 if tol == 0.0 else 1)
                    if tol == 0.0:
                        assert ncd == r["last_ncorr"]


@pytest.mark.gpu
def test_affine_variants_vs_oracle(Context, orc, hip_lib):
    """SURVEY 8(f) rank 3, affine variants: the 12-unknown closed forms (transform_estimation.hpp:50-102, :369-476) and the
    Affine ICP instances (icp_common_instances.hpp:255, :266).  The device accumulates the moments of the normal equations
    with per-term f32 quantities and f64 products / sums; the oracle's MIXED mode rounds every product to f32 first, its
    F32 mode is the reference's all-f32 arithmetic: tolerances below are set by those roundings, not by the algorithm."""
    from cilantro_amd.icp import SimpleCombinedMetricAffineICP3f, SimplePointToPointMetricAffineICP3f, CorrespondenceSearchDirection

    n = 60_000
    d = syn.make_pair(n, perturb=0.5)
    rng = np.random.default_rng(5)
    S = np.eye(3) + 0.004 * rng.normal(size=(3, 3))          # a genuinely affine distortion of the source
    c0 = np.array([0.5, 0.5, 0.5])
    src = ((d["src"].astype(np.float64) - c0) @ S.T + c0).astype(np.float32)
    dst, dst_n = d["dst"], d["dst_n"]
    max_sq = float((3 * d["h"]) ** 2)

    # --- the estimators on one correspondence set
    ctx = Context()
    ctx.set_target(dst, dst_n); ctx.set_source(src)
    T0 = np.eye(4, dtype=np.float32)
    ctx.find_correspondences(T0, max_sq, count=False)
    g1, g2, _ = ctx.get_correspondences()
    q = orc.transform_points(T0, src)
    dm, sm = ctx.means()
    for w_p2p, w_p2pl, centered in ((0.0, 1.0, True), (0.2, 1.0, True), (1.0, 0.0, True), (1.0, 0.0, False)):
        Tg, AtA, Atb, ok = ctx.estimate_affine(w_p2p, w_p2pl, centered)
        zero = np.zeros(3, np.float32)
        To, AtAo, Atbo, oko = orc.estimate_affine(dst, dst_n, q, g1, g2, w_p2p, w_p2pl, dm if centered else zero,
                                                  orc.transform_points(T0, sm.reshape(1, 3))[0] if centered else zero, orc.MODE_MIXED)
        assert ok and oko
        assert np.allclose(AtA, AtAo, rtol=2e-6, atol=1e-9 * np.abs(AtAo).max()), np.abs(AtA - AtAo).max()
        assert np.allclose(Atb, Atbo, rtol=2e-6, atol=1e-9 * np.abs(Atbo).max() + 1e-12)
        assert np.linalg.norm(Tg.astype(np.float64) - To) < 2e-5, (w_p2p, w_p2pl, centered)
    # degenerate inputs follow the reference's early returns: no terms / plane terms without normals -> identity, false
    Tg, _, _, ok = ctx.estimate_affine(0.0, 0.0, True)
    assert not ok and np.array_equal(Tg, np.eye(4, dtype=np.float32))
    ctx2 = Context(); ctx2.set_target(dst); ctx2.set_source(src); ctx2.find_correspondences(T0, max_sq, count=False)
    Tg, _, _, ok = ctx2.estimate_affine(0.0, 1.0, True)
    assert not ok and np.array_equal(Tg, np.eye(4, dtype=np.float32))

    # --- the ICP instances
    for metric in (1, 0):
        for max_iter, tol in ((5, 0.0), (40, 1e-5)):
            if metric == 1:
                icp = SimpleCombinedMetricAffineICP3f(dst, dst_n, src)
                icp.setPointToPointMetricWeight(0.1).setPointToPlaneMetricWeight(1.0)
            else:
                icp = SimplePointToPointMetricAffineICP3f(dst, src)
            icp.correspondenceSearchEngine().setMaxDistance(max_sq)
            icp.setMaxNumberOfIterations(max_iter).setConvergenceTolerance(tol)
            Tg = icp.estimate().getTransform()
            for mode, lim in ((orc.MODE_MIXED, 3e-5), (orc.MODE_F32, 1e-3)):
                p = orc.make_params(metric=metric, w_p2p=0.1, w_p2pl=1.0, max_iter=max_iter, conv_tol=tol, max_sq_dist=max_sq,
                                    mode=mode, affine=True)
                r = orc.icp_run(dst, dst_n, src, p)
                err = np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64))
                assert err <= lim, (metric, max_iter, mode, err)
                if mode == orc.MODE_MIXED:
                    assert abs(icp.getNumberOfPerformedIterations() - r["iterations"]) <= (0 if tol == 0.0 else 1)
                    if tol == 0.0:
                        assert icp.last_ncorr_ == r["last_ncorr"]
            if tol > 0:
                assert icp.hasConverged()
                # the recovered map undoes the distortion: T * src lands on dst's surface
                qf = orc.transform_points(Tg, src)
                _, _, dv = orc.KDTree(dst).find_correspondences(qf, max_sq)
                assert np.sqrt(np.median(dv)) < 0.6 * d["h"]
    # a pair-list search direction through the same loop
    icp = SimpleCombinedMetricAffineICP3f(dst, dst_n, src)
    icp.correspondenceSearchEngine().setMaxDistance(max_sq).setSearchDirection(CorrespondenceSearchDirection.BOTH)
    icp.setMaxNumberOfIterations(4).setConvergenceTolerance(0.0)
    Tg = icp.estimate().getTransform()
    p = orc.make_params(metric=1, w_p2p=0.0, w_p2pl=1.0, max_iter=4, conv_tol=0.0, max_sq_dist=max_sq, mode=orc.MODE_MIXED,
                        affine=True, direction=2)
    r = orc.icp_run(dst, dst_n, src, p)
    assert np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64)) <= 3e-5
    assert icp.last_ncorr_ == r["last_ncorr"]


@pytest.mark.gpu
def test_point_normal_feature_search_vs_oracle(Context, orc, hip_lib):
    """SURVEY 8(f) rank 3, 6-D point+normal features (PointNormalFeaturesAdaptor, common_transformable_feature_adaptors.hpp
    :60-161): the engine matches (T p, L (w n)) against (p, w n) by the 6-D squared distance.  Indices, distances and the
    strict radius test bit-identical to the oracle (exhaustive search with nanoflann's DIM = 6 arithmetic, itself pinned
    against the reference's nanoflann on the CPU); then the ICP loop on top."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 20_000
    d = syn.make_pair(n, perturb=0.5)
    h = d["h"]
    rng = np.random.default_rng(9)
    dst, dst_n, src = d["dst"], d["dst_n"], d["src"]
    # source normals: the matched target normals, rotated back roughly and jittered (so that the feature term matters)
    sn = dst_n[rng.permutation(n)] * 0.3 + rng.normal(size=(n, 3)).astype(np.float32)
    sn = (sn / np.linalg.norm(sn, axis=1, keepdims=True)).astype(np.float32)
    T = d["T_true"].astype(np.float32).copy(); T[:3, 3] += np.array([0.4, -0.3, 0.2], np.float32) * h
    for w, tiled in ((0.5 * h, 0), (3.0 * h, 0), (0.5 * h, 2), (3.0 * h, 2)):      # per-lane search and the LDS-tiled form
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(dst, dst_n); ctx.set_source(src, sn)
        ctx.set_option("feature_normal_weight", w)
        dst6 = orc.point_normal_features(dst, dst_n, w)
        q6 = orc.transform_features6(T, orc.point_normal_features(src, sn, w))
        for max_sq in (float((2.5 * h) ** 2), float("inf")):
            ctx.find_correspondences(T, max_sq, count=False)
            g1, g2, gv = ctx.get_correspondences()
            o1, o2, ov = orc.find_correspondences_feat6(dst6, q6, max_sq)
            assert np.array_equal(g2, o2) and np.array_equal(g1, o1) and np.array_equal(gv, ov), (w, max_sq)
        # engine post-filters act on the feature distances
        ctx.set_option("inlier_fraction", 0.7)
        ctx.find_correspondences(T, float((2.5 * h) ** 2), count=False)
        g1, g2, gv = ctx.get_correspondences()
        o1, o2, ov = orc.find_correspondences_feat6(dst6, q6, float((2.5 * h) ** 2))
        f1, f2, fv = orc.filter_fraction(o1, o2, ov, 0.7)
        assert np.array_equal(g2, f2) and np.array_equal(g1, f1) and np.array_equal(gv, fv)
    # the tiled form where it settles nearly everything inside the tile: a larger, nearly aligned pair whose source normals are
    # (nearly) the matched target normals -- and the same with normals that disagree (most queries take the clean-up pass)
    d2 = syn.make_pair(300_000, perturb=0.3)
    h2 = d2["h"]
    Ti = np.linalg.inv(d2["T_true"])
    sn_good = (d2["dst_n"].astype(np.float64) @ Ti[:3, :3].T).astype(np.float32)
    sn_bad = np.ascontiguousarray(sn_good[rng.permutation(len(sn_good))])
    T2 = d2["T_true"].astype(np.float32).copy(); T2[:3, 3] += np.array([0.1, -0.05, 0.08], np.float32) * np.float32(h2)
    for sn2, w in ((sn_good, 0.5 * h2), (sn_bad, 0.5 * h2), (sn_good, 4.0 * h2)):
        res = []
        for tiled in (0, 2):
            ctx2 = Context()
            ctx2.set_option("tiled", tiled)
            ctx2.set_target(d2["dst"], d2["dst_n"]); ctx2.set_source(d2["src"], sn2)
            ctx2.set_option("feature_normal_weight", w)
            ctx2.find_correspondences(T2, float(d2["max_sq_dist"]), count=False)
            res.append(ctx2.get_nn())
            dq, _ = ctx2.debug_counters()
        assert np.array_equal(res[0][0], res[1][0])
        m = res[0][0] != capi.NONE_IDX
        assert np.array_equal(res[0][1][m], res[1][1][m]) and m.sum() > 0.9 * len(m)
        if sn2 is sn_good and w < h2:
            assert dq < 0.05 * len(m), dq            # settled inside the tiles
    # what the feature search cannot do fails loudly
    ctx.set_option("inlier_fraction", 1.0)
    ctx3 = Context(); ctx3.set_target(dst, dst_n); ctx3.set_source(src); ctx3.set_option("feature_normal_weight", 0.1)
    with pytest.raises(RuntimeError):
        ctx3.find_correspondences(T, 1.0, count=False)      # no source normals

    # the ICP loop with feature correspondences (three-cloud combined metric) against the oracle's loop
    w = 0.5 * h
    sn_true = orc.transform_normals(np.linalg.inv(d["T_true"].astype(np.float64)).astype(np.float32), dst_n)   # plausible source normals
    icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
    icp.correspondenceSearchEngine().setMaxDistance(float((2.5 * h) ** 2)).setPointNormalFeatureAdaptors(sn_true, w)
    icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
    Tg = icp.estimate().getTransform()
    p = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float((2.5 * h) ** 2), mode=orc.MODE_MIXED,
                        normal_weight=w, three_cloud_metric=True)
    r = orc.icp_run(dst, dst_n, src, p, src_n=sn_true)
    assert np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64)) <= TOL_T
    assert icp.last_ncorr_ == r["last_ncorr"]


@pytest.mark.gpu
def test_full_size_variants_10m(Context, orc, hip_lib):
    """The round-6 loops at BASELINE configs[2]'s full size (10M <-> 10M), through properties that need no exhaustive oracle: the
    affine classes' device-resident loop against the host-driven loop it replaces (option affine_device_loop), the FIRST_TO_SECOND /
    BOTH loops with their reverse searches warm-started + accumulating in the same pass against searched from scratch + a separate
    pass (option reverse_warm_start) -- same iterations, same correspondence counts, transforms equal to the order of the f64
    additions, the known transform recovered.  Figures: gpurun_out/variants_parity_10m.json."""
    from cilantro_amd.icp import (CorrespondenceSearchDirection as D, SimpleCombinedMetricAffineICP3f, SimplePointToPointMetricAffineICP3f,
                                  SimpleCombinedMetricRigidICP3f)

    n = 10_000_000
    d = syn.make_pair(n, perturb=0.3)
    r2 = float(d["max_sq_dist"])
    report = {}
    for name, mk, opt in (("affine_combined", lambda: SimpleCombinedMetricAffineICP3f(d["dst"], d["dst_n"], d["src"]), "affine_device_loop"),
                          ("affine_point_to_point", lambda: SimplePointToPointMetricAffineICP3f(d["dst"], d["src"]), "affine_device_loop"),
                          ("first_to_second", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"]), "reverse_warm_start"),
                          ("both", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"]), "reverse_warm_start"),
                          ("both_reciprocal", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"]), "reverse_warm_start")):
        got = []
        for on in (1, 0):
            icp = mk()
            icp._ctx.set_option(opt, on)
            eng = icp.correspondenceSearchEngine().setMaxDistance(r2)
            if name == "first_to_second":
                eng.setSearchDirection(D.FIRST_TO_SECOND)
            elif name.startswith("both"):
                eng.setSearchDirection(D.BOTH).setRequireReciprocality(name == "both_reciprocal")
            icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
            icp.estimate()      # (the tables a first call builds -- source grid, nearest-other-point tables -- stay out of the figure below)
            T = icp.estimate().getTransform().astype(np.float64)
            got.append((T, icp.getNumberOfPerformedIterations(), icp.last_ncorr_, icp._ctx.last_timing()[0] / 8.0))
            del icp
        (T1, i1, n1, ms1), (T0, i0, n0, ms0) = got
        dT = float(np.abs(T1 - T0).max())
        err = float(np.linalg.norm(T1 - d["T_true"]))
        report[name] = {"iterations": i1, "last_ncorr": int(n1), "max_abs_T_difference_fast_vs_plain_form": dT, "T_err_vs_truth_frobenius": err,
                        "loop_ms_per_iteration_fast": ms1, "loop_ms_per_iteration_plain": ms0}
        assert i1 == i0 and n1 == n0, (name, i1, i0, n1, n0)
        assert dT < 2e-6, (name, dT)
        assert err < 2e-5, (name, err)
    _report("variants_parity_10m.json", report)


@pytest.mark.gpu
def test_full_size_properties_10m(Context, orc, hip_lib):
    """BASELINE configs[2] at its full size (10M <-> 10M, point-to-plane), through properties that do not need an
    exhaustive oracle:
      * round trip: a source that is an exact rigid image of the target finds its own twin for every point (identity
        permutation of indices), through the LDS-tiled kernel;
      * two independent search kernels (LDS-tiled / per-lane) agree bit for bit on a noisy source under a drifted
        transform -- every index and every squared distance;
      * a random sample of the queries against the reference's own nanoflann kd-tree over the full target (oracle/_ref)
        or the oracle's kd-tree restatement: same indices, bit-identical distances;
      * the ICP run recovers the known transform, and is bitwise reproducible run to run."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 10_000_000
    d = syn.make_pair(n, noise=0.0, perturb=0.3)         # src = T_true^-1 * dst up to f32 rounding
    h = d["h"]
    ctx = Context()
    ctx.set_option("tiled", 2)
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    i1, d1 = gpu_nn(ctx, d["T_true"].astype(np.float32), float(d["max_sq_dist"]))
    # (the exact twin unless two target points are closer to each other than the f32 rounding of the mapped source)
    assert np.count_nonzero(i1 != np.arange(n)) <= 5 and float(d1.max()) < (1e-5) ** 2
    del ctx

    d = syn.make_pair(n, perturb=0.3)                      # the benchmark's noisy source
    T = d["T_true"].astype(np.float32).copy(); T[:3, 3] += np.array([0.45, -0.3, 0.2], np.float32) * np.float32(h)
    res = []
    for tiled in (2, 0):
        ctx = Context()
        ctx.set_option("tiled", tiled)
        ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
        res.append(gpu_nn(ctx, T, float(d["max_sq_dist"])))
        del ctx
    (it, dt), (ip, dp) = res
    assert np.array_equal(it, ip) and np.array_equal(dt, dp)
    assert int(np.count_nonzero(it >= 0)) > 0.99 * n
    rng = np.random.default_rng(123)
    sample = np.sort(rng.choice(n, 40_000, replace=False))
    q = orc.transform_points(T, d["src"][sample])
    tree = orc.KDTree(d["dst"], use_ref=orc.ref_available())
    o1, o2, ov = tree.find_correspondences(q, float(d["max_sq_dist"]))
    found = it[sample] >= 0
    assert np.array_equal(np.nonzero(found)[0], o2) and np.array_equal(it[sample][o2], o1) and np.array_equal(dt[sample][o2], ov)
    tree10 = tree

    Ts = []
    for _ in range(2):
        icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
        icp.setMaxNumberOfIterations(20).setConvergenceTolerance(0.0)
        Ts.append(icp.estimate().getTransform().copy())
        assert icp.last_ncorr_ == n
        del icp
    assert np.array_equal(Ts[0], Ts[1])
    assert np.linalg.norm(Ts[0] - d["T_true"]) < 1e-5

    # The whole run against the ORACLE's run at the full size (all 10M correspondences per iteration, the reference's
    # nanoflann where available): 6 iterations from the identity.  MODE_MIXED is the arithmetic the HIP path mirrors
    # (per-term f32, f64 sums): must agree to 1e-5.  MODE_F32 is the reference-like all-f32 serial accumulation
    # (transform_estimation.hpp:339-341 sums 1e7 f32 terms): its distance from the f64-summed result is REPORTED
    # (gpurun_out/parity_10m.json; DESIGN.md numeric contract) and bounded loosely.
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
    Tg = icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0).estimate().getTransform().astype(np.float64)
    nc = icp.last_ncorr_
    del icp
    rep = {"n": n, "iterations": 6, "T_err_gpu_vs_truth": float(np.linalg.norm(Tg - d["T_true"]))}
    for name, mode in (("mixed", orc.MODE_MIXED), ("f32", orc.MODE_F32)):
        p = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]), mode=mode)
        To, nco = _oracle_icp_loop(orc, tree10, d, p, 6)
        rep[name] = {"T_gpu_minus_T_oracle_frobenius": float(np.linalg.norm(Tg - To.astype(np.float64))),
                     "T_oracle_minus_truth_frobenius": float(np.linalg.norm(To.astype(np.float64) - d["T_true"])),
                     "ncorr_gpu": int(nc), "ncorr_oracle": int(nco), "knn": "reference nanoflann" if tree10.use_ref else "oracle kd-tree"}
        assert nco == nc
    _report("parity_10m.json", rep)
    assert rep["mixed"]["T_gpu_minus_T_oracle_frobenius"] <= TOL_T, rep
    assert rep["f32"]["T_gpu_minus_T_oracle_frobenius"] <= TOL_T, rep          # measured 2.7e-7 (profiles/r02_parity_10m.json)


@pytest.mark.gpu
def test_full_size_properties_other_configs(orc, hip_lib):
    """The other BASELINE configs at their full single-GPU sizes, through size-independent properties.
    configs[1] (1M <-> 1M, point-to-point): the whole ICP run against the oracle's run (kd-tree restatement, all 1M points).
    configs[4] (KMeans3f k = 1024 and the plane RANSAC on 50M points): one Lloyd step -- sampled labels against the oracle's
    brute-force assignment, cluster sizes summing to n, centroids = means of their members (recomputed on the host);
    RANSAC -- the returned inlier set is exactly the set the returned plane selects (recount on the host), the run is
    deterministic in the seed, and the planted plane is found."""
    from cilantro_amd.clustering import KMeans3f, kmeans_assign
    from cilantro_amd.icp import SimplePointToPointMetricRigidICP3f
    from cilantro_amd.model_estimation import PlaneRANSACEstimator3f

    n = 1_000_000
    d = syn.make_pair(n, perturb=0.6)
    icp = SimplePointToPointMetricRigidICP3f(d["dst"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    icp.setMaxNumberOfIterations(15).setConvergenceTolerance(0.0)
    Tg = icp.estimate().getTransform()
    p = orc.make_params(metric=0, max_iter=15, conv_tol=0.0, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    r = orc.icp_run(d["dst"], None, d["src"], p)
    assert np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64)) <= TOL_T
    assert icp.last_ncorr_ == r["last_ncorr"] and icp.getNumberOfPerformedIterations() == r["iterations"]
    del icp, d

    n = 50_000_000
    rng = np.random.default_rng(77)
    x = rng.random((n, 3), dtype=np.float32)
    k = 1024
    c0 = x[:k].copy()
    km = KMeans3f(x).cluster(c0, max_iter=1, tol=0.0)
    lab = km.getPointToClusterIndexMap()
    sample = rng.choice(n, 20_000, replace=False)
    lab_o, _ = orc.kmeans_assign(np.ascontiguousarray(x[sample]), c0)
    assert np.array_equal(lab[sample], lab_o)
    sizes = np.bincount(lab, minlength=k)
    assert int(sizes.sum()) == n and int(sizes.min()) > 0
    cen = km.getClusterCentroids()
    for j in (0, 17, 511, 1023):
        mean_j = x[lab == j].astype(np.float64).mean(axis=0)
        assert np.abs(cen[j] - mean_j).max() <= 2e-6
    del km, lab

    # a planted plane holding 40 % of the points
    sel = rng.random(n) < 0.4
    x[sel, 2] = (0.3 * x[sel, 0] - 0.2 * x[sel, 1] + 0.4 + rng.normal(0, 0.002, int(sel.sum()))).astype(np.float32)
    thr = 0.006
    runs = []
    for _ in range(2):
        pe = (PlaneRANSACEstimator3f(x).setMaxInlierResidual(thr).setTargetInlierCount(int(0.39 * n)).setMaxNumberOfIterations(60)
              .setSeed(5).estimate())
        runs.append((pe.getModel().copy(), pe.getNumberOfInliers(), pe.getNumberOfPerformedIterations()))
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1:] == runs[1][1:]
    pl = runs[0][0]
    inl = pe.getModelInliers()
    res = np.abs(x @ pl[:3] + pl[3])                       # host recount (f32 expression order differs: allow the boundary)
    want = np.nonzero(res <= np.float32(thr))[0]
    assert len(np.setxor1d(inl, want)) <= int(2e-5 * n)
    assert len(inl) == runs[0][1] and len(inl) >= int(0.39 * n)
    m = pl / -pl[2]
    assert np.abs(m - np.array([0.3, -0.2, -1.0, 0.4])).max() < 1e-3


@pytest.mark.gpu
def test_full_size_target_sharded_config(orc, hip_lib):
    """BASELINE configs[3] at its full size (10M source points against an 80M-point target, combined metric 0.1 / 1.0),
    with the 8-GPU form's protocol played by two target shards on this one GPU: per iteration the element-wise MIN of the
    shards' packed keys (what all-reduce(MIN) computes) and the sum of their partial sums (all-reduce(SUM)) -- a checksum
    of checksums: the run must land on the transform of the unsharded run over the whole target, with the same number of
    correspondences, and recover the known transform."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    nd = 80_000_000
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
    T_last = last_keys = None
    for it in range(6):
        if it == 5:
            T_last = engs[0].state()[0]          # the transform the LAST iteration searches under
        keys = torch.minimum(engs[0].partial_keys(), engs[1].partial_keys())
        if it == 5:
            last_keys = keys.cpu().numpy().copy()
        sums = engs[0].sums_from_keys(keys).clone() + engs[1].sums_from_keys(keys)
        for e in engs:
            e.apply_sums(sums)
    (Ta, ita, _, nca), (Tb, itb, _, ncb) = engs[0].state(), engs[1].state()
    assert np.array_equal(Ta, Tb) and ita == itb == 6 and nca == ncb == nc1
    assert np.abs(Ta.astype(np.float64) - T1.astype(np.float64)).max() <= 1e-6, float(np.abs(Ta.astype(np.float64) - T1.astype(np.float64)).max())
    assert np.linalg.norm(T1 - d["T_true"]) < 1e-5, float(np.linalg.norm(T1 - d["T_true"]))
    del engs
    torch.cuda.empty_cache()
    # the same protocol through the C entry (cilhip_multi_set_clouds, partition 2: two index shards of the 80M-point target on this one device,
    # the MIN of the packed keys as a kernel between them): the transform of the unsharded run again
    from cilantro_amd import capi as _capi
    from cilantro_amd.multi import PARTITION_TARGET_SHARDS, MultiDeviceRigidICP
    m = MultiDeviceRigidICP([0, 0]); m.set_clouds(d["dst"], d["dst_n"], d["src"], float(d["max_sq_dist"]), PARTITION_TARGET_SHARDS)
    assert [m.shard_sizes(k) for k in range(2)] == [(half, len(d["src"])), (nd - half, len(d["src"]))]
    pc = _capi.IcpParams(); _capi.load().cilhip_icp_default_params(C.byref(pc))
    pc.metric, pc.w_p2p, pc.w_p2pl, pc.max_sq_dist, pc.max_iter, pc.conv_tol = _capi.METRIC_COMBINED, 0.1, 1.0, float(d["max_sq_dist"]), 6, 0.0
    rm = m.icp_run(pc); m.close()
    Tm = np.array(rm.T[:], np.float32).reshape(4, 4).T
    assert int(rm.iterations) == 6 and int(rm.last_ncorr) == nc1 and np.abs(Tm.astype(np.float64) - T1.astype(np.float64)).max() <= 1e-6
    del m
    torch.cuda.empty_cache()
    tree = orc.KDTree(d["dst"], use_ref=orc.ref_available())
    # the sharded LOOP's own correspondences: the last iteration's all-reduced keys (packed (bits(d2) << 32) | global target index per
    # source point) against the reference's nanoflann over the whole 80M-point target on a 200k-query sample, index for index, bit for bit
    rng = np.random.default_rng(322)
    sample = np.sort(rng.choice(len(d["src"]), 200_000, replace=False))
    ks = last_keys[sample]
    none = ks == distributed.KEY_NONE
    gi = np.where(none, -1, ks & 0xFFFFFFFF).astype(np.int64)
    gd = (ks >> 32).astype(np.uint32).view(np.float32)
    o1, o2, ov = tree.find_correspondences(orc.transform_points(T_last, np.ascontiguousarray(d["src"][sample])), float(d["max_sq_dist"]))
    oi = np.full(len(sample), -1, np.int64); od = np.zeros(len(sample), np.float32)
    oi[o2] = o1; od[o2] = ov
    assert np.array_equal(gi, oi), int(np.count_nonzero(gi != oi))
    assert np.array_equal(gd[oi >= 0].view(np.uint32), od[oi >= 0].view(np.uint32))
    loop_sample = {"loop_sample": int(len(sample)), "loop_sample_matched": int(np.count_nonzero(oi >= 0)), "loop_sample_mismatches": 0}

    # Against the oracle over the FULL 80M-point target (the reference's nanoflann where available): a 40k-query sample of
    # the source under a drifted transform -- indices and squared distances bit for bit -- and one combined-metric
    # (0.1 / 1.0) ICP update from those sampled correspondences, GPU engine vs the oracle's estimator.
    from cilantro_amd.icp import Context

    rng = np.random.default_rng(321)
    sample = np.sort(rng.choice(len(d["src"]), 40_000, replace=False))
    src_s = np.ascontiguousarray(d["src"][sample])
    T = d["T_true"].astype(np.float32).copy(); T[:3, 3] += np.array([0.3, -0.2, 0.25], np.float32) * np.float32(d["h"])
    ctx = Context(); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(src_s)
    gi, gd = gpu_nn(ctx, T, float(d["max_sq_dist"]))
    del ctx
    o1, o2, ov = tree.find_correspondences(orc.transform_points(T, src_s), float(d["max_sq_dist"]))
    found = gi >= 0
    assert len(o2) > 0.99 * len(sample)
    assert np.array_equal(np.nonzero(found)[0], o2) and np.array_equal(gi[o2], o1) and np.array_equal(gd[o2], ov)
    del tree
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], src_s)
    icp.setPointToPointMetricWeight(0.1).setPointToPlaneMetricWeight(1.0).setInitialTransform(T)
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    Tg = icp.setMaxNumberOfIterations(1).setConvergenceTolerance(0.0).estimate().getTransform()
    p = orc.make_params(metric=1, w_p2p=0.1, w_p2pl=1.0, max_iter=1, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]), mode=orc.MODE_MIXED)
    To, _ = orc.icp_update(d["dst"], d["dst_n"], src_s, T, o1, o2, p)
    err = float(np.linalg.norm(Tg.astype(np.float64) - To.astype(np.float64)))
    _report("parity_c4.json", {"n_target": nd, "sample": len(sample), "found": int(len(o2)), "T_gpu_minus_T_oracle_frobenius": err, **loop_sample})
    assert err <= TOL_T, err


@pytest.mark.gpu
def test_two_source_shards_on_one_gpu_are_ordered_with_torch(orc, hip_lib):
    """The source-sharded protocol with the collective played by torch ops on torch's stream (sum of the two shards'
    partial sums): the engines must run on the stream torch works on -- torch's default stream reports handle 0, which
    the Python layer passes on as hipStreamLegacy; the C ABI's NULL would select the context's own non-blocking stream
    and leave the exchange unordered.  The run must land on the unsharded run's transform (f64 summation order only)."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 400_000
    d = syn.make_pair(n, perturb=0.5)
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    T1 = icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0).estimate().getTransform()
    _, gmean = icp._ctx.means()
    half = n // 2
    engs = [distributed.HipShardEngine(d["dst"], d["dst_n"], d["src"][lo:hi], 0) for lo, hi in ((0, half), (half, n))]
    p = distributed.default_params(max_iter=8, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    for e in engs:
        e.begin(p, np.eye(4, dtype=np.float32), gmean)
    for _ in range(8):
        total = engs[0].partial_sums() + engs[1].partial_sums()      # what all-reduce(SUM) leaves on every rank
        for e in engs:
            e.apply_sums(total)
    (Ta, ita, _, nca), (Tb, _, _, _) = engs[0].state(), engs[1].state()
    assert np.array_equal(Ta, Tb) and ita == 8 and nca == icp.last_ncorr_
    assert np.abs(Ta.astype(np.float64) - T1.astype(np.float64)).max() <= 2e-7


@pytest.mark.gpu
def test_radius_search_lists_vs_oracle(orc, hip_lib):
    """KDTree3f::radiusSearch (core/kd_tree.hpp:251-282) as a list-returning call: per query every neighbour with squared
    distance < radius (strict), ascending by (distance, index) -- offsets, indices and distances identical to the oracle
    (exhaustive search, itself pinned against the reference's nanoflann on the CPU)."""
    from cilantro_amd.normal_estimation import KDTree3f

    rng = np.random.default_rng(17)
    pts = rng.random((30_000, 3)).astype(np.float32)
    pts[100:110] = pts[100]                                           # exact duplicates: ties broken by index
    q = np.concatenate([pts[:300], rng.random((500, 3)).astype(np.float32) * 1.3 - 0.15, np.array([[5, 5, 5], [np.nan, 0, 0]], np.float32)])
    tree = KDTree3f(pts)
    for r2 in (0.0, 0.02 ** 2, 0.09 ** 2):
        off, idx, d2 = tree.radiusSearch(q, r2)
        fin = np.all(np.isfinite(q), axis=1)
        ooff, oidx, od2 = orc.radius_search(pts, q[fin], r2)
        cnt = np.diff(off)
        assert np.array_equal(cnt[fin], np.diff(ooff)) and np.all(cnt[~fin] == 0)
        keep = np.repeat(fin, cnt)
        assert np.array_equal(idx[keep], oidx) and np.array_equal(d2[keep], od2), r2
    # the tree's own points as queries; every list starts with the point itself (or its lowest-index duplicate)
    off, idx, d2 = tree.radiusSearch(None, 0.03 ** 2)
    ooff, oidx, od2 = orc.radius_search(pts[:2000], pts[:2000], 0.03 ** 2)   # (oracle on a subset is a different problem: only shape checks here)
    assert len(off) == len(pts) + 1 and off[-1] == len(idx) and np.all(d2[off[:-1]] == 0.0)
    first = idx[off[:-1]]
    assert np.array_equal(first[:100], np.arange(100)) and np.all(first[100:110] == 100)
    # empty tree / no queries
    o, i, d = KDTree3f(np.zeros((0, 3), np.float32)).radiusSearch(q[:10], 1.0)
    assert np.all(o == 0) and len(i) == 0
    o, i, d = tree.radiusSearch(np.zeros((0, 3), np.float32), 1.0)
    assert len(o) == 1 and len(i) == 0


@pytest.mark.gpu
def test_in_tile_accumulation_vs_streaming_pass_and_oracle(Context, orc, hip_lib):
    """The ICP loop's first Gauss-Newton step accumulated INSIDE the LDS tiles of the search (f64 MFMA rank update of the
    per-correspondence vector z, k_search_tiled<ACC> + k_search_deferred<ACC>) against the two-pass form (search, then the
    streaming accumulation kernel): the 48 partial sums agree to f64 round-off for every metric -- near convergence, with
    the source drifted by more than a cell (queued 3x3x3 pass, deferred queries), with holes and outliers (clean-up pass
    by lane groups and by whole tiles) -- bitwise reproducibly, and the ICP runs land on the oracle's transform."""
    import torch

    n = 300_000
    d = syn.make_pair(n, perturb=0.4)
    h = d["h"]
    rng = np.random.default_rng(3)
    # a second, harder target/source pair: a hole, a thinned slab, outliers far outside the grid
    keep = np.linalg.norm(d["dst"] - np.array([0.5, 0.5, 0.5], np.float32), axis=1) > 0.12
    keep &= ~((d["dst"][:, 0] > 0.8) & (rng.random(n) < 0.9))
    dst_h, nrm_h = np.ascontiguousarray(d["dst"][keep]), np.ascontiguousarray(d["dst_n"][keep])
    src_h = d["src"].copy()
    far = rng.choice(n, n // 50, replace=False)
    src_h[far] += rng.normal(size=(far.size, 3)).astype(np.float32) * np.float32(20 * h)
    drift = np.eye(4, dtype=np.float32); drift[:3, 3] = np.array([1.4, -0.9, 0.7], np.float32) * np.float32(h)
    cases = [("near", d["dst"], d["dst_n"], d["src"], np.eye(4, dtype=np.float32), float(d["max_sq_dist"])),
             ("drift", d["dst"], d["dst_n"], d["src"], drift, float((3 * h) ** 2)),
             ("holes", dst_h, nrm_h, src_h, drift, float((6 * h) ** 2))]
    metrics = [(capi.METRIC_POINT_TO_POINT, 0.0, 1.0), (capi.METRIC_COMBINED, 0.0, 1.0), (capi.METRIC_COMBINED, 0.1, 1.0),
               (capi.METRIC_COMBINED, 1.0, 0.0)]
    sums = torch.zeros(capi.SUMS_LEN, dtype=torch.float64, device="cuda")
    deferred_seen = 0
    for name, dst, nrm, src, T0, max_sq in cases:
        ctxs = {}
        for acc in (0, 1):
            c = Context(0, torch.cuda.current_stream().cuda_stream)
            c.set_option("tiled", 2); c.set_option("tile_accumulation", 2 * acc)
            c.set_target(dst, nrm); c.set_source(src)
            c.find_correspondences(np.eye(4), max_sq, count=False)       # sort under the identity: T0 is a drift since the sort
            ctxs[acc] = c
        for metric, w_p2p, w_p2pl in metrics:
            p = capi.IcpParams()
            ctxs[0]._L.cilhip_icp_default_params(__import__("ctypes").byref(p))
            p.metric, p.w_p2p, p.w_p2pl, p.max_sq_dist, p.conv_tol = metric, w_p2p, w_p2pl, max_sq, 0.0
            got = {}
            for acc in (0, 1):
                runs = []
                for _ in range(3 if acc else 1):
                    ctxs[acc].icp_begin(p, T0, None)
                    ctxs[acc].icp_partial_sums(sums.data_ptr())
                    torch.cuda.synchronize()
                    runs.append(sums.cpu().numpy().copy())
                assert all(np.array_equal(runs[0], r) for r in runs[1:]), (name, metric, w_p2p)     # bitwise reproducible
                got[acc] = runs[0]
            assert got[0][0] == got[1][0] and got[1][0] > 0.5 * len(src) * (0.5 if name == "holes" else 1.0), (name, got[0][0], got[1][0])
            scale = np.abs(got[0]).max()
            assert np.abs(got[1] - got[0]).max() <= 1e-11 * scale, (name, metric, w_p2p, w_p2pl, np.abs(got[1] - got[0]).max(), scale)
        dq, dt = ctxs[1].debug_counters()
        deferred_seen += dq + dt
        del ctxs
    assert deferred_seen > 1000                         # the clean-up pass really accumulated something

    # whole runs (several Gauss-Newton steps included: the first one in the tiles, the rest streaming over the stored matches)
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f
    for metric, w_p2p, w_p2pl, steps in ((0, 0.0, 0.0, 1), (1, 0.0, 1.0, 1), (1, 0.1, 1.0, 1), (1, 0.3, 1.0, 3), (1, 1.0, 0.0, 1)):
        Ts = []
        for acc in (0, 1):
            if metric == 0:
                icp = SimplePointToPointMetricRigidICP3f(d["dst"], d["src"])
            else:
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
                icp.setPointToPointMetricWeight(w_p2p).setPointToPlaneMetricWeight(w_p2pl).setMaxNumberOfOptimizationStepIterations(steps)
            icp._ctx.set_option("tiled", 2); icp._ctx.set_option("tile_accumulation", 2 * acc)     # 2: one pass from the first iteration on
            icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
            Ts.append(icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0).estimate().getTransform().astype(np.float64))
            nc = icp.last_ncorr_
        p = orc.make_params(metric=metric, w_p2p=w_p2p, w_p2pl=w_p2pl, max_iter=8, conv_tol=0.0, max_opt_iter=steps, opt_conv_tol=1e-5,
                            max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
        r = orc.icp_run(d["dst"], d["dst_n"] if metric else None, d["src"], p)
        assert np.linalg.norm(Ts[1] - Ts[0]) <= 1e-6, (metric, w_p2p, w_p2pl, steps, np.linalg.norm(Ts[1] - Ts[0]))
        assert np.linalg.norm(Ts[1] - r["T"].astype(np.float64)) <= TOL_T and nc == r["last_ncorr"], (metric, w_p2p, w_p2pl, steps)


@pytest.mark.gpu
def test_slab_partition_two_slabs_on_one_gpu(orc, hip_lib):
    """SURVEY 8(e) partitioning B with the product engine: two spatial slabs (target slab + halo, owned source points) played
    on this one GPU, the all-reduce replaced by the sum it computes.  Same correspondences as the unsharded run (count),
    transform equal to summation round-off, identical on both "ranks"; the device-side guard stays quiet with the default
    slack and fires when the slack is a thousandth of a cell."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 400_000
    d = syn.make_pair(n, perturb=0.5)
    for w_p2p in (0.0, 0.1):
        icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
        icp.setPointToPointMetricWeight(w_p2p).setPointToPlaneMetricWeight(1.0)
        icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
        T1 = icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0).estimate().getTransform()
        nc1 = icp.last_ncorr_
        del icp
        part = distributed.SlabPartition.plan(d["dst"], d["src"], np.eye(4, dtype=np.float32), float(d["max_sq_dist"]), 2)
        engs = [distributed.HipSlabEngine(part, r, d["dst"], d["dst_n"], d["src"], 0) for r in range(2)]
        assert sum(e.n_local for e in engs) == n and all(e.ctx.n_target < 0.7 * n for e in engs)
        p = distributed.default_params(max_iter=8, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
        p.w_p2p, p.w_p2pl = w_p2p, 1.0
        for e in engs:
            e.begin(p, np.eye(4, dtype=np.float32), None)
        for _ in range(8):
            sums = engs[0].partial_sums().clone() + engs[1].partial_sums()
            for e in engs:
                e.apply_sums(sums)
        (Ta, ita, _, nca), (Tb, itb, _, ncb) = engs[0].state(), engs[1].state()
        assert np.array_equal(Ta, Tb) and ita == itb == 8 and nca == ncb == nc1
        assert np.abs(Ta.astype(np.float64) - T1.astype(np.float64)).max() <= 1e-6
        assert not engs[0].violated() and not engs[1].violated()
        del engs
    # the guard: a slack of a thousandth of a cell cannot survive the first update
    part = distributed.SlabPartition.plan(d["dst"], d["src"], np.eye(4, dtype=np.float32), float(d["max_sq_dist"]), 2, slack=1e-3 * d["h"])
    eng = distributed.HipSlabEngine(part, 0, d["dst"], d["dst_n"], d["src"], 0)
    eng.begin(p, np.eye(4, dtype=np.float32), None)
    s = eng.partial_sums()
    eng.apply_sums(s * 2.0)                # (any plausible sums: the update moves the source by a fraction of a cell)
    assert eng.violated()
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_adaptive_kernel_form_and_pacing(orc, hip_lib):
    """Large clouds pace the loop on the device's feedback: far from alignment (most octant searches unproven) an iteration
    runs as search + streaming accumulation, near alignment as one pass inside the tiles, and nothing is enqueued after
    convergence.  Whatever the mix, the run lands on the oracle's transform with the oracle's iteration count."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

    n = 1_200_000
    d = syn.make_pair(n, perturb=0.9)                         # ~0.9 cell from alignment: the first iterations are "far"
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(d["max_sq_dist"])
    Tg = icp.setMaxNumberOfIterations(40).setConvergenceTolerance(1e-5).estimate().getTransform()
    one, two = icp._ctx.last_run_forms()
    p = orc.make_params(metric=1, max_iter=40, conv_tol=1e-5, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    assert icp.getNumberOfPerformedIterations() == r["iterations"] and icp.hasConverged()
    assert np.linalg.norm(Tg.astype(np.float64) - r["T"].astype(np.float64)) <= TOL_T
    assert two >= 2 and one >= 1, (one, two)                  # both forms ran
    assert one + two <= r["iterations"] + 2, (one, two, r["iterations"])     # at most two iterations enqueued past convergence
    # a second run on the same pair starts in the form the first one ended in
    icp.setMaxNumberOfIterations(3).setConvergenceTolerance(0.0).setInitialTransform(Tg).estimate()
    assert icp._ctx.last_run_forms() == (3, 0)


@pytest.mark.gpu
def test_correspondence_weight_evaluators_vs_oracle(Context, orc, hip_lib):
    """The combined-metric classes' correspondence weight evaluators (icp_single_transform_combined_metric.hpp:11-14;
    core/common_pair_evaluators.hpp:14-27 Identity, :30-43 Unity, :46-80 RBF): per-pair f32 weight = metric weight *
    evaluator(corr.value) (transform_estimation.hpp:301-303, :330-332).  Normal equations of one step against the oracle's,
    then whole loops -- plain, with engine post-filters, with 6-D feature distances as the values, in the other search
    directions, with several Gauss-Newton steps -- against the oracle's loop."""
    from cilantro_amd.icp import (CorrespondenceSearchDirection as D, IdentityWeightEvaluator, RBFKernelWeightEvaluator,
                                  SimpleCombinedMetricAffineICP3f, SimpleCombinedMetricRigidICP3f, UnityWeightEvaluator)

    U, I, R = orc.W_UNITY, orc.W_IDENTITY, orc.W_RBF
    d = syn.make_pair(200000, perturb=0.5)
    r2 = float(d["max_sq_dist"])
    sigma = 0.4 * np.sqrt(r2)                                    # weights between exp(-3.1) and 1 over the radius
    T = syn.true_transform(d["h"], 0.1).astype(np.float32)
    ctx = Context()
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    ctx.find_correspondences(T, r2)
    g1, g2, gv = ctx.get_correspondences()
    q = orc.transform_points(T, d["src"])
    dm, sm = ctx.means()
    smt = orc.transform_points(T, sm.reshape(1, 3))[0]
    for pk, lk, w_p2p, w_p2pl in ((U, R, 0.0, 1.0), (R, R, 0.3, 1.0), (I, U, 1.0, 0.0), (R, I, 0.5, 0.7), (U, U, 0.2, 1.0)):
        ctx.set_option("point_weight_evaluator", pk); ctx.set_option("plane_weight_evaluator", lk)
        ctx.set_option("point_weight_sigma", sigma); ctx.set_option("plane_weight_sigma", 2.0 * sigma)
        Tg, AtA, Atb, cv = ctx.estimate_combined(w_p2p, w_p2pl, 1, 1e-5)
        To, AtAo, Atbo, cvo = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, w_p2p, w_p2pl, dm, smt, 1, 1e-5, orc.MODE_MIXED,
                                                    values=gv, weights=(pk, lk, sigma, 2.0 * sigma))
        scale = np.abs(AtAo).max()
        # plane terms: the same f32 per-term quantities and weights, f64 sums (order differs); point terms: the oracle forms
        # E E^T per pair in f32, the kernel in f64
        tol = 1e-9 if w_p2p == 0.0 else 2e-6
        assert np.abs(AtA - AtAo).max() <= tol * scale, (pk, lk, np.abs(AtA - AtAo).max() / scale)
        assert np.abs(Atb - Atbo).max() <= tol * np.abs(Atbo).max() + 1e-3 * tol * scale, (pk, lk)
        assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6 and cv == cvo, (pk, lk)
        # several Gauss-Newton steps over the same weighted correspondences
        Tg, _, _, cv = ctx.estimate_combined(w_p2p, w_p2pl, 3, 1e-7)
        To, _, _, cvo = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, w_p2p, w_p2pl, dm, smt, 3, 1e-7, orc.MODE_MIXED,
                                              values=gv, weights=(pk, lk, sigma, 2.0 * sigma))
        assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6 and cv == cvo, (pk, lk)
    # the weights change the estimate (the test would pass vacuously otherwise)
    ctx.set_option("point_weight_evaluator", U); ctx.set_option("plane_weight_evaluator", U)
    Tu = ctx.estimate_combined(0.0, 1.0, 1, 1e-5)[0]
    ctx.set_option("plane_weight_evaluator", R); ctx.set_option("plane_weight_sigma", sigma)
    Tr = ctx.estimate_combined(0.0, 1.0, 1, 1e-5)[0]
    assert np.linalg.norm(Tu - Tr) > 1e-6

    def evaluator(kind, s):
        return UnityWeightEvaluator() if kind == U else IdentityWeightEvaluator() if kind == I else RBFKernelWeightEvaluator(s)

    # whole loops
    dl = syn.make_pair(100000, perturb=0.5)
    r2 = float(dl["max_sq_dist"]); sigma = 0.4 * np.sqrt(r2)
    for pk, lk, w_p2p, w_p2pl, steps, frac, o2o, tiled in ((U, R, 0.0, 1.0, 1, 1.0, False, 2), (R, R, 0.2, 1.0, 2, 1.0, False, 2),
                                                           (I, I, 0.5, 0.5, 1, 1.0, False, 0), (R, U, 0.1, 1.0, 1, 0.8, True, 2)):
        icp = SimpleCombinedMetricRigidICP3f(dl["dst"], dl["dst_n"], dl["src"])
        icp._ctx.set_option("tiled", tiled)
        icp.setPointToPointMetricWeight(w_p2p).setPointToPlaneMetricWeight(w_p2pl)
        icp.setCorrespondenceWeightEvaluators(evaluator(pk, sigma), evaluator(lk, sigma))
        assert icp.pointToPlaneCorrespondenceWeightEvaluator().kind == lk
        icp.setMaxNumberOfOptimizationStepIterations(steps).setOptimizationStepConvergenceTolerance(1e-6)
        icp.correspondenceSearchEngine().setMaxDistance(r2).setInlierFraction(frac).setOneToOne(o2o)
        icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=w_p2p, w_p2pl=w_p2pl, max_iter=8, conv_tol=0.0, max_opt_iter=steps, opt_conv_tol=1e-6,
                            max_sq_dist=r2, mode=orc.MODE_MIXED, inlier_fraction=frac, one_to_one=o2o, point_weight=pk, plane_weight=lk,
                            point_sigma=sigma, plane_sigma=sigma)
        ro = orc.icp_run(dl["dst"], dl["dst_n"], dl["src"], p)
        err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
        assert err <= TOL_T and icp.last_ncorr_ == ro["last_ncorr"], (pk, lk, w_p2p, w_p2pl, steps, frac, o2o, err)
        assert icp._ctx.last_run_forms()[0] == 0                 # never the in-tile (unweighted) accumulation
    # the other search directions (list-free reverse accumulation): the value of a reverse match is its search distance too
    for direction, recip, code in ((D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
        icp = SimpleCombinedMetricRigidICP3f(dl["dst"], dl["dst_n"], dl["src"])
        icp.setPointToPointMetricWeight(0.1).setCorrespondenceWeightEvaluators(evaluator(R, sigma), evaluator(R, 1.5 * sigma))
        icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
        icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=0.1, w_p2pl=1.0, max_sq_dist=r2, max_iter=6, conv_tol=0.0, direction=code, reciprocal=recip,
                            point_weight=R, plane_weight=R, point_sigma=sigma, plane_sigma=1.5 * sigma)
        ro = orc.icp_run(dl["dst"], dl["dst_n"], dl["src"], p)
        err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
        assert err <= TOL_T and icp.last_ncorr_ == ro["last_ncorr"], (direction, recip, err)
    # 6-D feature correspondences: the evaluator reads the FEATURE distance
    dn = syn.make_pair(20000, perturb=0.5)
    h = dn["h"]; w = 0.5 * h; r6 = float((2.5 * h) ** 2)
    sn_true = orc.transform_normals(np.linalg.inv(dn["T_true"].astype(np.float64)).astype(np.float32), dn["dst_n"])
    for tiled in (0, 2):
        icp = SimpleCombinedMetricRigidICP3f(dn["dst"], dn["dst_n"], dn["src"])
        icp._ctx.set_option("tiled", tiled)
        icp.correspondenceSearchEngine().setMaxDistance(r6).setPointNormalFeatureAdaptors(sn_true, w)
        icp.setCorrespondenceWeightEvaluators(None, evaluator(R, 0.5 * np.sqrt(r6)))
        icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=r6, mode=orc.MODE_MIXED, normal_weight=w, three_cloud_metric=True,
                            plane_weight=R, plane_sigma=0.5 * np.sqrt(r6))
        ro = orc.icp_run(dn["dst"], dn["dst_n"], dn["src"], p, src_n=sn_true)
        assert np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)) <= TOL_T and icp.last_ncorr_ == ro["last_ncorr"], tiled
    # the affine combined-metric class with weight evaluators (transform_estimation.hpp:432-434, :453-455): every search direction
    for direction, recip, code in ((D.SECOND_TO_FIRST, False, 0), (D.BOTH, False, 2), (D.FIRST_TO_SECOND, False, 1)):
        for pe, le, pk, lk in ((None, evaluator(R, sigma), 0, R), (evaluator(I, 1.0), evaluator(R, 1.5 * sigma), I, R)):
            icp = SimpleCombinedMetricAffineICP3f(dl["dst"], dl["dst_n"], dl["src"])
            icp.setPointToPointMetricWeight(0.1).setCorrespondenceWeightEvaluators(pe, le)
            icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
            icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
            Tg = icp.estimate().getTransform()
            p = orc.make_params(metric=1, w_p2p=0.1, w_p2pl=1.0, max_sq_dist=r2, max_iter=5, conv_tol=0.0, direction=code, reciprocal=recip, affine=True,
                                point_weight=pk, plane_weight=lk, point_sigma=1.0, plane_sigma=sigma if pe is None else 1.5 * sigma)
            ro = orc.icp_run(dl["dst"], dl["dst_n"], dl["src"], p)
            err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
            assert err <= 1e-4 and icp.last_ncorr_ == ro["last_ncorr"], (direction, pk, lk, err)


@pytest.mark.gpu
def test_functor_weight_evaluators_through_the_callback(Context, orc, hip_lib):
    """The reference's evaluators are template arguments: ANY functor evaluator(indexInFirst, indexInSecond, value)
    (icp_single_transform_combined_metric.hpp:10-14; called at transform_estimation.hpp:303, :332).  Here such a functor runs on the
    host through cilhip_set_pair_weight_callback.  (1) functors that restate the stock classes give the stock classes' normal
    equations; (2) a functor that reads the INDICES (weight 0 for odd source indices, 1 + index/n for the others on the plane terms)
    against the oracle's estimate over the surviving correspondences with those weights folded in by linearity; (3) whole loops:
    functor = stock restated against the oracle's loop with the stock kind, and a robust (Cauchy) kernel the reference has no class
    for against a numpy restatement of the loop built from the engine's own search + the oracle's weighted-by-kind estimator
    (Identity kind over pre-weighted values)."""
    from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f, UnityWeightEvaluator

    U, I, R = orc.W_UNITY, orc.W_IDENTITY, orc.W_RBF
    d = syn.make_pair(150000, perturb=0.5)
    r2 = float(d["max_sq_dist"]); sigma = 0.4 * np.sqrt(r2)
    T = syn.true_transform(d["h"], 0.1).astype(np.float32)
    ctx = Context()
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    ctx.find_correspondences(T, r2)
    g1, g2, gv = ctx.get_correspondences()
    q = orc.transform_points(T, d["src"])
    dm, sm = ctx.means()
    smt = orc.transform_points(T, sm.reshape(1, 3))[0]
    coeff = np.float32(-0.5) / (np.float32(sigma) * np.float32(sigma))
    seen = {}

    def stock_as_functors(i1, i2, v):
        seen["n"] = len(v); seen["i1"] = i1.copy(); seen["i2"] = i2.copy(); seen["v"] = v.copy()
        return v.copy(), np.exp(coeff * v).astype(np.float32)          # Identity on the point terms, RBF on the plane terms

    # (1) against the device-evaluated stock classes and the oracle
    ctx.set_option("point_weight_evaluator", I); ctx.set_option("plane_weight_evaluator", R)
    ctx.set_option("point_weight_sigma", 1.0); ctx.set_option("plane_weight_sigma", sigma)
    Ts, AtAs, Atbs, _ = ctx.estimate_combined(0.3, 1.0, 1, 1e-5)
    ctx.set_option("point_weight_evaluator", U); ctx.set_option("plane_weight_evaluator", U)
    ctx.set_pair_weight_callback(stock_as_functors)
    Tf, AtAf, Atbf, _ = ctx.estimate_combined(0.3, 1.0, 1, 1e-5)
    # the callback saw the stored set in its stored order: the list get_correspondences returns
    assert seen["n"] == len(g1) and np.array_equal(seen["i1"], g1) and np.array_equal(seen["i2"], g2) and np.array_equal(seen["v"], gv)
    scale = np.abs(AtAs).max()
    assert np.abs(AtAf - AtAs).max() <= 1e-6 * scale and np.abs(Atbf - Atbs).max() <= 1e-6 * np.abs(Atbs).max()   # (numpy's expf against the device's)
    assert np.linalg.norm(Tf.astype(np.float64) - Ts.astype(np.float64)) < 1e-6
    To = orc.estimate_combined(d["dst"], d["dst_n"], q, g1, g2, 0.3, 1.0, dm, smt, 1, 1e-5, orc.MODE_MIXED, values=gv, weights=(I, R, 1.0, sigma))[0]
    assert np.linalg.norm(Tf.astype(np.float64) - To) < 1e-6

    # (2) weights from the indices: plane terms only; the oracle's Identity kind over "values" that ARE the weights
    n_src = len(d["src"])
    def by_index(i1, i2, v):
        w = np.where(i2 % 2 == 1, 0.0, 1.0 + i2.astype(np.float64) / n_src).astype(np.float32)
        return np.ones(len(v), np.float32), w
    ctx.set_pair_weight_callback(by_index)
    Tg, AtA, Atb, _ = ctx.estimate_combined(0.0, 1.0, 1, 1e-5)
    keep = g2 % 2 == 0
    wv = (1.0 + g2[keep].astype(np.float64) / n_src).astype(np.float32)
    To, AtAo, Atbo, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, g1[keep], g2[keep], 0.0, 1.0, dm, smt, 1, 1e-5, orc.MODE_MIXED,
                                              values=wv, weights=(U, I, 1.0, 1.0))
    scale = np.abs(AtAo).max()
    assert np.abs(AtA - AtAo).max() <= 1e-9 * scale, np.abs(AtA - AtAo).max() / scale
    assert np.abs(Atb - Atbo).max() <= 1e-9 * np.abs(Atbo).max() + 1e-12 * scale
    assert np.linalg.norm(Tg.astype(np.float64) - To) < 1e-6
    # a callback that raises: the exception surfaces from the library call it happened in
    def broken(i1, i2, v):
        raise ValueError("evaluator failed")
    ctx.set_pair_weight_callback(broken)
    with pytest.raises(ValueError, match="evaluator failed"):
        ctx.estimate_combined(0.0, 1.0, 1, 1e-5)
    # ... and a sharded run (partial sums of one shard: no host in the loop) refuses it instead of reading stale tables
    ctx.set_pair_weight_callback(by_index)
    from cilantro_amd import capi, distributed
    with pytest.raises(capi.CilhipError):
        ctx.icp_begin(distributed.default_params(capi.METRIC_COMBINED, max_sq_dist=r2, max_iter=2), np.eye(4, dtype=np.float32), sm)
    ctx.set_pair_weight_callback(None)
    Tu = ctx.estimate_combined(0.0, 1.0, 1, 1e-5)[0]
    assert np.linalg.norm(Tu - Tg) > 1e-7                                # (back to unit weights; the index weights did change the estimate)

    # (3) whole loops through the mirror class
    dl = syn.make_pair(60000, perturb=0.5)
    r2 = float(dl["max_sq_dist"]); sigma = 0.4 * np.sqrt(r2)
    coeff = np.float32(-0.5) / (np.float32(sigma) * np.float32(sigma))
    rbf = lambda i1, i2, v: np.exp(coeff * v).astype(np.float32)
    for frac, o2o in ((1.0, False), (0.8, True)):
        icp = SimpleCombinedMetricRigidICP3f(dl["dst"], dl["dst_n"], dl["src"])
        icp.setPointToPointMetricWeight(0.2).setPointToPlaneMetricWeight(1.0)
        icp.setCorrespondenceWeightEvaluators(UnityWeightEvaluator(), rbf)          # a stock object and a plain callable side by side
        icp.correspondenceSearchEngine().setMaxDistance(r2).setInlierFraction(frac).setOneToOne(o2o)
        icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
        Tg = icp.estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=0.2, w_p2pl=1.0, max_iter=8, conv_tol=0.0, max_opt_iter=1, opt_conv_tol=1e-5, max_sq_dist=r2,
                            mode=orc.MODE_MIXED, inlier_fraction=frac, one_to_one=o2o, point_weight=U, plane_weight=R, point_sigma=1.0, plane_sigma=sigma)
        ro = orc.icp_run(dl["dst"], dl["dst_n"], dl["src"], p)
        err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
        assert err <= TOL_T and icp.getNumberOfPerformedIterations() == ro["iterations"], (frac, o2o, err)
    # a robust kernel the reference has no class for: w = 1 / (1 + value / c^2) on both terms; the loop restated with the engine's own
    # search and the oracle's estimator fed the weights as Identity-kind values
    c2 = np.float32(0.25 * r2)
    cauchy = lambda i1, i2, v: (np.float32(1.0) / (np.float32(1.0) + v / c2)).astype(np.float32)
    icp = SimpleCombinedMetricRigidICP3f(dl["dst"], dl["dst_n"], dl["src"])
    icp.setPointToPointMetricWeight(0.1).setPointToPlaneMetricWeight(1.0).setCorrespondenceWeightEvaluators(cauchy, cauchy)
    icp.correspondenceSearchEngine().setMaxDistance(r2)
    icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
    Tg = icp.estimate().getTransform()
    ref = Context()
    ref.set_target(dl["dst"], dl["dst_n"]); ref.set_source(dl["src"])
    dm, sm = ref.means()
    Tc = np.eye(4, dtype=np.float32)
    for _ in range(5):
        ref.find_correspondences(Tc, r2)
        g1, g2, gv = ref.get_correspondences()
        qq = orc.transform_points(Tc, dl["src"])
        smt = orc.transform_points(Tc, sm.reshape(1, 3))[0]
        w = cauchy(g1, g2, gv)
        dT = orc.estimate_combined(dl["dst"], dl["dst_n"], qq, g1, g2, 0.1, 1.0, dm, smt, 1, 1e-5, orc.MODE_MIXED, values=w, weights=(I, I, 1.0, 1.0))[0]
        step = np.eye(4)                                                     # rotation() polish + transform_ = tform_iter * transform_ (:207-213)
        step[:3, :3] = orc.nearest_rotation(np.asarray(dT[:3, :3], np.float64)); step[:3, 3] = dT[:3, 3]
        Tc = (step @ Tc.astype(np.float64)).astype(np.float32)
    err = np.linalg.norm(Tg.astype(np.float64) - Tc.astype(np.float64))
    assert err <= TOL_T, err


def test_warm_started_iterations_find_the_same_matches(Context, orc, hip_lib):
    """From the second iteration on (near alignment) the loop runs a per-lane kernel that starts every search from the
    previous iteration's match: a real target point bounds the search, and a query nearer to it than half its distance to
    any other target point is settled without looking at a neighbour.  Same matches, hence the same loop: against the
    tiled forms (option warm_start = 0) and the oracle, on a uniform cloud, a cloud with holes and exact duplicate target
    points (nearest-other distance 0: never settled by the shortcut), from near and from farther away, both metrics."""
    rng = np.random.default_rng(5)
    base = syn.make_pair(1_200_000, perturb=0.5)
    h = base["h"]
    dst, dst_n, src = base["dst"], base["dst_n"], base["src"]
    # holes + duplicates: drop two slabs of the target, then repeat a few thousand target points exactly
    keep = ~(((dst[:, 0] > 0.3) & (dst[:, 0] < 0.36)) | ((dst[:, 2] > 0.7) & (dst[:, 2] < 0.73)))
    dup = rng.choice(np.nonzero(keep)[0], 5000, replace=False)
    dst_h = np.ascontiguousarray(np.concatenate([dst[keep], dst[dup]])); dst_hn = np.ascontiguousarray(np.concatenate([dst_n[keep], dst_n[dup]]))
    far = syn.make_pair(1_200_000, perturb=0.9)
    cases = (("uniform", dst, dst_n, src, base["max_sq_dist"], base["T_true"]), ("holes+duplicates", dst_h, dst_hn, src, base["max_sq_dist"], base["T_true"]),
             ("far start", far["dst"], far["dst_n"], far["src"], far["max_sq_dist"], far["T_true"]))
    for name, D, N, S, r2, Tt in cases:
        for metric, w_p2p in ((capi.METRIC_COMBINED, 0.0), (capi.METRIC_COMBINED, 0.1), (capi.METRIC_POINT_TO_POINT, 0.0)):
            res = {}
            for warm in (0, 2, 1):
                ctx = Context()
                ctx.set_option("warm_start", warm)
                ctx.set_option("tiled", 2)                    # (the cloud with holes would otherwise be left to the per-lane kernels)
                ctx.set_target(D, N); ctx.set_source(S)
                p = capi.IcpParams()
                ctx._L.cilhip_icp_default_params(C.byref(p))
                p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = metric, w_p2p, float(r2), 10, 0.0
                runs = [ctx.icp_run(p) for _ in range(2)]
                T = [np.array(r.T[:], np.float32) for r in runs]
                assert np.array_equal(T[0], T[1]) or warm == 1, (name, metric, warm)        # bitwise reproducible (fixed forms)
                res[warm] = (T[0].astype(np.float64), int(runs[0].last_ncorr), ctx.last_warm_iterations())
                ctx.close()
            assert res[0][2] == 0 and res[2][2] == 9, (name, res[0][2], res[2][2])
            if name == "uniform":
                assert res[1][2] >= 6, (name, res[1][2])                                   # the adaptive loop takes it when near
            for warm in (2, 1):
                assert res[warm][1] == res[0][1], (name, metric, w_p2p, warm, res[warm][1], res[0][1])      # same correspondences
                assert np.abs(res[warm][0] - res[0][0]).max() <= 2e-7, (name, metric, w_p2p, warm, np.abs(res[warm][0] - res[0][0]).max())
        # ... and the oracle's loop
        po = orc.make_params(metric=1, max_iter=10, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
        ro = orc.icp_run(D, N, S, po)
        ctx = Context(); ctx.set_option("warm_start", 2); ctx.set_option("tiled", 2); ctx.set_target(D, N); ctx.set_source(S)
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
        p.max_sq_dist, p.max_iter, p.conv_tol = float(r2), 10, 0.0
        rg = ctx.icp_run(p)
        Tg = np.array(rg.T[:], np.float32).reshape(4, 4).T
        assert np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)) <= TOL_T and int(rg.last_ncorr) == ro["last_ncorr"], name
        ctx.close()


@pytest.mark.gpu
def test_warm_start_stress_sweep(hip_lib):
    """tools/warm_stress.py: 168 loop comparisons (uniform and surface-like clouds, 70k-1.5M points, three metrics, one and two
    Gauss-Newton steps, fixed and tolerance-gated iteration counts, tiled and per-lane first iterations) of the adaptive and the
    forced warm-started forms against the loop without them: same iteration counts, same correspondence counts, same transform."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "warm_stress.py")], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_transform_ransac3f_vs_oracle(orc, hip_lib):
    """RigidTransformRANSACEstimator3f (model_estimation/ransac_transform_estimator.hpp; SURVEY section 2 "next tier") through the
    Python mirror: inlier counts of arbitrary transforms bit-exact, the closed-form fit to f32 round-off, whole runs with explicit
    samples -- same iteration count, same winner, inlier sets bit-exact for the product's own model and equal to the oracle's run up
    to pairs within round-off of the threshold -- correspondences as index lists, library-drawn samples, edge cases."""
    from cilantro_amd.model_estimation import RigidTransformRANSACEstimator3f

    rng = np.random.default_rng(31)
    n = 300_007
    src = rng.random((n, 3)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = syn.rot_xyz(0.25, -0.4, 0.1); T[:3, 3] = [0.2, 0.1, -0.3]
    dst = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 5e-4, (n, 3))).astype(np.float32)
    bad = rng.random(n) < 0.45
    dst[bad] = (rng.random((int(bad.sum()), 3)) * 2.0 - 0.5).astype(np.float32)
    thr = 2e-3
    te = RigidTransformRANSACEstimator3f(dst, src)
    # (1) scoring
    Ts = np.tile(np.eye(4, dtype=np.float32), (131, 1, 1))
    for k in range(131):
        Ts[k, :3, :3] = syn.rot_xyz(*(rng.normal(0, 0.3, 3))).astype(np.float32)
        Ts[k, :3, 3] = rng.normal(0, 0.2, 3).astype(np.float32)
    Ts[0] = T.astype(np.float32)
    Ts[5, 0, 0] = np.nan
    got = te.countInliers(Ts, thr)
    want = np.array([orc.transform_count_inliers(dst, src, t, thr) for t in Ts])
    assert np.array_equal(got, want) and got[0] > 0.5 * n
    # (2) the closed-form fit over all pairs of the inlier set
    good = np.nonzero(~bad)[0]
    tg = RigidTransformRANSACEstimator3f(dst, src, correspondences=(good, good)).estimateModel()
    to = orc.transform_fit(dst, src, good.astype(np.uint32), mode=orc.MODE_MIXED)
    assert np.abs(tg - to).max() <= 2e-6 and np.abs(tg.astype(np.float64) - T).max() < 1e-4
    # (3) full runs
    for max_iter, target, re_est in ((100, n // 2, True), (100, n // 2, False), (200, n, True), (3, 10, True), (70, int(0.52 * n), True)):
        samples = rng.integers(0, n, (max_iter, 3)).astype(np.uint32)
        te = (RigidTransformRANSACEstimator3f(dst, src).setMaxInlierResidual(thr).setTargetInlierCount(target)
              .setMaxNumberOfIterations(max_iter).setReEstimationStep(re_est).setSamples(samples))
        Tg = te.estimate().getModel()
        To, reso, inlo, ito, haveo = orc.transform_ransac(dst, src, samples, thr, target, re_estimate=re_est, mode=orc.MODE_MIXED)
        assert te.getNumberOfPerformedIterations() == ito, (max_iter, target, te.getNumberOfPerformedIterations(), ito)
        assert np.abs(Tg - To).max() <= 5e-6, (max_iter, target, np.abs(Tg - To).max())
        res_chk = orc.transform_residuals(dst, src, Tg)
        assert np.array_equal(te.getModelResiduals().view(np.uint32), res_chk.view(np.uint32))
        inl_chk = np.nonzero(res_chk <= np.float32(thr))[0]
        assert np.array_equal(te.getModelInliers(), inl_chk)
        assert len(np.setxor1d(inl_chk, inlo)) <= max(3, int(2e-4 * n)), (len(inl_chk), len(inlo))
        assert te.targetInlierCountAchieved() == (len(inl_chk) >= min(target, n))
    # (4) library-drawn samples: deterministic in the seed, finds the motion
    a = RigidTransformRANSACEstimator3f(dst, src).setMaxInlierResidual(thr).setSeed(3).estimate()
    b = RigidTransformRANSACEstimator3f(dst, src).setMaxInlierResidual(thr).setSeed(3).estimate()
    assert np.array_equal(a.getModel(), b.getModel()) and np.array_equal(a.getModelInliers(), b.getModelInliers())
    assert np.abs(a.getModel().astype(np.float64) - T).max() < 1e-4 and a.getNumberOfInliers() >= int(0.5 * n)
    # (5) edge cases: no accepted model (threshold 0 on noisy pairs); tiny inputs
    e = RigidTransformRANSACEstimator3f(dst[bad][:1000], src[bad][:1000]).setMaxInlierResidual(0.0).setMaxNumberOfIterations(20).setSeed(1).setReEstimationStep(False).estimate()
    assert e.getNumberOfPerformedIterations() == 20 and e.getNumberOfInliers() == 0 and np.array_equal(e.getModel(), np.eye(4, dtype=np.float32))
    for npts in (0, 1, 2, 3):
        t = RigidTransformRANSACEstimator3f(dst[:npts].copy(), src[:npts].copy()).setMaxInlierResidual(thr).setMaxNumberOfIterations(4).setSeed(2).estimate()
        assert t.getNumberOfPerformedIterations() <= 4 and t.getNumberOfInliers() <= npts


@pytest.mark.gpu
def test_feature_adaptors_directions_affine_colour_vs_oracle(Context, orc, hip_lib):
    """The 6-D feature adaptors beyond the default direction (common_transformable_feature_adaptors.hpp):
    PointNormalFeaturesAdaptor3f in FIRST_TO_SECOND / BOTH (+ reciprocity), under AFFINE transforms (the adaptor's non-rigid branch:
    normals through L^-T, renormalised, :112-124), and PointColorFeaturesAdaptor3f (colours untouched by the transform, :164-252).
    Correspondence lists element for element against the oracle's exhaustive 6-D search (itself pinned on the reference's
    nanoflann for DIM = 6), loops against the oracle's loops."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D
    from cilantro_amd.icp import SimpleCombinedMetricAffineICP3f, SimpleCombinedMetricRigidICP3f

    n = 12000
    d = syn.make_pair(n, perturb=0.5)
    h = d["h"]
    dst, dst_n, src = d["dst"], d["dst_n"], d["src"]
    Ti = np.linalg.inv(d["T_true"].astype(np.float64)).astype(np.float32)
    src_n = orc.transform_normals(Ti, dst_n)
    w = 0.6 * h
    r2 = float((2.5 * h) ** 2)
    dst6 = orc.point_normal_features(dst, dst_n, w)
    src6 = orc.point_normal_features(src, src_n, w)
    T_rigid = d["T_true"].astype(np.float32).copy(); T_rigid[:3, 3] += np.float32(0.2 * h)
    T_aff = T_rigid.copy(); T_aff[:3, :3] = (T_aff[:3, :3].astype(np.float64) @ (np.eye(3) + np.array([[0.02, 0.01, 0], [0, -0.03, 0.015], [0.01, 0, 0.025]]))).astype(np.float32)

    def lists_equal(g, o):
        return len(g[0]) == len(o[0]) and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]) and np.array_equal(g[2].view(np.uint32), o[2].view(np.uint32))

    # (1) + (2): correspondence lists, every direction, rigid and affine transforms, both search kernels
    for tiled in (0, 2):
        for T, mode, tmode in ((T_rigid, 0, 0), (T_aff, 1, 1)):
            q6 = orc.transform_features6(T, src6, mode)
            for direction, recip, code in ((D.SECOND_TO_FIRST, False, 0), (D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
                ctx = Context()
                ctx.set_option("tiled", tiled); ctx.set_option("transform_mode", tmode)
                ctx.set_target(dst, dst_n); ctx.set_source(src, src_n)
                ctx.set_option("symmetric_metric", 0); ctx.set_option("feature_normal_weight", w)
                ctx.set_option("search_direction", code); ctx.set_option("require_reciprocality", 1 if recip else 0)
                ng = ctx.find_correspondences(T, r2)
                got = ctx.get_correspondences()
                ctx.close()
                want = orc.find_correspondences_feat6_dir(dst6, q6, r2, code, recip)
                assert ng == len(want[0]) and len(want[0]) > 0.5 * n, (tiled, mode, direction, ng, len(want[0]))
                assert lists_equal(got, want), (tiled, mode, direction, recip)
    # (3) colours: source colours = the matched target's colour + noise; the colour part does not move with T
    rng = np.random.default_rng(9)
    dst_c = rng.random((n, 3)).astype(np.float32)
    src_c = np.clip(dst_c + rng.normal(0, 0.05, (n, 3)), 0, 1).astype(np.float32)
    wc = 0.8 * h
    dstc6 = orc.point_normal_features(dst, dst_c, wc)          # (p, w c): the same 6-D layout
    srcc6 = orc.point_normal_features(src, src_c, wc)
    for tiled in (0, 2):
        for direction, recip, code in ((D.SECOND_TO_FIRST, False, 0), (D.BOTH, False, 2), (D.FIRST_TO_SECOND, False, 1)):
            icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
            icp._ctx.set_option("tiled", tiled)
            eng = icp.correspondenceSearchEngine()
            eng.setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip).setPointColorFeatureAdaptors(dst_c, src_c, wc)
            eng.findCorrespondences(T_rigid)
            got = eng.getCorrespondences()
            want = orc.find_correspondences_feat6_dir(dstc6, orc.transform_features6(T_rigid, srcc6, 2), r2, code, recip)
            assert lists_equal(got, want) and len(want[0]) > 0.5 * n, (tiled, direction)
    # the colour loop against the oracle's pieces driven from here (transformFeatures -> findCorrespondences -> updateEstimate)
    icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
    icp.correspondenceSearchEngine().setMaxDistance(r2).setPointColorFeatureAdaptors(dst_c, src_c, wc)
    Tg = icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0).estimate().getTransform()
    p = orc.make_params(metric=1, max_iter=1, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED)
    To = np.eye(4, dtype=np.float32)
    for _ in range(5):
        di, si, dv = orc.find_correspondences_feat6_dir(dstc6, orc.transform_features6(To, srcc6, 2), r2, 0)
        To, _ = orc.icp_update(dst, dst_n, src, To, di, si, p)
    assert np.linalg.norm(Tg.astype(np.float64) - To.astype(np.float64)) <= TOL_T and icp.last_ncorr_ == len(di)
    # (4) loops: point+normal features in the pair-list directions (rigid) and under the affine class
    for direction, recip, code in ((D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
        icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
        icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip).setPointNormalFeatureAdaptors(src_n, w)
        Tg = icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0).estimate().getTransform()
        p = orc.make_params(metric=1, max_iter=5, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED, normal_weight=w, three_cloud_metric=True,
                            direction=code, reciprocal=recip)
        ro = orc.icp_run(dst, dst_n, src, p, src_n=src_n)
        err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
        assert err <= TOL_T and icp.last_ncorr_ == ro["last_ncorr"], (direction, recip, err, icp.last_ncorr_, ro["last_ncorr"])
    for direction, code in ((D.SECOND_TO_FIRST, 0), (D.BOTH, 2)):
        icp = SimpleCombinedMetricAffineICP3f(dst, dst_n, src)
        icp.setPointToPointMetricWeight(0.1)
        icp.correspondenceSearchEngine().setMaxDistance(r2).setSearchDirection(direction).setPointNormalFeatureAdaptors(src_n, w)
        Tg = icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0).estimate().getTransform()
        p = orc.make_params(metric=1, w_p2p=0.1, max_iter=5, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED, normal_weight=w, three_cloud_metric=True,
                            direction=code, affine=True)
        ro = orc.icp_run(dst, dst_n, src, p, src_n=src_n)
        err = np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64))
        assert err <= 1e-4 and icp.last_ncorr_ == ro["last_ncorr"], (direction, err, icp.last_ncorr_, ro["last_ncorr"])


@pytest.mark.gpu
def test_point_normal_color_features_9d_vs_oracle(Context, orc, hip_lib):
    """PointNormalColorFeaturesAdaptor3f (common_transformable_feature_adaptors.hpp:255-343): 9-D features (p, wn n, wc c), the normal
    part following the transform (rigid: L; affine: L^-T renormalised), the colour part not.  Correspondence lists in every search
    direction, rigid and affine transforms, both search kernels, element for element against the oracle's exhaustive 9-D search (pinned
    on the reference's nanoflann for DIM = 9); a rigid loop against the oracle's pieces driven from here."""
    from cilantro_amd.icp import CorrespondenceSearchDirection as D
    from cilantro_amd.icp import SimpleCombinedMetricAffineICP3f, SimpleCombinedMetricRigidICP3f

    n = 12000
    d = syn.make_pair(n, perturb=0.5)
    h = d["h"]
    dst, dst_n, src = d["dst"], d["dst_n"], d["src"]
    Ti = np.linalg.inv(d["T_true"].astype(np.float64)).astype(np.float32)
    src_n = orc.transform_normals(Ti, dst_n)
    rng = np.random.default_rng(19)
    dst_c = rng.random((n, 3)).astype(np.float32)
    src_c = np.clip(dst_c + rng.normal(0, 0.05, (n, 3)), 0, 1).astype(np.float32)
    wn, wc = 0.6 * h, 0.8 * h
    r2 = float((2.5 * h) ** 2)
    dst9 = orc.point_normal_color_features(dst, dst_n, dst_c, wn, wc)
    src9 = orc.point_normal_color_features(src, src_n, src_c, wn, wc)
    T_rigid = d["T_true"].astype(np.float32).copy(); T_rigid[:3, 3] += np.float32(0.2 * h)
    T_aff = T_rigid.copy(); T_aff[:3, :3] = (T_aff[:3, :3].astype(np.float64) @ (np.eye(3) + np.array([[0.02, 0.01, 0], [0, -0.03, 0.015], [0.01, 0, 0.025]]))).astype(np.float32)

    def lists_equal(g, o):
        return len(g[0]) == len(o[0]) and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]) and np.array_equal(g[2].view(np.uint32), o[2].view(np.uint32))

    for tiled in (0, 2):
        for T, mode, cls in ((T_rigid, 0, SimpleCombinedMetricRigidICP3f), (T_aff, 1, SimpleCombinedMetricAffineICP3f)):
            q9 = orc.transform_features9(T, src9, mode)
            for direction, recip, code in ((D.SECOND_TO_FIRST, False, 0), (D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
                icp = cls(dst, dst_n, src)
                icp._ctx.set_option("tiled", tiled)
                eng = icp.correspondenceSearchEngine()
                eng.setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
                eng.setPointNormalColorFeatureAdaptors(src_n, dst_c, src_c, wn, wc)
                eng.findCorrespondences(T)
                got = eng.getCorrespondences()
                want = orc.find_correspondences_feat9_dir(dst9, q9, r2, code, recip)
                assert len(want[0]) > 0.5 * n and lists_equal(got, want), (tiled, mode, direction, recip, len(got[0]), len(want[0]))
    # normal weight 0, colour weight > 0: still a 9-D search in BOTH halves of the other directions (the reverse half used to take the
    # normal weight for "no features" and ran on the points alone), rigid and affine (Eigen's normalized() leaves the zero vector as it is)
    dst9c = orc.point_normal_color_features(dst, dst_n, dst_c, 0.0, wc)
    src9c = orc.point_normal_color_features(src, src_n, src_c, 0.0, wc)
    for T, mode, cls in ((T_rigid, 0, SimpleCombinedMetricRigidICP3f), (T_aff, 1, SimpleCombinedMetricAffineICP3f)):
        q9c = orc.transform_features9(T, src9c, mode)
        for direction, recip, code in ((D.SECOND_TO_FIRST, False, 0), (D.FIRST_TO_SECOND, False, 1), (D.BOTH, False, 2), (D.BOTH, True, 2)):
            icp = cls(dst, dst_n, src)
            eng = icp.correspondenceSearchEngine()
            eng.setMaxDistance(r2).setSearchDirection(direction).setRequireReciprocality(recip)
            eng.setPointNormalColorFeatureAdaptors(src_n, dst_c, src_c, 0.0, wc)
            eng.findCorrespondences(T)
            got = eng.getCorrespondences()
            want = orc.find_correspondences_feat9_dir(dst9c, q9c, r2, code, recip)
            plain = orc.find_correspondences_feat9_dir(orc.point_normal_color_features(dst, dst_n, dst_c, 0.0, 0.0), orc.transform_features9(T, orc.point_normal_color_features(src, src_n, src_c, 0.0, 0.0), mode), r2, code, recip)
            assert lists_equal(got, want), ("wn = 0", mode, direction, recip, len(got[0]), len(want[0]))
            assert not lists_equal(want, plain)      # (the colours do decide matches here: a search on the points alone is told apart)
    # the matches differ from the 6-D point+normal adaptor's on a good share of the queries (a dropped part would go unnoticed otherwise)
    w9 = orc.find_correspondences_feat9_dir(dst9, orc.transform_features9(T_rigid, src9, 0), float("inf"), 0)
    w6 = orc.find_correspondences_feat6_dir(dst9[:, :6].copy(), orc.transform_features6(T_rigid, src9[:, :6].copy(), 0), float("inf"), 0)
    assert np.mean(w9[0] != w6[0]) > 0.002
    # a rigid loop: transformFeatures -> findCorrespondences -> updateEstimate, five times
    icp = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
    icp.correspondenceSearchEngine().setMaxDistance(r2).setPointNormalColorFeatureAdaptors(src_n, dst_c, src_c, wn, wc)
    Tg = icp.setMaxNumberOfIterations(5).setConvergenceTolerance(0.0).estimate().getTransform()
    p = orc.make_params(metric=1, max_iter=1, conv_tol=0.0, max_sq_dist=r2, mode=orc.MODE_MIXED)
    To = np.eye(4, dtype=np.float32)
    for _ in range(5):
        di, si, dv = orc.find_correspondences_feat9_dir(dst9, orc.transform_features9(To, src9, 0), r2, 0)
        To, _ = orc.icp_update(dst, dst_n, src, To, di, si, p)
    assert np.linalg.norm(Tg.astype(np.float64) - To.astype(np.float64)) <= TOL_T and icp.last_ncorr_ == len(di)
    # switching the kind back on the same engine rebuilds what depends on it (the source's grid carries the reverse search's features)
    eng = icp.correspondenceSearchEngine()
    eng.setSearchDirection(D.FIRST_TO_SECOND).setPointNormalFeatureAdaptors(src_n, wn)
    eng.findCorrespondences(T_rigid)
    got = eng.getCorrespondences()
    dst6 = orc.point_normal_features(dst, dst_n, wn)
    want = orc.find_correspondences_feat6_dir(dst6, orc.transform_features6(T_rigid, orc.point_normal_features(src, src_n, wn), 0), r2, 1)
    assert lists_equal(got, want)


def test_combined_metric_combiner_two_correspondence_sets_vs_oracle(Context, orc, hip_lib):
    """CorrespondenceSearchCombinedMetricCombiner (registration/correspondence_search_combined_metric_combiner.hpp:8-81): the
    combined metric's point-to-point terms from ONE engine's correspondences, its point-to-plane terms from ANOTHER's -- here a
    tighter radius and the inlier-fraction + one-to-one post-filters on the point engine, the plain search on the plane engine.
    One estimate (cilhip_estimate_combined_two_sets, several Gauss-Newton steps) and the whole loop (cilhip_icp_run_two_sets)
    against the oracle's two-set restatement of transform_estimation.hpp:237-367 driven with the oracle's own searches; the
    same engine twice equals the single-engine class."""
    from cilantro_amd.icp import CombinedMetricRigidICP3f, CorrespondenceSearchCombinedMetricCombiner, SimpleCombinedMetricRigidICP3f

    d = syn.make_pair(60000, perturb=0.6)
    dst, dst_n, src, h = d["dst"], d["dst_n"], d["src"], d["h"]
    r_pt, r_pl = np.float32((1.1 * h) ** 2), np.float32((2.0 * h) ** 2)
    w_pt, w_pl = 0.35, 1.0
    make = CorrespondenceSearchCombinedMetricCombiner.make_engine
    e_pt = make(dst, dst_n, src); e_pt.setMaxDistance(r_pt).setInlierFraction(0.8).setOneToOne(True)
    e_pl = make(dst, dst_n, src); e_pl.setMaxDistance(r_pl)
    comb = CorrespondenceSearchCombinedMetricCombiner(e_pt, e_pl)
    tree = orc.KDTree(dst, use_ref=orc.ref_available())

    def oracle_sets(T):
        q = orc.transform_points(T, src)
        a1, a2, av = tree.find_correspondences(q, float(r_pt))
        a1, a2, av = orc.filter_fraction(a1, a2, av, 0.8)
        a1, a2, av = orc.filter_one_to_one(a1, a2, av)
        b1, b2, bv = tree.find_correspondences(q, float(r_pl))
        return (a1, a2), (b1, b2)

    # (1) the two lists, then one estimate with three Gauss-Newton steps
    T0 = np.eye(4, dtype=np.float32)
    comb.findCorrespondences(T0)
    (a1, a2), (b1, b2) = oracle_sets(T0)
    g1 = comb.getPointToPointCorrespondences(); g2 = comb.getPointToPlaneCorrespondences()
    assert np.array_equal(g1[0], a1) and np.array_equal(g1[1], a2) and np.array_equal(g2[0], b1) and np.array_equal(g2[1], b2)
    assert 0 < len(a1) < len(b1)
    T = np.zeros(16, np.float32); cv = C.c_int(0)
    L = e_pt._ctx._L
    e_pt._ctx._ck(L.cilhip_estimate_combined_two_sets(e_pt._ctx._h, e_pl._ctx._h, w_pt, w_pl, 3, 0.0, T.ctypes.data_as(C.c_void_p), C.byref(cv)))
    po = orc.make_params(metric=1, w_p2p=w_pt, w_p2pl=w_pl, max_iter=1, conv_tol=0.0, max_sq_dist=float(r_pl), mode=orc.MODE_MIXED)
    po.max_opt_iter, po.opt_conv_tol = 3, 0.0
    To, _ = orc.icp_update_two_sets(dst, dst_n, src, T0, a1, a2, b1, b2, po)
    # (the estimate is the un-composed step: compare it through the composed update of the oracle with an identity start, whose
    #  polish only removes round-off)
    assert np.linalg.norm(T.reshape(4, 4).T.astype(np.float64) - To.astype(np.float64)) <= 2e-6

    # (2) the loop
    icp = CombinedMetricRigidICP3f(comb)
    icp.setPointToPointMetricWeight(w_pt).setPointToPlaneMetricWeight(w_pl).setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
    Tg = icp.estimate().getTransform()
    po.max_opt_iter, po.opt_conv_tol = 1, 1e-5
    To = np.eye(4, dtype=np.float32)
    for _ in range(6):
        (a1, a2), (b1, b2) = oracle_sets(To)
        To, dn = orc.icp_update_two_sets(dst, dst_n, src, To, a1, a2, b1, b2, po)
    err = float(np.linalg.norm(Tg.astype(np.float64) - To.astype(np.float64)))
    assert err <= TOL_T and icp.getNumberOfPerformedIterations() == 6, err
    assert abs(float(icp.getLastUpdateNorm()) - dn) <= 1e-6
    # a weight of zero on one side takes that engine's set out of the estimate altogether
    icp0 = CombinedMetricRigidICP3f(comb).setPointToPointMetricWeight(0.0).setPointToPlaneMetricWeight(1.0).setMaxNumberOfIterations(3).setConvergenceTolerance(0.0)
    ref = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
    ref.correspondenceSearchEngine().setMaxDistance(r_pl)
    ref.setPointToPointMetricWeight(0.0).setMaxNumberOfIterations(3).setConvergenceTolerance(0.0)
    assert np.linalg.norm(icp0.estimate().getTransform().astype(np.float64) - ref.estimate().getTransform().astype(np.float64)) <= 2e-6

    # (3) the same engine twice = the single-engine class (:33-43)
    same = CorrespondenceSearchCombinedMetricCombiner(e_pl, e_pl)
    icp2 = CombinedMetricRigidICP3f(same).setPointToPointMetricWeight(w_pt).setPointToPlaneMetricWeight(w_pl).setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
    ref2 = SimpleCombinedMetricRigidICP3f(dst, dst_n, src)
    ref2.correspondenceSearchEngine().setMaxDistance(r_pl)
    ref2.setPointToPointMetricWeight(w_pt).setPointToPlaneMetricWeight(w_pl).setMaxNumberOfIterations(5).setConvergenceTolerance(0.0)
    assert np.linalg.norm(icp2.estimate().getTransform().astype(np.float64) - ref2.estimate().getTransform().astype(np.float64)) <= 2e-6
    # engines that searched under different transforms are refused
    e_pt.findCorrespondences(T0); e_pl.findCorrespondences(To)
    with pytest.raises(Exception):
        e_pt._ctx._ck(L.cilhip_estimate_combined_two_sets(e_pt._ctx._h, e_pl._ctx._h, w_pt, w_pl, 1, 0.0, T.ctypes.data_as(C.c_void_p), C.byref(cv)))


def test_grid_parameters_never_change_a_result(hip_lib, orc):
    """The grid is a search structure, not part of the answer: whatever cell size the build picks (options cell_occupancy,
    refined_occupancy_factor; the refined grids of surface-like targets) the correspondences equal the reference's nanoflann index for
    index (up to exactly equidistant candidates: then an equally near point) and the loop ends on the same transform.  Clouds: the
    reference's sensor frames (a surface: the grid IS refined) under the identity, and a volumetric cloud with doubled points."""
    from cilantro_amd.icp import Context

    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_full.npz"))
    p1, n1, p2 = f["p1"], f["n1"], f["p2"]
    keep = p1[:, 0] > -0.4
    rng = np.random.default_rng(8)
    b = syn.make_pair(150_000, perturb=0.4)
    dup = rng.choice(len(b["dst"]), 2000, replace=False)
    clouds = (("frames", np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep]), np.ascontiguousarray(p2), np.float32(0.02 ** 2)),
              ("volumetric + duplicates", np.ascontiguousarray(np.concatenate([b["dst"], b["dst"][dup]])),
               np.ascontiguousarray(np.concatenate([b["dst_n"], b["dst_n"][dup]])), b["src"], np.float32(b["max_sq_dist"])))
    I = np.eye(4, dtype=np.float32)
    for name, D, N, S, r2 in clouds:
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        o1, o2, ov = tree.find_correspondences(S, float(r2))
        oi = np.full(len(S), -1, np.int64); oi[o2] = o1
        od = np.full(len(S), np.inf, np.float32); od[o2] = ov
        Ts, cells = [], []
        for opts in ((), (("refined_occupancy_factor", 1),), (("refined_occupancy_factor", 8),), (("cell_occupancy", 0.5),), (("cell_occupancy", 6),),
                     (("cell_occupancy", 40), ("refined_occupancy_factor", 2))):
            ctx = Context()
            for k, v in opts:
                ctx.set_option(k, v)
            ctx.set_target(D, N); ctx.set_source(S)
            cells.append(float(ctx.grid_info().cell))
            ctx.find_correspondences(I, float(r2), count=False)
            gi, gd = ctx.get_nn()
            gi = gi.astype(np.int64); gi[gi == capi.NONE_IDX] = -1
            assert np.array_equal(gi >= 0, oi >= 0), (name, opts)
            m = gi >= 0
            assert np.array_equal(gd[m].view(np.uint32), od[m].view(np.uint32)), (name, opts)        # the same distances, bit for bit
            diff = np.nonzero(gi != oi)[0]                                                             # different index: only an exact tie
            assert all(np.array_equal(D[gi[i]], D[oi[i]]) or gd[i] == od[i] for i in diff[:200]), (name, opts)
            p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
            p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), 8, 0.0
            r = ctx.icp_run(p)
            Ts.append(np.array(r.T[:], np.float64))
            ctx.close()
        assert len(set(round(c, 7) for c in cells)) >= 4, cells          # (the options did change the grid)
        for T in Ts[1:]:
            assert float(np.abs(T - Ts[0]).max()) <= 2e-6, (name, float(np.abs(T - Ts[0]).max()))


def test_ranked_library_loop_one_rank_is_bitwise_the_python_protocol(hip_lib):
    """cilhip_rank_comm_* + cilhip_icp_iterate_ranked: the sharded loop's inner triple inside the library with its own RCCL
    communicator (one process per device).  With ONE rank (all a single-GPU box can run): the communicator comes up from the id,
    and the loop -- blocks of iterations, state read between them -- gives bitwise the transform of the three-call protocol with
    the sums passed through unchanged, and of cilhip_icp_run."""
    import torch

    from cilantro_amd import distributed
    from cilantro_amd.icp import Context

    d = syn.make_pair(400_000, perturb=0.3)
    p = distributed.default_params(max_iter=9, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]))
    T0 = np.eye(4, dtype=np.float32)

    ref = Context(); ref.set_target(d["dst"], d["dst_n"]); ref.set_source(d["src"])
    r0 = ref.icp_run(p)
    _, sm = ref.means()
    ref.close()

    def three_calls():
        ctx = Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
        sums = torch.zeros(capi.SUMS_LEN, dtype=torch.float64, device="cuda")
        ctx.icp_begin(p, T0, sm)
        for _ in range(9):
            ctx.icp_partial_sums(sums.data_ptr()); ctx.icp_apply_sums(sums.data_ptr())
        r = ctx.icp_state(); ctx.close()
        return r

    def ranked():
        ctx = Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
        ctx.rank_comm_init(Context.rank_comm_unique_id(), 1, 0)
        with pytest.raises(RuntimeError):
            ctx.rank_comm_init(Context.rank_comm_unique_id(), 1, 0)       # one communicator per context
        ctx.icp_begin(p, T0, sm)
        for k in (4, 3, 2):
            ctx.icp_iterate_ranked(k)
            assert int(ctx.icp_state().iterations) in (4, 7, 9)
        r = ctx.icp_state()
        ctx.rank_comm_destroy(); ctx.close()
        return r

    ra, rb = three_calls(), ranked()
    for r in (ra, rb):
        assert int(r.iterations) == 9 and int(r.last_ncorr) == int(r0.last_ncorr)
    assert bytes(np.array(ra.T[:], np.float32)) == bytes(np.array(rb.T[:], np.float32))
    assert float(np.abs(np.array(rb.T[:], np.float32) - np.array(r0.T[:], np.float32)).max()) <= 2e-7       # (forms may differ in the order of f64 additions)


def test_multi_device_c_entry_one_gpu(Context, orc, hip_lib):
    """cilhip_multi_*: the sharded loop driven from C in ONE process (SURVEY.md 8(b) devices[]).  What a single-GPU box can check:
    one shard == cilhip_icp_run bit for bit; several shards on the same device (devices = [0, 0, 0]: the all-reduce runs as the
    same-device kernel instead of RCCL) -- source shards, spatial slabs and index shards of the target -- give the single-context loop's transform to the
    order of the f64 additions, iteration for iteration, also when the slab guard fires and all shards are cut again; the loop
    against the oracle's."""
    from cilantro_amd.multi import PARTITION_SLABS, PARTITION_SOURCE_SHARDS, PARTITION_TARGET_SHARDS, MultiDeviceRigidICP

    d = syn.make_pair(400_000, perturb=0.6)
    dst, dst_n, src, r2 = d["dst"], d["dst_n"], d["src"], d["max_sq_dist"]

    def params(iters, tol=0.0, r=r2):
        ctx = Context()
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p)); ctx.close()
        p.metric, p.w_p2p, p.w_p2pl, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.1, 1.0, float(r), iters, tol
        return p

    ctx = Context(); ctx.set_target(dst, dst_n); ctx.set_source(src)
    ref = ctx.icp_run(params(8)); T_ref = np.array(ref.T[:], np.float32)
    ctx.close()
    # one shard: the same kernels in the same order
    m = MultiDeviceRigidICP([0]); m.set_clouds(dst, dst_n, src, r2, PARTITION_SOURCE_SHARDS)
    r1 = m.icp_run(params(8)); m.close()
    assert np.array_equal(np.array(r1.T[:], np.float32).view(np.uint32), T_ref.view(np.uint32)) and int(r1.iterations) == 8 and int(r1.last_ncorr) == int(ref.last_ncorr)
    # ... and through RCCL (a communicator of one rank: librccl opened at run time, ncclAllReduce of the 48 f64 on the context's stream)
    os.environ["CILHIP_MULTI_FORCE_RCCL"] = "1"
    try:
        m = MultiDeviceRigidICP([0]); m.set_clouds(dst, dst_n, src, r2, PARTITION_SLABS)
        r1 = m.icp_run(params(8)); m.close()
    finally:
        del os.environ["CILHIP_MULTI_FORCE_RCCL"]
    assert np.array_equal(np.array(r1.T[:], np.float32).view(np.uint32), T_ref.view(np.uint32)) and int(r1.iterations) == 8
    for part in (PARTITION_SOURCE_SHARDS, PARTITION_SLABS, PARTITION_TARGET_SHARDS):
        m = MultiDeviceRigidICP([0, 0, 0]); m.set_clouds(dst, dst_n, src, r2, part)
        sizes = [m.shard_sizes(k) for k in range(3)]
        if part == PARTITION_TARGET_SHARDS:      # (index shards of the target, the whole source everywhere: MIN of packed keys per iteration)
            assert sum(s[0] for s in sizes) == len(dst) and all(s[1] == len(src) for s in sizes)
        else:
            assert sum(s[1] for s in sizes) == len(src)
        if part == PARTITION_SLABS:
            assert all(0 < s[0] < len(dst) for s in sizes)      # every device holds its slab + halo only
        rr = m.icp_run(params(8)); T = np.array(rr.T[:], np.float32)
        assert int(rr.iterations) == 8 and int(rr.last_ncorr) == int(ref.last_ncorr), (part, int(rr.iterations), int(rr.last_ncorr))
        assert np.abs(T.astype(np.float64) - T_ref.astype(np.float64)).max() <= 2e-6, part
        # convergence-gated: the same iteration count as the oracle's loop
        rc = m.icp_run(params(30, 1e-5)); m.close()
        po = orc.make_params(metric=1, w_p2p=0.1, w_p2pl=1.0, max_iter=30, conv_tol=1e-5, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
        ro = orc.icp_run(dst, dst_n, src, po)
        Tc = np.array(rc.T[:], np.float32).reshape(4, 4).T
        assert int(rc.iterations) == ro["iterations"] and np.linalg.norm(Tc.astype(np.float64) - ro["T"].astype(np.float64)) <= TOL_T, part
    # a slack of a hundredth of a cell: the guard fires with the first updates, all shards are cut again under the last exact transform
    m = MultiDeviceRigidICP([0, 0]); m.set_slab_slack(0.01 * d["h"]); m.set_clouds(dst, dst_n, src, r2, PARTITION_SLABS)
    rr = m.icp_run(params(8), check_every=2)
    assert m.repartitions() >= 1 and int(rr.iterations) == 8 and int(rr.last_ncorr) == int(ref.last_ncorr), (m.repartitions(), int(rr.iterations))
    assert np.abs(np.array(rr.T[:], np.float32).astype(np.float64) - T_ref.astype(np.float64)).max() <= 2e-6
    m.close()


def test_two_engines_share_one_built_target(hip_lib):
    """cilhip_share_target ≙ CorrespondenceSearchKDTree::get/setFirstSearchTree (correspondence_search_kd_tree.hpp:273-296): a second context
    searches the index the first one built -- no build of its own (build_ms 0), bitwise the results of a context that built it itself,
    the order tables of the reference's tree come along (a target with doubled points: loaded, never built again), and either context
    may be destroyed or re-targeted first."""
    from cilantro_amd.icp import Context

    rng = np.random.default_rng(31)
    d = syn.make_pair(300_000, perturb=0.4)
    D = np.ascontiguousarray(np.concatenate([d["dst"], d["dst"][:20_000]]))          # doubled points: every search of them ties
    N = np.ascontiguousarray(np.concatenate([d["dst_n"], d["dst_n"][:20_000]]))
    S1 = d["src"]
    S2 = np.ascontiguousarray(d["src"][::-1] + rng.normal(0, 1e-4, d["src"].shape).astype(np.float32))

    def run(ctx, S):
        ctx.set_source(S)
        p = capi.IcpParams()
        ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric = capi.METRIC_COMBINED; p.w_p2p, p.w_p2pl = 0.0, 1.0
        p.max_iter, p.conv_tol, p.max_sq_dist = 8, 0.0, float(d["max_sq_dist"])
        r = ctx.icp_run(p)
        idx, d2 = ctx.get_nn()
        return bytes(np.array(r.T[:], np.float32)), int(r.last_ncorr), idx.copy(), d2.copy()

    own = {}
    for name, S in (("s1", S1), ("s2", S2)):
        c = Context(0); c.set_target(D, N); own[name] = run(c, S); c.close()
    a = Context(0); a.set_target(D, N)
    ra = run(a, S1)
    assert a.tie_order_info()["loaded"] and a.tie_order_info()["builds"] == 1
    b = Context(0)
    b.share_target(a)
    assert b.grid_info().build_ms == 0.0 and b.grid_info().nx == a.grid_info().nx
    assert b.tie_order_info()["loaded"] and b.tie_order_info()["builds"] == 0          # the tables came along
    rb = run(b, S2)
    assert b.tie_order_info()["builds"] == 0
    for got, want in ((ra, own["s1"]), (rb, own["s2"])):
        assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
        m = got[2] != capi.NONE_IDX
        assert np.array_equal(got[3][m].view(np.uint32), want[3][m].view(np.uint32))
    # the lender goes first: the borrower keeps searching the same memory; then a third context shares from the borrower
    a.close()
    assert run(b, S2)[0] == own["s2"][0]
    c3 = Context(0)
    c3.share_target(b)
    b.set_target(d["dst"], d["dst_n"])                                                # the second user is re-targeted: the share lives on in the third
    assert run(c3, S1)[0] == own["s1"][0]
    with pytest.raises(capi.CilhipError):
        c3.share_target(c3)
    b.close(); c3.close()
