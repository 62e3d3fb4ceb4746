"""CPU tests (-m "not gpu"): the oracle against the reference's golden vectors, known answers and
independent numpy restatements.  No GPU, no /root/reference needed (golden fixtures are committed)."""
import os

import numpy as np
import pytest

from cilantro_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def knn_gold():
    return np.load(os.path.join(GOLD, "knn_golden.npz"))


def test_kd_tree_example_known_answer(orc, knn_gold):
    # examples/kd_tree.cpp:6-19 -> by hand: idx {0,3}, d2 {0.18, 0.38}
    ii, dd = orc.KDTree(knn_gold["cube"]).knn_in_radius(knn_gold["cube_q"], 2, 1.001)
    assert list(ii) == [0, 3]
    np.testing.assert_allclose(dd, [0.18, 0.38], rtol=1e-6)
    assert list(knn_gold["cube_idx"]) == [0, 3] and np.array_equal(dd, knn_gold["cube_d2"])


@pytest.mark.parametrize("name", ["r_small", "r_mid", "r_inf"])
def test_oracle_kdtree_matches_reference_nanoflann_golden(orc, knn_gold, name):
    dst, q, r2 = knn_gold["dst"], knn_gold["q"], knn_gold[f"{name}_r2"]
    tree = orc.KDTree(dst)
    gi, gd = knn_gold[f"{name}_knn3_idx"], knn_gold[f"{name}_knn3_d2"]
    for i in range(len(q)):
        ii, dd = tree.knn_in_radius(q[i], 3, r2)
        n = int((gi[i] >= 0).sum())
        assert len(ii) == n and np.array_equal(ii, gi[i, :n]) and np.array_equal(dd, gd[i, :n]), i
    di, si, dv = tree.find_correspondences(q, r2)
    assert np.array_equal(di, knn_gold[f"{name}_corr_first"])
    assert np.array_equal(si, knn_gold[f"{name}_corr_second"])
    assert np.array_equal(dv, knn_gold[f"{name}_corr_value"])
    # brute force agrees on distances everywhere and on indices wherever the distance is unique
    bi, bd = orc.nn_brute(dst, q, r2)
    kept = bi >= 0
    assert np.array_equal(np.nonzero(kept)[0], si)
    assert np.array_equal(bd[kept], dv)
    diff = bi[kept] != di
    for j in np.nonzero(diff)[0]:                      # only exact ties (duplicated points) may differ
        assert np.array_equal(dst[bi[kept][j]], dst[di[j]])
        assert bi[kept][j] < di[j]                     # brute force = lowest index


def test_oracle_kdtree_vs_live_reference(orc):
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    d = syn.make_pair(60000)
    q = orc.transform_points(np.eye(4), d["src"])
    a = orc.KDTree(d["dst"]).find_correspondences(q, d["max_sq_dist"])
    b = orc.KDTree(d["dst"], use_ref=True).find_correspondences(q, d["max_sq_dist"], num_threads=2)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_radius_is_strict_and_empty_inputs(orc):
    dst = np.array([[0, 0, 0], [4, 0, 0]], np.float32)
    t = orc.KDTree(dst)
    assert len(t.knn_in_radius([0.5, 0, 0], 1, np.float32(0.25))[0]) == 0     # d2 == r2 rejected
    assert list(t.knn_in_radius([0.5, 0, 0], 1, np.nextafter(np.float32(0.25), np.float32(1)))[0]) == [0]
    e = orc.KDTree(np.zeros((0, 3), np.float32))
    assert len(e.find_correspondences(np.zeros((5, 3), np.float32), 1.0)[0]) == 0   # kd_tree_utilities.hpp:16-19
    assert len(t.find_correspondences(np.zeros((0, 3), np.float32), 1.0)[0]) == 0


def test_transform_expression_is_pinned(orc):
    rng = np.random.default_rng(3)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = syn.rot_xyz(0.3, -0.2, 0.5).astype(np.float32)
    T[:3, 3] = [0.1, -0.4, 0.25]
    p = rng.random((1000, 3), dtype=np.float32)
    q = orc.transform_points(T, p)
    L = T[:3, :3]
    f = np.float32
    exp = np.stack([(f(L[r, 0]) * p[:, 0] + (f(L[r, 1]) * p[:, 1] + f(L[r, 2]) * p[:, 2])) + f(T[r, 3]) for r in range(3)], 1)
    assert np.array_equal(q, exp.astype(np.float32))


def test_svd_ldlt_rotation_vs_numpy(orc):
    rng = np.random.default_rng(0)
    for i in range(300):
        A = rng.standard_normal((3, 3))
        if i % 7 == 0:
            A[:, 2] = A[:, 0] * 2.0                       # rank deficient
        if i % 11 == 0:
            A = np.diag([1.0, 1.0, 1.0]) + 1e-9 * A       # near identity (the rotation() polish case)
        U, S, V = orc.svd3(A)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() < 1e-13 * max(1.0, np.abs(A).max())
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
        assert S[0] >= S[1] >= S[2] >= 0
        np.testing.assert_allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-13)
        R = orc.nearest_rotation(A)
        Un, _, Vnt = np.linalg.svd(A)
        if np.linalg.det(Un @ Vnt) < 0:
            Un[:, 0] *= -1                                # space_transformations.hpp:45-48 flips column 0
        if i % 7 != 0:
            np.testing.assert_allclose(R, Un @ Vnt, atol=1e-9)
        assert abs(np.linalg.det(R) - 1.0) < 1e-12
    for i in range(50):
        M = rng.standard_normal((30, 6))
        A = M.T @ M
        b = rng.standard_normal(6)
        np.testing.assert_allclose(orc.ldlt6_solve(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # singular system: Eigen's LDLT::solve pseudo-inverse-of-D semantics -> finite, residual orthogonal
    A = np.zeros((6, 6)); A[:3, :3] = np.eye(3)
    x = orc.ldlt6_solve(A, np.array([1, 2, 3, 4, 5, 6.0]))
    np.testing.assert_allclose(x, [1, 2, 3, 0, 0, 0])


def _numpy_kabsch(P, Q):
    mp, mq = P.mean(0), Q.mean(0)
    S = (P - mp).T @ (Q - mq) / len(P)
    U, _, Vt = np.linalg.svd(S)
    if np.linalg.det(U @ Vt) < 0:
        U[:, 2] *= -1
    R = U @ Vt
    return R, mp - R @ mq


def _numpy_gn_step(P, N, Q, w_p2p, w_p2pl, dm, sm):
    AtA = np.zeros((6, 6)); Atb = np.zeros(6)
    d = P - dm; s = Q - sm
    a = d + s; r = d - s
    if w_p2pl > 0:
        e = np.concatenate([np.cross(a, N), N], 1)
        res = (N * r).sum(1)
        AtA += w_p2pl * e.T @ e
        Atb += w_p2pl * e.T @ res
    if w_p2p > 0:
        for i in range(len(P)):
            ax = np.array([[0, -a[i, 2], a[i, 1]], [a[i, 2], 0, -a[i, 0]], [-a[i, 1], a[i, 0], 0]])
            E = np.concatenate([ax, np.eye(3)], 0)
            AtA += w_p2p * E @ E.T
            Atb += w_p2p * E @ r[i]
    return AtA, Atb


def test_estimators_vs_numpy(orc):
    d = syn.make_pair(3000, perturb=0.4)
    q = orc.transform_points(np.eye(4), d["src"])
    di, si, dv = orc.KDTree(d["dst"]).find_correspondences(q, d["max_sq_dist"])
    P = d["dst"][di].astype(np.float64); Q = q[si].astype(np.float64); N = d["dst_n"][di].astype(np.float64)
    R, t = _numpy_kabsch(P, Q)
    for mode, tol in ((orc.MODE_F64, 1e-9), (orc.MODE_MIXED, 1e-6), (orc.MODE_F32, 2e-5)):
        T, sums, ok = orc.estimate_p2p(d["dst"], q, di, si, mode)
        assert ok and np.abs(T[:3, :3] - R).max() < tol + 1e-7 and np.abs(T[:3, 3] - t).max() < tol + 1e-7
    assert sums[0] == len(di)
    np.testing.assert_allclose(sums[1:4], P.sum(0), rtol=1e-12)
    np.testing.assert_allclose(sums[7:16].reshape(3, 3), P.T @ Q, rtol=1e-12)
    dm = orc.mean3(d["dst"]); sm = orc.mean3(q)
    for w_p2p, w_p2pl in ((0.0, 1.0), (1.0, 0.0), (0.3, 1.0)):
        # the reference's weights are f32 (TransformT::Scalar): compare against the same rounded values
        AtA_np, Atb_np = _numpy_gn_step(P, N, Q, float(np.float32(w_p2p)), float(np.float32(w_p2pl)), dm.astype(np.float64), sm.astype(np.float64))
        T, AtA, Atb, cv = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, w_p2p, w_p2pl, dm, sm, 1, 1e-5, orc.MODE_F64)
        np.testing.assert_allclose(AtA, AtA_np, rtol=1e-9, atol=1e-9 * np.abs(AtA_np).max())
        np.testing.assert_allclose(Atb, Atb_np, rtol=1e-8, atol=1e-9 * np.abs(AtA_np).max())
        x = np.linalg.solve(AtA_np, Atb_np)
        na = np.linalg.norm(x[:3]); th = np.arctan(na); u = x[:3] / na
        K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
        Ra = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        L = Ra @ Ra; tt = Ra @ (np.cos(th) * x[3:])
        tt = tt - L @ sm.astype(np.float64) + dm.astype(np.float64)
        assert np.abs(T[:3, :3] - L).max() < 1e-6 and np.abs(T[:3, 3] - tt).max() < 1e-6
        # the three arithmetic modes agree to f32 accumulation accuracy
        T32, *_ = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, w_p2p, w_p2pl, dm, sm, 1, 1e-5, orc.MODE_F32)
        Tmx, *_ = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, w_p2p, w_p2pl, dm, sm, 1, 1e-5, orc.MODE_MIXED)
        assert np.abs(T32 - T).max() < 2e-5 and np.abs(Tmx - T).max() < 1e-6


def test_degenerate_estimates_are_identity(orc):
    d = syn.make_pair(100)
    e = np.zeros(0, np.int64)
    T, _, ok = orc.estimate_p2p(d["dst"], d["src"], e, e)
    assert not ok and np.array_equal(T, np.eye(4, dtype=np.float32))            # transform_estimation.hpp:20-23
    T, _, _, cv = orc.estimate_combined(d["dst"], d["dst_n"], d["src"], e, e, 0.0, 1.0, np.zeros(3), np.zeros(3))
    assert not cv and np.array_equal(T, np.eye(4, dtype=np.float32))            # :269-272


@pytest.mark.parametrize("metric", [0, 1])
def test_icp_converges_to_ground_truth_and_golden(orc, metric):
    g = np.load(os.path.join(GOLD, "icp_golden.npz"))
    d = syn.make_pair(int(g["n"]), perturb=float(g["perturb"]))
    np.testing.assert_array_equal(d["T_true"], g["T_true"])
    for mode in (orc.MODE_F32, orc.MODE_MIXED, orc.MODE_F64):
        p = orc.make_params(metric=metric, max_iter=6, conv_tol=0.0, max_sq_dist=d["max_sq_dist"], mode=mode, num_threads=1)
        r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
        assert np.abs(r["T"] - g[f"T_m{metric}_mode{mode}"]).max() <= 2e-7        # regression vector
        assert r["last_ncorr"] == int(g[f"ncorr_m{metric}_mode{mode}"])
        assert np.linalg.norm(r["T"] - d["T_true"]) < 5e-4                        # analytic anchor (noise-limited)
    p = orc.make_params(metric=metric, max_iter=50, conv_tol=1e-5, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    assert r["last_delta_norm"] < 1e-5 and r["iterations"] < 15


def test_synthetic_generator_is_deterministic():
    z = syn.splitmix64(42, 3)
    assert [int(v) for v in z] == [13679457532755275413, 2949826092126892291, 5139283748462763858]
    a = syn.make_pair(1000); b = syn.make_pair(1000)
    assert np.array_equal(a["dst"], b["dst"]) and np.array_equal(a["src"], b["src"]) and np.array_equal(a["dst_n"], b["dst_n"])
    assert a["dst"].min() >= 0 and a["dst"].max() < 1
    np.testing.assert_allclose(np.linalg.norm(a["dst_n"], axis=1), 1.0, atol=1e-6)


def test_plane_ransac_oracle_pieces(orc):
    """ransac_oracle.c: the Jacobi eigen-solver against numpy, PCA plane fit against an SVD fit, and the
    sequential RANSAC loop's bookkeeping (ransac_base.hpp:103-114) on a synthetic plane."""
    rng = np.random.default_rng(0)
    for t in range(300):
        B = rng.standard_normal((3, 3))
        A = B @ B.T if t % 3 else np.outer(B[0], B[0]) + np.outer(B[1], B[1])   # rank-deficient too
        w, V = orc.sym_eig3(A)
        assert np.allclose(A @ V, V * w, atol=1e-12 * max(1.0, np.abs(w).max()))
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-14) and np.linalg.det(V) > 0
        assert np.allclose(w, np.sort(np.linalg.eigvalsh(A))[::-1], atol=1e-12 * max(1.0, np.abs(w).max()))
    n = 20000
    x = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    x[:12000, 2] = (0.3 * x[:12000, 0] - 0.2 * x[:12000, 1] + 0.1 + rng.normal(0, 0.003, 12000)).astype(np.float32)
    idx = np.arange(12000, dtype=np.uint32)
    for mode in (0, 1):
        pl = orc.plane_fit(x, idx, mode=mode)
        xc = x[:12000].astype(np.float64)
        nrm = np.linalg.svd(xc - xc.mean(0), full_matrices=False)[2][2]
        s = 1.0 if np.dot(nrm, pl[:3]) > 0 else -1.0
        assert np.abs(s * nrm - pl[:3]).max() < (2e-4 if mode == 0 else 2e-7)
        assert abs(pl[3] + s * float(nrm @ xc.mean(0))) < (2e-4 if mode == 0 else 2e-7)
    samples = rng.integers(0, n, (100, 3)).astype(np.uint32)
    pl, res, inl, it = orc.plane_ransac(x, samples, 0.01, n // 2, re_estimate=True, mode=1)
    assert 0 < it <= 100 and len(inl) >= n // 2
    assert np.array_equal(inl, np.nonzero(res <= np.float32(0.01))[0])
    assert np.array_equal(res, orc.plane_residuals(x, pl))
    assert np.abs(pl / -pl[2] - np.array([0.3, -0.2, -1.0, 0.1])).max() < 1e-3
    # unreachable target: all iterations are spent, the best model is still reported
    pl2, _, inl2, it2 = orc.plane_ransac(x, samples, 0.01, n, re_estimate=False, mode=1)
    assert it2 == 100 and len(inl2) == max(orc.plane_count_inliers(x, orc.plane_fit(x, s3, mode=1), 0.01) for s3 in samples)
    # a degenerate sample (three times the same point) fits a NaN-free or NaN plane but never crashes
    _ = orc.plane_fit(x, np.zeros(3, np.uint32), mode=1)


def test_frame1_recipe_fixture_converges(orc):
    """BASELINE configs[0]: the reference's own test cloud examples/test_clouds/frame_1.ply through the
    recipe of examples/rigid_icp.cpp:25-65 (fixture tests/golden/frame1_c1.npz, made by make_golden.py), with the
    parameters of examples/rigid_icp.cpp:116-125.  The estimate must be the inverse of the applied motion (:132-133)."""
    f = np.load(os.path.join(GOLD, "frame1_c1.npz"))
    p = orc.make_params(metric=orc.METRIC_COMBINED, w_p2p=0.0, w_p2pl=1.0, max_iter=30, conv_tol=1e-4, max_opt_iter=1,
                        max_sq_dist=0.1 * 0.1, mode=orc.MODE_MIXED)
    r = orc.icp_run(f["dst"], f["dst_n"], f["src"], p)
    T_true = np.linalg.inv(f["T_ref"].astype(np.float64))
    assert r["iterations"] < 30 and r["last_delta_norm"] < 1e-4
    assert np.linalg.norm(r["T"] - T_true) < 5e-3, np.linalg.norm(r["T"] - T_true)     # jitter-limited (0.01 uniform noise)
    # the reference-like all-f32 mode lands on the same transform to f32 accumulation accuracy
    p32 = orc.make_params(metric=orc.METRIC_COMBINED, w_p2p=0.0, w_p2pl=1.0, max_iter=30, conv_tol=1e-4, max_opt_iter=1,
                          max_sq_dist=0.1 * 0.1, mode=orc.MODE_F32)
    r32 = orc.icp_run(f["dst"], f["dst_n"], f["src"], p32)
    assert np.linalg.norm(r32["T"] - r["T"]) < 1e-3


def test_affine_estimator_known_answer(orc):
    """The affine closed forms (transform_estimation.hpp:50-102, :369-476) recover a known affine map exactly from exact
    correspondences, in every arithmetic mode, for any mix of point / plane weights; and follow the reference's early
    returns (no terms, plane terms without normals: identity, false)."""
    rng = np.random.default_rng(0)
    n = 3000
    src = rng.random((n, 3)).astype(np.float32)
    A = np.eye(3) + 0.05 * rng.normal(size=(3, 3)); t = np.array([0.01, -0.02, 0.03])
    dst = (src.astype(np.float64) @ A.T + t).astype(np.float32)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True); nrm = nrm.astype(np.float32)
    idx = np.arange(n)
    dm, sm = orc.mean3(dst), orc.mean3(src)
    zero = np.zeros(3, np.float32)
    for mode, tol in ((orc.MODE_F32, 2e-5), (orc.MODE_MIXED, 5e-7), (2, 5e-7)):
        for w_p2p, w_p2pl, means in ((0.3, 1.0, (dm, sm)), (0.0, 1.0, (dm, sm)), (1.0, 0.0, (dm, sm)), (1.0, 0.0, (zero, zero))):
            T, AtA, Atb, ok = orc.estimate_affine(dst, nrm, src, idx, idx, w_p2p, w_p2pl, means[0], means[1], mode)
            assert ok and np.abs(AtA - AtA.T).max() <= 1e-6 * np.abs(AtA).max()
            assert np.abs(T[:3, :3] - A).max() < tol and np.abs(T[:3, 3] - t).max() < tol, (mode, w_p2p, w_p2pl)
    T, _, _, ok = orc.estimate_affine(dst, nrm, src, idx, idx, 0.0, 0.0, dm, sm, orc.MODE_MIXED)
    assert not ok and np.array_equal(T, np.eye(4))
    T, _, _, ok = orc.estimate_affine(dst, None, src, idx, idx, 0.0, 1.0, dm, sm, orc.MODE_MIXED)
    assert not ok and np.array_equal(T, np.eye(4))
    T, _, _, ok = orc.estimate_affine(dst, nrm, src, idx[:3], idx[:3], 1.0, 0.0, dm, sm, orc.MODE_MIXED)
    assert not ok            # fewer than Dim + 1 terms (:475-477)
    # the ICP instance on top: a distorted copy converges back onto the target
    p = orc.make_params(metric=1, w_p2p=0.1, w_p2pl=1.0, max_iter=30, conv_tol=1e-6, max_sq_dist=0.05 ** 2, affine=True)
    S = np.eye(3) + 0.01 * rng.normal(size=(3, 3))
    moved = ((dst.astype(np.float64) - 0.5) @ S.T + 0.5 + 0.004).astype(np.float32)
    r = orc.icp_run(dst, nrm, moved, p)
    back = orc.transform_points(r["T"], moved)
    assert np.abs(back - dst).max() < 5e-4 and r["iterations"] < 30


def test_point_normal_feature_search_vs_reference_nanoflann(orc):
    """6-D point+normal features (common_transformable_feature_adaptors.hpp:60-161): the oracle's exhaustive search with
    the pinned DIM = 6 distance arithmetic against the reference's own nanoflann instantiated for DIM = 6 -- same indices,
    bit-identical distances, same strict radius test."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(21)
    n = 6000
    dst = rng.random((n, 3)).astype(np.float32)
    dn = rng.normal(size=(n, 3)); dn /= np.linalg.norm(dn, axis=1, keepdims=True); dn = dn.astype(np.float32)
    src = (dst + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(0.01))[rng.permutation(n)[:4000]]
    sn = rng.normal(size=(len(src), 3)); sn /= np.linalg.norm(sn, axis=1, keepdims=True); sn = sn.astype(np.float32)
    ang = 0.05
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    T[:3, 3] = [0.01, -0.005, 0.002]
    for w in (0.02, 0.3):
        dst6 = orc.point_normal_features(dst, dn, w)
        assert np.array_equal(dst6[:, :3], dst) and np.array_equal(dst6[:, 3:], np.float32(w) * dn)
        q6 = orc.transform_features6(T, orc.point_normal_features(src, sn, w))
        assert np.array_equal(q6[:, :3], orc.transform_points(T, src))
        for max_sq in (0.05 ** 2, float("inf")):
            a1, a2, av = orc.find_correspondences_feat6(dst6, q6, max_sq)
            b1, b2, bv = orc.find_correspondences_feat6(dst6, q6, max_sq, use_ref=True)
            assert np.array_equal(a2, b2) and np.array_equal(a1, b1) and np.array_equal(av, bv)
        # with a heavy normal weight the feature match differs from the plain nearest point for many queries
    p1, _, _ = orc.KDTree(dst).find_correspondences(q6[:, :3].copy(), float("inf"))
    assert np.mean(p1 != a1) > 0.2


def test_feat9_oracle_vs_reference_nanoflann_dim9(orc):
    """PointNormalColorFeaturesAdaptor (common_transformable_feature_adaptors.hpp:255-343): the oracle's exhaustive 9-D search with
    nanoflann's DIM = 9 summation order (two groups of four, one tail term) against the reference's own nanoflann instantiated for
    DIM = 9 -- indices and squared distances bit for bit; the transform moves points and normals, not colours."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(23)
    n = 5000
    dst = rng.random((n, 3)).astype(np.float32)
    dn = rng.normal(size=(n, 3)); dn /= np.linalg.norm(dn, axis=1, keepdims=True); dn = dn.astype(np.float32)
    dc = rng.random((n, 3)).astype(np.float32)
    sel = rng.permutation(n)[:3000]
    src = (dst + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(0.01))[sel]
    sn = dn[sel]
    sc = np.clip(dc[sel] + rng.normal(0, 0.05, (len(sel), 3)), 0, 1).astype(np.float32)
    ang = 0.05
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    T[:3, 3] = [0.01, -0.005, 0.002]
    for wn, wc in ((0.02, 0.03), (0.3, 0.1), (0.0, 0.2)):
        d9 = orc.point_normal_color_features(dst, dn, dc, wn, wc)
        assert np.array_equal(d9[:, :3], dst) and np.array_equal(d9[:, 3:6], np.float32(wn) * dn) and np.array_equal(d9[:, 6:], np.float32(wc) * dc)
        s9 = orc.point_normal_color_features(src, sn, sc, wn, wc)
        q9 = orc.transform_features9(T, s9)
        assert np.array_equal(q9[:, :3], orc.transform_points(T, src)) and np.array_equal(q9[:, 6:], s9[:, 6:])
        assert np.array_equal(q9[:, :6], orc.transform_features6(T, s9[:, :6].copy()))
        for max_sq in (0.05 ** 2, float("inf")):
            a = orc.find_correspondences_feat9(d9, q9, max_sq)
            b = orc.find_correspondences_feat9(d9, q9, max_sq, use_ref=True)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a[0]) > 0
    # the 9-D distance is not the 6-D one plus a colour term in a different order: DIM = 9 groups (d4, d5) with the colour's first two
    a6 = orc.find_correspondences_feat6(d9[:, :6].copy(), q9[:, :6].copy(), float("inf"))
    assert len(a6[0]) == len(a[0])


def test_radius_search_oracle_vs_reference_nanoflann(orc):
    """KDTree::radiusSearch (core/kd_tree.hpp:251-282): the exhaustive oracle against the reference's own nanoflann with
    cilantro's RadiusSearchResultAdaptor -- same neighbour sets, bit-identical sorted distances, strict radius."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(31)
    pts = rng.random((4000, 3)).astype(np.float32)
    q = np.concatenate([pts[:50], rng.random((50, 3)).astype(np.float32) * 1.2 - 0.1])
    tree = orc.KDTree(pts, use_ref=True)
    for r2 in (0.0, 0.03 ** 2, 0.11 ** 2):
        off, idx, d2 = orc.radius_search(pts, q, r2)
        assert off[0] == 0 and off[-1] == len(idx)
        for i in range(len(q)):
            ri, rd = orc.ref_radius_search(tree, q[i], r2)
            mine_i, mine_d = idx[off[i]:off[i + 1]], d2[off[i]:off[i + 1]]
            assert np.array_equal(rd, mine_d)                      # sorted distances, bit for bit
            assert np.array_equal(np.sort(ri), np.sort(mine_i))    # the same set (tie order is unspecified in the reference)
            assert np.all(mine_d < np.float32(r2)) if len(mine_d) else True
    # a query point that is a tree point sits in its own list at distance 0 (first)
    off, idx, d2 = orc.radius_search(pts, pts[:5], 0.05 ** 2)
    assert all(idx[off[i]] == i and d2[off[i]] == 0.0 for i in range(5))


def test_weight_evaluators_vs_numpy(orc):
    """core/common_pair_evaluators.hpp:14-27 / :30-43 / :46-80 inside the combined-metric estimator
    (transform_estimation.hpp:301-303, :330-332): weighted normal equations against a direct numpy evaluation, the pinned
    exp against the correctly rounded one, and a weighted loop that still converges to the ground truth."""
    rng = np.random.default_rng(11)
    x = np.concatenate([-np.abs(rng.standard_normal(20000).astype(np.float32)) * 12, np.float32([0, -1e-8, -79.9, -80.0, -80.1, -1e4])])
    y = orc.pinned_expf(x)
    ref = np.exp(x.astype(np.float64))
    m = x >= -80.0
    assert np.all(np.abs(y[m].astype(np.float64) - ref[m]) <= 1.0 * np.spacing(ref[m].astype(np.float32)))   # within 1 ulp
    assert np.all(y[~m] == 0.0) and y[-6] == 1.0

    d = syn.make_pair(3000, perturb=0.4)
    q = orc.transform_points(np.eye(4), d["src"])
    di, si, dv = orc.KDTree(d["dst"]).find_correspondences(q, d["max_sq_dist"])
    P = d["dst"][di].astype(np.float64); Q = q[si].astype(np.float64); N = d["dst_n"][di].astype(np.float64)
    dm = orc.mean3(d["dst"]); sm = orc.mean3(q)
    sig_p, sig_l = 0.4 * float(np.sqrt(d["max_sq_dist"])), 0.8 * float(np.sqrt(d["max_sq_dist"]))

    def weights(kind, sigma, w_metric):
        v = dv.astype(np.float64)
        if kind == orc.W_UNITY:
            e = np.ones_like(v)
        elif kind == orc.W_IDENTITY:
            e = v
        else:
            e = np.exp(float(np.float32(-0.5) / (np.float32(sigma) * np.float32(sigma))) * v)
        return float(np.float32(w_metric)) * e

    for pk, lk, w_p2p, w_p2pl in ((orc.W_UNITY, orc.W_RBF, 0.0, 1.0), (orc.W_RBF, orc.W_RBF, 0.3, 1.0), (orc.W_IDENTITY, orc.W_UNITY, 1.0, 0.0),
                                  (orc.W_RBF, orc.W_IDENTITY, 0.5, 0.7)):
        wp, wl = weights(pk, sig_p, w_p2p), weights(lk, sig_l, w_p2pl)
        dd = P - dm.astype(np.float64); ss = Q - sm.astype(np.float64)
        a = dd + ss; r = dd - ss
        AtA_np = np.zeros((6, 6)); Atb_np = np.zeros(6)
        if w_p2pl > 0:
            e = np.concatenate([np.cross(a, N), N], 1)
            AtA_np += (e * wl[:, None]).T @ e
            Atb_np += (e * wl[:, None]).T @ (N * r).sum(1)
        if w_p2p > 0:
            for i in range(len(P)):
                ax = np.array([[0, -a[i, 2], a[i, 1]], [a[i, 2], 0, -a[i, 0]], [-a[i, 1], a[i, 0], 0]])
                E = np.concatenate([ax, np.eye(3)], 0)
                AtA_np += wp[i] * E @ E.T
                Atb_np += wp[i] * E @ r[i]
        Ts = {}
        for mode, tol in ((orc.MODE_F64, 1e-9), (orc.MODE_MIXED, 2e-6), (orc.MODE_F32, 1e-4)):
            T, AtA, Atb, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, w_p2p, w_p2pl, dm, sm, 1, 1e-5, mode,
                                                   values=dv, weights=(pk, lk, sig_p, sig_l))
            np.testing.assert_allclose(AtA, AtA_np, rtol=0, atol=tol * np.abs(AtA_np).max())
            np.testing.assert_allclose(Atb, Atb_np, rtol=0, atol=tol * np.abs(Atb_np).max() + 1e-3 * tol * np.abs(AtA_np).max())
            Ts[mode] = T
        assert np.abs(Ts[orc.MODE_MIXED] - Ts[orc.MODE_F64]).max() < 1e-6 and np.abs(Ts[orc.MODE_F32] - Ts[orc.MODE_F64]).max() < 5e-5
    # unity evaluators through the weighted entry point = the plain one, bit for bit
    T0, A0, b0, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, 0.3, 1.0, dm, sm, 2, 1e-7, orc.MODE_MIXED)
    T1, A1, b1, _ = orc.estimate_combined(d["dst"], d["dst_n"], q, di, si, 0.3, 1.0, dm, sm, 2, 1e-7, orc.MODE_MIXED, values=dv,
                                          weights=(orc.W_UNITY, orc.W_UNITY, 1.0, 1.0))
    assert np.array_equal(T0, T1) and np.array_equal(A0, A1) and np.array_equal(b0, b1)
    # a weighted loop still lands on the ground truth
    p = orc.make_params(metric=1, w_p2p=0.1, max_iter=30, conv_tol=1e-5, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED,
                        point_weight=orc.W_RBF, plane_weight=orc.W_RBF, point_sigma=sig_p, plane_sigma=sig_l)
    r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    assert r["last_delta_norm"] < 1e-5 and np.linalg.norm(r["T"] - d["T_true"]) < 2e-3


def test_transform_ransac_oracle_known_answers(orc):
    """RigidTransformRANSACEstimator3f restatement (oracle/ransac_oracle.c): the residual expression against numpy, the closed-form
    fit recovers a known rigid motion from exact pairs, the loop recovers it from pairs with 40 % gross outliers and stops at the
    reference's iteration (ransac_base.hpp:103-114), and the degenerate cases follow the reference (no accepted model)."""
    from cilantro_amd import synthetic as syn

    rng = np.random.default_rng(5)
    n = 5000
    src = rng.random((n, 3)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = syn.rot_xyz(0.3, -0.2, 0.5); T[:3, 3] = [0.1, -0.3, 0.2]
    dst = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    Tf = orc.transform_fit(dst, src, None, mode=orc.MODE_MIXED)
    assert np.abs(Tf.astype(np.float64) - T).max() < 2e-6
    res = orc.transform_residuals(dst, src, T.astype(np.float32))
    ref = np.linalg.norm((src.astype(np.float64) @ T.astype(np.float32).astype(np.float64)[:3, :3].T + T.astype(np.float32).astype(np.float64)[:3, 3]) - dst, axis=1)
    assert np.abs(res - ref).max() < 1e-6 and orc.transform_count_inliers(dst, src, T.astype(np.float32), 1e-4) == n
    # outliers
    bad = rng.random(n) < 0.4
    dst_o = dst.copy(); dst_o[bad] = rng.random((int(bad.sum()), 3)).astype(np.float32) * 3.0
    samples = rng.integers(0, n, (200, 3)).astype(np.uint32)
    Tr, resr, inl, it, have = orc.transform_ransac(dst_o, src, samples, 1e-3, n // 2, mode=orc.MODE_MIXED)
    assert have == 3 and it <= 200 and np.abs(Tr.astype(np.float64) - T).max() < 1e-5
    assert set(inl.tolist()) >= set(np.nonzero(~bad)[0].tolist()) and len(inl) <= int((~bad).sum()) + 5
    # the stopping iteration is the first one whose sample lies entirely among the good pairs
    good_iter = next(k for k in range(200) if not bad[samples[k]].any()) + 1
    assert it == good_iter
    # nothing reaches 3 inliers: no model; without re-estimation no inliers, with it the identity's
    Tn, _, inln, itn, haven = orc.transform_ransac(dst_o[bad][:500], src[bad][:500], samples[:20] % 500, 0.0, 250, re_estimate=False, mode=orc.MODE_MIXED)
    assert haven == 0 and itn == 20 and len(inln) == 0 and np.array_equal(Tn, np.eye(4, dtype=np.float32))
