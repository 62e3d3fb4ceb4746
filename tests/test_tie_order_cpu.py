"""tests/cpp/tie_order_host.hpp (the host restatement of the index build behind option "tie_rule"; the product builds the same tables on the
device: csrc/tie_build.hip, checked against this one in tests/test_gpu_tie_rule.py): the tree restatement names, for every query with several exactly
equidistant nearest target points, the point the reference's nanoflann returns (core/kd_tree.hpp:82-90; oracle/_ref when built,
the oracle's kd-tree restatement otherwise).  No GPU: the header is host-only, compiled by tests/cpp/build.sh into a shim."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial import cKDTree

from cilantro_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "cpp", "bin", "libtie_order_shim.so")
K = 12


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "tests", "cpp", "tie_order_shim.cpp"), "-o", SHIM])
    L = C.CDLL(SHIM)
    L.tie_shim_build.restype = C.c_void_p
    L.tie_shim_build.argtypes = [C.c_void_p, C.c_uint32]
    L.tie_shim_build_mt.restype = C.c_void_p
    L.tie_shim_build_mt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint]
    L.tie_shim_same_order.argtypes = [C.c_void_p, C.c_void_p]
    L.tie_shim_free.argtypes = [C.c_void_p]
    L.tie_shim_first_met.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_void_p]
    L.tie_shim_min_key.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


def _pinned_d2(q, p):      # (dx*dx + dy*dy) + dz*dz in f32, no contraction: the engine's and nanoflann's 3-D distance
    d = (q - p).astype(np.float32)
    return ((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])).astype(np.float32) + (d[..., 2] * d[..., 2]).astype(np.float32)


def tied_queries(D, Q, r2):
    """-> (query indices, candidates [n, K] ascending original index, counts)"""
    _, nb = cKDTree(D.astype(np.float64)).query(Q.astype(np.float64), k=K)
    d2 = _pinned_d2(Q[:, None, :], D[nb])
    m = d2.min(axis=1)
    tied = d2 == m[:, None]
    cnt = tied.sum(axis=1)
    assert cnt.max() < K
    sel = np.nonzero((cnt >= 2) & (m <= r2))[0]
    cand = np.zeros((len(sel), K), np.uint32)
    for k, i in enumerate(sel):
        c = np.sort(nb[i][tied[i]])
        cand[k, : len(c)] = c
    return sel, cand, cnt[sel].astype(np.int32)


def _first_met(L, D, Q, cand, counts):
    h = L.tie_shim_build(np.ascontiguousarray(D, np.float32).ctypes.data, len(D))
    q = np.ascontiguousarray(Q, np.float32)
    out = np.zeros(max(len(q), 1), np.uint32)
    L.tie_shim_first_met(h, q.ctypes.data, cand.ctypes.data, counts.ctypes.data, K, len(q), out.ctypes.data)
    L.tie_shim_free(h)
    return out[: len(q)].astype(np.int64)


def _clouds():
    f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
    p1, p2 = f["p1"], f["p2"]
    yield "frame_1 -> frame_2 (sensor lattice)", np.ascontiguousarray(p1[p1[:, 0] > -0.4]), np.ascontiguousarray(p2), float(np.float32(0.02 * 0.02))
    rng = np.random.default_rng(3)
    b = syn.make_pair(60_000, perturb=0.3)
    dup = rng.choice(len(b["dst"]), 3000, replace=False)
    D = np.ascontiguousarray(np.concatenate([b["dst"], b["dst"][dup], b["dst"][dup[:500]]]))      # doubled and tripled points
    yield "duplicated target points", D, b["src"], float(b["max_sq_dist"])


def test_first_met_is_the_reference_pick(shim, orc):
    for name, D, S, r2 in _clouds():
        sel, cand, counts = tied_queries(D, S, r2)
        assert len(sel) > 500, name
        got = _first_met(shim, D, S[sel], cand, counts)
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        o1, o2, _ = tree.find_correspondences(S, r2)
        oi = np.full(len(S), -1, np.int64)
        oi[o2] = o1
        assert np.array_equal(got, oi[sel]), (name, int(np.count_nonzero(got != oi[sel])))
        # (and the rule matters: the lowest index is a different point for a good part of them)
        assert np.count_nonzero(cand[:, 0].astype(np.int64) != oi[sel]) > len(sel) // 4, name


def test_smallest_traversal_key_is_the_reference_pick(shim, orc):
    """index shards of a target settle ties between shards by ONE number per candidate (TieOrderTree::traversal_key; on the device
    csrc/search_device.hpp tie_rank): the candidate with the smallest key is the one the reference's nanoflann returns, whatever the number of
    candidates; the keys' 58 levels are far from exhausted on these clouds"""
    for name, D, S, r2 in _clouds():
        sel, cand, counts = tied_queries(D, S, r2)
        h = shim.tie_shim_build(np.ascontiguousarray(D, np.float32).ctypes.data, len(D))
        q = np.ascontiguousarray(S[sel], np.float32)
        out = np.zeros(len(q), np.uint32); depth = np.zeros(1, np.uint32)
        shim.tie_shim_min_key(h, q.ctypes.data, cand.ctypes.data, counts.ctypes.data, K, len(q), out.ctypes.data, depth.ctypes.data)
        shim.tie_shim_free(h)
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        o1, o2, _ = tree.find_correspondences(S, r2)
        oi = np.full(len(S), -1, np.int64)
        oi[o2] = o1
        assert np.array_equal(out.astype(np.int64), oi[sel]), (name, int(np.count_nonzero(out.astype(np.int64) != oi[sel])))
        assert 8 <= int(depth[0]) <= 40, (name, int(depth[0]))


def test_threads_do_not_change_the_order(shim):
    """sub-trees are independent: the tables a pool of threads leaves are the serial build's (permutation, depths, splits, child sides)"""
    rng = np.random.default_rng(11)
    D = rng.random((400_000, 3), dtype=np.float32)
    D[rng.choice(len(D), 20_000, replace=False)] = D[rng.choice(len(D), 20_000, replace=False)]      # duplicated points
    D[:30_000, 2] = np.float32(0.25)                                                                 # a flat sheet: many equal coordinates on a split plane
    D = np.ascontiguousarray(D)
    a = shim.tie_shim_build_mt(D.ctypes.data, len(D), 1)
    b = shim.tie_shim_build_mt(D.ctypes.data, len(D), 8)
    c = shim.tie_shim_build_mt(D.ctypes.data, len(D), 3)
    try:
        assert shim.tie_shim_same_order(a, b) == 1
        assert shim.tie_shim_same_order(a, c) == 1
    finally:
        for h in (a, b, c):
            shim.tie_shim_free(h)


def test_degenerate_clouds_massive_duplicates_lines_tiny(shim, orc):
    """what the restated build must get right where nanoflann's splits degenerate: hundreds of copies of one point (zero-extent boxes: the
    plane split balances the copies by position), points on a line (two dimensions without extent), clouds smaller than a leaf.  Every
    query below has ALL copies of its nearest point as candidates; the pick (first met, and smallest traversal key) is the reference's."""
    rng = np.random.default_rng(23)
    A = np.array([0.25, 0.5, 0.75], np.float32); B = np.array([0.75, 0.25, 0.5], np.float32)
    clouds = {
        "300 + 300 copies of two points and 50 others": np.concatenate([np.tile(A, (300, 1)), np.tile(B, (300, 1)), rng.random((50, 3), dtype=np.float32)]),
        "a line with doubled points": np.concatenate([np.stack([np.linspace(0, 1, 400, dtype=np.float32), np.full(400, 0.5, np.float32), np.full(400, 0.5, np.float32)], 1)] * 2),
        "seven points, three of them the same": np.concatenate([rng.random((4, 3), dtype=np.float32), np.tile(A, (3, 1))]),
        "one point five times": np.tile(B, (5, 1)),
    }
    for name, D in clouds.items():
        D = np.ascontiguousarray(D[rng.permutation(len(D))].astype(np.float32))
        Q = np.ascontiguousarray(np.concatenate([rng.random((400, 3), dtype=np.float32), D[rng.integers(0, len(D), 100)] + np.float32(1e-3)]).astype(np.float32))
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        o1, o2, ov = tree.find_correspondences(Q, 1.0e9)
        assert len(o2) == len(Q)
        # all target points at exactly the nearest distance (the pinned expression), per query
        d2 = _pinned_d2(Q[:, None, :], D[None, :, :])
        tied = d2 == d2.min(axis=1, keepdims=True)
        cnt = tied.sum(axis=1).astype(np.int32)
        stride = int(cnt.max())
        cand = np.zeros((len(Q), stride), np.uint32)
        for k in range(len(Q)):
            cand[k, : cnt[k]] = np.nonzero(tied[k])[0]
        h = shim.tie_shim_build(D.ctypes.data, len(D))
        got = np.zeros(len(Q), np.uint32); got2 = np.zeros(len(Q), np.uint32); depth = np.zeros(1, np.uint32)
        shim.tie_shim_first_met(h, Q.ctypes.data, cand.ctypes.data, cnt.ctypes.data, stride, len(Q), got.ctypes.data)
        shim.tie_shim_min_key(h, Q.ctypes.data, cand.ctypes.data, cnt.ctypes.data, stride, len(Q), got2.ctypes.data, depth.ctypes.data)
        shim.tie_shim_free(h)
        want = np.zeros(len(Q), np.int64); want[o2] = o1
        assert np.array_equal(got.astype(np.int64), want), (name, int(np.count_nonzero(got.astype(np.int64) != want)))
        assert np.array_equal(got2.astype(np.int64), want), (name, "keys", int(np.count_nonzero(got2.astype(np.int64) != want)))
        assert int(depth[0]) <= 58, (name, int(depth[0]))
