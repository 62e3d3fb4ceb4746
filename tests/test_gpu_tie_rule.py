"""Option "tie_rule" = 1: correspondences equal the reference's where several target points are EXACTLY equidistant too.

The engine's default names the lowest target index among tied candidates, nanoflann the one its traversal meets first
(core/kd_tree.hpp:82-90) -- on the reference's raw sensor frames 660 of 120k queries are tied under the identity and 321 of them
get a different (equally near) point.  With the option the device lists the tied queries, the host walks a restatement of the
reference's tree for them (csrc/tie_order.hpp, pinned on the CPU by tests/test_tie_order_cpu.py) and the matches are re-pointed:
every index equals nanoflann's, and the loop equals the oracle's loop over the reference's searches to the same 1e-5 as clouds
without ties.
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


def test_tie_rule_reference_names_nanoflann_s_points(Context, orc):
    report = {}
    I = np.eye(4, dtype=np.float32)
    for name, D, N, S, r2 in _clouds():
        tree = orc.KDTree(D, use_ref=orc.ref_available())
        oi, _, ov = _ref_matches(tree, S, r2, len(S))
        # (1) a single search: default rule differs on ties, tie_rule = 1 equals the reference everywhere
        ctx = Context(); ctx.set_target(D, N); ctx.set_source(S)
        ctx.find_correspondences(I, float(r2), count=False)
        lo, _ = ctx.get_nn()
        differ_default = int(np.count_nonzero(_signed(lo) != oi))
        tied = ctx.tie_count(I, float(r2))
        ctx.set_option("tie_rule", 1)
        nfound = ctx.find_correspondences(I, float(r2))
        gi, gd = ctx.get_nn()
        gi = _signed(gi)
        seen, moved = ctx.tie_rule_stats()
        assert np.array_equal(gi, oi), (name, int(np.count_nonzero(gi != oi)))
        assert nfound == int(np.count_nonzero(oi >= 0))
        assert np.array_equal(gd[gi >= 0].view(np.uint32), ov.view(np.uint32)), name      # (ascending source order on both sides)
        assert seen == tied and moved == differ_default and moved > 0, (name, seen, tied, moved, differ_default)
        # the engine-level list too (ascending source index, correspondence_search_kd_tree.hpp:231)
        i1, i2, v = ctx.get_correspondences()
        assert np.array_equal(i1.astype(np.int64), oi[oi >= 0]) and np.array_equal(i2.astype(np.int64), np.nonzero(oi >= 0)[0])
        # (2) the loop: six iterations against the oracle's loop over the reference's searches
        po = orc.make_params(metric=1, max_iter=6, conv_tol=0.0, max_sq_dist=float(r2), mode=orc.MODE_MIXED)
        ro = orc.icp_run(D, N, S, po)
        p = capi.IcpParams()
        ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), 6, 0.0
        res = ctx.icp_run(p)
        Tg = np.array(res.T[:], np.float32).reshape(4, 4).T
        err = float(np.linalg.norm(Tg.astype(np.float64) - ro["T"].astype(np.float64)))
        seen_l, moved_l = ctx.tie_rule_stats()
        assert err <= 1e-5 and int(res.iterations) == 6, (name, err)
        assert ctx.last_warm_iterations() == 0
        # the set the loop leaves behind: the reference's at the last iteration's transform, index for index
        T = ctx.matches_transform()
        li, _ = ctx.get_nn()
        oi_l, _, _ = _ref_matches(tree, orc.transform_points(T, S), r2, len(S))
        assert np.array_equal(_signed(li), oi_l), (name, int(np.count_nonzero(_signed(li) != oi_l)))
        assert int(res.last_ncorr) == int(np.count_nonzero(oi_l >= 0))
        # (3) the default rule on the same pair, for the record: how far the ties move the loop
        ctx.set_option("tie_rule", 0)
        res0 = ctx.icp_run(p)
        T0 = np.array(res0.T[:], np.float32).reshape(4, 4).T
        ctx.close()
        report[name] = {"queries": int(len(S)), "tied_queries_identity": int(seen), "repointed_identity": int(moved),
                        "tied_queries_over_6_iterations": int(seen_l), "repointed_over_6_iterations": int(moved_l),
                        "T_minus_oracle_nanoflann_order_tie_rule_1": err,
                        "T_minus_oracle_nanoflann_order_tie_rule_0": float(np.linalg.norm(T0.astype(np.float64) - ro["T"].astype(np.float64)))}
    _report("tie_rule.json", report)


def test_tie_rule_refuses_what_it_does_not_cover(Context):
    d = syn.make_pair(20_000, perturb=0.3)
    ctx = Context(); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    ctx.set_option("tie_rule", 1)
    ctx.set_option("search_direction", 2)
    with pytest.raises(RuntimeError):
        ctx.find_correspondences(np.eye(4, dtype=np.float32), float(d["max_sq_dist"]))
    ctx.close()
