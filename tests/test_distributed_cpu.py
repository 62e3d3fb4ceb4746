"""CPU: the N>1 path (source-sharded ICP, one all-reduce(sum) per iteration) with world_size 2 over
gloo.  The per-rank compute is a test-only engine (the oracle); what is under test is the product's
sharding / reduction protocol in cilantro_amd/distributed.py."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

from cilantro_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(world, metric, n, mode=""):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(metric), str(n)] + ([mode] if mode else [])
    env = dict(os.environ, OMP_NUM_THREADS="1")
    # own process group, killed as a whole on a timeout: a worker that outlives the launcher would keep the host busy
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        stdout, stderr = proc.communicate()
        raise AssertionError("distributed worker timed out\n" + stdout[-2000:] + stderr[-2000:])
    assert proc.returncode == 0, stdout[-2000:] + stderr[-2000:]
    line = [l for l in stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("metric", [0, 1])
def test_sharded_icp_world2_matches_single_process(orc, metric):
    n = 6000
    r2 = _run(2, metric, n)
    assert r2["world"] == 2 and r2["identical"]          # every rank ends with the bit-identical transform
    d = syn.make_pair(n, perturb=0.5)
    p = orc.make_params(metric=metric, max_iter=12, conv_tol=1e-6, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    ref = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    T2 = np.array(r2["T"], np.float64)
    assert np.linalg.norm(T2 - ref["T"]) <= 1e-5
    assert abs(r2["iters"] - ref["iterations"]) <= 1 and r2["ncorr"] == ref["last_ncorr"]
    assert np.linalg.norm(T2 - d["T_true"]) < 5e-4


@pytest.mark.parametrize("metric", [0, 1])
def test_target_sharded_icp_world2_matches_single_process(orc, metric):
    """BASELINE configs[3] protocol: target split by index over 2 ranks, MIN all-reduce of packed
    (d2, global index) keys, each rank accumulates the pairs its shard won, SUM all-reduce."""
    n = 6000
    r2 = _run(2, metric, n, "tshard")
    assert r2["world"] == 2 and r2["identical"] and r2["iters"] == 8
    d = syn.make_pair(n, perturb=0.5)
    p = orc.make_params(metric=metric, max_iter=8, conv_tol=0.0, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    ref = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    assert np.linalg.norm(np.array(r2["T"], np.float64) - ref["T"]) <= 1e-5
    assert r2["ncorr"] == ref["last_ncorr"]


def test_target_sharded_ties_follow_the_reference_s_order_world2(orc):
    """Exactly equidistant nearest points inside the index shards of a target and ACROSS them (doubled and tripled points, shuffled):
    TargetShardedRigidICP runs once with one key per query, every rank notices (its own ties; a nearest point as far as the winner's
    that is not the winner), all agree (all-reduce MAX), load the whole target's order and run again with the third collective -- the
    MIN of the traversal keys.  The last iteration's pairs, gathered from the ranks that won them: every matched query won by exactly
    one rank, index for index the reference's nanoflann over the whole target (one key alone names another point for hundreds)."""
    r = _run(2, 1, 12000, "tshardties")
    assert r["world"] == 2 and r["identical"] and r["ordered"] == [True, True] and r["iters"] == 6
    assert r["won_once"] and r["mismatches"] == 0, r
    assert r["lowest_index_would_differ"] > 100, r
    assert r["ncorr"] == r["oracle_ncorr"] and r["T_err"] <= 1e-5, r


@pytest.mark.parametrize("metric", [0, 1])
def test_slab_sharded_icp_world2_matches_single_process(orc, metric):
    """SURVEY 8(e) partitioning B: target and source cut into spatial slabs (halo = radius + slack), one all-reduce(sum) of
    the partial sums per iteration and nothing else.  With the default slack the partition survives the run; with a slack
    of a hundredth of a cell the guard fires, the ranks re-partition under the last checked transform and continue --
    either way the result is the single-process run's."""
    n = 6000
    d = syn.make_pair(n, perturb=0.5)
    p = orc.make_params(metric=metric, max_iter=12, conv_tol=1e-6, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    ref = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    for mode, want_repart in (("slab", False), ("slab0.01", True)):
        r2 = _run(2, metric, n, mode)
        assert r2["world"] == 2 and r2["identical"]
        assert (r2["repartitions"] > 0) == want_repart, r2["repartitions"]
        assert sum(r2["n_src"]) == n and max(r2["n_src"]) - min(r2["n_src"]) <= 1          # every query owned exactly once, balanced
        assert all(nd < 0.9 * n for nd in r2["n_dst"])                                   # a slab + halo (wide at this tiny size: radius + slack = 6 h), not the whole target
        T2 = np.array(r2["T"], np.float64)
        assert np.linalg.norm(T2 - ref["T"]) <= 1e-5, np.linalg.norm(T2 - ref["T"])
        assert abs(r2["iters"] - ref["iterations"]) <= 1 and r2["ncorr"] == ref["last_ncorr"]
        # estimate() once more from the identity on the same object: an engine left by a re-partition is cut again under the start
        # (at least the one re-cut + the run's own), the result is the first run's
        assert r2["rerun"]["same"], r2["rerun"]
        if want_repart:
            assert r2["rerun"]["recut"] >= 1 + r2["repartitions"], r2["rerun"]


def test_loops_around_an_engine_that_runs_blocks_of_iterations_itself(orc):
    """cilhip_icp_iterate_ranked (one process per device: blocks of iterations inside the library, its own all-reduce) changes the
    CONTROL FLOW of ShardedRigidICP / SlabShardedRigidICP: blocks up to the next check, early stop on convergence, a new
    communicator for the engine a re-partition creates.  Played on the CPU by the test-only engine with the same interface
    (`native`, `enable_native_allreduce`, `iterate`) over gloo, world size 2: same results as the per-iteration protocol and as the
    single-process oracle run, with and without a guard that fires."""
    n = 6000
    d = syn.make_pair(n, perturb=0.5)
    p = orc.make_params(metric=1, max_iter=12, conv_tol=1e-6, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_MIXED)
    ref = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
    plain = _run(2, 1, n)
    r = _run(2, 1, n, "native")
    assert r["world"] == 2 and r["identical"]
    assert np.linalg.norm(np.array(r["T"], np.float64) - ref["T"]) <= 1e-5 and r["ncorr"] == ref["last_ncorr"]
    # (checked every 3 iterations instead of every one: the loop may run up to 2 iterations past the tolerance; converged either way)
    assert plain["iters"] <= r["iters"] <= plain["iters"] + 2
    for mode, want_repart in (("nslab", False), ("nslab0.01", True)):
        r2 = _run(2, 1, n, mode)
        base = _run(2, 1, n, mode[1:])
        assert r2["world"] == 2 and r2["identical"] and (r2["repartitions"] > 0) == want_repart
        assert r2["T"] == base["T"] and r2["iters"] == base["iters"] and r2["ncorr"] == base["ncorr"] and r2["repartitions"] == base["repartitions"]
        assert np.linalg.norm(np.array(r2["T"], np.float64) - ref["T"]) <= 1e-5


def test_rank_communicator_handshake_over_gloo():
    """distributed.init_rank_comm (what bench.py --gpus N and the engines call before cilhip_icp_iterate_ranked) with a stand-in
    context, world size 3 over gloo: the id is created once, on rank 0; every rank receives the same 128 bytes and joins with its own
    rank; a rank that cannot join -- or a rank 0 that cannot create the id -- makes every rank fall back, and leave what it joined."""
    r = _run(3, 0, 0, "rankcomm")
    assert r["world"] == 3
    rows = sorted(r["rows"], key=lambda x: x["rank"])
    want = bytes((np.arange(128) * 7 % 251).astype(np.uint8)).hex()
    for k, row in enumerate(rows):
        assert row["r1"] is True and row["got"] == [want, 3, k]
        assert row["made"] == (2 if k == 0 else 0)                       # (two successful creations, both on rank 0)
        assert row["r2"] is False and row["bad_destroyed"] == 1          # one rank failed: all leave
        assert row["r3"] is False and row["none_got"] is None            # no id: nobody tries to join
        # a rank that cannot even open librccl says so BEFORE the collective init: nobody enters it (entering it alone would hang)
        assert row["r4"] is False and row["r4_entered"] == 0
        assert row["r5"] is True and row["r5_entered"] == 1


def test_slab_partition_is_exact(orc):
    """Every owned query finds, inside its own slab + halo, exactly the neighbour the whole target gives it -- under the
    partition transform and under any transform that moves no point by more than the slack along the axis."""
    from cilantro_amd import distributed

    n = 20000
    d = syn.make_pair(n, perturb=0.5)
    h = d["h"]
    T0 = np.eye(4, dtype=np.float32)
    full = orc.KDTree(d["dst"])
    for world in (2, 3, 5):
        part = distributed.SlabPartition.plan(d["dst"], d["src"], T0, float(d["max_sq_dist"]), world)
        owned = np.zeros(n, np.int32)
        for T in (T0, d["T_true"].astype(np.float32)):
            ax = part.axis
            disp = np.abs(part._image_coord(T, d["src"], ax) - part._image_coord(T0, d["src"], ax)).max()
            assert disp <= part.slack                                      # (what the guard bounds from the box corners)
            for r in range(world):
                dst_l, _, src_l = part.select(r, d["dst"], d["dst_n"], d["src"])
                if T is T0:
                    owned[part.src_index] += 1
                q = orc.transform_points(T, src_l)
                li, ls, lv = orc.KDTree(dst_l).find_correspondences(q, d["max_sq_dist"])
                gi, gs, gv = full.find_correspondences(q, d["max_sq_dist"])
                assert np.array_equal(ls, gs) and np.array_equal(part.dst_index[li], gi) and np.array_equal(lv, gv)
        assert (owned == 1).all()


@pytest.mark.parametrize("case", ["", "empty", "tol", "kd"])
def test_sharded_kmeans_world2_is_the_one_shard_run(orc, case):
    """SURVEY.md 8(e), last row: KMeans with the points sharded, centroids replicated, all-reduce of k x (3 sums + count).  The sums
    are exact integers: the sharded loop's centroids, labels and iteration count equal the same loop over ONE shard bit for bit
    (also through the empty-cluster repair, whose farthest member lives on some rank, and the tolerance exit); against the oracle's
    KMeans (the reference's f32 serial sums) the centroids agree to 1e-5."""
    r = _run(2, 0, 20000, "kmeans" + case)
    assert r["world"] == 2 and r["same_centroids_on_all_ranks"] and r["equals_one_shard"], r
    assert r["centroid_err_vs_oracle"] <= 1e-5 and r["label_mismatches_vs_oracle"] <= 20 and abs(r["it"] - r["oracle_it"]) <= 1, r


def test_sharded_ransac_counts_world2(orc):
    """... and the RANSAC scoring pass: per-hypothesis inlier counts of the shards, summed"""
    r = _run(2, 0, 30000, "ransac")
    assert r["world"] == 2 and r["equal"] and r["best"] == 0, r


def test_kmeans_scale_exponent_python_and_library_agree():
    """every shard of a sharded KMeans run must use ONE fixed-point scale: the Python loop's restatement against the library's (pure host
    code: callable without a device)"""
    import ctypes as C

    from cilantro_amd import capi, distributed_models as dm

    L = capi.load()
    for maxabs in (0.0, 1e-30, 0.4999, 0.5, 1.0, 1.0000001, 3.99, 4.0, 1234.5, 3.0e38):
        for n in (1, 2, 3, 1000, 1 << 20, (1 << 20) + 1, 50_000_000, 400_000_000, 0xFFFFFFEF):
            assert dm.scale_exponent(maxabs, n) == L.cilhip_kmeans_scale_exponent(C.c_double(np.float32(maxabs)), C.c_size_t(n)), (maxabs, n)
