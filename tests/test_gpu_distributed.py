"""GPU: the sharded ICP loops of cilantro_amd/distributed.py with the PRODUCT's engines composed across PROCESSES -- one process per
rank (torch.distributed.run), both ranks on the box's one GPU, the all-reduce of the 48 partial sums over gloo.  (tests/
test_distributed_cpu.py drives the same protocols with the oracle as the per-rank engine; RCCL with more than one rank needs more than
one GPU: the driver's SCALE runs.)  Source shards and spatial slabs, a slab guard that fires (all ranks re-partition), and a target
with duplicated points: ties met on some rank make EVERY rank load the whole target's order tables and run again (one MAX over the
ranks), the result equals the oracle's loop over the reference's searches."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(world, mode, kind, n, iters):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), mode, kind, str(n), str(iters)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=500)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        stdout, stderr = proc.communicate()
        raise AssertionError("distributed GPU worker timed out\n" + stdout[-2000:] + stderr[-2000:])
    assert proc.returncode == 0, stdout[-3000:] + stderr[-3000:]
    line = [l for l in stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("mode,kind", [("source", "plain"), ("slab", "plain"), ("slab0.01", "plain"), ("source", "dup"), ("slab", "dup"), ("tshard", "plain"), ("tshard", "dup")])
def test_hip_engines_across_two_processes(orc, hip_lib, mode, kind):
    import _dist_gpu_worker as w

    n, iters = 200_000, 6
    r = _run(2, mode, kind, n, iters)
    rows = sorted(r["rows"], key=lambda x: x["rank"])
    assert r["world"] == 2 and all(row["it"] == iters for row in rows)
    T0, T1 = np.array(rows[0]["T"]), np.array(rows[1]["T"])
    assert np.array_equal(T0, T1)                               # the same sums, the same epilogue: bit-identical transforms on every rank
    d = w.clouds(kind, n)
    po = orc.make_params(metric=1, max_iter=iters, conv_tol=0.0, max_sq_dist=float(d["max_sq_dist"]), mode=orc.MODE_MIXED)
    ro = orc.icp_run(d["dst"], d["dst_n"], d["src"], po)
    err = float(np.linalg.norm(T0 - ro["T"].astype(np.float64)))
    assert err <= 2e-6 and rows[0]["nc"] == ro["last_ncorr"], (mode, kind, err, rows[0]["nc"], ro["last_ncorr"])
    if kind == "dup":
        assert all(row["tables_loaded"] for row in rows)       # ties were met: every rank loaded the order of the WHOLE target
    if mode == "slab0.01":
        assert all(row["repartitions"] >= 1 for row in rows)   # the guard fired: all ranks cut their slabs again


@pytest.mark.parametrize("kind", ["plain", "empty", "tol", "kd", "big"])
def test_sharded_kmeans_across_two_processes(hip_lib, kind):
    """SURVEY.md 8(e), last row: KMeans with the points sharded (HIP shards: cilhip_kmeans_shard_*), centroids replicated, one all-reduce
    of 4k + 1 integers per Lloyd iteration.  The sums are exact fixed-point integers: centroids, labels and the iteration count are the
    single-device cilhip_kmeans3f_ex's BIT FOR BIT -- through the pruned assignment (k = 1024), the kd branch with exact ties, the
    tolerance exit and the empty-cluster repair whose farthest member lives on one of the ranks."""
    n, iters = (400_000, 6) if kind == "big" else (120_000, 100 if kind == "tol" else 8)
    r = _run(2, "kmeans", kind, n, iters)
    rows = sorted(r["rows"], key=lambda x: x["rank"])
    assert r["world"] == 2 and rows[0]["cent"] == rows[1]["cent"] and rows[0]["it"] == rows[1]["it"]
    one_c, one_l = np.array(rows[0]["one_cent"]), np.array(rows[0]["one_lab"])
    lab = np.concatenate([np.array(row["lab"]) for row in rows])
    assert np.array_equal(np.array(rows[0]["cent"]), one_c), (kind, np.abs(np.array(rows[0]["cent"]) - one_c).max())
    assert np.array_equal(lab, one_l) and rows[0]["it"] == rows[0]["one_it"], (kind, int((lab != one_l).sum()), rows[0]["it"], rows[0]["one_it"])
    if kind == "tol":
        assert rows[0]["it"] < iters


def test_sharded_ransac_counts_across_two_processes(hip_lib):
    r = _run(2, "ransac", "plain", 300_000, 1)
    rows = sorted(r["rows"], key=lambda x: x["rank"])
    assert rows[0]["counts"] == rows[1]["counts"] == rows[0]["one_counts"] and max(rows[0]["counts"]) > 0
