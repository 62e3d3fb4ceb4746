#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native rigid ICP engine.

A "step" is ONE ICP iteration (LDS-tiled kNN correspondence search kernel, streaming residual
accumulation kernel, then the on-device 6x6 / 3x3 solve) over one synthetic cloud pair resident in HBM.

  N=1  : BASELINE.json configs[2]: 10M <-> 10M synthetic cloud with normals, point-to-plane
         (SimpleCombinedMetricRigidICP3f defaults w_p2p=0, w_p2pl=1), SURVEY.md 8(d) recipe.
  N>1  : weak scaling, one process per GPU: every rank holds the full 10M target and its own 10M
         source shard (source points are independent work units); the only exchange
         is the all-reduce(sum) of 48 doubles per iteration over RCCL (torch.distributed "nccl").

Timed region = exactly K iterations from T0 = identity with conv_tol = 0 (never early-exits),
bracketed by barrier + torch.cuda.synchronize(); MAX over ranks; rank 0 prints one JSON line.
`value` = correspondence pairs/s of the whole job (all ranks' source points x K / time);
`icp_iterations_per_sec` = K / time is reported beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=10_000_000, help="points per cloud (target and per-rank source)")
    ap.add_argument("--metric", choices=["p2plane", "p2p"], default="p2plane")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="source points of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(d, metric, n_sample, T):
    """Reference CPU path timed on this host: the reference's own nanoflann (oracle/_ref, OpenMP
    schedule(dynamic,256) loop as correspondence_search_kd_tree_utilities.hpp:26) for the kNN pass +
    the oracle's accumulation/solve, on a bounded sample of the source."""
    from oracle import oracle as orc

    cores = os.cpu_count() or 1
    use_ref = orc.ref_available()
    t0 = time.perf_counter()
    tree = orc.KDTree(d["dst"], use_ref=use_ref)      # single-thread build, as the reference
    t_build = time.perf_counter() - t0
    src = d["src"][:n_sample]
    q = orc.transform_points(T, src)
    best = None
    for _ in range(2):                                # warm + measured
        t0 = time.perf_counter()
        di, si, d2 = tree.find_correspondences(q, d["max_sq_dist"], num_threads=cores)
        t_knn = time.perf_counter() - t0
        best = t_knn if best is None else min(best, t_knn)
    p = orc.make_params(metric=1 if metric == "p2plane" else 0, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_F32)
    t0 = time.perf_counter()
    orc.icp_update(d["dst"], d["dst_n"], src, T, di, si, p)
    t_est = time.perf_counter() - t0
    pairs_s = len(src) / (best + t_est)
    return {
        "value": pairs_s, "unit": "pairs/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": f"{len(src)} of {len(d['src'])} source points vs full {len(d['dst'])}-point target, 1 iteration "
                  f"(kNN {best:.3f}s on {cores} OpenMP threads + accumulate/solve {t_est:.3f}s single thread); "
                  f"one-off kd-tree build {t_build:.2f}s (1 thread, excluded)",
        "knn_s": best, "estimate_s": t_est, "tree_build_s": t_build,
        "icp_iterations_per_sec_equiv": pairs_s / len(d["src"]),
    }


def main():
    a = parse()
    import torch

    from cilantro_amd import capi
    from cilantro_amd import synthetic as syn
    from cilantro_amd.icp import Context

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dist = None
    # CILHIP_BENCH_FORCE_SHARDED=1 (under torchrun): run the sharded protocol + RCCL even with one rank,
    # to exercise exactly the code path the multi-GPU runs take
    sharded = world > 1 or (os.environ.get("CILHIP_BENCH_FORCE_SHARDED") == "1" and "RANK" in os.environ)
    if sharded:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == a.gpus or world == 1, (world, a.gpus)

    n = a.n
    with_normals = a.metric == "p2plane"
    # rank r: same target, its own window of source points / noise stream
    d = syn.make_pair(n, n, with_normals=with_normals, src_offset=rank * 7919)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(local_rank, stream)
    dst_t = torch.from_numpy(d["dst"]).cuda()
    nrm_t = torch.from_numpy(d["dst_n"]).cuda() if with_normals else None
    src_t = torch.from_numpy(d["src"]).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.set_target(dst_t, nrm_t)
    ctx.set_source(src_t)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    gi = ctx.grid_info()

    import ctypes as C

    p = capi.IcpParams()
    ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric = capi.METRIC_COMBINED if with_normals else capi.METRIC_POINT_TO_POINT
    p.conv_tol = 0.0
    p.max_sq_dist = float(d["max_sq_dist"])
    T0 = np.eye(4, dtype=np.float32)

    sums = torch.zeros(capi.SUMS_LEN, dtype=torch.float64, device="cuda")
    gmean = None
    if sharded:
        _, sm = ctx.means()
        m = torch.tensor(sm.astype(np.float64) * n, dtype=torch.float64, device="cuda")
        dist.all_reduce(m)
        gmean = (m.cpu().numpy() / (n * world)).astype(np.float32)

    def run(iters, timing):
        p.max_iter = iters
        if not sharded:
            ctx.enable_kernel_timing(timing)
            return ctx.icp_run(p, T0)
        ctx.enable_kernel_timing(timing)
        ctx.icp_begin(p, T0, gmean)
        for _ in range(iters):
            ctx.icp_partial_sums(sums.data_ptr())
            dist.all_reduce(sums)
            ctx.icp_apply_sums(sums.data_ptr())
        return ctx.icp_state()

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    run(a.warmup, False)
    barrier()
    t0 = time.perf_counter()
    res = run(a.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert int(res.iterations) == a.steps, (res.iterations, a.steps)

    if rank == 0:
        T = np.array(res.T[:], np.float32).reshape(4, 4).T
        err_true = float(np.linalg.norm(T - d["T_true"]))
        loop_ms, _, launches = ctx.last_timing()
        search_ms, acc_ms = ctx.last_timing2()
        ns, nd, nc = n, n, int(res.last_ncorr)
        # Dominant kernel = the kNN correspondence-search kernel.  Algorithmic bytes of one kNN pass
        # (B_knn): read every source point once (12 B), every target point once (12 B),
        # write (index, d2) per source point (8 B)  =>  20*Ns + 12*Nd for a stand-alone search.
        # Inside the ICP loop (no post-filters) nothing reads the squared distances, so the kernel does not write them:
        # 4 B per source point less than SURVEY's B_knn -- counted as such, not inflated.
        alg_bytes = 16.0 * ns + 12.0 * nd
        # streaming accumulation kernel (point-to-plane): B_acc = 16*Ns + 24*Nc  
        acc_bytes = 16.0 * ns + (24.0 if with_normals else 12.0) * nc
        # HBM bytes per launch of the search kernel(s) from the committed PMC passes (separate rocprofv3 --pmc runs,
        # corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes); only quoted for the workload it was measured on
        traffic = None
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")))
            w = tj["workload"]
            if (w["n_target"], w["n_source_per_gpu"], bool(w["with_normals"])) == (nd, ns, bool(with_normals)):
                traffic = float(tj["traffic_bytes_per_launch"])
        except Exception:
            traffic = None
        roof = None
        if launches > 0:      # (sharded runs: rank 0's own kernels)
            avg_ms = search_ms / launches
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            # the measured device copy bandwidth beside the spec peak (1 GiB torch copy = read + write)
            copy_gbs = None
            try:
                xb = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); yb = torch.empty_like(xb)
                yb.copy_(xb); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    yb.copy_(xb)
                e1.record(); torch.cuda.synchronize()
                copy_gbs = 5 * 2 * xb.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                del xb, yb
            except Exception:
                copy_gbs = None
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "measured_copy_bandwidth_GBps": copy_gbs,
                    "traffic": traffic, "kernel": "k_search_tiled + k_search_todo (kNN correspondence search, LDS-tiled)",
                    "avg_kernel_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": alg_bytes,
                    "accumulate_kernel": {"avg_kernel_ms": acc_ms / launches,
                                          "algorithmic_bytes_per_launch": acc_bytes,
                                          "achieved_GBps": acc_bytes / max(acc_ms / launches * 1e-3, 1e-12) / 1e9}}
        out = {
            "metric": "ICP corr. pairs/sec (+ iterations/sec), synthetic uniform clouds",
            "value": n * world * a.steps / dt, "unit": "pairs/s",
            "icp_iterations_per_sec": a.steps / dt,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 search / f64 accumulate+solve",
            "data": "synthetic",
            "config": {"workload": f"{n/1e6:g}M<->{n/1e6:g}M synthetic float3 clouds"
                                   + (" with normals, point-to-plane (SimpleCombinedMetricRigidICP3f)" if with_normals
                                      else ", point-to-point (SimplePointToPointMetricRigidICP3f)"),
                       "n_target": nd, "n_source_per_gpu": ns, "max_sq_dist": float(d["max_sq_dist"]),
                       "iterations": a.steps, "conv_tol": 0.0, "sharding": "source-sharded, target replicated, all-reduce(sum) of 48 f64 per iteration" if sharded else "none",
                       "grid": [gi.nx, gi.ny, gi.nz], "grid_cell": gi.cell, "grid_avg_occupancy": gi.avg_occupancy},
            "setup_ms": t_setup * 1e3, "loop_ms_hip_events": loop_ms,
            "last_ncorr": nc, "T_err_vs_truth_frobenius": err_true,
            "roofline": roof,
        }
        if not sharded and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(d, a.metric, min(a.cpu_sample, n), T0)
            except Exception as e:  # the baseline is a report, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
