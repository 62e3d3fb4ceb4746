#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native rigid ICP engine.

A "step" is ONE ICP iteration over one synthetic cloud pair resident in HBM: the LDS-tiled kNN correspondence
search with the residual accumulation inside the tile (k_search_tiled<metric> + its clean-up pass), the cross-block
reduction and the on-device 6x6 / 3x3 solve.

  --config c3 (default)  BASELINE.json configs[2]: 10M <-> 10M synthetic cloud with normals, point-to-plane
                         (SimpleCombinedMetricRigidICP3f defaults w_p2p = 0, w_p2pl = 1), SURVEY.md 8(d) recipe --
                         the configuration BASELINE's metric is quoted on.
           c2            configs[1]: 1M <-> 1M, point-to-point (SimplePointToPointMetricRigidICP3f)
           c4 / c4_1gpu  configs[3]: 10M source points against an 80M-point target, combined metric 0.1 / 1.0
           kmeans        configs[4]: KMeans3f k = 1024 on 50M points (step = one Lloyd iteration); --gpus N: 50M points PER RANK, centroids
                         replicated, one all-reduce of 4k + 1 int64 per iteration (last row; weak scaling)
           ransac        configs[4]: plane RANSAC scoring on 50M points (step = one pass of 128 hypotheses)
  --gpus N (one process per GPU; backend "nccl" = RCCL).  Started without a launcher (`python bench.py --gpus N`) the script
      re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; started
      under one (RANK / WORLD_SIZE set) it checks that WORLD_SIZE == N.  Fewer than N visible devices: exits non-zero.
      c2 / c3 (default --scaling strong) and c4: spatial slabs (SURVEY.md 8(e) partitioning B, cilantro_amd.distributed.SlabPartition):
          BASELINE's metric is "10M<->10M at 1/2/4/8 GPUs" -- the SAME registration on N GPUs.  Rank r owns the target points of
          its slab plus a halo and the source points that fall into the slab; one all-reduce(sum) of 48 f64 per iteration
          ("per-rank partial covariances reduced via RCCL all-reduce").  Total work is fixed as N grows.
      --scaling weak: every rank holds the full target and its own source shard of the config's size (source points are
          independent work units); the same single all-reduce.
      The SCALE commands: `python bench.py --gpus N` for N = 1, 2, 4, 8 (c3) and `python bench.py --config c4 --gpus 8`.

Timed region = exactly K steps from T0 = identity with conv_tol = 0 (never early-exits), bracketed by barrier +
torch.cuda.synchronize(); MAX over ranks; rank 0 prints one JSON line.  `value` = correspondence pairs/s of the whole
job (source points of all ranks x K / time); `icp_iterations_per_sec` = K / time is reported beside it.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_NOFMA_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: 157.3 TFLOP/s f32 vector peak counts FMA as 2; pinned, uncontracted f32 ops run at half of it

ICP_CONFIGS = {
    # name: (n_target, n_source, source stride, metric, w_p2p, w_p2pl, label)
    "c2": (1_000_000, 1_000_000, 1, "p2p", 0.0, 0.0, "point-to-point (SimplePointToPointMetricRigidICP3f)"),
    "c3": (10_000_000, 10_000_000, 1, "p2plane", 0.0, 1.0, "with normals, point-to-plane (SimpleCombinedMetricRigidICP3f)"),
    "c4": (80_000_000, 10_000_000, 8, "combined", 0.1, 1.0, "with normals, combined metric w_p2p = 0.1, w_p2pl = 1 (SimpleCombinedMetricRigidICP3f)"),
}
ICP_CONFIGS["c4_1gpu"] = ICP_CONFIGS["c4"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c4_1gpu", "kmeans", "ransac"], default="c3")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="--gpus > 1: strong (default; slabs, total work fixed) or weak (c4 is always strong)")
    ap.add_argument("--selftest-spawn", action="store_true", help="(tests) exercise the self-launch path on CPU: gloo, one all-reduce, no GPU work")
    ap.add_argument("--n", type=int, default=None, help="override the number of target points (source scaled alike): quick runs only")
    ap.add_argument("--metric", choices=["p2plane", "p2p"], default=None, help="(compatibility) c3 sizes with another metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold-start / converging-trajectory measurements")
    ap.add_argument("--no-other-configs", action="store_true", help="default line only: skip the c2 / kmeans / ransac / c4_1gpu figures it carries as `other_configs`")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="source points of the CPU baseline sample")
    return ap.parse_args()


TIMING_STRIDE = 4      # kernel timing inside the timed region: iterations 0-2 and every 4th one carry events


def stratified_form_timing(ft, forms, timed):
    """Per-form kernel time of a run whose iterations were timed by SAMPLE (option kernel_timing_stride), as {form: (ms, n)} with
    ms / n the estimate of the form's mean launch.  The first iteration of a stretch of one form is not like the others (the first
    warm-started iteration after a cold one searches a hundred times more queries than the ones after it) and the sample always
    holds it: the mean is formed per stratum -- first of a stretch / the rest -- and weighted by how many iterations of the run
    each stratum has, not by how many of them happened to be timed.  forms: the run's form per iteration (its trace); timed:
    [(iteration, ms)].  Without a trace (sharded runs) the plain sums of `ft` are returned."""
    if not forms or not timed:
        return ft
    strat = lambda i: (forms[i], i == 0 or forms[i - 1] != forms[i])
    count, tsum, tn = {}, {}, {}
    for i in range(len(forms)):
        count[strat(i)] = count.get(strat(i), 0) + 1
    for i, ms in timed:
        if i < len(forms):
            tsum[strat(i)] = tsum.get(strat(i), 0.0) + ms
            tn[strat(i)] = tn.get(strat(i), 0) + 1
    out = {}
    for f in set(forms):
        num = den = 0.0
        n_timed = 0
        for first in (True, False):
            k = (f, first)
            if count.get(k, 0) and tn.get(k, 0):
                num += tsum[k] / tn[k] * count[k]
                den += count[k]
                n_timed += tn[k]
        if den > 0:
            out[f] = (num / den * n_timed, n_timed)      # (ms, n) with ms / n = the weighted mean
    return {f: out.get(f, ft.get(f, (0.0, 0))) for f in set(ft) | set(out)}


def source_hash(files=("kernels.hip", "warm.hip", "epilogue.hip", "search_device.hpp", "affine_device.hpp", "internal.hpp", "c_api.hip", "grid_build.hip", "solve.hpp")):
    """sha256 over the kernel sources: profile-derived numbers (roofline.traffic) are only quoted for the build they were measured on."""
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "cilantro_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = {"kmeans": ("kmeans.hip", "internal.hpp", "tie_build.hip"), "ransac": ("ransac.hip",)}


def config_traffic(config, workload):
    """(HBM bytes per launch of the configuration's dominant kernel, note) from profiles/r*_traffic_<config>.json (tools/make_traffic_json.py
    --config), quoted only for the build (hash of the kernel's sources) and the workload it was counted on; (None, why) otherwise."""
    import glob
    note = "no PMC measurement of this build / workload committed"
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{config}.json")), reverse=True):
        try:
            tj = json.load(open(tf))
            if tj.get("source_hash") == source_hash(KERNEL_SOURCES[config]) and tj.get("workload") == workload and float(tj.get("traffic_bytes_per_launch") or 0.0) > 0.0:
                return float(tj["traffic_bytes_per_launch"]), os.path.basename(tf) + ": " + tj.get("method", "")
            if note.startswith("no PMC"):
                note = "profiles/" + os.path.basename(tf) + " was measured on another build or workload: not quoted"
        except Exception:
            pass
    return None, note


def cpu_baseline_icp(d, metric, w_p2p, w_p2pl, n_sample, T):
    """Reference CPU path timed on this host: the reference's own nanoflann (oracle/_ref, OpenMP schedule(dynamic,256) loop as
    correspondence_search_kd_tree_utilities.hpp:26) for the kNN pass + the oracle's accumulation / solve in the reference's f32
    arithmetic, on a bounded sample of the source against the full target.  Warm-up, then the median of 5."""
    from oracle import oracle as orc

    cores = os.cpu_count() or 1
    use_ref = orc.ref_available()
    t0 = time.perf_counter()
    tree = orc.KDTree(d["dst"], use_ref=use_ref)      # single-thread build, as the reference
    t_build = time.perf_counter() - t0
    src = d["src"][:n_sample]
    q = orc.transform_points(T, src)
    knn = []
    for k in range(6):                                # 1 warm-up + 5 measured
        t0 = time.perf_counter()
        di, si, d2 = tree.find_correspondences(q, d["max_sq_dist"], num_threads=cores)
        if k:
            knn.append(time.perf_counter() - t0)
    p = orc.make_params(metric=0 if metric == "p2p" else 1, w_p2p=w_p2p, w_p2pl=w_p2pl, max_sq_dist=d["max_sq_dist"], mode=orc.MODE_F32)
    est = []
    # (the reference's default build reduces the estimator's loops with OpenMP: transform_estimation.hpp:284-344)
    est_threads = cores if metric != "p2p" else 1
    orc.set_estimator_threads(est_threads)
    try:
        for k in range(6):
            t0 = time.perf_counter()
            orc.icp_update(d["dst"], d["dst_n"], src, T, di, si, p)
            if k:
                est.append(time.perf_counter() - t0)
    finally:
        orc.set_estimator_threads(1)
    t_knn, t_est = statistics.median(knn), statistics.median(est)
    pairs_s = len(src) / (t_knn + t_est)
    return {
        "value": pairs_s, "unit": "pairs/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": f"{len(src)} of {len(d['src'])} source points vs full {len(d['dst'])}-point target, 1 iteration, median of 5 after a warm-up "
                  f"(kNN {t_knn:.3f}s on {cores} OpenMP threads + accumulate/solve {t_est:.3f}s, its accumulation loops on {est_threads} OpenMP thread(s) as the "
                  f"reference's default build reduces them, f32 as the reference); "
                  f"one-off kd-tree build {t_build:.2f}s (1 thread, excluded)",
        "knn_s": t_knn, "knn_s_min_max": [min(knn), max(knn)], "estimate_s": t_est, "tree_build_s": t_build,
        "icp_iterations_per_sec_equiv": pairs_s / len(d["src"]),
    }


def copy_bandwidth(torch):
    """the measured device copy bandwidth beside the spec peak (1 GiB torch copy = read + write)."""
    try:
        xb = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); yb = torch.empty_like(xb)
        yb.copy_(xb); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            yb.copy_(xb)
        e1.record(); torch.cuda.synchronize()
        return 5 * 2 * xb.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    except Exception:
        return None


def bench_icp(a, torch, rank, world, local_rank, emit=True):
    from cilantro_amd import capi, distributed
    from cilantro_amd import synthetic as syn
    from cilantro_amd.icp import Context

    nd, ns, stride, metric, w_p2p, w_p2pl, label = ICP_CONFIGS[a.config]
    if a.metric is not None and a.config == "c3":
        metric = a.metric
        label = ICP_CONFIGS["c2"][6] if metric == "p2p" else label
    if a.n is not None:
        ns = max(1, a.n * ns // nd); nd = a.n
    with_normals = metric != "p2p"
    dist = None
    # CILHIP_BENCH_FORCE_SHARDED=1 (under torchrun): run the sharded protocol + RCCL even with one rank,
    # to exercise exactly the code path the multi-GPU runs take
    sharded = world > 1 or (os.environ.get("CILHIP_BENCH_FORCE_SHARDED") == "1" and "RANK" in os.environ)
    rccl_ranks = 1
    if sharded:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)                      # how many ranks the RCCL all-reduce really spans
        rccl_ranks = int(ones.item())
        if rccl_ranks != a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: the RCCL all-reduce spans {rccl_ranks} rank(s)")
    strong = sharded and (a.config in ("c4", "c4_1gpu") or a.scaling != "weak")

    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(local_rank, stream)
    p = capi.IcpParams()
    ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric = capi.METRIC_POINT_TO_POINT if metric == "p2p" else capi.METRIC_COMBINED
    p.w_p2p, p.w_p2pl = (w_p2p, w_p2pl) if metric != "p2p" else (0.0, 1.0)
    p.conv_tol = 0.0
    T0 = np.eye(4, dtype=np.float32)
    slab_info = None
    if strong:
        # every rank generates the whole pair (synthetic, deterministic) and keeps its slab: target points of the slab + halo,
        # source points whose T0-image falls into the slab
        d = syn.make_pair(nd, ns, with_normals=with_normals, src_stride=stride)
        part = distributed.SlabPartition.plan(d["dst"], d["src"], T0, float(d["max_sq_dist"]), world)
        dst_l, nrm_l, src_l = part.select(rank, d["dst"], d["dst_n"], d["src"])
        n_src_rank = len(src_l)
        slab_info = part.describe(rank, len(dst_l), len(src_l))
    else:
        # weak scaling / single GPU: rank r: same target, its own window of source points / noise stream
        d = syn.make_pair(nd, ns, with_normals=with_normals, src_stride=stride, src_offset=rank * 7919)
        dst_l, nrm_l, src_l = d["dst"], d["dst_n"], d["src"]
        n_src_rank = ns
    p.max_sq_dist = float(d["max_sq_dist"])
    dst_t = torch.from_numpy(dst_l).cuda()
    nrm_t = torch.from_numpy(nrm_l).cuda() if with_normals else None
    src_t = torch.from_numpy(src_l).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.set_target(dst_t, nrm_t)
    ctx.set_source(src_t)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    gi = ctx.grid_info()

    sums = torch.zeros(capi.SUMS_LEN, dtype=torch.float64, device="cuda")
    gmean = None
    if sharded:
        if strong:
            gdm, gmean = part.global_means(d["dst"], d["src"])
            ctx.set_shard_info(0, dst_mean=gdm)
            part.arm_guard(ctx, T0)
        else:
            _, sm = ctx.means()
            m = torch.tensor(sm.astype(np.float64) * n_src_rank, dtype=torch.float64, device="cuda")
            dist.all_reduce(m)
            gmean = (m.cpu().numpy() / (n_src_rank * world)).astype(np.float32)

    # the per-rank loop inside the library with its own RCCL communicator (cilhip_icp_iterate_ranked); CILHIP_BENCH_TORCH_ALLREDUCE=1 or a
    # rank without a loadable librccl: three calls + torch's all-reduce per iteration (same sums either way)
    native = sharded and os.environ.get("CILHIP_BENCH_TORCH_ALLREDUCE") != "1" and distributed.init_rank_comm(ctx, dist, None, "cuda")

    ctx.set_option("kernel_timing_stride", TIMING_STRIDE)

    def run(iters, timing):
        p.max_iter = iters
        ctx.enable_kernel_timing(timing and os.environ.get("CILHIP_BENCH_NO_KERNEL_TIMING") != "1")      # (dev: what the events themselves cost)
        if not sharded:
            return ctx.icp_run(p, T0)
        ctx.icp_begin(p, T0, gmean)
        if native:
            ctx.icp_iterate_ranked(iters)
            return ctx.icp_state()
        for _ in range(iters):
            ctx.icp_partial_sums(sums.data_ptr())
            dist.all_reduce(sums)
            ctx.icp_apply_sums(sums.data_ptr())
        return ctx.icp_state()

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    sort_ms = ctx.prepare_source(T0, force=True) if not a.no_extras else None   # (the first run below would do it lazily)
    run(a.warmup, False)
    if sharded:
        # option tie_rule in the sharded protocol driven by hand here: did some rank's warm-up searches meet exactly equidistant nearest points
        # without order tables?  (cilhip_icp_run does this by itself; the sharded loops of distributed.py too.)  One MAX over the ranks -- all take
        # the same decision --, then every rank loads the order of the WHOLE target (a slab: the entries of its points) before the timed region.
        try:
            pend = torch.tensor([1.0 if (ctx.tie_order_info()["pending"] and not ctx.tie_order_info()["loaded"]) else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(pend, op=dist.ReduceOp.MAX)
            if pend.item() > 0:
                if strong:
                    ctx.load_tie_order(distributed._tie_order_of(d["dst"]), np.ascontiguousarray(part.dst_index, np.uint32))
                else:
                    ctx.build_tie_order()
                run(a.warmup, False)
        except Exception as e:      # (a report's refinement, never a reason to lose the run)
            print(f"[bench] tie order in the sharded run skipped: {e!r}", file=sys.stderr)
    barrier()
    t0 = time.perf_counter()
    res = run(a.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    n_src_total = n_src_rank
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(n_src_rank)], dtype=torch.float64, device="cuda")
        dist.all_reduce(c)
        n_src_total = int(c.item())
        if strong:
            assert not ctx.slab_violation(), "a source point left its slab's halo: the partition had to be redone (not expected in the bench)"
    assert int(res.iterations) == a.steps, (res.iterations, a.steps)
    sf = None
    if sharded:
        # every rank's own figures of the timed region (sampled iterations: kernel_timing_stride), gathered for the line
        ftr = ctx.last_form_timing()
        k_ms = sum(ms for ms, n in ftr.values()) / max(sum(n for ms, n in ftr.values()), 1)
        ar_ms, ar_n = ctx.last_allreduce_timing() if native else (0.0, 0)
        sf = scaling_fields(torch, dist, world, "cuda", k_ms, (ar_ms * 1e3 / ar_n) if ar_n else None, ctx.last_host_enqueue_time() if native else None)

    out = None
    if rank == 0:
        T = np.array(res.T[:], np.float32).reshape(4, 4).T
        err_true = float(np.linalg.norm(T - d["T_true"]))
        loop_ms, _, launches = ctx.last_timing()
        search_ms, acc_ms = ctx.last_timing2()
        ns_l, nd_l, nc = len(src_l), len(dst_l), int(res.last_ncorr)
        one_pass_iters, two_pass_iters = ctx.last_run_forms()
        # Algorithmic (compulsory) bytes, SURVEY.md 8(d): every datum touched once.
        #   search + accumulation in ONE pass (the default: no index round trip):  12 Ns + 12 Nd + 12 Nc (the matched normals;
        #   point-to-point: the matched points are part of the 12 Nd)
        #   search alone inside the loop: 16 Ns + 12 Nd (the match index is written, the squared distance is not: nothing reads
        #   it); the streaming accumulation pass that then follows: 16 Ns + 24 Nc (point-to-point: 12 Nc)
        nc_l = nc if not sharded else ns_l       # (rank 0's own pairs; ncorr is the job's total)
        one_pass_bytes = 12.0 * ns_l + 12.0 * nd_l + (12.0 * nc_l if with_normals else 0.0)
        # warm-started iteration: of the target only the MATCHED points (and normals) are data of the computation -- the search
        # starts from the previous matches instead of reading the target (equal to the line above when Nd = Nc, as in C3)
        warm_bytes = 12.0 * ns_l + 12.0 * nc_l + (12.0 * nc_l if with_normals else 0.0)
        FORMS = {
            0: ("kNN correspondence search alone (k_search_tiled<none> + clean-up pass, or the per-lane search: small / sparse-source clouds); "
                "a streaming accumulation pass follows", 16.0 * ns_l + 12.0 * nd_l),
            1: ("k_search_tiled<metric> + k_search_deferred<metric> (LDS-tiled kNN search with the accumulation inside the tile)", one_pass_bytes),
            2: ("k_warm<metric, 1> (first warm-started iteration: search from the previous matches, gathers, writes the match records)", warm_bytes),
            3: ("k_warm<metric, 2> (warm-started iteration: search from the previous matches read as records, accumulation on the matrix cores)",
                warm_bytes),
            4: ("k_iter<metric, search> (per-lane fused search + accumulation)", one_pass_bytes),
        }
        ft = ctx.last_form_timing()
        # launches of each form in the timed region (the run's trace), and how many of them carried events (kernel_timing_stride)
        tr_forms = [int(t["form"]) & 0x7f for t in (ctx.last_run_trace() if not sharded else [])]
        n_form = {f: (tr_forms.count(f) if tr_forms else ft[f][1]) for f in ft}
        ft = stratified_form_timing(ft, tr_forms, ctx.last_iteration_timing() if not sharded else [])
        forms = {str(f): {"launches": n_form[f], "timed_launches": n, "avg_kernel_ms": ms / n} for f, (ms, n) in ft.items() if n > 0}
        # the form the timed region spent most kernel time in (average of the timed launches x all launches of the form)
        dom = max((f for f in ft if ft[f][1] > 0), key=lambda f: ft[f][0] / ft[f][1] * n_form[f], default=None) if launches > 0 else None
        fused = dom is not None and dom != 0
        traffic, traffic_note, traffic_cold = None, "no PMC measurement of this build / workload committed", None
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_c*.json")), reverse=True):      # newest round first
            try:
                tj = json.load(open(tf))
                w = tj["workload"]
                # (a file without a positive figure -- its passes did not find the kernel -- is not a measurement)
                if tj.get("source_hash") == source_hash() and (w["n_target"], w["n_source_per_gpu"], w["metric"]) == (nd_l, ns_l, metric) and tj.get("form") == dom \
                        and float(tj.get("traffic_bytes_per_launch") or 0.0) > 0.0:
                    traffic, traffic_note = float(tj["traffic_bytes_per_launch"]), os.path.basename(tf) + ": " + tj.get("method", "")
                    traffic_cold = tj.get("cold_forms")      # {"plain_tile": bytes per launch, "record_writing_tile": ...} of the same passes
                    break
                if traffic_note.startswith("no PMC"):      # (name the newest file only)
                    traffic_note = "profiles/" + os.path.basename(tf) + " was measured on another build, workload or kernel form: not quoted"
            except Exception:
                pass
        roof = None
        if dom is not None:      # (sharded runs: rank 0's own kernels)
            kern, alg_bytes = FORMS[dom]
            avg_ms = ft[dom][0] / ft[dom][1]
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "measured_copy_bandwidth_GBps": copy_bandwidth(torch), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": kern, "form": dom, "avg_kernel_ms": avg_ms, "launches": n_form[dom], "timed_launches": ft[dom][1], "algorithmic_bytes_per_launch": alg_bytes,
                    "timing": "hipEvents attached to the kernels' own dispatch packets on the context's stream, inside the timed region: iterations 0-2 and "
                              f"every {TIMING_STRIDE}th one (an event between two dependent kernels idles the device for ~6 us, two per iteration are a tenth of a "
                              "warm-started iteration: all of them timed costs the run 9 %); the dominant kernel is the form with the largest share of the "
                              "timed region's kernel time; a form's average = per stratum (first iteration of a stretch of the form / the rest) the mean of the timed "
                              "launches, weighted by the stratum's share of the run's iterations",
                    "forms_in_timed_region": forms,
                    # HBM bytes actually moved / time / peak (the counter traffic of the same build: wasted re-reads count as work here)
                    "frac_moved_bytes": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    # every search / one-pass kernel of the timed region, launch-weighted: sum over the forms of (average launch x launches) / steps
                    # (not in it: the reduction and epilogue kernels of an iteration, ~15 us, and the gaps between dependent launches)
                    "kernel_ms_per_step": sum(ms / n * n_form[f] for f, (ms, n) in ft.items() if n > 0) / max(a.steps, 1),
                    "timed_launches_unweighted_avg_kernel_ms": search_ms / launches}
            if 1 in ft and ft[1][1] > 0 and dom != 1:
                # the COLD form beside the headline: the LDS-tiled search with the accumulation inside the tile -- what the first
                # iterations of every registration, and every iteration of a source that is not the target's points plus small
                # noise (`independent_source` below), run in
                cold_ms = ft[1][0] / ft[1][1]
                out_cold = {"bound": "hbm", "kernel": FORMS[1][0], "form": 1, "avg_kernel_ms": cold_ms, "launches": n_form[1],
                            "algorithmic_bytes_per_launch": one_pass_bytes, "achieved": one_pass_bytes / (cold_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": one_pass_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                # the two tile kernels behind that average, by the iteration they ran in: iteration 0 is the plain tile, the later ones also
                # write the warm-started form's match records (k_search_tiled<metric, false, true>)
                it_ms = dict(ctx.last_iteration_timing()) if not sharded else {}
                for name, its in (("plain_tile", [0]), ("record_writing_tile", [i for i in range(1, len(tr_forms)) if tr_forms[i] == 1])):
                    ms_l = [it_ms[i] for i in its if i in it_ms and i < len(tr_forms) and tr_forms[i] == 1]
                    if ms_l:
                        m = sum(ms_l) / len(ms_l)
                        tb = (traffic_cold or {}).get(name)
                        out_cold[name] = {"avg_kernel_ms": m, "timed_launches": len(ms_l), "frac": one_pass_bytes / (m * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "traffic": tb, "frac_moved_bytes": (tb / (m * 1e-3) / 1e9 / HBM_PEAK_GBS) if tb else None}
            else:
                out_cold = None
            if dom == 0:
                acc_bytes = 16.0 * ns_l + (24.0 if with_normals else 12.0) * nc_l
                roof["accumulate_kernel"] = {"avg_kernel_ms": acc_ms / launches, "algorithmic_bytes_per_launch": acc_bytes,
                                             "achieved_GBps": acc_bytes / max(acc_ms / launches * 1e-3, 1e-12) / 1e9}
        sharding = "none"
        if sharded:
            sharding = ("spatial slabs: per rank the target points of its slab + halo and the source points inside the slab; all-reduce(sum) of 48 f64 per iteration"
                        if strong else "source-sharded, target replicated, all-reduce(sum) of 48 f64 per iteration")
        out = {
            "metric": "ICP corr. pairs/sec (+ iterations/sec), synthetic uniform clouds",
            "value": n_src_total * a.steps / dt, "unit": "pairs/s",
            "icp_iterations_per_sec": a.steps / dt,
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32 search / f64 accumulate+solve", "data": "synthetic",
            "config": {"workload": f"{a.config}: {ns/1e6:g}M<->{nd/1e6:g}M synthetic float3 clouds " + label,
                       "n_target": nd, "n_source_per_gpu": ns_l, "n_source_total": n_src_total, "max_sq_dist": float(d["max_sq_dist"]),
                       "iterations": a.steps, "conv_tol": 0.0, "sharding": sharding, "slab": slab_info,
                       "allreduce": ("library loop, ncclAllReduce on the context's stream (cilhip_icp_iterate_ranked)" if native else "torch.distributed.all_reduce per iteration") if sharded else None,
                       "grid": [gi.nx, gi.ny, gi.nz], "grid_cell": gi.cell, "grid_avg_occupancy": gi.avg_occupancy},
            "setup_ms": t_setup * 1e3, "source_sort_ms": sort_ms, "loop_ms_hip_events": loop_ms,
            "last_ncorr": nc, "T_err_vs_truth_frobenius": err_true,
            "iterations_one_pass": one_pass_iters, "iterations_two_pass": two_pass_iters, "iterations_warm_started": ctx.last_warm_iterations(),
            # option "tie_rule" (default 2): exactly equidistant nearest points take the reference's kd-tree order, resolved on the device; the order
            # tables are built on the device the first time a search of this target meets a tie (here: in the warm-up run, if at all)
            "tie_order": dict(ctx.tie_order_info(), tied_queries_resolved_in_timed_run=ctx.tie_rule_stats()[0], not_the_lowest_index=ctx.tie_rule_stats()[1]),
            "roofline": roof, "roofline_cold": out_cold if dom is not None else None,
        }
        if sf is not None:
            out.update(sf)
    if not sharded and not a.no_extras:
        extras = {}
        # cost of a whole estimate() the way the reference's example calls it (15 iterations), sort included: what a caller sees
        cold = []
        for _ in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.prepare_source(T0, force=True)
            p.max_iter = 15
            ctx.enable_kernel_timing(False)
            ctx.icp_run(p, T0)
            ctx.synchronize()
            cold.append((time.perf_counter() - t0) * 1e3)
        extras["icp_estimate_ms_15iter_cold"] = statistics.median(cold)
        # the same timed region with the warm-started form switched off (every iteration through the LDS-tiled kernels): the
        # matches -- and so the loop -- are the same, only the way they are found differs
        ctx.set_option("warm_start", 0)
        ctx.prepare_source(T0, force=True)
        p.max_iter = a.warmup; ctx.icp_run(p, T0)
        ctx.synchronize(); t0 = time.perf_counter()
        p.max_iter = a.steps; r0 = ctx.icp_run(p, T0)
        ctx.synchronize(); dt0 = time.perf_counter() - t0
        T_nw = np.array(r0.T[:], np.float32).reshape(4, 4).T
        extras["without_warm_start"] = {"ms_per_step": dt0 * 1e3 / a.steps, "icp_iterations_per_sec": a.steps / dt0, "last_ncorr": int(r0.last_ncorr),
                                        "max_abs_T_difference_to_the_timed_run": float(np.abs(T_nw - np.array(res.T[:], np.float32).reshape(4, 4).T).max())}
        ctx.set_option("warm_start", 1)
        extras["source_sort_ms_runs"] = [ctx.prepare_source(T0, force=True) for _ in range(3)]
        # a converging trajectory (SURVEY.md 8(d) "recipe sanity": perturbation 0.8 h, tolerance 1e-5) beside the fixed-point run
        try:
            dc = syn.make_pair(nd, ns, with_normals=False, src_stride=stride, perturb=0.8)
            ctx.set_source(torch.from_numpy(dc["src"]).cuda())
            p.max_iter, p.conv_tol = 50, 1e-5
            ctx.icp_run(p, T0)                       # (first call: sort + run, warm)
            ctx.synchronize(); t0 = time.perf_counter()
            ctx.prepare_source(T0, force=True)
            rc = ctx.icp_run(p, T0)
            ctx.synchronize(); tc = (time.perf_counter() - t0) * 1e3
            Tc = np.array(rc.T[:], np.float32).reshape(4, 4).T
            extras["converging_run"] = {"perturbation_h": 0.8, "conv_tol": 1e-5, "iterations": int(rc.iterations), "ms_total_incl_sort": tc,
                                        "iterations_one_pass_two_pass": list(ctx.last_run_forms()),
                                        "ms_per_iteration": tc / max(int(rc.iterations), 1), "T_err_vs_truth_frobenius": float(np.linalg.norm(Tc - dc["T_true"]))}
            p.conv_tol = 0.0
        except Exception as e:
            extras["converging_run"] = {"error": repr(e)}
        # A source that is NOT the target's points plus small noise: an independent uniform sample of the same volume (matches at
        # about half the point spacing): the regime every registration of two separately sampled clouds lives in.  The margin
        # proof settles 96 % of its queries per iteration; the rest -- nearly equidistant first and second
        # neighbours -- is searched again in every iteration.
        try:
            rng = np.random.default_rng(3)
            si = rng.random((ns, 3), dtype=np.float32)
            Ti = np.linalg.inv(d["T_true"].astype(np.float64))
            si = (si.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
            ctx.set_source(torch.from_numpy(si).cuda())
            p.conv_tol = 0.0
            p.max_iter = a.steps; ctx.enable_kernel_timing(False); ctx.icp_run(p, T0)       # (sort + warm-up run)
            ctx.enable_kernel_timing(True)
            ctx.synchronize(); t0 = time.perf_counter()
            ri = ctx.icp_run(p, T0)
            ctx.synchronize(); dti = time.perf_counter() - t0
            tri = [int(t["form"]) & 0x7f for t in ctx.last_run_trace()]
            fti = stratified_form_timing(ctx.last_form_timing(), tri, ctx.last_iteration_timing())
            nci = int(ri.last_ncorr)
            domi = max((f for f in fti if fti[f][1] > 0), key=lambda f: fti[f][0] / fti[f][1] * max(tri.count(f), 1))
            bytes_i = {0: 16.0 * ns + 12.0 * nd, 1: 12.0 * ns + 12.0 * nd + (12.0 * nci if with_normals else 0.0)}.get(domi, 12.0 * ns + (24.0 if with_normals else 12.0) * nci)
            ms_i = fti[domi][0] / max(fti[domi][1], 1)
            extras["independent_source"] = {
                "workload": f"{ns/1e6:g}M independent uniform source points against the same {nd/1e6:g}M-point target, {a.steps} iterations, tolerance 0",
                "ms_per_step": dti * 1e3 / a.steps, "icp_iterations_per_sec": a.steps / dti, "last_ncorr": nci,
                "iterations_one_pass_two_pass": list(ctx.last_run_forms()), "iterations_warm_started": ctx.last_warm_iterations(),
                "forms": {str(f): {"launches": tri.count(f), "timed_launches": n, "avg_kernel_ms": ms / n} for f, (ms, n) in fti.items() if n > 0},
                "roofline": {"bound": "hbm", "form": domi, "avg_kernel_ms": ms_i, "algorithmic_bytes_per_launch": bytes_i,
                             "achieved": bytes_i / (ms_i * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_i / (ms_i * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        except Exception as e:
            extras["independent_source"] = {"error": repr(e)}
        # The other ICP instance families / engine directions on the SAME clouds (driver-timed companions of `ms_per_step`): the affine classes
        # (device-resident loop: one-pass moments on the matrix cores, 12-unknown solve in the epilogue kernel) and the FIRST_TO_SECOND / BOTH
        # search directions (reverse searches warm-started from the previous reverse matches), a.steps iterations each, tolerance 0.
        try:
            import copy
            variants = {}
            for vname, vopts, vmetric in (("affine_combined", {"transform_mode": 1}, capi.METRIC_COMBINED),
                                          ("affine_point_to_point", {"transform_mode": 1}, capi.METRIC_POINT_TO_POINT),
                                          ("first_to_second", {"search_direction": 1}, capi.METRIC_COMBINED),
                                          ("both", {"search_direction": 2}, capi.METRIC_COMBINED),
                                          ("both_reciprocal", {"search_direction": 2, "require_reciprocality": 1}, capi.METRIC_COMBINED),
                                          # PointNormalFeaturesAdaptor on both clouds (6-D search, normal weight h / 2, the source's normals = its twin's), three-cloud metric
                                          # the four-cloud constructor: the symmetric point-to-plane objective (source normals = the twins')
                                          ("symmetric_metric", {"symmetric_metric": 1}, capi.METRIC_COMBINED),
                                          ("features_point_normal", {"feature_kind": 0, "feature_normal_weight": 0.5 * float(d["h"]), "symmetric_metric": 0}, capi.METRIC_COMBINED)):
                if not with_normals and vmetric == capi.METRIC_COMBINED:
                    continue
                if vname in ("features_point_normal", "symmetric_metric") and ns != nd:
                    continue
                cv = Context(local_rank, stream)
                cv.set_target(dst_t, nrm_t); cv.set_source(src_t)
                if vname in ("features_point_normal", "symmetric_metric"):
                    from cilantro_amd.icp import _as_cloud
                    qn, _, memn, _keep = _as_cloud(nrm_t)
                    cv._ck(cv._L.cilhip_set_source_normals(cv._h, qn, memn))
                for k, v in vopts.items():
                    cv.set_option(k, v)
                pv = copy.copy(p)
                pv.metric = vmetric
                pv.w_p2p, pv.w_p2pl = (w_p2p, w_p2pl) if with_normals else (0.0, 1.0)
                pv.conv_tol = 0.0; pv.max_iter = a.steps
                cv.icp_run(pv, T0)                   # (sort, grids / tables of the variant, warm-up)
                cv.synchronize(); t0 = time.perf_counter()
                rv = cv.icp_run(pv, T0)
                cv.synchronize(); dtv = time.perf_counter() - t0
                lv = cv.last_timing()[0]
                Tv = np.array(rv.T[:], dtype=np.float64).reshape(4, 4).T
                variants[vname] = {"ms_per_step": dtv * 1e3 / a.steps, "loop_ms_per_step_hip_events": lv / a.steps, "iterations": int(rv.iterations),
                                   "last_ncorr": int(rv.last_ncorr), "iterations_warm_started": cv.last_warm_iterations(),
                                   "T_err_vs_truth_frobenius": float(np.linalg.norm(Tv - d["T_true"]))}
                cv.close()
            extras["variants"] = variants
        except Exception as e:
            extras["variants"] = {"error": repr(e)}
        # the sharded loop driven from C (cilhip_multi_icp_run: what a C / C++ caller with several devices uses) on ONE shard: the cost of
        # the protocol itself -- three calls per iteration instead of one enqueue-ahead loop -- against `ms_per_step` above
        try:
            from cilantro_amd.multi import PARTITION_SLABS, MultiDeviceRigidICP
            ctx.close()
            mm = MultiDeviceRigidICP([local_rank])
            mm.set_clouds(d["dst"], d["dst_n"] if with_normals else None, d["src"], float(d["max_sq_dist"]), PARTITION_SLABS)
            p.conv_tol = 0.0
            p.max_iter = a.warmup; mm.icp_run(p, T0, check_every=1 << 20)
            p.max_iter = a.steps; mm.icp_run(p, T0, check_every=1 << 20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rm = mm.icp_run(p, T0, check_every=1 << 20)
            torch.cuda.synchronize(); dtm = time.perf_counter() - t0
            Tm = np.array(rm.T[:], np.float32).reshape(4, 4).T
            extras["multi_device_c_loop_one_shard"] = {
                "entry": "cilhip_multi_icp_run, devices = [0], spatial slabs (one slab), state read once at the end",
                "ms_per_step": dtm * 1e3 / a.steps, "icp_iterations_per_sec": a.steps / dtm,
                "relative_to_cilhip_icp_run": (a.steps / dtm) / (a.steps / dt),
                "max_abs_T_difference_to_the_timed_run": float(np.abs(Tm - np.array(res.T[:], np.float32).reshape(4, 4).T).max())}
            mm.close()
            # ... and EIGHT shards on this one GPU (repeated ordinal: the all-reduce runs as the same-device kernel): what one GPU can show
            # of an 8-device run's HOST side -- the enqueue calls of an iteration per shard, with one host thread per shard (the default)
            # and with one thread walking the shards (CILHIP_MULTI_THREADS=0, round 4); the shards' kernels share the one device, so
            # ms_per_step here is not a scaling figure
            leg = {}
            for label, thr in (("one_host_thread_per_shard", "1"), ("one_host_thread", "0")):
                os.environ["CILHIP_MULTI_THREADS"] = thr
                m8 = MultiDeviceRigidICP([local_rank] * 8)
                m8.set_clouds(d["dst"], d["dst_n"] if with_normals else None, d["src"], float(d["max_sq_dist"]), PARTITION_SLABS)
                p.max_iter = a.warmup; m8.icp_run(p, T0, check_every=1 << 20)
                p.max_iter = a.steps; m8.icp_run(p, T0, check_every=1 << 20)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r8 = m8.icp_run(p, T0, check_every=1 << 20)
                torch.cuda.synchronize(); dt8 = time.perf_counter() - t0
                leg[label] = {"ms_per_step": dt8 * 1e3 / a.steps, "host_enqueue_us_per_iteration_per_shard": m8.last_host_time(),
                              "max_abs_T_difference_to_the_timed_run": float(np.abs(np.array(r8.T[:], np.float32).reshape(4, 4).T - np.array(res.T[:], np.float32).reshape(4, 4).T).max())}
                m8.close()
            os.environ.pop("CILHIP_MULTI_THREADS", None)
            extras["multi_device_c_loop_8_shards_one_gpu"] = dict(leg, entry="cilhip_multi_icp_run, devices = [0] x 8, spatial slabs, state read once at the end")
            # partitioning A from C: the TARGET in two index shards on this one GPU, the whole source on both (every query is searched twice,
            # a MIN of one 64-bit key per query between the shards, each shard accumulates what it won): the protocol's cost, not a scaling figure
            from cilantro_amd.multi import PARTITION_TARGET_SHARDS
            m2 = MultiDeviceRigidICP([local_rank] * 2)
            m2.set_clouds(d["dst"], d["dst_n"] if with_normals else None, d["src"], float(d["max_sq_dist"]), PARTITION_TARGET_SHARDS)
            p.max_iter = a.warmup; m2.icp_run(p, T0, check_every=1 << 20)
            p.max_iter = a.steps; m2.icp_run(p, T0, check_every=1 << 20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r2 = m2.icp_run(p, T0, check_every=1 << 20)
            torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
            extras["multi_device_c_loop_2_target_shards_one_gpu"] = {
                "entry": "cilhip_multi_icp_run, devices = [0] x 2, partition 2 (index shards of the target, MIN of packed keys per iteration), state read once at the end",
                "ms_per_step": dt2 * 1e3 / a.steps,
                "max_abs_T_difference_to_the_timed_run": float(np.abs(np.array(r2.T[:], np.float32).reshape(4, 4).T - np.array(res.T[:], np.float32).reshape(4, 4).T).max())}
            m2.close()
        except Exception as e:
            extras["multi_device_c_loop_one_shard"] = {"error": repr(e)}
        out.update(extras)
    if rank == 0 and not sharded and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline_icp(d, metric, w_p2p, w_p2pl, min(a.cpu_sample, ns), T0)
        except Exception as e:  # the baseline is a report, never the product path
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0 and emit and not sharded and a.config == "c3" and not a.no_extras and not a.no_other_configs and a.n is None:
        out["other_configs"] = other_configs(a, torch, local_rank)
    if rank == 0 and emit:
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()
    if not emit:
        ctx.close()      # (a figure for another line: its context and clouds go before the next configuration is set up)
    return out


def other_configs(a, torch, local_rank, budget_s=150.0):
    """The default line is the one the driver times; the other BASELINE configurations ride on it as `other_configs` -- ms_per_step and
    roofline.frac each, measured by the very functions their own `--config` lines come from (`value`, `config`, `roofline` of the line
    itself stay C3's).  A wall-clock budget keeps the default run inside a few minutes: what does not fit is named, not dropped silently."""
    import copy
    import gc

    res, t_begin = {}, time.perf_counter()
    for cfg in ("c2", "kmeans", "ransac", "c4_1gpu"):
        spent = time.perf_counter() - t_begin
        if spent > budget_s:
            res[cfg] = {"skipped": f"wall-clock budget of {budget_s:.0f} s spent ({spent:.0f} s)"}
            continue
        b = copy.copy(a)
        b.config, b.no_extras, b.no_cpu_baseline, b.n, b.metric = cfg, True, True, None, None
        t0 = time.perf_counter()
        try:
            if cfg == "kmeans":
                o = bench_kmeans(b, torch, emit=False)
            elif cfg == "ransac":
                o = bench_ransac(b, torch, emit=False)
            else:
                o = bench_icp(b, torch, 0, 1, local_rank, emit=False)
            res[cfg] = {"workload": o["config"]["workload"], "metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                        "roofline": {k: o["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_kernel_ms", "traffic")},
                        "wall_s": time.perf_counter() - t0}
        except Exception as e:      # (a report's refinement, never a reason to lose the line)
            res[cfg] = {"error": repr(e)}
        gc.collect()
        torch.cuda.empty_cache()
    return res


def bench_kmeans_sharded(a, torch, rank, world, local_rank):
    """--config kmeans --gpus N: the points sharded over the ranks (50M per rank: weak scaling), centroids replicated, ONE all-reduce of
    4k + 1 int64 per Lloyd iteration over RCCL (cilantro_amd/distributed_models.py: ShardedKMeans3f; SURVEY.md 8(e), last row).
    Results are the single-device run's bit for bit (exact integer sums); value = point-centroid distances per second over all ranks."""
    import torch.distributed as dist

    from cilantro_amd import distributed_models as dm, synthetic as syn

    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n, k = (a.n or 50_000_000), 1024
    x = syn.make_dst(n, offset=rank * n)       # (counter-based generator: the ranks' shards are consecutive pieces of ONE cloud of world * n points)
    c0 = syn.make_dst(k).copy()                # ... whose first k points are the initial centroids, on every rank
    xd = torch.from_numpy(x).cuda()
    eng = dm.HipKMeansShard(xd, k, rank * n, local_rank)
    km = dm.ShardedKMeans3f(eng, dist, device="cuda")
    km.cluster(c0, max_iter=max(a.warmup, 1), tol=0.0, fetch_labels=False)

    def timed(iters):
        dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        km.cluster(c0, max_iter=iters, tol=0.0, fetch_labels=False)
        torch.cuda.synchronize(); dist.barrier()
        return time.perf_counter() - t0

    t1 = timed(1)                              # fixed costs (scale, labels) cancel in the difference
    tk = timed(a.steps + 1)
    assert km.getNumberOfPerformedIterations() == a.steps + 1
    t = torch.tensor([tk - t1], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        evals = float(n) * world * k * a.steps
        print(json.dumps({"metric": "KMeans3f point-centroid distance evaluations/sec (k = 1024), brute-force equivalent", "value": evals / dt, "unit": "distances/s",
                          "n_gpus": world, "rccl_ranks": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32 distances / exact fixed-point sums", "data": "synthetic",
                          "config": {"workload": f"kmeans: KMeans3f k = {k}, {n/1e6:g}M uniform points PER RANK, initial centroids replicated, tol = 0", "n_points_per_gpu": n, "k": k,
                                     "sharding": "points; centroids replicated", "allreduce": f"{4 * k + 1} int64 (sum) per Lloyd iteration over RCCL"},
                          "roofline": {"bound": "hbm", "achieved": 20.0 * n * world * a.steps / dt / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                       "frac": 20.0 * n * a.steps / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "k_assign_grid",
                                       "note": "per rank as on one device (see --gpus 1: the pruned pass is VALU-issue-bound); the exchange is 32 KB per iteration"}}))
    eng.close()
    dist.destroy_process_group()


def bench_kmeans(a, torch, emit=True):
    """BASELINE configs[4], first half: KMeans3f k = 1024 on 50M points, explicit initial centroids (the first k points).
    Step = one Lloyd iteration (brute-force assignment + centroid update, one pass over the points)."""
    from cilantro_amd import synthetic as syn
    from cilantro_amd.clustering import KMeans3f

    n, k = (a.n or 50_000_000), 1024
    x = syn.make_dst(n)
    xd = torch.from_numpy(x).cuda()
    c0 = x[:k].copy()
    KMeans3f(xd).cluster(c0, max_iter=max(a.warmup, 1), tol=0.0)

    def timed(iters, kd=False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        km = KMeans3f(xd).cluster(c0, max_iter=iters, tol=0.0, use_kd_tree=kd)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, km

    def per_step(kd=False):
        t1, _ = timed(1, kd)                  # fixed costs (centroid upload, label download) cancel in the difference
        tk, km = timed(a.steps + 1, kd)
        assert km.getNumberOfPerformedIterations() == a.steps + 1
        return tk - t1

    from cilantro_amd import clustering
    dt = per_step()                           # the default: the assignment pruned exactly (centroid grid, proof, fallbacks: kmeans.hip)
    clustering.set_pruning(False)
    try:
        KMeans3f(xd).cluster(c0, max_iter=1, tol=0.0)
        dt_ex = per_step()                    # the exhaustive pass of the same library (n * k distances): what round 4's line measured
    finally:
        clustering.set_pruning(True)
    # ... and the reference's use_kd_tree branch (the mode its examples/kmeans.cpp runs): the same pruned pass with nanoflann's rounding, the
    # order tables of the tree over the centroids built in the iterations that meet exactly equidistant centroids
    dt_kd = None
    if emit:
        try:
            KMeans3f(xd).cluster(c0, max_iter=1, tol=0.0, use_kd_tree=True)
            dt_kd = per_step(True)
        except Exception:
            dt_kd = None
    evals = float(n) * k * a.steps
    flops_ex = 8.0 * evals                    # 3 sub, 3 mul, 2 add per point-centroid distance, each individually rounded
    read_evals = 36.0 * float(n) * a.steps    # distances the pruned pass evaluates at least: four records of each of the nine runs of a point's block
    bytes_step = 20.0 * float(n)              # 12 B point + 4 B label read + 4 B label written
    km_traffic, km_note = config_traffic("kmeans", {"n_points": n, "k": k})
    out = {"metric": "KMeans3f point-centroid distance evaluations/sec (k = 1024), brute-force equivalent", "value": evals / dt, "unit": "distances/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 distances / exact fixed-point sums", "data": "synthetic",
           "config": {"workload": f"kmeans: KMeans3f k = {k} on {n/1e6:g}M uniform points, initial centroids = the first k points, tol = 0", "n_points": n, "k": k,
                      "assignment": "pruned exactly: labels bit-identical to the exhaustive argmin (tests/test_gpu_parity.py::test_kmeans_pruned_assignment_is_the_exhaustive_one)"},
           # HBM and the arithmetic ceiling are both far away (fractions below); what binds the pruned pass is VALU ISSUE: roofline.limiter
           "roofline": {"bound": "hbm", "achieved": bytes_step * a.steps / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_step * a.steps / dt / 1e9 / HBM_PEAK_GBS,
                        "traffic": km_traffic, "traffic_note": km_note, "kernel": "k_assign_grid", "algorithmic_bytes_per_launch": bytes_step,
                        "valu_frac_of_evaluated_distances": 8.0 * read_evals / dt / 1e12 / VALU_NOFMA_PEAK_TFLOPS,
                        "note": "12 B point + label read + label written per point against 8 TB/s; the distances it does evaluate (>= 36 per point, 8 individually rounded f32 ops each) "
                                "against the 78.6 TFLOP/s non-FMA vector ceiling: valu_frac_of_evaluated_distances -- neither binds; `limiter` does",
                        # counters of this kernel on this workload (profiles/r06_config_c5_pmc_summary.txt: SQ_INSTS_VALU of a steady-state launch) against the measured
                        # issue rate of the instructions it is made of (profiles/r06_valu_rate_probe.txt: 2.0 ns per wave-instruction and SIMD, 1024 SIMDs)
                        "limiter": ({"bound": "valu issue", "wave_instructions_per_launch": 6.15e8, "ns_per_wave_instruction_per_simd": 2.0, "simds": 1024,
                                     "frac": 6.15e8 * 2.0e-9 / 1024.0 / (dt / a.steps),
                                     "lds_array_busy": "about 0.7 (SQ_LDS_IDX_ACTIVE 5.5e8 cycles per launch over 256 CUs; 62 % of them bank conflicts: the lanes of a wave sit in different centroid cells)",
                                     "source": "profiles/r06_config_c5_pmc_summary.txt, profiles/r06_valu_rate_probe.txt (counted on n = 50M, k = 1024)"}
                                    if (n == 50_000_000 and k == 1024) else None)},
           "kd_branch": ({"ms_per_step": dt_kd * 1e3 / a.steps, "note": "use_kd_tree = true (clustering/kmeans.hpp:86-94): labels equal the reference's tree search, exact ties included"}
                         if dt_kd is not None else None),
           "exhaustive_pass": {"ms_per_step": dt_ex * 1e3 / a.steps, "distances_per_sec": evals / dt_ex, "kernel": "k_assign_accumulate",
                               "roofline": {"bound": "valu", "achieved": flops_ex / dt_ex / 1e12, "peak": VALU_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops_ex / dt_ex / 1e12 / VALU_NOFMA_PEAK_TFLOPS,
                                            "note": "8 individually rounded f32 ops per distance (no FMA contraction: labels must match the reference bit for bit); peak = half of the 157.3 TFLOP/s FMA figure"}}}
    if not a.no_cpu_baseline:
        try:
            from oracle import oracle as orc

            m = min(n, 4_000_000)
            ts = []
            for r in range(4):
                t0 = time.perf_counter(); orc.kmeans_assign(x[:m], c0); t = time.perf_counter() - t0
                if r:
                    ts.append(t)
            cores = os.cpu_count() or 1
            out["cpu_baseline"] = {"value": m * k / statistics.median(ts), "unit": "distances/s", "cores": cores, "kind": "port",
                                   "sample": f"assignment of {m} of {n} points to the {k} centroids, oracle (C restatement of clustering/kmeans.hpp:95-119, "
                                             f"OpenMP parallel for over the points as the reference's :100, {cores} threads), median of 3 after a warm-up"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    if emit:
        print(json.dumps(out))
    return out


def bench_ransac(a, torch, emit=True):
    """BASELINE configs[4], second half: plane RANSAC inlier counting on 50M points.  Step = one scoring pass of 128
    hypotheses over all points (the estimator scores its hypotheses 128 at a time)."""
    from cilantro_amd.model_estimation import PlaneRANSACEstimator3f

    n = a.n or 50_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
    k = int(0.6 * n)
    x[:k, 2] = 0.3 * x[:k, 0] - 0.2 * x[:k, 1] + 0.1 + 0.004 * torch.randn(k, device="cuda", generator=g)
    rng = np.random.default_rng(3)
    nrm = rng.normal(size=(128 * a.steps, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    planes = np.concatenate([nrm, rng.uniform(-0.5, 0.5, (len(nrm), 1))], axis=1).astype(np.float32)
    pe = PlaneRANSACEstimator3f(x).setMaxInlierResidual(0.01)
    pe.countInliers(planes[:128 * max(a.warmup, 1)])
    # (each call pays a fixed cost besides its passes -- stream, buffers, upload of the hypotheses, download of the counts: the time of
    #  a.steps passes is the difference between a call with a.steps + 1 passes and a call with one; the MINIMUM of three of each, one
    #  host hiccup in either would otherwise land in the difference)
    t1s, tks = [], []
    more = np.concatenate([planes, planes[:128]])
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cnt1 = pe.countInliers(planes[:128])
        torch.cuda.synchronize(); t1s.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cnt = pe.countInliers(more)
        torch.cuda.synchronize(); tks.append(time.perf_counter() - t0)
    t1, tk = min(t1s), min(tks)
    dt = tk - t1                                  # a.steps passes of 128 hypotheses
    tests = float(n) * 128 * a.steps
    alg = 12.0 * n                                # one read of the points per 128-hypothesis pass
    rs_traffic, rs_note = config_traffic("ransac", {"n_points": n})
    out = {"metric": "plane RANSAC point-plane tests/sec", "value": tests / dt, "unit": "tests/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"ransac: PlaneRANSACEstimator3f inlier counting, {n/1e6:g}M points, 128 hypotheses per pass", "n_points": n},
           "roofline": {"bound": "valu", "achieved": 6.0 * tests / dt / 1e12, "peak": VALU_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": 6.0 * tests / dt / 1e12 / VALU_NOFMA_PEAK_TFLOPS, "traffic": rs_traffic, "traffic_note": rs_note, "kernel": "k_score",
                        "note": "6 individually rounded f32 operations per point-plane test (3 multiplies + 3 additions of n.p + offset, hyperplane.hpp absDistance; the "
                                "|.| <= threshold compare and the ballot/popcount are not counted), no FMA contraction: counts must match the reference bit for bit; "
                                "peak = half of the 157.3 TFLOP/s FMA figure.  128 hypotheses share one read of the points, so HBM is not the bound:",
                        "hbm_GBps": alg / (dt / a.steps) / 1e9, "hbm_frac": alg / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
                        # what the instruction stream itself allows: per PAIR of tests of a lane six packed f32 operations (v_pk_mul / v_pk_add: two tests
                        # each) and two compares (|r| <= thr, one test each; the counts go through ballots on the scalar unit), at the issue rates
                        # tools/valu_rate_probe measured on this part (profiles/r06_valu_rate_probe.txt: 2.0 ns per wave-instruction and SIMD for
                        # v_pk_*, 1.45 ns for single-operation f32 instructions), 1024 SIMDs
                        "issue_bound": {"ns_per_128_lane_tests": 6 * 2.0 + 2 * 1.45, "bound_ms_per_step": float(n) * 128 / 128.0 * (6 * 2.0 + 2 * 1.45) * 1e-6 / 1024.0,
                                        "frac": float(n) * 128 / 128.0 * (6 * 2.0 + 2 * 1.45) * 1e-6 / 1024.0 / (dt * 1e3 / a.steps)}}}
    if not a.no_cpu_baseline:
        try:
            from oracle import oracle as orc

            m = min(n, 20_000_000)
            xs = np.ascontiguousarray(x[:m].cpu().numpy())
            ts, ts1 = [], []
            for r in range(4):
                t0 = time.perf_counter()
                for j in range(4):
                    orc.plane_count_inliers_mt(xs, planes[j], 0.01)
                t = time.perf_counter() - t0
                t0 = time.perf_counter()
                orc.plane_count_inliers(xs[: m // 10], planes[0], 0.01)
                t1 = time.perf_counter() - t0
                if r:
                    ts.append(t); ts1.append(t1)
            cores = os.cpu_count() or 1
            out["cpu_baseline"] = {"value": 4 * m / statistics.median(ts), "unit": "tests/s", "cores": cores, "kind": "port",
                                   "sample": f"4 hypotheses x {m} of {n} points, oracle (C restatement of ransac_hyperplane_estimator.hpp:47-55) with an OpenMP "
                                             f"parallel for over the points on {cores} threads, median of 3 after a warm-up; the reference itself evaluates the "
                                             f"residuals as a SERIAL Eigen expression: `serial_value` is the same loop on 1 thread",
                                   "serial_value": (m // 10) / statistics.median(ts1)}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    assert int(cnt[0]) == int(cnt1[0])
    if emit:
        print(json.dumps(out))
    return out


def scaling_fields(torch, dist, world, device, kernel_ms_per_step, allreduce_us, enqueue_us=None):
    """What a scaling curve is read against, on the line of an N-rank run: how many ranks the collective spans, every rank's own kernel
    time per iteration and the time its all-reduce takes on the stream (hipEvents around the collective).  One all-gather over the
    launcher's process group (RCCL on the GPUs; the CPU test of the launch path plays it over gloo)."""
    mine = torch.tensor([float(kernel_ms_per_step), float(allreduce_us) if allreduce_us is not None else -1.0,
                         float(enqueue_us) if enqueue_us is not None else -1.0], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = [p.cpu().tolist() for p in parts]
    ar = [r[1] for r in rows if r[1] >= 0.0]
    enq = [r[2] for r in rows if r[2] >= 0.0]
    return {"rccl_ranks": world, "kernel_ms_per_step_per_rank": [r[0] for r in rows],
            "allreduce_us_per_iteration": (max(ar) if ar else None), "allreduce_us_per_iteration_per_rank": ar or None,
            "host_enqueue_us_per_iteration_per_rank": enq or None}


def relaunch_under_torchrun(a):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1).  Never prints a 1-GPU line for an N-GPU request: fewer visible devices than ranks is an error."""
    import socket
    import subprocess

    if not a.selftest_spawn:
        import torch

        nvis = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if nvis < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {nvis} HIP device(s) visible on this node -- refusing to report a {a.gpus}-GPU number")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def selftest_spawn(a, rank, world):
    """CPU-only check of the launch path (tests/test_bench_launch.py): the ranks rendezvous over gloo and all-reduce once."""
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo")
    ones = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(ones)
    # the per-rank fields of an N-rank line, gathered exactly as bench_icp gathers them (stand-in figures: rank r reports r + 1)
    sf = scaling_fields(torch, dist, world, "cpu", 0.1 * (rank + 1), 10.0 * (rank + 1), 1.0 * (rank + 1))
    if rank == 0:
        print(json.dumps(dict({"selftest_spawn": True, "n_gpus": world, "ranks_in_all_reduce": int(ones.item()), "requested": a.gpus}, **sf)))
    dist.destroy_process_group()


def main():
    a = parse()
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not launched and a.config != "ransac":
        relaunch_under_torchrun(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not (world == 1 and a.config == "ransac"):
        raise SystemExit(f"bench.py --gpus {a.gpus} started with WORLD_SIZE = {world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus}) or let bench.py launch itself")
    if a.selftest_spawn:
        return selftest_spawn(a, rank, world)
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    if a.config == "kmeans":
        if world > 1:
            bench_kmeans_sharded(a, torch, rank, world, local_rank)
        else:
            bench_kmeans(a, torch)
    elif a.config == "ransac":
        if rank == 0:
            bench_ransac(a, torch)
    else:
        bench_icp(a, torch, rank, world, local_rank)


if __name__ == "__main__":
    main()
