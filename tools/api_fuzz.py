#!/usr/bin/env python
"""dev: differential fuzz of the context's STATE.  One long-lived context is driven through a random sequence of operations (new targets
/ sources of changing sizes -- with duplicated points now and then --, option changes, searches, estimates, whole loops, shared targets);
after every search / estimate / loop the same question is put to a FRESH context configured the same way.  Everything a context
caches between calls (grid, order tables, sorted source, matches, records, margins, pair lists, weight tables) has to be invisible:
correspondence lists element for element, transforms bit for bit (the loop is deterministic) -- any difference is a stale cache.
usage: api_fuzz.py [steps] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)

OPTS = {"search_direction": (0, 1, 2), "require_reciprocality": (0, 1), "inlier_fraction": (1.0, 0.7), "one_to_one": (0, 1), "tie_rule": (2, 1, 0),
        "tiled": (1, 0, 2), "warm_start": (1, 0, 2), "tile_accumulation": (1, 0, 2), "point_weight_evaluator": (0, 1, 2), "plane_weight_evaluator": (0, 2),
        "group_search": (-1, 0, 8), "symmetric_metric": (1, 0), "transform_mode": (0, 1), "affine_device_loop": (1, 0), "reverse_warm_start": (1, 0)}
DEFAULTS = {k: v[0] for k, v in OPTS.items()}


def cloud(n, dup):
    d = syn.make_pair(n, perturb=float(rng.uniform(0.1, 0.9)))
    if dup:      # doubled target points: exact ties in every search
        pick = rng.choice(n, max(n // 20, 1), replace=False)
        d["dst"] = np.ascontiguousarray(np.concatenate([d["dst"], d["dst"][pick]]))
        d["dst_n"] = np.ascontiguousarray(np.concatenate([d["dst_n"], d["dst_n"][pick]]))
    return d


class Model:
    """what the long-lived context has been told: enough to configure a fresh one identically"""

    def __init__(self):
        self.opts = dict(DEFAULTS)
        self.d = None
        self.src = None

    def fresh(self):
        c = Context()
        for k, v in self.opts.items():
            if v != DEFAULTS[k]:
                c.set_option(k, v)
        c.set_target(self.d["dst"], self.d["dst_n"])
        c.set_source(self.src)
        return c


def params(m, metric, iters):
    p = capi.IcpParams()
    capi.load().cilhip_icp_default_params(C.byref(p))
    p.metric, p.max_sq_dist, p.max_iter, p.conv_tol = metric, float(m.d["max_sq_dist"]), iters, 0.0
    if metric == capi.METRIC_COMBINED:
        p.w_p2p, p.w_p2pl = 0.1, 1.0
    return p


def same_lists(a, b, what, log):
    ok = len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a, b))
    if not ok:
        log.append(what)
    return ok


def plan(steps):
    """the whole sequence of operations with their random parameters, drawn up front (so that a replay may skip steps: FUZZ_ONLY)"""
    ops = [("target", int(rng.integers(2000, 60000)), float(rng.uniform(0.1, 0.9)), False, int(rng.integers(1 << 30)))]
    for _ in range(steps):
        op = str(rng.choice(["target", "source", "option", "search", "estimate", "loop", "loop"], p=[0.08, 0.12, 0.25, 0.2, 0.1, 0.15, 0.1]))
        if op == "target":
            ops.append((op, int(rng.integers(500, 150000)) if rng.random() < 0.7 else int(rng.integers(150000, 400000)), float(rng.uniform(0.1, 0.9)), bool(rng.random() < 0.3), int(rng.integers(1 << 30))))
        elif op == "source":
            ops.append((op, float(rng.random()), int(rng.integers(1 << 30))))
        elif op == "option":
            k = str(rng.choice(list(OPTS)))
            ops.append((op, k, OPTS[k][int(rng.integers(len(OPTS[k])))]))
        elif op == "search":
            ops.append((op, float(rng.uniform(0.0, 0.6)) if rng.random() < 0.7 else None, float(rng.choice([1.0, 0.3, 4.0]))))
        elif op == "estimate":
            ops.append((op,))
        else:
            ops.append((op, int(rng.choice([capi.METRIC_COMBINED, capi.METRIC_POINT_TO_POINT])), int(rng.integers(1, 9))))
    return ops


def make_cloud(n, perturb, dup, sd):
    d = syn.make_pair(n, perturb=perturb)
    if dup:      # doubled target points: exact ties in every search
        pick = np.random.default_rng(sd).choice(n, max(n // 20, 1), replace=False)
        d["dst"] = np.ascontiguousarray(np.concatenate([d["dst"], d["dst"][pick]]))
        d["dst_n"] = np.ascontiguousarray(np.concatenate([d["dst_n"], d["dst_n"][pick]]))
    return d


def main():
    ops = plan(steps)
    only = os.environ.get("FUZZ_ONLY")      # "3,17,40-60": the steps of the plan to execute (targets / sources / options among them change the model alike)
    keep = None
    if only:
        keep = set()
        for part in only.split(","):
            lo, _, hi = part.partition("-")
            keep.update(range(int(lo), int(hi or lo) + 1))
    live, m = Context(), Model()
    log, bad, done = [], [], {"search": 0, "estimate": 0, "loop": 0}
    for step, o in enumerate(ops):
        op = o[0]
        if keep is not None and step not in keep and step != 0 and op != "option":      # (options always apply: they are the configuration the kept steps run under)
            continue
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"[{step}] {o}  opts={ {k: v for k, v in m.opts.items() if v != DEFAULTS[k]} } nd={len(m.d['dst']) if m.d else 0} ns={len(m.src) if m.src is not None else 0}", file=sys.stderr, flush=True)
        try:
            if op == "target":
                m.d = make_cloud(o[1], o[2], o[3], o[4])
                m.src = m.d["src"]
                live.set_target(m.d["dst"], m.d["dst_n"]); live.set_source(m.src)
                log.append(f"{step}: target n={len(m.d['dst'])}")
            elif op == "source":
                k = max(100, int(o[1] * len(m.d["src"])))
                m.src = np.ascontiguousarray(m.d["src"][np.random.default_rng(o[2]).permutation(len(m.d["src"]))[:k]])
                live.set_source(m.src)
                log.append(f"{step}: source n={len(m.src)}")
            elif op == "option":
                live.set_option(o[1], o[2]); m.opts[o[1]] = o[2]
                log.append(f"{step}: {o[1]}={o[2]}")
            elif op == "search":
                T = syn.true_transform(m.d["h"], o[1]).astype(np.float32) if o[1] is not None else np.eye(4, dtype=np.float32)
                r2 = float(m.d["max_sq_dist"]) * o[2]
                live.find_correspondences(T, r2); a = live.get_correspondences()
                f = m.fresh(); f.find_correspondences(T, r2); b = f.get_correspondences(); f.close()
                done["search"] += 1
                if not same_lists(a, b, f"{step}: SEARCH differs (live {len(a[0])} pairs, fresh {len(b[0])})", bad):
                    bad.append("   history: " + " | ".join(log[-12:]))
            elif op == "estimate":
                T = syn.true_transform(m.d["h"], 0.3).astype(np.float32)
                r2 = float(m.d["max_sq_dist"])
                f = m.fresh()
                outs = []
                for c in (live, f):
                    c.find_correspondences(T, r2)
                    outs.append(c.estimate_combined(0.2, 1.0, 2, 1e-6)[0])
                f.close()
                done["estimate"] += 1
                if not np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)):
                    bad.append(f"{step}: ESTIMATE differs by {np.abs(outs[0] - outs[1]).max():.3g}"); bad.append("   history: " + " | ".join(log[-12:]))
            else:
                p = params(m, o[1], o[2])
                ra = live.icp_run(p); Ta = np.array(ra.T[:], np.float32); ca = live.get_correspondences()
                f = m.fresh(); rb = f.icp_run(p); Tb = np.array(rb.T[:], np.float32); cb = f.get_correspondences(); f.close()
                done["loop"] += 1
                okT = np.array_equal(Ta.view(np.uint32), Tb.view(np.uint32)) and int(ra.iterations) == int(rb.iterations) and int(ra.last_ncorr) == int(rb.last_ncorr)
                if not okT:
                    bad.append(f"{step}: LOOP differs (metric {o[1]}, {o[2]} iterations): |dT| = {np.abs(Ta - Tb).max():.3g}, ncorr {int(ra.last_ncorr)} / {int(rb.last_ncorr)}")
                    bad.append("   history: " + " | ".join(log[-12:]))
                elif not same_lists(ca, cb, f"{step}: the loop's last correspondence set differs", bad):
                    bad.append("   history: " + " | ".join(log[-12:]))
        except capi.CilhipError as e:
            log.append(f"{step}: {op} refused ({str(e)[:60]})")      # (a refusal -- an unsupported combination of options -- is an answer too)
    live.close()
    print(f"api_fuzz: {steps} steps (seed {seed}): {done['search']} searches, {done['estimate']} estimates, {done['loop']} loops compared with fresh contexts; {len([b for b in bad if not b.startswith('   ')])} differences")
    for b in bad[:40]:
        print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
