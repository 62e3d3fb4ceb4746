"""dev: randomised A/B of round 6's loops against the forms they replace -- the affine classes' device-resident loop (affine_device_loop), the fused
warm-started reverse pass of FIRST_TO_SECOND / BOTH (reverse_warm_start), the warm-started feature search (feature_warm_start) -- over random
sizes, start transforms, noise levels, radii, weights, Gauss-Newton steps, duplicated points.  Same iterations / correspondence counts, transforms
equal to the order of the f64 additions.  usage: variants_stress.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import (CorrespondenceSearchDirection as D, SimpleCombinedMetricRigidICP3f, SimplePointToPointMetricRigidICP3f,
                              SimpleCombinedMetricAffineICP3f, SimplePointToPointMetricAffineICP3f)

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    kind = str(rng.choice(["affine", "reverse", "feature", "symmetric"]))
    n = int(rng.integers(70_000, 700_000)) if kind != "feature" else int(rng.integers(400_000, 900_000))
    ns = n if kind == "feature" else int(n * rng.uniform(0.3, 1.0))
    stride = 1 if ns == n else max(1, n // ns)
    ns = min(ns, n // stride)
    perturb, noise = float(rng.uniform(0.05, 0.9)), float(rng.uniform(0.0, 0.3))
    d = syn.make_pair(n, ns, with_normals=True, perturb=perturb, noise=noise, src_stride=stride)
    src = d["src"]
    if kind != "feature" and rng.random() < 0.3:      # doubled source points
        pick = rng.choice(len(src), max(len(src) // 50, 1), replace=False)
        src = np.ascontiguousarray(np.concatenate([src, src[pick]]))
    r2 = float(d["max_sq_dist"]) * float(rng.choice([0.25, 1.0, 4.0]))
    iters = int(rng.integers(2, 14))
    wts = [(0.0, 1.0), (1.0, 0.0), (0.2, 1.0)][int(rng.integers(3))]
    metric = int(rng.integers(2))
    steps = int(rng.choice([1, 1, 2, 3]))
    fw = float(rng.uniform(0.1, 1.0)) * float(d["h"])      # the feature adaptor's normal weight
    desc = f"{kind} n={n} ns={len(src)} perturb={perturb:.2f} noise={noise:.2f} r2x={r2 / float(d['max_sq_dist']):g} iters={iters} metric={metric} wts={wts} gn={steps}"
    got = []
    try:
        for on in (1, 0):
            if kind == "affine":
                icp = SimpleCombinedMetricAffineICP3f(d["dst"], d["dst_n"], src) if metric else SimplePointToPointMetricAffineICP3f(d["dst"], src)
                icp._ctx.set_option("affine_device_loop", on)
            elif kind == "symmetric":      # the four-cloud classes: warm-started (k_warm<., ., SYM>) against search + streaming pass every iteration
                sn = d["dst_n"][(np.arange(len(src)) * stride) % n]
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], src, np.ascontiguousarray(sn))
                icp._ctx.set_option("warm_start", (1 + case % 2) if on else 0)
            else:
                icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], src) if metric else SimplePointToPointMetricRigidICP3f(d["dst"], src)
            if metric or kind == "symmetric":
                icp.setPointToPointMetricWeight(wts[0]).setPointToPlaneMetricWeight(wts[1] if kind != "symmetric" else 1.0)
                if kind != "affine":
                    icp.setMaxNumberOfOptimizationStepIterations(steps).setOptimizationStepConvergenceTolerance(1e-7)
            eng = icp.correspondenceSearchEngine().setMaxDistance(r2)
            if kind == "reverse":
                direction, recip = [(D.FIRST_TO_SECOND, False), (D.BOTH, False), (D.BOTH, True)][case % 3]
                eng.setSearchDirection(direction).setRequireReciprocality(recip)
                icp._ctx.set_option("reverse_warm_start", on)
            if kind == "feature":
                eng.setPointNormalFeatureAdaptors(d["dst_n"], fw)
                icp._ctx.set_option("feature_warm_start", on)
            icp.setMaxNumberOfIterations(iters).setConvergenceTolerance(0.0)
            T = icp.estimate().getTransform()
            got.append((T.copy(), icp.getNumberOfPerformedIterations(), icp.last_ncorr_, icp._ctx.last_warm_iterations()))
    except Exception as e:      # (a refusal is an answer; anything else is reported)
        print(f"[{case}] {desc}: {type(e).__name__}: {str(e)[:120]}", flush=True)
        continue
    (T1, i1, n1, w1), (T0, i0, n0, w0) = got
    dT = float(np.abs(T1.astype(np.float64) - T0).max())
    tol = 5e-6
    ok = i1 == i0 and n1 == n0 and dT <= tol
    bad += 0 if ok else 1
    print(f"[{case}] {'ok ' if ok else 'DIFF'} {desc}: iterations {i1}/{i0} ncorr {n1}/{n0} max|dT| {dT:.3g} warm {w1}/{w0}", flush=True)
print(f"variants_stress: {cases} cases, {bad} differences")
