"""Plane RANSAC on a large synthetic cloud: device time per scored hypothesis (tools only, not the bench)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from cilantro_amd.model_estimation import PlaneRANSACEstimator3f

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
k = int(0.6 * n)
x[:k, 2] = 0.3 * x[:k, 0] - 0.2 * x[:k, 1] + 0.1 + 0.004 * torch.randn(k, device="cuda", generator=g)
torch.cuda.synchronize()
for target, label in ((n, "no early exit"), (n // 2, "default target")):
    for rep in range(2):
        pe = PlaneRANSACEstimator3f(x).setMaxInlierResidual(0.01).setTargetInlierCount(target).setMaxNumberOfIterations(iters).setSeed(7)
        t0 = time.perf_counter(); pe.estimate(); t1 = time.perf_counter()
    ms = pe.getDeviceMilliseconds()
    it = pe.getNumberOfPerformedIterations()
    print(f"plane RANSAC n={n} max_iter={iters} ({label}): performed {it}, inliers {pe.getNumberOfInliers()}, device {ms:.3f} ms "
          f"({n * min(iters, ((it + 127) // 128) * 128) / ms / 1e6:.1f} G point-plane tests/s), wall {1e3 * (t1 - t0):.1f} ms, model {pe.getModel()}")
