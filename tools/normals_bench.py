"""NormalEstimation3f / k-NN on a large synthetic cloud (tools only)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd.normal_estimation import KDTree3f, NormalEstimation3f

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(1)
x = rng.random((n, 3), dtype=np.float32)
x[: n // 2, 2] = 0.2 * x[: n // 2, 0] + 0.1 * np.sin(6 * x[: n // 2, 1])
for rep in range(2):
    t0 = time.perf_counter(); nrm, cur = NormalEstimation3f(x).setViewPoint([0, 0, 10]).getNormalsAndCurvatureKNN(k); t1 = time.perf_counter()
print(f"normals k={k} n={n}: {1e3*(t1-t0):.1f} ms wall (upload + grid + k-NN + PCA + download) = {n/(t1-t0)/1e6:.1f} M points/s")
for rep in range(2):
    t0 = time.perf_counter(); idx, d2, cnt = KDTree3f(x).kNNSearch(None, k); t1 = time.perf_counter()
print(f"self k-NN k={k} n={n}: {1e3*(t1-t0):.1f} ms wall = {n/(t1-t0)/1e6:.1f} M queries/s")
