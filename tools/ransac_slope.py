import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cilantro_amd.model_estimation import PlaneRANSACEstimator3f
n = 50_000_000
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
rng = np.random.default_rng(3)
nrm = rng.normal(size=(128 * 64, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
planes = np.concatenate([nrm, rng.uniform(-0.5, 0.5, (len(nrm), 1))], axis=1).astype(np.float32)
pe = PlaneRANSACEstimator3f(x).setMaxInlierResidual(0.01)
pe.countInliers(planes[:256])
for k in (1, 2, 4, 8, 16, 32, 64, 1, 64):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pe.countInliers(planes[:128 * k]); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(k, "passes:", round(dt * 1e3, 3), "ms total,", round(dt * 1e3 / k, 4), "ms/pass", flush=True)
