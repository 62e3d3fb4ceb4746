import time, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from cilantro_amd import synthetic as syn
from cilantro_amd.clustering import KMeans3f
from cilantro_amd.normal_estimation import set_knn_tie_rule
n, k = 50_000_000, 1024
x = syn.make_dst(n); xd = torch.from_numpy(x).cuda(); c0 = x[:k].copy()
for rule in (2, 0, 2, 0):
    set_knn_tie_rule(rule)
    KMeans3f(xd).cluster(c0, max_iter=2, tol=0.0, use_kd_tree=True)
    def timed(it):
        torch.cuda.synchronize(); t0 = time.perf_counter(); KMeans3f(xd).cluster(c0, max_iter=it, tol=0.0, use_kd_tree=True); torch.cuda.synchronize(); return time.perf_counter() - t0
    t1 = timed(1); t11 = timed(11)
    print(f"tie rule {rule}: {(t11 - t1) / 10 * 1e3:.3f} ms per Lloyd iteration (kd branch, k = {k}, n = {n})", flush=True)
set_knn_tie_rule(2)
