#!/usr/bin/env python
"""BASELINE configs[4] timing: KMeans3f k=1024 on 50M points, explicit initial centroids (first k points), 10 iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cilantro_amd import synthetic as syn
from cilantro_amd.clustering import KMeans3f
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
x = syn.make_dst(n)
xd = torch.from_numpy(x).cuda()
c0 = x[:k].copy()
KMeans3f(xd).cluster(c0, max_iter=1, tol=0.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
km = KMeans3f(xd).cluster(c0, max_iter=10, tol=0.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
it = km.getNumberOfPerformedIterations()
print(f"KMeans3f n={n} k={k}: {it} iterations in {dt*1e3:.1f} ms = {dt*1e3/max(it,1):.2f} ms/iter; "
      f"{n*k*it/dt/1e12:.2f} T point-centroid distances/s; min cluster size {np.bincount(km.getPointToClusterIndexMap(), minlength=k).min()}")
