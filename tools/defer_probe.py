#!/usr/bin/env python
"""dev: how many queries one tiled search hands to its clean-up pass (cilhip_debug_counters), recipe pair, identity and the true transform"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
d = syn.make_pair(n, n, with_normals=True)
for tr in (0, 2):
    ctx = Context(); ctx.set_option("tiled", 2); ctx.set_option("tie_rule", tr); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
    for name, T in (("identity", np.eye(4, dtype=np.float32)), ("true transform", d["T_true"].astype(np.float32))):
        nf = ctx.find_correspondences(T, float(d["max_sq_dist"]))
        print(f"n={n} tie_rule={tr} {name}: found {nf}, deferred (queries, whole tiles) = {ctx.debug_counters()}, exact ties by the diagnostic = {ctx.tie_count(T, float(d['max_sq_dist']))}", flush=True)
    ctx.close()
