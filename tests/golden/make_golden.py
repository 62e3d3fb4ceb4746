#!/usr/bin/env python
"""Generates tests/golden/*.npz.  Run in the BUILD container (needs oracle/_ref, i.e. /root/reference):

    python tests/golden/make_golden.py

knn_golden.npz : outputs of the REFERENCE's own nanoflann 1.7.1 (oracle/_ref/libref_nanoflann.so, compiled
                 from /root/reference/include/cilantro/3rd_party/nanoflann/nanoflann.hpp) driven exactly as
                 cilantro drives it (core/kd_tree.hpp:63-109,162-170,283-291;
                 correspondence_search_kd_tree_utilities.hpp:7-51) on small seeded inputs.
icp_golden.npz : regression vectors of the ORACLE's ICP loop (the estimator half of the reference cannot be
                 run here -- Eigen3 is absent -- so these pin the oracle against itself over time and carry
                 the analytic ground truth T_true the recipe converges to).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cilantro_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    assert orc.ref_available(), "oracle/_ref missing: run `make -C oracle` where /root/reference exists"
    rng = np.random.default_rng(20250629)
    dst = rng.random((3000, 3), dtype=np.float32)
    dst[1500:1520] = dst[:20]                                  # exact duplicates (tie cases)
    q = (rng.random((800, 3), dtype=np.float32) * 1.2 - 0.1).astype(np.float32)
    q[:10] = dst[:10]
    tree = orc.KDTree(dst, use_ref=True)
    out = {"dst": dst, "q": q}
    for name, r2 in (("r_small", np.float32(0.002)), ("r_mid", np.float32(0.02)), ("r_inf", np.float32(3.0e38))):
        idx = np.full((len(q), 3), -1, np.int64)
        d2 = np.full((len(q), 3), np.nan, np.float32)
        for i in range(len(q)):
            ii, dd = tree.knn_in_radius(q[i], 3, r2)
            idx[i, : len(ii)] = ii
            d2[i, : len(ii)] = dd
        di, si, dv = tree.find_correspondences(q, r2, num_threads=1)
        out.update({f"{name}_r2": r2, f"{name}_knn3_idx": idx, f"{name}_knn3_d2": d2,
                    f"{name}_corr_first": di, f"{name}_corr_second": si, f"{name}_corr_value": dv})
    # examples/kd_tree.cpp:6-19
    cube = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1], [1, 1, 0], [1, 1, 1]], np.float32)
    ii, dd = orc.KDTree(cube, use_ref=True).knn_in_radius([0.1, 0.1, 0.4], 2, 1.001)
    out.update({"cube": cube, "cube_q": np.array([0.1, 0.1, 0.4], np.float32), "cube_idx": ii, "cube_d2": dd})
    np.savez_compressed(os.path.join(HERE, "knn_golden.npz"), **out)

    d = syn.make_pair(4000, perturb=0.5)
    icp = {"n": 4000, "perturb": 0.5, "T_true": d["T_true"], "max_sq_dist": d["max_sq_dist"]}
    for metric in (0, 1):
        for mode in (orc.MODE_F32, orc.MODE_MIXED, orc.MODE_F64):
            p = orc.make_params(metric=metric, max_iter=6, conv_tol=0.0, max_sq_dist=d["max_sq_dist"], mode=mode, num_threads=1)
            r = orc.icp_run(d["dst"], d["dst_n"], d["src"], p)
            icp[f"T_m{metric}_mode{mode}"] = r["T"]
            icp[f"ncorr_m{metric}_mode{mode}"] = r["last_ncorr"]
    np.savez_compressed(os.path.join(HERE, "icp_golden.npz"), **icp)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
