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


def read_ply_xyz_normals(path):
    """the reference's test clouds: binary_little_endian, vertex float x y z nx ny nz + uchar red green blue
    (what utilities/ply_io.hpp:43-106 reads through tinyply); read with the product's own PLY reader"""
    from cilantro_amd.ply_io import read_ply
    c = read_ply(path)
    return c["points"], c["normals"]


def make_frame1_fixture(out_path, ply="/root/reference/examples/test_clouds/frame_1.ply", keep=20000):
    """BASELINE configs[0] / SURVEY 8(d) C1: frame_1.ply through the recipe of examples/rigid_icp.cpp:25-65 (voxel
    downsample 0.005, src = dst + 0.01*uniform jitter, dst keeps x > -0.4, src moved by tf_ref), with OUR seeded jitter and a
    seeded subsample so the fixture stays small.  Real sensor data: a surface, strongly non-uniform density."""
    pts, nrm = read_ply_xyz_normals(ply)
    ok = np.isfinite(pts).all(1) & np.isfinite(nrm).all(1)
    pts, nrm = pts[ok], nrm[ok]
    # gridDownsample(0.005f): one averaged point / re-normalised normal per occupied voxel
    key = np.floor(pts / np.float32(0.005)).astype(np.int64)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    m = inv.max() + 1
    cnt = np.bincount(inv, minlength=m).astype(np.float64)
    dpts = np.stack([np.bincount(inv, pts[:, c].astype(np.float64), m) / cnt for c in range(3)], 1).astype(np.float32)
    dn = np.stack([np.bincount(inv, nrm[:, c].astype(np.float64), m) for c in range(3)], 1)
    dn = (dn / np.maximum(np.linalg.norm(dn, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    rng = np.random.default_rng(20250629)
    sel = np.sort(rng.permutation(m)[:keep])
    dpts, dn = dpts[sel], dn[sel]
    src = (dpts + np.float32(0.01) * rng.uniform(-1, 1, dpts.shape).astype(np.float32)).astype(np.float32)
    keep_dst = dpts[:, 0] > -0.4
    dst, dst_n = dpts[keep_dst], dn[keep_dst]
    cz, sz, cy, sy, cx, sx = np.cos(-0.1), np.sin(-0.1), np.cos(0.1), np.sin(0.1), np.cos(-0.1), np.sin(-0.1)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    R = (Rz @ Ry @ Rx).astype(np.float32)
    t = np.array([-0.20, -0.05, 0.10], np.float32)
    src = (src @ R.T + t).astype(np.float32)                       # src.transform(tf_ref)
    T_ref = np.eye(4, dtype=np.float32); T_ref[:3, :3] = R; T_ref[:3, 3] = t
    np.savez_compressed(out_path, dst=dst, dst_n=dst_n, src=src, T_ref=T_ref)
    return len(dst), len(src)


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


def make_frames_full_fixture(out_path, d="/root/reference/examples/test_clouds"):
    """The reference's two sensor frames at FULL resolution (120k points each, no voxel grid): frame_1 with its normals (the
    target of examples/rigid_icp.cpp), frame_2's points (a second, independent sampling of the same scene from a slightly
    different pose).  Large enough for every adaptive kernel form (tiles forced, per-lane, warm-started: >= 65 536 points)."""
    p1, n1 = read_ply_xyz_normals(os.path.join(d, "frame_1.ply"))
    ok = np.isfinite(p1).all(1) & np.isfinite(n1).all(1)
    p1, n1 = p1[ok], n1[ok]
    p2, _ = read_ply_xyz_normals(os.path.join(d, "frame_2.ply"))
    p2 = p2[np.isfinite(p2).all(1)]
    np.savez_compressed(out_path, p1=p1.astype(np.float32), n1=n1.astype(np.float32), p2=p2.astype(np.float32))
    return len(p1), len(p2)


def main_frame1():
    nd, ns = make_frame1_fixture(os.path.join(HERE, "frame1_c1.npz"))
    print(f"frame1_c1.npz: dst {nd} points, src {ns} points")
    n1, n2 = make_frames_full_fixture(os.path.join(HERE, "frames_full.npz"))
    print(f"frames_full.npz: frame_1 {n1} points (+ normals), frame_2 {n2} points")


if __name__ == "__main__":
    main()
    main_frame1()
