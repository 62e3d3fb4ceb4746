"""Deterministic synthetic clouds for tests and bench (SURVEY.md section 8(d) recipe).

RNG: counter-based splitmix64 -- x_i = mix(seed + (i+1)*0x9E3779B97F4A7C15), u = (x >> 40) * 2^-24.
Streams: dst xyz seed 42, dst normals seed 43, source noise seed 44.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    """n outputs of the splitmix64 stream started at ``seed`` (skipping ``offset`` outputs)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1 + offset, n + 1 + offset, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n, offset=0):
    """float32 uniform in [0,1) with 24 random bits."""
    return ((splitmix64(seed, n, offset) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def rot_xyz(rx, ry, rz):
    """Rz(rz) @ Ry(ry) @ Rx(rx) as float64 3x3 (examples/rigid_icp.cpp:57-60 shape)."""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_dst(n, seed=42, offset=0):
    return uniform01(seed, 3 * n, 3 * offset).reshape(n, 3)


def make_normals(n, seed=43, offset=0):
    u = uniform01(seed, 2 * n, 2 * offset).reshape(n, 2)
    z = 2.0 * u[:, 0].astype(np.float64) - 1.0
    phi = 2.0 * np.pi * u[:, 1].astype(np.float64)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    nrm = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(nrm, dtype=np.float32)


def true_transform(h, scale=0.3):
    """T_true = Rz(-s h) Ry(s h) Rx(-s h), t = (-s h, s h/3, 2 s h/3) -> 4x4 float64."""
    a = scale * h
    T = np.eye(4)
    T[:3, :3] = rot_xyz(-a, a, -a)
    T[:3, 3] = [-a, a / 3.0, 2.0 * a / 3.0]
    return T


def make_pair(n_dst, n_src=None, with_normals=True, noise=0.05, perturb=0.3, src_stride=1,
              src_offset=0, seed_noise=44):
    """src = subset of dst + uniform noise (+-noise*h per axis), mapped by T_true^-1.

    Returns dict(dst, dst_n, src, T_true (4x4 f64: the transform ICP should recover), h, max_sq_dist).
    ``src_offset`` selects a different window of dst points / noise (used for source sharding).
    """
    n_src = n_dst if n_src is None else n_src
    dst = make_dst(n_dst)
    dst_n = make_normals(n_dst) if with_normals else None
    h = float(n_dst) ** (-1.0 / 3.0)
    sel = (src_offset + np.arange(n_src, dtype=np.int64) * src_stride) % n_dst
    base = dst[sel].astype(np.float64)
    nz = (uniform01(seed_noise, 3 * n_src, 3 * src_offset).reshape(n_src, 3).astype(np.float64) * 2.0 - 1.0) * (noise * h)
    T_true = true_transform(h, perturb)
    Ti = np.linalg.inv(T_true)
    src = ((base + nz) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return {
        "dst": np.ascontiguousarray(dst), "dst_n": dst_n, "src": np.ascontiguousarray(src),
        "T_true": T_true, "h": h, "max_sq_dist": np.float32((2.0 * h) ** 2),
    }
