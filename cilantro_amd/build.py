"""Build libcilantro_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cilantro_amd.build [--force]

Output: cilantro_amd/lib/libcilantro_hip.so (in-tree; git-ignored, travels with gpurun snapshots).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libcilantro_hip.so")
SOURCES = ["kernels.hip", "warm.hip", "epilogue.hip", "affine.hip", "feat_warm.hip", "extract.hip", "grid_build.hip", "filters.hip", "kmeans.hip", "ransac.hip", "ransac_transform.hip", "knn.hip", "bidir.hip", "tie_build.hip", "c_api.hip", "multi.hip"]
HEADERS = ["internal.hpp", "solve.hpp", "rccl_api.hpp", "search_device.hpp", "affine_device.hpp", os.path.join("..", "..", "include", "cilantro_hip", "c_api.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the pinned f32 expressions (d2, T*s, per-term residuals) must round exactly as
# written on host and device; f64 accumulations use explicit fma().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-result", "-DNDEBUG"]
# kernels.hip: the SLP vectoriser pairs scalar f32 operations of DIFFERENT candidates into v_pk_* instructions and pays
# for it in v_mov's that gather the operands (measured: +7 % VALU in the search kernel's hot block); the packed math that
# pays is written explicitly (f32x2).
EXTRA_FLAGS = {f: ["-fno-slp-vectorize"] for f in ("kernels.hip", "warm.hip", "epilogue.hip", "affine.hip", "extract.hip")}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), tag=""):
    """tag/defines: build a variant library libcilantro_hip<tag>.so with extra -D flags (dev A/B runs;
    select it with CILHIP_LIB_PATH)."""
    global OBJDIR, LIB
    if tag:
        OBJDIR = os.path.join(HERE, "lib", "obj" + tag)
        LIB = os.path.join(LIBDIR, f"libcilantro_hip{tag}.so")
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + [f"-D{d}" for d in defines] + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tags = [a[6:] for a in sys.argv[1:] if a.startswith("--tag=")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, tag=tags[0] if tags else ""))
