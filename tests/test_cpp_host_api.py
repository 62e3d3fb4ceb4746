"""The C++ host mirror (include/cilantro_hip/icp.hpp) compiled with g++ against the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "test_icp")
BIN2 = os.path.join(ROOT, "tests", "cpp", "bin", "test_model_estimation")


def _ensure_built(hip_lib, orc):
    if not (os.path.exists(BIN) and os.path.exists(BIN2)):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cpp", "build.sh")])


def test_cpp_header_compiles_and_fails_loudly_without_device(hip_lib, orc):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cpp", "build.sh")])   # compile check of the header
    for b in (BIN, BIN2):
        out = subprocess.run([b, "--expect-no-device"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_api_parity_on_gpu(hip_lib, orc):
    _ensure_built(hip_lib, orc)
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_model_estimation_mirrors_on_gpu(hip_lib, orc):
    """PlaneRANSACEstimator3f / KMeans3f through include/cilantro_hip/model_estimation.hpp"""
    _ensure_built(hip_lib, orc)
    out = subprocess.run([BIN2], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


def test_solve_fast_paths_on_host(hip_lib, orc):
    """cilantro_amd/csrc/solve.hpp is shared by the host API and the single-lane device epilogue: its fast paths (polar
    iteration for the rotation() polish, unpivoted register-resident LDL^T) against the general ones (SVD, pivoted LDL^T
    with pseudo-inverse), compiled for the host -- no GPU needed."""
    binp = os.path.join(ROOT, "tests", "cpp", "bin", "test_solve")
    if not os.path.exists(binp):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cpp", "build.sh")])
    out = subprocess.run([binp], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
