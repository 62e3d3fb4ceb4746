"""CPU: the C-ABI library loads and exports every symbol include/cilantro_hip/c_api.h declares.
No compute calls are made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from cilantro_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cilantro_hip", "c_api.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cilhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_what_the_binding_lists():
    assert sorted(capi.SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(hip_lib):
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} is declared in c_api.h but not exported by libcilantro_hip.so"


def test_header_cites_reference_interfaces():
    txt = open(HEADER).read()
    for needle in ("icp_base.hpp:68-87", "kd_tree.hpp:162-170", "transform_estimation.hpp:237-367",
                   "transform_estimation.hpp\n * :11-48", "correspondence_search_kd_tree.hpp:107-229"):
        assert needle in txt, needle


def test_no_device_fails_loudly(hip_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert hip_lib.cilhip_create(ctypes.byref(h), 0) == capi.ERR_NO_DEVICE and not h.value
    from cilantro_amd.icp import Context

    with pytest.raises(capi.CilhipError):
        Context(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cilantro_amd/ may reference it."""
    pkg = os.path.join(ROOT, "cilantro_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|liboracle|libref_nanoflann|#include\s*[<\"][^>\"]*oracle|oracle/_ref", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not bad.search(src), os.path.join(dirpath, f)
    for f in ("c_api.h", "icp.hpp"):
        p = os.path.join(ROOT, "include", "cilantro_hip", f)
        if os.path.exists(p):
            assert not bad.search(open(p).read())
    # and the shared library has no dynamic dependency on it
    import subprocess

    deps = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps and "nanoflann" not in deps


def test_every_option_key_is_documented_in_the_header():
    """cilhip_set_option's keys (cilantro_amd/csrc/c_api.hip) all appear, quoted, in include/cilantro_hip/c_api.h"""
    import re

    src = open(os.path.join(ROOT, "cilantro_amd", "csrc", "c_api.hip")).read()
    hdr = open(os.path.join(ROOT, "include", "cilantro_hip", "c_api.h")).read()
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', src)))
    assert len(keys) >= 15
    missing = [k for k in keys if '"%s"' % k not in hdr]
    assert not missing, missing


def test_option_table_documents_every_option(hip_lib):
    """the typed option table (enum cilhip_option / cilhip_option_info): one row per key cilhip_set_option accepts, in the enum's order, each
    with a default inside its range and a line of documentation; every key is also named, quoted, by at least one test or dev tool."""
    import re

    opts = capi.options()
    hdr = open(HEADER).read()
    enum = re.search(r"typedef enum cilhip_option \{(.*?)\} cilhip_option;", hdr, re.S).group(1)
    names = [n.strip().split("=")[0].strip() for n in enum.replace("\n", " ").split(",") if n.strip()]
    assert names[-1] == "CILHIP_OPT_COUNT" and len(names) - 1 == len(opts)
    for i, (o, n) in enumerate(zip(opts, names)):
        assert o["id"] == i and n == "CILHIP_OPT_" + o["key"].upper(), (i, o["key"], n)
        assert len(o["doc"]) >= 20 and o["min"] <= o["default"] <= o["max"], o
    src = open(os.path.join(ROOT, "cilantro_amd", "csrc", "c_api.hip")).read()
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', src)))
    assert sorted(o["key"] for o in opts) == keys
    assert hip_lib.cilhip_option_info(-1) is None or not hip_lib.cilhip_option_info(-1)
    assert not hip_lib.cilhip_option_info(len(opts))
    # named by a test (or, for the dev-only A/B switches, by a dev tool)
    corpus = ""
    for d in ("tests", "tools"):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith(".py") and f != os.path.basename(__file__):
                corpus += open(os.path.join(ROOT, d, f), errors="ignore").read()
    corpus += open(os.path.join(ROOT, "bench.py")).read() + open(os.path.join(ROOT, "cilantro_amd", "icp.py")).read()
    unnamed = [o["key"] for o in opts if '"%s"' % o["key"] not in corpus]
    assert not unnamed, unnamed


@pytest.mark.gpu
def test_every_option_takes_its_default_and_reads_back():
    from cilantro_amd.icp import Context

    ctx = Context(0)
    for o in capi.options():
        ctx.set_option(o["key"], o["default"])
        assert ctx.get_option(o["key"]) == pytest.approx(o["default"]), o["key"]
        ctx._ck(ctx._L.cilhip_set_option_id(ctx._h, o["id"], o["default"]))
    for key, bad in (("tie_rule", 3), ("group_search", 5), ("search_direction", 7), ("kernel_timing_stride", 0), ("refined_occupancy_factor", 0.5), ("tiled", float("nan"))):
        with pytest.raises(capi.CilhipError):
            ctx.set_option(key, bad)
    with pytest.raises(capi.CilhipError):
        ctx.get_option("no_such_option")
    ctx.close()
