"""The C-ABI library loads on a machine without a GPU, exports every symbol include/mi355rt.h declares,
and refuses to render without a device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT, scene_path


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    assert declared_functions() == sorted(pkg.ABI_SYMBOLS + pkg.MULTI_ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    """libmi355rt.so exports the single-GPU ABI, libmi355rt_multi.so (the only one that links RCCL) the rt_*_multi entry points."""
    lib = ctypes.CDLL(pkg.LIB_PATH)
    multi = ctypes.CDLL(pkg.MULTI_LIB_PATH)
    for name in declared_functions():
        assert hasattr(multi if name in pkg.MULTI_ABI_SYMBOLS else lib, name), name
    assert lib.rt_abi_version() == 3
    needed = subprocess.run(["readelf", "-d", pkg.MULTI_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "librccl" in needed and "libmi355rt.so" in needed
    base = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "librccl" not in base


def test_multi_refuses_bad_device_lists(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    sc = pkg.Scene.load_from_file(scene_path("quadratic"))
    with pytest.raises(pkg.RtError) as e:
        pkg.MultiRenderer(sc, [0, 1])
    assert e.value.code == pkg.RT_ERR_NO_DEVICE


def test_update_backend_exports_reference_contract(pkg):
    """libmi355rt_update.so defines the three C++ symbols of the reference's include/update.h:6-8."""
    out = subprocess.run(["nm", "-D", "-C", "--defined-only", pkg.UPDATE_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert re.search(r"\binit_update\(unsigned int, Scene const&\)", out)
    assert re.search(r"\bupdate\(glm::mat<4, 4, double, \(glm::qualifier\)0> const&\)", out)
    assert re.search(r"\bcleanup_update\(\)", out)


def test_headers_cite_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    for cite in ("include/update.h:6", "include/update.h:7", "include/update.h:8", "include/scene.h", "src/scene.cpp"):
        assert cite in text


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sc = pkg.Scene.load_from_file(scene_path("cubic"))
    with pytest.raises(pkg.RtError) as e:
        pkg.Renderer(sc)
    assert e.value.code == pkg.RT_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_bad_arguments_are_reported_not_crashed(pkg):
    lib = pkg.lib()
    assert lib.rt_scene_get_desc(None, None) == -1
    assert b"null" in lib.rt_last_error()
    assert lib.rt_render(None, None, None, None, None) == -1
    assert lib.rt_destroy(None) == 0


def test_product_does_not_link_or_import_the_oracle(pkg):
    """The oracle is test infrastructure: nothing under cuda-ray-tracer_amd/ or include/ may reference it."""
    bad = []
    for base in (os.path.join(ROOT, "cuda-ray-tracer_amd"), os.path.join(ROOT, "include"), os.path.join(ROOT, "tools")):
        for dp, dn, fn in os.walk(base):
            if "build" in dp.split(os.sep):
                continue
            for f in fn:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".sh")) or f == "Makefile":
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"rt_oracle|librt_oracle|from oracle|import oracle|oracle/", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_no_kernel_spills_vgprs():
    """`make` refuses to link a library whose kernels spill VGPRs to scratch (tools/check_spills.py: with this compiler a
    spill can be stored under a narrowed EXEC mask and reloaded under the full one) or reserve a private segment at all;
    the report it leaves must say so."""
    report = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-ray-tracer_amd", "build", "spills.txt")
    if not os.path.exists(report):
        pytest.skip("library not built by this tree's Makefile")
    text = open(report).read()
    assert "VGPR SPILL" not in text and "PRIVATE SEGMENT" not in text and text.strip().endswith("no VGPR spills, no private segment")
    assert all(line.rstrip().endswith("scratch 0") for line in text.splitlines() if " scratch " in line)   # (a private segment costs ~0.7 us per launch)
    assert text.count("rt_wavefront") >= 32   # 16 instantiations x strict / fast
