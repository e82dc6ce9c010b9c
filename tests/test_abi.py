"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/kpb.h declares,
and the ctypes table mirrors the header.  No compute calls (there is no GPU here)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "kpb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kpb_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from keypoint_bench_amd import build, _lib
    so = build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(re.findall(r"\bT (kpb_[a-z0-9_]+)", out))
    declared = _header_functions()
    assert declared and set(declared) <= exported, sorted(set(declared) - exported)
    assert exported <= set(declared), "exported but undeclared: %s" % sorted(exported - set(declared))
    lib = _lib.load()
    assert set(_lib.SIGNATURES) == set(declared)
    assert lib.kpb_version() == 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.models.ALike import alike_t
    with pytest.raises(RuntimeError):
        detection(torch.rand(1, 1, 64, 64), None)
    with pytest.raises(RuntimeError):
        alike_t()(torch.rand(1, 3, 64, 64))
    # the C entry point itself refuses too
    import ctypes
    from keypoint_bench_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.kpb_ctx_create(0, None, ctypes.byref(h)) != 0
    assert b"no HIP device" in lib.kpb_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "keypoint_bench_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "kpb_oracle" not in txt, f
