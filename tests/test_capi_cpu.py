"""CPU: the C-ABI library builds, loads, exports every symbol include/rescan_hip.h declares, its
host-side helpers match the golden vectors, and compute entry points FAIL LOUDLY without a GPU
(no CPU fallback anywhere in the product path)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def lib():
    from rescan_amd import build, capi
    build.build()
    return capi.load()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "rescan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    from rescan_amd import capi
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rescan_hip.h but not exported"
        assert n in capi.SIGNATURES, f"{n} has no ctypes signature in rescan_amd/capi.py"
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (rs_hip_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_code_object_is_gfx950_only():
    """Every device code object bundled in the library targets gfx950 (the .hip_fatbin entries are named
    `hipv4-amdgcn-amd-amdhsa--<arch>`; rocPRIM's host-side tuning tables mention other arch names as plain
    strings, which is not code)."""
    from rescan_amd import capi
    blob = open(capi.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--([a-z0-9:+-]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_does_not_touch_the_oracle():
    """rescan_amd/ and include/ never import, link or name anything under oracle/."""
    for base in ("rescan_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "pyoracle" not in txt and "rs_oracle" not in txt and "libref" not in txt, os.path.join(dp, f)
    from rescan_amd import capi
    needed = subprocess.check_output(["readelf", "-d", capi.LIB_PATH], text=True)
    assert "oracle" not in needed


def test_host_math_vs_golden(lib):
    from rescan_amd import capi
    g = load_golden("mat4.npz")
    for k in range(len(g["m"])):
        assert (capi.mat4_inverse(g["m"][k]) == g["inverse"][k]).all()
        assert (capi.mat4_mul(g["m"][k], g["b"][k]) == g["mul"][k]).all()


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rescan_amd import capi
    with pytest.raises(capi.RescanHipError):
        capi.init(0)
    with pytest.raises(capi.RescanHipError):
        capi.Cloud(np.zeros((8, 3), np.float32), np.zeros((8, 3), np.float32))
    rc = lib.rs_hip_synchronize()
    assert rc == -1 and b"no HIP device" in lib.rs_hip_last_error()


def test_missing_extension_is_an_error(monkeypatch, tmp_path):
    from rescan_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.RescanHipError):
        capi.load()


def test_graft_entry_build():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(os.path.join(ROOT, "rescan_amd", "librescan_hip.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "librs_oracle.so"))
