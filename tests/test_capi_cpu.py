"""CPU: the C-ABI library builds, loads, exports every symbol include/rescan_hip.h declares, its
host-side helpers match the golden vectors, and compute entry points FAIL LOUDLY without a GPU
(no CPU fallback anywhere in the product path)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

LIB = os.path.join(ROOT, "rescan_amd", "librescan_hip.so")


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
    """rescan_amd/, include/, shadow/ and tools/ never import, link or name anything under oracle/ (only tests/,
    __graft_entry__.smoke() and bench.py's CPU-baseline leg do)."""
    for base in ("rescan_amd", "include", "shadow", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
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


def test_device_sincosf_model_is_the_host_libm(lib):
    """The reference composes its poses with libm's sinf/cosf (msh_vec_math.h:2091-2092), which are not correctly
    rounded (0.85 % of the arguments differ from the rounded double result); the ICP loop runs on the device, so the
    device carries a restatement of glibc's algorithm.  It must return libm's bits: ICP-step-sized angles, the whole
    polynomial range, the range-reduced one, and the edges of each."""
    import ctypes as C
    from rescan_amd import capi
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = C.c_float; libm.sinf.argtypes = [C.c_float]
    libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(1)
    x = np.concatenate([
        rng.uniform(-0.2, 0.2, 60000), rng.uniform(-0.75, 0.75, 40000), rng.uniform(-120, 120, 40000),
        10.0 ** rng.uniform(-7, -2, 20000) * rng.choice([-1, 1], 20000),
        np.array([0.0, -0.0, 2.0 ** -12, np.nextafter(np.float32(2.0 ** -12), np.float32(0)), 0.75, np.nextafter(np.float32(0.75), np.float32(0)),
                  np.pi / 4, np.pi / 2, np.pi, 119.99999, 120.0, 1e5, -1e5]),
    ]).astype(np.float32)
    s, c = capi.sincosf_model(x)
    want_s = np.array([libm.sinf(float(v)) for v in x], np.float32)
    want_c = np.array([libm.cosf(float(v)) for v in x], np.float32)
    # (beyond |x| = 120 the model rounds the double-precision function once: equal to libm's third branch
    #  except in its rare misroundings — not an ICP step, not asserted bit for bit)
    small = np.abs(x) < 120
    assert (s[small].view(np.uint32) == want_s[small].view(np.uint32)).all()
    assert (c[small].view(np.uint32) == want_c[small].view(np.uint32)).all()
    assert np.abs(s[~small] - want_s[~small]).max() < 1e-6 and np.abs(c[~small] - want_c[~small]).max() < 1e-6
    # the point of the exercise: libm is NOT the correctly rounded function
    assert (want_s != np.sin(x.astype(np.float64)).astype(np.float32)).sum() > 100


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


def test_spin_flags_hand_work_between_threads():
    """tools/benchaux (the harness's own helper library, NOT the product ABI): a worker thread and the caller ping-pong through two
    int32 flags without ever sleeping in a queue; a wait that cannot succeed times out; and none of these helpers is exported by
    librescan_hip.so any more (round-2 verdict: bench plumbing out of the product ABI)."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import benchaux
    flags = np.zeros(4, np.int32)
    addr = lambda k: flags.ctypes.data + 4 * k          # noqa: E731
    seen = []

    def worker():
        for k in range(1, 201):
            benchaux.spin_wait(addr(0), k, 30.0)
            seen.append(k)
            benchaux.spin_post(addr(1), k)

    t = threading.Thread(target=worker)
    t.start()
    for k in range(1, 201):
        (benchaux.spin_post if k % 2 else benchaux.spin_post_holding_gil)(addr(0), k)
        benchaux.spin_wait(addr(1), k, 30.0)
        assert seen[-1] == k
    t.join()
    with pytest.raises(RuntimeError):
        benchaux.spin_wait(addr(0), 10 ** 6, 0.05)
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    assert not re.search(r"rs_hip_(spin_|post_|probe_placement)", out)


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
