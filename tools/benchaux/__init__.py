"""Harness helpers that are NOT part of the product ABI: spin flags for bench.py's RoleRunner and the workgroup-placement probe
of the CU-mask diagnostics (tools/benchaux/rs_benchaux.hip -> tools/benchaux/librs_benchaux.so, built in-tree for gfx950)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rs_benchaux.hip")
LIB = os.path.join(HERE, "librs_benchaux.so")
_lib = None
_pylib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", SRC, "-o", LIB])
    return LIB


def _bind(lib):
    lib.rsb_spin_post.restype = C.c_int; lib.rsb_spin_post.argtypes = [C.c_void_p, C.c_int32]
    lib.rsb_spin_wait.restype = C.c_int; lib.rsb_spin_wait.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    lib.rsb_probe_placement.restype = C.c_int
    lib.rsb_probe_placement.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), C.c_int32]
    return lib


def load():
    global _lib, _pylib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(LIB))        # calls release the interpreter lock (spin_wait must)
        _pylib = _bind(C.PyDLL(LIB))     # ... and these do not (post_then_call's store)
    return _lib


def spin_post(flag_addr, value):
    """*flag = value (release)."""
    if load().rsb_spin_post(flag_addr, int(value)):
        raise RuntimeError("spin_post: null flag")


def spin_post_holding_gil(flag_addr, value):
    """The same store WITHOUT releasing the interpreter lock: a thread busy-waiting (lock-free, in spin_wait) for the flag wakes
    up at once but can only go on when the posting thread next releases the lock — i.e. when it has entered its own native
    call.  bench.py posts this way right before a consumer's library call, so that the longest consumer is under way before
    the others are released."""
    load()
    if _pylib.rsb_spin_post(flag_addr, int(value)):
        raise RuntimeError("spin_post: null flag")


def spin_wait(flag_addr, at_least, timeout_s=60.0):
    """Busy-waits (without the interpreter lock) until *flag >= at_least."""
    rc = load().rsb_spin_wait(flag_addr, int(at_least), float(timeout_s))
    if rc:
        raise RuntimeError("spin_wait: timed out" if rc == -3 else "spin_wait: null flag")


def probe_placement(n_blocks=4096, stream=None):
    """(xcc, se, sh, cu) of every workgroup of a probe launch on `stream` (default: the calling thread's library stream,
    rs_hip_get_stream) — which CUs a CU mask really selects (tools/cu_mask_probe.py)."""
    from rescan_amd import capi
    if stream is None:
        stream = capi.get_stream()
    out = np.zeros(n_blocks, np.uint32)
    if load().rsb_probe_placement(C.c_void_p(stream), out, n_blocks):
        raise RuntimeError("probe_placement failed")
    hw = out >> 8
    return out & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15
