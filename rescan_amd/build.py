"""Builds librescan_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librescan_hip.so")
DROPIN = os.path.join(HERE, "librescan_dropin.so")
SOURCES = ["rs_icp_search.hip", "rs_icp_estimate.hip", "rs_score.hip", "rs_rows.hip", "rs_build.hip", "rs_api.hip"]
HEADERS = ["rs_device.h", "rs_math.h", "rs_search.h", "rs_icp.h", "rs_dropin.cpp", os.path.join("..", "..", "include", "rescan_hip.h"),
           os.path.join("..", "..", "include", "rescan_dropin.h")]
# -ffp-contract=off: the neighbour-deciding arithmetic must round exactly like the reference's
# scalar SSE2 code (no FMA); see DESIGN.md.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def sources_sha():
    """Short digest of the device/host sources the library is built from: profiles that describe kernels carry it, and
    bench.py refuses a PMC traffic file made for other sources."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + [x for x in HEADERS if x.endswith(".h") and not x.startswith("..")]):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(DROPIN):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    hdrs = [os.path.join(CSRC, h) for h in HEADERS if h.endswith(".h")] + [os.path.abspath(__file__)]
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        deps = [os.path.join(CSRC, s)] + hdrs
        if not force and os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps):
            continue                    # this object is current (rs_build.hip pulls in hipCUB: ~20 s)
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    # the reference-named entry points (icp_align, msh_hash_grid_*) live in their own library so
    # that they never collide with a reference build loaded in the same process
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
           os.path.join(CSRC, "rs_dropin.cpp"), "-o", DROPIN, "-L" + HERE, "-lrescan_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
