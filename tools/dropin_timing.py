"""What a caller of the reference-named icp_align (librescan_dropin.so) pays, host arrays in, pose out:
first call (hash + upload + device index build + ICP), repeated call (hash + cache hit + ICP), next to the
native entry point on resident clouds.  The PCIe-inclusive figures DESIGN.md quotes."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rescan_amd import capi, synth
from test_dropin import Mat4, DROPIN
capi.init(0)
lib = C.CDLL(DROPIN)
lib.icp_align.restype = C.c_float
lib.icp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Mat4), Mat4, C.c_float, C.c_float, C.c_bool]
lib.rsd_cache_clear.restype = None
I4 = np.eye(4, dtype=np.float32).ravel()
for n in (50_000, 1_000_000):
    s0 = synth.scene_for_point_count(n, seed=11, timestep=0); s1 = synth.scene_for_point_count(n, seed=11, timestep=1)
    T0 = synth.perturbed_pose(I4, np.random.default_rng(16), 0.01, 0.01)
    T2 = Mat4(); T2.data[:] = [float(x) for x in I4]
    def call():
        T = Mat4(); T.data[:] = [float(x) for x in T0]
        t = time.perf_counter()
        e = lib.icp_align(s1["points"].ctypes.data, s1["normals"].ctypes.data, len(s1["points"]), s0["points"].ctypes.data, s0["normals"].ctypes.data,
                          len(s0["points"]), C.byref(T), T2, 0.1, float(np.deg2rad(60.0)), False)
        return time.perf_counter() - t, e
    call(); lib.rsd_cache_clear()
    cold = min(call()[0] for _ in range(1)); warm = min(call()[0] for _ in range(3))
    lib.rsd_cache_clear(); cold2 = call()[0]
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0))
    t = time.perf_counter(); e, T, it = capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0)); nat = time.perf_counter() - t
    print(f"{len(s1['points']):8d} -> {len(s0['points']):8d} points, {it} iterations: shim first call {1e3*cold2:7.2f} ms, repeated {1e3*warm:7.2f} ms, native on resident clouds {1e3*nat:7.2f} ms")
