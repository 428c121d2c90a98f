"""Wall time of icp_align (to convergence) on small resident clouds: the app's regime (object level 2 vs scan level 2)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
out = []
for n in [int(x) for x in (sys.argv[1:] or ["50000"])]:
    s0 = synth.scene_for_point_count(n, seed=11, timestep=0); s1 = synth.scene_for_point_count(n, seed=11, timestep=1)
    T0 = synth.perturbed_pose(I4, np.random.default_rng(16), 0.01, 0.01)
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    obj = s1["objects"][0]; oc = capi.Cloud(obj["pos"], obj["nor"])
    To = synth.perturbed_pose(obj["pose"], np.random.default_rng(2), 0.03, 0.03)
    for name, src, T in (("scan->scan", b, T0), ("object->scan", oc, To)):
        capi.icp_align(src, a, T, I4, 0.1, np.deg2rad(60.0))
        ts = []
        for _ in range(5):
            t = time.perf_counter(); e, Tr, it = capi.icp_align(src, a, T, I4, 0.1, np.deg2rad(60.0)); ts.append(time.perf_counter() - t)
        out.append(f"{name} {src.n}->{a.n}: {1e3*min(ts):.2f} ms / {it} it = {1e6*min(ts)/it:.0f} us per iteration")
print(" | ".join(out))
