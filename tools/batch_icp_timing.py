"""Batched ICP: n start poses of one object against one scan in a single call (rs_hip_icp_align_batch)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
s = synth.scene_for_point_count(250_000, seed=11, timestep=1)
scan = capi.Cloud(s["points"], s["normals"])
obj = s["objects"][0]; oc = capi.Cloud(obj["pos"], obj["nor"])
rng = np.random.default_rng(2)
for n in (1, 4, 16, 64):
    T = np.stack([synth.perturbed_pose(obj["pose"], rng, 0.03, 0.03) for _ in range(n)])
    capi.icp_align_batch(oc, scan, T, I4, 0.1, np.deg2rad(60.0))
    ts = []
    for _ in range(3):
        t = time.perf_counter(); e, Tr, it = capi.icp_align_batch(oc, scan, T, I4, 0.1, np.deg2rad(60.0)); ts.append(time.perf_counter() - t)
    one = []
    for k in range(min(n, 4)):
        t = time.perf_counter(); capi.icp_align(oc, scan, T[k], I4, 0.1, np.deg2rad(60.0)); one.append(time.perf_counter() - t)
    print(f"{n:3d} poses x {oc.n} pts -> {scan.n}: batch {1e3*min(ts):7.2f} ms ({1e3*min(ts)/n:6.3f} ms per pose, iterations {it.min()}..{it.max()}); one at a time {1e3*np.mean(one):6.2f} ms per pose")
