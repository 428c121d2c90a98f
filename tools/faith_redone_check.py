import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
r0 = capi.icp_faith_redone()
tot_it = 0
for n in (20000, 50000):
    s0 = synth.scene_for_point_count(n, seed=11, timestep=0); s1 = synth.scene_for_point_count(n, seed=11, timestep=1)
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    for k, o in enumerate(s1["objects"][:3]):
        oc = capi.Cloud(o["pos"], o["nor"])
        for seed in range(4):
            To = synth.perturbed_pose(o["pose"], np.random.default_rng(seed), 0.03, 0.03)
            e, T, it = capi.icp_align(oc, a, To, I4, 0.1, np.deg2rad(60.0)); tot_it += it
print("iterations", tot_it, "one-pass statistics+centroids redone", capi.icp_faith_redone() - r0)
