import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
for n, centred in ((50000, False), (150000, False), (1170000, True)):
    s0 = synth.scene_for_point_count(int(n if n < 1e6 else n * 0.84), seed=11, timestep=0); s1 = synth.scene_for_point_count(int(n if n < 1e6 else n * 0.84), seed=11, timestep=1)
    sh = -np.median(s1["points"], axis=0).astype(np.float32) if centred else np.zeros(3, np.float32)
    a, b = capi.Cloud(s0["points"] + sh, s0["normals"]), capi.Cloud(s1["points"] + sh, s1["normals"])
    T0 = synth.perturbed_pose(I4, np.random.default_rng(16), 0.01, 0.01)
    if n > 1e6: capi.icp_reference_order_below(0); capi.icp_replay_below(0); capi.icp_exact_centroids(2)
    e, T, it = capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=6, fixed_iters=True)
    print(b.n, "points, centred" if centred else "points", ": segments", (b.n + 127) // 128, "re-added one by one over", it, "iterations (all rows):", capi.icp_replay_redone())
