"""RS_HIP_CHAIN_DEBUG=1 python tools/chain_debug.py [points]: which segments the centroid chains' walks add up addend by addend, and why."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
capi.icp_reference_order_below(0); capi.icp_replay_below(0); capi.icp_exact_centroids(1)
centred = "centred" in sys.argv[1:]          # both scans shifted so that the median point is the origin: chains that wander around zero
iters = max([int(a[6:]) for a in sys.argv[1:] if a.startswith("iters=")] or [3])
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [20_000, 300_000]:
    s0 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=0); s1 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=1)
    if centred:
        shift = -np.median(s1["points"], axis=0).astype(np.float32)
        s0["points"] = s0["points"] + shift; s1["points"] = s1["points"] + shift
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
    print(n, "points:", capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=iters, fixed_iters=True), flush=True)
    a.close(); b.close()
