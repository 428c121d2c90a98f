"""One whole-scan ICP with the centroid chains (grid chains), for rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_170_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s0 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=0); s1 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=1)
a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
capi.icp_reference_order_below(0); capi.icp_replay_below(0); capi.icp_exact_centroids(mode)
for _ in range(3):
    capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
