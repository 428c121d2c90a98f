"""One whole-scan ICP (10 fixed iterations) with the seven centroid sums by pass 2 of the replay (rs_hip_icp_exact_centroids( 2 ): what
a problem the grid chains give up is run with), for rocprofv3 --kernel-trace --stats.  `centred` as an argument: coordinates that straddle zero."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
n = 1_170_000
s0 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=0); s1 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=1)
if "centred" in sys.argv[1:]:
    shift = -np.median(s1["points"], axis=0).astype(np.float32)
    s0["points"] = s0["points"] + shift; s1["points"] = s1["points"] + shift
a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
capi.icp_exact_centroids(int(os.environ.get("MODE", "2")))
for _ in range(3):
    capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
print("gave up:", capi.icp_chains_gave_up())
