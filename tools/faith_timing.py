"""RS_HIP_LIB=.../librescan_hip_ftime<pass>.so python tools/faith_timing.py: cycles the waves of the sequential estimator spend at work / at the
barrier over one pass (tools/variant.sh ftime3 -DRS_FAITH_TIMING=3; ftime12 -DRS_FAITH_TIMING=12), one 54 k-point problem."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
s0 = synth.scene_for_point_count(20000, seed=11, timestep=0); s1 = synth.scene_for_point_count(20000, seed=11, timestep=1)
a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
T0 = synth.perturbed_pose(I4, np.random.default_rng(16), 0.01, 0.01)
print(capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=3, fixed_iters=True)[2], "iterations,", b.n, "source points")
capi.icp_faith_redone()
