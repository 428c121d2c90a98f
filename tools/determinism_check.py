"""ICP run to run: the pose must be bit-identical."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
n = int(sys.argv[1]); reps = int(sys.argv[2])
s0 = synth.scene_for_point_count(n, seed=11, timestep=0); s1 = synth.scene_for_point_count(n, seed=11, timestep=1)
a = capi.Cloud(s0["points"], s0["normals"]); b = capi.Cloud(s1["points"], s1["normals"])
I4 = np.eye(4, dtype=np.float32).ravel()
T0 = synth.perturbed_pose(I4, np.random.default_rng(16), 0.01, 0.01)
seen = {}
for r in range(reps):
    e, T, it = capi.icp_align(b, a, T0, I4, 0.10, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
    key = (np.float32(e).tobytes(), np.asarray(T, np.float32).tobytes())
    seen[key] = seen.get(key, 0) + 1
print("distinct results:", len(seen), sorted(seen.values(), reverse=True))
