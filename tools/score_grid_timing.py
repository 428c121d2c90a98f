"""The app's grid-search regime: thousands of poses of a coarse (level-4 / level-3) object against one scene."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
s = synth.scene_for_point_count(500_000, seed=11, timestep=1)      # ~ level 1 of a 1 M-point scan
scan = capi.Cloud(s["points"], s["normals"])
rng = np.random.default_rng(2)
op, on = synth.make_object("chair", 5, density=3800.0)
for stride, name in ((64, "level 4"), (16, "level 3"), (4, "level 2")):
    sub = rng.permutation(len(op))[::stride]
    oc = capi.Cloud(op[sub].copy(), on[sub].copy())
    for n_poses in (1000, 11500):
        lo, hi = s["points"].min(0), s["points"].max(0)
        poses = np.stack([synth.pose_matrix(rng.uniform(0, 2 * np.pi), np.array([rng.uniform(lo[0], hi[0]), 0.0, rng.uniform(lo[2], hi[2])]))
                          for _ in range(n_poses)]).astype(np.float32)          # the grid search: floor positions x rotations about the up axis
        capi.alignment_scores(oc, scan, poses[:10], 0.1, 64)
        t = time.perf_counter(); sc = capi.alignment_scores(oc, scan, poses, 0.1, 64); dt = time.perf_counter() - t
        print(f"{name}: {oc.n:5d} pts x {n_poses:6d} poses: {1e3*dt:8.2f} ms ({1e6*dt/n_poses:6.2f} us per pose, {oc.n*n_poses/dt/1e9:5.2f} G pairs/s), mean score {sc.mean():.3f}")
