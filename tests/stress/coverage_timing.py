"""Scene-coverage term: GPU (one arrangement per call, and batched) next to the reference on the host."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth  # noqa: E402
from oracle.pyoracle import Oracle, RefAO  # noqa: E402

capi.init(0)
s = synth.scene_for_point_count(200_000, seed=11, timestep=1)        # ~ level 2 of a 1M-point scan
pts = s["points"]
bmin, bmax = pts.min(0), pts.max(0)
rng = np.random.default_rng(0)
objs = s["objects"]
poses = [[synth.perturbed_pose(o["pose"], rng, 0.05, 0.05) for o in objs] for _ in range(256)]
cov = capi.Coverage(bmin, bmax, pts)
clouds = [capi.Cloud(o["pos"], o["nor"]) for o in objs]
n_obj_pts = sum(len(o["pos"]) for o in objs)
print(f"scene {len(pts)} pts, grid {tuple(cov.res)} = {cov.n_cells} voxels ({cov.valid_cells} active), {len(objs)} objects, {n_obj_pts} object points")
arrs = [[(clouds[k], poses[a][k], 0) for k in range(len(objs))] for a in range(256)]
for _ in range(3):
    cov.scores(arrs[:1]); cov.scores(arrs)
t = time.perf_counter()
for a in range(64):
    cov.scores(arrs[a:a + 1])
one = (time.perf_counter() - t) / 64
t = time.perf_counter(); sc, _ = cov.scores(arrs); batch = (time.perf_counter() - t) / 256
if RefAO.available():
    R = RefAO(synth.CLASS_IDX, pts, bmin, bmax)
    idx = [R.add_object(o["pos"], o["class_idx"], o["uidx"]) for o in objs]
    t = time.perf_counter(); ref = [R.coverage(idx, poses[a]) for a in range(32)]; cpu = (time.perf_counter() - t) / 32
    assert (np.array(ref, np.float32) == sc[:32]).all()
    print(f"reference (host, 1 thread): {1e6 * cpu:.0f} us per arrangement")
print(f"GPU: {1e6 * one:.0f} us per call (one arrangement), {1e6 * batch:.1f} us per arrangement in a batch of 256")
