"""Cost of the reference-order estimator: icp_align with a fixed number of iterations, both estimators, by source size."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
s0 = synth.scene_for_point_count(200_000, seed=11, timestep=0)
a = capi.Cloud(s0["points"], s0["normals"])
rng = np.random.default_rng(4)
T0 = synth.perturbed_pose(I4, rng, 0.01, 0.01)
IT = 20
for n in [int(x) for x in (sys.argv[1:] or ["2000", "8000", "16000", "32000"])]:
    sel = rng.choice(len(s0["points"]), n, replace=False)
    src = capi.Cloud(s0["points"][sel], s0["normals"][sel])
    row = []
    for below in (1 << 30, 0):
        capi.icp_reference_order_below(below)
        capi.icp_align(src, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=IT, fixed_iters=True)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); capi.icp_align(src, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=IT, fixed_iters=True); ts.append(time.perf_counter() - t)
        row.append(1e6 * min(ts) / IT)
    print(f"n {n:6d}: reference order {row[0]:7.1f} us/iteration, fp64 moments {row[1]:6.1f} us/iteration, difference {row[0]-row[1]:7.1f} us = {1e3*(row[0]-row[1])/n:.2f} ns/point", flush=True)
