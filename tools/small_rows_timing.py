"""Latency of one small msh_hash_grid_radius_search-shaped call (the unchanged pose_proposal's score loop: ~130 object points,
K = 64, r = 0.1 against a level-1 scene): wall-clock per call and the kernel's own time (HIP events)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
s = synth.make_scene(seed=100, density=3700.0, timestep=1)          # ~ a level-1 scene of a 124 k-point scan
scene = capi.Cloud(s["points"], None, cell_size=0.1)
rng = np.random.default_rng(0)
for nq, k in ((130, 64), (130, 32), (1000, 64), (10000, 64), (130, 1)):
    q = s["points"][rng.integers(0, len(s["points"]), nq)] + rng.normal(0, 0.01, (nq, 3)).astype(np.float32)
    for _ in range(20): capi.radius_search(scene, q, 0.1, k)
    t = time.perf_counter()
    for _ in range(300): capi.radius_search(scene, q, 0.1, k)
    wall = (time.perf_counter() - t) / 300
    capi.profile_enable(True); capi.profile_reset()
    for _ in range(100): d, i, nn, tot = capi.radius_search(scene, q, 0.1, k)
    n, ms = capi.profile_read("nn_rows"); capi.profile_enable(False)
    print(f"nq {nq:6d} K {k:3d}: {wall*1e6:7.1f} us per call, kernel {ms/n*1e3:6.1f} us, mean neighbours within r {tot/nq:.0f} (capped at K)", flush=True)
