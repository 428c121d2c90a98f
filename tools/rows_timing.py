"""msh_hash_grid_radius_search through rs_hip_radius_search (full rows, the compatibility path)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
s = synth.scene_for_point_count(1_000_000, seed=11, timestep=0)
c = capi.Cloud(s["points"], s["normals"])
rng = np.random.default_rng(0)
q = s["points"][rng.integers(0, len(s["points"]), 200_000)] + rng.normal(0, 0.005, (200_000, 3)).astype(np.float32)
for k, r in ((1, 0.05), (8, 0.05), (16, 0.1), (64, 0.1)):
    capi.radius_search(c, q[:1000], r, k)
    t = time.perf_counter(); d, i, nn, tot = capi.radius_search(c, q, r, k); dt = time.perf_counter() - t
    print(f"K={k:3d} r={r}: {len(q)} queries in {1e3*dt:8.2f} ms ({len(q)/dt/1e6:7.2f} M queries/s), mean neighbours {nn.mean():.1f}")
