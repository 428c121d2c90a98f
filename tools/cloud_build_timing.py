"""How long rs_hip_cloud_create takes (host index build + upload) for typical cloud sizes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
for n in (10_000, 50_000, 250_000, 1_000_000):
    s = synth.scene_for_point_count(n, seed=3, timestep=0)
    p, q = s["points"], s["normals"]
    capi.Cloud(p[:1000].copy(), q[:1000].copy())
    ts = []
    for _ in range(3):
        t = time.perf_counter(); c = capi.Cloud(p, q); ts.append(time.perf_counter() - t); del c
    print(f"n = {len(p):8d}: cloud_create {1e3 * min(ts):8.2f} ms  ({1e9 * min(ts) / len(p):.0f} ns/point)")
