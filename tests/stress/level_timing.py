"""Level builder (rs_pointcloud__compute_level_poisson) on the GPU against the oracle: sample sets, rounds, time,
for the generator's point order and for a raster (z, y, x) vertex order."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
from oracle.pyoracle import Oracle, LEVEL_VOXEL, level_max_n_neigh
capi.init(0)
O = Oracle()
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
s = synth.scene_for_point_count(n_pts, seed=11, timestep=0)
bad = 0
for name, pts in (("generator order", s["points"]), ("raster order", s["points"][np.lexsort((s["points"][:, 0], s["points"][:, 1], s["points"][:, 2]))])):
    pts = np.ascontiguousarray(pts)
    t = time.perf_counter(); cloud = capi.Cloud(pts, None); t_build = time.perf_counter() - t
    for level in (1, 2, 3, 4):
        r, k = LEVEL_VOXEL[level], level_max_n_neigh(level)
        t = time.perf_counter(); want = O.level_poisson(pts, r, k); t_cpu = time.perf_counter() - t
        capi.level_samples(cloud, r, k)
        t = time.perf_counter(); got, rounds = capi.level_samples(cloud, r, k); t_gpu = time.perf_counter() - t
        ok = len(want) == len(got) and (want == got).all()
        bad += 0 if ok else 1
        print(f"{name:16s} level {level} (r {r}): {len(pts)} -> {len(got)} samples, {'IDENTICAL' if ok else 'DIFFERENT (%d vs %d)' % (len(want), len(got))}, "
              f"{rounds} rounds, GPU {1e3*t_gpu:.2f} ms (cloud index {1e3*t_build:.1f} ms), CPU oracle {1e3*t_cpu:.1f} ms", flush=True)
# the whole pyramid of a scan as device-resident clouds (levels 1-4: samples + gather + index, rs_hip_cloud_create_level)
pts = np.ascontiguousarray(s["points"]); nor = np.ascontiguousarray(s["normals"])
base = capi.Cloud(pts, nor)
for rep in range(2):
    t = time.perf_counter()
    lv = [capi.Cloud.level_of(base, LEVEL_VOXEL[l], level_max_n_neigh(l))[0] for l in (1, 2, 3, 4)]
    dt = time.perf_counter() - t
print(f"levels 1-4 of {len(pts)} points as device clouds: {1e3*dt:.1f} ms ({', '.join(str(c.n) for c in lv)} points)")
sys.exit(1 if bad else 0)
