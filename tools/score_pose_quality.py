"""How the score batch's cost depends on the quality of the poses (fraction of object points with a match)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
s1 = synth.scene_for_point_count(840_000, seed=11, timestep=1)
scan = capi.Cloud(s1["points"], s1["normals"])
op, on = synth.make_object("table", 11 * 13 + 1, density=3800.0)
oc = capi.Cloud(op, on)
tbl = [o for o in s1["objects"] if o["kind"] == "table"][0]
rng = np.random.default_rng(16)
for rot, tr in ((0.0, 0.0), (0.02, 0.01), (0.1, 0.05), (0.3, 0.12), (0.6, 0.25), (1.5, 1.0)):
    poses = np.stack([synth.perturbed_pose(tbl["pose"], rng, rot, tr) for _ in range(256)])
    capi.alignment_scores(oc, scan, poses, 0.1, 64)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); sc = capi.alignment_scores(oc, scan, poses, 0.1, 64); ts.append(time.perf_counter() - t)
    print(f"perturbation rot {rot:4.2f} rad, trans {tr:4.2f} m: {1e3*min(ts):6.3f} ms for 256 poses x {oc.n} pts, mean score {sc.mean():.3f}", flush=True)
