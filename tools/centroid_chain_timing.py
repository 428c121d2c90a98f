"""Per-iteration cost of the estimator on a whole-scan source (10 fixed iterations, resident clouds): plain fp64 moments, the fp64
step centred on the reference's centroid chains (grid chains: the default above 262 144 source points; the same sums through
pass 2 of the replay: the cross-check), and the fully reference-order parallel chains."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
# `centred` as an argument: both scans shifted so that the coordinates straddle zero (the median point at the origin) — chains that wander
# around zero change binade, and sign, far more often than those of a room in the positive octant
centred = "centred" in sys.argv[1:]
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [300_000, 1_170_000]:
    s0 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=0); s1 = synth.scene_for_point_count(int(n * 0.84), seed=11, timestep=1)
    if centred:
        shift = -np.median(s1["points"], axis=0).astype(np.float32)
        s0["points"] = s0["points"] + shift; s1["points"] = s1["points"] + shift
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
    capi.icp_reference_order_below(0)
    out, poses = [], {}
    for name, rp, ec in (("fp64 moments", 0, 0), ("grid chains", 0, 1), ("replay pass 2", 0, 2), ("parallel chains (all 45)", 1 << 30, 0)):
        capi.icp_replay_below(rp); capi.icp_exact_centroids(ec)
        capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            e, T, it = capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
            best = min(best, (time.perf_counter() - t) / 10)
        poses[name] = T
        out.append(f"{name}: {best*1e6:7.1f} us/iter" + (f" ({capi.icp_replay_redone()} segments added one by one in the last iteration)" if ec == 1 else ""))
    d = lambda x, y: float(np.linalg.norm(poses[x].astype(np.float64) - poses[y]))
    print(f"{len(s1['points']):8d} source points: " + " | ".join(out), flush=True)
    print(f"          pose distance to the parallel chains' (= the reference's bits): fp64 {d('fp64 moments', 'parallel chains (all 45)'):.2e}, grid chains {d('grid chains', 'parallel chains (all 45)'):.2e}; grid chains == replay pass 2: {bool((poses['grid chains'] == poses['replay pass 2']).all())}", flush=True)
    a.close(); b.close()
capi.icp_reference_order_below(65536); capi.icp_replay_below(262144); capi.icp_exact_centroids(1)
