"""The three estimators per ICP iteration (search + estimator + loop bookkeeping; resident clouds), scan-to-scan at several sizes:
fp64 moments, the reference's sequential chains (k_icp_faithful), the same chains computed in parallel (replay)."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
for n in (5_000, 10_000, 20_000, 40_000, 60_000, 134_000, 260_000, 500_000):
    s0 = synth.scene_for_point_count(int(n * 0.84), seed=3, timestep=0); s1 = synth.scene_for_point_count(int(n * 0.84), seed=3, timestep=1)
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
    out = []
    for name, ro, rp in (("fp64 moments", 0, 0), ("sequential chains", 1 << 30, 0), ("parallel chains", 0, 1 << 30)):
        if name == "sequential chains" and len(s1["points"]) > 300_000:
            out.append(f"{name}: (skipped)"); continue
        capi.icp_reference_order_below(ro); capi.icp_replay_below(rp)
        capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        t = time.perf_counter()
        e, T, it = capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        dt = (time.perf_counter() - t) / 10
        out.append(f"{name}: {dt*1e6:7.1f} us/iter" + (f" (re-added segments, last iteration: {capi.icp_replay_redone()})" if rp else ""))
        if name == "sequential chains": Tseq = T
        if name == "parallel chains" and len(s1["points"]) <= 300_000: out.append("bits equal" if (T == Tseq).all() else "BITS DIFFER")
    print(f"{len(s1['points']):8d} source points: " + " | ".join(out), flush=True)
    a.close(); b.close()
capi.icp_reference_order_below(65536); capi.icp_replay_below(262144)
