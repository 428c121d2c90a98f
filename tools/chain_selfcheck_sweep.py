"""RS_HIP_CHAIN_DEBUG=1 python tools/chain_selfcheck_sweep.py 2> log; grep -c differs log
Whole icp_align runs over rooms of several sizes — as generated, moved to the origin, moved along one axis — with the walks'
self-check on (every step compared with the plain sum of its chain, rs_icp_estimate.hip: chain_walk_row): any line with "differs" or
"DIFFERS" in the log is a wrong step."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rescan_amd import capi, synth
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
capi.icp_reference_order_below(0); capi.icp_replay_below(0); capi.icp_exact_centroids(1)
rng = np.random.default_rng(17)
for n, seeds in ((60_000, (1, 2, 3)), (330_000, (4, 5, 6)), (700_000, (7, 8)), (1_150_000, (9, 10)), (2_600_000, (12,))):
    for seed in seeds:
        s0 = synth.scene_for_point_count(n, seed=seed, timestep=0); s1 = synth.scene_for_point_count(n, seed=seed, timestep=1)
        med = np.median(s1["points"], axis=0).astype(np.float32)
        for name, shift in (("as is", 0 * med), ("centred", -med), ("x only", -med * np.array([1, 0, 0], np.float32)), ("far", med * 20)):
            a, b = capi.Cloud(s0["points"] + shift, s0["normals"]), capi.Cloud(s1["points"] + shift, s1["normals"])
            for it in (1, 4):
                T0 = synth.perturbed_pose(I4, rng, 0.02, 0.01)
                g = capi.icp_chains_gave_up()
                print(f"[sweep] n {b.n} seed {seed} {name} iterations {it}", file=sys.stderr, flush=True)
                capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=it, fixed_iters=True)
                print(f"[sweep]   gave up: {capi.icp_chains_gave_up() - g}", file=sys.stderr, flush=True)
            a.close(); b.close()
print("done")
