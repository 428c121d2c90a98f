"""Where can an ICP run still differ from the reference's bits?  Voxel-averaged level-2 clouds (rs_pointcloud.h:905-975,
2 cm) of an object against a scan, from good and from bad start poses: correspondences GPU vs oracle, and for every
mismatch the two candidates' distances (an exact fp32 distance tie is decided by the reference's heap/sort order)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rescan_amd import capi, synth
from oracle.pyoracle import Oracle
capi.init(0)
O = Oracle()
I4 = np.eye(4, dtype=np.float32).ravel()
ang = np.float32(np.deg2rad(60.0))


from tie_hunt_lib import level

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
runs = bad_runs = corr_mismatch = tie_runs = 0
for seed in range(1, n_seeds + 1):
    rng = np.random.default_rng(100 + seed)
    s0 = synth.make_scene(seed=seed, density=2000.0, timestep=0); s1 = synth.make_scene(seed=seed, density=2000.0, timestep=1)
    sp, sn = level(s0["points"], s0["normals"], 0.02)
    a = capi.Cloud(sp, sn)
    for k, o in enumerate(s1["objects"][:3]):
        op, on = level(o["pos"], o["nor"], 0.02)
        oc = capi.Cloud(op, on)
        for trial in range(6):
            T0 = synth.perturbed_pose(o["pose"], rng, *( (0.03, 0.03) if trial < 2 else (0.6, 0.5) ))
            r = float(rng.choice([0.05, 0.1]))
            e_o, T_o, it_o = O.icp_align(op, on, sp, sn, T0, I4, r, ang)
            e_g, T_g, it_g = capi.icp_align(oc, a, T0, I4, r, float(ang))
            runs += 1
            if T_o.tobytes() == np.asarray(T_g, np.float32).tobytes() and it_o == it_g:
                continue
            bad_runs += 1
            # Replay the oracle's iterations (find_corrs + estimate; each find_corrs builds its grid with the CURRENT
            # radius as cell, icp_align builds it once with the initial one, icp.h:437).  If the GPU equals the replay,
            # the reference's own answer depends on its grid's cell size at this input: an exact fp32 distance tie
            # (between the winner and a rival, or at the K-th place), decided by its bin traversal / heap / sort order.
            T = np.asarray(T0, np.float32).copy(); md = np.float32(r); corr_differs = False
            for it in range(it_o):
                w = O.icp_find_corrs(op, on, sp, sn, T, I4, float(md), ang)
                g = capi.icp_find_corrs(oc, a, T, I4, float(md), float(ang))
                if not all(x.shape == y.shape and (x == y).all() for x, y in zip(w, g)):
                    corr_differs = True; corr_mismatch += 1
                    break
                if len(w[4]) == 0: break
                _, T = O.icp_estimate_pt2pl(w[0], w[2], w[3], w[4], T)
                md = np.float32(max(float(md) * 0.95, 0.05))
            replay_is_gpu = (not corr_differs) and T.tobytes() == np.asarray(T_g, np.float32).tobytes()
            tie_runs += replay_is_gpu
            print(f"seed {seed} object {k} ({len(op)} pts vs {len(sp)}) trial {trial} r {r}: it {it_o} vs {it_g}, dT {np.linalg.norm(T_o.astype(np.float64) - T_g):.2e}: "
                  + ("correspondences differ from the oracle's at the same pose" if corr_differs else
                     "GPU == replay with per-radius grids: the reference's result depends on its grid cell here (distance tie)" if replay_is_gpu else "unexplained"), flush=True)
print(f"runs {runs}, bit-identical {runs - bad_runs}, grid-dependent ties in the reference {tie_runs}, differing correspondences {corr_mismatch}, unexplained {bad_runs - tie_runs - corr_mismatch}")
sys.exit(1 if bad_runs - tie_runs else 0)
