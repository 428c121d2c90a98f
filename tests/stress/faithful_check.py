"""Reference-order estimator vs the oracle, bit for bit: poses, errors, iteration counts of whole ICP runs
(object-sized sources against a scene, and whole scans with the threshold lifted), and what it costs."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
from oracle.pyoracle import Oracle
capi.init(0)
O = Oracle()
I4 = np.eye(4, dtype=np.float32).ravel()
ang = np.float32(np.deg2rad(60.0))
bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def same(a, b):
    return np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes()


for seed in range(1, n_seeds + 1):
    rng = np.random.default_rng(seed)
    dens = float(rng.choice([600, 1500, 3000]))
    s0 = synth.make_scene(seed=seed, density=dens, timestep=0); s1 = synth.make_scene(seed=seed, density=dens, timestep=1)
    a = capi.Cloud(s0["points"], s0["normals"])
    # objects of scan t1 aligned to scan t0 (apps/pose_proposal regime)
    for k, o in enumerate(s1["objects"][:4]):
        oc = capi.Cloud(o["pos"], o["nor"])
        r = float(rng.choice([0.05, 0.075, 0.1]))
        T0 = synth.perturbed_pose(o["pose"], rng, 0.05, 0.05)
        e_o, T_o, it_o = O.icp_align(o["pos"], o["nor"], s0["points"], s0["normals"], T0, I4, r, ang)
        out = {}
        for mode, below in (("ref-order", 1 << 30), ("fp64", 0)):
            capi.icp_reference_order_below(below)
            capi.icp_align(oc, a, T0, I4, r, float(ang))
            t = time.perf_counter(); e_g, T_g, it_g = capi.icp_align(oc, a, T0, I4, r, float(ang)); dt = time.perf_counter() - t
            out[mode] = (same(T_o, T_g) and same(e_o, e_g) and it_o == it_g, np.linalg.norm(T_o.astype(np.float64) - T_g), it_g, dt)
        ok = out["ref-order"][0]
        bad += 0 if ok else 1
        print(f"seed {seed} object {k} n {len(o['pos']):6d} r {r}: oracle it {it_o}; ref-order {'BIT-EXACT' if ok else 'DIFFERS'} "
              f"(dT {out['ref-order'][1]:.1e}, it {out['ref-order'][2]}, {out['ref-order'][3]*1e3:.2f} ms) | fp64 dT {out['fp64'][1]:.1e}, it {out['fp64'][2]}, {out['fp64'][3]*1e3:.2f} ms", flush=True)
    # whole scan to whole scan, threshold lifted
    b = capi.Cloud(s1["points"], s1["normals"])
    T0 = synth.perturbed_pose(I4, rng, 0.03, 0.03)
    e_o, T_o, it_o = O.icp_align(s1["points"], s1["normals"], s0["points"], s0["normals"], T0, I4, 0.1, ang)
    capi.icp_reference_order_below(1 << 30)
    t = time.perf_counter(); e_g, T_g, it_g = capi.icp_align(b, a, T0, I4, 0.1, float(ang)); dt = time.perf_counter() - t
    ok = same(T_o, T_g) and same(e_o, e_g) and it_o == it_g
    bad += 0 if ok else 1
    capi.icp_reference_order_below(0)
    t = time.perf_counter(); e_f, T_f, it_f = capi.icp_align(b, a, T0, I4, 0.1, float(ang)); dt_f = time.perf_counter() - t
    print(f"seed {seed} scan n {len(s1['points']):7d}: oracle it {it_o}; ref-order {'BIT-EXACT' if ok else 'DIFFERS'} (dT {np.linalg.norm(T_o.astype(np.float64) - T_g):.1e}, it {it_g}, {dt*1e3:.1f} ms) "
          f"| fp64 dT {np.linalg.norm(T_o.astype(np.float64) - T_f):.1e}, it {it_f}, {dt_f*1e3:.1f} ms", flush=True)
    # the estimator entry point on the correspondences of the first iteration
    c = O.icp_find_corrs(s1["points"], s1["normals"], s0["points"], s0["normals"], T0, I4, 0.1, ang)
    e_o, T_o = O.icp_estimate_pt2pl(c[0], c[2], c[3], c[4], T0)
    capi.icp_reference_order_below(1 << 30)
    e_g, T_g = capi.icp_estimate_pt2pl(c[0], c[2], c[3], c[4], T0)
    ok = same(T_o, T_g) and same(e_o, e_g)
    bad += 0 if ok else 1
    print(f"seed {seed} estimate on {len(c[4])} correspondences: {'BIT-EXACT' if ok else 'DIFFERS'} (dT {np.linalg.norm(T_o.astype(np.float64) - T_g):.1e})", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
