"""How the GPU and the oracle drift apart over ICP iterations on one hard case (seed 7 of tests/stress/stress_parity.py)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
from oracle.pyoracle import Oracle
capi.init(0); O = Oracle()
I4 = np.eye(4, dtype=np.float32).ravel(); ang = np.float32(np.deg2rad(60.0))
seed = 7
rng = np.random.default_rng(seed); dens = float(rng.choice([600, 1500, 3000, 5000]))
s0 = synth.make_scene(seed=seed, density=dens, timestep=0); s1 = synth.make_scene(seed=seed, density=dens, timestep=1)
a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
r = float(rng.choice([0.05, 0.075, 0.1])); T0 = synth.perturbed_pose(I4, rng, rng.choice([0.005, 0.03, 0.1]), rng.choice([0.005, 0.03, 0.1]))
print("radius", r)
# oracle step by step: re-run with capped iterations is not exposed; emulate with find_corrs + estimate
To = T0.copy(); Tg = T0.copy(); md = np.float32(r)
for it in range(12):
    c = O.icp_find_corrs(s1["points"], s1["normals"], s0["points"], s0["normals"], To, I4, float(md), ang)
    eo, To = O.icp_estimate_pt2pl(c[0], c[2], c[3], c[4], To)
    eg, Tg2, _ = capi.icp_align(b, a, Tg, I4, float(md), float(ang), max_iter=1, fixed_iters=True)
    # same-start comparison: GPU one iteration from the ORACLE's pose
    eg_s, Tg_s, _ = capi.icp_align(b, a, c and To_prev if it else T0, I4, float(md), float(ang), max_iter=1, fixed_iters=True) if False else (0, None, 0)
    Tg = Tg2
    print(f"it {it:2d} max_dist {md:.5f}: |T_gpu - T_oracle| = {np.linalg.norm(To.astype(np.float64) - Tg):.3e}   err oracle {eo:.7f} gpu {eg:.7f}   n_corr {len(c[0])}")
    nd = float(md) * 0.95; md = np.float32(nd if nd > 0.05 else 0.05)

# fixed points: one more iteration of each solver from the ORACLE's converged pose, same radius
c = O.icp_find_corrs(s1["points"], s1["normals"], s0["points"], s0["normals"], To, I4, 0.05, ang)
eo, To2 = O.icp_estimate_pt2pl(c[0], c[2], c[3], c[4], To)
eg, Tg2, _ = capi.icp_align(b, a, To, I4, 0.05, float(ang), max_iter=1, fixed_iters=True)
print(f"from the oracle's pose: oracle moves {np.linalg.norm(To2.astype(np.float64) - To):.3e}, gpu moves {np.linalg.norm(Tg2.astype(np.float64) - To):.3e}, apart {np.linalg.norm(Tg2.astype(np.float64) - To2):.3e}")
# the estimate-only entry point on the oracle's own correspondences and weights
Te = capi.icp_estimate_pt2pl(c[0], c[2], c[3], c[4], To)
print(f"gpu estimate on the oracle's correspondences: apart from oracle {np.linalg.norm(np.asarray(Te[1], np.float64) - To2):.3e}")
# float64 numpy evaluation of the same normal equations (reference formula, exact arithmetic)
p1, p2, n2, w = [x.astype(np.float64) for x in (c[0], c[2], c[3], c[4])]
c1 = (w[:, None] * p1).sum(0) / w.sum(); c2 = (w[:, None] * p2).sum(0) / w.sum()
P = p1 - c1; Q = p2 - c2; D = P - Q; Cx = np.cross(P, n2); S = (D * n2).sum(1)
A = np.zeros((6, 6)); J = np.concatenate([Cx, n2], 1); A = (J * w[:, None]).T @ J; bb = (J * (w * S)[:, None]).sum(0)
x = np.linalg.solve(A, -bb)
print("exact-arithmetic step x =", x)
