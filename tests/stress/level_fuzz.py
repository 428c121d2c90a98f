"""Level builder against the oracle on many small clouds: random densities, radii, point orders (shuffled, raster,
Hilbert-like by object), duplicated points, lattices (distances exactly on the radius)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rescan_amd import capi, synth
from oracle.pyoracle import Oracle
capi.init(0)
O = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(1000 + case)
    kind = case % 5
    if kind == 0:
        pts = synth.make_scene(seed=case, density=float(rng.choice([300, 800, 2000])), timestep=0)["points"]
    elif kind == 1:
        pts = synth.make_object(str(rng.choice(["chair", "table"])), case, density=float(rng.choice([2000, 6000])))[0]
    elif kind == 2:      # lattice: many distances exactly equal to the radius
        g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.0078125)
        pts = g
    elif kind == 3:
        pts = rng.uniform(0, 1, (int(rng.integers(1, 5000)), 3)).astype(np.float32)
    else:
        base = rng.uniform(0, 0.6, (1500, 3)).astype(np.float32)
        pts = np.concatenate([base, base[rng.integers(0, 1500, 700)]])     # duplicates
    order = int(rng.integers(0, 3))
    if order == 1: pts = pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]
    if order == 2: pts = pts[rng.permutation(len(pts))]
    pts = np.ascontiguousarray(pts, np.float32)
    radius = float(rng.choice([0.0078125, 0.01, 0.02, 0.04, 0.08, 0.015625]))
    cap = int(rng.choice([256, 512, 1024]))
    want = O.level_poisson(pts, radius, cap)
    cloud = capi.Cloud(pts, None, cell_size=float(rng.choice([-1.0, 0.05, 0.2])))
    try:
        got, rounds = capi.level_samples(cloud, radius, cap)
        ok = len(got) == len(want) and (got == want).all()
        msg = f"{len(got)} samples, {rounds} steps"
    except capi.RescanHipError as e:
        # refused: legitimate only if some point really has more than `cap` points within the radius
        d, i, nn, _ = capi.radius_search(cloud, pts[:: max(1, len(pts) // 2000)], radius, min(cap + 1, 2048)) if cap + 1 <= 2048 else (None, None, np.array([cap + 1]), None)
        ok = int(nn.max()) > cap
        msg = "refused (cap), max neighbours in a sample of points %d" % int(nn.max())
    bad += 0 if ok else 1
    print(f"case {case:3d} kind {kind} order {order} n {len(pts):6d} r {radius:.4f} cap {cap}: {'OK' if ok else 'MISMATCH'} ({msg})", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
