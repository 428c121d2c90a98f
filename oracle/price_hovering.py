"""TEST INFRASTRUCTURE (CPU, build container).  VERDICT r05 item 4: what pose does "integer chains where the binade is high, fp64 over
hovering stretches" return on the three CENTRED rooms (bench.py --centre: the regime in which the reference's centroid sums wander
around zero), against the reference's — and what does the rigorous bound it would carry say.  oracle/rs_oracle.c:
orc_icp_iterate_variant, mode 2 (the shipped estimator's arithmetic: reference centroid chains + exact moments) and mode 4 (the same
with every chain summed in fp64 while |s| < 2^14), ten fixed iterations and icp_align with its stop test, on
tests/golden/bench_seed{11,31,32}_centre.npz.   ->  profiles/r06/hovering_fp64_variant.txt"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle.pyoracle import Oracle  # noqa: E402
from price_estimators import variant  # noqa: E402
import bench  # noqa: E402

I4 = np.eye(4, dtype=np.float32).ravel()
O = Oracle()
O.lib.orc_hover_bound.restype = None
O.lib.orc_hover_bound.argtypes = [C.POINTER(C.c_double)]
out = []
say = lambda s: (print(s, flush=True), out.append(s))  # noqa: E731
say("# pose distance (Frobenius) from the REFERENCE's pose, centred rooms (coordinates of both signs); mode 2 = reference centroid chains + exact moments")
say("# (what ships), mode 4 = the same with every chain in fp64 while |s| < 2^14; `bound` = rigorous sum of half-ulps over mode 4's hovering addends,")
say("# as a centroid error: worst chain's bound / total weight, in metres (the verdict's acceptance rule: <= 2e-5 m)")
for seed in (11, 31, 32):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "bench_seed%d_centre.npz" % seed)))
    w = bench.build_inputs(1_000_000, seed=seed, centre=True)
    s0, s1 = w["s0"], w["s1"]
    for label, T0, md, ma, n_it, stop, T_ref, it_ref in (
            ("10 fixed iterations", w["icp_T0"], 0.10, np.deg2rad(60.0), bench.ICP_ITERS, 0, g["icp_pose"], None),
            ("icp_align, stop test", g["stop_T0"], float(g["stop_params"][0]), float(g["stop_params"][1]), 100, 1, g["stop_pose"], int(g["stop_iters"]))):
        cells = []
        for m in (2, 4):
            t = time.time()
            e, T, it = variant(O, s1["points"], s1["normals"], s0["points"], s0["normals"], T0, I4, md, ma, n_it, stop, m)
            d = float(np.linalg.norm(T.astype(np.float64) - np.asarray(T_ref, np.float64).ravel()))
            cell = f"mode {m}: {d:.2e} ({it} it{'' if it_ref is None or it == it_ref else ' != ' + str(it_ref)})"
            if m == 4:
                hb = (C.c_double * 8)(); O.lib.orc_hover_bound(hb)
                cell += f", bound {max(hb[k] for k in range(7)) / hb[7]:.2e} m (chains: " + " ".join(f"{hb[k]:.1f}" for k in range(7)) + f"; total weight {hb[7]:.0f})"
            cells.append(cell + f" [{time.time() - t:.0f} s]")
        say(f"seed {seed} centred, {label}: " + " | ".join(cells))
os.makedirs(os.path.join(ROOT, "profiles", "r06"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "r06", "hovering_fp64_variant.txt"), "w").write("\n".join(out) + "\n")
