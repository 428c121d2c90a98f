"""Round 6: what the plain early iterations cost in pose distance from the REFERENCE and buy in time, by the number of chain iterations kept at
the end of a fixed-length call (RS_HIP_EARLY_TAIL, read at library load: one child process per setting).  Cases: the headline's ten fixed
iterations on every 1 M-point room with a fixture, the eight strong_icp refine units (50 k points each, one rs_hip_icp_align_multi call).
python tools/early_plain_table.py  ->  profiles/r06/early_plain.txt"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import torch; torch.cuda.init()
import bench
from rescan_amd import capi
capi.init(0)
I4 = np.eye(4, dtype=np.float32).ravel()
out = {}
for seed in (11, 23, 31, 32, 33, 34):
    g = dict(np.load(%r + "/tests/golden/bench_seed%%d.npz" %% seed))
    w = bench.build_workload(1_000_000, seed=seed, knn="hash")
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        e, T, it = capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], I4, 0.10, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        best = min(best, time.perf_counter() - t)
    out["room%%d" %% seed] = [float(np.linalg.norm(T.astype(np.float64) - g["icp_pose"].astype(np.float64))), best * 1e3]
    if seed == 11:
        si = w["strong_icp"]; plc = w["plc"][:8]
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            errs, Ts, its = capi.icp_align_multi([p["cloud"] for p in plc], w["scan1"], si["T0s"], I4, si["max_dist"], si["max_angle"], max_iter=10, fixed_iters=True)
            best = min(best, time.perf_counter() - t)
        d = np.linalg.norm(Ts.astype(np.float64).reshape(-1, 16) - g["strong_icp_pose"].astype(np.float64).reshape(-1, 16), axis=1)
        out["strong8"] = [float(d.max()), best * 1e3]
    for c in [w["scan0"], w["scan1"], w["obj_score"]] + [p["cloud"] for p in w["plc"]]:
        c.close()
print(json.dumps(out))
''' % (ROOT, ROOT, ROOT)
rows = []
for keep in (10, 7, 5, 4, 3, 2, 1):
    env = dict(os.environ, RS_HIP_EARLY_TAIL=str(keep))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    rows.append((keep, d))
    print(f"chain iterations kept at the end: {keep:2d} | " + " | ".join(f"{k}: {v[0]:.2e} {v[1]:.2f} ms" for k, v in d.items()), flush=True)
