"""Host-side timeline of one bench step: when each consumer's call starts and ends (three threads)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rescan_amd import capi  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

capi.init(0)
w = bench.build_workload(1_000_000, 11, "hash")
I4 = np.eye(4, dtype=np.float32).ravel()
pool = ThreadPoolExecutor(max_workers=3)


def timed(fn):
    def run():
        a = time.perf_counter(); r = fn(); b = time.perf_counter()
        return a, b, r
    return run


ops = {
    "icp": lambda: capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], I4, 0.10, np.deg2rad(60.0), max_iter=bench.ICP_ITERS, fixed_iters=True),
    "score": lambda: capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64),
    "label": lambda: capi.arrangement_to_labels(w["scan1"], w["plc_poses"], [p["cloud"] for p in w["plc"]], [0] * len(w["plc"]),
                                                 [p["cls"] for p in w["plc"]], 0.05, False),
}
for rep in range(6):
    t0 = time.perf_counter()
    fut = {k: pool.submit(timed(f)) for k, f in ops.items()}
    res = {k: f.result() for k, f in fut.items()}
    t1 = time.perf_counter()
    if rep >= 3:
        print(f"step {1e3 * (t1 - t0):.3f} ms: " + "  ".join(f"{k} {1e3 * (a - t0):.3f}->{1e3 * (b - t0):.3f}" for k, (a, b, _) in res.items()))
for k, f in ops.items():
    for _ in range(2):
        f()
    a = time.perf_counter(); f(); b = time.perf_counter()
    print(f"alone: {k} {1e3 * (b - a):.3f} ms")
