"""What the K-nearest cap costs the score batch: the same batch with K = 64 (the reference's) and with K so large that
the rank of a match never has to be established."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
for k in (64, 32, 1_000_000):
    capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, k)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); sc = capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, k); ts.append(time.perf_counter() - t)
    print(f"K {k:8d}: {1e3*min(ts):.3f} ms, mean score {sc.mean():.4f}")
