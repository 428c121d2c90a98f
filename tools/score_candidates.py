"""Candidates staged by the score batch of the bench (per (tile, pose) wave)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64)
capi.profile_enable(True); capi.profile_reset()
capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64)
c = capi.profile_read("candidates")[0]; ms = capi.profile_read("nn_score")
n_tiles = -(-w["n_obj"] // 64)
print(f"score: {c/1e6:.1f} M candidates staged, {ms[1]:.3f} ms, ~{c/(n_tiles*256):.0f} per (tile, pose) wave assuming {n_tiles} tiles x 256 poses")
capi.profile_reset()
capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], bench.I4, 0.10, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
c = capi.profile_read("candidates")[0]
print(f"icp 10 iterations: {c/1e6:.1f} M candidates staged = {c/10/(w['n_scan1']/64):.0f} per tile and iteration")
