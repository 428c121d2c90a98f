"""Candidates staged per consumer of the bench step (sharded device counters; each staged candidate is tested by the 64 lanes of
its wave, 16 in the per-row sweeps) and the kernel times next to them."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
def run(name, fn, pairs):
    fn()
    capi.profile_enable(True); capi.profile_reset()
    fn()
    c = capi.profile_read("candidates")[0]
    ms = {k: capi.profile_read(k)[1] for k in ("nn_icp", "icp_moments", "nn_score", "nn_label")}
    print(f"{name}: {c/1e6:.1f} M candidates staged = {c*64/pairs:.0f} evaluations per point pair; kernel ms {dict((k, round(v,3)) for k,v in ms.items() if v)}", flush=True)
    capi.profile_enable(False)
run("score", lambda: capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64), w["pairs"]["score"])
run("icp  ", lambda: capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], bench.I4, 0.10, np.deg2rad(60.0), max_iter=10, fixed_iters=True), w["pairs"]["icp"])
run("label", lambda: capi.arrangement_to_labels(w["scan1"], w["plc_poses"], [p["cloud"] for p in w["plc"]], [0] * len(w["plc"]), [p["cls"] for p in w["plc"]], 0.05, False), w["pairs"]["label"])
