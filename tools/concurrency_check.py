"""The three consumers issued from three host threads must return exactly what they return one after another."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 11, "hash")
ref = bench.run_step(w, concurrent=False)
bad = 0
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    for mode in (True, False):
        got = bench.run_step(w, concurrent=mode)
        same = (np.float32(got[0]) == np.float32(ref[0]) and (np.asarray(got[1]) == np.asarray(ref[1])).all()
                and (got[2] == ref[2]).all() and (got[3]["labels"] == ref[3]["labels"]).all() and (got[3]["min_dists"] == ref[3]["min_dists"]).all())
        bad += 0 if same else 1
        if not same:
            what = []
            if np.float32(got[0]) != np.float32(ref[0]): what.append("icp err")
            if not (np.asarray(got[1]) == np.asarray(ref[1])).all(): what.append("icp pose (max abs diff %.3g)" % np.abs(np.asarray(got[1], np.float64) - np.asarray(ref[1])).max())
            if not (got[2] == ref[2]).all(): what.append("scores (%d of %d, max abs diff %.3g)" % ((got[2] != ref[2]).sum(), len(ref[2]), np.abs(got[2].astype(np.float64) - ref[2]).max()))
            if not (got[3]["labels"] == ref[3]["labels"]).all(): what.append("labels (%d)" % (got[3]["labels"] != ref[3]["labels"]).sum())
            if not (got[3]["min_dists"] == ref[3]["min_dists"]).all(): what.append("min_dists (%d)" % (got[3]["min_dists"] != ref[3]["min_dists"]).sum())
            print("step", k, "concurrent" if mode else "serial", "differs:", ", ".join(what))
print("mismatching steps:", bad)
sys.exit(1 if bad else 0)
