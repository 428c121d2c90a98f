"""RS_HIP_CHAIN_DEBUG=1 python tools/chain_debug_bench.py [centre] [iters=N]: the centroid chains' walks on bench.py's own ICP problem."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rescan_amd import capi
capi.init(0)
iters = max([int(a[6:]) for a in sys.argv[1:] if a.startswith("iters=")] or [4])
w = bench.build_workload(1_000_000, seed=11, knn="hash", centre="centre" in sys.argv[1:])
print(capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], bench.I4, 0.10, np.deg2rad(60.0), max_iter=iters, fixed_iters=True), flush=True)
