"""Per-iteration queue / match / certificate counts of the bench's ICP problem (RS_HIP_DEBUG=1 set here)."""
import os, sys
os.environ.setdefault("RS_HIP_DEBUG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], bench.I4, 0.10, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
