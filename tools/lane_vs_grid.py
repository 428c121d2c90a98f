"""Single-problem ICP calls on object-sized sources: the lane chains against the grid chains by source size (us per iteration, wall clock of
the call / iterations, best of 5; ten fixed iterations).  Sources: the first refine unit of bench.py --scaling strong (50 k points) thinned
to the sizes asked for.  python tools/lane_vs_grid.py [sizes...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

if torch.cuda.is_available():
    torch.cuda.init()
import bench  # noqa: E402
from rescan_amd import capi  # noqa: E402

I4 = np.eye(4, dtype=np.float32).ravel()


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [17000, 20000, 25000, 30000, 40000, 50000, 65000]
    capi.init(0)
    w = bench.build_workload(1_000_000, seed=11, knn="hash")
    si = w["strong_icp"]
    p = w["plc"][0]
    pos, nor = p["np"]
    prev = (capi.icp_reference_order_below(0), capi.icp_replay_below(0), capi.icp_lane_chains_below(-1))
    try:
        for n in sizes:
            # (sizes above the unit's 50 k: the unit twice, the copy shifted by a millimetre)
            if n <= len(pos):
                sel = np.sort(np.random.default_rng(n).permutation(len(pos))[:n]); P, N = pos[sel], nor[sel]
            else:
                extra = n - len(pos); P = np.concatenate([pos, pos[:extra] + np.float32(1e-3)]); N = np.concatenate([nor, nor[:extra]])
            c = capi.Cloud(np.ascontiguousarray(P), np.ascontiguousarray(N), cell_size=0.1)
            out = []
            for name, ln in (("lane", 1 << 30), ("grid", 0)):
                capi.icp_lane_chains_below(ln)
                best = 1e9
                for _ in range(5):
                    t = time.perf_counter()
                    e, T, it = capi.icp_align(c, w["scan1"], si["T0s"][0], I4, si["max_dist"], si["max_angle"], max_iter=10, fixed_iters=True)
                    best = min(best, time.perf_counter() - t)
                out.append((name, best * 1e6 / 10, T))
            d = float(np.linalg.norm(out[0][2].astype(np.float64) - out[1][2].astype(np.float64)))
            print(f"n {n:6d}: lane chains {out[0][1]:7.1f} us / iteration | grid chains {out[1][1]:7.1f} | poses {d:.1e} apart", flush=True)
            c.close()
    finally:
        capi.icp_reference_order_below(prev[0]); capi.icp_replay_below(prev[1]); capi.icp_lane_chains_below(prev[2])


if __name__ == "__main__":
    main()
