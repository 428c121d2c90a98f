"""bench.py's synthetic inputs (no device): --centre moves the whole world — scans, object poses and everything drawn from them — so
that the scan's median point is the origin, and changes nothing else."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_centre_moves_the_whole_world():
    import bench
    a = bench.build_inputs(20_000, 11)
    b = bench.build_inputs(20_000, 11, centre=True)
    sh = -np.median(a["s1"]["points"], axis=0).astype(np.float32)
    assert np.abs(np.median(b["s1"]["points"], axis=0)).max() < 1e-6
    assert (b["s0"]["points"] == a["s0"]["points"] + sh).all() and (b["s1"]["points"] == a["s1"]["points"] + sh).all()
    assert (b["s1"]["normals"] == a["s1"]["normals"]).all()
    # coordinates of both signs on every axis that the room spans
    assert (b["s1"]["points"].min(axis=0) < 0).all() and (b["s1"]["points"].max(axis=0) > 0).all()
    # poses: the same rotations, translations moved with the world (up to the float rounding of drawing the perturbation around another origin)
    for k in ("score_poses", "plc_poses"):
        pa, pb = a[k].reshape(-1, 16), b[k].reshape(-1, 16)
        assert pa.shape == pb.shape
        assert np.abs(pb[:, :12] - pa[:, :12]).max() < 1e-5
        assert np.abs(pb[:, 12:15] - (pa[:, 12:15] + sh)).max() < 0.5       # (a perturbation's rotation acts about the object's own position: close to the moved pose)
    assert a["pairs"] == b["pairs"] and a["n_scan1"] == b["n_scan1"]
