"""CPU: the HOST logic of the batched grid-search driver (shadow/apps/pose_proposal_batched.cpp) against the
reference app.  oracle/_ref/pose_proposal_batched_cpu is the reference's unchanged apps/pose_proposal with its
mgs_propose_poses bound to that driver and rsd_alignment_scores answered by the reference's own score function
(oracle/ref_scores_stub.cpp): whatever differs from oracle/_ref/pose_proposal's proposal file is the driver's enumeration,
per-cell best rotation, thresholds or survivor bookkeeping.  (The same driver on the GPU: tests/test_app_dropin.py.)"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref")
BINS = [os.path.join(REF, b) for b in ("seg2rsdb", "pose_proposal", "pose_proposal_batched_cpu")]

pytestmark = pytest.mark.skipif(not all(os.path.exists(b) for b in BINS), reason="oracle/_ref apps not built")


def test_batched_driver_host_logic_is_the_reference_loop(tmp_path):
    from rescan_amd import synth
    seq = tmp_path / "seq"
    seq.mkdir()
    for t in (0, 1):
        synth.write_ply(str(seq / f"t{t}.ply"), synth.make_scene(seed=9, density=700.0, timestep=t))
    synth.write_class_table(str(tmp_path / "classes.rsdb"))
    run = lambda *a: subprocess.run(list(a), cwd=str(tmp_path), capture_output=True, text=True, timeout=900)  # noqa: E731
    r = run(BINS[0], "seq/t0.ply", "classes.rsdb", "seq/t0.rsdb", "-v")
    assert os.path.exists(seq / "t0.rsdb"), r.stdout[-500:] + r.stderr[-500:]      # (its exit code is unreliable)
    a = run(BINS[1], "seq/t0.rsdb", "seq/t1.ply", "seq/t1_ref.rsdb", "-v")
    assert a.returncode == 0, a.stdout[-800:]
    b = run(BINS[2], "seq/t0.rsdb", "seq/t1.ply", "seq/t1_bat.rsdb", "-v")
    assert b.returncode == 0, b.stdout[-800:] + b.stderr[-800:]
    assert "(batched)" in b.stdout and "scored in one batch" in b.stdout
    fa, fb = (seq / "t1_ref" / "t1_ref.bin").read_bytes(), (seq / "t1_bat" / "t1_bat.bin").read_bytes()
    n = np.frombuffer(fa[:4], np.int32)[0]
    counts = np.frombuffer(fa[4:4 + 4 * n], np.int32)
    assert counts.sum() >= 3, "the scene must produce proposals for the comparison to mean anything"
    assert fa == fb, "the batched driver's proposals differ from the reference app's"
