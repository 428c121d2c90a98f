"""The seed sweeps of tests/stress/ as GPU tests, at sizes that keep them short (the scripts themselves take more seeds:
tests/stress/README.md).  Each runs as a child process — the scripts are sweeps with an exit code, not importable cases — and
must end with exit code 0: fresh scenes against the oracle (correspondences, ICP, scores, labels), the same with the whole world
moved so that coordinates have both signs, the hunt for ICP runs that are not the reference's bits (every one of them must be
explained by an exact distance tie), and the pose margin by source size."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
STRESS = os.path.join(ROOT, "tests", "stress")


def run(script, *args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(STRESS, script), *[str(a) for a in args]], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout


def test_stress_parity_fresh_scenes():
    out = run("stress_parity.py", 1, 4)
    assert "mismatches: 0" in out


def test_stress_parity_coordinates_of_both_signs():
    out = run("stress_parity.py", 4, 6, env={"STRESS_SHIFT": "centre"})
    assert "mismatches: 0" in out


def test_tie_hunt_every_difference_is_an_exact_distance_tie():
    """54 ICP runs of voxel-averaged level-2 objects against a level-2 scan (seeds 1-3).  Seed 3 holds the one run in 288 that is NOT
    the reference's bits (ties): an exact fp32 distance tie, which the reference decides by the cell size its grid was
    built with — it must be there, be explained as such (the GPU result equals a replay of the reference's loop on per-radius
    grids), stay within north_star's tolerance, and be the only one."""
    import re
    out = run("tie_hunt.py", 3)
    assert "unexplained 0" in out and "differing correspondences 0" in out
    ties = re.findall(r"dT ([0-9.e+-]+): GPU == replay with per-radius grids", out)
    assert len(ties) == 1 and 0.0 < float(ties[0]) < 1e-4, ties


def test_icp_margin_by_source_size():
    run("icp_margin.py", 2)
