"""The drop-in claim at application level: the reference's own apps/pose_proposal, compiled
UNMODIFIED from /root/reference with shadow/icp first on the include path and linked against
librescan_dropin.so (oracle/Makefile: _ref/pose_proposal_hip), against the same sources built
the reference's way (_ref/pose_proposal), on a synthetic 2-timestep scene (BASELINE.json
configs[0]).  The binaries are built in the container that has /root/reference and travel to
the GPU box as prebuilt files; the test skips where they are absent.

Round 5: apps/segment_transfer likewise — minus its graph-cut smoothing.  gco-v3.0 is not vendored and no stand-in for it is written;
oracle/Makefile builds the reference's own text minus three lines (two gco includes and main.cpp's one call of rspf_smooth_labels),
the reference's way and against the shim: the app's three call sites on the hot path (the refine of the optimised poses, the label
loops, the database update's icp include) reached THROUGH THE APP.  Nothing here says anything about rspf_smooth_labels."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref")
BINS = [os.path.join(REF, b) for b in ("seg2rsdb", "pose_proposal", "pose_proposal_hip", "pose_proposal_hip2", "pose_proposal_hip3")]

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not all(os.path.exists(b) for b in BINS), reason="oracle/_ref apps not built")]


def read_pose_bin(path):
    """apps/pose_proposal/main.cpp:61-89: int32 n_obj; int32 count[n_obj]; {float[16] pose; float score} x sum(count)."""
    b = open(path, "rb").read()
    n = struct.unpack("<i", b[:4])[0]
    counts = struct.unpack("<%di" % n, b[4:4 + 4 * n])
    off = 4 + 4 * n
    out = []
    for c in counts:
        rec = np.frombuffer(b[off:off + 68 * c], np.float32).reshape(c, 17)
        off += 68 * c
        out.append(rec)
    return out


def computed_in(stdout):
    """The app's own timer: "Computed poses in ..." (apps/pose_proposal/main.cpp:208)."""
    import re
    m = re.search(r"Computed poses in\s*([0-9.]+)", stdout)
    return float(m.group(1)) if m else float("nan")


@pytest.mark.parametrize("which", ["icp", "icp+grid", "icp+grid+batched"])
def test_pose_proposal_app_links_against_the_shim(tmp_path, which):
    """which = "icp": shadow/icp only (icp_align on the GPU); "icp+grid": shadow/icp + shadow/grid — also the app's own
    mgs_compute_object_alignment_score (83 % of its run time), its level builder and its level grids run on
    the shim's msh_hash_grid_*; "icp+grid+batched": in addition the app's grid search (mgs_propose_poses) is the batched driver
    shadow/apps/pose_proposal_batched.cpp: one rsd_alignment_scores (k_score) per object and level instead of one small search
    per pose — the proposal file must still be the reference build's."""
    from rescan_amd import synth
    shim_bin = {"icp": BINS[2], "icp+grid": BINS[3], "icp+grid+batched": BINS[4]}[which]
    seq = tmp_path / "seq"
    seq.mkdir()
    for t in (0, 1):
        synth.write_ply(str(seq / f"t{t}.ply"), synth.make_scene(seed=7, density=2000.0, timestep=t))
    synth.write_class_table(str(tmp_path / "classes.rsdb"))
    run = lambda *a: subprocess.run(list(a), cwd=str(tmp_path), capture_output=True, text=True, timeout=900)  # noqa: E731
    r = run(BINS[0], "seq/t0.ply", "classes.rsdb", "seq/t0.rsdb", "-v")
    assert os.path.exists(seq / "t0.rsdb"), r.stdout[-500:] + r.stderr[-500:]      # (its exit code is unreliable)
    cpu = run(BINS[1], "seq/t0.rsdb", "seq/t1.ply", "seq/t1_cpu.rsdb", "-v")
    assert cpu.returncode == 0, cpu.stdout[-800:]
    hip = run(shim_bin, "seq/t0.rsdb", "seq/t1.ply", "seq/t1_hip.rsdb", "-v")
    print("Computed poses in: reference build %.3f s, shim build (%s) %.3f s" % (computed_in(cpu.stdout), which, computed_in(hip.stdout)))
    assert hip.returncode == 0, hip.stdout[-800:] + hip.stderr[-800:]
    assert "[rescan_hip]" not in hip.stderr, hip.stderr[-800:]                      # the shim reported no failure
    if which == "icp+grid+batched":
        assert "scored in one batch" in hip.stdout
    a = read_pose_bin(str(seq / "t1_cpu" / "t1_cpu.bin"))
    b = read_pose_bin(str(seq / "t1_hip" / "t1_hip.bin"))
    assert len(a) == len(b)
    n_dyn = 0
    worst = []
    for pa, pb in zip(a, b):
        assert len(pa) == len(pb), "different number of surviving proposals"
        # Proposals are sorted by score; compare pose by pose.  The app refines on level-2 clouds (2 cm voxels, a few
        # hundred points per chair): object-sized sources, for which the library runs the estimator in the reference's
        # own accumulation order with libm's sinf/cosf, so the refined poses are the reference's bit for bit (26 of the
        # 27 proposals of this scene; the 27th, a weak one, is 7e-7 away — an exact distance tie, which the reference
        # decides by its grid traversal order).  Asserted: good proposals (score > 0.9) at most 1e-4
        # apart with a median of exactly 0; weak ones within 1e-2.
        for ra, rb in zip(pa, pb):
            dpose = np.linalg.norm(ra[:16].astype(np.float64) - rb[:16])
            worst.append((float(ra[16]), dpose))
            assert dpose < (1e-4 if ra[16] > 0.9 else 1e-2), (ra[16], dpose)
            assert abs(float(ra[16]) - float(rb[16])) < (1e-3 if ra[16] > 0.9 else 1e-2)
        n_dyn += len(pa) > 1
    assert n_dyn >= 3                                   # table + two chairs were refined by icp_align
    good = [d for sc, d in worst if 0.9 < sc < 9.0]
    assert len(good) >= 3 and np.median(good) == 0.0
    assert np.mean([d == 0.0 for _, d in worst]) > 0.9
    print("pose deltas: good proposals median %.2e max %.2e, all max %.2e over %d proposals"
          % (np.median(good), max(good), max(d for _, d in worst), len(worst)))
    print("nonzero:", [(round(sc, 4), "%.2e" % d) for sc, d in worst if d > 0])


ST_BINS = [os.path.join(REF, b) for b in ("segment_transfer_nosmooth", "segment_transfer_nosmooth_hip2")]


def read_prediction_ply(path):
    """The segmented cloud segment_transfer writes (rs_pointcloud__save_ply: binary little-endian vertices); returns the header's
    property names and the vertex block as bytes per property."""
    b = open(path, "rb").read()
    end = b.index(b"end_header\n") + len(b"end_header\n")
    hdr = b[:end].decode()
    n = int([l for l in hdr.splitlines() if l.startswith("element vertex")][0].split()[-1])
    props = [l.split()[1:] for l in hdr.splitlines() if l.startswith("property") and "list" not in l]
    np_t = {"float": "<f4", "uchar": "u1", "int": "<i4", "uint": "<u4", "double": "<f8", "short": "<i2", "ushort": "<u2", "char": "i1"}
    dt = np.dtype([(name, np_t[t]) for t, name in props])
    return np.frombuffer(b[end:end + n * dt.itemsize], dt)


@pytest.mark.skipif(not all(os.path.exists(b) for b in ST_BINS), reason="oracle/_ref segment_transfer builds absent")
def test_segment_transfer_app_links_against_the_shim(tmp_path):
    """seg2rsdb -> pose_proposal (reference build) -> segment_transfer minus its smoothing, reference build against shim build
    (shadow/icp + shadow/grid) on the same proposal file: the poses of the optimised and REFINED arrangement in the output .rsdb
    (rsdb_refine_alignment_of_objects_to_scene -> icp_align, lib/rs/rs_database.h:216-232) and the per-vertex class / instance ids
    of the segmented cloud (rspf_arrangement_to_labels + the wall / floor relabelling, before any smoothing) must be the reference build's."""
    import shutil
    from rescan_amd import synth
    seq = tmp_path / "seq"
    seq.mkdir()
    for t in (0, 1):
        synth.write_ply(str(seq / f"t{t}.ply"), synth.make_scene(seed=7, density=2000.0, timestep=t))
    synth.write_class_table(str(tmp_path / "classes.rsdb"))
    run = lambda *a: subprocess.run(list(a), cwd=str(tmp_path), capture_output=True, text=True, timeout=900)  # noqa: E731
    run(BINS[0], "seq/t0.ply", "classes.rsdb", "seq/t0.rsdb", "-v")
    assert os.path.exists(seq / "t0.rsdb")
    pp = run(BINS[1], "seq/t0.rsdb", "seq/t1.ply", "seq/t1_pp.rsdb", "-v")
    assert pp.returncode == 0, pp.stdout[-800:]
    outs = {}
    for tag, b in zip(("cpu", "hip"), ST_BINS):
        r = run(b, "seq/t1_pp.rsdb", "-o", f"seq/t1_{tag}.rsdb", "-v")
        assert r.returncode == 0, r.stdout[-1200:] + r.stderr[-800:]
        if tag == "hip":
            assert "[rescan_hip]" not in r.stderr, r.stderr[-800:]
        import re
        stages = dict(re.findall(r"(Refining optimized poses done|Segmentation finished|Optimization finished) in ([0-9.]+)s", r.stdout))
        print(tag, {k: float(v) for k, v in stages.items()})
        rsdb = open(seq / f"t1_{tag}.rsdb").read()
        poses = [l for l in rsdb.splitlines() if l.strip().startswith("pose")]
        pred = read_prediction_ply(str(seq / "predictions" / f"t1_{tag}.ply"))
        outs[tag] = (poses, pred)
        shutil.rmtree(seq / f"t1_{tag}", ignore_errors=True)
    (pa, ca), (pb, cb) = outs["cpu"], outs["hip"]
    assert len(pa) == len(pb) and len(pa) >= 6
    assert pa == pb, [x for x in zip(pa, pb) if x[0] != x[1]][:3]                   # every pose line of the .rsdb, character for character
    names = ca.dtype.names
    ids = [n for n in names if "class" in n or "instance" in n]
    assert len(ids) >= 2, names
    for n in ids:
        assert (ca[n] == cb[n]).all(), (n, int((ca[n] != cb[n]).sum()))
    assert len(np.unique(ca[ids[0]])) >= 3                                          # (floor, walls, furniture: the labels are not trivial)
