"""TEST INFRASTRUCTURE.  tests/golden/formats/: the two on-disk formats either side of the hot path
(SURVEY §8f row 4), written by the REFERENCE's own binaries (oracle/_ref/seg2rsdb, oracle/_ref/pose_proposal,
compiled from /root/reference in place) on a small synthetic 2-timestep scene:

  t1.bin     pose proposals (apps/pose_proposal/main.cpp:61-89)
  t1.rsdb    the database text with its `pose` lines (lib/rs/rs_database.h:590-606)

Usage: python oracle/gen_golden_formats.py     (build container only; needs oracle/_ref)"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rescan_amd import synth  # noqa: E402
from test_app_dropin import write_ply  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "formats")


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    seq = os.path.join(tmp, "seq")
    os.makedirs(seq)
    for t in (0, 1):
        write_ply(os.path.join(seq, f"t{t}.ply"), synth.make_scene(seed=7, density=900.0, timestep=t))
    with open(os.path.join(tmp, "classes.rsdb"), "w") as f:
        f.write("rsdb 0.1\n")
        for k, v in synth.CLASS_IDX.items():
            f.write(f"class {k} {v}\n")
    run = lambda *a: subprocess.run(list(a), cwd=tmp, capture_output=True, text=True, timeout=1800)  # noqa: E731
    run(os.path.join(REF, "seg2rsdb"), "seq/t0.ply", "classes.rsdb", "seq/t0.rsdb", "-v")
    r = run(os.path.join(REF, "pose_proposal"), "seq/t0.rsdb", "seq/t1.ply", "seq/t1.rsdb", "-v")
    assert r.returncode == 0, r.stdout[-1000:]
    shutil.copy(os.path.join(seq, "t1", "t1.bin"), os.path.join(OUT, "t1.bin"))
    shutil.copy(os.path.join(seq, "t1.rsdb"), os.path.join(OUT, "t1.rsdb"))
    shutil.copy(os.path.join(seq, "t0.rsdb"), os.path.join(OUT, "t0.rsdb"))
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)})
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
