"""Score batch alone (bench workload: 256 poses x 10 157 points against the 0.98 M-point scan): kernel time, candidates staged and
parity with tests/golden/bench_seed11.npz for the library selected by RS_HIP_LIB and the switches in the environment
(e.g. RS_HIP_SCORE_SCENE=0, RS_HIP_SCORE_CULL=0, RS_HIP_SCORE_NBIN=0, RS_HIP_SCORE_PARENT=2; with the diagnostic build — tools/variant.sh dbg -DRS_DBG=1 — RS_HIP_SCORE_HIST=1 prints where the candidates are streamed: by shell and number of unsettled lanes, and by the rank pass).
usage: python tools/score_batch_alone.py [repeats]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rescan_amd import capi
capi.init(0)
w = bench.build_workload(1_000_000, seed=11, knn="hash")
g = np.load(os.path.join(ROOT, "tests", "golden", "bench_seed11.npz"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64)
capi.profile_enable(True); capi.profile_reset()
for _ in range(reps):
    sc = capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64)
c = capi.profile_read("candidates")[0] / reps; n, ms = capi.profile_read("nn_score")
n_tiles = -(-w["n_obj"] // 64)
same = int((sc == g["scores"]).sum()); err = float(np.abs(sc.astype(np.float64) - g["scores"]).max())
print(f"lib {os.environ.get('RS_HIP_LIB', 'default')}: {ms / n:.3f} ms per batch, {c / 1e6:.1f} M candidates staged "
      f"(~{c / (n_tiles * 256):.0f} per (tile, pose) wave); scores bit-identical to the reference's: {same} of {len(sc)}, max abs err {err:.2e}")
# K = 32 (main.cpp:199's refine scoring) and another radius/K against the one-radius search of the same library
for K in (32, 16):
    a = capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"][:64], 0.1, K)
    print(f"   K = {K}: first scores {a[:3].tolist()}")
