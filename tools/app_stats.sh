#!/bin/bash
# Where the unchanged pose_proposal spends its time when it runs on the shim (shadow/icp + shadow/grid): the app's own stage
# timers and the shim's per-route counters (RS_DROPIN_STATS=1).  Run on the GPU box: bash tools/app_stats.sh [density]
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
W=$(mktemp -d); cd "$W"
ROOT="$ROOT" DENS="${1:-6400}" python - <<'PY'
import os, sys
sys.path.insert(0, os.environ["ROOT"])
from rescan_amd import synth
os.makedirs("seq", exist_ok=True)
for t in (0, 1):
    synth.write_ply(f"seq/t{t}.ply", synth.make_scene(seed=100, density=float(os.environ["DENS"]), timestep=t))
synth.write_class_table("classes.rsdb")
PY
"$ROOT/oracle/_ref/seg2rsdb" seq/t0.ply classes.rsdb seq/t0.rsdb > /dev/null 2>&1 || true
for b in pose_proposal_hip2 pose_proposal_hip3; do
  echo "== $b"
  RS_DROPIN_STATS=1 "$ROOT/oracle/_ref/$b" seq/t0.rsdb seq/t1.ply seq/t1_$b.rsdb -v 2> stats_$b.txt | grep -i "computed poses\|processing time\|IO: Done\|Read a scene\|pose proposals made\|Done in\|ICP\|refin\|time" || true
  grep stats stats_$b.txt || true
done
