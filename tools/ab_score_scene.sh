#!/bin/bash
# Score batch alone (256 poses x 10 157 points against the 0.98 M-point scan): the object-space launch, the scene-space route and its switches.
#   bash tools/ab_score_scene.sh        -> gpurun_out/sq_ab.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/sq_ab.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/score_batch_alone.py 10 2>&1 | grep -v "K = " >> $out; }
run RS_HIP_SCORE_SCENE=0
run RS_HIP_SCORE_SCENE=1
run RS_HIP_SCORE_NBIN=0
run RS_HIP_SCORE_CULL=0
run RS_HIP_SCORE_PARENT=2
run RS_HIP_SCORE_PARENT=0.5
run RS_HIP_SCORE_SCENE=1
cut -c1-170 $out
