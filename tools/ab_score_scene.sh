#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/sq_ab.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/score_batch_alone.py 10 2>&1 | grep -v "K = " >> $out; }
run RS_HIP_SCORE_SCENE=1
for v in NO_RANK DIST_ONLY NO_SEARCH; do run RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_exp_$v.so; done
cut -c1-150 $out
