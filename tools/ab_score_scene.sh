#!/bin/bash
# Score batch alone under the scene-space route's switches: bash tools/ab_score_scene.sh
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/sq_ab.txt; : > $out
run() { echo "== $*" >> $out; env "$@" python tools/score_batch_alone.py 10 2>&1 | grep -v "K = " >> $out; }
hist() { echo "== hist $*" >> $out; env RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_dbg.so RS_HIP_SCORE_HIST=1 "$@" python tools/score_batch_alone.py 1 2>&1 | grep -v "K = " | tail -10 >> $out; }
run RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_nco.so
run RS_HIP_SCORE_NBIN=2
run RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_nco.so
run RS_HIP_SCORE_NBIN=2
hist RS_HIP_SCORE_NBIN=2
cut -c1-170 $out
