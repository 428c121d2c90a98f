#!/bin/bash
# The GPU parity suite under every performance switch: they must not change a single result.
#   bash tools/switch_matrix.sh [first [last]]     (a slice of the list: the whole matrix takes ~half an hour)
cd "$(dirname "$0")/.."
first=${1:-0}; last=${2:-999}; k=-1
for e in "RS_X=1" "RS_HIP_NO_WARM=1" "RS_HIP_NO_CERT=1" "RS_HIP_NO_RANK_CERT=1" "RS_HIP_NO_BY_ROWS=1" "RS_HIP_NO_SEED=1" "RS_HIP_NO_LPT=1" "RS_HIP_SOLO_STAGES=64" "RS_HIP_SOLO_STAGES=100000" "RS_HIP_HANDOFF_K=1" "RS_HIP_ICP_CHUNK=1" "RS_HIP_ICP_CHUNK=7" "RS_HIP_COOP_ALL_BELOW=0" "RS_HIP_COOP_ALL_BELOW=0 RS_HIP_NO_WARM=1" "RS_HIP_COOP_ALL_BELOW=0 RS_HIP_NO_CERT=1" "RS_HIP_COOP_ALL_BELOW=100000000" "RS_HIP_COOP_WAVES=4" "RS_HIP_REF_ORDER_BELOW=0" "RS_HIP_REF_ORDER_BELOW=100000000" "RS_HIP_NO_BOUNDED_ONLY=1" "RS_HIP_HEAVY_TOTAL=0 RS_HIP_HEAVY_HANDOFF=100000000 RS_HIP_HEAVY_STREAMED=100000000" "RS_HIP_HEAVY_TOTAL=64" "RS_HIP_SCORE_SCENE=0" "RS_HIP_SCORE_SCENE_MIN=0" "RS_HIP_SCORE_SCENE_MIN=0 RS_HIP_SCORE_CULL=0" "RS_HIP_SCORE_SCENE_MIN=0 RS_HIP_SCORE_NBIN=0" "RS_HIP_SCORE_SCENE_MIN=0 RS_HIP_SCORE_PARENT=2" "RS_HIP_REF_ORDER_BELOW=0 RS_HIP_REPLAY_BELOW=100000000" "RS_HIP_NO_ROWS_WAVE=1" "RS_HIP_ROWS_ZERO_COPY_BELOW=0" "RS_HIP_ROWS_TILED_FROM=0" "RS_DROPIN_HOST_QUERIES=0" "RS_DROPIN_FULL_HASH=1" "RS_HIP_ICP_BATCH_BYTES=1" "RS_HIP_FAITH_GUESS=0" "RS_HIP_FAITH_GUESS=500"; do
  k=$((k+1)); if [ $k -lt $first ] || [ $k -gt $last ]; then continue; fi
  printf "%-60s " "$e"; env $e python -m pytest tests/test_gpu_parity.py tests/test_dropin.py tests/test_gpu_headline.py -x -q -m gpu 2>&1 | tail -1
done
