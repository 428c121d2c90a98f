#!/bin/bash
# bash tools/lane_kernel_times.sh [points] [tag]: rocprofv3 kernel stats of one ICP on a source of `points` points with the lane chains (tools/chain_profile.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd "$ROOT"
n=${1:-160000}; tag=${2:-x}; rm -rf gpurun_out/prof_lane_$tag
export RS_HIP_LANE_CHAINS_BELOW=1000000000
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lane_$tag -- python tools/chain_profile.py $n 1 > gpurun_out/lane_prof_$tag.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_lane_$tag/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_lane' in r['Name'] or 'k_icp' in r['Name'] or 'k_chain' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us  min {float(r['MinNs'])/1000:8.1f}")
PY
