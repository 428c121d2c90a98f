#!/bin/bash
# bash tools/chain_kernel_times.sh [tag]: rocprofv3 kernel stats of one whole-scan ICP with the centroid chains (tools/chain_profile.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd "$ROOT"
tag=${1:-x}; rm -rf gpurun_out/prof_chain_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain_$tag -- python tools/chain_profile.py > gpurun_out/chain_prof_$tag.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_chain_$tag/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_chain' in r['Name'] or 'k_icp' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us  min {float(r['MinNs'])/1000:8.1f}")
PY
