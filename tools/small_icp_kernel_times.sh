#!/bin/bash
# bash tools/small_icp_kernel_times.sh [points]: rocprofv3 kernel stats of icp_align on object-sized clouds (tools/small_icp.py)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd "$ROOT"
n=${1:-50000}; rm -rf gpurun_out/prof_small_icp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_icp -- python tools/small_icp.py $n > gpurun_out/small_icp_prof.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_small_icp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_icp' in r['Name'] or 'k_replay' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  min {float(r['MinNs'])/1000:8.1f}  max {float(r['MaxNs'])/1000:8.1f}")
PY
