#!/bin/bash
# bash tools/replay2_kernel_times.sh [centred]: rocprofv3 kernel stats of one whole-scan ICP whose centroid sums go through pass 2 of the replay
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd "$ROOT"
tag=${1:-octant}; rm -rf gpurun_out/prof_replay2_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_replay2_$tag -- python tools/replay2_profile.py $1 > gpurun_out/replay2_prof_$tag.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_replay2_$tag/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if 'k_build' in r['Name'] or 'hipcub' in r['Name'] or 'rocprim' in r['Name']: continue
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
