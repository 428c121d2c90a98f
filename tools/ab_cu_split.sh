#!/bin/bash
# bash tools/ab_cu_split.sh: the bench's CU partition measured again (interleaved, 2 rounds): fraction of the CU mask for the ICP chain x where the label pass runs
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for r in 1 2; do
  for cfg in "0.75 chain" "0.75 batch" "0.625 chain" "0.875 chain" "0.75 all" "0 chain"; do
    set -- $cfg
    RS_BENCH_CU_SPLIT=$1 RS_BENCH_LABEL_ON=$2 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('split $1 label on $2:', round(d['ms_per_step'],4), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
  done
done
