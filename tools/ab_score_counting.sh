#!/bin/bash
# the score batch's keys ordered by counting against the radix sort: parity tests, then the batch alone (serial bench) and beside the chain
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "score" 2>&1 | tail -3 || exit 1
for r in 1 2; do
  for cfg in "RS_HIP_SCORE_COUNTING=0" "RS_HIP_SCORE_COUNTING=22"; do
    for mode in "--serial" ""; do
      env $cfg python bench.py --no-cpu-baseline --no-extras --steps 20 $mode 2>/dev/null > /tmp/ab.json
      python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$cfg $mode:', round(d['ms_per_step'],4), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'scores', d['parity']['score_max_abs_err'])"
    done
  done
done
