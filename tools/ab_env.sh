#!/bin/bash
# bench (serial + concurrent) under different environment settings: tools/ab_env.sh "A=1" "A=2 B=3" ...
cd "$(dirname "$0")/.."
one() { env $1 python bench.py --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', '$2', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
for e in "$@"; do one "$e" --serial; one "$e" ""; done
