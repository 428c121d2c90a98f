#!/bin/bash
# Step time under different CU partitions / label placements (interleaved, two rounds): bash tools/ab_cu_split.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/ab_cu_split.txt; : > $out
one() { r=$(env "$@" python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'median', round(d['ms_per_step_spread']['median'],3), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"); echo "$* -> $r" >> $out; }
for rep in 1 2; do
one RS_BENCH_CU_SPLIT=0.75 RS_BENCH_LABEL_ON=chain
one RS_BENCH_CU_SPLIT=1.0 RS_BENCH_CU_BATCH_LO=0.75 RS_BENCH_LABEL_ON=batch
one RS_BENCH_CU_SPLIT=1.0 RS_BENCH_CU_BATCH_LO=0.75 RS_BENCH_LABEL_ON=chain
one RS_BENCH_CU_SPLIT=0.875 RS_BENCH_CU_BATCH_LO=0.75 RS_BENCH_LABEL_ON=batch
done
cat $out
