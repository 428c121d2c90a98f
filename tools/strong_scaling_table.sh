#!/bin/bash
# Per-rank compute time of bench.py --scaling strong at worlds 1 / 2 / 4 / 8, simulated on ONE device (rank 0's share, no exchange):
#   bash tools/strong_scaling_table.sh [problems ...]      -> gpurun_out/strong_scaling_simulated_worlds.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/strong_scaling_simulated_worlds.txt
echo "bench.py --scaling strong --strong-problems K, RS_BENCH_SIM_WORLD=W: rank 0's share of a W-rank world on one MI355X, ms per step (compute of the three consumers; no exchange)" > $out
for K in ${@:-8 64 512}; do
  for W in 1 2 4 8; do
    RS_BENCH_SIM_WORLD=$W python bench.py --scaling strong --strong-problems $K --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('problems %4d  world %d: step %7.3f ms  rank-0 compute %7.3f ms  kernels %s' % ($K, $W, d['ms_per_step'], c['rank0_compute_ms_per_step'], {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()}))" >> $out
  done
done
cat $out
