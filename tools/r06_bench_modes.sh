#!/bin/bash
# Round 6, VERDICT r05 item 3 — the bench lines that let an N = 8 number be split into join mechanism, exchange and compute, and the
# rehearsals of the N > 1 path on a one-GPU box: bash tools/r06_bench_modes.sh   (-> gpurun_out/r06_*.json)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { name=$1; shift; "$@" > gpurun_out/r06_$name.json 2> gpurun_out/r06_$name.err || echo "$name: exit code $?"; }
run default_n1            python bench.py --no-cpu-baseline
RS_BENCH_SPIN=0 run n1_spin0          python bench.py --no-cpu-baseline
RS_BENCH_FORCE_DIST=1 run n1_force_dist_rccl  python bench.py --no-cpu-baseline
run strong_n1             python bench.py --no-cpu-baseline --scaling strong --steps 10
run strong_k512           python bench.py --no-cpu-baseline --scaling strong --strong-problems 512 --steps 5
run centre_n1             python bench.py --no-cpu-baseline --centre
run timesteps4_n1         python bench.py --no-cpu-baseline --timesteps 4 --steps 10
RS_BENCH_ONE_DEVICE=1 run gpus2_one_device_gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu-baseline
RS_BENCH_ONE_DEVICE=1 run gpus4_one_device_gloo timeout -k 10 400 python bench.py --gpus 4 --steps 5 --no-cpu-baseline
# the failure path: rank 1 exits hard in warm-up step 2 while rank 0 has an exchange in flight -> the launcher ends rank 0, exit code 7, no hang
RS_BENCH_ONE_DEVICE=1 RS_BENCH_DIE=1:2 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu-baseline > gpurun_out/r06_gpus2_rank_dies.json 2> gpurun_out/r06_gpus2_rank_dies.err
echo "rank dies mid-run: launcher exit code $? (expected 7)" | tee gpurun_out/r06_gpus2_rank_dies.txt
grep "launcher" gpurun_out/r06_gpus2_rank_dies.err | tee -a gpurun_out/r06_gpus2_rank_dies.txt
for f in default_n1 n1_spin0 n1_force_dist_rccl strong_n1 strong_k512 centre_n1 timesteps4_n1 gpus2_one_device_gloo gpus4_one_device_gloo; do
  echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r06_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d.get("kernel_ms_per_step"))
    print("   parity:", d["parity"])
except Exception as e:
    print("unreadable:", e)
PY
done
