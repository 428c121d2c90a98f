#!/bin/bash
# The bench's modes added in round 5, on a one-GPU box: bash tools/r05_bench_modes.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || echo "default failed"
python bench.py --no-cpu-baseline --timesteps 4 --steps 10 > gpurun_out/bench_timesteps4.json 2> gpurun_out/bench_timesteps4.err || echo "timesteps failed"
RS_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --no-cpu-baseline > gpurun_out/bench_gpus2_launcher_one_device_gloo.json 2> gpurun_out/bench_gpus2_launcher.err || echo "gpus2 failed"
RS_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --timesteps 4 --no-cpu-baseline > gpurun_out/bench_gpus2_timesteps4_one_device_gloo.json 2> gpurun_out/bench_gpus2_timesteps4.err || echo "gpus2 timesteps failed"
for f in default timesteps4 gpus2_launcher_one_device_gloo gpus2_timesteps4_one_device_gloo; do
  echo "== $f"; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"]["timesteps"], d["config"]["pairs_per_step"], d["parity"], d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("unreadable:", e)
PY
done
tail -3 gpurun_out/bench_gpus2_launcher.err
