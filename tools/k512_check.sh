#!/bin/bash
# after a change to the lane chains / the batched searches: the tests that guard them, then the two strong-scaling bench lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py -q -m gpu -x -k "lane or multi or strong or icp_align or stop_test or batch" 2>&1 | tail -4 || exit 1
python bench.py --no-cpu-baseline --no-extras --scaling strong --strong-problems 512 --steps 5 > gpurun_out/k512.json 2>/dev/null || exit 1
python bench.py --no-cpu-baseline --no-extras --scaling strong --steps 10 > gpurun_out/strong1.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("k512", "strong1"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 3), {k: round(x, 3) for k, x in d["kernel_ms_per_step"].items()}, d["parity"]["pose_dist"])
PY
