#!/bin/bash
# BASELINE.json configs[2]: the LDS spatial-hash cells against the brute tile at 1 M points per scan, with rocprof HBM GB/s.
#   bash tools/cfg2_profile.sh [points]      (on the GPU box, through gpurun)
# Per layout (--knn hash | brute), one serial step of bench.py: the bench line itself, rocprofv3 --kernel-trace --stats (average
# duration per kernel), and FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE is
# doubled for 16-byte-per-lane reads on gfx950).  Output: gpurun_out/cfg2_<knn>_<points>.json (the bench line) and
# gpurun_out/cfg2_<knn>_<points>.txt (per search kernel: us per launch, HBM-side MB per launch, GB/s, fraction of 8 TB/s).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out
pts=${1:-1000000}
for knn in hash brute; do
  args="--steps 2 --warmup 1 --no-cpu-baseline --serial --knn $knn --points $pts"
  python bench.py $args > gpurun_out/cfg2_${knn}_${pts}.json 2> gpurun_out/cfg2_${knn}_${pts}.err
  rm -rf gpurun_out/cfg2_stats_$knn
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg2_stats_$knn -- python bench.py $args > /dev/null 2> gpurun_out/cfg2_stats_$knn.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/cfg2_${c}_$knn
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/cfg2_${c}_$knn -- python bench.py $args > /dev/null 2> gpurun_out/cfg2_${c}_$knn.err
  done
  python - "$knn" "$pts" <<'PY' > gpurun_out/cfg2_${knn}_${pts}.txt
import csv, glob, json, sys
sys.path.insert(0, "tools")
from pmc_summary import per_kernel
knn, pts = sys.argv[1], sys.argv[2]
line = json.loads(open(f"gpurun_out/cfg2_{knn}_{pts}.json").read().strip().splitlines()[-1])
F = per_kernel(glob.glob(f"gpurun_out/cfg2_FETCH_SIZE_{knn}/*/*counter_collection.csv")[0])
W = per_kernel(glob.glob(f"gpurun_out/cfg2_WRITE_SIZE_{knn}/*/*counter_collection.csv")[0])
dur = {}
for r in csv.DictReader(open(glob.glob(f"gpurun_out/cfg2_stats_{knn}/**/*kernel_stats.csv", recursive=True)[0])):
    dur[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print(f"configs[2], --knn {knn}, {pts} points per scan, one serial step: {line['ms_per_step']:.3f} ms per step, "
      f"{line['value'] / 1e9:.3f} G point-pairs/s; candidate evaluations: {json.dumps(line.get('candidate_evals'))}")
print(f"{'kernel':34s} {'launches':>8s} {'us/launch':>10s} {'fetch MB':>9s} {'write MB':>9s} {'HBM-side MB':>11s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for k in sorted(F):
    if not any(t in k for t in ("k_icp_corr", "k_score", "k_label(", "k_label<", "rs::k_label")):
        continue
    f = F[k]["FETCH_SIZE"][0] * 1024; w = W.get(k, {}).get("WRITE_SIZE", (0.0, 0))[0] * 1024
    us, calls = dur.get(k, (float("nan"), 0))
    hbm = 2 * f + w
    print(f"{k[:34]:34s} {calls:8d} {us:10.1f} {f / 1e6:9.2f} {w / 1e6:9.2f} {hbm / 1e6:11.2f} {hbm / (us * 1e-6) / 1e9:8.1f} {hbm / (us * 1e-6) / 8e12:9.4f}")
PY
  cat gpurun_out/cfg2_${knn}_${pts}.txt
done
