#!/bin/bash
# Per-kernel times and SQ instruction counters of the score batch alone (scene-space route): bash tools/score_scene_profile.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out; out=gpurun_out/sq_profile.txt; : > $out
for kv in "$@"; do export "$kv"; done
echo "== switches: $*" >> $out
rm -rf gpurun_out/prof_sq gpurun_out/prof_sq_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sq -- python tools/score_batch_alone.py 10 > /dev/null 2> gpurun_out/prof_sq.err
python - >> $out <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_sq/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 256-pose batch: from its k_score_keys (grid.y = 256) to its k_score_gather
starts = [i for i, r in enumerate(rows) if "k_score_keys" in r["Kernel_Name"] and r["Grid_Size_Y"] == "256"]
if starts:
    i = starts[-1]; t0 = int(rows[i]["Start_Timestamp"]); agg = collections.OrderedDict()
    for r in rows[i:]:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += d
        if "k_score_gather" in n:
            print(f"one 256-pose batch under the profiler: {(int(r['End_Timestamp']) - t0) / 1e3:.1f} us from the first kernel's start to the last one's end")
            break
    for n, (c, d) in agg.items():
        print(f"  {d:8.1f} us  x{c:<3d} {n}")
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_sq_pmc -- python tools/score_batch_alone.py 2 > /dev/null 2> gpurun_out/prof_sq_pmc.err
python - >> $out <<'PY'
import glob, sys
sys.path.insert(0, "tools")
from pmc_summary import per_kernel
pk = per_kernel(glob.glob("gpurun_out/prof_sq_pmc/*/*counter_collection.csv")[0])
for k, d in pk.items():
    if "k_score" in k:
        print(k[:60], {c: round(v[0] / 1e6, 2) for c, v in sorted(d.items())}, "(millions per launch, averaged over", int(max(v[1] for v in d.values())), "launches incl. the small K = 32 / 16 batches)")
PY
cat $out
