#!/bin/bash
# VALU wave-instructions of every k_icp_corr / k_icp_corr_coop dispatch of one serial bench step, under the
# environment switches given as arguments (e.g. RS_HIP_NO_CERT=1): tools/valu_per_iteration.sh [VAR=1 ...]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/valu_it
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/valu_it -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial > /dev/null 2> gpurun_out/valu_it.err
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/valu_it/*/*counter_collection.csv")[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_icp_corr" not in k: continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": k})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for i, (did, d) in enumerate(rows.items()):
    print(f'{d["k"]:28s} VALU {d.get("SQ_INSTS_VALU",0)/1e6:7.2f} M  SALU {d.get("SQ_INSTS_SALU",0)/1e6:6.2f} M  LDS {d.get("SQ_INSTS_LDS",0)/1e6:6.2f} M')
PY
