cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > /dev/null 2> gpurun_out/tr.err
f=$(find gpurun_out/tr -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
rm -rf gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial > /dev/null 2> gpurun_out/pmc2.err
f=$(find gpurun_out/pmc2 -name "*counter_collection.csv" | head -1)
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:28]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,v in agg.items():
    if "icp_corr" in k or "score" in k or "label" in k:
        print(k, cnt[k], {a:round(b/max(1,cnt[k])/1e6,2) for a,b in v.items()})
PY
