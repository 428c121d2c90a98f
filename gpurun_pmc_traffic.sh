cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial > /dev/null 2> gpurun_out/pmc_$c.err
done
python - <<'PY'
import csv,collections,glob,json
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]; agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    out[c]={k:(agg[k]/cnt[k],cnt[k]) for k in agg}
for k in sorted(out["FETCH_SIZE"]):
    print(k, "launches",out["FETCH_SIZE"][k][1], "FETCH_SIZE avg KB", round(out["FETCH_SIZE"][k][0],1), "WRITE_SIZE avg KB", round(out["WRITE_SIZE"].get(k,(0,0))[0],1))
json.dump(out, open("gpurun_out/pmc_traffic_raw.json","w"), indent=1)
PY
