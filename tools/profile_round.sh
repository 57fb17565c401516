#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command + HBM traffic PMC passes (separate runs,
# counters only).  Run on the GPU box:  bash tools/profile_round.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $OUT/bench_stats_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/bench_pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/bench_pmc_write_stdout.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<PY
import csv, glob, json, collections, os
out="$OUT"
def agg(path, counter):
    tot=collections.defaultdict(float); n=collections.Counter(); seen=set()
    for fn in glob.glob(path+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"]!=counter: continue
            k=row["Kernel_Name"].split("(")[0][:60]
            tot[k]+=float(row["Counter_Value"]); n[k]+=1
    return {k:(tot[k]/n[k], n[k]) for k in tot}
fetch=agg(out+"/pmc_fetch","FETCH_SIZE"); write=agg(out+"/pmc_write","WRITE_SIZE")
res={}
for k in sorted(set(fetch)|set(write)):
    f=fetch.get(k,(0,0)); w=write.get(k,(0,0))
    res[k]={"FETCH_SIZE_KiB_per_launch":f[0],"WRITE_SIZE_KiB_per_launch":w[0],"launches":max(f[1],w[1])}
json.dump(res, open(out+"/pmc_traffic_raw.json","w"), indent=1)
for k,v in res.items(): print(k, v)
PY
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do echo "== $f"; cat $f | head -12; done
