#!/bin/bash
# dynamic instruction mix of the kernels (rocprofv3 PMC pass; counters only, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc1
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/perf_probe.py ${1:-65536} > $OUT/stdout.log 2>&1
ls $OUT | head
python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*counter_collection.csv")
print(f)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for fn in f:
    for row in csv.DictReader(open(fn)):
        k=row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
        key=(row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k]+=1
for k,v in agg.items():
    n=cnt[k]
    w=v.get("SQ_WAVES",1)/n
    print(k, "dispatches",n, "waves/disp",w)
    for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES"):
        if c in v: print("    %-18s per wave %.1f" % (c, v[c]/n/w))
PY
