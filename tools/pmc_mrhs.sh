#!/bin/bash
# where does an MRHS streaming pass wait?  (rocprofv3 PMC pass on tools/mrhs_probe.py; counters only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_mrhs
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/tools/mrhs_probe.py > $OUT/stdout_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/b -o pmc -- python $R/tools/mrhs_probe.py > $OUT/stdout_b.log 2>&1
python - <<PY
import csv,glob,collections
for sub in ("a","b"):
    f=glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
    for fn in f:
        for row in csv.DictReader(open(fn)):
            k=row["Kernel_Name"]
            if "mrhs_stream" not in k and "mrhs_coop" not in k: continue
            k="mode0" if k.rstrip(")").split(",")[-1].strip().startswith("0") or "0>(" in k else "mode1"
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
            key=(k,row["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k]+=1
    for k,v in agg.items():
        print(sub,k,"dispatches",cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
PY
tail -3 $OUT/stdout_b.log
