#!/bin/bash
# instruction counts of the Gram fit kernel at cfg4 (separate PMC pass): bash tools/pmc_cfg4.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_cfg4
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/raw -o c4 -- python $R/tools/cfg4_hist.py 8192 > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for fn in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "fitg" not in r["Kernel_Name"]: continue
        rows[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
o = {c: sum(v.values()) / len(v) for c, v in rows.items()}
o["launches"] = len(dur); o["avg_duration_ns_under_pmc"] = sum(dur.values()) / max(1, len(dur))
print(json.dumps(o, indent=1)); json.dump(o, open("$OUT/cfg4_pmc.json", "w"), indent=1)
PY
tail -1 $OUT/stdout.log | cut -c1-200
rm -rf $OUT/raw
