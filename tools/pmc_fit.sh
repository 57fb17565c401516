#!/bin/bash
# VALU instruction count / issue utilisation of the fit kernels (separate PMC pass, counters only), wave kernel and
# slot kernel side by side:  bash tools/pmc_fit.sh [tag]   -> gpurun_out/<tag>/fit_pmc.json
TAG=${1:-pmc_fit}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/raw -o valu -- python $R/tools/slot_probe.py 65536 > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
dur = collections.defaultdict(dict)
for fn in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r["Kernel_Name"]
        key = "fit2_kernel" if "fit2_kernel" in name else ("fit_kernel" if "fit_kernel" in name else None)
        if key is None: continue
        rows[key][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        dur[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for key in rows:
    o = {}
    for c, v in rows[key].items():
        vals = list(v.values()); o[c] = sum(vals) / len(vals)
    d = list(dur[key].values()); o["launches"] = len(d); o["avg_duration_ns_under_pmc"] = sum(d) / len(d)
    out[key] = o
    print(key, json.dumps(o))
json.dump(out, open("$OUT/fit_pmc.json", "w"), indent=1)
PY
tail -12 $OUT/stdout.log
rm -rf $OUT/raw
