#!/bin/bash
# VALU instruction count / issue utilisation of the fit kernel (separate PMC pass, counters only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/valu
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -o valu -- python $R/tools/perf_probe.py 65536 > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "fit_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 65536 * 64:
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in rows.items():
    vals = [sum(x) for x in v.values()]
    print("%-22s per launch %.4g  (%d launches)" % (k, sum(vals) / len(vals), len(vals)))
d = list(dur.values())
print("duration ns (under PMC) avg %.0f" % (sum(d) / len(d)))
PY
tail -4 $OUT/stdout.log
