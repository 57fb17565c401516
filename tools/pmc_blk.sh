#!/bin/bash
# Instruction counts / wait cycles of the streamed fit kernel on the bench's m = 10 000 leg (separate PMC passes, counters
# only):  bash tools/pmc_blk.sh [tag]   -> gpurun_out/<tag>/blk_pmc.json
TAG=${1:-pmc_blk}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/raw1 -o a -- python $R/tools/stream_ab_probe.py first > $OUT/stdout1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/raw2 -o b -- python $R/tools/stream_ab_probe.py first > $OUT/stdout2.log 2>&1
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for fn in glob.glob("$OUT/raw*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "blk_fit_kernel" not in r["Kernel_Name"]: continue
        rows[r["Counter_Name"]][fn + r["Dispatch_Id"]] += float(r["Counter_Value"])
        dur[fn + r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
o = {c: sum(v.values()) / len(v) for c, v in rows.items()}
o["avg_duration_ns_under_pmc"] = sum(dur.values()) / max(1, len(dur)); o["launches"] = len(dur)
print(json.dumps(o, indent=1)); json.dump(o, open("$OUT/blk_pmc.json", "w"), indent=1)
PY
tail -3 $OUT/stdout1.log; tail -3 $OUT/stdout2.log
rm -rf $OUT/raw1 $OUT/raw2
