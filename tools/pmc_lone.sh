#!/bin/bash
# issue rate of a LONE wavefront in fit_kernel (B = 256: at most one wave per SIMD): bash tools/pmc_lone.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_lone
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/raw -o lone -- python $R/tools/lone_probe.py 256 > $OUT/stdout.log 2>&1
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for fn in glob.glob("$OUT/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "fit_kernel" not in r["Kernel_Name"]: continue
        rows[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
o = {c: sum(v.values()) / len(v) for c, v in rows.items()}
o["launches"] = len(dur); o["avg_duration_ns_under_pmc"] = sum(dur.values()) / max(1, len(dur))
# SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (4 clocks)
if o.get("SQ_WAVE_CYCLES"):
    o["valu_instructions_per_wave"] = o["SQ_INSTS_VALU"] / o["SQ_WAVES"]
    o["cycles_per_valu_instruction_wave_resident"] = 4.0 * o["SQ_WAVE_CYCLES"] / o["SQ_INSTS_VALU"]
    o["valu_busy_fraction_of_wave_residency"] = o["SQ_ACTIVE_INST_VALU"] / o["SQ_WAVE_CYCLES"]
print(json.dumps(o, indent=1))
json.dump(o, open("$OUT/lone_pmc.json", "w"), indent=1)
PY
tail -2 $OUT/stdout.log
rm -rf $OUT/raw
