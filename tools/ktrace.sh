#!/bin/bash
# per-kernel durations of any probe script: tools/ktrace.sh <tag> <python script + args>   (rocprofv3 --kernel-trace)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktrace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o $TAG -- python $R/"$@" > $OUT/stdout.log 2>&1
tail -8 $OUT/stdout.log
python - <<PY
import csv, glob, collections, re, json
agg = collections.defaultdict(list)
for fn in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        m = re.match(r"(?:void )?(?:vp::)?(?:gen::)?([A-Za-z0-9_]+)(<.*>)?\(", n)
        k = m.group(1) if m else n[:40]
        if m and m.group(2) and ("stream" in k or "coop" in k): k += "_mode" + m.group(2).rstrip(">").split(",")[-1].strip()
        agg[(k, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(({"kernel": k, "grid": g, "calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3, "total_ms": sum(v) / 1e6}
              for (k, g), v in agg.items()), key=lambda r: -r["total_ms"])
json.dump(rows, open("$OUT/kernels.json", "w"), indent=1)
for r in rows[:16]: print("%-34s grid %9d calls %5d avg %9.1f min %9.1f max %9.1f us  total %8.2f ms" % (r["kernel"], r["grid"], r["calls"], r["avg_us"], r["min_us"], r["max_us"], r["total_ms"]))
PY
