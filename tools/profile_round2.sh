#!/bin/bash
# Profile of the bench command on the GPU box (bash tools/profile_round2.sh [tag]; round 3: tag r03):
#   1. rocprofv3 --kernel-trace --stats           -> per-kernel durations + the timestamped trace of the pipelined leg
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE    -> HBM traffic per kernel (separate passes, counters only)
#   3. rocprofv3 --pmc SQ_* (tools/pmc_fit.sh)    -> VALU instruction counts of the fit kernels
# Summaries land in gpurun_out/<tag>/summary/ ; copy them to profiles/ to commit.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT/summary
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --emulate-shards 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD > $OUT/summary/${TAG}_bench_under_rocprof_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/bench_pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/bench_pmc_write_stdout.log 2>&1
python - <<PY
import csv, glob, json, collections, re
out = "$OUT"; tag = "$TAG"
def short(name):
    m = re.match(r"(?:void )?(?:vp::)?(?:(?:blk|ext|gen)::)?([A-Za-z0-9_]+)<(.*)>\(", name)
    if not m: return name.split("(")[0][:50]
    k, targs = m.group(1), m.group(2)
    if k in ("blk_fit_kernel", "blk_evaluate_kernel") and "RtModel" in targs: return k + "_rt"
    if k == "ext_evaluate_kernel":  # <T, N, P, R, WITH_D, W>
        wv = targs.split(",")[-1].strip()
        return k + ("" if "true" in targs else "_no_derivatives") + ("" if wv in ("1", "true", "false") else "_w" + wv)
    if k == "mrhs_stream_kernel": return "%s_mode%s" % (k, targs.split(",")[-1].strip())
    if k == "evaluate_kernel":  # MODE (0: c / cost, 1: + r, 2: + r + J) is the 5th template argument
        flat = re.sub(r"<[^<>]*>", "", targs).split(",")
        mode = flat[4].strip() if len(flat) > 4 else "2"
        return k + ("_f32" if targs.startswith("float") else "") + ("" if mode == "2" else "_mode" + mode)
    if k in ("fit_kernel", "fit2_kernel", "basis_kernel"):
        return k + ("_f32" if targs.startswith("float") else "")
    return k
# ---- 1. kernel stats + trace ----
trace = []
for fn in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        trace.append((short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                      r.get("Queue_Id", ""), int(r.get("VGPR_Count", 0) or 0), int(r.get("Scratch_Size", 0) or 0)))
agg = collections.defaultdict(list)
for k, s, e, g, q, v, sc in trace: agg[(k, g)].append(e - s)
rows = [{"kernel": k, "grid_size": g, "calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3,
         "total_ms": sum(v) / 1e6} for (k, g), v in agg.items()]
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline", "kernels": rows},
          open(out + "/summary/%s_bench_kernel_stats.json" % tag, "w"), indent=1)
for r in rows[:14]: print("%-34s grid %9d calls %4d avg %9.1f us  total %8.2f ms" % (r["kernel"], r["grid_size"], r["calls"], r["avg_us"], r["total_ms"]))
# pipelined overlap: the headline-size fit launches, in start order; the pipelined leg alternates two queues
fits = sorted([(s, e, q) for k, s, e, g, q, v, sc in trace if k == "fit2_kernel" and g >= 2048 * 64 - 1 and (e - s) > 1.5e6], key=lambda t: t[0])
pairs = []
for (s0, e0, q0), (s1, e1, q1) in zip(fits, fits[1:]):
    if q0 != q1 and s1 < e0: pairs.append({"first_start_ns": s0, "first_end_ns": e0, "second_start_ns": s1, "second_end_ns": e1,
                                           "overlap_us": (e0 - s1) / 1e3, "queues": [q0, q1]})
ov = {"what": "consecutive fit2_kernel launches of the pipelined leg (step k on stream k mod 2) whose execution intervals overlap; "
              "timestamps from rocprofv3 --kernel-trace", "overlapping_pairs": len(pairs),
      "mean_overlap_us": (sum(p["overlap_us"] for p in pairs) / len(pairs)) if pairs else 0.0,
      "mean_kernel_us": (sum(e - s for s, e, q in fits) / len(fits) / 1e3) if fits else 0.0, "examples": pairs[:6]}
json.dump(ov, open(out + "/summary/%s_pipelined_overlap.json" % tag, "w"), indent=1)
print("pipelined overlap: %d pairs, mean overlap %.0f us of mean kernel %.0f us" % (ov["overlapping_pairs"], ov["mean_overlap_us"], ov["mean_kernel_us"]))
# ---- 2. PMC traffic ----
res = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = collections.defaultdict(list)
    for fn in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % kind, recursive=True):
        per = collections.defaultdict(float); meta = {}
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != ctr: continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        for did, v in per.items(): d[meta[did]].append(v)
    for (name, grid), v in d.items():
        e = res.setdefault("%s@grid%d" % (name, grid), {"kernel": name, "grid_size": grid})
        e[ctr + "_KiB_per_launch"] = sum(v) / len(v); e["launches_" + ctr] = len(v)
tr = {"command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline (tools/profile_round2.sh; separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, --kernel-trace only)",
      "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md HBM section) -> hbm bytes = WRITE_SIZE + 2*FETCH_SIZE; counters are in KiB",
      "per_kernel_and_grid": {}}
big = {}
for key, e in res.items():
    if "FETCH_SIZE_KiB_per_launch" in e and "WRITE_SIZE_KiB_per_launch" in e:
        e["hbm_bytes_per_launch_corrected"] = (e["WRITE_SIZE_KiB_per_launch"] + 2.0 * e["FETCH_SIZE_KiB_per_launch"]) * 1024.0
        tr["per_kernel_and_grid"][key] = e
        if e["kernel"] not in big or e["hbm_bytes_per_launch_corrected"] > big[e["kernel"]]["hbm_bytes_per_launch_corrected"]: big[e["kernel"]] = e
for k, e in big.items(): tr[k] = e   # the largest launch of every kernel under its plain name (bench.py reads these)
json.dump(tr, open(out + "/summary/%s_pmc_traffic.json" % tag, "w"), indent=1)
for k, e in sorted(big.items()): print("%-34s grid %9d  HBM %10.1f MB per launch" % (k, e["grid_size"], e["hbm_bytes_per_launch_corrected"] / 1e6))
PY
bash $R/tools/pmc_fit.sh ${TAG}_valu > /dev/null 2>&1
cp $R/gpurun_out/${TAG}_valu/fit_pmc.json $OUT/summary/${TAG}_fit_kernels_valu_pmc.json 2>/dev/null
for f in $(find $OUT/stats -name "*kernel_stats.csv" | head -1); do cp $f $OUT/summary/${TAG}_bench_kernel_stats.csv; done
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $R/gpurun_out/${TAG}_valu
ls -la $OUT/summary
tail -3 $OUT/summary/${TAG}_bench_under_rocprof_stdout.log | cut -c1-600
