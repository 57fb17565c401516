#!/bin/bash
# where do the MRHS kernels of cfg2 spend their wave residency?  (rocprofv3 PMC passes on tools/mrhs_probe.py; counters
# only, two passes; per kernel, averaged over the REAL dispatches -- an idle graph iteration issues < 10 % of the
# instructions of a real one).  Writes gpurun_out/pmc_mrhs/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_mrhs
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -o pmc -- python $R/tools/mrhs_probe.py > $OUT/stdout_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/b -o pmc -- python $R/tools/mrhs_probe.py > $OUT/stdout_b.log 2>&1
python - <<PY
import csv, glob, collections, json, re
res = {}
for sub, key_ctr in (("a", "SQ_ACTIVE_INST_VALU"), ("b", "SQ_INSTS_VALU")):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
    for fn in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"]
            m = re.match(r"(?:void )?(?:vp::)?([A-Za-z0-9_]+)(<.*>)?\(", k)
            if not m or not m.group(1).startswith("mrhs_"): continue
            short = m.group(1)
            if short == "mrhs_coop_out_kernel": short += "_r_and_J" if m.group(2).rstrip(">").endswith(", 2, 1") else "_r_only"
            did = (short, row["Dispatch_Id"]); name[did] = short
            per[did][row["Counter_Name"]] += float(row["Counter_Value"])
    byk = collections.defaultdict(list)
    for did, c in per.items(): byk[name[did]].append(c)
    for k, lst in byk.items():
        top = max(c.get(key_ctr, 0.0) for c in lst)
        real = [c for c in lst if c.get(key_ctr, 0.0) >= 0.1 * top]
        e = res.setdefault(k, {})
        e["dispatches_%s" % sub] = len(lst); e["real_dispatches_%s" % sub] = len(real)
        for ctr in sorted({x for c in real for x in c}): e[ctr] = sum(c.get(ctr, 0.0) for c in real) / len(real)
for k, e in res.items():
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
        e["valu_active_frac_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"]
json.dump({"command": "tools/pmc_mrhs.sh (rocprofv3 --pmc, two passes, tools/mrhs_probe.py: cfg2 at full size)", "per_kernel_real_dispatch_average": res},
          open("$OUT/summary.json", "w"), indent=1)
for k, e in sorted(res.items()):
    print(k, {c: (round(v, 3) if v < 10 else round(v)) for c, v in e.items()})
PY
