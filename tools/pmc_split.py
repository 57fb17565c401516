"""profiles/<tag>_pmc_traffic.json from the PMC passes of tools/profile_round.sh: per kernel AND grid size."""
import csv, collections, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
res = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open("gpurun_out/%s/pmc_%s/%s_counter_collection.csv" % (tag, kind, tag))):
        if row["Counter_Name"] != ctr: continue
        for name in ("fit_kernel", "basis_kernel"):
            if name in row["Kernel_Name"]: d[(name, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    for (name, grid), v in d.items():
        e = res.setdefault("%s@grid%d" % (name, grid), {})
        e[ctr + "_KiB_per_launch"] = sum(v) / len(v)
        e["launches_" + ctr] = len(v)
def corrected(e): return (e["WRITE_SIZE_KiB_per_launch"] + 2.0 * e["FETCH_SIZE_KiB_per_launch"]) * 1024.0
out = {"batch": 65536, "m": 1024,
       "command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline (tools/profile_round.sh %s; separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, --kernel-trace only)" % tag,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md HBM section) -> hbm bytes = WRITE_SIZE + 2*FETCH_SIZE; counters are in KiB",
       "grid_note": "Grid_Size 4194304 = 65536 problems x 64 lanes (the bench workload); 262144 = the 4096-problem configs[1] side measurement"}
for name in ("basis_kernel", "fit_kernel"):
    e = dict(res["%s@grid4194304" % name]); e["hbm_bytes_per_launch_corrected"] = corrected(e); out[name] = e
e = dict(res["fit_kernel@grid262144"]); e["hbm_bytes_per_launch_corrected"] = corrected(e); out["fit_kernel_B4096"] = e
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
