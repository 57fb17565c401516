"""Parity census of ALL of BASELINE configs[3]: 524 288 double-exponential problems = 8 shards of 65 536 (the contiguous
split of varpro_amd/distributed.py:shard_range), every shard fitted on the device and by the oracle (16 threads), problem by
problem.  usage: PYTHONPATH=. python tools/census_cfg3.py [out.json]"""
import json
import sys
import time

import numpy as np

import varpro_amd as vp
from oracle import census as CS
from oracle import oracle as O
from varpro_amd import distributed as vd
from varpro_amd import synth

G, B, m = 8, 65536, 1024
thr = min(16, O.max_threads())
out = {"what": "vp_fit vs the oracle on every problem of BASELINE configs[3] (8 shards x 65536, m = 1024, fp64, noise 1e-3)",
       "oracle_threads": thr, "shards": []}
tot = {"problems": 0, "success_class_disagreements": 0, "failed_device": 0, "failed_oracle": 0, "failed_on_both": 0,
       "sum_evals_device": 0, "sum_evals_oracle": 0}
worst_obj, within3 = 0.0, 0.0
bp = None
for g in range(G):
    first, count = vd.shard_range(G * B, g, G)
    d = synth.double_exp_batch(count, m=m, first_problem=first, noise=1e-3)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    if bp is None:
        bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    else:
        bp.set_observations(d["Y"])
    a, c, rep = bp.fit(d["tau_guess"])
    t0 = time.time()
    ao, co, ro, _s = O.fit_batch(mdl, d["x"], d["Y"], d["tau_guess"], n_threads=thr)
    res = CS.census(rep, a, ro, ao)
    res["shard"] = g
    res["first_problem"] = int(first)
    res["oracle_seconds"] = round(time.time() - t0, 2)
    out["shards"].append(res)
    for k in tot:
        tot[k] += res[k]
    worst_obj = max(worst_obj, res["objective_rel_diff_max_common_successes"])
    within3 += res["share_evals_within_3"] * count
    print("shard %d: same class %.6f  disagreements %d  failed %d / %d  obj median %.2e max %.2e  evals within 3: %.4f  max evals %d / %d"
          % (g, res["same_success_class"], res["success_class_disagreements"], res["failed_device"], res["failed_oracle"],
             res["objective_rel_diff_median_common_successes"], res["objective_rel_diff_max_common_successes"],
             res["share_evals_within_3"], res["max_evals_device"], res["max_evals_oracle"]), flush=True)
tot["objective_rel_diff_max_common_successes"] = worst_obj
tot["share_evals_within_3"] = within3 / (G * B)
tot["same_success_class"] = 1.0 - tot["success_class_disagreements"] / float(G * B)
out["total"] = tot
print(json.dumps(tot))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
