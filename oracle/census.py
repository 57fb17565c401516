"""Parity census at batch scale (TEST INFRASTRUCTURE, like everything under oracle/): the device's batched LM fit against
the oracle's fit of the SAME problems, problem by problem -- what `LevMarSolver::fit` returns for each of them
(/root/reference/src/solvers/levmar/mod.rs:238-254: the Ok / Err split by `termination.was_successful()`, src/fit.rs:113-122:
parameters, objective, number of evaluations).  Used by tests/test_gpu_census.py and by bench.py's cpu_baseline leg."""
import numpy as np

TERM = {1: "ResidualsZero", 2: "Orthogonal", 3: "Converged{ftol}", 4: "Converged{xtol}", 5: "Converged{ftol,xtol}", 0: "NotRun",
        -1: "User", -2: "Numerical", -3: "NoImprovementPossible", -4: "LostPatience", -5: "NoParameters", -6: "NoResiduals",
        -7: "WrongDimensions"}


def census(rep_dev, alpha_dev, rep_orc, alpha_orc, max_listed=40):
    """rep_*: structured arrays (termination, n_evals, objective) of the same problems; alpha_*: (B, q).
    Returns a dict of agreement figures; `disagreements` lists every problem whose success CLASS differs (first
    `max_listed` in full) with both termination codes, evaluation counts, objectives and the largest relative parameter
    difference -- the material to explain each of them."""
    td, to = np.asarray(rep_dev["termination"]), np.asarray(rep_orc["termination"])
    nd, no = np.asarray(rep_dev["n_evals"]).astype(np.int64), np.asarray(rep_orc["n_evals"]).astype(np.int64)
    od, oo = np.asarray(rep_dev["objective"], dtype=np.float64), np.asarray(rep_orc["objective"], dtype=np.float64)
    ad, ao = np.asarray(alpha_dev, dtype=np.float64), np.asarray(alpha_orc, dtype=np.float64)
    B = td.size
    okd, oko = td > 0, to > 0
    both = okd & oko
    rel_obj = np.abs(od - oo)[both] / np.maximum(np.abs(oo[both]), 1e-300)
    rel_a = (np.abs(ad - ao) / np.maximum(np.abs(ao).max(1, keepdims=True), 1e-300)).max(1)
    dis = np.nonzero(okd != oko)[0]
    listed = []
    for b in dis[:max_listed]:
        listed.append({"problem": int(b), "device": TERM.get(int(td[b]), int(td[b])), "oracle": TERM.get(int(to[b]), int(to[b])),
                       "evals_device": int(nd[b]), "evals_oracle": int(no[b]), "objective_device": float(od[b]),
                       "objective_oracle": float(oo[b]), "rel_objective_diff": float(abs(od[b] - oo[b]) / max(abs(oo[b]), 1e-300)),
                       "max_rel_alpha_diff": float(rel_a[b])})

    def by_code(t):
        codes, counts = np.unique(t[t <= 0], return_counts=True)
        return {TERM.get(int(c), str(int(c))): int(k) for c, k in zip(codes, counts)}

    dn = np.abs(nd - no)
    return {
        "problems": int(B),
        "same_success_class": float((okd == oko).mean()), "success_class_disagreements": int(dis.size),
        "same_termination_code": float((td == to).mean()),
        "failed_device": int((~okd).sum()), "failed_oracle": int((~oko).sum()), "failed_on_both": int((~okd & ~oko).sum()),
        "failures_by_code_device": by_code(td), "failures_by_code_oracle": by_code(to),
        "objective_rel_diff_median_common_successes": float(np.median(rel_obj)) if rel_obj.size else None,
        "objective_rel_diff_max_common_successes": float(rel_obj.max()) if rel_obj.size else None,
        "objective_rel_diff_p999_common_successes": float(np.quantile(rel_obj, 0.999)) if rel_obj.size else None,
        "objective_rel_diff_p90_common_successes": float(np.quantile(rel_obj, 0.9)) if rel_obj.size else None,
        "objective_rel_diff_p99_common_successes": float(np.quantile(rel_obj, 0.99)) if rel_obj.size else None,
        "share_objective_within_1e-6": float((rel_obj <= 1e-6).mean()) if rel_obj.size else None,
        "share_objective_within_1e-3": float((rel_obj <= 1e-3).mean()) if rel_obj.size else None,
        "alpha_rel_diff_median_common_successes": float(np.median(rel_a[both])) if both.any() else None,
        "share_evals_within_3": float((dn <= 3).mean()), "share_evals_equal": float((dn == 0).mean()),
        "sum_evals_device": int(nd.sum()), "sum_evals_oracle": int(no.sum()),
        "max_evals_device": int(nd.max()), "max_evals_oracle": int(no.max()),
        "disagreements": listed,
    }
