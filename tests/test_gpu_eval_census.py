"""Trait-level parity at BATCH scale: ONE evaluation (set_params + residuals + jacobian,
/root/reference/src/solvers/levmar/mod.rs:42-73, 91-95, 101-201) of every one of the 65 536 problems of the headline
workload (bench.py's configs[3] shard) at its initial guess, device against oracle, problem by problem -- for the
descriptor-language handle (`vp_evaluate`: the split `evaluate_kernel`) AND for the external-model handle given the same
columns (`vp_evaluate_with_basis`, `ext_evaluate_kernel`; reference plugin contract src/model/mod.rs:239-363).

The fit census (tests/test_gpu_census.py) compares what a whole fit returns; this one compares what ONE trait call
returns, so that a caller's own solver sees the oracle's numbers on every problem of a batch, not on the <= 48 problems the
fixed-alpha tests hold.

Contract (north_star: 1e-10 relative on c, r, J), asserted on EVERY problem:
  * r to 1e-10 max|y|, c to 1e-10 max|c|, cost to 1e-10 -- flat (measured: 2e-13 / 3e-12 / 2e-13 at worst);
  * J_k = -P_perp (D_k c) (Kaufman, :156-171) to 1e-10 max|J_k| on >= 99.8 % of the problems (measured 99.90 %; median
    2e-13).  The rest are guesses at which D_k c lies almost inside span(Phi) -- decay times within ~0.5 % of each other
    (d/dtau exp(-x/tau) is the limit of the difference of the two columns; cond(Phi) 3e3 - 2e4, c_1 ~ -c_2 large), or a slow
    decay whose derivative the three columns happen to represent well at cond(Phi) ~ 20 -- so that the projector cancels
    most of its input and two fp64 evaluations differ by up to 1.3e-7 of what is left.  WHO is right there is decided,
    problem by problem, by an 80-bit long-double evaluation (three-pass Gram-Schmidt QR): the device -- descriptor handle
    and external-model handle -- must be CLOSER to it than the oracle on >= 90 % of these problems, never more than 10 x
    further, and within 1e-7.  Measured: unweighted m = 1024: closer on 67 of 67, median 19 x; weighted m = 1000: closer
    on 53 of 56, median 14 x (the printed `long_double_arbitration` record).  Both are backward stable -- the error of P_perp T is c(m) eps cond(Phi) |T| in
    either -- and differ in c(m): the oracle, like the reference's nalgebra loops, sums its m-term dot products one after
    another (c ~ m at worst: 1024 eps x 1.7e4 = 3.9e-9 |T|, measured 3.2e-9), the device reduces them as lane-local
    partial sums + a wave tree (c ~ log m).
"""
import json

import numpy as np
import pytest
import torch

import varpro_amd as vp
from oracle import oracle as O
from varpro_amd import synth

pytestmark = pytest.mark.gpu

B, M = 65536, 1024


def _cond_phi(x, tau, w):
    """2-norm condition number of W [exp(-x/tau1) exp(-x/tau2) 1] per problem, from the 3 x 3 R of a QR in chunks"""
    out = np.empty(len(tau))
    for lo in range(0, len(tau), 4096):
        t = tau[lo:lo + 4096]
        Phi = np.stack([np.exp(-x[None, :] / t[:, 0:1]), np.exp(-x[None, :] / t[:, 1:2]), np.ones((len(t), len(x)))], axis=2)
        Phi = Phi * w[None, :, None]
        s = np.linalg.svd(np.linalg.qr(Phi, mode="r"), compute_uv=False)
        out[lo:lo + 4096] = s[:, 0] / s[:, -1]
    return out


def _dkc_scale(x, tau, Cref, w):
    """max_i |W D_k c| per problem and parameter: D_k c = c_k x / tau_k^2 exp(-x / tau_k)"""
    out = np.empty((len(tau), 2))
    for lo in range(0, len(tau), 8192):
        t = tau[lo:lo + 8192]
        for k in range(2):
            out[lo:lo + 8192, k] = np.abs(Cref[lo:lo + 8192, k:k + 1] * (w * x)[None, :] / t[:, k:k + 1] ** 2
                                          * np.exp(-x[None, :] / t[:, k:k + 1])).max(axis=1)
    return out


def _long_double_reference(x, y, tau, w):
    """c, r and the Kaufman J_k = -P_perp W D_k c of one problem in 80-bit long double (three-pass Gram-Schmidt QR of W Phi)"""
    LD = np.longdouble
    x, y, tau, w = x.astype(LD), y.astype(LD) * w.astype(LD), tau.astype(LD), w.astype(LD)
    e = [np.exp(-x / tau[0]), np.exp(-x / tau[1])]
    Phi = np.stack([e[0] * w, e[1] * w, w], axis=1)
    Q, R = np.zeros_like(Phi), np.zeros((3, 3), dtype=LD)
    for j in range(3):
        v = Phi[:, j].copy()
        for _ in range(3):
            for i in range(j):
                h = Q[:, i] @ v
                R[i, j] += h
                v = v - h * Q[:, i]
        R[j, j] = np.sqrt(v @ v)
        Q[:, j] = v / R[j, j]
    b = Q.T @ y
    c = np.zeros(3, dtype=LD)
    for i in (2, 1, 0):
        c[i] = (b[i] - R[i, i + 1:] @ c[i + 1:]) / R[i, i]
    r = y - Phi @ c
    J = []
    for k in range(2):
        T = w * e[k] * x / tau[k] ** 2 * c[k]
        for _ in range(2):
            T = T - Q @ (Q.T @ T)
        J.append(-T)
    return c, r, np.stack(J)


def _errors(ev, ref, ymax, dkc):
    r = np.abs(ev["r"] - ref["r"]).max(axis=1) / ymax
    c = np.abs(ev["C"] - ref["C"]).max(axis=1) / np.abs(ref["C"]).max(axis=1)
    dJ = np.abs(ev["J"] - ref["J"]).max(axis=2)
    J = (dJ / np.abs(ref["J"]).max(axis=2)).max(axis=1)
    Jt = (dJ / dkc).max(axis=1)
    cost = np.abs(ev["cost"] - ref["cost"]) / ref["cost"]
    return dict(r=r, c=c, J=J, J_rel_DkC=Jt, cost=cost)


def _to_np(d):
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items() if v is not None}


# (1024, unweighted): the headline workload -- the split FULL / UNIFORM evaluate kernel; (1000, weighted): the same batch
# cut to a length that fills no lane evenly, with per-row weights -- the general masked kernels of the same set
@pytest.mark.parametrize("m,weighted", [(M, False), (1000, True)])
def test_one_evaluation_of_every_headline_problem_matches_the_oracle(m, weighted):
    d = synth.double_exp_batch(B, m=m, noise=1e-3)
    x, Y, guess = np.asarray(d["x"], dtype=np.float64), d["Y"], np.asarray(d["tau_guess"], dtype=np.float64)
    w = np.random.default_rng(7).uniform(0.3, 2.0, m) if weighted else None
    w1 = w if weighted else np.ones(m)
    mdl = vp.multi_exponential_model(x, guess[0])
    ref = O.evaluate_batch(mdl, x, Y, guess, w=w, n_threads=min(16, O.max_threads()))
    ymax = np.abs(Y * w1[None, :]).max(axis=1)
    cond = _cond_phi(x, guess, w1)
    dkc = _dkc_scale(x, guess, ref["C"], w1)

    dev = torch.device("cuda:0")
    Yd = torch.as_tensor(Y, device=dev)
    gd = torch.as_tensor(guess, device=dev)
    bp = vp.BatchProblem(mdl, Yd, x=x, weights=w)
    ev = _to_np(bp.evaluate(gd))
    # the same columns through the external-model boundary
    phi, dphi = bp.basis(gd)
    bpx = vp.BatchProblem(vp.ExternalModel(3, 2, [(0, 0), (1, 1)]), Yd, weights=w)
    evx = _to_np(bpx.evaluate_with_basis(gd, phi, dphi))
    bp.close()
    bpx.close()

    summary = {"problems": B, "m": m, "weighted": weighted, "cond_phi": {"median": float(np.median(cond)), "p99": float(np.quantile(cond, 0.99)),
                                           "max": float(cond.max())}}
    ok = ref["status"] == 0
    assert ok.all()
    errs = {}
    for name, e in (("vp_evaluate", ev), ("vp_evaluate_with_basis", evx)):
        assert np.array_equal(e["status"], ref["status"]), name
        errs[name] = E = _errors(e, ref, ymax, dkc)
        summary[name] = {q: {"median": float(np.median(v)), "max": float(v.max()), "share_within_1e-10": float((v <= 1e-10).mean())}
                         for q, v in E.items()}
        print(json.dumps({name: summary[name]}))
    print(json.dumps({"m": m, "weighted": weighted, "cond_phi": summary["cond_phi"]}))

    # who is right where the two differ: every problem on which either device path is further than 1e-10 max|J_k| from the
    # oracle is recomputed in long double; the device must be closer to that than the oracle on >= 90 % of them and never
    # more than 10 x further
    over = np.nonzero((errs["vp_evaluate"]["J"] > 1e-10) | (errs["vp_evaluate_with_basis"]["J"] > 1e-10))[0]
    over = over[np.argsort(-errs["vp_evaluate"]["J"][over])]
    assert len(over) <= 0.002 * B
    arb = {"problems_above_1e-10": int(len(over)), "device_closer_than_oracle": 0, "worst_device_vs_long_double": 0.0, "worst_oracle_vs_long_double": 0.0,
           "median_oracle_error_over_device_error": None}
    ratios = []
    for i, b in enumerate(over):
        c, r, J = _long_double_reference(x, Y[b], guess[b], w1)
        sc = np.abs(J).max(axis=1, keepdims=True)
        row = {"problem": int(b), "tau_guess": guess[b].tolist(), "cond_phi": float(cond[b]),
               "device_vs_oracle": float(errs["vp_evaluate"]["J"][b]),
               "device_vs_long_double": float((np.abs(ev["J"][b] - J) / sc).max()),
               "external_vs_long_double": float((np.abs(evx["J"][b] - J) / sc).max()),
               "oracle_vs_long_double": float((np.abs(ref["J"][b] - J) / sc).max())}
        if i < 8:
            print(json.dumps(row))
        for k in ("device_vs_long_double", "external_vs_long_double"):
            assert row[k] <= 10.0 * row["oracle_vs_long_double"] and row[k] <= 1e-7, row
            arb["worst_device_vs_long_double"] = max(arb["worst_device_vs_long_double"], row[k])
        arb["worst_oracle_vs_long_double"] = max(arb["worst_oracle_vs_long_double"], row["oracle_vs_long_double"])
        arb["device_closer_than_oracle"] += int(row["device_vs_long_double"] <= row["oracle_vs_long_double"])
        ratios.append(row["oracle_vs_long_double"] / max(row["device_vs_long_double"], 1e-300))
    if ratios:
        arb["median_oracle_error_over_device_error"] = float(np.median(ratios))
    print(json.dumps({"long_double_arbitration": arb}))
    assert arb["device_closer_than_oracle"] >= 0.9 * len(over)

    for name in ("vp_evaluate", "vp_evaluate_with_basis"):
        S = summary[name]
        for q in ("r", "c", "cost"):
            assert S[q]["max"] <= 1e-10, (name, q, S[q])
            assert S[q]["median"] <= 1e-13, (name, q, S[q])
        assert S["J"]["share_within_1e-10"] >= 0.998 and S["J"]["median"] <= 1e-12, (name, S["J"])
