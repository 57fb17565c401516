#!/usr/bin/env python
"""Generate the high-precision golden vectors that pin the 1e-10 contract.

The reference (Rust) cannot be run in the build image, and its own tests assert nothing tighter than
1e-8 on this path, so the 1e-10 tolerance of BASELINE.json is pinned against an INDEPENDENT evaluation
of the reference's formulas in 50-digit arithmetic (mpmath):

    Phi_w = W Phi(alpha);  C = argmin ||Y_w - Phi_w C||  (normal equations are exact enough at 50 digits);
    R = Y_w - Phi_w C;  J_k = -(I - Phi_w Phi_w^+) (W dPhi/dalpha_k) C          (src/solvers/levmar/mod.rs:42-201)

Inputs are float64 values (exactly representable), outputs are rounded to float64 at the end.
Run:  python tests/golden/make_golden.py   (writes tests/golden/golden_eval.npz; ~1 minute)
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from varpro_amd import synth  # noqa: E402

mp.mp.dps = 50


def basis(kind, t, p):
    if kind == 0:
        return mp.mpf(1), []
    if kind == 1:
        e = mp.exp(-t / p[0])
        return e, [e * t / (p[0] * p[0])]
    if kind == 3:
        ex = mp.exp(-p[0] * t)
        return ex * mp.cos(p[1] * t), [-t * ex * mp.cos(p[1] * t), -t * ex * mp.sin(p[1] * t)]
    raise ValueError(kind)


def evaluate(kinds, pidx, q, x, y, alpha, w=None):
    m, n = len(x), len(kinds)
    X = [mp.mpf(float(v)) for v in x]
    Yv = [mp.mpf(float(v)) for v in y]
    A = [mp.mpf(float(v)) for v in alpha]
    W = [mp.mpf(1)] * m if w is None else [mp.mpf(float(v)) for v in w]
    Phi = mp.zeros(m, n)
    D = [mp.zeros(m, n) for _ in range(q)]
    for i in range(m):
        for j in range(n):
            p = [A[k] for k in pidx[j]]
            f, df = basis(kinds[j], X[i], p)
            Phi[i, j] = W[i] * f
            for a, k in enumerate(pidx[j]):
                D[k][i, j] += W[i] * df[a]
    yw = mp.matrix([W[i] * Yv[i] for i in range(m)])
    G = Phi.T * Phi
    c = mp.lu_solve(G, Phi.T * yw)
    r = yw - Phi * c
    J = []
    for k in range(q):
        T = D[k] * c
        proj = Phi * mp.lu_solve(G, Phi.T * T)
        J.append(proj - T)
    tof = lambda v: np.array([float(z) for z in v])
    return tof(c), tof(r), np.stack([tof(j) for j in J])


def main():
    out = {}
    # case A: BASELINE configs[0] inputs (quirk grid, cond(Phi) up to 9e4) at the initial guess and near the solution
    c0 = synth.config0()
    kinds, pidx = [1, 1, 0], [(0,), (1,), ()]
    for tag, alpha in (("cfg0_guess", c0["tau_guess"]), ("cfg0_near", c0["tau_true"] * np.array([1.05, 0.97]))):
        c, r, J = evaluate(kinds, pidx, 2, c0["x"], c0["y"], alpha)
        out[tag + "_x"], out[tag + "_y"], out[tag + "_alpha"] = c0["x"], c0["y"], np.asarray(alpha, float)
        out[tag + "_c"], out[tag + "_r"], out[tag + "_J"] = c, r, J
        print(tag, "done")
    # case B: configs[1]-style problem with noise and weights, m = 256
    d = synth.double_exp_batch(1, m=256, noise=1e-3)
    rng = np.random.default_rng(1)
    w = 0.5 + rng.random(256)
    c, r, J = evaluate(kinds, pidx, 2, d["x"], d["Y"][0], d["tau_guess"][0], w)
    out["cfg1w_x"], out["cfg1w_y"], out["cfg1w_alpha"], out["cfg1w_w"] = d["x"], d["Y"][0], d["tau_guess"][0], w
    out["cfg1w_c"], out["cfg1w_r"], out["cfg1w_J"] = c, r, J
    print("cfg1w done")
    # case C: O'Leary example (shared parameters, exp*cos), weighted
    t = np.array([0., 0.1, 0.22, 0.31, 0.46, 0.50, 0.63, 0.78, 0.85, 0.97])
    y = np.array([6.9842, 5.1851, 2.8907, 1.4199, -0.2473, -0.5243, -1.0156, -1.0260, -0.9165, -0.6805])
    wt = np.array([1.0, 1.0, 1.0, 0.5, 0.5, 1.0, 0.5, 1.0, 0.5, 0.5])
    alpha = np.array([0.5, 2., 3.])
    c, r, J = evaluate([3, 3], [(1, 2), (0, 1)], 3, t, y, alpha, wt)
    out["oleary_x"], out["oleary_y"], out["oleary_alpha"], out["oleary_w"] = t, y, alpha, wt
    out["oleary_c"], out["oleary_r"], out["oleary_J"] = c, r, J
    print("oleary done")
    # case D: ill-conditioned triple exponential + offset (close decay times, cond(Phi) 1e3 .. 1e5), m = 129:
    # where QR- and SVD-based evaluations drift apart, the 50-digit values say who is right
    rng = np.random.default_rng(77)
    m3 = 129
    x3 = np.linspace(0.0, 12.0, m3)
    taus = np.array([[1.0, 1.05, 1.1], [2.0, 2.1, 2.2], [1.0, 1.02, 1.04], [3.0, 3.3, 3.6]])
    for i, tau in enumerate(taus):
        cc = rng.uniform(1, 50, 4)
        y3 = sum(cc[j] * np.exp(-x3 / tau[j]) for j in range(3)) + cc[3]
        y3 = y3 + 1e-4 * np.abs(y3).max() * rng.standard_normal(m3)
        alpha3 = tau * rng.uniform(0.9, 1.1, 3)
        c, r, J = evaluate([1, 1, 1, 0], [(0,), (1,), (2,), ()], 3, x3, y3, alpha3)
        tag = "triple_close%d" % i
        out[tag + "_x"], out[tag + "_y"], out[tag + "_alpha"] = x3, y3, alpha3
        out[tag + "_c"], out[tag + "_r"], out[tag + "_J"] = c, r, J
        print(tag, "done")
    # case F: fit statistics at the cfg1w point (src/statistics/mod.rs:352-441, 481-511): J = [Phi, (dPhi/dalpha_k c)_k]
    # unweighted, H = W J, sigma^2 = ||r_w||^2 / (m - n - q), Cov = sigma^2 (H^T H)^-1, conf sigma_i = sqrt(j_i^T Cov j_i)
    def statistics(kinds_, pidx_, q_, x_, y_, alpha_, w_):
        mm, nn = len(x_), len(kinds_)
        X = [mp.mpf(float(v)) for v in x_]
        A = [mp.mpf(float(v)) for v in alpha_]
        Wv = [mp.mpf(float(v)) for v in w_]
        c_, r_, _ = evaluate(kinds_, pidx_, q_, x_, y_, alpha_, w_)
        cm = [mp.mpf(float(v)) for v in c_]   # float64-rounded coefficients, as the kernel sees them
        Jm = mp.zeros(mm, nn + q_)
        for i in range(mm):
            for j in range(nn):
                f, df = basis(kinds_[j], X[i], [A[k] for k in pidx_[j]])
                Jm[i, j] = f
                for a_, k in enumerate(pidx_[j]):
                    Jm[i, nn + k] += df[a_] * cm[j]
        H = mp.zeros(mm, nn + q_)
        for i in range(mm):
            for j in range(nn + q_):
                H[i, j] = Wv[i] * Jm[i, j]
        rr = sum(mp.mpf(float(v)) ** 2 for v in r_)
        s2 = rr / (mm - nn - q_)
        Cov = s2 * mp.inverse(H.T * H)
        sig = []
        for i in range(mm):
            ji = Jm[i, :]
            sig.append(mp.sqrt((ji * Cov * ji.T)[0, 0]))
        K = nn + q_
        return (np.array([[float(Cov[a_, b_]) for b_ in range(K)] for a_ in range(K)]), float(s2),
                np.array([float(v) for v in sig]))
    cov, chi2, sig = statistics(kinds, pidx, 2, d["x"], d["Y"][0], d["tau_guess"][0], w)
    out["cfg1w_cov"], out["cfg1w_chi2"], out["cfg1w_sigma"] = cov, np.array([chi2]), sig
    print("cfg1w statistics done")
    # case E: multiple right-hand sides (one alpha, S = 3 columns), weighted, m = 40: per column c_s, r_s and
    # J_k[:, s] = -P_perp (W dPhi/dalpha_k) c_s   (src/solvers/levmar/mod.rs:147-186); layouts [S][n], [S][m], [q][S][m]
    rng = np.random.default_rng(5)
    xm = np.linspace(0.0, 9.0, 40)
    wm = 0.5 + rng.random(40)
    alpham = np.array([1.3, 3.8])
    Cm = rng.uniform(1, 10, (3, 3))
    Ym = Cm[:, :1] * np.exp(-xm / 1.0) + Cm[:, 1:2] * np.exp(-xm / 3.0) + Cm[:, 2:] + 1e-3 * rng.standard_normal((3, 40))
    cs, rs, Js = [], [], []
    for si in range(3):
        c, r, J = evaluate(kinds, pidx, 2, xm, Ym[si], alpham, wm)
        cs.append(c); rs.append(r); Js.append(J)
    out["mrhs3_x"], out["mrhs3_y"], out["mrhs3_alpha"], out["mrhs3_w"] = xm, Ym, alpham, wm
    out["mrhs3_c"], out["mrhs3_r"] = np.stack(cs), np.stack(rs)
    out["mrhs3_J"] = np.stack(Js, axis=1)  # [q][S][m]
    print("mrhs3 done")
    np.savez_compressed(os.path.join(HERE, "golden_eval.npz"), **out)


if __name__ == "__main__":
    main()
