"""Randomised parity sweep for the multiple-right-hand-side path (GPU vs oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import varpro_amd as vp
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(N):
    nexp = int(rng.integers(1, 4)); off = True if nexp == 3 else bool(rng.integers(0, 2))
    mmax = 2048 if off and nexp >= 2 else 1024
    m = int(min(mmax, rng.choice([rng.integers(nexp + 3, 130), rng.integers(130, mmax + 1), mmax, 1000])))
    S = int(rng.integers(2, 50)); weighted = rng.random() < 0.4
    x = np.linspace(0.0, rng.uniform(6, 15), m)
    tau = np.sort(rng.uniform(0.4, 3.0, nexp)) + 1.5 * np.arange(nexp)
    Cm = rng.uniform(1, 50, (S, nexp + 1))
    Y = sum(Cm[:, j:j + 1] * np.exp(-x / tau[j]) for j in range(nexp)) + (Cm[:, -1:] if off else 0.0)
    Y = Y + rng.choice([1e-6, 1e-4, 1e-3]) * np.abs(Y).max() * rng.standard_normal(Y.shape)
    w = rng.uniform(0.3, 2.0, m) if weighted else None
    guess = tau * rng.uniform(0.85, 1.2, nexp)
    mdl = vp.multi_exponential_model(x, guess, offset=off)
    tag = "it %d: nexp %d off %d m %d S %d weighted %d" % (it, nexp, off, m, S, weighted)
    try:
        bp = vp.BatchProblem(mdl, Y[None], x=x, weights=w)
    except vp.VarproHipError as e:
        print("UNSUPPORTED", tag, e); continue
    ev = bp.evaluate(guess[None])
    ref = O.Problem(mdl, x, Y, w=w); ref.set_params(guess)
    yw = Y if w is None else Y * w
    ec = np.abs(ev["C"][0] - ref.linear_coefficients()).max() / np.abs(ref.linear_coefficients()).max()
    er = np.abs(ev["r"][0] - ref.residuals()).max() / np.abs(yw).max()
    Jr = ref.jacobian(); eJ = max(np.abs(ev["J"][0, k] - Jr[k]).max() / np.abs(Jr[k]).max() for k in range(nexp))
    if ec > 1e-10 or er > 1e-10 or eJ > 1e-8: print("PARITY C %.1e r %.1e J %.1e" % (ec, er, eJ), tag); bad += 1
    a, C, rep = bp.fit(guess[None])
    res = O.Problem(mdl, x, Y, w=w)
    res.set_params(guess)
    repo = res.fit()
    ao = np.asarray(res.params())
    okg, oko = rep["termination"][0] > 0, repo.termination > 0
    if okg != oko:
        print("FIT FLAG MISMATCH gpu %d oracle %d" % (rep["termination"][0], repo.termination), tag); bad += 1
    elif okg:
        da = np.abs(a[0] - ao).max() / np.abs(ao).max()
        do = abs(rep["objective"][0] - repo.objective) / max(repo.objective, 1e-300)
        if da > 1e-5 or do > 1e-6: print("FIT MISMATCH dalpha %.1e dobj %.1e evals %d/%d" % (da, do, rep["n_evals"][0], repo.n_evals), tag); bad += 1
    bp.close()
print("done: %d configurations, %d flagged" % (N, bad))
