"""cfg4: which fits end differently under two builds of the Gram kernel?
  python tools/cfg4_variant_diff.py run out.npz          (under VARPRO_HIP_LIBRARY=...: fit configs[4], save terminations / traces)
  python tools/cfg4_variant_diff.py diff a.npz b.npz     (problems whose success class differs; where their iterates part)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(out):
    import varpro_amd as vp
    from varpro_amd import synth
    B = 8192
    d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
    mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, C, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=100)
    r = bp.report_to_numpy(rep)
    np.savez(out, term=r["termination"], nev=r["n_evals"], obj=r["objective"], alpha=np.asarray(a), trace=np.asarray(tr))

def diff(fa, fb):
    A, Bz = np.load(fa), np.load(fb)
    sa, sb = A["term"] > 0, Bz["term"] > 0
    print("successes: %d vs %d; class differs for %d problems" % (sa.sum(), sb.sum(), (sa != sb).sum()))
    idx = np.nonzero(sa != sb)[0]
    q = A["alpha"].shape[1]
    for i in idx[:40]:
        ta, tb = A["trace"][i], Bz["trace"][i]
        n = min(A["nev"][i], Bz["nev"][i], ta.shape[0])
        k = 0
        while k < n and np.allclose(ta[k, :q], tb[k, :q], rtol=1e-6, atol=0):
            k += 1
        print("prob %5d: term %2d (%3d ev) vs %2d (%3d ev); iterates part at row %d" % (i, A["term"][i], A["nev"][i], Bz["term"][i], Bz["nev"][i], k))
        for nm, t, te, nv in (("A", ta, A["term"][i], A["nev"][i]), ("B", tb, Bz["term"][i], Bz["nev"][i])):
            last = min(nv, t.shape[0]) - 1
            for rr in range(max(0, min(k, last) - 1), min(last + 1, k + 3)):
                print("   %s row %2d: alpha %s fnorm1 %.6g ratio %.3g delta %.3g par %.3g" % (nm, rr, np.array2string(t[rr, :q], precision=5), t[rr, q], t[rr, q + 1], t[rr, q + 2], t[rr, q + 3]))
            print("   %s last row %d: alpha %s fnorm1 %.6g" % (nm, last, np.array2string(t[last, :q], precision=5), t[last, q]))

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        diff(sys.argv[2], sys.argv[3])
