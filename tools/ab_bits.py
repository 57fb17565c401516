"""Bit-level comparison of library builds: fits the bench's 65 536 headline problems (and 4 096 at m = 1000, the general-length
slot kernel) with each library in its own process and prints a hash of (parameters, coefficients, reports).
usage: python tools/ab_bits.py libA.so libB.so ..."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import varpro_amd as vp
    from varpro_amd import synth
    for B, m, fg in ((65536, 1024, None), (16384, 1000, "slots"), (4096, 1024, None)):
        d = synth.double_exp_batch(B, m=m, noise=1e-3)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
        bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
        if fg and hasattr(bp, "set_fit_kernel"): bp.set_fit_kernel(fg)
        a, C, rep = bp.fit(d["tau_guess"])
        h = hashlib.sha256(np.ascontiguousarray(a).tobytes() + np.ascontiguousarray(C).tobytes() + np.ascontiguousarray(rep["n_evals"]).tobytes()
                           + np.ascontiguousarray(rep["objective"]).tobytes() + np.ascontiguousarray(rep["termination"]).tobytes()).hexdigest()[:16]
        print("B=%d m=%d: evals %d  sha %s" % (B, m, int(rep["n_evals"].sum()), h))
        bp.close()
    sys.exit(0)
for lib in [a for a in sys.argv[1:] if a.endswith(".so")]:
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, VARPRO_HIP_LIBRARY=os.path.abspath(lib), PYTHONPATH=ROOT),
                       capture_output=True, text=True, cwd=ROOT)
    for ln in o.stdout.splitlines():
        if ln.startswith("B="): print("%-28s %s" % (os.path.basename(lib), ln))
    if o.returncode != 0: print(os.path.basename(lib), "FAILED", o.stderr[-600:])
