"""Per-section cycle profile of the fit kernel (diagnostic library built with -DVP_FIT_CLOCKS:
`make -C varpro_amd/csrc ab TAG=clk EXTRA=-DVP_FIT_CLOCKS`).  Runs B fits and prints, per LM iteration, the mean
wave-clock cycles each section took (s_memtime deltas; with B <= #SIMDs every wave runs alone, with B large the
sections include the time the wave waited for its SIMD neighbour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VARPRO_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         "varpro_amd", "lib", "ab", "libvarpro_hip_clk.so"))
import numpy as np
import varpro_amd as vp
from varpro_amd import synth

NAMES = ["park+load y", "build columns", "house_qr", "solve+norm", "LM accept/terminate", "jacobian + jac_qrfac",
         "gnorm/diag", "lmpar", "prered/next step", "-", "-", "-"]
CFG4 = "cfg4" in sys.argv  # fp32 five exponentials + offset, m = 4096, four waves per problem
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [256, 65536]:
    if CFG4:
        d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05,
                                  dtype=np.float32)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
    else:
        d = synth.double_exp_batch(B, m=1024, noise=1e-3)
        mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
    bp = vp.BatchProblem(mdl, d["Y"], x=d["x"])
    a, c, rep, tr = bp.fit_trace(d["tau_guess"], max_rows=4)   # 4 rows x (q+4)=6 -> 24 doubles >= 12
    clk = tr.reshape(B, -1)[:, :12].astype(np.float64)
    ne = rep["n_evals"].astype(np.float64)
    per_iter = clk.sum(0) / ne.sum()
    tot = per_iter.sum()
    print("B = %d   mean evaluations/fit %.2f   cycles per LM iteration (100 MHz s_memtime ticks x ?): total %.0f" % (B, ne.mean(), tot))
    for k in range(9):
        print("   %-24s %9.1f  %5.1f %%" % (NAMES[k], per_iter[k], 100 * per_iter[k] / tot))
    bp.close()
