"""cfg4 timeline (debug build with -DVP_FITG_TIMELINE=1: start / finish clock of every fit in status / cost, 100 MHz):
when do fits of which length start and finish, and what does a round cost as a function of the fit's length?
  make -C varpro_amd/csrc ab TAG=tline AB_SRCS="vp_api.hip vp_inst_generic.hip vp_inst_ext.hip vp_inst_ext_b_f64.hip vp_inst_me5_f32.hip vp_inst_blk_me_f32.hip" EXTRA=-DVP_FITG_TIMELINE=1
  VARPRO_HIP_LIBRARY=varpro_amd/lib/ab/libvarpro_hip_tline.so python tools/cfg4_timeline.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import varpro_amd as vp
from varpro_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = synth.multi_exp_batch(B, 5, 4096, [0.5, 1.5, 3.0, 6.0, 12.0], noise=1e-3, spread=0.1, guess_spread=0.05, dtype=np.float32)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0], dtype=np.float32)
dev = torch.device("cuda", 0)
Y = torch.from_numpy(d["Y"]).to(dev); g = torch.from_numpy(d["tau_guess"]).to(dev)
bp = vp.BatchProblem(mdl, Y, x=torch.from_numpy(d["x"]).to(dev))
bp.set_timing(True)
for _ in range(3):
    a, C, rep = bp.fit(g)
ms = bp.last_kernel_ms(_lib.VP_KERNEL_FIT)
r = bp.report_to_numpy(rep)
ne = r["n_evals"].astype(np.int64)
t1 = bp.cost().cpu().numpy() if hasattr(bp.cost(), "cpu") else np.asarray(bp.cost())
t0 = bp.status().cpu().numpy() if hasattr(bp.status(), "cpu") else np.asarray(bp.status())
t0 = t0.astype(np.float64); t1 = t1.astype(np.float64)
base = t0.min()
s_us = (t0 - base) / 100.0; f_us = (t1 - base) / 100.0
dur = f_us - s_us
print("kernel %.3f ms; last finish %.1f us; fits started after t=10us: %d" % (ms, f_us.max(), int((s_us > 10).sum())))
print("%8s %6s %10s %10s %10s" % ("evals", "fits", "us/round", "start us", "finish us"))
for lo, hi in ((1, 8), (8, 16), (16, 24), (24, 32), (32, 48), (48, 64), (64, 1000)):
    k = (ne >= lo) & (ne < hi)
    if k.any():
        print("%3d-%-4d %6d %10.1f %10.1f %10.1f" % (lo, hi, k.sum(), (dur[k] / ne[k]).mean(), s_us[k].mean(), f_us[k].mean()))
order = np.argsort(-f_us)[:12]
print("the last fits to finish: (prob, evals, start, finish, us/round)")
for i in order:
    print("  %6d %4d %8.1f %8.1f %6.1f" % (i, ne[i], s_us[i], f_us[i], dur[i] / ne[i]))
# share of fits finished over time
for t in (250, 500, 750, 1000, 1250, 1500, 1750, 2000, 2250):
    print("t=%5d us: %5.1f%% finished, %d still running" % (t, 100.0 * (f_us <= t).mean(), int((f_us > t).sum())))

# per-wave activity (host-pointer handle: the debug build writes it into the trace buffer)
bh = vp.BatchProblem(mdl, d["Y"], x=d["x"])
a, C, rep, tr = bh.fit_trace(d["tau_guess"], max_rows=8)
flat = tr.reshape(-1)
nblk = min(256, (B + 31) // 32 if B >= 256 * 32 else 256)
st = flat[: 256 * 8 * 8].reshape(256, 8, 8)
sc = st[:, 0, :]; ok = sc[:, 0] == 0.0
print("scalar waves (%d): total %.0f us, busy %.0f us (%.0f%%), trips %.1f, slots served per trip %.2f, us per trip %.1f" % (
    ok.sum(), sc[ok, 1].mean() / 100, sc[ok, 2].mean() / 100, 100 * sc[ok, 2].sum() / sc[ok, 1].sum(), sc[ok, 3].mean(),
    sc[ok, 4].sum() / sc[ok, 3].sum(), sc[ok, 2].sum() / sc[ok, 3].sum() / 100))
sw = st[:, 1:, :].reshape(-1, 8); ok = sw[:, 0] == 1.0
sw = sw[ok]
print("stream waves (%d): total %.0f us, busy %.0f us (%.0f%%), whole passes %.1f, parts %.1f, own bookkeeping %.0f us; us per pass-or-part (incl. own bookkeeping) %.1f" % (
    len(sw), sw[:, 1].mean() / 100, sw[:, 2].mean() / 100, 100 * sw[:, 2].sum() / sw[:, 1].sum(), sw[:, 3].mean(), sw[:, 4].mean(),
    sw[:, 5].mean() / 100, sw[:, 2].sum() / (sw[:, 3].sum() + sw[:, 4].sum()) / 100))
