"""CPU oracle scaling on the host: fits/s for 1..N OpenMP threads (python tools/cpu_scaling_probe.py)"""
import os, sys, time
os.environ.setdefault("OMP_PLACES", "cores"); os.environ.setdefault("OMP_PROC_BIND", "spread")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
import varpro_amd as vp
from varpro_amd import synth
from oracle import oracle as O
d = synth.double_exp_batch(8192, m=1024, noise=1e-3)
mdl = vp.multi_exponential_model(d["x"], d["tau_guess"][0])
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    n = min(8192, 64 * nt)
    t = time.perf_counter(); a, c, rep, s = O.fit_batch(mdl, d["x"], d["Y"][:n], d["tau_guess"][:n], n_threads=nt); dt = time.perf_counter() - t
    print("%4d threads: %8.0f fits/s inside fits (%.0f per thread), wall %.3f s" % (nt, n / s, n / s / nt, dt), flush=True)
