"""A/B of library builds on ONE box with the bench's schedule (one batch at a time / two in flight): runs tools/refit_cost_probe.py in a
subprocess per library, alternating.  usage: python tools/ab_inflight.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, VARPRO_HIP_LIBRARY=os.path.abspath(lib))
        o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refit_cost_probe.py")], env=env, capture_output=True, text=True, cwd=ROOT)
        for ln in o.stdout.splitlines():
            if ln.startswith("refit"):
                print("%-28s %s" % (os.path.basename(lib), ln))
        if o.returncode != 0:
            print(os.path.basename(lib), "FAILED", o.stderr[-400:])
