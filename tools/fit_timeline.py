"""Kernel timeline of the LAST global fit in a rocprofv3 --kernel-trace CSV (tools/ktrace.sh <tag> tools/mrhs_probe.py):
python tools/fit_timeline.py gpurun_out/ktrace_<tag>/t/<tag>_kernel_trace.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(?:vp::)?(?:gen::)?([A-Za-z0-9_]+)", n)
    return m.group(1) if m else n[:30]
idx = [i for i, r in enumerate(rows) if 'mrhs_finish' in r['Kernel_Name']]
e = idx[-1]; s = idx[-2] + 1
t0 = None; prev = None
for r in rows[s:e + 1]:
    st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    if t0 is None: t0 = st
    print("%-28s start %8.1f dur %6.1f gap %6.1f" % (short(r['Kernel_Name']), (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3 if prev else 0)); prev = en
