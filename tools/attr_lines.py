"""Attribute the instructions of one kernel in a -gline-tables-only .s file to source (file, function-ish line ranges).
usage: attr_lines.py file.s '<mangled-prefix>' """
import re, sys, collections
path, prefix = sys.argv[1], sys.argv[2]
files = {}
cur = None; inside = False
cnt = collections.Counter(); valu = collections.Counter()
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"', line)
    if m:
        files[int(m.group(1))] = m.group(m.lastindex).split("/")[-1]
        continue
    if not inside:
        if line.startswith(prefix) and line.rstrip().endswith(":") or (line.startswith(prefix) and ":" in line[:len(prefix) + 200] and "@" in line):
            inside = True
        continue
    if re.match(r"\s*s_endpgm", line):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+([vsdb][a-z_0-9]+)", line)
    if m and cur:
        cnt[cur] += 1
        if m.group(1).startswith("v_"): valu[cur] += 1
# bucket by file + 10-line ranges
by = collections.Counter(); byv = collections.Counter()
for (f, l), n in cnt.items():
    by[(f, l)] += n; byv[(f, l)] += valu[(f, l)]
tot = sum(by.values()); totv = sum(byv.values())
print("total instrs", tot, "VALU", totv)
perfile = collections.Counter()
for (f, l), n in byv.items(): perfile[f] += n
for f, n in perfile.most_common(): print("  %-20s VALU %5d" % (f, n))
print("top lines:")
for (f, l), n in byv.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("  %-18s:%4d  VALU %5d  all %5d" % (f, l, n, by[(f, l)]))
