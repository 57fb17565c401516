"""Print per-kernel VGPR / scratch / occupancy / code size from a gfx950 .s file (hipcc -save-temps)."""
import re, subprocess, sys
name = None
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m: name = m.group(1)
    m = re.match(r"; (NumVgprs|ScratchSize|Occupancy|codeLenInByte)\W+(\d+)", line)
    if m and name:
        rows.append((name, m.group(1), int(m.group(2))))
import collections
d = collections.OrderedDict()
for n, k, v in rows: d.setdefault(n, {})[k] = v
names = list(d)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, dn in zip(names, dem):
    if flt and flt not in dn: continue
    short = re.sub(r"\(.*", "", dn).replace("void vp::", "").replace("vp::", "")
    v = d[n]
    print("%-88s vgpr %3d scratch %4d occ %d code %6d" % (short[:88], v.get("NumVgprs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1), v.get("codeLenInByte", -1)))
