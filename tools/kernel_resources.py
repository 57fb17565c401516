"""Per-kernel code-object metadata (VGPRs, spilled VGPRs / SGPRs, scratch bytes, LDS) of a built object file or of
the whole library: `python tools/kernel_resources.py varpro_amd/csrc/build/vp_inst_ext_b_f64.o [substring]`
(llvm-objdump --offloading extracts the gfx950 code object, llvm-readelf --notes prints its metadata).
`--worst N` lists the N kernels with the most spilled VGPRs over all objects of varpro_amd/csrc/build."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    out = []
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), dst)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", dst], capture_output=True, cwd=td)
        for co in glob.glob(dst + ".*gfx950*"):
            txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
                out.append(dict(name=g("name"), vgpr=int(g("vgpr_count")), vspill=int(g("vgpr_spill_count")),
                                sspill=int(g("sgpr_spill_count")), scratch=int(g("private_segment_fixed_size")),
                                lds=int(g("group_segment_fixed_size"))))
    names = [k["name"] for k in out]
    if names:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        for k, d in zip(out, dem):
            k["dem"] = re.sub(r"\(.*", "", d).replace("void vp::", "").replace("vp::", "")
    return out


def main():
    if sys.argv[1] == "--worst":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
        allk = []
        for obj in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "varpro_amd", "csrc", "build", "*.o"))):
            for k in kernels_of(obj):
                k["obj"] = os.path.basename(obj)
                allk.append(k)
        allk.sort(key=lambda k: -k["vspill"])
        print("%d kernels; %d with spilled VGPRs, %d with more than 64" % (len(allk), sum(k["vspill"] > 0 for k in allk),
                                                                          sum(k["vspill"] > 64 for k in allk)))
        for k in allk[:n]:
            print("%-26s %-100s vgpr %3d vspill %4d sspill %3d scratch %5d" % (k["obj"], k["dem"][:100], k["vgpr"], k["vspill"], k["sspill"], k["scratch"]))
        return
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels_of(sys.argv[1]):
        if flt in k["dem"]:
            print("%-100s vgpr %3d vspill %4d sspill %3d scratch %5d lds %6d" % (k["dem"][:100], k["vgpr"], k["vspill"], k["sspill"], k["scratch"], k["lds"]))


if __name__ == "__main__":
    main()
