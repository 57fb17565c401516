"""VGPRs / spilled VGPRs / scratch / LDS of every kernel in a built object or library (code-object metadata).
usage: python tools/kernel_resources.py [varpro_amd/lib/libvarpro_hip.so] [--worst N] [--json out.json]"""
import json, os, re, subprocess, sys, tempfile, shutil
LLVM = "/opt/rocm/lib/llvm/bin"
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] not in ("--worst", "--json")]
obj = os.path.abspath(args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "varpro_amd", "lib", "libvarpro_hip.so"))
worst = int(sys.argv[sys.argv.index("--worst") + 1]) if "--worst" in sys.argv else 15
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
tmp = tempfile.mkdtemp()
try:
    shutil.copy(obj, os.path.join(tmp, "x.o"))
    blob = open(obj, "rb").read()
    if blob.find(b"CCOB") >= 0:
        # compressed offload bundles (--offload-compress): llvm-objdump --offloading mis-extracts a library that holds several
        # of them; cut each "CCOB" blob out (header: magic, u16 version, u16 method, u64 blob size, ...) and let
        # clang-offload-bundler decompress + unbundle it
        import struct
        i, n = blob.find(b"CCOB"), 0
        while i >= 0:
            version = struct.unpack_from("<H", blob, i + 4)[0]
            size = struct.unpack_from("<Q", blob, i + 8)[0] if version >= 3 else struct.unpack_from("<I", blob, i + 8)[0]
            open(os.path.join(tmp, "blob"), "wb").write(blob[i:i + size])
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + os.path.join(tmp, "blob"), "--output=" + os.path.join(tmp, "x.%d.amdgcn.co" % n)], capture_output=True)
            n += 1
            i = blob.find(b"CCOB", i + max(size, 4))
        os.remove(os.path.join(tmp, "blob"))
    else:
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", "x.o"], cwd=tmp, capture_output=True)
    kernels = []
    for fn in sorted(os.listdir(tmp)):
        if "amdgcn" not in fn: continue
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(tmp, fn)], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m: continue
            k, v = m.group(1), m.group(2).strip()
            if k == "name" and cur.get("_in_kernel"):
                cur["name"] = v
            if k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "sgpr_count"):
                cur[k] = int(v); cur["_in_kernel"] = True
            if k == "wavefront_size":
                if "name" in cur or "symbol" in cur: kernels.append(cur)
                cur = {}
            if k == "symbol": cur["symbol"] = v
    ks = [k for k in kernels if "vgpr_count" in k]
    def nm(k): return subprocess.run([shutil.which("c++filt") or "cat", k.get("symbol", k.get("name", "?")).replace(".kd", "")], capture_output=True, text=True).stdout.strip()[:150]
    sp = sorted(ks, key=lambda k: -k.get("vgpr_spill_count", 0))
    summary = {"object": os.path.relpath(obj), "bytes": os.path.getsize(obj), "compressed_bundles": blob.find(b"CCOB") >= 0,
               "code_object_bytes": sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if "amdgcn" in f), "kernels": len(ks),
               "kernels_with_spilled_vgprs": sum(1 for k in ks if k.get("vgpr_spill_count", 0) > 0),
               "kernels_above_64_spilled_vgprs": sum(1 for k in ks if k.get("vgpr_spill_count", 0) > 64),
               "worst": [dict(kernel=nm(k), vgprs=k["vgpr_count"], spilled_vgprs=k.get("vgpr_spill_count", 0), spilled_sgprs=k.get("sgpr_spill_count", 0),
                              scratch_bytes=k.get("private_segment_fixed_size", 0)) for k in sp[:worst]]}
    print(json.dumps({k: v for k, v in summary.items() if k != "worst"}))
    for w in summary["worst"]: print("%4d spilled  %3d VGPRs  %5d B scratch  %s" % (w["spilled_vgprs"], w["vgprs"], w["scratch_bytes"], w["kernel"]))
    if out_json: json.dump(summary, open(out_json, "w"), indent=1)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
