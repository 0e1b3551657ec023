#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of a library's kernels from the code object's notes (llvm-readelf --notes of the bundled gfx950
code object).  usage: tools/kernel_resources.py lib.so [name-substring ...] [--json out.json]"""
import json, re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin/"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out: args.remove(out)
lib, subs = args[0], args[1:]
notes = ""
with tempfile.TemporaryDirectory() as d:
    # the fat binary (section .hip_fatbin of an object or of the shared library: one bundle per translation unit, back to back)
    fb = os.path.join(d, "fb")
    subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
    blob = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    for k in range(len(at) - 1):
        piece, co = os.path.join(d, "p%d" % k), os.path.join(d, "co%d" % k)
        open(piece, "wb").write(blob[at[k]:at[k + 1]])
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co):
            notes += subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
res = {}
cur = {}
for ln in notes.splitlines():
    m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", ln)
    if not m: continue
    k, v = m.group(1), m.group(2).strip().strip("'")
    if k == "agpr_count" and cur.get("name"):
        pass
    if ln.lstrip().startswith("- ") and cur.get("name"):
        res[cur["name"]] = cur; cur = {}
    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
        cur[k] = v if k == "name" else int(v)
if cur.get("name"): res[cur["name"]] = cur
sel = {}
for n, c in res.items():
    dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if subs and not any(s in dem for s in subs): continue
    c = dict(c); c.pop("name"); sel[dem] = c
for n, c in sorted(sel.items()):
    print("%-48s vgpr %3d spill %3d | sgpr %3d spill %3d | scratch %4d B | lds %6d B" % (n[:48], c.get("vgpr_count", 0), c.get("vgpr_spill_count", 0), c.get("sgpr_count", 0), c.get("sgpr_spill_count", 0), c.get("private_segment_fixed_size", 0), c.get("group_segment_fixed_size", 0)))
if out:
    json.dump({"library": os.path.basename(lib), "source": "llvm-readelf --notes of the library's gfx950 code object (tools/kernel_resources.py)", "kernels": sel}, open(out, "w"), indent=1, sort_keys=True)
