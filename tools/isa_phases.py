#!/usr/bin/env python3
"""Per-phase instruction counts of ku_frames from the code object: every instruction's address goes through llvm-symbolizer with inlined
frames, and an instruction belongs to the phase of the line of its kf_frame frame (the call chain's frame inside kf_frame), whatever was
inlined there.  usage: tools/isa_phases.py code_object mangled_kernel 'name:line,...' [out.json]
(build the code object with -gline-tables-only -save-temps=obj: the *.out file)"""
import json, re, subprocess, sys, collections
LLVM = "/opt/rocm/lib/llvm/bin/"
obj, kern = sys.argv[1], sys.argv[2]
phases = sorted((int(p.split(":")[1]), p.split(":")[0]) for p in sys.argv[3].split(","))
dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + kern, obj], capture_output=True, text=True).stdout
insts = []
for ln in dis.splitlines():
    m = re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]{12}):", ln)
    if m: insts.append((int(m.group(2), 16), m.group(1)))
sym = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + obj, "-i", "-f", "--output-style=JSON"], input="\n".join(hex(a) for a, _ in insts), capture_output=True, text=True).stdout
def kind(op):
    if op.startswith("scratch_load"): return "scratch_load"
    if op.startswith("scratch_store"): return "scratch_store"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "vmem_load"
    if op.startswith(("global_store", "flat_store", "buffer_store")): return "vmem_store"
    if op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")): return "vmem_atomic"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("v_writelane", "v_readlane")): return "sgpr_spill_lane"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    return None
agg = collections.defaultdict(collections.Counter)
for (addr, op), ln in zip(insts, sym.strip().splitlines()):
    fr = json.loads(ln).get("Symbol", [])
    name = "kernel_body"
    for s in fr:
        if "kf_frame" in s.get("FunctionName", ""):
            name = "before"
            for l0, n in phases:
                if s["Line"] >= l0: name = n
            break
    agg[name]["insts"] += 1
    k = kind(op)
    if k: agg[name][k] += 1
keys = ["insts", "scratch_load", "scratch_store", "sgpr_spill_lane", "vmem_load", "vmem_store", "vmem_atomic", "lds", "waitcnt", "barrier"]
print("%-14s" % "phase" + "".join("%14s" % k for k in keys))
tot = collections.Counter()
for n in [n for _, n in phases] + ["before", "kernel_body"]:
    c = agg.get(n)
    if c:
        print("%-14s" % n + "".join("%14d" % c[k] for k in keys)); tot.update(c)
print("%-14s" % "total" + "".join("%14d" % tot[k] for k in keys))
if len(sys.argv) > 4:
    json.dump({"kernel": kern, "phases": {n: dict(c) for n, c in agg.items()}, "total": dict(tot)}, open(sys.argv[4], "w"), indent=1)
