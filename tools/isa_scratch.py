#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses are: counts scratch_load / scratch_store and vector-memory instructions of ONE function in a
`hipcc -save-temps -gline-tables-only` assembly file, attributed to the source line of the nearest `.loc` in FILE (inlined bodies count
for the line they were inlined at when --top is given: the innermost location of s3a_utt.hip).
usage: tools/isa_scratch.py file.s mangled_function_prefix [source file name] [phase table: 'name:line,name:line,...']"""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else "s3a_utt.hip"
phases = []
if len(sys.argv) > 4:
    for p in sys.argv[4].split(","):
        n, l = p.split(":"); phases.append((int(l), n))
    phases.sort()
files = {}
cur_file, cur_line = None, 0
in_fn = False
cnt = collections.Counter(); per_line = collections.defaultdict(collections.Counter)
loc_re = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
file_re = re.compile(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
top_line = 0
with open(path) as fh:
    for ln in fh:
        m = file_re.match(ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)); continue
        if not in_fn:
            if ln.startswith(fn) and ":" in ln:
                in_fn = True
            continue
        if ln.startswith(".Lfunc_end"):
            break
        m = loc_re.match(ln)
        if m:
            f = files.get(int(m.group(1)), "")
            if f.endswith(src):
                top_line = int(m.group(2))
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        kind = None
        if op.startswith("scratch_load"): kind = "scratch_load"
        elif op.startswith("scratch_store"): kind = "scratch_store"
        elif op.startswith(("global_load", "flat_load", "buffer_load")): kind = "vmem_load"
        elif op.startswith(("global_store", "flat_store", "buffer_store")): kind = "vmem_store"
        elif op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")): kind = "vmem_atomic"
        elif op.startswith("ds_"): kind = "lds"
        elif op.startswith(("v_writelane", "v_readlane")): kind = "sgpr_spill_lane"
        elif op.startswith("s_waitcnt"): kind = "waitcnt"
        elif op.startswith("s_barrier"): kind = "barrier"
        cnt["insts"] += 1; per_line[top_line]["insts"] += 1
        if kind:
            cnt[kind] += 1; per_line[top_line][kind] += 1
print("function", fn, dict(cnt))
if phases:
    agg = collections.defaultdict(collections.Counter)
    for line, c in per_line.items():
        name = "before"
        for l0, n in phases:
            if line >= l0: name = n
        agg[name].update(c)
    keys = ["insts", "scratch_load", "scratch_store", "sgpr_spill_lane", "vmem_load", "vmem_store", "vmem_atomic", "lds", "waitcnt", "barrier"]
    print("%-18s" % "phase" + "".join("%16s" % k for k in keys))
    for l0, n in [(0, "before")] + phases:
        c = agg.get(n)
        if c: print("%-18s" % n + "".join("%16d" % c[k] for k in keys))
else:
    for line, c in sorted(per_line.items(), key=lambda t: -(t[1]["scratch_load"] + t[1]["scratch_store"]))[:40]:
        print(line, dict(c))
