#!/bin/bash
# round 5 evidence on the GPU box: rocprofv3 kernel stats of the bench's timed regime, then the PMC passes (FETCH_SIZE, WRITE_SIZE,
# each its own run: MI355X_MICROARCH.md "HBM"), summarised into gpurun_out/$NAME/ (copy what is to be judged into profiles/).
# usage: tools/gpu_round5.sh NAME [bench arguments]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r5}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
BENCH="python $R/bench.py --plain --no-cpu --steps 2 --warmup 1 $*"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH > $OUT/plain_under_rocprof.json 2> $OUT/plain_under_rocprof.err
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv
rm -rf /tmp/prof_kt
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -o pmc -- $BENCH > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(out, "bench_kernel_stats.csv")))) if os.path.exists(os.path.join(out, "bench_kernel_stats.csv")) else []
for r in rows[:12]:
    print("%-60s calls %7s avg %12.1f us total %9.1f ms %6s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].strip()
            a = agg.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
            a[c].append(float(row["Counter_Value"]))
line = {}
try:
    line = json.loads(open(os.path.join(out, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line under the PMC pass:", e)
frames = None
res = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two runs) of `bench.py --plain --no-cpu --steps 2 --warmup 1` (gpurun_out/{os.path.basename(out)}): "
                 "KB at the L2's memory-side interface, per dispatch; FETCH_SIZE doubled for the streaming scoring kernel only (gfx950: wide coalesced "
                 "reads are tallied at half), scattered access patterns and WRITE_SIZE taken as they are (uncalibrated)",
       "bench_line_under_pmc": line, "kernels": {}}
for k, a in agg.items():
    fe, wr = a["FETCH_SIZE"], a["WRITE_SIZE"]
    corr = 2.0 if k == "ku_score_window" else 1.0
    res["kernels"][k] = {"launches": max(len(fe), len(wr)), "fetch_size_kb_per_launch": round(sum(fe) / max(len(fe), 1), 1), "write_size_kb_per_launch": round(sum(wr) / max(len(wr), 1), 1),
                         "fetch_correction": corr, "hbm_bytes_per_launch": int((corr * sum(fe) / max(len(fe), 1) + sum(wr) / max(len(wr), 1)) * 1024)}
FR = 1155127            # frames of the bench's batch (a ku_frames launch decodes all of them)
if "ku_frames" in res["kernels"]:
    v = res["kernels"]["ku_frames"]
    v["lane_frames_per_launch"] = FR
    v["hbm_bytes_per_lane_frame"] = round(v["hbm_bytes_per_launch"] / FR, 1)
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda t: -t[1]["hbm_bytes_per_launch"])[:8]:
    print(k, v)
PY
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
tail -c 400 $OUT/plain_under_rocprof.json
