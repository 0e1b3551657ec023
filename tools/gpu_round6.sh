#!/bin/bash
# round 6 evidence on the GPU box: rocprofv3 kernel stats of the bench's timed regime, then the PMC passes (FETCH_SIZE, WRITE_SIZE, each its own
# run: MI355X_MICROARCH.md "HBM"), summarised into gpurun_out/$NAME/ (copy what is to be judged into profiles/).  A ku_frames CALL is a chain of
# launches (the relay): the traffic per lane-frame is the SUM over a call's dispatches / the call's frames.  The summary is stamped with the hash
# of the kernels' sources (bench.py: csrc_hash) -- bench.py uses it only for the library it was measured on.
# usage: tools/gpu_round6.sh NAME [bench arguments]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export TMPDIR=/tmp S3A_ON_GPU_BOX=1
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
sys.path.insert(0, os.getcwd())
import bench
out = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(out, "bench_kernel_stats.csv")))) if os.path.exists(os.path.join(out, "bench_kernel_stats.csv")) else []
for r in rows[:10]:
    print("%-60s calls %7s avg %12.1f us total %9.1f ms %6s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].strip()
            agg.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})[c].append(float(row["Counter_Value"]))
line = {}
try: line = json.loads(open(os.path.join(out, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
except Exception as e: print("no bench line under the PMC pass:", e)
calls = int(line.get("steps", 2)) + int(line.get("warmup", 1))
FR = int(line.get("config", {}).get("frames_per_step", 0)) or 1155127
res = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two runs) of `bench.py --plain --no-cpu --steps 2 --warmup 1` (gpurun_out/{os.path.basename(out)}, tools/gpu_round6.sh): "
                 "KB at the L2's memory-side interface; FETCH_SIZE doubled for the streaming scoring kernel only (gfx950: wide coalesced reads are tallied at "
                 "half), scattered access patterns and WRITE_SIZE taken as they are (uncalibrated); ku_frames: the SUM over the dispatches of a call (the relay's "
                 "chain) / the call's lane-frames",
       "csrc_hash": bench.csrc_hash(), "calls": calls, "frames_per_call": FR, "bench_line_under_pmc": line, "kernels": {}}
for k, a in agg.items():
    fe, wr = a["FETCH_SIZE"], a["WRITE_SIZE"]
    corr = 2.0 if k == "ku_score_window" else 1.0
    res["kernels"][k] = {"dispatches": max(len(fe), len(wr)), "fetch_size_kb_total": round(sum(fe), 1), "write_size_kb_total": round(sum(wr), 1), "fetch_correction": corr,
                         "hbm_bytes_per_dispatch": int((corr * sum(fe) / max(len(fe), 1) + sum(wr) / max(len(wr), 1)) * 1024)}
if "ku_frames" in res["kernels"]:
    v = res["kernels"]["ku_frames"]
    v["dispatches_per_call"] = v["dispatches"] / calls
    v["fetch_bytes_per_lane_frame"] = round(v["fetch_size_kb_total"] * 1024 / calls / FR, 1)
    v["write_bytes_per_lane_frame"] = round(v["write_size_kb_total"] * 1024 / calls / FR, 1)
    v["hbm_bytes_per_lane_frame"] = round(v["fetch_bytes_per_lane_frame"] + v["write_bytes_per_lane_frame"], 1)
    print("ku_frames per lane-frame: fetch %.0f + write %.0f = %.0f B; dispatches per call %.1f" % (v["fetch_bytes_per_lane_frame"], v["write_bytes_per_lane_frame"], v["hbm_bytes_per_lane_frame"], v["dispatches_per_call"]))
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
PY
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
tail -c 600 $OUT/plain_under_rocprof.json
