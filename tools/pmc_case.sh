#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of ku_frames (a run each: MI355X_MICROARCH.md "HBM") for one build / setting of the library in the bench's plain regime.
# usage: tools/pmc_case.sh NAME label lib.so|- [bench arguments]   -> gpurun_out/NAME/pmc_<label>.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; LABEL=$2; LIB=$3; shift; shift; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
export TMPDIR=/tmp S3A_ON_GPU_BOX=1
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_tree_pmc.so
[ "$LIB" != "-" ] && cp cmusphinx_amd/$LIB cmusphinx_amd/libcmusphinx_amd.so
BENCH="python $R/bench.py --plain --no-cpu --steps 1 --warmup 1 $*"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_$c
    timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$c -o pmc -- $BENCH > $OUT/pmc_${LABEL}_$c.json 2> $OUT/pmc_${LABEL}_$c.err
done
cd $R
cp /tmp/lib_tree_pmc.so cmusphinx_amd/libcmusphinx_amd.so
python3 - $OUT $LABEL <<'PY'
import csv, glob, json, os, sys
out, label = sys.argv[1], sys.argv[2]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].strip()
            agg.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})[c].append(float(row["Counter_Value"]))
line = {}
try: line = json.loads(open(os.path.join(out, f"pmc_{label}_FETCH_SIZE.json")).read().strip().splitlines()[-1])
except Exception as e: print("no bench line under the PMC pass:", e)
FR = int(line.get("config", {}).get("frames_per_step", 0)) or 1155127
res = {"label": label, "value_under_pmc": line.get("value"), "frames_per_launch": FR, "kernels": {}}
for k, a in agg.items():
    fe, wr = a["FETCH_SIZE"], a["WRITE_SIZE"]
    # (the warm-up step's launch and the timed step's: the same work; the average)
    res["kernels"][k] = {"launches": max(len(fe), len(wr)), "fetch_kb_per_launch": round(sum(fe) / max(len(fe), 1), 1), "write_kb_per_launch": round(sum(wr) / max(len(wr), 1), 1)}
if "ku_frames" in res["kernels"]:
    v = res["kernels"]["ku_frames"]
    v["fetch_bytes_per_lane_frame"] = round(v["fetch_kb_per_launch"] * 1024 / FR, 1); v["write_bytes_per_lane_frame"] = round(v["write_kb_per_launch"] * 1024 / FR, 1)
    print(label, "ku_frames per lane-frame: fetch %.0f B + write %.0f B = %.0f B" % (v["fetch_bytes_per_lane_frame"], v["write_bytes_per_lane_frame"], v["fetch_bytes_per_lane_frame"] + v["write_bytes_per_lane_frame"]))
json.dump(res, open(os.path.join(out, f"pmc_{label}.json"), "w"), indent=1)
PY
rm -rf /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE
