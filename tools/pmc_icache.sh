#!/bin/bash
# instruction-cache counters of ku_frames in the bench's regime (one rocprofv3 --pmc pass, no trace options)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-ic}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
make -s -C oracle oracle >/dev/null 2>&1
BENCH="python $R/bench.py --plain --no-cpu --steps 1 --warmup 1 $*"
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_IFETCH"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/prof_ic -o pmc -- $BENCH > $OUT/run_$tag.json 2> $OUT/run_$tag.err
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/prof_ic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ","").strip()
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k,v in agg.items():
    if k in ("ku_frames","ku_score_window"): print(k, {c:(x, cnt[(k,c)]) for c,x in v.items()})
PY
  rm -rf /tmp/prof_ic
done
