#!/bin/bash
# rocprofv3 kernel trace of a synthetic-task decode through the device path (single stream)
# usage: tools/prof_task.sh OUTNAME hub4|wsj N_UTT N_FRAMES [extra decoder args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/${1:-task}; KIND=${2:-hub4}; NU=${3:-2}; NF=${4:-600}; shift 4
mkdir -p $OUT
T=/tmp/task_$KIND
rm -rf $T; python -m cmusphinx_amd.synth_task $KIND $T n_utt=$NU n_frames=$NF > $T.args || exit 1
ARGS="$(cut -d';' -f2 $T.args) $@"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o decode -- $R/oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/prof.match > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/decode_kernel_stats.csv
grep "^INFO.*tst shim t" $OUT/prof.log | cut -c24-300 > $OUT/decode_timing.txt
rm -rf $OUT/prof
cat $OUT/decode_timing.txt
python3 - "$OUT/decode_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-56s calls %7s  avg %9.0f ns  total %7.1f ms  %5s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
