#!/bin/bash
# rocprofv3 kernel trace of a batched synthetic-task decode (N decoders sharing every launch)
# usage: tools/prof_batch.sh OUTNAME hub4|wsj N_UTT N_FRAMES N_DECODERS GROUPS
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/${1:-batch}; KIND=${2:-hub4}; NU=${3:-16}; NF=${4:-600}; ND=${5:-16}; NG=${6:-1}
mkdir -p $OUT
T=/tmp/task_$KIND
rm -rf $T; python -m cmusphinx_amd.synth_task $KIND $T n_utt=$NU n_frames=$NF > $T.args || exit 1
ARGS="$(cut -d';' -f2 $T.args)"
export TMPDIR=/tmp
cd /tmp
S3A_BATCH=$NG S3A_STREAMS=$ND rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o decode -- $R/oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/prof.match > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/batch_kernel_stats.csv
grep "^INFO.*tst shim t\|^INFO.*batched engine" $OUT/prof.log | cut -c24-300 > $OUT/batch_timing.txt
rm -rf $OUT/prof
cat $OUT/batch_timing.txt
python3 - "$OUT/batch_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-56s calls %7s  avg %9.0f ns  total %7.1f ms  %5s%%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
