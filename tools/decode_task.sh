#!/bin/bash
# Full mode-4 decode of a synthetic hub4- or WSJ-shaped task (cmusphinx_amd/synth_task.py): unmodified CPU
# reference vs the device path (fused frame) with N in-process decoder streams; outputs must be identical.
# usage: tools/decode_task.sh hub4|wsj|small N_UTT N_FRAMES "STREAMS..." [extra decoder args]
#        STREAMS entries: N = N decoders on N HIP streams; bN = N decoders batched into shared launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
KIND=${1:-hub4}; NU=${2:-8}; NF=${3:-1000}; STREAMS=${4:-"1 4"}; shift 4
T=/tmp/task_$KIND
rm -rf $T; python -m cmusphinx_amd.synth_task $KIND $T n_utt=$NU n_frames=$NF > $T.args || exit 1
ARGS="$(cut -d';' -f2 $T.args) $@"
s=$(date +%s%N); oracle/_ref/sphinx3_decode $ARGS -hyp $T/ref.match -hypseg $T/ref.seg > $T/ref.log 2>&1; e=$(date +%s%N)
echo "reference rc=$? wall $(( (e - s) / 1000000 )) ms"; grep "^INFO: stat.c.*SUMMARY" $T/ref.log | cut -c1-260
for N in $STREAMS; do
  B=0; case $N in b*) B=1; N=${N#b};; esac      # "b16" = 16 decoders sharing every launch (s3a_batch_*)
  case $N in *x*) B=${N#*x}; N=${N%x*};; esac     # "b16x2" = the same in 2 groups that alternate on the GPU
  s=$(date +%s%N); S3A_BATCH=$B S3A_STREAMS=$N oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp $T/s$N.match -hypseg $T/s$N.seg > $T/s$N.log 2>&1; rc=$?; e=$(date +%s%N)
  echo "streams=$N batch=$B rc=$rc wall $(( (e - s) / 1000000 )) ms $(cmp $T/s$N.match $T/ref.match && cmp $T/s$N.seg $T/ref.seg && echo IDENTICAL-to-reference)"
  grep "^INFO.*tst shim t\|^INFO.*histogram\|^INFO.*batched engine\|^FATAL\|^ERROR" $T/s$N.log | cut -c24-300 | head -6
done
