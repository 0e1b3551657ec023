#!/bin/bash
# A synthetic task (cmusphinx_amd/synth_task.py) through the unmodified reference and through S3A_UTT=L lanes.
# usage: tools/utt_task.sh hub4|wsj|small N_UTT N_FRAMES "LANES..." [extra decoder args]   (SKIP_REF=1: no reference run)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
KIND=${1:-hub4}; NU=${2:-8}; NF=${3:-600}; LANES=${4:-"1 8"}; shift 4
T=/tmp/task_$KIND; O=gpurun_out/utt; mkdir -p $O
rm -rf $T; python -m cmusphinx_amd.synth_task $KIND $T n_utt=$NU n_frames=$NF > $T.args || exit 1
ARGS="$(cut -d';' -f2 $T.args) $@"
if [ -z "$SKIP_REF" ]; then
  s=$(date +%s%N); oracle/_ref/sphinx3_decode $ARGS -hyp $T/ref.match -hypseg $T/ref.seg > $T/ref.log 2>&1; e=$(date +%s%N)
  echo "reference rc=$? wall $(( (e - s) / 1000000 )) ms"; grep "^INFO: stat.c.*SUMMARY" $T/ref.log | cut -c1-200
fi
for L in $LANES; do
  s=$(date +%s%N); S3A_UTT=$L timeout 900 oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp $T/u$L.match -hypseg $T/u$L.seg > $O/task_$KIND.u$L.log 2>&1; rc=$?; e=$(date +%s%N)
  echo "$KIND lanes=$L rc=$rc wall $(( (e - s) / 1000000 )) ms $([ -z "$SKIP_REF" ] && cmp $T/u$L.match $T/ref.match && cmp $T/u$L.seg $T/ref.seg && echo IDENTICAL-to-reference)"
  grep "^INFO.*tst shim utt\|^INFO.*tst shim thr\|^INFO.*histogram\|^FATAL\|uttdec" $O/task_$KIND.u$L.log | cut -c24-330 | head -8
done
