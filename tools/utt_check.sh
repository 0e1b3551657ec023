#!/bin/bash
# Whole-utterance device decoding (S3A_UTT=L) against the unmodified reference: tidigits, RM1 and a synthetic task.
# usage: tools/utt_check.sh "LANES..." [task_kind n_utt n_frames [extra decoder args]]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
LANES=${1:-"1 4"}; KIND=${2:-none}; NU=${3:-4}; NF=${4:-300}; shift 4
O=gpurun_out/utt; mkdir -p $O
D=tests/golden/tidigits_decode; AM=tests/golden/tidigits
TID="-dict $D/dictionary -fdict $D/fillerdict -hmm $AM -cepdir $D/cepstra -agc none -varnorm no -cmn current -lw 9.5 -ctl $D/tidigits.length.arb.regression -op_mode 4 -lm $D/tidigits.DMP"
RM=tests/_local_data/rm1
RMA="-mdef $RM/mdef -fdict $RM/fillerdict -dict $RM/RM.dictionary -mean $RM/means -var $RM/variances -mixw $RM/mixture_weights -tmat $RM/transition_matrices -agc none -varnorm no -cmn current -epl 4 -fillprob 0.02 -maxwpf 10 -wip 0.2 -lm $RM/RM.2845.trigram.arpa.DMP -lw 14 -beam 1e-140 -wbeam 1e-100 -cepdir $RM/feat -cepext .mfc -ctl $RM/rm.ctl -ctlcount 20 -op_mode 4"
run() { # name args...
  local name=$1; shift
  oracle/_ref/sphinx3_decode "$@" -hyp $O/$name.ref.match -hypseg $O/$name.ref.seg > $O/$name.ref.log 2>&1
  for L in $LANES; do
    s=$(date +%s%N); S3A_UTT=$L timeout 600 oracle/_ref/ref_s3amd_tst_decode "$@" -hyp $O/$name.u$L.match -hypseg $O/$name.u$L.seg > $O/$name.u$L.log 2>&1; rc=$?; e=$(date +%s%N)
    echo "$name lanes=$L rc=$rc wall $(( (e - s) / 1000000 )) ms $(cmp $O/$name.u$L.match $O/$name.ref.match && cmp $O/$name.u$L.seg $O/$name.ref.seg && echo IDENTICAL-to-reference)"
    grep "^INFO.*tst shim utt\|^INFO.*tst shim thr\|^INFO.*histogram\|^FATAL\|^ERROR.*shim\|uttdec" $O/$name.u$L.log | cut -c1-330 | head -8
  done
}
run tidigits $TID
run tidigits_hp20 $TID -maxhmmpf 20
[ -d $RM ] && run rm1 $RMA
[ -d $RM ] && run rm1_hp800 $RMA -maxhmmpf 800
if [ $KIND != none ]; then
  T=/tmp/task_$KIND; rm -rf $T; python -m cmusphinx_amd.synth_task $KIND $T n_utt=$NU n_frames=$NF > $T.args || exit 1
  run task_$KIND $(cut -d';' -f2 $T.args) "$@"
fi
