#!/bin/bash
# the pocketsphinx first pass on the hub4-shaped synthetic task: lanes sweep on the device + the unmodified decoder on CPU
cd $(dirname $0)/..
O=${1:-gpurun_out/psbench}; NU=${2:-64}; NF=${3:-1000}; shift 3
mkdir -p $O
D=/tmp/pstask
[ -f $D/ctl ] || python -m cmusphinx_amd.synth_task hub4 $D n_utt=$NU n_frames=$NF sorted_names=1 > $O/task.txt 2>&1
PSA="-mdef $D/mdef -mean $D/means -var $D/variances -mixw $D/mixture_weights -tmat $D/transition_matrices -senmgau .cont. -dict $D/dict -fdict $D/fillerdict -lm $D/lm.arpa -feat 1s_c -ceplen 39 -cmn none -agc none -varnorm no -cepdir $D/feat -cepext .mfc -ctl $D/ctl -fwdflat no -bestpath no"
head -4 $D/ctl > $O/ctl4
t0=$(date +%s.%N); oracle/_ref/ref_ps_fwd ${PSA/-ctl $D\/ctl/-ctl $O\/ctl4} -fresh yes -hyp $O/ref4.match -hypseg $O/ref4.seg > $O/ref4.log 2>&1; t1=$(date +%s.%N)
grep -E "AVERAGE|TOTAL" $O/ref4.log | tail -3
awk -v a=$t0 -v b=$t1 'BEGIN { printf "reference, 4 utterances, new decoder each: %.1f s wall\n", b - a }'
for L in "$@"; do
  oracle/_ref/ref_ps_amdfwd $PSA -fresh yes -batch $L -hyp $O/amd$L.match -hypseg $O/amd$L.seg > $O/amd$L.log 2>&1 || { echo "lanes $L FAILED"; grep -E "ERROR|FATAL" $O/amd$L.log | tail -3; continue; }
  grep "ms on the device" $O/amd$L.log | sed "s/^.*batch of/lanes $L: batch of/" | head -3
  head -4 $O/amd$L.match | cmp -s - $O/ref4.match && echo "lanes $L: first 4 hypotheses identical to the reference's" || echo "lanes $L: HYP DIFF"
  head -4 $O/amd$L.seg | cmp -s - $O/ref4.seg || echo "lanes $L: SEG DIFF"
done
# the whole control file as one queue over Q lanes: QUEUE="256 512" bash tools/psfwd_bench.sh ...
for L in $QUEUE; do
  oracle/_ref/ref_ps_amdfwd $PSA -fresh yes -batch $L -queue yes -hyp $O/q$L.match -hypseg $O/q$L.seg > $O/q$L.log 2>&1 || { echo "queue lanes $L FAILED"; grep -E "ERROR|FATAL" $O/q$L.log | tail -3; continue; }
  grep "ms on the device" $O/q$L.log | sed "s/^.*batch of/queue lanes $L: batch of/" | head -3
  head -4 $O/q$L.match | cmp -s - $O/ref4.match && echo "queue lanes $L: first 4 hypotheses identical to the reference's" || echo "queue lanes $L: HYP DIFF"
  head -4 $O/q$L.seg | cmp -s - $O/ref4.seg || echo "queue lanes $L: SEG DIFF"
done
