#!/bin/bash
# rocprofv3 kernel statistics of the pocketsphinx first pass on the hub4-shaped task at the given lane counts
cd $(dirname $0)/..
R=$(pwd); O=$R/${1:-gpurun_out/psprof}; NU=${2:-64}; NF=${3:-1000}; shift 3
mkdir -p $O
D=/tmp/pstask
[ -f $D/ctl ] || python -m cmusphinx_amd.synth_task hub4 $D n_utt=$NU n_frames=$NF sorted_names=1 > $O/task.txt 2>&1
PSA="-mdef $D/mdef -mean $D/means -var $D/variances -mixw $D/mixture_weights -tmat $D/transition_matrices -senmgau .cont. -dict $D/dict -fdict $D/fillerdict -lm $D/lm.arpa -feat 1s_c -ceplen 39 -cmn none -agc none -varnorm no -cepdir $D/feat -cepext .mfc -fwdflat no -bestpath no"
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  # QUEUE=1: the whole control file as one queue over L lanes (what bench.py's ps_fwdtree leg runs); else the first L utterances in lock step
  if [ -n "$QUEUE" ]; then cp $D/ctl /tmp/ctl$L; Q="-queue yes"; else head -$L $D/ctl > /tmp/ctl$L; Q=""; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/l$L -o p -- $R/oracle/_ref/ref_ps_amdfwd $PSA -ctl /tmp/ctl$L -fresh yes -batch $L $Q -hyp /tmp/x.match > $O/prof$L.log 2>&1
  grep "ms on the device" $O/prof$L.log | head -1
  f=$(find $O/l$L -name "*kernel_stats.csv" | head -1)
  cp $f $O/psfwd_${L}lanes_kernel_stats.csv
  head -7 $f | cut -c1-160
  rm -rf $O/l$L
done
