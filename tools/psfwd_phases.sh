#!/bin/bash
# per-phase time of the pocketsphinx search's frame (lane 0; a -DPSF_TIMING build of s3a_psfwd.hip in cmusphinx_amd/variants/lib_timing.so)
cd $(dirname $0)/..
O=gpurun_out/psphases; mkdir -p $O
D=/tmp/pstask
[ -f $D/ctl ] || python -m cmusphinx_amd.synth_task hub4 $D n_utt=1024 n_frames=1000 sorted_names=1 > $O/task.txt 2>&1
PSA="-mdef $D/mdef -mean $D/means -var $D/variances -mixw $D/mixture_weights -tmat $D/transition_matrices -senmgau .cont. -dict $D/dict -fdict $D/fillerdict -lm $D/lm.arpa -feat 1s_c -ceplen 39 -cmn none -agc none -varnorm no -cepdir $D/feat -cepext .mfc -fwdflat no -bestpath no"
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
cp cmusphinx_amd/variants/lib_timing.so cmusphinx_amd/libcmusphinx_amd.so
for L in "$@"; do
  head -$L $D/ctl > /tmp/ctl$L
  oracle/_ref/ref_ps_amdfwd $PSA -ctl /tmp/ctl$L -fresh yes -batch $L -hyp /tmp/x.match > $O/ph$L.log 2>&1
  echo "lanes $L: $(grep 'ms on the device' $O/ph$L.log | sed 's/^.*batch of/batch of/' | head -1)"
  grep PSF_TIMING $O/ph$L.log | head -2
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
