#!/bin/bash
# aggregate full-decode throughput of N decoder streams (host threads + HIP streams) in ONE process on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
RM=tests/_local_data/rm1
ARGS="-mdef $RM/mdef -fdict $RM/fillerdict -dict $RM/RM.dictionary -mean $RM/means -var $RM/variances -mixw $RM/mixture_weights -tmat $RM/transition_matrices -agc none -varnorm no -cmn current -epl 4 -fillprob 0.02 -maxwpf 10 -wip 0.2 -lm $RM/RM.2845.trigram.arpa.DMP -lw 14 -beam 1e-140 -wbeam 1e-100 -cepdir $RM/feat -cepext .mfc -ctl $RM/rm.ctl -op_mode 4"
oracle/_ref/sphinx3_decode $ARGS -hyp /tmp/ref.match -hypseg /tmp/ref.seg > /tmp/ref.log 2>&1
grep "^INFO: stat.c.*SUMMARY" /tmp/ref.log | cut -c1-250
for N in ${@:-1 2 4 8 16}; do
  B=0; case $N in b*) B=1; N=${N#b};; esac      # "b16" = 16 decoders sharing every launch (s3a_batch_*)
  case $N in *x*) B=${N#*x}; N=${N%x*};; esac     # "b16x2" = the same in 2 groups that alternate on the GPU
  S3A_BATCH=$B S3A_STREAMS=$N oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/s$N.match -hypseg /tmp/s$N.seg > /tmp/s$N.log 2>&1
  echo "streams=$N batch=$B rc=$? $(cmp /tmp/s$N.match /tmp/ref.match && cmp /tmp/s$N.seg /tmp/ref.seg && echo IDENTICAL-to-reference)"
  grep "^INFO.*tst shim t\|^INFO.*batched engine" /tmp/s$N.log | cut -c24-300
done
