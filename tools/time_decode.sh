#!/bin/bash
# wall-clock of RM1 decodes on the GPU box: unmodified reference vs scoring-only shim vs full-device search,
# single stream and P concurrent streams on one GPU.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
RM=tests/_local_data/rm1
ARGS="-mdef $RM/mdef -fdict $RM/fillerdict -dict $RM/RM.dictionary -mean $RM/means -var $RM/variances -mixw $RM/mixture_weights -tmat $RM/transition_matrices -agc none -varnorm no -cmn current -epl 4 -fillprob 0.02 -maxwpf 10 -wip 0.2 -lm $RM/RM.2845.trigram.arpa.DMP -lw 14 -beam 1e-140 -wbeam 1e-100 -cepdir $RM/feat -cepext .mfc -ctl $RM/rm.ctl -op_mode 4"
for b in sphinx3_decode ref_s3amd_decode ref_s3amd_tst_decode; do
  s=$(date +%s%N); oracle/_ref/$b $ARGS -hyp /tmp/$b.match > /tmp/$b.log 2>&1; e=$(date +%s%N)
  echo "$b: wall $(( (e - s) / 1000000 )) ms"; grep "^INFO: stat.c.*SUMMARY" /tmp/$b.log | cut -c1-260
done
for P in 4 16 32; do
  s=$(date +%s%N)
  for i in $(seq $P); do oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/p$i.match > /tmp/p$i.log 2>&1 & done; wait
  e=$(date +%s%N); echo "full-device x$P concurrent streams: wall $(( (e - s) / 1000000 )) ms for $P x 75.77 s of audio"
  cmp /tmp/p1.match /tmp/ref_s3amd_tst_decode.match && cmp /tmp/p$P.match /tmp/sphinx3_decode.match && echo "  hyps identical"
done
