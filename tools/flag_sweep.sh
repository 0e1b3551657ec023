#!/bin/bash
# parity sweep: the reference decoder vs the device path (one decoder and batched) on tidigits under
# decoder options that change the search's control flow
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=tests/golden/tidigits_decode; AM=tests/golden/tidigits
COMMON="-dict $D/dictionary -fdict $D/fillerdict -hmm $AM -cepdir $D/cepstra -agc none -varnorm no -cmn current -lw 9.5 -ctl $D/tidigits.length.arb.regression -op_mode 4 -lm $D/tidigits.DMP"
i=0
while read -r FLAGS; do
  [ -z "$FLAGS" ] && continue
  i=$((i+1))
  oracle/_ref/sphinx3_decode $COMMON $FLAGS -hyp /tmp/fs_ref.match -hypseg /tmp/fs_ref.seg > /tmp/fs_ref.log 2>&1; r0=$?
  oracle/_ref/ref_s3amd_tst_decode $COMMON $FLAGS -hyp /tmp/fs_a.match -hypseg /tmp/fs_a.seg > /tmp/fs_a.log 2>&1; r1=$?
  S3A_STREAMS=4 S3A_BATCH=2 oracle/_ref/ref_s3amd_tst_decode $COMMON $FLAGS -hyp /tmp/fs_b.match -hypseg /tmp/fs_b.seg > /tmp/fs_b.log 2>&1; r2=$?
  a=DIFF; cmp -s /tmp/fs_a.match /tmp/fs_ref.match && cmp -s /tmp/fs_a.seg /tmp/fs_ref.seg && a=same
  b=DIFF; cmp -s /tmp/fs_b.match /tmp/fs_ref.match && cmp -s /tmp/fs_b.seg /tmp/fs_ref.seg && b=same
  echo "[$i] rc=$r0/$r1/$r2 one=$a batched=$b :: $FLAGS"
  [ "$r1" != 0 ] && grep "^FATAL" /tmp/fs_a.log | head -2 | cut -c1-300
done <<'LIST'
-ptranskip 1
-ptranskip 2 -beam 1e-80 -pbeam 1e-100 -wbeam 1e-40
-ptranskip 3
-wend_beam 1e-30
-maxwpf 2 -maxhistpf 5
-Nlextree 5 -epl 2
-Nlextree 1
-beam 1e-30 -pbeam 1e-30 -wbeam 1e-10
-beam 1e-200 -pbeam 1e-150 -wbeam 1e-120
-fillprob 0.5 -silprob 0.3
-wip 0.01 -silprob 0.05
-ds 3 -ci_pbeam 1e-3 -tighten_factor 0.3
-maxhmmpf 10 -maxwpf 3
-pl_window 3
-ctloffset 5 -ctlcount 7
-bestpath 1
LIST
