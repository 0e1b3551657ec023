#!/bin/bash
# The pocketsphinx first pass on the device against the unmodified pocketsphinx, on the GPU box (everything it
# needs travels with the snapshot: oracle/_ref binaries, tests/golden, tests/_local_data).  Prints one line per case.
cd $(dirname $0)/..
G=tests/golden; L=tests/_local_data/ps; O=${1:-gpurun_out/psfwd}; B=oracle/_ref
mkdir -p $O
awk '{print $1}' $G/tidigits_decode/tidigits.length.arb.regression > $O/tdc.ctl
printf "goforward\nnumbers\nsomething\n" > $O/raw.ctl
CONT="-mdef $G/tidigits/mdef -mean $G/tidigits/means -var $G/tidigits/variances -mixw $G/tidigits/mixture_weights -tmat $G/tidigits/transition_matrices -senmgau .cont. -topn 4 -dict $G/tidigits_decode/tidigits.ps.dic -fdict $G/tidigits_decode/fillerdict -lm $G/tidigits_decode/tidigits.DMP -ctl $O/tdc.ctl -cepdir $G/tidigits_decode/cepstra"
SC="-hmm $G/ps_tidigits_sc -lm $G/tidigits_decode/tidigits.DMP -dict $G/ps_tidigits_sc/tidigits.dic -ctl $G/ps_tidigits_sc/tidigits.ctl -cepdir $G/ps_tidigits_sc/cepstra"
TU="-hmm $L/hub4wsj_sc_8k -lm $G/ps_turtle/turtle.DMP -dict $G/ps_turtle/turtle.dic -ctl $O/raw.ctl -cepdir $L/raw -cepext .raw -adcin yes"
ZH="-hmm $L/tdt_sc_8k -lm $L/zh_CN/gigatdt.5000.DMP -dict $L/zh_CN/mandarin_notone.dic -ctl $O/raw.ctl -cepdir $L/raw -cepext .raw -adcin yes"
run() { # tag refargs... -- amdextra...
  tag=$1; shift; ref=(); while [ "$1" != "--" ]; do ref+=("$1"); shift; done; shift
  $B/ref_ps_fwd "${ref[@]}" -hyp $O/$tag.ref.match -hypseg $O/$tag.ref.seg -bpdump $O/$tag.ref.bp > $O/$tag.ref.log 2>&1 || echo "$tag: ref FAILED"
  t0=$(date +%s.%N)
  timeout 900 $B/ref_ps_amdfwd "${ref[@]}" "$@" -hyp $O/$tag.amd.match -hypseg $O/$tag.amd.seg -bpdump $O/$tag.amd.bp > $O/$tag.amd.log 2>&1 || { echo "$tag: amd FAILED: $(grep -E 'ERROR|FATAL' $O/$tag.amd.log | tail -2)"; return; }
  awk -v a=$t0 -v b=$(date +%s.%N) 'BEGIN { printf "%.1f s\n", b - a }' > $O/$tag.amd.time
  m=ok; cmp -s $O/$tag.ref.match $O/$tag.amd.match || m=MATCH-DIFF
  s=ok; cmp -s $O/$tag.ref.seg $O/$tag.amd.seg || s=SEG-DIFF
  b=$(python3 tests/psfwd_dump.py $O/$tag.ref.bp $O/$tag.amd.bp | head -3 | tr '\n' ';')
  echo "$tag: match $m seg $s bp $b ($(cat $O/$tag.amd.time))"
}
run cont_sync $CONT -fwdflat no -bestpath no --
run cont_default $CONT --
run cont_fresh_b1 $CONT -fwdflat no -bestpath no -fresh yes -- -batch 1
run cont_fresh_b8 $CONT -fwdflat no -bestpath no -fresh yes -- -batch 8
run cont_fresh_b31 $CONT -fwdflat no -bestpath no -fresh yes -- -batch 31
run cont_fresh_b31_all $CONT -fwdflat no -bestpath no -fresh yes -compallsen yes -- -batch 31
run sc_default $SC --
run sc_sync $SC -fwdflat no -bestpath no --
run turtle $TU -fwdflat no -bestpath no --
if [ "$2" != "quick" ]; then
run zh $ZH -fwdflat no -bestpath no --
run zh_pruned $ZH -fwdflat no -bestpath no -maxhmmpf 800 -maxwpf 5 -beam 1e-60 -wbeam 1e-30 --
run zh_default $ZH --
fi
