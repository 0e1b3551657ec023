gcc -shared -fPIC -o /tmp/segv.so tools/segv_trace.c
R=$(pwd); D=$R/tests/golden/tidigits_decode; AM=$R/tests/golden/tidigits; C=$R/tests/golden/tidigits_clm
mkdir -p /tmp/clm && cd /tmp/clm
printf "{ $C/digits.probdef }\n$C/digits.cls.lm digitclass {\n[low]\n[high]\n}\n$D/tidigits.DMP plain\n" > lmctl
awk '{print (NR%3==0) ? "plain" : "digitclass"}' $D/tidigits.length.arb.regression > ctl_lm
S3A_UTT=4 LD_PRELOAD=/tmp/segv.so $R/oracle/_ref/ref_s3amd_tst_decode -dict $D/dictionary -fdict $D/fillerdict -hmm $AM -cepdir $D/cepstra -agc none -varnorm no -cmn current -lw 9.5 -ctl $D/tidigits.length.arb.regression -op_mode 4 -lmctlfn /tmp/clm/lmctl -ctl_lm /tmp/clm/ctl_lm -lmname plain -hyp /tmp/clm/g.match -hypseg /tmp/clm/g.seg 2>&1 | tail -25
