#!/bin/bash
# rocprofv3 kernel trace of one full RM1 decode through the fused device frame (single stream):
# per-kernel GPU time per frame.  Output: gpurun_out/$1/decode_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-decode}
mkdir -p $OUT
RM=$R/tests/_local_data/rm1
ARGS="-mdef $RM/mdef -fdict $RM/fillerdict -dict $RM/RM.dictionary -mean $RM/means -var $RM/variances -mixw $RM/mixture_weights -tmat $RM/transition_matrices -agc none -varnorm no -cmn current -epl 4 -fillprob 0.02 -maxwpf 10 -wip 0.2 -lm $RM/RM.2845.trigram.arpa.DMP -lw 14 -beam 1e-140 -wbeam 1e-100 -cepdir $RM/feat -cepext .mfc -ctl $RM/rm.ctl -ctlcount ${2:-20} -op_mode 4"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o decode -- $R/oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/prof.match > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/decode_kernel_stats.csv
grep "^INFO.*tst shim t" $OUT/prof.log | cut -c24-300 > $OUT/decode_timing.txt
grep "^INFO: stat.c.*SUMMARY" $OUT/prof.log | cut -c1-250 >> $OUT/decode_timing.txt
rm -rf $OUT/prof
cat $OUT/decode_timing.txt
python3 - "$OUT/decode_kernel_stats.csv" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print("%-60s calls %8s  avg %9.0f ns  total %6.1f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
EOF
