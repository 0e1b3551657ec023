#!/bin/bash
# SQ / cache counters per kernel of the whole-utterance engine (bench workload, LANES lanes) -- rocprofv3 --pmc passes,
# no trace domains.  usage: tools/pmc_utt.sh NAME LANES
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-pmc}; LANES=${2:-128}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu --no-scoring --frames 120 --utts 32 --lanes $LANES"
pass() { # name counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  rocprofv3 --output-format csv --pmc "$@" -d /tmp/pmc_$n -o d -- $CMD > $OUT/pass_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split('(')[0][:28]
    if not k.startswith("ku_") and "ku_" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    n = max(len(cnt[k]), 1)
    print("%-28s launches %5d " % (k, n) + " ".join("%s/l %.0f" % (c, v / n) for c, v in sorted(agg[k].items())))
PY
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR > $OUT/sq.txt 2>&1
pass mem SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM > $OUT/mem.txt 2>&1
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum > $OUT/tcc.txt 2>&1
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum > $OUT/tcp.txt 2>&1
cat $OUT/sq.txt $OUT/mem.txt $OUT/tcc.txt $OUT/tcp.txt
grep -c . $OUT/list_avail.txt
