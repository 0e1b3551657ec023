#!/bin/bash
# what two lanes on a CU contend for: counter passes of ku_frames with 512 lanes (two per CU) and 256 lanes (one per CU) -- L2 hit rate, the address
# path's busy share, the waves' wait share.  usage: tools/pmc_diag.sh NAME   -> gpurun_out/NAME/pmc_diag.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-pmcd}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export TMPDIR=/tmp S3A_ON_GPU_BOX=1
cd /tmp
timeout 120 rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -o -E "\b(TCC_[A-Z0-9_]*(HIT|MISS|REQ)[A-Za-z0-9_]*|TA_[A-Z0-9_]*BUSY[A-Za-z0-9_]*|TCP_[A-Z0-9_]*(STALL|LATENCY|REQ)[A-Za-z0-9_]*|TD_[A-Z0-9_]*BUSY[A-Za-z0-9_]*|SQ_WAIT[A-Z0-9_]*|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_[A-Z0-9_]*|SQ_ACTIVE_INST_[A-Z0-9_]*|GRBM_GUI_ACTIVE|LDSBankConflict|SQ_LDS_[A-Z0-9_]*)\b" $OUT/counters_all.txt | sort -u > $OUT/counters_of_interest.txt
wc -l $OUT/counters_of_interest.txt
for lanes in 512 256; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/prof_d
    timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/prof_d -o pmc -- python $R/bench.py --plain --no-cpu --steps 1 --warmup 0 --lanes $lanes > $OUT/run_${lanes}_$tag.json 2> $OUT/run_${lanes}_$tag.err
    python3 - $lanes "$set" >> $OUT/pmc_diag.txt <<'PY'
import csv, glob, sys
lanes, names = sys.argv[1], sys.argv[2].split()
agg = {}
for f in glob.glob("/tmp/prof_d/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ku_frames" not in row["Kernel_Name"]: continue
        agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
print("lanes", lanes, {k: (sum(v) / len(v), len(v)) for k, v in agg.items()} if agg else "NO DATA for " + " ".join(names))
PY
  done
done
cat $OUT/pmc_diag.txt
