#!/bin/bash
# SQ counters per kernel of a batched decode (16 decoders, one group) -- rocprofv3 --pmc pass, no trace domains
R=$GRAFT_REPO_ROOT; cd $R
T=/tmp/task_hub4; rm -rf $T; python -m cmusphinx_amd.synth_task hub4 $T n_utt=16 n_frames=300 > $T.args
ARGS="$(cut -d';' -f2 $T.args)"
export TMPDIR=/tmp; cd /tmp
S3A_BATCH=1 S3A_STREAMS=16 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_ANY -d /tmp/pmcb -o d -- $R/oracle/_ref/ref_s3amd_tst_decode $ARGS -hyp /tmp/p.match > /tmp/pmcb.log 2>&1
f=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split('(')[0][:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k in sorted(agg, key=lambda k:-agg[k].get("SQ_BUSY_CYCLES",0))[:10]:
    a=agg[k]; n=max(cnt[k],1)
    print("%-40s launches %5d waves/l %8.0f busy/l %9.0f wavecyc/l %11.0f wait%% %4.1f vmemrd/l %8.0f vmemwr/l %7.0f valu/l %9.0f" % (k,n,a["SQ_WAVES"]/n,a["SQ_BUSY_CYCLES"]/n,a["SQ_WAVE_CYCLES"]/n,100*a["SQ_WAIT_INST_ANY"]/max(a["SQ_WAVE_CYCLES"],1),a["SQ_INSTS_VMEM_RD"]/n,a["SQ_INSTS_VMEM_WR"]/n,a["SQ_INSTS_VALU"]/n))
PY
