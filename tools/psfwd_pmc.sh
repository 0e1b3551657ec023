#!/bin/bash
# HBM-side bytes of the pocketsphinx first pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): the queue of NU utterances over L lanes
cd $(dirname $0)/..
R=$(pwd); O=$R/${1:-gpurun_out/pspmc}; NU=${2:-256}; L=${3:-128}
mkdir -p $O
D=/tmp/pstask
[ -f $D/ctl ] || python -m cmusphinx_amd.synth_task hub4 $D n_utt=1024 n_frames=1000 sorted_names=1 > $O/task.txt 2>&1
PSA="-mdef $D/mdef -mean $D/means -var $D/variances -mixw $D/mixture_weights -tmat $D/transition_matrices -senmgau .cont. -dict $D/dict -fdict $D/fillerdict -lm $D/lm.arpa -feat 1s_c -ceplen 39 -cmn none -agc none -varnorm no -cepdir $D/feat -cepext .mfc -fwdflat no -bestpath no"
head -$NU $D/ctl > /tmp/ctlpmc
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/$C -o p -- $R/oracle/_ref/ref_ps_amdfwd $PSA -ctl /tmp/ctlpmc -fresh yes -batch $L -queue yes -hyp /tmp/x.match > $O/$C.log 2>&1; echo "$C rc=$?"
  grep "ms on the device" $O/$C.log | sed 's/^.*batch of/batch of/' | head -1
done
python - <<PY
import csv, glob, json
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of ref_ps_amdfwd: $NU utterances as one queue over $L lanes, hub4-shaped task", "kernels": {}}
frames = 0
for l in open("$O/FETCH_SIZE.log", errors="ignore"):
    if "ms on the device" in l: frames = int(l.split("utterances,")[1].split("frames")[0])
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                k = r["Kernel_Name"].replace("void ", "").split("(")[0]
                e = out["kernels"].setdefault(k, {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
                e[C + "_KB"] += float(r["Counter_Value"])
                if C == "FETCH_SIZE": e["launches"] += 1
out["frames"] = frames
for k, e in out["kernels"].items():
    e["bytes_per_frame"] = round((e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024 / frames, 1) if frames else None
json.dump(out, open("$O/pmc_ps.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
