#!/bin/bash
# one utterance alone (configs[2]): where the frame's time goes -- host enqueue against device, per-kernel table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-ol}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
export S3A_ON_GPU_BOX=1
S3A_UTT_TIMES=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-scoring --lanes 1 --engines 1 --utts 4 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
grep "s3a_uttdec_decode" $OUT/bench.err | tail -6
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "xRT", d.get("xRT_per_gpu"), "identical", d.get("identical_to_reference"), "dev_ms", d.get("device_ms_per_step"))
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["us_per_frame"]): print(f"  {k:20s} {v['avg_launch_us']:8.2f} us/launch {v['us_per_frame']:8.2f} us/frame")
PY
