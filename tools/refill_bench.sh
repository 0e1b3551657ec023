#!/bin/bash
# lane refill against lock-step groups, same batch: usage tools/refill_bench.sh NAME "<bench arg sets separated by ;>"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-rf}; BENCHES=${2:---refill 0;--refill 1}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
export S3A_ON_GPU_BOX=1
i=0
IFS=';' read -ra SETS <<< "$BENCHES"
for B in "${SETS[@]}"; do
  i=$((i+1))
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-scoring $B > $OUT/bench_$i.json 2> $OUT/bench_$i.err; echo "bench [$B] rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$i.json"))
    print("value", d["value"], "xRT", d.get("xRT_per_gpu"), "identical", d.get("identical_to_reference"), "dev_ms", d.get("device_ms_per_step"), "proj", (d.get("strong_scaling_projection") or {}).get("frames_per_sec_per_gpu"))
except Exception as e:
    print("no json:", e); print(open("$OUT/bench_$i.err").read()[-2000:])
PY
done
