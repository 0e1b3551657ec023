#!/bin/bash
# round-3 iteration on the GPU box: the utterance-engine tests, then short benches with the per-kernel table
# usage: tools/r3_iter.sh NAME "<pytest selection>" [bench arg sets separated by ';']
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-it}; SEL=${2:-tests/test_gpu_uttdec.py tests/test_gpu_dropin.py}; BENCHES=${3:---lanes 512 --engines 4}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle
export S3A_ON_GPU_BOX=1
if [ -n "$SEL" ]; then
  timeout 1500 python -m pytest $SEL -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -15 $OUT/pytest.log
fi
i=0
IFS=';' read -ra SETS <<< "$BENCHES"
for B in "${SETS[@]}"; do
  i=$((i+1))
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-scoring $B > $OUT/bench_$i.json 2> $OUT/bench_$i.err; echo "bench [$B] rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$i.json"))
    print("value", d["value"], "xRT", d.get("xRT_per_gpu"), "identical", d.get("identical_to_reference"), "dev_ms", d.get("device_ms_per_step"))
    print({k: v["avg_launch_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("no json:", e); print(open("$OUT/bench_$i.err").read()[-2000:])
PY
done
