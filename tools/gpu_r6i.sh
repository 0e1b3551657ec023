#!/bin/bash
# round 6: the whole GPU suite, smoke, then the full bench line (as the driver runs it)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6i}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 3000 python -m pytest tests/ -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -8 $OUT/pytest_gpu.txt
timeout 900 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "identical", d["identical_to_reference"], "roofline", d["roofline"].get("frac"), d["roofline"].get("traffic"))
print("single", d.get("single_utterance")); print("proj", d.get("strong_scaling_projection"))
w = d.get("wide_beam", {}); print("wide", w.get("frames_per_sec"), w.get("identical_to_reference"), w.get("lanes_128"))
print("ps", {k: d.get("ps_fwdtree", {}).get(k) for k in ("frames_per_sec", "identical_to_pocketsphinx")}); print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
