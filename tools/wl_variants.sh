#!/bin/bash
# bench with the word-level workgroup-size variants built into cmusphinx_amd/variants/
cd $(dirname $0)/..
O=gpurun_out/wlvar; mkdir -p $O
cp cmusphinx_amd/libcmusphinx_amd.so /tmp/lib_base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so; else cp cmusphinx_amd/variants/lib_$v.so cmusphinx_amd/libcmusphinx_amd.so; fi
  python bench.py --no-cpu --no-scoring --no-ps --no-wide-beam --steps 2 --warmup 1 > $O/$v.json 2> $O/$v.err || { echo "$v FAILED"; tail -3 $O/$v.err; continue; }
  python - <<PY
import json
r=json.load(open("$O/$v.json"))
k=r["kernels"]
print("$v", r["value"], "identical", r["identical_to_reference"]["hyp"], "| emit_word", k["ku_emit_word"]["avg_launch_us"], k["ku_emit_word"]["avg_launch_us_alone"], "| hist_sort", k["ku_hist_sort"]["avg_launch_us"], "| search us/frame", r["search"]["us_per_frame"], r["search"]["us_per_frame_alone"])
PY
done
cp /tmp/lib_base.so cmusphinx_amd/libcmusphinx_amd.so
