#!/bin/bash
# One optimisation iteration on the GPU box: the word-level / utterance tests, the hub4 task (1 and 32... lanes) with the
# word-level phase ticks, a short bench.  usage: tools/gpu_iter.sh NAME [bench lanes...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-it}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle
export S3A_ON_GPU_BOX=1
python -m pytest tests/test_gpu_wordlevel.py tests/test_gpu_uttdec.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
S3A_UTT_TICKS=1 tools/utt_task.sh hub4 16 600 "1 16" > $OUT/hub4.txt 2>&1
grep "word-level phase" gpurun_out/utt/task_hub4.u1.log | cut -c24-200 > $OUT/ticks_u1.txt
grep "word-level phase" gpurun_out/utt/task_hub4.u16.log | cut -c24-200 > $OUT/ticks_u16.txt
S3A_UTT=1 SKIP_REF=1 tools/prof_task.sh ${NAME}_hub4_utt1 hub4 4 600 > $OUT/hub4_utt1_prof.txt 2>&1
for L in ${@:-32}; do
  python bench.py --steps 4 --warmup 1 --no-cpu --no-scoring --lanes $L --engines 1 > $OUT/bench_l$L.json 2> $OUT/bench_l$L.err; echo "bench lanes=$L rc=$?"
  python - <<PY
import json
d = json.load(open("$OUT/bench_l$L.json"))
print("lanes $L value", d["value"], "xRT", d.get("xRT_per_gpu"), "identical", d.get("identical_to_reference"))
print({k: v["avg_launch_us"] for k, v in d.get("kernels", {}).items()})
PY
done
tail -3 $OUT/pytest.log; grep -v histogram $OUT/hub4.txt; cat $OUT/ticks_u1.txt; head -16 $OUT/hub4_utt1_prof.txt
