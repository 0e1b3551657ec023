#!/bin/bash
# round 6: ku_frames with the wide-beam word level inside (tests), the N-best test on the hub4-shaped task again; same-box A/B of the bench's task
# (lib_prev.so = before) and the wide-beam leg both ways (64 and 128 lanes: launches = S3A_UTT_PERSIST=-1 against ku_frames)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6h}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 3000 python -m pytest tests/test_gpu_kframes.py "tests/test_gpu_dag.py::test_nbest_lists_on_the_hub4_shaped_task" "tests/test_gpu_dropin.py::test_wsj_shaped_wide_beam_decode_matches_reference" -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
tools/ab_multi.sh $NAME 2 "prev|lib_prev.so|" "tree|-|"
for lanes in 64 128; do
  for p in -1 0; do
    S3A_UTT_PERSIST=$p timeout 1200 python bench.py --steps 1 --warmup 0 --no-cpu --no-scoring --no-ps --utts 64 --lanes 64 --wide-lanes $lanes > $OUT/wide_${lanes}_p$p.json 2> $OUT/wide_${lanes}_p$p.err
    python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/wide_${lanes}_p$p.json").read().strip().splitlines()[-1])
    print("wide beam lanes $lanes persist $p:", json.dumps(d.get("wide_beam"))[:700])
except Exception as e:
    print("wide $lanes $p FAILED", e); print(open("$OUT/wide_${lanes}_p$p.err").read()[-800:])
PY
  done
done
