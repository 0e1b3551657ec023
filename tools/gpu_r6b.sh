#!/bin/bash
# round 6, second GPU job: the relay's tests, then same-box A/B: relay on / off (and the lanes' busy times of each)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6b}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 1500 python -m pytest tests/test_gpu_kframes.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
tools/ab_multi.sh $NAME 2 "norelay|-|--variant kf_no_relay=1" "relay|-|"
python3 - <<PY
import json
for l in ("norelay", "relay"):
    d = json.loads(open("$OUT/%s_1.json" % l).read().strip().splitlines()[-1])
    print(l, d.get("lanes_busy_ms"), d.get("kernels"))
PY
