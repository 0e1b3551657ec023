#!/bin/bash
# round 6: the second pass's tests with the library's N-best (tests/test_gpu_dag.py), the ku_frames / queue files once more, smoke
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6f}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 2400 python -m pytest tests/test_gpu_dag.py tests/test_gpu_kframes.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 900 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
