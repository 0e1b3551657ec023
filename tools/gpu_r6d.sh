#!/bin/bash
# round 6: the relay's tests again (test hook fixed), the whole ku_frames / queue files with the frame as a function, then same-box A/B:
# the frame inlined into the kernel (lib_nocall.so) against the frame as a function of its own (the tree's)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6d}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 1500 python -m pytest tests/test_gpu_kframes.py tests/test_gpu_queue.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
tools/ab_multi.sh $NAME 2 "inline|lib_nocall.so|" "call|-|"
