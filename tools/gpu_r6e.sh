#!/bin/bash
# round 6: ku_frames' tests with the eight-positions-per-thread scan, then same-box A/B against d_dec_scan_t (lib_scan0.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6e}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 1500 python -m pytest tests/test_gpu_kframes.py tests/test_gpu_queue.py tests/test_gpu_dropin.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
tools/ab_multi.sh $NAME 2 "scan0|lib_scan0.so|" "scan8|-|"
