#!/bin/bash
# round 6: the second pass's tests (library N-best), ku_frames' tests with -maxcdsenpf inside the kernel, the word level's new test, RM1 drop-in cases;
# the one-frame scoring floor (tools/score_floor.hip); same-box A/B: the library before -maxcdsenpf moved into ku_frames (lib_prev.so) against the tree's
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6g}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 3000 python -m pytest tests/test_gpu_dag.py tests/test_gpu_kframes.py tests/test_gpu_wordlevel.py "tests/test_gpu_dropin.py::test_rm1_identical_to_live_reference" -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/score_floor.hip -o /tmp/score_floor && /tmp/score_floor > $OUT/score_floor.txt 2>&1; cat $OUT/score_floor.txt
tools/ab_multi.sh $NAME 2 "prev|lib_prev.so|" "tree|-|"
