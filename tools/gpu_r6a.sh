#!/bin/bash
# round 6, first GPU job: the tests of what changed on the host side (queue order, score_rows_max parts / fallback, clusters at scale,
# the exchange's rendezvous), then same-box A/Bs: queue order (longest first vs in order), and what the spills cost: 256 lanes with the
# 128-VGPR build (spills) against the 256-VGPR build (lib_occ2.so: none), throughput and PMC traffic of both.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-r6a}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle >/dev/null 2>&1
export S3A_ON_GPU_BOX=1
timeout 1500 python -m pytest tests/test_gpu_kframes.py tests/test_gpu_queue.py tests/test_gpu_gather.py tests/test_gpu_bench_multirank.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
tools/ab_multi.sh $NAME 2 "inorder|-|--variant kf_queue_in_order=1" "lpt|-|"
tools/ab_multi.sh $NAME 1 "l256|-|--lanes 256" "l256occ2|lib_occ2.so|--lanes 256"
tools/pmc_case.sh $NAME l256 - --lanes 256
tools/pmc_case.sh $NAME l256occ2 lib_occ2.so --lanes 256
