#!/bin/bash
# Validation of a state on the GPU box without the profiling passes: all gpu tests, smoke, the default bench.
# usage: tools/gpu_check.sh NAME [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=${1:-chk}; shift
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
make -s -C oracle oracle
export S3A_ON_GPU_BOX=1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
( time timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?" >> $OUT/bench.err
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -3 $OUT/bench.err; cat $OUT/bench.time; head -c 700 $OUT/bench.json; echo
