#!/bin/bash
# quick GPU check: gpu tests + bench line (no CPU baseline, no profiling)
cd ${GRAFT_REPO_ROOT:-.}
make -s -C oracle oracle
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu "$@"
